"""LBVH build timings (median of 10 after 2 warm-ups, buffers pre-allocated) with the stage split.
usage: python scripts/bench_lbvh.py [n ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from conftest import load_cpt
cpt = load_cpt()
for n in [int(a) for a in sys.argv[1:]] or [1_000_000, 10_000_000]:
    print(n, bench.lbvh_timings(cpt, n), flush=True)
    print(n, "onesweep", bench.lbvh_timings(cpt, n, sort_variant=1), flush=True)
