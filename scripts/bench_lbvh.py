"""LBVH build timings (median of 10 after 2 warm-ups, buffers pre-allocated) with the stage split.
usage: python scripts/bench_lbvh.py [soup|heightfield] [n ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from conftest import load_cpt
cpt = load_cpt()
args = sys.argv[1:]
kind = args.pop(0) if args and not args[0].isdigit() else "soup"
for n in [int(a) for a in args] or [1_000_000, 10_000_000]:
    print(n, kind, bench.lbvh_timings(cpt, n, kind=kind), flush=True)
    print(n, kind, "onesweep", bench.lbvh_timings(cpt, n, sort_variant=1, kind=kind), flush=True)
