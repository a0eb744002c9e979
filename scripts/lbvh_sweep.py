"""LBVH parity sweep beyond the test suite's fixed sizes: seeded random inputs -- sizes around the boundaries of the
build's tiles and groups (multiples of 512, 1024, 2048, 4096 and of the 3,000,000-key switch between the two launch
shapes, each with one key less and one more) and log-uniform sizes in between; leaf boxes uniform, clustered (many equal
Morton codes), flat along an axis, or all equal -- built by cpt_build_accel and by the oracle and compared bit for bit
(sorted keys, nodes, parent links, every box).  Runs until the time budget is used and prints one JSON object.
usage: python scripts/lbvh_sweep.py [seconds=150] [first seed=1]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import load_cpt
from helpers import oracle

cpt = load_cpt()
L = oracle()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 150.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
edges = [512, 1024, 2048, 4096, 8192, 65536, 131072, 1 << 20, 3_000_000, 3_001_344]
t_begin = time.time()
out = {"builds": 0, "primitives": 0, "mismatching_builds": 0, "cases": []}
while time.time() - t_begin < budget:
    rng = np.random.Generator(np.random.PCG64(seed))
    if seed % 3 == 0:
        e = edges[(seed // 3) % len(edges)]
        n = max(1, e * int(rng.integers(1, 4 if e < 1_000_000 else 2)) + int(rng.integers(-1, 2)))
    else:
        n = int(10 ** rng.uniform(0.0, 6.3))
    shape = ("uniform", "clustered", "flat", "equal")[int(rng.integers(0, 4))] if n < 1_500_000 else "uniform"
    c = rng.random((n, 3)).astype(np.float32)
    if shape == "clustered":
        c = (np.floor(c * 6.0) / 6.0 + 1e-3 * rng.random((n, 3))).astype(np.float32)
    elif shape == "flat":
        c[:, int(rng.integers(0, 3))] = np.float32(0.25)
    elif shape == "equal":
        c[:] = c[0]
    ext = (rng.random((n, 3)) * (0.0 if shape in ("flat", "equal") and seed % 2 else 0.01)).astype(np.float32)
    boxes = np.ones((n, 8), np.float32)
    boxes[:, 0:3] = c - ext
    boxes[:, 4:7] = c + ext
    merged = np.ones(8, np.float32)
    merged[0:3] = boxes[:, 0:3].min(axis=0)
    merged[4:7] = boxes[:, 4:7].max(axis=0)
    nn = 2 * n - 1
    o_boxes = np.zeros((nn, 8), np.float32); o_boxes[n - 1:] = boxes
    o_nodes = np.zeros((nn, 2), np.uint32); o_keys = np.zeros(n, np.uint64); o_par = np.zeros(nn, np.uint32)
    L.orc_build_accel(o_nodes.reshape(-1), o_boxes.reshape(-1), np.ascontiguousarray(merged), n, o_keys.ctypes.data, o_par.ctypes.data)
    g_nodes, g_boxes, g_keys, g_par, _ = cpt.build_accel(boxes, merged)
    same = (np.array_equal(g_keys, o_keys) and np.array_equal(g_nodes, o_nodes) and np.array_equal(g_par[1:], o_par[1:])
            and np.array_equal(g_boxes.view(np.uint32), o_boxes.view(np.uint32)))
    out["builds"] += 1
    out["primitives"] += n
    out["mismatching_builds"] += 0 if same else 1
    if not same or len(out["cases"]) < 40:
        out["cases"].append({"seed": seed, "n": n, "boxes": shape, "same": bool(same)})
    print(f"seed {seed} n {n} {shape}: {'same' if same else 'DIFFERENT'}, {time.time() - t_begin:.0f} s", file=sys.stderr, flush=True)
    seed += 1
out["seconds"] = round(time.time() - t_begin, 1)
print(json.dumps(out))
