"""How the difference between the device's film and the oracle's shrinks as both converge.  Both sides draw the same
random numbers; a path parts from its twin where a last-place difference between the device's and the host's sinf / cosf /
powf flips one of its decisions, and from there the two are independent samples of the same pixel -- so the relative L2
distance of the films falls like Monte Carlo noise restricted to the parted paths.  Scenes: the Cornell box of
configuration 1 (512 x 512 in the configuration, 256 x 256 here; blinn-phong, max depth 5), cornell-box-glass of configuration 2
(max depth 8) and the same with unbounded depth; sums of samples 1 .. N on both sides for N = 16 ... 4096.
and two synthetic scenes (heightfield of 1M triangles, soup of 100k) under the benchmark's camera.
usage: python scripts/render_convergence.py [largest N = 4096] [film side = 256] [only scenes whose name contains this]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import load_cpt
from helpers import cornell, OracleScene

cpt = load_cpt()
top = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
side = int(sys.argv[2]) if len(sys.argv) > 2 else 256
out = {"film": [side, side], "scenes": {}}
from helpers import synthetic_scene
BENCH_CAMERA = {"pos": np.array([0.5, 0.3, 0.1], np.float32), "look_at": np.array([0.5, 0.0, 0.45], np.float32), "fov": 40.0}
cases = [("cornell-box, max depth 5", False, 5), ("cornell-box-glass, max depth 8", True, 8), ("cornell-box-glass, unbounded depth", True, -1),
         ("heightfield 1M triangles, max depth 5", "heightfield", 5), ("soup 100k triangles, max depth 5", "soup", 5)]
if len(sys.argv) > 3:
    cases = [c for c in cases if sys.argv[3] in c[0]]
for name, glass, depth in cases:
    if isinstance(glass, str):
        objs, cam = synthetic_scene(glass, 1_000_000 if glass == "heightfield" else 100_000)
        cam = dict(cam, **BENCH_CAMERA)
    else:
        objs, cam = cornell(glass=glass, bsdf_type=2)
    osc = OracleScene(objs, cam)
    sc = cpt.Scene()
    for o in objs:
        sc.add_object(o["name"], o["positions"], o["normals"], o["texcoords"], o["indices"], o["material"])
    sc.set_camera(cam["pos"], cam["look_at"], cam["up"], cam["fov"], side, side)
    tr = cpt.Tracer(sc, side, side, max_depth=depth)
    tr.build_buffers()
    tr.reset()
    ref = np.zeros((side, side, 4), np.float32)
    rows, done, n = [], 0, 16
    while n <= top:
        tr.render_samples(done + 1, n - done, True)
        tr.synchronize()
        block, _ = osc.render(side, side, done + 1, n + 1, depth, sum_mode=1, nthreads=16)
        ref = ref + block                       # sums of blocks: the oracle's own block sums added in float32, as the device adds launches
        done = n
        img = tr.film()
        a, b = img[..., :3].astype(np.float64) / n, ref[..., :3].astype(np.float64) / n
        lum = np.maximum(np.abs(b).max(axis=-1), 1e-3)
        rows.append({"spp": n, "rel_l2": float(np.sqrt(((a - b) ** 2).sum() / (b ** 2).sum())),
                     "mean_rgb_relative_difference": [float(x) for x in np.abs(a.mean(axis=(0, 1)) - b.mean(axis=(0, 1))) / b.mean(axis=(0, 1))],
                     "pixels_off_by_more_than_1e-3": float((np.abs(a - b).max(axis=-1) > 1e-3 * lum).mean()),
                     "alpha_equal": bool(np.array_equal(img[..., 3], ref[..., 3]))})
        print(name, rows[-1], file=sys.stderr, flush=True)
        n *= 4
    out["scenes"][name] = rows
    del tr, sc, osc
print(json.dumps(out))
