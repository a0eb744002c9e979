"""The two kernels of cpt_tracer_first_hit on one scene: one ray per lane for the whole launch against the persistent kernel
that hands a lane the next ray when its own is finished; exact and ordered traversal, camera rays and rays from inside the
scene; results compared bit for bit, rates.
usage: python scripts/compare_query.py <cornell|heightfield|soup> [ntris width height]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import load_cpt
from helpers import synthetic_scene, SCENES

cpt = load_cpt()
kind = sys.argv[1] if len(sys.argv) > 1 else "heightfield"
ntris = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
W = int(sys.argv[3]) if len(sys.argv) > 3 else 1920
H = int(sys.argv[4]) if len(sys.argv) > 4 else 1080
if kind == "cornell":
    d = os.path.join(SCENES, "cornell-box-glass")
    sc, lo, hi = cpt.Scene().load_obj(os.path.join(d, "cornell-box.obj"), 2).load_json(os.path.join(d, "cornell-box.json")), np.float32([-1, 0, -1]), np.float32([1, 2, 1])
else:
    objs, cam = synthetic_scene(kind, ntris)
    sc = cpt.Scene()
    for o in objs:
        sc.add_object(o["name"], o["positions"], o["normals"], o["texcoords"], o["indices"], o["material"])
    sc.set_camera([0.5, 0.3, 0.1], [0.5, 0.0, 0.45], [0, 1, 0], 40.0, W, H)
    lo, hi = np.float32([0, -0.2, 0]), np.float32([1, 1.2, 1])
rng = np.random.Generator(np.random.PCG64(5))
m = (1 << 21) + 37                                       # not a multiple of a segment
o = (lo + (hi - lo) * rng.random((m, 3))).astype(np.float32)
dd = rng.standard_normal((m, 3)).astype(np.float32)
dd /= np.linalg.norm(dd, axis=1, keepdims=True)
rays = np.zeros((m, 8), np.float32)
rays[:, 0:3] = o; rays[:, 3] = 1e-4; rays[:, 4:7] = dd; rays[:, 7] = np.finfo(np.float32).max
rays[: m // 64, 4:7] = np.float32([[0.0, -1.0, 0.0]]); rays[m // 64 : m // 32, 4:7] = np.float32([[-0.0, 0.0, 1.0]])

tr = cpt.Tracer(sc, W, H, max_depth=5)
tr.build_buffers()
n = W * H
d_cam = cpt.DeviceArray((n, 8), np.float32)
cpt.check(cpt.lib.cpt_tracer_camera_rays(tr.h, W, H, 2, d_cam.ptr))
d_int = cpt.DeviceArray.from_host(rays)
out = {"scene": kind, "ntris": ntris, "film": [W, H]}
for trav, tname in ((0, "exact"), (1, "fast")):
    tr.set_traversal(trav)
    for name, d_rays, count in (("camera_rays", d_cam, n), ("interior_rays", d_int, m)):
        got = {}
        row = {}
        for mode, mname in ((0, "one_ray_per_lane"), (1, "persistent")):
            tr.set_query_kernel(mode)
            d_inst, d_prim, d_t = cpt.DeviceArray((count,), np.uint32), cpt.DeviceArray((count,), np.uint32), cpt.DeviceArray((count,), np.float32)
            d_uv = cpt.DeviceArray((count, 2), np.float32)
            for a in (d_inst, d_prim, d_t, d_uv):
                a.zero()
            cpt.check(cpt.lib.cpt_tracer_first_hit(tr.h, d_rays.ptr, count, d_inst.ptr, d_prim.ptr, d_t.ptr, d_uv.ptr))
            tr.synchronize(); tr.reset_counters(); tr.synchronize()
            reps = 10
            tr.timer_begin()
            for _ in range(reps):
                cpt.check(cpt.lib.cpt_tracer_first_hit(tr.h, d_rays.ptr, count, d_inst.ptr, d_prim.ptr, d_t.ptr, d_uv.ptr))
            ms = tr.timer_end() / reps
            c = tr.counters()
            row[mname] = {"mrays_per_s": round(count / ms / 1e3, 1), "records_per_ray": round(c["node_visits"] / c["rays"], 2), "rays_counted": c["rays"] // reps}
            got[mode] = (d_inst.to_host(), d_prim.to_host(), d_t.to_host().view(np.uint32), d_uv.to_host().view(np.uint32))
        a, b = got[0], got[1]
        row["results_that_differ"] = int(((a[0] != b[0]) | (a[1] != b[1]) | (a[2] != b[2]) | (a[3] != b[3]).any(axis=1)).sum())
        row["speedup"] = round(row["persistent"]["mrays_per_s"] / row["one_ray_per_lane"]["mrays_per_s"], 3)
        out[f"{tname}/{name}"] = row
print(json.dumps(out), flush=True)
