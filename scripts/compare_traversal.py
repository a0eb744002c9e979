"""Exact against fast traversal on one scene: mismatching first hits (ids, distances), node records per ray, Mrays/s
of the batch query, Mpaths/s of the bounce loop.
usage: python scripts/compare_traversal.py <cornell|heightfield|soup> [ntris width height spp depth two_level_records]
Prints one JSON line."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import load_cpt
from helpers import synthetic_scene, SCENES

cpt = load_cpt()
kind = sys.argv[1] if len(sys.argv) > 1 else "cornell"
ntris = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
W = int(sys.argv[3]) if len(sys.argv) > 3 else 1920
H = int(sys.argv[4]) if len(sys.argv) > 4 else 1080
spp = int(sys.argv[5]) if len(sys.argv) > 5 else 16
depth = int(sys.argv[6]) if len(sys.argv) > 6 else 5
two_level = int(sys.argv[7]) if len(sys.argv) > 7 else 1

if kind == "cornell":
    d = os.path.join(SCENES, "cornell-box-glass")
    sc = cpt.Scene().load_obj(os.path.join(d, "cornell-box.obj"), 2).load_json(os.path.join(d, "cornell-box.json"))
    lo, hi = np.float32([-1, 0, -1]), np.float32([1, 2, 1])
    label = "cornell-box-glass"
else:
    objs, cam = synthetic_scene(kind, ntris)
    sc = cpt.Scene()
    for o in objs:
        sc.add_object(o["name"], o["positions"], o["normals"], o["texcoords"], o["indices"], o["material"])
    sc.set_camera([0.5, 0.3, 0.1], [0.5, 0.0, 0.45], [0, 1, 0], 40.0, W, H)
    lo, hi = np.float32([0, -0.2, 0]), np.float32([1, 1.2, 1])
    label = f"{kind} {len(objs[0]['indices']) // 3}"
tr = cpt.Tracer(sc, W, H, max_depth=depth)
tr.set_integrator(3)
tr.set_node_records(two_level)
tr.build_buffers()

n = W * H
d_cam = cpt.DeviceArray((n, 8), np.float32)
cpt.check(cpt.lib.cpt_tracer_camera_rays(tr.h, W, H, 2, d_cam.ptr))
rng = np.random.Generator(np.random.PCG64(5))
m = 1 << 21
o = (lo + (hi - lo) * rng.random((m, 3))).astype(np.float32)
dd = rng.standard_normal((m, 3)).astype(np.float32)
dd /= np.linalg.norm(dd, axis=1, keepdims=True)
rays = np.zeros((m, 8), np.float32)
rays[:, 0:3] = o; rays[:, 3] = 1e-4; rays[:, 4:7] = dd; rays[:, 7] = np.finfo(np.float32).max
d_int = cpt.DeviceArray.from_host(rays)
del rays, o, dd

out = {"scene": label, "film": [W, H], "depth": depth, "two_level_records": bool(two_level)}
for name, d_rays, count in (("camera_rays", d_cam, n), ("interior_rays", d_int, m)):
    res = {}
    hits = {}
    for mode, tag in ((0, "exact"), (1, "fast")):
        tr.set_traversal(mode)
        d_inst, d_prim, d_t = cpt.DeviceArray((count,), np.uint32), cpt.DeviceArray((count,), np.uint32), cpt.DeviceArray((count,), np.float32)
        cpt.check(cpt.lib.cpt_tracer_first_hit(tr.h, d_rays.ptr, count, d_inst.ptr, d_prim.ptr, d_t.ptr, None))
        tr.synchronize(); tr.reset_counters(); tr.synchronize()
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            cpt.check(cpt.lib.cpt_tracer_first_hit(tr.h, d_rays.ptr, count, d_inst.ptr, d_prim.ptr, d_t.ptr, None))
        tr.synchronize()
        dt = (time.perf_counter() - t0) / reps
        c = tr.counters()
        hits[tag] = (d_inst.to_host(), d_prim.to_host(), d_t.to_host())
        res[tag] = {"mrays_per_s": round(count / dt / 1e6, 1), "node_records_per_ray": round(c["node_visits"] / c["rays"], 2),
                    "triangles_per_ray": round(c["tri_tests"] / c["rays"], 2), "stack_overflows": tr.stack_overflows()}
    e, f = hits["exact"], hits["fast"]
    ids = (e[0] != f[0]) | (e[1] != f[1])
    res["rays"] = count
    res["id_mismatches"] = int(ids.sum())
    res["distance_mismatches"] = int((e[2].view(np.uint32) != f[2].view(np.uint32)).sum())
    res["fast_hit_farther"] = int((f[2] > e[2]).sum())
    res["fast_hit_nearer"] = int((f[2] < e[2]).sum())
    res["speedup"] = round(res["fast"]["mrays_per_s"] / res["exact"]["mrays_per_s"], 3)
    out[name] = res
    dump = os.environ.get("CPT_DUMP_MISMATCHES")
    if dump and ids.any():
        idx = np.nonzero(ids)[0][:4096]
        np.savez(f"{dump}_{name}.npz", rays=d_rays.to_host()[idx], e_inst=e[0][idx], e_prim=e[1][idx], e_t=e[2][idx], f_inst=f[0][idx], f_prim=f[1][idx], f_t=f[2][idx])

paths = {}
films = {}
for mode, tag in ((0, "exact"), (1, "fast")):
    tr.set_traversal(mode)
    tr.reset()
    tr.update(2); tr.synchronize(); tr.reset()
    tr.reset_counters()
    ms = tr.update_timed(spp)
    c = tr.counters()
    films[tag] = tr.film()
    paths[tag] = {"mpaths_per_s": round(W * H * spp / ms / 1e3, 1), "mrays_per_s": round(c["rays"] / ms / 1e3, 1),
                  "node_records_per_ray": round(c["node_visits"] / c["rays"], 2), "ms_per_sample": round(ms / spp, 4)}
a, b = films["exact"][..., :3], films["fast"][..., :3]
paths["speedup"] = round(paths["fast"]["mpaths_per_s"] / paths["exact"]["mpaths_per_s"], 3)
paths["film_rel_l2"] = float(np.sqrt(((a - b) ** 2).sum() / max((a ** 2).sum(), 1e-30)))
paths["pixels_differing"] = int(np.any(a != b, axis=-1).sum())
paths["spp"] = spp
out["path_tracing"] = paths
print(json.dumps(out), flush=True)
