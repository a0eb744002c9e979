"""Binary node records against two-level ones (exact traversal) on one scene: first-hit results compared bit for bit, rates.
usage: python scripts/compare_wide.py <cornell|heightfield|soup> [ntris width height]"""
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

def make_scene():
    if kind == "cornell":
        d = os.path.join(SCENES, "cornell-box-glass")
        return cpt.Scene().load_obj(os.path.join(d, "cornell-box.obj"), 2).load_json(os.path.join(d, "cornell-box.json")), np.float32([-1, 0, -1]), np.float32([1, 2, 1])
    objs, cam = synthetic_scene(kind, ntris)
    sc = cpt.Scene()
    for o in objs:
        sc.add_object(o["name"], o["positions"], o["normals"], o["texcoords"], o["indices"], o["material"])
    sc.set_camera([0.5, 0.3, 0.1], [0.5, 0.0, 0.45], [0, 1, 0], 40.0, W, H)
    return sc, np.float32([0, -0.2, 0]), np.float32([1, 1.2, 1])

sc, lo, hi = make_scene()
rng = np.random.Generator(np.random.PCG64(5))
m = 1 << 21
o = (lo + (hi - lo) * rng.random((m, 3))).astype(np.float32)
dd = rng.standard_normal((m, 3)).astype(np.float32)
dd /= np.linalg.norm(dd, axis=1, keepdims=True)
rays = np.zeros((m, 8), np.float32)
rays[:, 0:3] = o; rays[:, 3] = 1e-4; rays[:, 4:7] = dd; rays[:, 7] = np.finfo(np.float32).max
# a share of axis-parallel directions (zeros of both signs) and of origins on box planes
rays[: m // 64, 4:7] = np.float32([[0.0, -1.0, 0.0]]); rays[m // 64 : m // 32, 4:7] = np.float32([[-0.0, 0.0, 1.0]])

out = {"scene": kind, "ntris": ntris}
results = {}
for tag in ("binary", "wide"):
    tr = cpt.Tracer(sc, W, H, max_depth=5)
    tr.set_node_records(1 if tag == "wide" else 0)
    tr.set_integrator(3)
    tr.build_buffers()
    n = W * H
    d_cam = cpt.DeviceArray((n, 8), np.float32)
    cpt.check(cpt.lib.cpt_tracer_camera_rays(tr.h, W, H, 2, d_cam.ptr))
    d_int = cpt.DeviceArray.from_host(rays)
    res = {}
    for name, d_rays, count in (("camera_rays", d_cam, n), ("interior_rays", d_int, m)):
        d_inst, d_prim, d_t = cpt.DeviceArray((count,), np.uint32), cpt.DeviceArray((count,), np.uint32), cpt.DeviceArray((count,), np.float32)
        d_uv = cpt.DeviceArray((count, 2), np.float32)
        cpt.check(cpt.lib.cpt_tracer_first_hit(tr.h, d_rays.ptr, count, d_inst.ptr, d_prim.ptr, d_t.ptr, d_uv.ptr))
        tr.synchronize(); tr.reset_counters(); tr.synchronize()
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            cpt.check(cpt.lib.cpt_tracer_first_hit(tr.h, d_rays.ptr, count, d_inst.ptr, d_prim.ptr, d_t.ptr, d_uv.ptr))
        tr.synchronize()
        dt = (time.perf_counter() - t0) / reps
        c = tr.counters()
        res[name] = {"mrays_per_s": round(count / dt / 1e6, 1), "records_per_ray": round(c["node_visits"] / c["rays"], 2), "triangles_per_ray": round(c["tri_tests"] / c["rays"], 2),
                     "stack_overflows": tr.stack_overflows()}
        d_occ = cpt.DeviceArray((count,), np.uint8)
        cpt.check(cpt.lib.cpt_tracer_occlude(tr.h, d_rays.ptr, count, d_occ.ptr))
        tr.synchronize()
        results[(tag, name)] = (d_inst.to_host(), d_prim.to_host(), d_t.to_host().view(np.uint32), d_uv.to_host().view(np.uint32), d_occ.to_host())
    out[tag] = res
    del tr
for name in ("camera_rays", "interior_rays"):
    a, b = results[("binary", name)], results[("wide", name)]
    out[name + "_mismatches"] = {"inst": int((a[0] != b[0]).sum()), "prim": int((a[1] != b[1]).sum()), "t": int((a[2] != b[2]).sum()), "uv": int((a[3] != b[3]).any(axis=1).sum()),
                                 "occluded": int((a[4] != b[4]).sum())}
    out[name + "_speedup"] = round(out["wide"][name]["mrays_per_s"] / out["binary"][name]["mrays_per_s"], 3)
print(json.dumps(out), flush=True)
