"""Parity sweep beyond the test suite's fixed cases: seeded random scenes (triangle soup and heightfield of random size,
random camera), camera rays and rays from random points inside the scene, first hits of both node-record kinds and of
both query kernels compared with the oracle bit for bit (instance, primitive, distance, barycentrics), any-hit answers
compared exactly.  Runs until the time budget is used and prints one JSON object.
usage: python scripts/parity_sweep.py [seconds=240] [first seed=1000]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import load_cpt
from helpers import synthetic_scene, OracleScene

cpt = load_cpt()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
t_begin = time.time()
out = {"scenes": 0, "rays_compared": 0, "shadow_rays_compared": 0, "mismatching_rays": 0, "mismatching_shadow_rays": 0, "cases": []}
while time.time() - t_begin < budget:
    rng = np.random.Generator(np.random.PCG64(seed))
    kind = "soup" if seed % 2 else "heightfield"
    ntris = int(10 ** rng.uniform(2.0, 5.7))
    objs, cam = synthetic_scene(kind, ntris, seed)
    cam = dict(cam)
    cam["pos"] = np.array([rng.uniform(0.1, 0.9), rng.uniform(0.1, 0.6), rng.uniform(0.05, 0.9)], dtype=np.float32)
    cam["look_at"] = np.array([rng.uniform(0.2, 0.8), rng.uniform(-0.2, 0.3), rng.uniform(0.2, 0.8)], dtype=np.float32)
    cam["fov"] = float(rng.uniform(25.0, 80.0))
    W, H = 512, 288
    osc = OracleScene(objs, cam)
    sc = cpt.Scene()
    for o in objs:
        sc.add_object(o["name"], o["positions"], o["normals"], o["texcoords"], o["indices"], o["material"])
    sc.set_camera(cam["pos"], cam["look_at"], cam["up"], cam["fov"], W, H)
    n_in = 150000
    o = rng.random((n_in, 3)).astype(np.float32)
    d = rng.standard_normal((n_in, 3)).astype(np.float32)
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    d[:64, 0] = 0.0                                   # some axis-parallel directions
    d[64:128, 1] = -0.0
    inside = np.zeros((n_in, 8), np.float32)
    inside[:, 0:3] = o; inside[:, 3] = 1e-4; inside[:, 4:7] = d; inside[:, 7] = np.finfo(np.float32).max
    inside[n_in // 2:, 7] = rng.uniform(0.01, 0.5, n_in - n_in // 2).astype(np.float32)      # short rays
    rays = np.concatenate([osc.camera_rays(W, H, 1 + seed % 5), inside])
    o_inst, o_prim, o_t, o_uv, _ = osc.first_hit(rays, nthreads=16)
    o_occ = osc.occlude(inside)
    bad = 0
    for node_records in (1, 0):
        tr = cpt.Tracer(sc, W, H, max_depth=5)
        tr.set_node_records(node_records)
        tr.build_buffers()
        for query_kernel in (0, 1):
            tr.set_query_kernel(query_kernel)
            g_inst, g_prim, g_t, g_uv = tr.first_hit(rays)
            differ = (g_inst != o_inst) | (g_prim != o_prim) | (g_t.view(np.uint32) != o_t.view(np.uint32)) | np.any(g_uv != o_uv, axis=-1)
            bad += int(differ.sum())
            out["rays_compared"] += len(rays)
        g_occ = tr.occlude(inside)
        out["mismatching_shadow_rays"] += int((np.asarray(g_occ) != np.asarray(o_occ)).sum())
        out["shadow_rays_compared"] += n_in
        del tr
    out["mismatching_rays"] += bad
    out["scenes"] += 1
    out["cases"].append({"seed": seed, "kind": kind, "triangles": sum(len(ob["indices"]) // 3 for ob in objs), "hit_fraction": round(float((o_inst != 0xFFFFFFFF).mean()), 3), "mismatches": bad})
    print(f"seed {seed} {kind} {out['cases'][-1]['triangles']} triangles: {bad} mismatches, {time.time() - t_begin:.0f} s", file=sys.stderr, flush=True)
    seed += 1
out["seconds"] = round(time.time() - t_begin, 1)
print(json.dumps(out))
