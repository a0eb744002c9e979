"""One case of scripts/render_sweep.py (by seed) rendered by every integrator of the device and by the oracle: ray counts,
relative L2 error against the oracle, and whether the integrators' films are equal bit for bit.
usage: python scripts/render_case.py <seed> [spp override]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import load_cpt
from helpers import cornell, OracleScene

cpt = load_cpt()
seed = int(sys.argv[1])
rng = np.random.Generator(np.random.PCG64(seed))
glass = bool(rng.integers(0, 2)); bsdf = int(rng.integers(0, 4)); depth = int(rng.choice([1, 2, 3, 5, 8, -1])); integrator = int(rng.choice([0, 1, 2, 4, 5]))
W, H = int(rng.integers(33, 161)), int(rng.integers(33, 121)); spp = int(rng.integers(4, 33)); first = int(rng.integers(1, 50))
objs, cam = cornell(glass=glass, bsdf_type=bsdf)
cam = dict(cam)
cam["pos"] = (np.asarray(cam["pos"], np.float32) + rng.uniform(-0.15, 0.15, 3).astype(np.float32)).astype(np.float32)
moved = bool(rng.integers(0, 2))
if moved:
    for o in objs:
        if any(k in o["name"].lower() for k in ("tall", "short", "light")):
            o["trs"] = ([float(x) for x in rng.uniform(-0.08, 0.08, 3)], [float(x) for x in rng.uniform(-25.0, 25.0, 3)], [float(x) for x in rng.uniform(0.8, 1.2, 3)])
if len(sys.argv) > 2:
    spp = int(sys.argv[2])
print(dict(glass=glass, bsdf=bsdf, depth=depth, integrator=integrator, film=[W, H], spp=spp, first=first, moved=moved), flush=True)
osc = OracleScene(objs, cam)
ref, ocnt = osc.render(W, H, first, first + spp, depth, sum_mode=1, nthreads=16)
print("oracle rays", int(ocnt[0]), flush=True)
sc = cpt.Scene()
for o in objs:
    sc.add_object(o["name"], o["positions"], o["normals"], o["texcoords"], o["indices"], o["material"])
    if "trs" in o:
        sc.set_trs(sc.num_objects() - 1, *o["trs"])
sc.set_camera(cam["pos"], cam["look_at"], cam["up"], cam["fov"], W, H)
films = {}
for mode in (0, 1, 2, 4, 5):
    tr = cpt.Tracer(sc, W, H, max_depth=depth)
    tr.set_integrator(mode)
    tr.build_buffers()
    tr.reset()
    tr.render_samples(first, spp, True)
    tr.synchronize()
    img = tr.film()
    c = tr.counters()
    a, b = img[..., :3].astype(np.float64), ref[..., :3].astype(np.float64)
    films[mode] = img
    print(f"integrator {mode}: rays {c['rays']}, paths {c['paths']}, rel L2 {np.sqrt(((a - b) ** 2).sum() / (b ** 2).sum()):.3e}, pixels that differ from the oracle {np.any(a != b, axis=-1).mean():.4f}, "
          f"equal to integrator 0 bit for bit: {np.array_equal(films[0].view(np.uint32), img.view(np.uint32))}, stack overflows {c.get('stack_overflows')}", flush=True)
    del tr
# per-pixel: one sample at a time, where do the device and the oracle part?
tr = cpt.Tracer(sc, W, H, max_depth=depth)
tr.set_integrator(4)
tr.build_buffers()
for s in range(first, first + spp):
    tr.reset(); tr.reset_counters()
    tr.render_samples(s, 1, True)
    tr.synchronize()
    img = tr.film()
    c = tr.counters()
    ref1, oc1 = osc.render(W, H, s, s + 1, depth, sum_mode=1, nthreads=16)
    diff = np.any(img[..., :3] != ref1[..., :3], axis=-1)
    if c["rays"] != int(oc1[0]) or diff.any():
        ys, xs = np.nonzero(diff)
        worst = np.abs(img[..., :3] - ref1[..., :3]).max(axis=-1)
        k = np.argsort(-worst.ravel())[:3]
        print(f"sample {s}: rays {c['rays']} / {int(oc1[0])}, {int(diff.sum())} pixels differ; largest: " +
              ", ".join(f"({i % W},{i // W}) device {img.reshape(-1, 4)[i, :3]} oracle {ref1.reshape(-1, 4)[i, :3]}" for i in k), flush=True)
