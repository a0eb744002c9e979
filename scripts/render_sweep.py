"""Bounce-loop parity sweep beyond the test suite's fixed cases: the Cornell box (plain and glass) with a random BSDF
kind, maximum depth, film size, camera offset and random object transforms (translation, rotation about three axes,
non-uniform scale on the blocks and the light), rendered by a random integrator of the device and by the oracle with
the same sample indices.  Both sides draw the same random numbers and the ray / scene arithmetic is bit-exact, so the
films differ only where a last-place difference between the device's and the host's sinf / cosf / powf flips a
decision of a path.  Reported per case: relative L2 error, fraction of pixels that differ at all, ray counts.
usage: python scripts/render_sweep.py [seconds=150] [first seed=1]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import load_cpt
from helpers import cornell, OracleScene

cpt = load_cpt()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 150.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
t_begin = time.time()
out = {"cases": [], "tolerance_rel_l2": 1e-3}
while time.time() - t_begin < budget:
    rng = np.random.Generator(np.random.PCG64(seed))
    glass = bool(rng.integers(0, 2))
    bsdf = int(rng.integers(0, 4))
    depth = int(rng.choice([1, 2, 3, 5, 8, -1]))
    integrator = int(rng.choice([0, 1, 2, 4, 5]))
    W, H = int(rng.integers(33, 161)), int(rng.integers(33, 121))
    spp = int(rng.integers(4, 33))
    first = int(rng.integers(1, 50))
    objs, cam = cornell(glass=glass, bsdf_type=bsdf)
    cam = dict(cam)
    cam["pos"] = (np.asarray(cam["pos"], np.float32) + rng.uniform(-0.15, 0.15, 3).astype(np.float32)).astype(np.float32)
    moved = bool(rng.integers(0, 2))
    if moved:
        for o in objs:
            if any(k in o["name"].lower() for k in ("tall", "short", "light")):
                o["trs"] = ([float(x) for x in rng.uniform(-0.08, 0.08, 3)], [float(x) for x in rng.uniform(-25.0, 25.0, 3)], [float(x) for x in rng.uniform(0.8, 1.2, 3)])
    osc = OracleScene(objs, cam)
    sc = cpt.Scene()
    for o in objs:
        sc.add_object(o["name"], o["positions"], o["normals"], o["texcoords"], o["indices"], o["material"])
        if "trs" in o:
            sc.set_trs(sc.num_objects() - 1, *o["trs"])
    sc.set_camera(cam["pos"], cam["look_at"], cam["up"], cam["fov"], W, H)
    tr = cpt.Tracer(sc, W, H, max_depth=depth)
    tr.set_integrator(integrator)
    tr.build_buffers()
    tr.reset()
    tr.render_samples(first, spp, True)            # sums of sample indices first .. first + spp - 1
    tr.synchronize()
    img = tr.film()
    ref, ocnt = osc.render(W, H, first, first + spp, depth, sum_mode=1, nthreads=16)
    c = tr.counters()
    a, b = img[..., :3].astype(np.float64), ref[..., :3].astype(np.float64)
    err = float(np.sqrt(((a - b) ** 2).sum() / max((b ** 2).sum(), 1e-30)))
    differing = float(np.any(a != b, axis=-1).mean())
    case = {"seed": seed, "glass": glass, "bsdf": bsdf, "max_depth": depth, "integrator": integrator, "film": [W, H], "spp": spp, "first_sample": first, "moved_objects": moved,
            "rel_l2": err, "pixels_that_differ": differing, "finite": bool(np.all(np.isfinite(img))), "alpha_equal": bool(np.array_equal(img[..., 3], ref[..., 3])),
            "rays_device": int(c["rays"]), "rays_oracle": int(ocnt[0])}
    out["cases"].append(case)
    print(f"seed {seed}: rel L2 {err:.2e}, pixels that differ {differing:.4f}, rays {c['rays']} / {int(ocnt[0])}, {time.time() - t_begin:.0f} s", file=sys.stderr, flush=True)
    del tr, sc, osc
    seed += 1
errs = np.array([c["rel_l2"] for c in out["cases"]])
out["summary"] = {"cases": len(errs), "rel_l2_max": float(errs.max()), "rel_l2_median": float(np.median(errs)), "cases_above_tolerance": int((errs > 1e-3).sum()),
                  "bit_identical_films": int((errs == 0.0).sum()), "pixels_that_differ_max": float(max(c["pixels_that_differ"] for c in out["cases"])),
                  "all_finite": all(c["finite"] for c in out["cases"]), "all_alphas_equal": all(c["alpha_equal"] for c in out["cases"]),
                  "ray_count_rel_diff_max": float(max(abs(c["rays_device"] - c["rays_oracle"]) / max(c["rays_oracle"], 1) for c in out["cases"]))}
out["seconds"] = round(time.time() - t_begin, 1)
print(json.dumps(out))
