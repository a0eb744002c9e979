"""Development aid: at which maximum depth the device's and the oracle's paths begin to part.  Cornell box, plain and
glass, 512 x 512, four single samples per depth: the share of pixel samples whose values differ at all and by more than
1e-4 relative.  With a lift, the short block is raised by that much through its object transform first (its bottom face
lies in the floor's plane as the scene file has it).  usage: python scripts/depth_probe.py [lift]"""
import os, sys
ROOT = os.getcwd()
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import load_cpt
from helpers import cornell, OracleScene
cpt = load_cpt()
side = 512
lift = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
for glass in (False, True):
    objs, cam = cornell(glass=glass, bsdf_type=2)
    if lift != 0.0:
        for o in objs:
            if "short" in o["name"].lower():
                o["trs"] = ([0.0, lift, 0.0], [0.0, 0.0, 0.0], [1.0, 1.0, 1.0])
    osc = OracleScene(objs, cam)
    sc = cpt.Scene()
    for o in objs:
        sc.add_object(o["name"], o["positions"], o["normals"], o["texcoords"], o["indices"], o["material"])
        if "trs" in o:
            sc.set_trs(sc.num_objects() - 1, *o["trs"])
    sc.set_camera(cam["pos"], cam["look_at"], cam["up"], cam["fov"], side, side)
    for depth in ((1, 2, 3, 4, 5, 6, 8) if lift == 0.0 else (5, 8)):
        tr = cpt.Tracer(sc, side, side, max_depth=depth); tr.build_buffers()
        anyd = big = tot = 0
        for s in (1, 2, 3, 4):
            tr.reset(); tr.render_samples(s, 1, True); tr.synchronize()
            img = tr.film()[..., :3]
            ref = osc.render(side, side, s, s + 1, depth, sum_mode=1, nthreads=16)[0][..., :3]
            anyd += int(np.any(img != ref, axis=-1).sum())
            big += int((np.abs(img - ref).max(axis=-1) > 1e-4 * np.maximum(np.abs(ref).max(axis=-1), 1e-3)).sum())
            tot += side * side
        print(f"glass {glass} depth {depth}: pixel-samples that differ at all {anyd / tot:.2e}, by more than 1e-4 relative {big / tot:.2e}", flush=True)
        del tr
