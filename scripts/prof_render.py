"""Small driver for rocprofv3 passes: N launches of the bounce-loop kernel on cornell-box-glass
(no torch, no oracle).  usage: python scripts/prof_render.py [steps] [width] [depth] [integrator] [wavefront tail] [samples per launch] [records in LDS 1|0] [scene]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_cpt
cpt = load_cpt()
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 16
width = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
depth = int(sys.argv[3]) if len(sys.argv) > 3 else 8
mode = int(sys.argv[4]) if len(sys.argv) > 4 else 0
d = os.path.join(ROOT, "tests", "golden", "scenes", sys.argv[8] if len(sys.argv) > 8 else "cornell-box-glass")
scene = cpt.Scene().load_obj(os.path.join(d, "cornell-box.obj"), 2).load_json(os.path.join(d, "cornell-box.json"))
tr = cpt.Tracer(scene, width, width, max_depth=depth)
tr.set_integrator(mode)
if len(sys.argv) > 5:
    tr.set_wavefront_tail(int(sys.argv[5]))
if len(sys.argv) > 6:
    tr.set_samples_per_launch(int(sys.argv[6]))
tr.build_buffers()
in_lds = tr.set_scene_in_lds(int(sys.argv[7]) != 0) if len(sys.argv) > 7 else None
tr.update(steps); tr.synchronize(); tr.reset_counters()       # same launch shape as the timed run
ms = tr.update_timed(steps)
c = tr.counters()
print("" if in_lds is None else f"records in LDS: {in_lds}; ", end="")
print(f"integrator {mode}: {steps} samples {ms/steps:.4f} ms each, {width*width*steps/ms/1e3:.1f} Mpaths/s, {c['rays']/ms/1e3:.1f} Mrays/s", c)
