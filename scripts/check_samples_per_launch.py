"""One-off check at configuration-4 size: the film of the 10M-triangle heightfield at 3840x2160 after N samples is the same
bit for bit with 1 and with 32 samples per launch.  usage: python scripts/check_samples_per_launch.py [ntris width height spp]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import load_cpt
from helpers import synthetic_scene
cpt = load_cpt()
ntris = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
W = int(sys.argv[2]) if len(sys.argv) > 2 else 3840
H = int(sys.argv[3]) if len(sys.argv) > 3 else 2160
spp = int(sys.argv[4]) if len(sys.argv) > 4 else 48
objs, cam = synthetic_scene("heightfield", ntris)
scene = cpt.Scene()
for o in objs:
    scene.add_object(o["name"], o["positions"], o["normals"], o["texcoords"], o["indices"], o["material"])
scene.set_camera([0.5, 0.3, 0.1], [0.5, 0.0, 0.45], [0, 1, 0], 40.0, W, H)
films = []
for per_launch in (1, 32):
    tr = cpt.Tracer(scene, W, H, max_depth=-1)
    tr.set_integrator(3)
    tr.set_samples_per_launch(per_launch)
    tr.build_buffers()
    ms = tr.update_timed(spp)
    films.append(tr.film())
    print(f"{per_launch} per launch: {ms / spp:.3f} ms per sample", flush=True)
same = np.array_equal(films[0].view(np.uint32), films[1].view(np.uint32))
print("films identical:", same)
sys.exit(0 if same else 1)
