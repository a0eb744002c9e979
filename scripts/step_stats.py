"""Where a persistent megakernel spends its time, by kind of step (development build: make FLAGS+=-DCPT_STEP_STATS).
usage: python scripts/step_stats.py [spp] [traversal 0|1] [cornell|heightfield|soup] [ntris] [width height depth]
The Cornell box runs the pooled kernel (integrator 5), the synthetic scenes the one that keeps traversal state across
shading (integrator 4)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import load_cpt
from helpers import SCENES, synthetic_scene

cpt = load_cpt()
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 32
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 0
kind = sys.argv[3] if len(sys.argv) > 3 else "cornell"
ntris = int(sys.argv[4]) if len(sys.argv) > 4 else 1_000_000
W = int(sys.argv[5]) if len(sys.argv) > 5 else (1024 if kind == "cornell" else 1920)
H = int(sys.argv[6]) if len(sys.argv) > 6 else (1024 if kind == "cornell" else 1080)
depth = int(sys.argv[7]) if len(sys.argv) > 7 else (8 if kind == "cornell" else 5)
if kind == "cornell":
    d = os.path.join(SCENES, "cornell-box-glass")
    sc = cpt.Scene().load_obj(os.path.join(d, "cornell-box.obj"), 2).load_json(os.path.join(d, "cornell-box.json"))
    integrator = 5
else:
    objs, cam = synthetic_scene(kind, ntris)
    sc = cpt.Scene()
    for o in objs:
        sc.add_object(o["name"], o["positions"], o["normals"], o["texcoords"], o["indices"], o["material"])
    sc.set_camera([0.5, 0.3, 0.1], [0.5, 0.0, 0.45], cam["up"], 40.0, W, H)      # the camera of the measurements (scenes.BENCH_CAMERA)
    integrator = 4
tr = cpt.Tracer(sc, W, H, max_depth=depth)
tr.set_integrator(integrator)
tr.set_traversal(mode)
tr.build_buffers()
tr.update(spp); tr.synchronize(); tr.reset()
out = (ctypes.c_ulonglong * 18)()
fn = cpt.lib.cpt_debug_step_stats
fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
assert fn(out, 1) == 0
ms = tr.update_timed(spp)
assert fn(out, 0) == 0
v = np.array(list(out), dtype=np.float64)
time, trips, lanes = v[0:6], v[6:12], v[12:18]
names = ["node", "triangle", "instance", "pop", "shade", "kernel"]
print(f"{kind}: {W * H * spp / ms / 1e3:.1f} Mpaths/s (instrumented build, integrator {integrator})")
for k in range(5):
    share = time[k] / time[5]
    per_trip = time[k] / max(trips[k], 1)
    occ = lanes[k] / max(trips[k], 1) / 64
    print(f"{names[k]:9s} time share {share:6.3f}  trips {trips[k]:.3e}  clocks/trip {per_trip:8.1f}  lanes/trip {occ:5.3f}")
print(f"other     time share {1 - time[0:5].sum() / time[5]:6.3f}")
print(f"lanes with a ray per traversal trip {lanes[5] / max(trips[0:3].sum(), 1) / 64:5.3f}, of them in the voted kind {lanes[0:3].sum() / max(lanes[5], 1):5.3f}")
