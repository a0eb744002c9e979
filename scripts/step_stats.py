"""Where the pooled megakernel spends its time, by kind of step (development build: make FLAGS+=-DCPT_STEP_STATS).
usage: python scripts/step_stats.py [spp] [traversal 0|1]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import load_cpt
from helpers import SCENES

cpt = load_cpt()
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 32
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 0
d = os.path.join(SCENES, "cornell-box-glass")
sc = cpt.Scene().load_obj(os.path.join(d, "cornell-box.obj"), 2).load_json(os.path.join(d, "cornell-box.json"))
tr = cpt.Tracer(sc, 1024, 1024, max_depth=8)
tr.set_integrator(5)
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
print(f"{1024 * 1024 * spp / ms / 1e3:.1f} Mpaths/s (instrumented build)")
for k in range(6):
    share = time[k] / time[5]
    per_trip = time[k] / max(trips[k], 1)
    occ = lanes[k] / max(trips[k], 1) / 64
    print(f"{names[k]:9s} time share {share:6.3f}  trips {trips[k]:.3e}  clocks/trip {per_trip:8.1f}  lanes/trip {occ:5.3f}")
print(f"other     time share {1 - time[0:5].sum() / time[5]:6.3f}")
