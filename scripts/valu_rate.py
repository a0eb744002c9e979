"""Vector-instruction issue rate of the device at 1..8 waves per SIMD (cpt_valu_issue_rate), both instruction mixes.
usage: python scripts/valu_rate.py [out.json]"""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_cpt
cpt = load_cpt()
cpt.require_device()
out = {}
for mix, name in ((0, "v_fma_f32"), (1, "slab mix (sub, mul, cmp+cndmask, max, min)"), (2, "v_pk_fma_f32 (two fp32 operations per instruction)")):
    rows = {}
    for w in (1, 2, 3, 4, 5, 6, 8):
        g, c, mhz = C.c_double(0), C.c_double(0), C.c_double(0)
        cpt.check(cpt.lib.cpt_valu_issue_rate(mix, w, C.byref(g), C.byref(c), C.byref(mhz)))
        rows[str(w)] = {"ginst_per_s": round(g.value, 1), "cycles_per_inst_per_wave": round(c.value, 3), "simd_issue_interval_cycles": round(c.value / w, 3), "shader_mhz": round(mhz.value, 1)}
        print(name, w, rows[str(w)], flush=True)
    out[name] = rows
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
