import sys, os, ctypes as C
sys.path.insert(0, "tests")
from conftest import load_cpt
cpt = load_cpt()
for b in (1 << 28, 1 << 30, 1 << 32):
    r = C.c_double(0)
    cpt.check(cpt.lib.cpt_stream_copy_rate(b, 10, C.byref(r)))
    print(b, round(r.value, 1), "GB/s")
