import ctypes, sys, os
sys.path.insert(0, '/root/repo')
from psoap_amd import _lib
L = _lib.load_bench()
for ab in (0, 1, 2, 3):
    u = ctypes.c_double()
    rc = L.psoap_microbench_potrf(0, ab, ctypes.byref(u))
    print("potrf microbench ablate", ab, rc, round(u.value, 2), "us")
