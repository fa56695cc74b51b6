import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from psoap_amd import _lib
L = _lib.load_bench()
for ab, name in enumerate(["full", "no chol16", "no B/C mfma", "no W output"]):
    us = ctypes.c_double()
    _lib.check_bench(L.psoap_microbench_potrf(0, ab, ctypes.byref(us)), "potrf bench")
    print(f"{name:14s} {us.value:8.2f} us")
us = ctypes.c_double()
_lib.check_bench(L.psoap_microbench_potrf(0, 9, ctypes.byref(us)), "potrf bench")
print("stamped build", us.value, "us")
