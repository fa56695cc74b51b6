import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from psoap_amd import _lib
L = _lib.load_bench()
names = {0: "full loop", 1: "no loads/stores", 2: "no barrier", 3: "no loads/stores, no barrier", 4: "no frag reads",
         5: "no frag reads, no loads/stores", 6: "no frag reads, no barrier", 7: "MFMA only"}
for abl in range(8):
    t = ctypes.c_double()
    _lib.check_bench(L.psoap_microbench_tile_engine(0, 16 + abl, ctypes.byref(t)), "tile")
    print(f"ablation {abl} ({names[abl]:32s}): {t.value:6.2f} TF")
