"""Tile-engine staging comparison: register staging (variants 0/1) vs LDS-DMA (8/9)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from psoap_amd import _lib
L = _lib.load_bench()
for rep in range(2):
    for v, name in ((1, "reg staging, L2 operands"), (0, "reg staging, HBM operands"),
                    (9, "LDS-DMA, L2 operands"), (8, "LDS-DMA, HBM operands")):
        t = ctypes.c_double()
        _lib.check_bench(L.psoap_microbench_tile_engine(0, v, ctypes.byref(t)), "tile")
        print(f"{name:28s}: {t.value:6.2f} TF")
