"""Tile engine alone, two workgroups per compute unit and ONE (+ 64: what a workgroup gets while its neighbour is outside its
K-loop), for the bench library PSOAP_BENCH_LIB points at (A/B of two builds: tools/engine_ab.py under each)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from psoap_amd import _lib
L = _lib.load_bench()
for rep in range(3):
    row = []
    for v, name in ((8, "2 WG/CU HBM"), (8 + 64, "1 WG/CU HBM"), (9, "2 WG/CU L2"), (9 + 64, "1 WG/CU L2")):
        t = ctypes.c_double()
        _lib.check_bench(L.psoap_microbench_tile_engine(0, v, ctypes.byref(t)), "tile")
        row.append(f"{name}: {t.value:6.2f}")
    print(os.environ.get("PSOAP_BENCH_LIB", "in-tree").split("/")[-1], " | ".join(row), "TFLOP/s")
