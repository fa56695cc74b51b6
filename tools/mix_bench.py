"""fp64 MFMA waves beside fp64 vector-arithmetic waves on the same SIMDs (psoap_microbench_mix): each alone, then
together -- what the fused-fill epilogue of one workgroup costs the K-loop of its neighbour and vice versa."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from psoap_amd import _lib
L = _lib.load_bench()
IM, IE = 4000, 2500
def run(mode):
    out = np.zeros(3)
    rc = L.psoap_microbench_mix(0, mode, IM, IE, out.ctypes.data_as(_lib._dp))
    _lib.check_bench(rc, "psoap_microbench_mix")
    return out
for name, mode in (("MFMA alone", 1), ("exp alone", 2), ("MFMA + exp", 3), ("FMA chains alone", 6), ("MFMA + FMA chains", 7),
                   ("fp32 FMA chains alone", 8), ("MFMA + fp32 FMA chains", 9), ("int32 chains alone", 16), ("MFMA + int32 chains", 17),
                   ("MFMA + exp at priority 3", 35), ("MFMA + fp32 at priority 3", 41)):
    o = run(mode)
    print(f"{name:20s}: MFMA waves {o[0]:8.1f} us ({IM*4} MFMAs, {IM*4*64/o[0]/1e3 if o[0] > 1 else 0:.2f} GHz-equivalent issue)   "
          f"vector waves {o[1]:8.1f} us ({IE} batches of 4)   same-SIMD pairs {o[2]:.2f}")

out = np.zeros(1)
for name, v in (("tile engine, 2 workgroups per CU", 8), ("tile engine, 1 workgroup per CU", 8 + 64)):
    _lib.check_bench(L.psoap_microbench_tile_engine(0, v, out.ctypes.data_as(_lib._dp)), "tile engine")
    print(f"{name}: {out[0]:.1f} TFLOP/s")
