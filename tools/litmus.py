"""Coherence litmus tests between XCDs (psoap_amd/csrc/litmus_kernels.hpp):   python tools/litmus.py [ITERS]
What does a wave read of a 256-byte unit that a wave on another XCD has just rewritten and announced?  One row per
combination of: how the reader touched the unit before (plant), how the writer stored, how the reader reads, writer on the
same / another XCD, chip idle / streaming 512 MiB through the L2s."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from psoap_amd import _lib

L = _lib.load_bench()
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
PLANT = ["none", "sc1 load", "plain load", "acq+plain"]
WRITE = ["sc1 st", "plain st+rel", "sc0sc1 st"]
READ = ["sc1 ld", "acq+sc1 ld", "acq+plain ld", "rmw", "plain ld", "acq+lds-dma", "sc0sc1 ld"]
out = (ctypes.c_ulonglong * 8)()
print(f"{'plant':11s} {'writer':13s} {'reader':13s} xcd  bg | iters plant_stale read_stale never max_us mean_us (reader xcc, writer xcc)")
for bg in (0, 1):
    for same in (0, 1):
        for plant in range(4):
            for wr in range(3):
                for rd in range(7):
                    if bg and (plant in (0, 3) or rd in (4, 6)):      # (the loaded runs: the combinations that matter)
                        continue
                    n_it = iters if not bg else iters // 4
                    if rd == 4:
                        n_it = min(n_it, 1000)          # (no acquire in front of a plain load: stale for milliseconds, by design)
                    rc = L.psoap_litmus_l2(0, plant, wr, rd, same, n_it, bg, out)
                    if rc:
                        print("error:", L.psoap_bench_last_error().decode())
                        sys.exit(1)
                    it, ps, rs, nv, mx, sm, rx, wx = [int(x) for x in out]
                    mean = (sm / max(rs - nv, 1)) / 100.0
                    print(f"{PLANT[plant]:11s} {WRITE[wr]:13s} {READ[rd]:13s} {'same' if same else 'diff'} {bg:3d} | {it:6d} {ps:8d} {rs:8d} {nv:6d} "
                          f"{mx / 100.0:7.2f} {mean:7.2f}  ({rx}, {wx})", flush=True)
rc = L.psoap_litmus_writeback(0, iters, out)
if rc:
    print("error:", L.psoap_bench_last_error().decode())
    sys.exit(1)
print(f"write-back of a partly rewritten line: {int(out[0])} iterations, another XCD's words overwritten in {int(out[2])}, "
      f"the plain-stored word lost in {int(out[1])}")
