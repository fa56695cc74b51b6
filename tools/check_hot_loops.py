"""For every k_chol_dag instantiation of the installed library: the scratch (spill) accesses inside the basic blocks
that hold the 64-MFMA K-loop stages.  A reload there costs more than its latency: the s_waitcnt vmcnt(0) behind it
also waits for the LDS-DMA of the next stage.  (psoap_amd/asmcheck.py: scan_hot_loops)

    python tools/check_hot_loops.py        # exit code 1 if any K-loop stage block touches scratch
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from psoap_amd import asmcheck, build  # noqa: E402

text = build.device_asm()
res = asmcheck.scan_hot_loops(text)
meta = asmcheck.kernel_resources(text)
bad = 0
for name, (n_hot, spills) in sorted(res.items()):
    r = meta.get(name, {})
    print(f"{name}: {n_hot} K-loop stage blocks, {spills} scratch accesses inside them; "
          f"vgpr_spill_count {r.get('vgpr_spill_count')}, private segment {r.get('private_segment_fixed_size')} B")
    bad += spills
sys.exit(1 if bad else 0)
