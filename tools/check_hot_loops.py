"""Compile the HIP library to assembly and report, for every k_chol_dag instantiation, the scratch (spill)
loads / stores inside the basic blocks that hold the 64-MFMA K-loop stages.  A reload there costs more than
its latency: the s_waitcnt vmcnt(0) behind it also waits for the LDS-DMA of the next stage.

    python tools/check_hot_loops.py        # exit code 1 if any K-loop stage block touches scratch
"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(tempfile.mkdtemp(prefix="psoap_asm_"), "psoap.s")
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                       os.path.join(ROOT, "psoap_amd", "csrc", "psoap_gp.hip"), "-o", out], stderr=subprocess.DEVNULL)
lines = open(out).read().split("\n")
bad = 0
i = 0
while i < len(lines):
    m = re.match(r"^(_ZN5psoap10k_chol_dag\w+):", lines[i])
    if not m:
        i += 1
        continue
    name = m.group(1)
    j = i
    while not lines[j].strip().startswith(".Lfunc_end"):
        j += 1
    cur, blocks = "entry", {}
    for l in lines[i:j]:
        mm = re.match(r"^(\.LBB\d+_\d+):", l)
        if mm:
            cur = mm.group(1)
        b = blocks.setdefault(cur, [0, 0, 0])
        if "v_mfma" in l: b[0] += 1
        if "scratch_load" in l: b[1] += 1
        if "scratch_store" in l: b[2] += 1
    hot = {k: v for k, v in blocks.items() if v[0] == 64}     # one K-loop stage = 64 MFMAs per wave
    spills = sum(v[1] + v[2] for v in hot.values())
    tmpl = re.search(r"k_chol_dagILi(\d)ELb(\d)ELb(\d)E", name)
    print(f"k_chol_dag<C={tmpl.group(1)}, AUG={tmpl.group(2)}, LAT={tmpl.group(3)}>: {len(hot)} K-loop stage blocks, "
          f"{spills} scratch accesses inside them")
    bad += spills
    i = j
sys.exit(1 if bad else 0)
