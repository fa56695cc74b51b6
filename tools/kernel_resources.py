"""Registers, spills and scratch of every kernel in a device assembly file (the metadata hipcc writes with -save-temps).
    python tools/kernel_resources.py [libpsoap_gp.device.s] [name filter]"""
import os
import re
import sys

path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                          "psoap_amd", "csrc", "libpsoap_gp.device.s")
flt = sys.argv[2] if len(sys.argv) > 2 else ""
keys = ("agpr_count", "name", "private_segment_fixed_size", "sgpr_count", "sgpr_spill_count", "vgpr_count", "vgpr_spill_count",
        "group_segment_fixed_size")
recs, cur = [], None
with open(path) as fh:
    for ln in fh:
        m = re.match(r"\s+(?:- )?\.(\w+):\s+(\S+)\s*$", ln)
        if not m or m.group(1) not in keys:
            continue
        k, v = m.groups()
        if k == "agpr_count":
            cur = {}
            recs.append(cur)
        if cur is not None:
            cur[k] = v


def short(sym):
    """k_chol_dag<C, AUG, LAT, STREAM, WPE> from the mangled name; other kernels: the bare name"""
    m = re.match(r"_ZN5psoap\d+(\w+?)I((?:L[ib]\d+E)+)E", sym)
    if m:
        args = re.findall(r"L([ib])(\d+)E", m.group(2))
        return m.group(1) + "<" + ", ".join(a[1] for a in args) + ">"
    m = re.match(r"_ZN5psoap\d+([A-Za-z_0-9]+?)E", sym)
    return m.group(1) if m else sym


print(f"{'kernel':58s} vgpr agpr sgpr  vspill sspill scratch_B")
for r in recs:
    if "name" not in r:
        continue
    n = short(r["name"])
    if flt and flt not in n:
        continue
    print(f"{n:58s} {r.get('vgpr_count', '?'):>4s} {r.get('agpr_count', '?'):>4s} {r.get('sgpr_count', '?'):>4s}  "
          f"{r.get('vgpr_spill_count', '?'):>6s} {r.get('sgpr_spill_count', '?'):>6s} {r.get('private_segment_fixed_size', '?'):>9s}")
