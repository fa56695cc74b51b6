"""Scan device assembly for the hipcc defect behind round 2's "build-dependent GPU fault of the latency-scheme kernels"
(vector-register writes ahead of an exec restore; psoap_amd/asmcheck.py has the description and the scanner).

    python tools/check_exec_restore.py                  # the installed library's assembly (builds it if stale)
    python tools/check_exec_restore.py file.s
    python tools/check_exec_restore.py -DFLAG ...        # compile psoap_gp.hip with extra flags and scan that
"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from psoap_amd import asmcheck, build  # noqa: E402

scan = asmcheck.scan_exec_restore
short = asmcheck.short


def main(argv):
    if argv and argv[0].endswith(".s"):
        with open(argv[0]) as fh:
            text = fh.read()
    elif argv:
        with tempfile.TemporaryDirectory() as tmp:
            out = os.path.join(tmp, "dev.s")
            subprocess.check_call([build.hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                                   *argv, os.path.join(build.CSRC, "psoap_gp.hip"), "-o", out], stderr=subprocess.DEVNULL)
            with open(out) as fh:
                text = fh.read()
    else:
        text = build.device_asm()
    hits = scan(text)
    for fn, label, line, pend in hits:
        print(f"{short(fn)}: join block {label} (line {line}): {len(pend)} vector-register write(s) ahead of the exec restore")
        for ln, ins in pend[:12]:
            print(f"    {ln}: {ins}")
    print(f"{len(hits)} join block(s) with vector-register writes under the partial exec mask")
    return 1 if hits else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
