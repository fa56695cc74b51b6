"""Build the gfx950 shared library in-tree with hipcc (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_PATH = os.path.join(CSRC, "libpsoap_gp.so")
SOURCES = ["psoap_gp.hip"]
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith(".hpp")) + [os.path.join("..", "..", "include", "psoap_gp.h")]


def _stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    for f in SOURCES + HEADERS:
        p = os.path.join(CSRC, f)
        if os.path.exists(p) and os.path.getmtime(p) > t:
            return True
    return False


def build(force: bool = False, verbose: bool = False) -> str:
    """hipcc --offload-arch=gfx950 -> psoap_amd/csrc/libpsoap_gp.so"""
    if not force and not _stale():
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           *[os.path.join(CSRC, s) for s in SOURCES], "-o", LIB_PATH]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))
