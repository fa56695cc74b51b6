"""Build the gfx950 shared library in-tree with hipcc (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_PATH = os.path.join(CSRC, "libpsoap_gp.so")
SOURCES = ["psoap_gp.hip"]
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith(".hpp")) + [os.path.join("..", "..", "include", "psoap_gp.h")]


HASH_PATH = LIB_PATH + ".srchash"


def source_hash() -> str:
    """SHA-256 over the kernel sources the library is built from (names and contents)."""
    import hashlib
    h = hashlib.sha256()
    for f in SOURCES + HEADERS:
        p = os.path.join(CSRC, f)
        h.update(os.path.basename(p).encode())
        with open(p, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def _stale() -> bool:
    """The library is missing or was not built from the sources as they are now (content hash kept beside it:
    modification times say nothing after a `git checkout` of a kernel header over an experimental build)."""
    if not os.path.exists(LIB_PATH) or not os.path.exists(HASH_PATH):
        return True
    with open(HASH_PATH) as fh:
        return fh.read().strip() != source_hash()


def build(force: bool = False, verbose: bool = False) -> str:
    """hipcc --offload-arch=gfx950 -> psoap_amd/csrc/libpsoap_gp.so"""
    if not force and not _stale():
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           *[os.path.join(CSRC, s) for s in SOURCES], "-o", LIB_PATH]
    if verbose:
        print(" ".join(cmd))
    digest = source_hash()
    subprocess.check_call(cmd)
    with open(HASH_PATH, "w") as fh:
        fh.write(digest + "\n")
    return LIB_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))
