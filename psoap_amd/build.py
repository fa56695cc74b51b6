"""Build the gfx950 shared library in-tree with hipcc (cross-compiles without a GPU), and refuse to install a binary
whose device assembly shows the compiler defect described in ``psoap_amd/asmcheck.py`` (vector-register writes ahead
of an exec restore: the GPU fault of round 2's latency-scheme kernels).

    python -m psoap_amd.build            # rebuild if the sources / compiler / flags changed
    python -m psoap_amd.build --force
"""
from __future__ import annotations

import hashlib
import json
import os
import shutil
import subprocess
import tempfile

from . import asmcheck

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_PATH = os.path.join(CSRC, "libpsoap_gp.so")
SOURCES = ["psoap_gp.hip"]
# (microbench_kernels.hpp belongs to the measurement library below, not to the product)
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith(".hpp") and f not in ("microbench_kernels.hpp", "litmus_kernels.hpp")) + \
    [os.path.join("..", "..", "include", "psoap_gp.h")]
# The measurement library (include/psoap_bench.h): micro-benchmarks and the exp() self-check that bench.py, tools/ and
# one GPU test load.  Product kernels only in libpsoap_gp.so.
BENCH_LIB_PATH = os.path.join(CSRC, "libpsoap_bench.so")
BENCH_SOURCES = ["psoap_bench.hip"]
BENCH_HEADERS = ["common.hpp", "gemm_core.hpp", "potrf_blocked.hpp", "fill_kernels.hpp", "microbench_kernels.hpp", "litmus_kernels.hpp",
                 os.path.join("..", "..", "include", "psoap_bench.h")]
BENCH_HASH_PATH = BENCH_LIB_PATH + ".srchash"
HASH_PATH = LIB_PATH + ".srchash"                      # JSON: what the library beside it was built from
ASM_PATH = os.path.join(CSRC, "libpsoap_gp.device.s")   # device assembly of that build (kept for the CPU tests)

BASE_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared"]
# Flag sets tried in order until the device assembly is clean.  The second drops the out-of-line routines of the
# latency-scheme kernels: no following strip solves, and the fused diagonal task compiled into the kernels (no function
# call, so no values parked around one): 2-10 % slower single evaluations, and the form that never showed the defect.
FLAG_LADDER = [[], ["-DPSOAP_NO_FOLLOW", "-DPSOAP_DIAG_INLINE"]]


class BuildError(RuntimeError):
    pass


def hipcc() -> str:
    return os.environ.get("HIPCC", "hipcc")


def source_hash(files=None) -> str:
    """SHA-256 over the kernel sources the library is built from (names and contents)."""
    h = hashlib.sha256()
    for f in (SOURCES + HEADERS if files is None else files):
        p = os.path.join(CSRC, f)
        h.update(os.path.basename(p).encode())
        with open(p, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def compiler_id() -> str:
    """First line of `hipcc --version` that names the compiler build (the defect is a property of the compiler)."""
    try:
        out = subprocess.run([hipcc(), "--version"], capture_output=True, text=True).stdout
    except OSError:
        return "unknown"
    for ln in out.splitlines():
        if "clang version" in ln:
            return ln.strip()
    return out.strip().splitlines()[0] if out.strip() else "unknown"


def extra_flags() -> list[str]:
    """PSOAP_BUILD_FLAGS: extra hipcc flags (experiments, tools/lat_variants.py)"""
    return os.environ.get("PSOAP_BUILD_FLAGS", "").split()


def build_record() -> dict | None:
    if not os.path.exists(HASH_PATH):
        return None
    try:
        with open(HASH_PATH) as fh:
            return json.load(fh)
    except ValueError:
        return None            # a hash file of the round-2 format


def _stale() -> bool:
    """The library is missing or was not built from the sources as they are now with these extra flags (content hash
    kept beside it: modification times say nothing after a `git checkout` of a kernel header over an experimental
    build).  The compiler: a library whose sources match is REBUILT for another compiler only where that is asked for
    (PSOAP_REBUILD_ON_COMPILER_CHANGE=1, or `--force`) -- a machine without hipcc, or with a hipcc whose version line
    differs, loads the shipped library as it is (its SHA-256 is what the profiles cite) and says so once."""
    rec = build_record()
    if not os.path.exists(LIB_PATH) or rec is None:
        return True
    if rec.get("sources") != source_hash() or rec.get("extra_flags") != extra_flags():
        return True
    cid = compiler_id()
    if rec.get("compiler") != cid:
        if cid != "unknown" and os.environ.get("PSOAP_REBUILD_ON_COMPILER_CHANGE") == "1":
            return True
        global _warned_compiler
        if not _warned_compiler:
            _warned_compiler = True
            import warnings
            warnings.warn(f"libpsoap_gp.so was built by {rec.get('compiler')!r}; this machine has {cid!r}: using the "
                          "library as shipped (PSOAP_REBUILD_ON_COMPILER_CHANGE=1 rebuilds and re-checks it)")
    return False


_warned_compiler = False


def compile_once(flags: list[str], out_dir: str) -> tuple[str, str]:
    """hipcc -> (shared library, device assembly) in out_dir"""
    so = os.path.join(out_dir, "libpsoap_gp.so")
    cmd = [hipcc(), *BASE_FLAGS, "-save-temps=obj", *flags, *[os.path.join(CSRC, s) for s in SOURCES], "-o", so]
    subprocess.check_call(cmd, cwd=out_dir)
    asm = [f for f in os.listdir(out_dir) if f.endswith(".s") and "amdgcn" in f]
    if len(asm) != 1:
        raise BuildError(f"expected one device assembly file from -save-temps, found {asm}")
    return so, os.path.join(out_dir, asm[0])


def compile_checked(extra: list[str], out_dir: str, verbose: bool = False, scan=asmcheck.scan_exec_restore
                    ) -> tuple[str, str, dict]:
    """Walk FLAG_LADDER until the device assembly is clean -> (library, assembly, build record), all inside out_dir.
    ``scan``: the assembly check (tests substitute one that rejects a chosen rung)."""
    tried = []
    for k, rung in enumerate(FLAG_LADDER):
        flags = extra + rung
        sub = os.path.join(out_dir, f"rung{k}")
        os.makedirs(sub, exist_ok=True)
        if verbose:
            print(f"{hipcc()} {' '.join(BASE_FLAGS + flags)} psoap_gp.hip")
        so, asm_path = compile_once(flags, sub)
        with open(asm_path) as fh:
            hits = scan(fh.read())
        tried.append((flags, hits))
        if verbose and not hits:
            print("  assembly scan clean (asmcheck.scan_exec_restore: the ONE known code shape of the exec-restore defect, "
                  "within basic blocks; not a proof of absence)")
        if hits:
            if verbose:
                for fn, label, line, pend in hits:
                    print(f"  exec-restore defect in {asmcheck.short(fn)}, block {label}: {len(pend)} vector writes")
            continue
        rec = {"sources": source_hash(), "compiler": compiler_id(), "extra_flags": extra, "flags": BASE_FLAGS + flags,
               "fallback_rung": k,
               "rejected": [{"flags": f, "kernels": sorted({asmcheck.short(h[0]) for h in hs})} for f, hs in tried[:-1]]}
        return so, asm_path, rec
    raise BuildError("every flag set produced device code with vector-register writes ahead of an exec restore "
                     "(psoap_amd/asmcheck.py): " +
                     "; ".join(f"{' '.join(f) or '(default)'} -> {sorted({asmcheck.short(h[0]) for h in hs})}"
                               for f, hs in tried))


def build(force: bool = False, verbose: bool = False) -> str:
    """hipcc --offload-arch=gfx950 -> psoap_amd/csrc/libpsoap_gp.so, after the assembly checks"""
    if not force and not _stale():
        return LIB_PATH
    # (a FIXED scratch directory: hipcc derives unit ids from the paths it is given, and a random temporary directory made
    # every build of the same sources a different binary -- 759 bytes of the fat binary -- so that the library a profile
    # names by its SHA-256 could not be rebuilt)
    tmp = os.path.join(tempfile.gettempdir(), "psoap_gfx950_build")
    import fcntl
    with open(tmp + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)           # one build at a time in that directory (ranks, test workers)
        if not force and not _stale():
            return LIB_PATH                        # somebody else built it while we waited
        shutil.rmtree(tmp, ignore_errors=True)
        os.makedirs(tmp)
        try:
            so, asm_path, rec = compile_checked(extra_flags(), tmp, verbose)
            shutil.copy(so, LIB_PATH)
            shutil.copy(asm_path, ASM_PATH)
            with open(HASH_PATH, "w") as fh:
                json.dump(rec, fh, indent=1)
                fh.write("\n")
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    return LIB_PATH


def library_sha256(path: str = LIB_PATH) -> str | None:
    if not os.path.exists(path):
        return None
    h = hashlib.sha256()
    with open(path, "rb") as fh:
        for blk in iter(lambda: fh.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def provenance() -> dict:
    """What bench.py prints beside its numbers: which binary ran."""
    rec = build_record() or {}
    return {"sha256": library_sha256(), "source_hash": rec.get("sources"), "compiler": rec.get("compiler"),
            "fallback_rung": rec.get("fallback_rung"), "extra_flags": rec.get("extra_flags"),
            "sources_match_tree": rec.get("sources") == source_hash()}


def build_bench(force: bool = False) -> str:
    """hipcc -> psoap_amd/csrc/libpsoap_bench.so (measurement kernels; no assembly gate: nothing in it is shipped)"""
    want = source_hash(BENCH_SOURCES + BENCH_HEADERS)
    if not force and os.path.exists(BENCH_LIB_PATH) and os.path.exists(BENCH_HASH_PATH):
        try:
            with open(BENCH_HASH_PATH) as fh:
                if json.load(fh).get("sources") == want:
                    return BENCH_LIB_PATH
        except ValueError:
            pass
    tmp = os.path.join(tempfile.gettempdir(), "psoap_gfx950_build_bench")
    import fcntl
    with open(tmp + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        shutil.rmtree(tmp, ignore_errors=True)
        os.makedirs(tmp)
        try:
            so = os.path.join(tmp, "libpsoap_bench.so")
            subprocess.check_call([hipcc(), *BASE_FLAGS, *[os.path.join(CSRC, f) for f in BENCH_SOURCES], "-o", so], cwd=tmp)
            shutil.copy(so, BENCH_LIB_PATH)
            with open(BENCH_HASH_PATH, "w") as fh:
                json.dump({"sources": want, "compiler": compiler_id(), "flags": BASE_FLAGS}, fh, indent=1)
                fh.write("\n")
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    return BENCH_LIB_PATH


def device_asm() -> str:
    """Device assembly of the installed library (rebuilding first if it is stale)."""
    build()
    if not os.path.exists(ASM_PATH):
        build(force=True)
    with open(ASM_PATH) as fh:
        return fh.read()


if __name__ == "__main__":
    import sys
    print(build(force="--force" in sys.argv, verbose=True))
    print(json.dumps(build_record(), indent=1))
    print(build_bench(force="--force" in sys.argv))
