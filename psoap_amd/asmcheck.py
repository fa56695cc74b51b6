"""Checks on the device assembly hipcc produced for the library -- run by ``psoap_amd.build`` on every build (a library
that fails them is not installed) and by the CPU tests.

1. ``scan_exec_restore``: vector-register writes under a stale exec mask.  hipcc (ROCm 7.2, AMD clang 22.0.0git) lowers
   ``if (threadIdx.x == 0) { ... }`` to

        s_and_saveexec_b64 s[A:B], vcc        ; exec := the lanes that take the branch, old exec -> s[A:B]
        s_cbranch_execz    .LBB_join
        ...                                   ; then-block, exec partial
    .LBB_join:
        s_or_b64 exec, exec, s[A:B]           ; all lanes back

   and everything between the label and the ``s_or_b64`` still runs under the PARTIAL mask (exec == 0 in the waves that
   skipped).  When its register allocator needs copies or reloads at the top of the join block -- e.g. to park values that
   live across a function call in callee-saved VGPRs -- it sometimes puts them AHEAD of the ``s_or_b64``: the copy happens
   only in the lanes of the ``if``, and the copy back after the call hands every other lane garbage.  This was the
   "build-dependent GPU fault of the latency-scheme kernels" of round 2: in ``k_chol_dag<3, true, true>`` the value was
   threadIdx.x, parked around the call of ``dag_diag_fast`` behind the one-lane poll of ``dag_wait_ge``; the next LDS-DMA
   staging took its row index from the garbage (``profiles/r3_lat_fault_rocgdb.txt``; prediction against
   outcome for 30 builds in ``profiles/r3_lat_variant_matrix.md``).  Whether it happens depends on the allocator's split
   decisions, i.e. on unrelated code.

2. ``scan_hot_loops``: scratch (spill) accesses inside the 64-MFMA K-loop stages of the ``k_chol_dag`` kernels -- a
   reload there also waits for the LDS-DMA of the next stage (``s_waitcnt vmcnt(0)`` counts both) and serialises it with
   the MFMAs.  A performance defect, not a correctness one: reported, not fatal for a build.
"""
from __future__ import annotations

import re

_VWRITE = re.compile(r"^(v_|scratch_load|global_load|flat_load|ds_read|ds_bpermute|ds_permute|buffer_load|"
                     r"global_atomic|flat_atomic|image_)")
# no vector destination, or (v_writelane: an SGPR spill into ONE lane) a write that ignores the exec mask
_NO_VDST = re.compile(r"^(v_cmp|v_cmpx|v_readlane|v_readfirstlane|v_writelane|global_load_lds|v_nop)")
_EXEC_RESTORE = re.compile(r"^s_or_b64\s+exec,\s*(exec,\s*s\[\d+:\d+\]|s\[\d+:\d+\],\s*exec)")
# the other ways this compiler re-enables lanes at a join: a plain copy of the saved mask, or-and-save (the else mask)
_EXEC_RESTORE_ALT = re.compile(r"^(s_mov_b64\s+exec,\s*s\[\d+:\d+\]|s_or_saveexec_b64\s+s\[\d+:\d+\],\s*s\[\d+:\d+\])")
_BRANCH_Z = re.compile(r"^s_cbranch_execz\s+(\S+)")
_LABEL = re.compile(r"^([A-Za-z_.$][\w.$]*):")


def short(mangled: str | None) -> str:
    """k_chol_dag<3,true,true> for _ZN5psoap10k_chol_dagILi3ELb1ELb1EEE...; other names are returned as they are"""
    m = re.match(r"_ZN5psoap10k_chol_dagILi(\d)ELb([01])ELb([01])E(?:Lb([01])E)?(?:Li(\d)E)?EE", mangled or "")
    if m:
        # (the fourth parameter, STREAM, and the fifth, one wave per SIMD, are printed only when set: the names of rounds
        # 1-3 stay as they were)
        return "k_chol_dag<%s,%s,%s%s%s>" % (m.group(1), "true" if m.group(2) == "1" else "false",
                                             "true" if m.group(3) == "1" else "false", ",stream" if m.group(4) == "1" else "",
                                             ",wide" if m.group(5) == "1" else "")
    m = re.match(r"_ZN5psoap11dag_specialILi(\d)ELb([01])E(?:Li(\d)E)?EE", mangled or "")
    if m:
        return "dag_special<%s,%s%s>" % (m.group(1), "true" if m.group(2) == "1" else "false", ",wide" if m.group(3) == "1" else "")
    return mangled or "?"


def _narrows_exec(code, lo: int, k: int) -> bool:
    """``s_mov_b64 exec, s[A:B]`` at line k, where inside the block (lines lo .. k) s[A:B] was last written by
    ``s_and_b64 s[A:B], s[X:Y], ...`` (either operand order) and s[X:Y] by ``s_mov_b64 s[X:Y], exec`` with no write of exec
    in between: hipcc's long form of ``s_and_saveexec`` -- the START of a masked region.  The new mask is a subset of the
    one the block ran under, so this is not the instruction that hands parked values to lanes that did not park them."""
    m = re.match(r"^s_mov_b64\s+exec,\s*(s\[\d+:\d+\])$", code[k])
    if not m:
        return False
    mask = m.group(1)
    saved = None
    for j in range(k - 1, lo - 1, -1):
        s = code[j]
        if not s or s.startswith(".") or _LABEL.match(s):
            continue
        parts = s.replace(",", " ").split()
        op, dst = parts[0], (parts[1] if len(parts) > 1 else "")
        if dst == "exec" or "saveexec" in op:
            return False
        if saved is None:
            if dst == mask:
                if op != "s_and_b64" or len(parts) < 4:
                    return False
                others = [x for x in parts[2:4] if x != mask]
                if not others:
                    return False
                saved = others[0]
        elif dst == saved:
            return op == "s_mov_b64" and len(parts) > 2 and parts[2] == "exec"
    return False


def scan_exec_restore(text: str):
    """-> [(function, join block, line of its label, [(line, instruction), ...]), ...]: the join blocks -- targets of an
    ``s_cbranch_execz``, or the fall-through exit of a loop closed by ``s_cbranch_execnz`` -- in which an instruction that
    writes a vector register sits ahead of the instruction that re-enables the lanes (``s_or_b64 exec, exec, s[..]``,
    ``s_mov_b64 exec, s[..]``, ``s_or_saveexec_b64``).

    SCOPE: this finds ONE code shape -- the one the defect was traced to -- inside one basic block.  A clean scan says
    that shape is absent, not that the binary is free of exec-mask defects: a restore behind an intermediate label or
    branch, or through another instruction form, is not followed.  The evidence that the shipped code shapes are sound
    is the GPU variant matrix (tools/lat_variants.py: 68 builds, prediction against outcome) and the GPU suites."""
    code = [ln.split(";")[0].strip() for ln in text.split("\n")]
    fn_of = []
    cur = None
    for s in code:
        m = _LABEL.match(s)
        if m and not m.group(1).startswith(".L"):
            cur = m.group(1)
        fn_of.append(cur)
    skip_targets = set()
    for n, s in enumerate(code):
        b = _BRANCH_Z.match(s)
        if b:
            skip_targets.add((fn_of[n], b.group(1)))
    hits = []
    for n, s in enumerate(code):
        m = _LABEL.match(s)
        if m and (fn_of[n], m.group(1)) in skip_targets:
            name = m.group(1)
        elif s.startswith("s_cbranch_execnz") and not (n + 1 < len(code) and _LABEL.match(code[n + 1])):
            name = "(exit of the loop that ends at line %d)" % (n + 1)   # falls through with the lanes that left
        else:
            continue
        pend = []
        for k in range(n + 1, min(n + 400, len(code))):
            s2 = code[k]
            if not s2 or s2.startswith("."):
                continue
            if _LABEL.match(s2):
                break
            op = s2.split()[0]
            if _EXEC_RESTORE.match(s2) or _EXEC_RESTORE_ALT.match(s2):
                if _narrows_exec(code, n + 1, k):
                    break       # exec := exec & mask, spelled with a copy: no lane comes back, nothing was parked for one
                if pend:
                    hits.append((fn_of[n], name, n + 1, pend))
                break
            if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_swappc", "s_endpgm", "s_barrier")):
                break
            dst = s2.split(None, 1)[1].split(",")[0] if " " in s2 else ""
            if re.search(r"\bexec", dst) or "saveexec" in op:
                break           # exec rewritten some other way (else-mask, loop mask): not this pattern
            if _VWRITE.match(op) and not _NO_VDST.match(op):
                pend.append((k + 1, s2))
    return hits


def scan_serial_loads(text: str, min_run: int = 16):
    """-> {function: [(line, length), ...]}: runs of at least ``min_run`` global / flat loads each of which is followed by
    ``s_waitcnt vmcnt(0)`` before the next load is issued -- a tile read element by element, one round trip to memory per
    element.  hipcc schedules a plain C++ loop over a tile that way wherever it has only a few vector registers to spare
    (behind a K-loop); rounds 1-5 carried five such loops -- the fold of a partial tile (27 us per tile, 52 round trips on the
    row-to-row chain of every single evaluation) and the read-modify-write epilogues of the staged kernels -- until
    tile_load16 (gemm_core.hpp).  A performance defect: reported by the tests, not fatal for a build."""
    code = [ln.split(";")[0].strip() for ln in text.split("\n")]
    out = {}
    cur = None
    run_start, run_len, last_hit = 0, 0, -100
    load = re.compile(r"^(global|flat)_load_dword")

    def close():
        nonlocal run_len
        if cur is not None and run_len >= min_run:
            out.setdefault(short(cur), []).append((run_start + 1, run_len))
        run_len = 0

    for n, s in enumerate(code):
        m = _LABEL.match(s)
        if m and not m.group(1).startswith(".L"):
            close()
            cur = m.group(1)
            continue
        if not load.match(s) or "lds" in s:
            continue
        waited = False
        for k in range(n + 1, min(n + 8, len(code))):
            if load.match(code[k]):
                break
            if code[k].startswith("s_waitcnt") and "vmcnt(0)" in code[k]:
                waited = True
                break
        if not waited:
            continue
        if n - last_hit > 14:
            close()
            run_start = n
        run_len += 1
        last_hit = n
    close()
    return out


def scan_hot_loops(text: str):
    """-> {function: (number of 64-MFMA K-loop stage blocks, scratch accesses inside them)} for every k_chol_dag kernel
    and every dag_special<C, AUG> (the out-of-line routine that runs the following strip-solve tasks, K-loops included)"""
    lines = text.split("\n")
    out = {}
    i = 0
    while i < len(lines):
        m = re.match(r"^(_ZN5psoap(?:10k_chol_dag|11dag_special|11k_chol_solo)\w+):", lines[i])
        if not m:
            i += 1
            continue
        name = m.group(1)
        j = i
        while j < len(lines) and not lines[j].strip().startswith(".Lfunc_end"):
            j += 1
        cur, blocks = "entry", {}
        for ln in lines[i:j]:
            mm = re.match(r"^(\.LBB\d+_\d+):", ln)
            if mm:
                cur = mm.group(1)
            b = blocks.setdefault(cur, [0, 0])
            if "v_mfma" in ln:
                b[0] += 1
            if "scratch_load" in ln or "scratch_store" in ln:
                b[1] += 1
        hot = [v for v in blocks.values() if v[0] == 64]     # one K-loop stage = 64 MFMAs per wave
        out[short(name)] = (len(hot), sum(v[1] for v in hot))
        i = j
    return out


def kernel_resources(text: str):
    """-> {kernel: {"vgpr_spill_count": n, "private_segment_fixed_size": n, ...}} from the code-object metadata"""
    out = {}
    cur = {}
    for ln in text.split("\n"):
        m = re.match(r"\s+-?\s*\.(\w+):\s+(.*)$", ln)
        if not m:
            continue
        key, val = m.group(1), m.group(2).strip()
        if key in ("vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size", "vgpr_count", "agpr_count",
                   "group_segment_fixed_size"):
            cur[key] = int(val)
        elif key == "symbol":
            sym = val.strip("'\"")[:-3] if val.strip("'\"").endswith(".kd") else val.strip("'\"")
            cur["symbol"] = sym
        elif key == "name" and "symbol" not in cur:
            cur["name"] = val.strip("'\"")
        if key == "wavefront_size":      # last field of a kernel record in hipcc's emission order
            name = cur.get("symbol") or cur.get("name")
            if name:
                out[short(name)] = {k: v for k, v in cur.items() if k not in ("symbol", "name")}
            cur = {}
    return out
