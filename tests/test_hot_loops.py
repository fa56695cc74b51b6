"""CPU (needs hipcc, no GPU): checks on the device assembly of the library as built from the sources in the tree
(psoap_amd/asmcheck.py).

* The 64-MFMA K-loop stages of every k_chol_dag instantiation must not touch scratch.  A spill reload inside a stage
  costs more than its latency -- the s_waitcnt vmcnt(0) behind it also waits for the LDS-DMA of the next stage and
  serialises it with the MFMAs (measured: 39.5 -> 44.2 ms per 32-walker step) -- and hipcc introduces one whenever the
  kernel around the loops grows in the wrong place.
* No vector-register write ahead of an exec restore: the hipcc defect that made round 2's latency-scheme kernels fault
  on the GPU in about half of all builds.  psoap_amd.build refuses to install such a binary; here the
  scanner itself is pinned on the join block of the faulting build (profiles/r3_lat_fault_joinblock.s)."""
import json
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def _have_hipcc():
    from psoap_amd import build
    return shutil.which(build.hipcc()) is not None          # (the HIPCC variable, as the build itself resolves it)


needs_hipcc = pytest.mark.skipif(not _have_hipcc(), reason="needs hipcc")


@needs_hipcc
def test_library_is_built_from_the_sources_as_they_are():
    """The in-tree library travels to the GPU box as built here: it must come from the kernel sources in the tree, the
    compiler on this machine and no experimental flags (record kept beside it), not from an experiment built earlier."""
    from psoap_amd import build
    build.build()                      # rebuilds only when the record differs
    assert not build._stale()
    rec = build.build_record()
    assert rec["sources"] == build.source_hash() and rec["compiler"] == build.compiler_id()
    assert rec["extra_flags"] == []
    # the shipped sources must not need the inline fallback with the compiler of this image
    assert rec["fallback_rung"] == 0 and rec["rejected"] == [], rec


@needs_hipcc
def test_no_spill_reloads_inside_mfma_loop_stages():
    from psoap_amd import asmcheck, build
    res = asmcheck.scan_hot_loops(build.device_asm())
    kernels = {k: v for k, v in res.items() if k.startswith("k_chol_dag")}
    special = {k: v for k, v in res.items() if k.startswith("dag_special")}
    # C = 1, 2, 3  x  AUG  x  LAT, plus the streamed forms (round 4: C x LAT, never AUG) and the LAT kernels compiled for one
    # wave per SIMD (C x AUG; their out-of-line routine is an instantiation of its own)
    assert len(kernels) == 24 and len(special) == 12, res
    # 3 K-loop stage blocks per kernel, 5 in the out-of-line routine (its following update has two more): no scratch access
    assert all(v == (3, 0) for v in kernels.values()) and all(v == (5, 0) for v in special.values()), res


@needs_hipcc
def test_single_evaluation_kernels_keep_their_chain_phases_in_registers():
    """The LAT kernels that run single evaluations and predict (at most one workgroup per compute unit: compiled for one
    wave per SIMD, 512 registers per lane) spill no vector register and have no frame to speak of -- the argument record of
    the out-of-line routine travels through LDS, and the routine is compiled without callee-saved registers (it must not be
    tail-called for that: dag_kernel.hpp).  Code-object metadata; the bound asked for is 16 spilled VGPRs / 256 B."""
    from psoap_amd import asmcheck, build
    res = asmcheck.kernel_resources(build.device_asm())
    wide = {k: v for k, v in res.items() if k.startswith("k_chol_dag") and k.endswith(",wide>")}
    assert len(wide) == 6, sorted(res)
    for k, v in wide.items():
        assert v["vgpr_spill_count"] <= 16 and v["private_segment_fixed_size"] <= 256, (k, v)
        assert v["agpr_count"] > 128, (k, v)          # the accumulator tile lives in the AccVGPR half
    # the kernels for two workgroups per compute unit have 256 registers: their chain phases do spill (recorded, not bounded)
    narrow = {k: v for k, v in res.items() if k.startswith("k_chol_dag") and not k.endswith(",wide>")}
    assert all(v["agpr_count"] == 0 and v["vgpr_count"] == 256 for v in narrow.values()), narrow
    # ... but the headline kernels (throughput scheme, SB1 / SB2) stay all but clean (none of it inside a K-loop stage: the
    # test above; late round 5: 8 VGPRs / 36 B in the SB2 batch kernel, with which it runs 1.1 % FASTER than with 4 / 20 --
    # profiles/r5_handover.txt)
    for k in ("k_chol_dag<1,false,false>", "k_chol_dag<2,false,false>", "k_chol_dag<1,false,false,stream>", "k_chol_dag<2,false,false,stream>"):
        assert narrow[k]["vgpr_spill_count"] <= 8 and narrow[k]["private_segment_fixed_size"] <= 48, (k, narrow[k])


@needs_hipcc
def test_no_vector_write_ahead_of_an_exec_restore():
    from psoap_amd import asmcheck, build
    hits = asmcheck.scan_exec_restore(build.device_asm())
    assert hits == [], [(asmcheck.short(h[0]), h[1], h[3][:3]) for h in hits]


@needs_hipcc
def test_build_falls_back_when_the_default_flags_show_the_defect(tmp_path, monkeypatch):
    """The build's flag ladder: a binary whose assembly shows the defect is rejected and the next rung (diagonal routine
    inlined: no call, nothing parked around one) is taken and recorded; if every rung shows it, the build fails.  Which
    source shapes make THIS compiler produce the defect changes with every edit of the kernels (the 68-build matrix in
    profiles/ is the record of that), so the ladder is driven here by a scanner that reports the real faulting join block
    for the first rung only."""
    from psoap_amd import asmcheck, build
    with open(os.path.join(ROOT, "profiles", "r3_lat_fault_joinblock.s")) as fh:
        faulty = "_ZN5psoap10k_chol_dagILi3ELb1ELb1EEEv:\n\ts_cbranch_execz .LBB46_1318\n" + fh.read()
    calls = []

    def scan_first_rung_faulty(text):
        calls.append(len(text))
        return asmcheck.scan_exec_restore(faulty if len(calls) == 1 else text)

    # (rung 0 is the library in the tree -- built from these sources with these flags, test above -- so only the fallback
    # rung is compiled here: it has to keep compiling, and one compilation of the library is two and a half minutes)
    build.build()
    real_compile = build.compile_once

    def compile_rung(flags, out_dir):
        if flags == []:
            return build.LIB_PATH, build.ASM_PATH
        return real_compile(flags, out_dir)

    monkeypatch.setattr(build, "compile_once", compile_rung)
    so, asm, rec = build.compile_checked([], str(tmp_path), scan=scan_first_rung_faulty)
    assert len(calls) == 2 and rec["fallback_rung"] == 1
    assert rec["rejected"] == [{"flags": [], "kernels": ["k_chol_dag<3,true,true>"]}]
    assert os.path.exists(so) and "-DPSOAP_DIAG_INLINE" in rec["flags"] and "-DPSOAP_NO_FOLLOW" in rec["flags"]
    monkeypatch.setattr(build, "compile_once", lambda flags, out_dir: (so, asm))      # no third and fourth compile
    with pytest.raises(build.BuildError, match="every flag set"):
        build.compile_checked([], str(tmp_path / "all"), scan=lambda text: asmcheck.scan_exec_restore(faulty))


def test_no_tile_is_read_one_round_trip_per_element():
    """No kernel of the library reads a tile as a run of load / `s_waitcnt vmcnt(0)` pairs (asmcheck.scan_serial_loads): the
    shape hipcc gives a plain loop over a tile behind a K-loop, 27 us per partial tile until round 5.  The
    scanner is checked on that shape itself."""
    from psoap_amd import asmcheck, build
    build.build()
    with open(build.ASM_PATH) as fh:
        assert asmcheck.scan_serial_loads(fh.read()) == {}
    pair = "\tglobal_load_dwordx2 v[0:1], v[2:3], off\n\ts_waitcnt vmcnt(0)\n\tv_add_f64 v[4:5], v[4:5], -v[0:1]\n"
    assert asmcheck.scan_serial_loads("_Zk:\n" + 20 * pair) == {"_Zk": [(2, 20)]}
    batch = 16 * "\tglobal_load_dwordx2 v[0:1], v[2:3], off\n" + "\ts_waitcnt vmcnt(0)\n"
    assert asmcheck.scan_serial_loads("_Zk:\n" + 4 * batch) == {}


def test_exec_restore_scanner_tells_a_narrowed_mask_from_a_restored_one():
    """`s_mov_b64 exec, s[A:B]` re-enables lanes only when s[A:B] is a saved mask; hipcc also spells the START of a masked
    region that way (copy exec, and it with the condition, move it back): no lane comes back there, and the vector writes
    ahead of it ran under the mask the block was entered with."""
    from psoap_amd import asmcheck
    head = "_Zf:\n\ts_cbranch_execz .LBB0_1\n.LBB0_1:\n\tv_mov_b32 v1, v2\n"
    narrowed = head + "\ts_mov_b64 s[0:1], exec\n\ts_and_b64 s[14:15], s[0:1], s[14:15]\n\ts_mov_b64 exec, s[14:15]\n"
    restored = head + "\ts_mov_b64 exec, s[14:15]\n"
    rewritten = head + "\ts_mov_b64 s[0:1], exec\n\ts_and_b64 s[14:15], s[0:1], s[14:15]\n\ts_mov_b64 s[14:15], s[2:3]\n\ts_mov_b64 exec, s[14:15]\n"
    assert asmcheck.scan_exec_restore(narrowed) == []
    assert len(asmcheck.scan_exec_restore(restored)) == 1 and len(asmcheck.scan_exec_restore(rewritten)) == 1


def test_exec_restore_scanner_flags_the_faulting_join_block():
    """The join block of k_chol_dag<3, true, true> from the build that faulted on the GPU (rocgdb session in
    profiles/r3_lat_fault_rocgdb.txt): ten vector writes -- `v_mov_b32 v172, v244` among them -- ahead of the
    `s_or_b64 exec, exec, s[0:1]`."""
    from psoap_amd import asmcheck
    with open(os.path.join(ROOT, "profiles", "r3_lat_fault_joinblock.s")) as fh:
        block = fh.read()
    text = "_ZN5psoap10k_chol_dagILi3ELb1ELb1EEEv:\n\ts_cbranch_execz .LBB46_1318\n" + block
    hits = asmcheck.scan_exec_restore(text)
    assert len(hits) == 1 and hits[0][1] == ".LBB46_1318"
    assert any("v_mov_b32_e32 v172, v244" in ins for _, ins in hits[0][3])
    assert asmcheck.short(hits[0][0]) == "k_chol_dag<3,true,true>"
    # the same block with the restore in front is clean; v_writelane (an SGPR spill into one lane) ignores exec
    lines = block.split("\n")
    k = next(i for i, ln in enumerate(lines) if "s_or_b64 exec, exec" in ln)
    first = next(i for i, ln in enumerate(lines) if ln.startswith(".LBB46_1318:"))
    fixed = lines[:first + 1] + [lines[k], "\tv_writelane_b32 v254, s72, 57"] + lines[first + 1:k] + lines[k + 1:]
    assert asmcheck.scan_exec_restore("f:\n\ts_cbranch_execz .LBB46_1318\n" + "\n".join(fixed)) == []
    bad = lines[:first + 1] + ["\tv_writelane_b32 v254, s72, 57", lines[k]]
    assert asmcheck.scan_exec_restore("f:\n\ts_cbranch_execz .LBB46_1318\n" + "\n".join(bad)) == []


def test_variant_matrix_predictions_match_the_gpu_outcomes():
    """profiles/r3_lat_variant_matrix.jsonl: 30 builds, detector verdict (CPU) against what the GPU did."""
    rows = [json.loads(ln) for ln in open(os.path.join(ROOT, "profiles", "r3_lat_variant_matrix.jsonl"))]
    assert len(rows) >= 30
    for d in rows:
        predicted_fault = bool(d["detector_hits"])
        faulted = d["check_rc"] != 0 or d["repro_rc"] != 0
        assert predicted_fault == faulted, d["name"]
    assert sum(bool(d["detector_hits"]) for d in rows) >= 7
