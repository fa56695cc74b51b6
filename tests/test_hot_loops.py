"""CPU (needs hipcc, no GPU): the 64-MFMA K-loop stages of every k_chol_dag instantiation must not touch scratch.
A spill reload inside a stage costs more than its latency -- the s_waitcnt vmcnt(0) behind it also waits for the
LDS-DMA of the next stage and serialises it with the MFMAs (measured: 39.5 -> 44.2 ms per 32-walker step) -- and
hipcc introduces one whenever the kernel around the loops grows in the wrong place."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="needs hipcc")
def test_no_spill_reloads_inside_mfma_loop_stages():
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_hot_loops.py")], capture_output=True, text=True,
                         timeout=900)
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("k_chol_dag<")]
    assert len(lines) == 12, res.stdout + res.stderr          # C = 1, 2, 3  x  AUG  x  LAT
    assert all(" 3 K-loop stage blocks, 0 scratch accesses" in ln for ln in lines), res.stdout
    assert res.returncode == 0


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="needs hipcc")
def test_library_is_built_from_the_sources_as_they_are():
    """The in-tree library travels to the GPU box as built here: it must come from the kernel sources in the tree
    (content hash kept beside it), not from an experiment built over them earlier."""
    sys.path.insert(0, ROOT)
    from psoap_amd import build
    build.build()                      # rebuilds only when the hash differs
    assert not build._stale()
    with open(build.HASH_PATH) as fh:
        assert fh.read().strip() == build.source_hash()
