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
