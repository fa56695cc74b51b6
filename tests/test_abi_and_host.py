"""CPU-only checks of the boundary: the C-ABI library loads and exports every symbol the
header declares; host-side logic (sharding, argument checks, bench formulas)."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols(name="psoap_gp.h"):
    text = open(os.path.join(ROOT, "include", name)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(psoap_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from psoap_amd import build, _lib
    path = build.build()
    L = ctypes.CDLL(path)
    syms = _header_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(L, s), f"{s} declared in include/psoap_gp.h but not exported"
    # the ctypes table binds exactly the declared set
    assert sorted(_lib.SIGNATURES) == syms
    assert _lib.load().psoap_version() == 1


def test_bench_library_exports_every_declared_symbol_and_the_product_has_no_measurement_kernels():
    """include/psoap_bench.h <-> libpsoap_bench.so; the product library holds product entry points only."""
    from psoap_amd import build, _lib
    L = ctypes.CDLL(build.build_bench())
    syms = _header_symbols("psoap_bench.h")
    assert len(syms) >= 7 and sorted(_lib.BENCH_SIGNATURES) == syms
    for s in syms:
        assert hasattr(L, s), f"{s} declared in include/psoap_bench.h but not exported"
    P = ctypes.CDLL(build.build())
    assert not any(hasattr(P, s) for s in syms)
    assert not any("microbench" in s for s in _header_symbols())


def test_no_cpu_fallback_when_library_missing(monkeypatch, tmp_path):
    from psoap_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.PsoapError, match="no CPU fallback"):
        _lib.load()


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "psoap_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "psoap_oracle" not in src, f


def test_negative_hyperparameters_short_circuit_without_gpu():
    """covariance.py:317-318: -inf before any work -- so no device is touched."""
    from psoap_amd import covariance
    x = np.zeros(4)
    assert covariance.lnlike_f(None, x, x, x, -1.0, 5.0) == -np.inf
    assert covariance.lnlike_f_g(None, x, x, x, x, 0.1, 5.0, 0.1, -5.0) == -np.inf
    assert covariance.lnlike_f_g_h(None, x, x, x, x, x, 0.1, 5.0, 0.1, 5.0, -0.1, 5.0) == -np.inf
    assert set(covariance.lnlike) == {"SB1", "SB2", "ST1", "ST2", "ST3"}


def test_predict_assertions_match_reference_messages():
    from psoap_amd import covariance
    a, b = np.zeros(5), np.zeros(4)
    with pytest.raises(AssertionError, match="Input wavelengths must be the same length."):
        covariance.predict_f_g(a, b, a, a, a, a, 0, 1, 1, 0, 1, 1)
    with pytest.raises(AssertionError, match="Prediction wavelengths must be the same length."):
        covariance.predict_f_g(a, a, a, a, a, b, 0, 1, 1, 0, 1, 1)
    with pytest.raises(AssertionError, match="Prediction wavelengths must be the same length."):
        covariance.predict_f_g_h(a, a, a, a, a, a, a, b, 0, 0, 0, 1, 1, 1, 1, 1, 1)


def test_owned_chunks_partition():
    from psoap_amd.ensemble import owned_chunks
    for n, w in [(8, 1), (8, 2), (8, 8), (7, 3), (3, 2)]:
        seen = sorted(k for r in range(w) for k in owned_chunks(n, w, r))
        assert seen == list(range(n))
    assert owned_chunks(8, 4, 1) == [1, 5]


def test_bench_flop_formulas():
    import bench
    assert abs(bench.flops_eval(6000) - 7.2072e10) / 7.2072e10 < 1e-4
    N = 6000
    # the panel update is the bulk of N^3/3 and never exceeds it
    share = bench.flops_panel_update(N) / (N ** 3 / 3.0)
    assert 0.93 < share < 1.0


def test_synthetic_generator_is_deterministic():
    from psoap_amd import synthetic as syn
    a = syn.make_config_chunk(3, chunk_index=2)
    b = syn.make_config_chunk(3, chunk_index=2)
    assert a.N == 6000 and np.array_equal(a.fl, b.fl) and np.array_equal(a.lwls, b.lwls)
    assert not np.array_equal(a.fl, syn.make_config_chunk(3, chunk_index=3).fl)
    w = syn.make_walkers(2, 32, seed=1)
    assert w.shape == (32, 4) and np.all(w > 0) and np.allclose(w[0], syn.GP_BASE[2])
