"""GPU: parity at the FULL BASELINE.json shapes against outputs of the reference itself
(tests/golden/golden_full_v1.npz, made by tests/golden/make_golden_full.py):

* configs[4]: ``predict_f_g_h`` at N = 8192, M = 2 n_pix = 1024 (scripts/psoap_retrieve_ST3.py:97-109),
  through the reference-signature function and through the handle-resident C entry point;
* configs[3]: 32 walkers x 8 chunks at N = 6000 in ONE ChunkGroup launch (per-(chunk, walker) values of
  the reference for the first 4 walkers; the other 28 against single-chunk launches).
"""
import os

import numpy as np
import pytest

from psoap_amd import synthetic as syn

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LNP_RTOL = 1e-10      # |dlnp| <= 1e-10 max(1, |lnp|)
MU_ATOL = 1e-10       # predict mean, absolute
SIGMA_ATOL = 1e-9     # predict covariance, absolute (cancellation in A - W^T W)


@pytest.fixture(scope="module")
def full():
    return dict(np.load(os.path.join(ROOT, "tests", "golden", "golden_full_v1.npz")))


def _cfg5():
    ch = syn.make_config_chunk(5)
    M = 2 * ch.n_pix
    pred = np.linspace(np.min(ch.lwls[0]), np.max(ch.lwls[0]), num=M)
    return ch, M, pred


def _check_predict(full, mu, Sig, M):
    assert mu.shape == (3 * M,) and Sig.shape == (3 * M, 3 * M)
    assert np.max(np.abs(mu - full["cfg5_pred_mu"])) <= MU_ATOL
    assert np.max(np.abs(np.diag(Sig) - full["cfg5_pred_diag"])) <= SIGMA_ATOL
    rows = full["cfg5_pred_row_index"]
    assert np.max(np.abs(Sig[rows] - full["cfg5_pred_rows"])) <= SIGMA_ATOL
    assert np.array_equal(Sig, Sig.T)          # mirrored tiles: bitwise symmetric


def test_predict_cfg5_retrieve_shape_reference_signature(full):
    from psoap_amd import covariance
    ch, M, pred = _cfg5()
    assert (ch.N, M) == tuple(full["cfg5_pred_meta"][1:])
    mu, Sig = covariance.predict_f_g_h(*ch.lwls, ch.fl, ch.sigma, pred, pred, pred, 0.0, 0.0, 0.0, *syn.GP_BASE[3])
    _check_predict(full, mu, Sig, M)
    t = covariance.last_predict_timings()
    assert t["device_ms"] > 0.0 and t["flops"] > 4e11
    # second call on the kept workspace: identical bits, nothing re-allocated
    mu2, Sig2 = covariance.predict_f_g_h(*ch.lwls, ch.fl, ch.sigma, pred, pred, pred, 0.0, 0.0, 0.0, *syn.GP_BASE[3])
    assert np.array_equal(mu, mu2) and np.array_equal(Sig, Sig2)
    covariance.release_handles()


def test_predict_cfg5_handle_resident(full):
    from psoap_amd.chunk import ChunkHandle
    ch, M, pred = _cfg5()
    with ChunkHandle(ch.fl, ch.sigma, max_batch=1) as h:
        mu, Sig = h.predict(0, ch.lwls, np.stack([pred] * 3), np.zeros(3), syn.GP_BASE[3])
        _check_predict(full, mu, Sig, M)
        first = h.predict_timings()["total_ms"]
        mu_only = h.predict(0, ch.lwls, np.stack([pred] * 3), np.zeros(3), syn.GP_BASE[3], want_sigma=False)
        assert np.array_equal(mu_only, mu)
        assert h.predict_timings()["total_ms"] < first          # no allocation, no Sigma
        # the likelihood of the same handle is unaffected by the predict workspace
        lnp = h.lnlike(ch.lwls, syn.GP_BASE[3])
    g1 = dict(np.load(os.path.join(ROOT, "tests", "golden", "golden_v1.npz")))
    want = float(g1["lnlike_vals"][list(g1["lnlike_names"]).index("cfg5_st3_n8192")])
    assert abs(lnp - want) <= LNP_RTOL * max(1.0, abs(want))


def test_cfg4_eight_chunks_32_walkers_one_launch(full):
    from psoap_amd.chunk import ChunkGroup, ChunkHandle
    B = 32
    gf = full["cfg4_lnlike"]                                 # (8, 4) reference values
    gps = syn.make_walkers(2, B, seed=4500)
    chunks = [syn.make_config_chunk(4, k) for k in range(8)]
    lws = [syn.walker_lwls(ch, syn.make_walker_velocities(ch, B, seed=4501 + k)) for k, ch in enumerate(chunks)]
    handles = [ChunkHandle(ch.fl, ch.sigma, max_batch=B) for ch in chunks]
    try:
        with ChunkGroup(handles) as g:
            for h, lw in zip(handles, lws):
                h.upload(lw, gps)
            g.eval()
            table = np.stack([h.fetch() for h in handles])            # (8, 32)
        assert table.shape == (8, B)
        nw = gf.shape[1]
        assert np.all(np.abs(table[:, :nw] - gf) <= LNP_RTOL * np.maximum(1.0, np.abs(gf))), (table[:, :nw], gf)
        # walkers beyond the golden prefix: the same chunk evaluated alone (other split factors, same tolerance)
        for k in (0, 5):
            single = handles[k].lnlike_batch(lws[k], gps)
            assert np.all(np.abs(table[k] - single) <= LNP_RTOL * np.maximum(1.0, np.abs(single)))
        # the ensemble sum of the reference's gather (sample_parallel.py:387), fixed chunk order
        want = np.zeros(nw)
        got = np.zeros(B)
        for k in range(8):
            want = want + gf[k]
            got = got + table[k]
        assert np.all(np.abs(got[:nw] - want) <= LNP_RTOL * np.abs(want))
    finally:
        for h in handles:
            h.close()


def test_predict_cfg5_variance_only(full):
    """psoap_chunk_predict_var: mean and diag(Sigma) without forming Sigma -- what the retrieve scripts use of it
    (sqrt(diag(Sigma)), psoap_retrieve_ST3.py:111) -- against the reference's diagonal at the retrieve shape."""
    from psoap_amd.chunk import ChunkHandle
    ch, M, pred = _cfg5()
    with ChunkHandle(ch.fl, ch.sigma, max_batch=1) as h:
        mu, var = h.predict(0, ch.lwls, np.stack([pred] * 3), np.zeros(3), syn.GP_BASE[3], want_sigma="diag")
        assert mu.shape == (3 * M,) and var.shape == (3 * M,)
        assert np.max(np.abs(mu - full["cfg5_pred_mu"])) <= MU_ATOL
        assert np.max(np.abs(var - full["cfg5_pred_diag"])) <= SIGMA_ATOL
        t = h.predict_timings()
        assert t["download_ms"] < 1.0          # nothing R^2-sized travels
