"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle and the
golden vectors generated from the reference.  Run with `-m gpu` on an MI355X."""
import os

import numpy as np
import pytest

from psoap_amd import synthetic as syn

pytestmark = pytest.mark.gpu

# tolerance contract: |dlnp| <= 1e-10 * max(1, |lnp|); fills <= 4 ulp
LNP_RTOL = 1e-10
FILL_ULP = 4


def ulp_diff(a, b):
    a = np.ascontiguousarray(a, dtype=np.float64).view(np.int64)
    b = np.ascontiguousarray(b, dtype=np.float64).view(np.int64)
    return np.abs(a - b)


def lnp_close(got, want):
    return abs(got - want) <= LNP_RTOL * max(1.0, abs(want))


@pytest.fixture(scope="module")
def mf():
    from psoap_amd import matrix_functions
    return matrix_functions


@pytest.fixture(scope="module")
def cov():
    from psoap_amd import covariance
    yield covariance
    covariance.release_handles()


# ------------------------------------------------------------------------------ fills
def test_fill_golden(golden, mf):
    c, ne, npx, seed, M = golden["fill_meta"]
    ch = syn.make_chunk(c, ne, npx, seed=seed)
    N = ch.N
    gp = syn.GP_BASE[3]
    m = np.empty((N, N))
    mf.fill_V11_f(m, ch.lwls[0], *gp[:2])
    assert ulp_diff(m, golden["fill_f"]).max() <= FILL_ULP
    assert np.array_equal(m, m.T)
    mf.fill_V11_f_g(m, ch.lwls[0], ch.lwls[1], *gp[:4])
    assert ulp_diff(m, golden["fill_f_g"]).max() <= FILL_ULP
    mf.fill_V11_f_g_h(m, *ch.lwls, *gp)
    assert ulp_diff(m, golden["fill_f_g_h"]).max() <= FILL_ULP
    assert np.array_equal(np.diag(m), np.diag(golden["fill_f_g_h"]))   # diagonal rule is exact
    pred = np.linspace(ch.lwls[0].min(), ch.lwls[0].max(), M)
    r = np.empty((M, N))
    mf.fill_V12_f(r, pred, ch.lwls[0], *gp[:2])
    assert ulp_diff(r, golden["fill_cross_40xN"]).max() <= FILL_ULP
    r = np.empty((N, M))
    mf.fill_V12_f(r, ch.lwls[1], pred, *gp[2:4])
    assert ulp_diff(r, golden["fill_cross_Nx40"]).max() <= FILL_ULP


@pytest.mark.parametrize("c,ne,npx", [(1, 3, 43), (2, 5, 77), (3, 9, 100), (2, 1, 1), (1, 2, 64)])
def test_fill_vs_oracle_ragged(oracle, mf, c, ne, npx):
    ch = syn.make_chunk(c, ne, npx, seed=900 + npx, masked_fraction=0.1 if npx > 10 else 0.0)
    N = ch.N
    gp = syn.GP_BASE[c]
    want = np.empty((N, N))
    oracle.fill_sym(want, ch.lwls, gp)
    got = np.full((N, N), np.nan)
    [mf.fill_V11_f, mf.fill_V11_f_g, mf.fill_V11_f_g_h][c - 1](got, *ch.lwls, *gp)
    assert ulp_diff(got, want).max() <= FILL_ULP
    # rectangular, odd sizes, non-contiguous target
    M = max(1, N // 3)
    pred = np.linspace(ch.lwls[0].min(), ch.lwls[0].max(), M)
    want = np.empty((M, N))
    oracle.fill_V12_f(want, pred, ch.lwls[0], gp[0], gp[1])
    big = np.full((M, 2 * N), np.nan)
    view = big[:, ::2]
    mf.fill_V12_f(view, pred, ch.lwls[0], gp[0], gp[1])
    assert ulp_diff(np.ascontiguousarray(view), want).max() <= FILL_ULP


def test_fill_mostly_bit_exact(oracle, mf):
    """The arithmetic around exp() is contraction-free, so the only difference from the
    reference is the device exp(): the large majority of entries must match bit for bit."""
    ch = syn.make_chunk(2, 6, 100, seed=77)
    want = np.empty((ch.N, ch.N))
    oracle.fill_sym(want, ch.lwls, syn.GP_BASE[2])
    got = np.empty_like(want)
    mf.fill_V11_f_g(got, *ch.lwls, *syn.GP_BASE[2])
    d = ulp_diff(got, want)
    assert d.max() <= FILL_ULP
    assert (d == 0).mean() > 0.5


# ------------------------------------------------------------------------------ lnlike
def _cases(golden):
    for name, meta, val in zip(golden["lnlike_names"], golden["lnlike_meta"], golden["lnlike_vals"]):
        c, ne, npx, seed, mf100, N = [int(x) for x in meta]
        yield str(name), c, ne, npx, seed, mf100 / 100.0, N, float(val)


def test_lnlike_golden_all_configs(golden, cov):
    V11 = None
    for name, c, ne, npx, seed, mfrac, N, val in _cases(golden):
        ch = syn.make_chunk(c, ne, npx, seed=seed, masked_fraction=mfrac)
        fn = [cov.lnlike_f, cov.lnlike_f_g, cov.lnlike_f_g_h][c - 1]
        got = fn(V11, *ch.lwls, ch.fl, ch.sigma, *syn.GP_BASE[c])
        assert lnp_close(got, val), (name, got, val, got - val)
        cov.release_handles()


def test_lnlike_dispatch_and_conventions(golden, cov):
    ch = syn.make_chunk(2, 8, 32, seed=105)
    assert cov.lnlike["SB2"] is cov.lnlike_f_g and cov.lnlike["ST2"] is cov.lnlike_f_g
    assert cov.lnlike["SB1"] is cov.lnlike_f and cov.lnlike["ST1"] is cov.lnlike_f
    assert cov.lnlike["ST3"] is cov.lnlike_f_g_h
    V = np.empty((ch.N, ch.N))
    v = cov.lnlike["SB2"](V, *ch.lwls, ch.fl, ch.sigma, *syn.GP_BASE[2], mu_GP=0.9)
    assert lnp_close(v, float(golden["lnlike_mu0p9"]))
    assert cov.lnlike_f(V, ch.lwls[0], ch.fl, ch.sigma, -0.2, 5.0) == -np.inf
    assert cov.lnlike_f_g(V, *ch.lwls, ch.fl, ch.sigma, 0.2, 5.0, 0.1, -7.0) == -np.inf
    assert cov.lnlike_f_g_h(V, *ch.lwls, ch.lwls[0], ch.fl, ch.sigma, 0.2, 5.0, 0.1, 7.0, -0.05, 6.0) == -np.inf
    lw = ch.lwls.copy()
    lw[:, 1] = lw[:, 0]
    assert cov.lnlike_f_g(V, *lw, ch.fl, np.zeros_like(ch.sigma), *syn.GP_BASE[2]) == -np.inf   # not PD
    v = cov.lnlike_f(V, ch.lwls[0], ch.fl, ch.sigma, 0.0, 5.0)
    assert lnp_close(v, float(golden["lnlike_zero_amp"]))


@pytest.mark.parametrize("N", [1, 2, 63, 127, 128, 129, 255, 257, 640])
def test_lnlike_vs_oracle_edge_sizes(oracle, N):
    from psoap_amd.chunk import ChunkHandle
    ch = syn.make_chunk(2, 1, N, seed=4000 + N)
    with ChunkHandle(ch.fl, ch.sigma, max_batch=1) as h:
        got = h.lnlike(ch.lwls, syn.GP_BASE[2])
    want = oracle.lnlike(ch.lwls, ch.fl, ch.sigma, syn.GP_BASE[2])
    assert lnp_close(got, want), (N, got, want)


def test_batch_matches_singles_and_oracle(golden, oracle):
    from psoap_amd.chunk import ChunkHandle
    ch = syn.make_chunk(2, 6, 100, seed=321)          # N = 600
    nw = 7
    gps = syn.make_walkers(2, nw, seed=11)
    gps[3, 1] = -1.0                                  # one rejected proposal inside the batch
    lw = syn.walker_lwls(ch, syn.make_walker_velocities(ch, nw, seed=12))
    with ChunkHandle(ch.fl, ch.sigma, max_batch=8) as h:
        results = {}
        for mode, groups in (("dag", 1), ("staged", 1), ("staged", 2), ("staged", 3)):
            h.set_mode(mode)
            h.set_stream_groups(groups)
            got = h.lnlike_batch(lw, gps)
            results[(mode, groups)] = got
            for w in range(nw):
                want = oracle.lnlike(lw[w], ch.fl, ch.sigma, gps[w])
                if w == 3:
                    assert got[w] == -np.inf and want == -np.inf
                else:
                    assert lnp_close(got[w], want), (mode, groups, w, got[w], want)
            # determinism: same inputs, same bits, every time
            for _ in range(3):
                assert np.array_equal(h.lnlike_batch(lw, gps), got)
        # staged results do not depend on the stream grouping
        assert np.array_equal(results[("staged", 1)], results[("staged", 3)])
        # profiling mode (per-launch events) gives the same numbers
        h.set_mode("staged")
        h.set_profiling(True)
        prof = h.lnlike_batch(lw, gps)
        assert np.array_equal(prof, results[("staged", 1)])
        t = h.timings()
        assert t["panel_update"]["launches"] == 4 and t["potrf"]["launches"] == 5
        h.set_mode("dag")
        prof = h.lnlike_batch(lw, gps)
        assert np.array_equal(prof, results[("dag", 1)])
        assert h.timings()["dag"]["launches"] == 1
        h.set_profiling(False)


@pytest.mark.gpu
@pytest.mark.parametrize("nw", [9, 12, 13, 17, 25])
def test_batches_that_are_not_a_multiple_of_eight(oracle, nw):
    """dag_queue_count: 9, 13, 17 and 25 matrices go through ONE ticket queue, 12 through four -- every value against the
    oracle, the same bits on every repetition, and the same values (parity tolerance) whichever number of queues is forced."""
    from psoap_amd.chunk import ChunkHandle
    ch = syn.make_chunk(2, 7, 100, seed=77)           # N = 700: 6 block rows
    gps = syn.make_walkers(2, nw, seed=5)
    gps[nw // 2, 2] = -0.5                            # one rejected proposal inside the batch
    lw = syn.walker_lwls(ch, syn.make_walker_velocities(ch, nw, seed=6))
    want = np.array([oracle.lnlike(lw[w], ch.fl, ch.sigma, gps[w]) for w in range(nw)])
    got = {}
    try:
        for nq in ("", "8", "4", "2", "1"):
            if nq:
                os.environ["PSOAP_DAG_QUEUES"] = nq
            else:
                os.environ.pop("PSOAP_DAG_QUEUES", None)
            with ChunkHandle(ch.fl, ch.sigma, max_batch=nw) as h:
                got[nq] = h.lnlike_batch(lw, gps)
                for _ in range(3):
                    assert np.array_equal(h.lnlike_batch(lw, gps), got[nq])
            assert got[nq][nw // 2] == -np.inf and want[nw // 2] == -np.inf
            for w in range(nw):
                assert w == nw // 2 or lnp_close(got[nq][w], want[w]), (nq, w, got[nq][w], want[w])
    finally:
        os.environ.pop("PSOAP_DAG_QUEUES", None)


@pytest.mark.gpu
@pytest.mark.parametrize("c,ne,npx,nw", [(1, 8, 250, 1), (2, 10, 300, 1), (3, 4, 128, 1), (1, 4, 128, 4), (2, 6, 100, 8)])
def test_single_evaluation_kernels_give_the_same_bits_in_512_and_in_256_registers(c, ne, npx, nw):
    """Launches with at most one workgroup per compute unit run the LAT kernels compiled for one wave per SIMD
    (k_chol_dag<.., WPE = 1>); PSOAP_DAG_WIDE=0 keeps them on the 256-register forms: the same source, the same arithmetic,
    bit-identical results."""
    from psoap_amd.chunk import ChunkHandle
    ch = syn.make_chunk(c, ne, npx, seed=31 + c)
    gps = syn.make_walkers(c, nw, seed=9)
    lw = syn.walker_lwls(ch, syn.make_walker_velocities(ch, nw, seed=10))
    got = {}
    try:
        for wide in ("1", "0"):
            os.environ["PSOAP_DAG_WIDE"] = wide
            with ChunkHandle(ch.fl, ch.sigma, max_batch=nw) as h:
                got[wide] = h.lnlike_batch(lw, gps)
                assert np.array_equal(h.lnlike_batch(lw, gps), got[wide])
    finally:
        os.environ.pop("PSOAP_DAG_WIDE", None)
    assert np.all(np.isfinite(got["1"])) and np.array_equal(got["1"], got["0"])


@pytest.mark.parametrize("mode", ["dag", "staged"])
def test_modes_full_size_batch(golden, mode):
    """Both execution modes at BASELINE config 3 with a batch that oversubscribes the persistent grid."""
    from psoap_amd.chunk import ChunkHandle
    ch = syn.make_config_chunk(3)
    gps = syn.make_walkers(2, 4, seed=3500)
    lw = syn.walker_lwls(ch, syn.make_walker_velocities(ch, 4, seed=3501))
    reps = 3
    with ChunkHandle(ch.fl, ch.sigma, max_batch=4 * reps) as h:
        h.set_mode(mode)
        got = h.lnlike_batch(np.concatenate([lw] * reps), np.concatenate([gps] * reps))
        for w in range(4 * reps):
            assert lnp_close(got[w], golden["walkers_cfg3"][w % 4]), (mode, w, got[w])
        # identical proposals in different batch slots give identical bits
        assert np.array_equal(got[:4], got[4:8]) and np.array_equal(got[:4], got[8:])


def test_walker_batch_cfg3_golden(golden):
    """BASELINE config 3 (SB2, N=6000): 4 walkers against values from the reference itself."""
    from psoap_amd.chunk import ChunkHandle
    ch = syn.make_config_chunk(3)
    gps = syn.make_walkers(2, 4, seed=3500)
    vels = syn.make_walker_velocities(ch, 4, seed=3501)
    lw = syn.walker_lwls(ch, vels)
    with ChunkHandle(ch.fl, ch.sigma, max_batch=4) as h:
        got = h.lnlike_batch(lw, gps)
        for w in range(4):
            assert lnp_close(got[w], golden["walkers_cfg3"][w]), (w, got[w], golden["walkers_cfg3"][w])
        # device-side Doppler shift (replicate_wls on the GPU) must give the same answers
        h.set_grid(ch.lwl, ch.epoch_index, ch.n_epochs)
        h.upload_velocities(vels, gps)
        h.eval()
        got2 = h.fetch()
        assert np.array_equal(got2, got)


def test_full_size_properties():
    """Size-independent properties at the metric's full size (N=6000, SB2):
    (1) K = sigma^2 I (zero amplitudes) has the closed form -0.5*(|r|^2/s^2 + N log s^2);
    (2) scaling fl-mu, sigma and amp by a changes lnp by exactly -N log a;
    (3) permuting the pixels (a symmetric permutation of K) leaves lnp unchanged."""
    from psoap_amd.chunk import ChunkHandle
    ch = syn.make_config_chunk(3)
    N = ch.N
    gp = np.array(syn.GP_BASE[2])
    r = ch.fl - 1.0
    with ChunkHandle(ch.fl, ch.sigma, max_batch=2) as h:
        base = h.lnlike(ch.lwls, gp)
        zero = h.lnlike(ch.lwls, [0.0, 5.0, 0.0, 7.0])
        want0 = -0.5 * (np.sum(r * r / ch.sigma**2) + np.sum(np.log(ch.sigma**2)))
        assert abs(zero - want0) <= 1e-10 * abs(want0)
        a = 2.0   # power of two: the scaled problem is the same problem bit for bit
        h.set_data(1.0 + a * r, a * ch.sigma)
        scaled = h.lnlike(ch.lwls, gp * np.array([a, 1.0, a, 1.0]))
        assert abs(scaled - (base - N * np.log(a))) <= 1e-10 * abs(base)
        perm = np.random.default_rng(5).permutation(N)
        h.set_data(ch.fl[perm], ch.sigma[perm])
        permuted = h.lnlike(np.ascontiguousarray(ch.lwls[:, perm]), gp)
        assert abs(permuted - base) <= 1e-10 * abs(base)


# ------------------------------------------------------------------------------ predict
def test_predict_golden_small(golden, cov):
    c, ne, npx, seed, M, c2, ne2, npx2, seed2 = golden["pred_meta"]
    ch = syn.make_chunk(c, ne, npx, seed=seed)
    pg = np.linspace(ch.lwls[0].min(), ch.lwls[0].max(), M)
    mu, Sig = cov.predict_f_g(ch.lwls[0], ch.lwls[1], ch.fl, ch.sigma, pg, pg, 0.0, 0.2, 5.0, 0.0, 0.1, 7.0)
    np.testing.assert_allclose(mu, golden["pred_fg_mu"], rtol=0, atol=1e-10)
    np.testing.assert_allclose(Sig, golden["pred_fg_Sigma"], rtol=0, atol=1e-9)
    mu = cov.predict_f_g(ch.lwls[0], ch.lwls[1], ch.fl, ch.sigma, pg, pg + 1e-5, 0.3, 0.2, 5.0, 0.7, 0.1, 7.0,
                         get_Sigma=False)
    np.testing.assert_allclose(mu, golden["pred_fg_mu_only"], rtol=0, atol=1e-10)
    mu, Sig = cov.predict_f_g_h(*ch.lwls, ch.fl, ch.sigma, pg, pg, pg, 0.0, 0.0, 0.0, *syn.GP_BASE[3])
    np.testing.assert_allclose(mu, golden["pred_fgh_mu"], rtol=0, atol=1e-10)
    np.testing.assert_allclose(Sig, golden["pred_fgh_Sigma"], rtol=0, atol=1e-9)
    mu, Sig = cov.predict_f_g_sum(ch.lwls[0], ch.lwls[1], ch.fl, ch.sigma, pg, pg, 1.0, 0.2, 5.0, 0.1, 7.0)
    np.testing.assert_allclose(mu, golden["pred_fg_sum_mu"], rtol=0, atol=1e-10)
    np.testing.assert_allclose(Sig, golden["pred_fg_sum_Sigma"], rtol=0, atol=1e-9)
    chq = syn.make_chunk(c2, ne2, npx2, seed=seed2)
    mu, Sig = cov.predict_f_g_h_sum(*chq.lwls, chq.fl, chq.sigma, *chq.lwls, 1.0, *syn.GP_BASE[3])
    np.testing.assert_allclose(mu, golden["pred_fgh_sum_mu"], rtol=0, atol=1e-10)
    np.testing.assert_allclose(Sig, golden["pred_fgh_sum_Sigma"], rtol=0, atol=1e-9)


def test_predict_golden_retrieve_shape(golden, cov):
    c, ne, npx, seed, M = golden["predL_meta"]
    ch = syn.make_chunk(c, ne, npx, seed=seed)
    pg = np.linspace(ch.lwls[0].min(), ch.lwls[0].max(), M)
    mu, Sig = cov.predict_f_g_h(*ch.lwls, ch.fl, ch.sigma, pg, pg, pg, 0.0, 0.0, 0.0, *syn.GP_BASE[3])
    np.testing.assert_allclose(mu, golden["predL_fgh_mu"], rtol=0, atol=1e-10)
    np.testing.assert_allclose(np.diag(Sig), golden["predL_fgh_diag"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(Sig[[0, 399, 400, 777, 1199]], golden["predL_fgh_rows"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(Sig, Sig.T, rtol=0, atol=1e-12)
    mu, Sig = cov.predict_f_g(ch.lwls[0], ch.lwls[1], ch.fl, ch.sigma, pg, pg, 0.0, 0.2, 5.0, 0.0, 0.1, 7.0)
    np.testing.assert_allclose(mu, golden["predL_fg_mu"], rtol=0, atol=1e-10)
    np.testing.assert_allclose(np.diag(Sig), golden["predL_fg_diag"], rtol=0, atol=1e-9)


def test_predict_sum_transposed_mean_and_predict_f(oracle, cov):
    """predict_f_g_h_sum with M == N but *different* grids exercises the V12.T of
    covariance.py:294; predict_f is checked against the oracle's c=1 conditional."""
    ch = syn.make_chunk(3, 3, 50, seed=31)              # N = 150
    pred = [w + 0.3 * 2.7 / syn.C_KMS for w in ch.lwls]
    mu, Sig = cov.predict_f_g_h_sum(*ch.lwls, ch.fl, ch.sigma, *pred, 1.0, *syn.GP_BASE[3])
    mu_o, Sig_o = oracle.predict_sum(ch.lwls, ch.fl, ch.sigma, pred, 1.0, syn.GP_BASE[3])
    np.testing.assert_allclose(mu, mu_o, rtol=0, atol=1e-10)
    np.testing.assert_allclose(Sig, Sig_o, rtol=0, atol=1e-9)
    with pytest.raises(ValueError):
        cov.predict_f_g_h_sum(*ch.lwls, ch.fl, ch.sigma, *[p[:100] for p in pred], 1.0, *syn.GP_BASE[3])
    pg = np.linspace(ch.lwls[0].min(), ch.lwls[0].max(), 77)
    mu, Sig = cov.predict_f(ch.lwls[0], ch.fl, ch.sigma, pg, 0.2, 5.0, mu_GP=1.0)
    # c=1 joint conditional with offset 1.0 and prior mean 1.0 is predict_f with mu_GP = 1
    mu_o, Sig_o = oracle.predict_components(ch.lwls[:1], ch.fl, ch.sigma, [pg], [1.0], syn.GP_BASE[1])
    np.testing.assert_allclose(mu, mu_o, rtol=0, atol=1e-10)
    np.testing.assert_allclose(Sig, Sig_o, rtol=0, atol=1e-9)
    with pytest.raises(AssertionError, match="Input wavelengths must be the same length."):
        cov.predict_f_g(ch.lwls[0], ch.lwls[1][:-1], ch.fl, ch.sigma, pg, pg, 0.0, 0.2, 5.0, 0.0, 0.1, 7.0)


def test_batched_exp_is_bit_identical_to_library_exp():
    """The fused-fill epilogue evaluates exp() four at a time with the library's own operation sequence;
    every non-positive argument (normal, denormal result, underflow, -0, huge) and NaN must give the same bits."""
    import ctypes
    from psoap_amd import _lib
    rng = np.random.default_rng(77)
    x = np.concatenate([
        -rng.uniform(0.0, 760.0, 400000),                 # the whole useful range incl. denormal results
        -np.exp(rng.uniform(-40.0, 7.0, 200000)),         # log-uniform magnitudes down to 1e-18
        -rng.uniform(700.0, 1100.0, 50000),               # around the underflow thresholds
        np.array([0.0, -0.0, -745.1332191019411, -745.1332191019412, -746.0, -1074.9, -1075.0, -1075.1,
                  -1e10, -1e300, -np.inf, np.nan, -5e-324, -2.2250738585072014e-308]),
    ])
    x = np.ascontiguousarray(x[: len(x) - len(x) % 4])
    bad = ctypes.c_longlong(-1)
    _lib.check_bench(_lib.load_bench().psoap_microbench_exp_check(_lib.default_device(), len(x), _lib.dptr(x), ctypes.byref(bad)),
               "psoap_microbench_exp_check")
    assert bad.value == 0


@pytest.mark.parametrize("c,ne,npx,M,seed", [
    (1, 2, 37, 1, 51),        # N = 74 (below one tile), a single prediction point
    (2, 3, 43, 5, 52),        # N = 129 (one past a tile edge), M far below a tile
    (2, 5, 60, 130, 53),      # M one past a tile edge
    (3, 4, 70, 129, 54),      # c M = 387 prediction columns: four column tiles, ragged last one
    (3, 9, 100, 16, 55),      # N = 900 (ragged), masked epochs
])
def test_predict_edge_shapes_vs_oracle(oracle, cov, c, ne, npx, M, seed):
    """The predict family factors [B | Cx^T] inside the persistent kernel with the prediction columns as
    extra column tiles; shapes that leave those tiles ragged, for the joint and the summed conditionals."""
    ch = syn.make_chunk(c, ne, npx, seed=seed, masked_fraction=0.1 if npx >= 60 else 0.0)
    lo, hi = ch.lwls[0].min(), ch.lwls[0].max()
    pred = [np.linspace(lo, hi, M) + k * 1e-6 for k in range(c)] if M > 1 else [np.array([0.5 * (lo + hi)])] * c
    gp = syn.GP_BASE[c]
    mus = [0.1 * k for k in range(c)]
    if c == 1:
        mu, Sig = cov.predict_f(ch.lwls[0], ch.fl, ch.sigma, pred[0], *gp, mu_GP=1.0)
        mu_o, Sig_o = oracle.predict_components(ch.lwls, ch.fl, ch.sigma, pred, [1.0], gp)
    elif c == 2:
        mu, Sig = cov.predict_f_g(ch.lwls[0], ch.lwls[1], ch.fl, ch.sigma, pred[0], pred[1], mus[0], gp[0], gp[1],
                                  mus[1], gp[2], gp[3])
        mu_o, Sig_o = oracle.predict_components(ch.lwls, ch.fl, ch.sigma, pred, mus, gp)
        mu_s, Sig_s = cov.predict_f_g_sum(ch.lwls[0], ch.lwls[1], ch.fl, ch.sigma, pred[0], pred[1], 0.9, *gp)
        mu_so, Sig_so = oracle.predict_sum(ch.lwls, ch.fl, ch.sigma, pred, 0.9, gp)
        np.testing.assert_allclose(mu_s, mu_so, rtol=0, atol=1e-10)
        np.testing.assert_allclose(Sig_s, Sig_so, rtol=0, atol=1e-9)
    else:
        mu, Sig = cov.predict_f_g_h(*ch.lwls, ch.fl, ch.sigma, *pred, *mus, *gp)
        mu_o, Sig_o = oracle.predict_components(ch.lwls, ch.fl, ch.sigma, pred, mus, gp)
    assert mu.shape == (c * M,) and Sig.shape == (c * M, c * M)
    np.testing.assert_allclose(mu, mu_o, rtol=0, atol=1e-10)
    np.testing.assert_allclose(Sig, Sig_o, rtol=0, atol=1e-9)


def test_non_finite_data_conventions(cov):
    """Degenerate inputs (NaN / inf in flux, uncertainties, wavelengths, hyper-parameters, mean; l == 0; a negative
    hyper-parameter beside them): the shim must do what the REFERENCE does with the same call -- ValueError,
    ZeroDivisionError, -inf or a number -- as recorded from the reference itself in golden_conventions_v1.json
    (tests/golden/make_golden_conventions.py).  Behind the C ABI the device keeps its own convention."""
    import json
    import os
    from psoap_amd import _convention_cases as cc
    from psoap_amd.chunk import ChunkHandle
    with open(os.path.join(os.path.dirname(__file__), "golden", "golden_conventions_v1.json")) as fh:
        want = json.load(fh)
    cases = cc.cases()
    assert sorted(want) == sorted(name for name, *_ in cases) and len(cases) >= 30
    kinds = set()
    for name, fname, args, kwargs in cases:
        got = cc.outcome(getattr(cov, fname), args, kwargs, None)
        assert got["kind"] == want[name]["kind"], (name, got, want[name])
        if got["kind"] == "finite":
            assert abs(got["value"] - want[name]["value"]) <= LNP_RTOL * max(1.0, abs(want[name]["value"])), (name, got, want[name])
        kinds.add(got["kind"])
    assert kinds == {"ValueError", "ZeroDivisionError", "-inf", "finite"}
    # the same dict entry the reference's Worker calls (sample_parallel.py:193)
    name, fname, args, kwargs = next(c for c in cases if c[0] == "fg_fl_nan")
    with pytest.raises(ValueError, match="infs or NaNs"):
        cov.lnlike["SB2"](None, *args)
    cov.release_handles()
    # ---- the C ABI itself: no exceptions, NaN / -inf
    ch = syn.make_chunk(2, 3, 50, seed=77)
    fl = ch.fl.copy()
    fl[17] = np.nan
    sg = ch.sigma.copy()
    sg[5] = np.nan
    with ChunkHandle(fl, ch.sigma, max_batch=1) as h:
        assert np.isnan(h.lnlike(ch.lwls, syn.GP_BASE[2]))
    with ChunkHandle(ch.fl, sg, max_batch=1) as h:
        assert h.lnlike(ch.lwls, syn.GP_BASE[2]) == -np.inf


@pytest.mark.parametrize("mode,c", [(0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 1)])
def test_predict_variance_only_matches_the_diagonal_of_sigma(mode, c):
    """diag(Sigma) from the variance-only entry (prior variance minus the column norms of W) against the diagonal
    of the full Sigma = A - W^T W of the same call, every mode (components / sum incl. the 1e-8 nugget / predict_f);
    ragged sizes (N, M not multiples of 128)."""
    from psoap_amd.chunk import ChunkHandle
    ch = syn.make_chunk(c, 5, 117, seed=600 + 10 * mode + c)           # N = 585
    M = ch.N if (mode == 1 and c == 3) else 150                        # predict_f_g_h_sum needs M == N (covariance.py:294)
    pred = np.linspace(ch.lwls[0].min(), ch.lwls[0].max(), M)
    mu_c = np.full(c, 0.25)
    with ChunkHandle(ch.fl, ch.sigma, max_batch=1) as h:
        mu, Sigma = h.predict(mode, ch.lwls, np.stack([pred] * c), mu_c, syn.GP_BASE[c])
        mu2, var = h.predict(mode, ch.lwls, np.stack([pred] * c), mu_c, syn.GP_BASE[c], want_sigma="diag")
    assert np.array_equal(mu, mu2)
    assert np.max(np.abs(var - np.diag(Sigma))) <= 1e-12
