"""Pins the CPU oracle (oracle/) against golden vectors generated from the
reference (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

from psoap_amd import synthetic as syn

# tolerance contract: |dlnp| <= 1e-10 * max(1, |lnp|)
LNP_RTOL = 1e-10


def ulp_diff(a, b):
    a = np.ascontiguousarray(a, dtype=np.float64).view(np.int64)
    b = np.ascontiguousarray(b, dtype=np.float64).view(np.int64)
    return np.abs(a - b)


def test_fills_bit_exact(golden, oracle):
    c, ne, npx, seed, M = golden["fill_meta"]
    ch = syn.make_chunk(c, ne, npx, seed=seed)
    N = ch.N
    gp = syn.GP_BASE[3]
    m = np.empty((N, N))
    oracle.fill_V11_f(m, ch.lwls[0], *gp[:2])
    assert np.array_equal(m, golden["fill_f"])
    oracle.fill_V11_f_g(m, ch.lwls[0], ch.lwls[1], *gp[:4])
    assert np.array_equal(m, golden["fill_f_g"])
    oracle.fill_V11_f_g_h(m, *ch.lwls, *gp)
    assert np.array_equal(m, golden["fill_f_g_h"])
    pred = np.linspace(ch.lwls[0].min(), ch.lwls[0].max(), M)
    r = np.empty((M, N))
    oracle.fill_V12_f(r, pred, ch.lwls[0], *gp[:2])
    assert np.array_equal(r, golden["fill_cross_40xN"])
    r = np.empty((N, M))
    oracle.fill_V12_f(r, ch.lwls[1], pred, *gp[2:4])
    assert np.array_equal(r, golden["fill_cross_Nx40"])


def test_replicate_wls_matches_reference(golden):
    c, ne, npx, seed = golden["replicate_meta"]
    ch = syn.make_chunk(c, ne, npx, seed=seed, masked_fraction=0.2)
    assert np.array_equal(ch.lwls, golden["replicate_masked"])


def _cases(golden):
    for name, meta, val in zip(golden["lnlike_names"], golden["lnlike_meta"], golden["lnlike_vals"]):
        c, ne, npx, seed, mf100, N = meta
        yield str(name), int(c), int(ne), int(npx), int(seed), mf100 / 100.0, int(N), float(val)


def test_lnlike_scipy_layer_all_sizes(golden, oracle):
    for name, c, ne, npx, seed, mf, N, val in _cases(golden):
        ch = syn.make_chunk(c, ne, npx, seed=seed, masked_fraction=mf)
        assert ch.N == N
        got = oracle.lnlike(ch.lwls, ch.fl, ch.sigma, syn.GP_BASE[c])
        assert abs(got - val) <= LNP_RTOL * max(1.0, abs(val)), (name, got, val)


def test_lnlike_c_layer_small_sizes(golden, oracle):
    for name, c, ne, npx, seed, mf, N, val in _cases(golden):
        if N > 2000:
            continue
        ch = syn.make_chunk(c, ne, npx, seed=seed, masked_fraction=mf)
        got = oracle.lnlike_c(ch.lwls, ch.fl, ch.sigma, syn.GP_BASE[c])
        assert abs(got - val) <= LNP_RTOL * max(1.0, abs(val)), (name, got, val)


def test_lnlike_conventions(golden, oracle):
    ch = syn.make_chunk(2, 8, 32, seed=105)
    for fn in (oracle.lnlike, oracle.lnlike_c):
        v = fn(ch.lwls, ch.fl, ch.sigma, syn.GP_BASE[2], mu_GP=0.9)
        assert abs(v - golden["lnlike_mu0p9"]) <= LNP_RTOL * abs(v)
        assert np.all(np.isneginf(golden["lnlike_infs"]))
        assert fn(ch.lwls[:1], ch.fl, ch.sigma, (-0.2, 5.0)) == -np.inf
        assert fn(ch.lwls, ch.fl, ch.sigma, (0.2, 5.0, 0.1, -7.0)) == -np.inf
        lw3 = np.vstack([ch.lwls, ch.lwls[:1]])
        assert fn(lw3, ch.fl, ch.sigma, (0.2, 5.0, 0.1, 7.0, -0.05, 6.0)) == -np.inf
        lw = ch.lwls.copy()
        lw[:, 1] = lw[:, 0]
        assert fn(lw, ch.fl, np.zeros_like(ch.sigma), syn.GP_BASE[2]) == -np.inf
        v = fn(ch.lwls[:1], ch.fl, ch.sigma, (0.0, 5.0))
        assert abs(v - golden["lnlike_zero_amp"]) <= LNP_RTOL * abs(v)


def test_walker_batch(golden, oracle):
    ch = syn.make_config_chunk(3)
    gps = syn.make_walkers(2, 4, seed=3500)
    lw = syn.walker_lwls(ch, syn.make_walker_velocities(ch, 4, seed=3501))
    for w in (0, 3):
        got = oracle.lnlike(lw[w], ch.fl, ch.sigma, gps[w])
        val = golden["walkers_cfg3"][w]
        assert abs(got - val) <= LNP_RTOL * abs(val)


def test_predict_small(golden, oracle):
    c, ne, npx, seed, M, c2, ne2, npx2, seed2 = golden["pred_meta"]
    ch = syn.make_chunk(c, ne, npx, seed=seed)
    pg = np.linspace(ch.lwls[0].min(), ch.lwls[0].max(), M)
    mu, Sig = oracle.predict_components(ch.lwls[:2], ch.fl, ch.sigma, [pg, pg], [0.0, 0.0], syn.GP_BASE[2])
    np.testing.assert_allclose(mu, golden["pred_fg_mu"], rtol=0, atol=1e-10)
    np.testing.assert_allclose(Sig, golden["pred_fg_Sigma"], rtol=0, atol=1e-9)
    mu = oracle.predict_components(ch.lwls[:2], ch.fl, ch.sigma, [pg, pg + 1e-5], [0.3, 0.7], syn.GP_BASE[2],
                                   get_Sigma=False)
    np.testing.assert_allclose(mu, golden["pred_fg_mu_only"], rtol=0, atol=1e-10)
    mu, Sig = oracle.predict_components(ch.lwls, ch.fl, ch.sigma, [pg, pg, pg], [0.0] * 3, syn.GP_BASE[3])
    np.testing.assert_allclose(mu, golden["pred_fgh_mu"], rtol=0, atol=1e-10)
    np.testing.assert_allclose(Sig, golden["pred_fgh_Sigma"], rtol=0, atol=1e-9)
    mu, Sig = oracle.predict_sum(ch.lwls[:2], ch.fl, ch.sigma, [pg, pg], 1.0, syn.GP_BASE[2])
    np.testing.assert_allclose(mu, golden["pred_fg_sum_mu"], rtol=0, atol=1e-10)
    np.testing.assert_allclose(Sig, golden["pred_fg_sum_Sigma"], rtol=0, atol=1e-9)
    chq = syn.make_chunk(c2, ne2, npx2, seed=seed2)
    mu, Sig = oracle.predict_sum(chq.lwls, chq.fl, chq.sigma, chq.lwls, 1.0, syn.GP_BASE[3])
    np.testing.assert_allclose(mu, golden["pred_fgh_sum_mu"], rtol=0, atol=1e-10)
    np.testing.assert_allclose(Sig, golden["pred_fgh_sum_Sigma"], rtol=0, atol=1e-9)


def test_predict_retrieve_shape(golden, oracle):
    c, ne, npx, seed, M = golden["predL_meta"]
    ch = syn.make_chunk(c, ne, npx, seed=seed)
    pg = np.linspace(ch.lwls[0].min(), ch.lwls[0].max(), M)
    mu, Sig = oracle.predict_components(ch.lwls, ch.fl, ch.sigma, [pg] * 3, [0.0] * 3, syn.GP_BASE[3])
    np.testing.assert_allclose(mu, golden["predL_fgh_mu"], rtol=0, atol=1e-10)
    np.testing.assert_allclose(np.diag(Sig), golden["predL_fgh_diag"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(Sig[[0, 399, 400, 777, 1199]], golden["predL_fgh_rows"], rtol=0, atol=1e-9)
    mu, Sig = oracle.predict_components(ch.lwls[:2], ch.fl, ch.sigma, [pg] * 2, [0.0] * 2, syn.GP_BASE[2])
    np.testing.assert_allclose(mu, golden["predL_fg_mu"], rtol=0, atol=1e-10)
    np.testing.assert_allclose(np.diag(Sig), golden["predL_fg_diag"], rtol=0, atol=1e-9)
