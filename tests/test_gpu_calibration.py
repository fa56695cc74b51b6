"""GPU: the device calibration solve (two augmented Cholesky passes) against vectors from the reference
and against the CPU oracle."""
import os
import sys

import numpy as np
import pytest

from psoap_amd import synthetic as syn

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
CAL_RTOL = 1e-8      # of max|fl_cor| / max|X|: two fp64 Cholesky passes with condition numbers up to ~1e3-1e5


@pytest.fixture(scope="module")
def ghost():
    return dict(np.load(os.path.join(ROOT, "tests", "golden", "golden_host_v1.npz")))


def test_device_calibration_matches_reference(ghost, oracle):
    from psoap_amd import covariance as cov
    from test_calibration_oracle import cal_cases, close
    for name, c, order, case in cal_cases():
        # kernel form: blocks evaluated on the device
        fl_cor, X = cov.optimize_calibration_components(case["lwl0"], case["lwl1"], case["lwl_cal"], case["lwls_cal"],
                                                        case["fl_cal"], case["sigma_cal"], case["lwls_fixed"],
                                                        case["fl_fixed"], case["sigma_fixed"], case["gp"], order=order)
        assert close(fl_cor, ghost[f"cal_{name}_fl"], CAL_RTOL) and close(X, ghost[f"cal_{name}_X"], CAL_RTOL), name
        # explicit form: the reference's signature, caller-filled A, B, C
        A, B, C = oracle.calibration_blocks(case["lwls_cal"], case["sigma_cal"], case["lwls_fixed"],
                                            case["sigma_fixed"], case["gp"])
        fl2, X2 = cov.optimize_calibration(case["lwl0"], case["lwl1"], case["lwl_cal"], case["fl_cal"],
                                           case["fl_fixed"], A, B, C, order=order)
        assert close(fl2, ghost[f"cal_{name}_fl"], CAL_RTOL) and close(X2, ghost[f"cal_{name}_X"], CAL_RTOL), name
        if c == 1:
            fs, Xs = cov.optimize_calibration_static(case["lwl0"], case["lwl1"], case["lwl_cal"], case["fl_cal"],
                                                     case["sigma_cal"], case["lwls_fixed"][0], case["fl_fixed"],
                                                     case["sigma_fixed"], case["gp"][0], case["gp"][1], order=order)
            assert close(fs, ghost[f"cal_{name}_static_fl"], CAL_RTOL)
            assert close(Xs, ghost[f"cal_{name}_static_X"], CAL_RTOL)


def test_calibration_mu_and_larger_problem(oracle):
    from psoap_amd import covariance as cov
    from make_golden_host import cal_case
    from test_calibration_oracle import close
    case = cal_case(syn, 3, 6, 400, 960, 0.05, 0.9, limit_array=4)        # M ~ 380, N ~ 1520: several tiles
    A, B, C = oracle.calibration_blocks(case["lwls_cal"], case["sigma_cal"], case["lwls_fixed"], case["sigma_fixed"],
                                        case["gp"])
    for order, mu in ((1, 1.0), (4, 0.97)):
        want_fl, want_X = oracle.optimize_calibration(case["lwl0"], case["lwl1"], case["lwl_cal"], case["fl_cal"],
                                                      case["fl_fixed"], A, B, C, order=order, mu_GP=mu)
        fl_cor, X = cov.optimize_calibration_components(case["lwl0"], case["lwl1"], case["lwl_cal"], case["lwls_cal"],
                                                        case["fl_cal"], case["sigma_cal"], case["lwls_fixed"],
                                                        case["fl_fixed"], case["sigma_fixed"], case["gp"], order=order,
                                                        mu_GP=mu)
        assert close(fl_cor, want_fl, CAL_RTOL) and close(X, want_X, CAL_RTOL), (order, mu)


def test_calibration_failure_raises():
    from psoap_amd import covariance as cov
    M, N = 40, 90
    rng = np.random.RandomState(0)
    lwl = np.sort(rng.uniform(8.5, 8.501, M))
    B = np.eye(N)
    B[3, 3] = -1.0                                        # not positive definite
    with pytest.raises(np.linalg.LinAlgError):
        cov.optimize_calibration(8.5, 8.501, lwl, np.ones(M), np.ones(N), np.eye(M), B, np.zeros((M, N)))
    with pytest.raises(np.linalg.LinAlgError):            # C' = A - C B^-1 C^T indefinite
        cov.optimize_calibration(8.5, 8.501, lwl, np.ones(M), np.ones(N), -np.eye(M), np.eye(N), np.zeros((M, N)))
    from psoap_amd._lib import PsoapError
    with pytest.raises(PsoapError):
        cov.optimize_calibration(8.5, 8.501, lwl, np.ones(M), np.ones(N), np.eye(M), np.eye(N), np.zeros((M, N)), order=40)


def test_cycle_calibration_pulls_epochs_together():
    from psoap_amd import covariance as cov
    from psoap_amd import data as pdata
    rng = np.random.RandomState(4)
    n_ep, n_pix = 5, 90
    lwl = np.tile(np.linspace(8.5560, 8.5568, n_pix), (n_ep, 1)) + 1e-6 * rng.standard_normal((n_ep, 1))
    truth = 1.0 + 0.08 * np.sin(2e4 * (lwl - 8.556))
    scale = np.array([1.0, 1.08, 0.93, 1.05, 0.97])[:, None]
    fl = truth * scale + 0.004 * rng.standard_normal((n_ep, n_pix))
    sigma = np.full_like(fl, 0.004)
    out = cov.cycle_calibration(lwl, fl, sigma, 0.1, 10.0, ncycles=2, order=1, limit_array=3)
    spread_before = np.std(fl / truth, axis=0).mean()
    spread_after = np.std(out / truth, axis=0).mean()
    assert out.shape == fl.shape and spread_after < 0.25 * spread_before
    # mask-aware chunk form: masked pixels are corrected with the fitted polynomial too
    mask = rng.uniform(size=fl.shape) > 0.1
    ch = pdata.Chunk(lwl.copy(), fl.copy(), sigma.copy(), np.zeros_like(fl), mask)
    ch.lwl = None
    cov.cycle_calibration_chunk(ch, 0.1, 10.0, 2, order=1, limit_array=3)
    assert np.std(ch.fl / truth, axis=0).mean() < 0.25 * spread_before


def test_optimize_GP_f_matches_oracle_driven_fit(oracle):
    """Same Nelder-Mead driver (covariance.py:405-422) on the device likelihood and on the oracle likelihood."""
    from scipy.optimize import minimize
    from psoap_amd import covariance as cov
    ch = syn.make_chunk(1, 3, 80, seed=321)               # N = 240
    got = cov.optimize_GP_f(ch.lwls[0], ch.fl, ch.sigma, 0.3, 8.0)
    want = minimize(lambda x: -oracle.lnlike(ch.lwls, ch.fl, ch.sigma, [x[0], x[1]]), np.array([0.3, 8.0]),
                    method="Nelder-Mead")["x"]
    assert np.allclose(got, want, rtol=1e-5)
    assert cov.lnlike_f(None, ch.lwls[0], ch.fl, ch.sigma, *got) >= cov.lnlike_f(None, ch.lwls[0], ch.fl, ch.sigma, 0.3, 8.0)


@pytest.mark.parametrize("c,ne,npx,order,limit,seed", [
    (1, 2, 9, 0, 1, 971),        # M = 9, N = 9: everything below one tile, constant correction
    (2, 3, 129, 5, 2, 972),      # M = 129 (one past a tile edge), order 5
    (3, 5, 200, 2, 4, 973),      # N = 800, three components
])
def test_calibration_edge_shapes_vs_oracle(oracle, c, ne, npx, order, limit, seed):
    from psoap_amd import covariance as cov
    from make_golden_host import cal_case
    from test_calibration_oracle import close
    case = cal_case(syn, c, ne, npx, seed, 0.0, 1.04, limit_array=limit)
    A, B, C = oracle.calibration_blocks(case["lwls_cal"], case["sigma_cal"], case["lwls_fixed"], case["sigma_fixed"],
                                        case["gp"])
    want_fl, want_X = oracle.optimize_calibration(case["lwl0"], case["lwl1"], case["lwl_cal"], case["fl_cal"],
                                                  case["fl_fixed"], A, B, C, order=order)
    fl_cor, X = cov.optimize_calibration_components(case["lwl0"], case["lwl1"], case["lwl_cal"], case["lwls_cal"],
                                                    case["fl_cal"], case["sigma_cal"], case["lwls_fixed"],
                                                    case["fl_fixed"], case["sigma_fixed"], case["gp"], order=order)
    assert X.shape == (order + 1,)
    # high Chebyshev orders make the normal equations ill-conditioned: compare the corrected flux tightly,
    # the coefficients relative to the largest one
    assert close(fl_cor, want_fl, 1e-8) and close(X, want_X, 1e-6 if order >= 5 else 1e-8), (X, want_X)
    fl2, X2 = cov.optimize_calibration(case["lwl0"], case["lwl1"], case["lwl_cal"], case["fl_cal"], case["fl_fixed"],
                                       A, B, C, order=order)
    assert close(fl2, want_fl, 1e-8)
