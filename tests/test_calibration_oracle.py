"""CPU: the calibration oracle against vectors from the reference's optimize_calibration /
optimize_calibration_static (tests/golden/make_golden_host.py)."""
import os
import sys

import numpy as np
import pytest

from psoap_amd import synthetic as syn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
CAL_RTOL = 1e-9      # relative to max|fl_cor| / max|X|; both sides are LAPACK, conditioning ~1e5


@pytest.fixture(scope="module")
def ghost():
    return dict(np.load(os.path.join(ROOT, "tests", "golden", "golden_host_v1.npz")))


def cal_cases():
    from make_golden_host import CAL_CASES, cal_case
    for name, (c, ne, npx, seed, mf, order, scale) in CAL_CASES.items():
        yield name, c, order, cal_case(syn, c, ne, npx, seed, mf, scale)


def close(got, want, rtol):
    return np.max(np.abs(np.asarray(got) - want)) <= rtol * np.max(np.abs(want))


def test_oracle_calibration_matches_reference(ghost, oracle):
    seen = 0
    for name, c, order, case in cal_cases():
        A, B, C = oracle.calibration_blocks(case["lwls_cal"], case["sigma_cal"], case["lwls_fixed"],
                                            case["sigma_fixed"], case["gp"])
        fl_cor, X = oracle.optimize_calibration(case["lwl0"], case["lwl1"], case["lwl_cal"], case["fl_cal"],
                                                case["fl_fixed"], A, B, C, order=order)
        assert X.shape == (order + 1,) and fl_cor.shape == case["fl_cal"].shape
        assert close(fl_cor, ghost[f"cal_{name}_fl"], CAL_RTOL) and close(X, ghost[f"cal_{name}_X"], CAL_RTOL), name
        if c == 1:
            # the static form is the explicit form with the fills done inside (covariance.py:628-707); its
            # kernel abscissa for the epoch is the Chebyshev abscissa itself (the golden call passes lwl_cal)
            A, B, C = oracle.calibration_blocks([case["lwl_cal"]], case["sigma_cal"], case["lwls_fixed"],
                                                case["sigma_fixed"], case["gp"])
            fs, Xs = oracle.optimize_calibration(case["lwl0"], case["lwl1"], case["lwl_cal"], case["fl_cal"],
                                                 case["fl_fixed"], A, B, C, order=order)
            assert close(fs, ghost[f"cal_{name}_static_fl"], CAL_RTOL)
            assert close(Xs, ghost[f"cal_{name}_static_X"], CAL_RTOL)
        # the correction undoes most of the injected mis-scaling: X0 ~ 1 / scale
        assert 0.8 < X[0] < 1.2
        seen += 1
    assert seen == 4
