"""CPU: the orbit oracle and the parameter plumbing against vectors generated from the reference
(tests/golden/make_golden_orbit.py)."""
import os
import sys

import numpy as np
import pytest

from psoap_amd import synthetic as syn
from psoap_amd import utils

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VEL_ATOL = 1e-8   # km/s; the reference's own Kepler solve is fsolve with xtol 1.5e-8


@pytest.fixture(scope="module")
def gorb():
    return dict(np.load(os.path.join(ROOT, "tests", "golden", "golden_orbit_v1.npz")))


@pytest.fixture(scope="module")
def orbit_oracle():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orbit_oracle
    return orbit_oracle


def test_oracle_velocities_match_reference(gorb, orbit_oracle):
    dates = gorb["dates"]
    assert np.array_equal(dates, syn.make_dates(20, seed=77))
    for model in ("SB1", "SB2", "ST1", "ST2", "ST3"):
        P = syn.make_orbit_proposals(model, 6, seed=500)
        for i, p in enumerate(P):
            v = orbit_oracle.velocities(model, p, dates)
            np.testing.assert_allclose(v, gorb[f"vel_{model}"][i], rtol=0, atol=VEL_ATOL)
    v = orbit_oracle.velocities("SB2", gorb["p_SB2_ecc"], dates)
    np.testing.assert_allclose(v, gorb["vel_SB2_ecc"], rtol=0, atol=VEL_ATOL)


def test_convert_vector_matches_reference(gorb):
    o, g = utils.convert_vector(np.arange(1.0, 11.0), "SB2", ["gamma"], gamma=3.5)
    assert np.array_equal(o, gorb["cv_SB2_orb"]) and np.array_equal(g, gorb["cv_SB2_gp"])
    o, g = utils.convert_vector(np.arange(1.0, 17.0), "ST3", ["gamma", "e_out", "omega_out"],
                                gamma=1.5, e_out=0.1, omega_out=20.0)
    assert np.array_equal(o, gorb["cv_ST3_orb"]) and np.array_equal(g, gorb["cv_ST3_gp"])
    assert utils.n_params_orb == {"SB1": 6, "SB2": 7, "ST1": 11, "ST2": 12, "ST3": 13}
    with pytest.raises(ValueError):
        utils.convert_vectors(np.zeros((2, 3)), "SB2", ["gamma"], gamma=0.0)


def test_worker_lnprob_restated_with_oracles(gorb, orbit_oracle, oracle):
    """The CPU restatement of Worker.lnprob (orbit oracle -> replicate_wls -> lnlike oracle) reproduces
    the reference's numbers; this is what the GPU lnprob(p) path is compared with."""
    ch = syn.make_chunk(2, 8, 75, seed=610, masked_fraction=0.1)
    P = syn.make_orbit_proposals("SB2", 5, seed=611)
    G = syn.make_walkers(2, 5, seed=612)
    for i in range(5):
        vel = orbit_oracle.velocities("SB2", P[i], ch.dates)
        lw = syn.replicate_wls(ch.lwl, vel, ch.mask)
        got = oracle.lnlike(lw, ch.fl, ch.sigma, G[i])
        assert abs(got - gorb["lnprob_SB2"][i]) <= 1e-8 * abs(got)
