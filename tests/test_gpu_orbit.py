"""GPU: device orbit evaluation and the lnprob(p) boundary against vectors from the reference."""
import os

import numpy as np
import pytest

from psoap_amd import synthetic as syn

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VEL_ATOL = 1e-8      # km/s
LNP_RTOL = 1e-8      # lnprob(p): the reference's fsolve tolerance feeds through the Doppler shift


@pytest.fixture(scope="module")
def gorb():
    return dict(np.load(os.path.join(ROOT, "tests", "golden", "golden_orbit_v1.npz")))


def test_device_velocities_match_reference(gorb):
    from psoap_amd import orbit
    dates = gorb["dates"]
    for model in ("SB1", "SB2", "ST1", "ST2", "ST3"):
        P = syn.make_orbit_proposals(model, 6, seed=500)
        v = orbit.velocities(model, P, dates)
        np.testing.assert_allclose(v, gorb[f"vel_{model}"], rtol=0, atol=VEL_ATOL)
        # class interface, dates passed positionally as sample_parallel.py:183 does
        one = orbit.models[model](*P[2], dates).get_velocities()
        assert np.array_equal(one, v[2])
    v = orbit.SB2(*gorb["p_SB2_ecc"], obs_dates=dates).get_velocities()
    np.testing.assert_allclose(v, gorb["vel_SB2_ecc"], rtol=0, atol=VEL_ATOL)
    with pytest.raises(AssertionError, match="Eccentricity"):
        orbit.SB1(10.0, 1.2, 0.0, 5.0, 0.0, 0.0, dates).get_velocities()
    with pytest.raises(RuntimeError):
        orbit.SB1(10.0, 0.2, 0.0, 5.0, 0.0, 0.0).get_velocities()


@pytest.mark.parametrize("model,c,shape,seeds,key", [
    ("SB2", 2, (8, 75, 0.1), (610, 611, 612), "lnprob_SB2"),
    ("ST3", 3, (6, 60, 0.0), (620, 621, 622), "lnprob_ST3"),
    ("SB1", 1, (5, 40, 0.0), (630, 631, 632), "lnprob_SB1"),
])
def test_lnprob_of_p_matches_reference(gorb, model, c, shape, seeds, key):
    from psoap_amd.lnprob import ChunkWorker
    from psoap_amd.utils import registered_params
    ne, npx, mf = shape
    ch = syn.make_chunk(c, ne, npx, seed=seeds[0], masked_fraction=mf)
    want = gorb[key]
    P = syn.make_orbit_proposals(model, len(want), seed=seeds[1])
    G = syn.make_walkers(c, len(want), seed=seeds[2])
    fit = np.hstack([P, G])                       # every registered parameter is fitted
    w = ChunkWorker(model, ch.lwl, ch.fl, ch.sigma, ch.epoch_index, ch.dates, max_batch=len(want))
    assert fit.shape[1] == len(registered_params[model])
    try:
        got = w.lnprob_batch(fit)
        for i in range(len(want)):
            assert abs(got[i] - want[i]) <= LNP_RTOL * max(1.0, abs(want[i])), (model, i, got[i], want[i])
        one = w.lnprob(fit[1])               # a single evaluation is scheduled differently from the batch: a few ulp
        assert abs(one - got[1]) <= 1e-13 * abs(got[1]) and w.lnprob(fit[1]) == one
        if model == "SB2":
            fast = fit[0].copy()
            fast[1] = 4.0e5                       # K: faster than light -> -inf (sample_parallel.py:186-187)
            assert w.lnprob(fast) == -np.inf and np.isneginf(gorb["lnprob_SB2_fast"])
            neg = fit[0].copy()
            neg[-1] = -1.0                        # l_g < 0 -> -inf (covariance.py:339-340)
            assert w.lnprob(neg) == -np.inf
    finally:
        w.close()
