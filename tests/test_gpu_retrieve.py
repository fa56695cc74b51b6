"""GPU: the reconstruction helper (retrieve scripts' body) against the same pipeline on the CPU oracles."""
import os
import sys

import numpy as np
import pytest

from psoap_amd import synthetic as syn

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def _chunk2d(s):
    from psoap_amd import data as pdata
    def full(v, fill):
        out = np.full(s.mask.shape, fill)
        out[s.mask] = v
        return out
    date = np.broadcast_to(s.dates[:, None], s.mask.shape).copy()
    return pdata.Chunk(np.exp(full(s.lwl, 8.5)), full(s.fl, 1.0), full(s.sigma, 1.0), date, s.mask.copy())


@pytest.mark.parametrize("model,c,seed", [("SB2", 2, 881), ("ST3", 3, 882)])
def test_retrieve_matches_oracle_pipeline(oracle, tmp_path, model, c, seed):
    import orbit_oracle
    from psoap_amd import retrieve
    from psoap_amd.utils import registered_params
    s = syn.make_chunk(c, 6, 48, seed=seed, masked_fraction=0.1)
    ch = _chunk2d(s)
    p_orb = syn.make_orbit_proposals(model, 1, seed=seed + 1)[0]
    gp = syn.GP_BASE[c]
    pars = dict(zip(registered_params[model], list(p_orb) + list(gp)))
    res = retrieve.retrieve_components(model, ch, pars)
    M = 2 * ch.n_pix
    assert res["mu"].shape == (c * M,) and res["Sigma"].shape == (c * M, c * M)
    # the same steps on the CPU oracles
    vel = orbit_oracle.velocities(model, p_orb, s.dates)
    lwls = s.lwl[None, :] - vel[:, s.epoch_index] / 2.99792458e5
    np.testing.assert_allclose(res["lwls"], lwls, rtol=0, atol=1e-13)
    pred = np.linspace(lwls[0].min(), lwls[0].max(), M)
    mu_o, Sig_o = oracle.predict_components(lwls, s.fl, s.sigma, [pred] * c, [0.0] * c, gp)
    np.testing.assert_allclose(res["mu"], mu_o, rtol=0, atol=1e-9)
    np.testing.assert_allclose(res["Sigma"], Sig_o, rtol=0, atol=1e-9)
    np.testing.assert_allclose(res["mu_g"], mu_o[M:2 * M], rtol=0, atol=1e-9)
    # variance-only form: the same per-component curves, no covariance matrix
    res_d = retrieve.retrieve_components(model, ch, pars, get_Sigma="diag")
    assert res_d["Sigma"] is None and np.array_equal(res_d["mu"], res["mu"])
    for k in "fgh"[:c]:
        np.testing.assert_allclose(res_d["sigma_" + k], res["sigma_" + k], rtol=0, atol=1e-9)
    retrieve.save_components(res, str(tmp_path / "plots"))
    f = np.load(tmp_path / "plots" / "f.npy")
    assert f.shape == (3, M) and np.array_equal(f[0], res["wl_predict"]) and np.array_equal(f[1], res["mu_f"])
    assert (tmp_path / "plots" / "Sigma.npy").exists() and (tmp_path / "plots" / ("h.npy" if c == 3 else "g.npy")).exists()
    ch.apply_mask()
    with pytest.raises(ValueError):
        retrieve.retrieve_components(model, ch, pars)
