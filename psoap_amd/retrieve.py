"""Component-spectrum reconstruction of one chunk: the body of the reference's retrieve scripts
(/root/reference/scripts/psoap_retrieve_SB2.py:40-160, psoap_retrieve_ST3.py:40-170) without the plotting.

``retrieve_components`` evaluates the orbit at the chunk's dates, shifts the ln-wavelengths to every
component's rest frame, masks, predicts all components on a grid twice as fine as the data spanning the
primary's rest-frame range (``n_pix_predict = 2 n_pix``, SB2:96-105 / ST3:97-109) with prior means 0, and
splits ``mu`` / ``sqrt(diag(Sigma))`` per component.  Orbit solve and the conditional run on the device.
``save_components`` writes what the scripts write: ``f.npy``, ``g.npy`` (, ``h.npy``) as
``vstack((wl_predict, mu_k, sigma_k))``, ``mu.npy`` and ``Sigma.npy`` (SB2:150-155).
"""
from __future__ import annotations

import os

import numpy as np

from . import covariance, orbit
from .data import c_kms, epoch_index_of
from .utils import N_COMPONENTS, n_params_orb, registered_params

_NAMES = "fgh"


def retrieve_components(model, chunk, parameters, n_pix_predict=None, get_Sigma=True):
    """``chunk``: an UNMASKED ``psoap_amd.data.Chunk`` (2-D arrays); ``parameters``: the config's
    ``parameters`` dictionary (every registered parameter of ``model``).  Returns a dict with
    ``wl_predict``, ``lwl_predict``, ``mu`` (c * M,), ``Sigma`` (c * M, c * M) or None, and per
    component ``mu_f`` / ``sigma_f`` (, ``_g``, ``_h``)."""
    if model not in ("SB2", "ST3", "SB1"):
        raise ValueError("retrieve_components supports SB1, SB2 and ST3 (the models the reference has scripts for)")
    c = N_COMPONENTS[model]
    reg = registered_params[model]
    p_orb = [parameters[n] for n in reg[:n_params_orb[model]]]
    p_gp = [parameters[n] for n in reg[n_params_orb[model]:]]
    mask = np.asarray(chunk.mask, dtype=bool)
    lwl2d = np.log(np.asarray(chunk.wl, dtype=np.float64)) if np.ndim(chunk.wl) == 2 else None
    if lwl2d is None:
        raise ValueError("retrieve_components needs the 2-D chunk: call it before apply_mask()")
    vel = np.atleast_2d(orbit.models[model](*p_orb, np.asarray(chunk.date1D)).get_velocities())   # (c, n_epochs)
    ep = epoch_index_of(mask)
    lwl = lwl2d[mask]
    lwls = lwl[None, :] + (-vel[:, ep]) / c_kms          # lredshift(lwls, -v) per epoch (data.py:37)
    fl = np.asarray(chunk.fl, dtype=np.float64)[mask]
    sigma = np.asarray(chunk.sigma, dtype=np.float64)[mask]
    M = int(n_pix_predict) if n_pix_predict is not None else 2 * int(chunk.n_pix)
    lwl_predict = np.linspace(np.min(lwls[0]), np.max(lwls[0]), num=M)
    pred = [lwl_predict] * c
    if isinstance(get_Sigma, str):
        # get_Sigma="diag": the per-component means and standard deviations only -- what the scripts plot and save as
        # f.npy / g.npy / h.npy -- without the (c M)^2 covariance matrix (psoap_predictor_run_var)
        if get_Sigma != "diag":
            raise ValueError('get_Sigma must be True, False or "diag"')
        mu, var = covariance.predict_components_var(lwls, fl, sigma, pred, [0.0] * c, p_gp)
        out = {"wl_predict": np.exp(lwl_predict), "lwl_predict": lwl_predict, "mu": mu, "Sigma": None, "lwls": lwls}
        sd = np.sqrt(var)
        for k in range(c):
            out["mu_" + _NAMES[k]] = mu[k * M:(k + 1) * M]
            out["sigma_" + _NAMES[k]] = sd[k * M:(k + 1) * M]
        return out
    if c == 1:
        res = covariance.predict_f(lwls[0], fl, sigma, lwl_predict, *p_gp, mu_GP=0.0)
        mu, Sigma = res
    elif c == 2:
        res = covariance.predict_f_g(lwls[0], lwls[1], fl, sigma, pred[0], pred[1], 0.0, p_gp[0], p_gp[1], 0.0,
                                     p_gp[2], p_gp[3], get_Sigma=get_Sigma)
        mu, Sigma = res if get_Sigma else (res, None)
    else:
        mu, Sigma = covariance.predict_f_g_h(lwls[0], lwls[1], lwls[2], fl, sigma, *pred, 0.0, 0.0, 0.0, *p_gp)
    out = {"wl_predict": np.exp(lwl_predict), "lwl_predict": lwl_predict, "mu": mu, "Sigma": Sigma, "lwls": lwls}
    sd = np.sqrt(np.diag(Sigma)) if Sigma is not None else None
    for k in range(c):
        out["mu_" + _NAMES[k]] = mu[k * M:(k + 1) * M]
        if sd is not None:
            out["sigma_" + _NAMES[k]] = sd[k * M:(k + 1) * M]
    return out


def save_components(result, outdir):
    """Write ``f.npy``, ``g.npy`` (, ``h.npy``), ``mu.npy``, ``Sigma.npy`` into ``outdir`` (the scripts'
    ``plots_chunk_*`` directory)."""
    os.makedirs(outdir, exist_ok=True)
    for k in _NAMES:
        if "mu_" + k in result and "sigma_" + k in result:
            np.save(os.path.join(outdir, k + ".npy"), np.vstack((result["wl_predict"], result["mu_" + k], result["sigma_" + k])))
    np.save(os.path.join(outdir, "mu.npy"), result["mu"])
    if result["Sigma"] is not None:
        np.save(os.path.join(outdir, "Sigma.npy"), result["Sigma"])
