"""Parameter plumbing of the sampling drivers, restated for the lnprob(p) boundary.

``registered_params`` / ``n_params_orb`` / ``convert_vector`` follow
/root/reference/psoap/utils.py:4-14,27-69: a fitted vector ``p`` holds only the non-fixed
parameters; it is unrolled to the model's full parameter vector and split at ``gamma`` into the
orbital and the GP parts.
"""
from __future__ import annotations

import numpy as np

# utils.py:4-8
registered_params = {
    "SB1": ["K", "e", "omega", "P", "T0", "gamma", "amp_f", "l_f"],
    "SB2": ["q", "K", "e", "omega", "P", "T0", "gamma", "amp_f", "l_f", "amp_g", "l_g"],
    "ST1": ["K_in", "e_in", "omega_in", "P_in", "T0_in", "K_out", "e_out", "omega_out", "P_out", "T0_out", "gamma",
            "amp_f", "l_f"],
    "ST2": ["q_in", "K_in", "e_in", "omega_in", "P_in", "T0_in", "K_out", "e_out", "omega_out", "P_out", "T0_out",
            "gamma", "amp_f", "l_f"],
    "ST3": ["q_in", "K_in", "e_in", "omega_in", "P_in", "T0_in", "q_out", "K_out", "e_out", "omega_out", "P_out",
            "T0_out", "gamma", "amp_f", "l_f", "amp_g", "l_g", "amp_h", "l_h"],
}
# utils.py:14
n_params_orb = {m: registered_params[m].index("gamma") + 1 for m in registered_params}
MODEL_ID = {"SB1": 0, "SB2": 1, "ST1": 2, "ST2": 3, "ST3": 4}
N_COMPONENTS = {"SB1": 1, "SB2": 2, "ST1": 1, "ST2": 2, "ST3": 3}


def convert_vector(p, model, fix_params, **kwargs):
    """(par_orb, par_GP) for one fitted vector (utils.py:27-69)."""
    orb, gp = convert_vectors(np.atleast_2d(np.asarray(p, dtype=np.float64)), model, fix_params, **kwargs)
    return orb[0], gp[0]


def convert_vectors(ps, model, fix_params, **kwargs):
    """Batched ``convert_vector``: ps (B, n_fit) -> (B, n_orb), (B, n_gp)."""
    reg = registered_params[model]
    fit_ind = [i for i, name in enumerate(reg) if name not in fix_params]
    fix_ind = [reg.index(name) for name in fix_params]
    ps = np.asarray(ps, dtype=np.float64)
    if ps.ndim != 2 or ps.shape[1] != len(fit_ind):
        raise ValueError(f"expected (B, {len(fit_ind)}) fitted parameters for model {model}")
    full = np.empty((ps.shape[0], len(reg)))
    full[:, fit_ind] = ps
    for i, name in zip(fix_ind, fix_params):
        full[:, i] = kwargs[name]
    k = n_params_orb[model]
    return np.ascontiguousarray(full[:, :k]), np.ascontiguousarray(full[:, k:])


def convert_dict(model, fix_params, **kwargs):
    """Fitted-parameter vector from a ``{name: value}`` dictionary such as config.yaml's ``parameters`` or
    ``jumps`` (utils.py:71-84): the registered parameters that are not fixed, in registered order."""
    return np.array([kwargs[name] for name in registered_params[model] if name not in fix_params], dtype=np.float64)


def estimate_covariance(flatchain, ndim=0):
    """Proposal covariance for the next run from a flatchain, ``2.38**2 / d * cov(flatchain)`` (the
    ``opt_jump.npy`` of the drivers; utils.py:168-201, written by plot_samples.py:130-131).  The
    reference also saves a correlation-coefficient plot; that stays with its plotting script."""
    flatchain = np.asarray(flatchain, dtype=np.float64)
    d = flatchain.shape[1] if ndim == 0 else ndim
    return 2.38 ** 2 / d * np.cov(flatchain, rowvar=0)
