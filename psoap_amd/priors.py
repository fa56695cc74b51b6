"""Default flat priors of the sampling drivers, vectorised over a batch of proposals.

/root/reference/psoap/sample_parallel.py:330-358 defines ``prior_SB1``, ``prior_SB2`` and ``prior_ST3``:
0.0 inside the box, ``-inf`` when a semi-amplitude, mass ratio, period, GP amplitude or length scale is
negative, an eccentricity leaves [0, 1] or an argument of periastron leaves [-90, 450] degrees (strict
inequalities, so the bounds themselves are allowed).  ST1 and ST2 have no prior in the reference (its
``priors`` dict, :369, would raise KeyError); they get the same rule applied to the parameters they have.
A user ``prior.py`` in the working directory overrides the default exactly as :362-366 does.
"""
from __future__ import annotations

import numpy as np

from .utils import registered_params

_NONNEG = ("q", "K", "P", "amp_f", "l_f", "amp_g", "l_g", "amp_h", "l_h")


def _rule(name):
    stem = name[:-3] if name.endswith("_in") else name[:-4] if name.endswith("_out") else name
    if stem == "e":
        return "ecc"
    if stem == "omega":
        return "omega"
    if stem in _NONNEG:
        return "nonneg"
    return None                                   # T0, gamma: unconstrained


def box_prior_full(model, full):
    """(B,) prior of full parameter vectors (B, n_registered) in ``registered_params[model]`` order."""
    full = np.atleast_2d(np.asarray(full, dtype=np.float64))
    bad = np.zeros(full.shape[0], dtype=bool)
    for i, name in enumerate(registered_params[model]):
        rule = _rule(name)
        x = full[:, i]
        if rule == "nonneg":
            bad |= x < 0.0
        elif rule == "ecc":
            bad |= (x < 0.0) | (x > 1.0)
        elif rule == "omega":
            bad |= (x < -90) | (x > 450)
    return np.where(bad, -np.inf, 0.0)


def make_prior(model, fix_params=(), **defaults):
    """``prior(P)`` over fitted vectors (B, n_fit) -> (B,); fixed parameters take their config values
    (the reference's priors see them through ``convert_vector_p``, sample_parallel.py:331)."""
    reg = registered_params[model]
    fit_ind = [i for i, n in enumerate(reg) if n not in fix_params]

    def prior(P):
        P = np.atleast_2d(np.asarray(P, dtype=np.float64))
        full = np.empty((P.shape[0], len(reg)))
        full[:, fit_ind] = P
        for name in fix_params:
            full[:, reg.index(name)] = defaults[name]
        return box_prior_full(model, full)

    return prior


def load_user_prior(directory="."):
    """A ``prior(p)`` from ``prior.py`` in ``directory`` if present (sample_parallel.py:362-366), else None.
    The user function is scalar; it is wrapped to take a batch."""
    import importlib.util
    import os
    path = os.path.join(directory, "prior.py")
    if not os.path.exists(path):
        return None
    spec = importlib.util.spec_from_file_location("psoap_user_prior", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    scalar = mod.prior
    return lambda P: np.array([scalar(p) for p in np.atleast_2d(P)], dtype=np.float64)
