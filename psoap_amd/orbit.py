"""Radial-velocity orbit models evaluated on the device.

Same constructor signatures and ``get_velocities`` contract as /root/reference/psoap/orbit.py
(``SB1`` :20-115, ``SB2`` :117-170, ``ST1`` :172-320, ``ST2`` :323-417, ``ST3`` :420-487, ``models``
:490): positional orbital parameters followed by ``obs_dates``; ``get_velocities(dates=None)`` returns
a ``(n_components, n_dates)`` array in km/s.  The Kepler solve is a batched Newton iteration in the
HIP library (psoap_orbit_velocities) instead of one ``scipy.optimize.fsolve`` call per date.
"""
from __future__ import annotations

import numpy as np

from . import _lib
from ._lib import as_f64, check, dptr
from .utils import MODEL_ID, N_COMPONENTS, n_params_orb


def velocities(model: str, p_orb, dates, device=None) -> np.ndarray:
    """Batched: p_orb (B, n_orb) -> (B, c, n_dates) km/s."""
    p_orb = as_f64(np.atleast_2d(p_orb))
    dates = as_f64(np.atleast_1d(dates))
    B, n_orb = p_orb.shape
    if n_orb != n_params_orb[model]:
        raise ValueError(f"model {model} takes {n_params_orb[model]} orbital parameters, got {n_orb}")
    out = np.empty((B, N_COMPONENTS[model], dates.shape[0]))
    dev = _lib.default_device() if device is None else device
    check(_lib.load().psoap_orbit_velocities(dev, MODEL_ID[model], B, dptr(p_orb), dates.shape[0], dptr(dates),
                                             dptr(out)), "psoap_orbit_velocities")
    return out


class _Orbit:
    model = ""

    def __init__(self, *params, obs_dates=None, **kwargs):
        n = n_params_orb[self.model]
        if len(params) == n + 1 and obs_dates is None:      # the drivers pass the dates positionally
            params, obs_dates = params[:n], params[n]
        if len(params) != n:
            raise TypeError(f"{self.model} takes {n} orbital parameters")
        self.params = np.array([float(x) for x in params])
        self.obs_dates = obs_dates

    def get_velocities(self, dates=None):
        if dates is None and self.obs_dates is None:
            raise RuntimeError("Must provide input dates or specify observation dates upon creation of orbit object.")
        if dates is None:
            dates = self.obs_dates
        try:
            return velocities(self.model, self.params[None, :], np.atleast_1d(dates))[0]
        except _lib.PsoapError as e:
            if "Eccentricity" in str(e):
                raise AssertionError("Eccentricity must be between [0, 1)") from e   # orbit.py:35
            raise


class SB1(_Orbit):
    model = "SB1"


class SB2(_Orbit):
    model = "SB2"


class ST1(_Orbit):
    model = "ST1"


class ST2(_Orbit):
    model = "ST2"


class ST3(_Orbit):
    model = "ST3"


models = {"SB1": SB1, "SB2": SB2, "ST1": ST1, "ST2": ST2, "ST3": ST3}
