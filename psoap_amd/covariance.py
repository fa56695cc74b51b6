"""Drop-in for the GP likelihood / prediction functions of ``psoap.covariance``.

Signatures, argument meaning, return types and error behaviour follow
/root/reference/psoap/covariance.py (``lnlike_f`` :299-331, ``lnlike_f_g`` :333-354,
``lnlike_f_g_h`` :356-376, ``lnlike`` :379, ``predict_f`` :25-54, ``predict_f_g``
:81-148, ``predict_f_g_sum`` :151-187, ``predict_f_g_h`` :190-251,
``predict_f_g_h_sum`` :253-297) so that ``psoap.sample_parallel``'s
``covariance.lnlike[model](self.V11, *lwls, self.fl, self.sigma, *p_GP)`` (:193)
works unchanged.  All arithmetic runs in the HIP library behind include/psoap_gp.h.

The ``V11`` argument is the caller's scratch matrix; the reference leaves
unspecified contents in it and no caller reads it back, so it is ignored here --
the matrix lives in HBM.  HIP is initialised lazily on the first call, i.e. in
the worker process after the fork.
"""
from __future__ import annotations

import ctypes
from collections import OrderedDict

import numpy as np

from . import _lib
from ._lib import as_f64, check, dptr
from .chunk import ChunkHandle

_MAX_CACHED_CHUNKS = 4
_handles: "OrderedDict[tuple, ChunkHandle]" = OrderedDict()


def _chunk_for(fl, sigma) -> ChunkHandle:
    """One persistent device handle per (fl, sigma) pair, keyed on the buffers and
    verified by content, so a Worker re-uses its resident chunk on every proposal."""
    fl = as_f64(fl)
    sigma = as_f64(sigma)
    key = (fl.ctypes.data, sigma.ctypes.data, fl.shape[0])
    h = _handles.get(key)
    if h is not None and np.array_equal(h.fl, fl) and np.array_equal(h.sigma, sigma):
        _handles.move_to_end(key)
        return h
    if h is not None:
        h.close()
        del _handles[key]
    h = ChunkHandle(fl.copy(), sigma.copy(), max_batch=1)
    _handles[key] = h
    while len(_handles) > _MAX_CACHED_CHUNKS:
        _, old = _handles.popitem(last=False)
        old.close()
    return h


def release_handles():
    """Free every cached device chunk."""
    while _handles:
        _, h = _handles.popitem()
        h.close()


def _lnlike(lwls, fl, sigma, gp, mu_GP):
    gp = [float(g) for g in gp]
    if any(g < 0.0 for g in gp):          # covariance.py:317-318,339-340,362-363
        return -np.inf
    h = _chunk_for(fl, sigma)
    return np.float64(h.lnlike(np.stack([as_f64(w) for w in lwls]), gp, mu_GP))


def lnlike_f(V11, wl_f, fl, sigma, amp_f, l_f, mu_GP=1.):
    return _lnlike([wl_f], fl, sigma, [amp_f, l_f], mu_GP)


def lnlike_f_g(V11, wl_f, wl_g, fl, sigma, amp_f, l_f, amp_g, l_g, mu_GP=1.):
    return _lnlike([wl_f, wl_g], fl, sigma, [amp_f, l_f, amp_g, l_g], mu_GP)


def lnlike_f_g_h(V11, wl_f, wl_g, wl_h, fl, sigma, amp_f, l_f, amp_g, l_g, amp_h, l_h, mu_GP=1.):
    return _lnlike([wl_f, wl_g, wl_h], fl, sigma, [amp_f, l_f, amp_g, l_g, amp_h, l_h], mu_GP)


# covariance.py:379
lnlike = {"SB1": lnlike_f, "SB2": lnlike_f_g, "ST1": lnlike_f, "ST2": lnlike_f_g, "ST3": lnlike_f_g_h}


def _predict(mode, lwls, fl, sigma, lwls_predict, mu_c, gp, want_sigma=True):
    lwls = np.stack([as_f64(w) for w in lwls])
    pred = np.stack([as_f64(w) for w in lwls_predict])
    c, N = lwls.shape
    M = pred.shape[1]
    fl = as_f64(fl, (N,))
    sigma = as_f64(sigma, (N,))
    mu_c = as_f64(mu_c)
    gp = as_f64(gp, (2 * c,))
    R = c * M if mode == 0 else M
    mu = np.empty(R)
    Sigma = np.empty((R, R)) if want_sigma else None
    status = ctypes.c_int(0)
    check(_lib.load().psoap_predict(_lib.default_device(), mode, c, N, M, dptr(lwls), dptr(fl), dptr(sigma),
                                    dptr(pred), dptr(mu_c), dptr(gp), dptr(mu),
                                    None if Sigma is None else dptr(Sigma), ctypes.byref(status)),
          "psoap_predict")
    if status.value != 0:
        # cho_factor raises here in the reference (covariance.py:113,182,222,292)
        raise np.linalg.LinAlgError("data covariance matrix is not positive definite")
    return (mu, Sigma) if want_sigma else mu


def predict_f(lwl_known, fl_known, sigma_known, lwl_predict, amp_f, l_f, mu_GP=1.0):
    """Single-component conditional.  The reference body is unrunnable
    (NameError ``wl_predict``, covariance.py:38); this implements the evidently
    intended ``N = len(lwl_predict)``."""
    return _predict(2, [lwl_known], fl_known, sigma_known, [lwl_predict], [mu_GP], [amp_f, l_f])


def predict_f_g(lwl_f, lwl_g, fl_fg, sigma_fg, lwl_f_predict, lwl_g_predict, mu_f, amp_f, l_f, mu_g, amp_g, l_g,
                get_Sigma=True):
    assert len(lwl_f) == len(lwl_g), "Input wavelengths must be the same length."
    assert len(lwl_f_predict) == len(lwl_g_predict), "Prediction wavelengths must be the same length."
    return _predict(0, [lwl_f, lwl_g], fl_fg, sigma_fg, [lwl_f_predict, lwl_g_predict], [mu_f, mu_g],
                    [amp_f, l_f, amp_g, l_g], want_sigma=bool(get_Sigma))


def predict_f_g_sum(lwl_f, lwl_g, fl_fg, sigma_fg, lwl_f_predict, lwl_g_predict, mu_fg, amp_f, l_f, amp_g, l_g):
    assert len(lwl_f) == len(lwl_g), "Input wavelengths must be the same length."
    return _predict(1, [lwl_f, lwl_g], fl_fg, sigma_fg, [lwl_f_predict, lwl_g_predict], [mu_fg],
                    [amp_f, l_f, amp_g, l_g])


def predict_f_g_h(lwl_f, lwl_g, lwl_h, fl_fgh, sigma_fgh, lwl_f_predict, lwl_g_predict, lwl_h_predict,
                  mu_f, mu_g, mu_h, amp_f, l_f, amp_g, l_g, amp_h, l_h):
    assert len(lwl_f) == len(lwl_g), "Input wavelengths must be the same length."
    assert len(lwl_f) == len(lwl_h), "Input wavelengths must be the same length."
    assert len(lwl_f_predict) == len(lwl_g_predict), "Prediction wavelengths must be the same length."
    assert len(lwl_f_predict) == len(lwl_h_predict), "Prediction wavelengths must be the same length."
    return _predict(0, [lwl_f, lwl_g, lwl_h], fl_fgh, sigma_fgh, [lwl_f_predict, lwl_g_predict, lwl_h_predict],
                    [mu_f, mu_g, mu_h], [amp_f, l_f, amp_g, l_g, amp_h, l_h])


def predict_f_g_h_sum(lwl_f, lwl_g, lwl_h, fl_fgh, sigma_fgh, lwl_f_predict, lwl_g_predict, lwl_h_predict,
                      mu_fgh, amp_f, l_f, amp_g, l_g, amp_h, l_h):
    assert len(lwl_f) == len(lwl_g), "Input wavelengths must be the same length."
    if len(lwl_f_predict) != len(lwl_f):
        # the reference fails here too: np.dot(V12.T, ...) at covariance.py:294 only conforms for M == N
        raise ValueError("shapes not aligned: predict_f_g_h_sum needs len(lwl_predict) == len(lwl) "
                         "(covariance.py:294)")
    return _predict(1, [lwl_f, lwl_g, lwl_h], fl_fgh, sigma_fgh, [lwl_f_predict, lwl_g_predict, lwl_h_predict],
                    [mu_fgh], [amp_f, l_f, amp_g, l_g, amp_h, l_h])
