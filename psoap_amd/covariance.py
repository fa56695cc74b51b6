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


def _chunk_for(fl, sigma):
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
    from . import server as _server
    if _server.wanted():
        # PSOAP_GPU_SERVER: the chunk lives in the process that owns the GPU (psoap_amd/server.py), this worker never
        # initialises HIP -- the reference's worker-per-chunk model with dozens of workers per device
        h = _server.connect_chunk(fl.copy(), sigma.copy())
    else:
        h = ChunkHandle(fl.copy(), sigma.copy(), max_batch=1)
    _handles[key] = h
    while len(_handles) > _MAX_CACHED_CHUNKS:
        _, old = _handles.popitem(last=False)
        old.close()
    return h


def release_handles():
    """Free every cached device chunk and the predict workspaces."""
    while _handles:
        _, h = _handles.popitem()
        h.close()
    while _predictors:
        _, p = _predictors.popitem()
        _lib.load().psoap_predictor_destroy(p)


# one reusable predict workspace per device: the retrieve loop predicts chunk after chunk
# (scripts/psoap_retrieve_ST3.py:148) and allocates only when a chunk outgrows the previous ones
_predictors: dict = {}


def _predictor():
    dev = _lib.default_device()
    p = _predictors.get(dev)
    if p is None:
        p = ctypes.c_void_p()
        check(_lib.load().psoap_predictor_create(ctypes.byref(p), dev), "psoap_predictor_create")
        _predictors[dev] = p
    return p


def last_predict_timings() -> dict:
    """Timings (ms) and algorithmic flops of the last ``predict_*`` call on the default device."""
    t = _lib.PredictTimings()
    check(_lib.load().psoap_predictor_timings(_predictor(), ctypes.byref(t)), "psoap_predictor_timings")
    return t.as_dict()


_NONFINITE = "array must not contain infs or NaNs"      # the message of scipy's check_finite


def _matrix_is_finite(lw: np.ndarray, sigma, gp) -> bool:
    """Would the reference's filled matrix (fill_V11_* + sigma**2 on the diagonal) be free of NaN / inf?  Entries are
    ``amp**2 * exp(-0.5 c**2 r**2 / l**2)`` with ``r`` a wavelength difference: a NaN wavelength, amplitude or length scale,
    an infinite amplitude, two infinite wavelengths of one sign in one component (inf - inf), or a non-finite
    uncertainty poison it; ONE infinite wavelength (exp(-inf) = 0) or an infinite length scale (exp(-0) = 1) do not."""
    if not np.all(np.isfinite(np.asarray(sigma, dtype=np.float64))):
        return False
    amps, ls = gp[0::2], gp[1::2]
    if not all(np.isfinite(a) for a in amps) or any(np.isnan(l) for l in ls):
        return False
    if np.any(np.isnan(lw)):
        return False
    if lw.shape[1] > 1 and (np.any(np.sum(np.isposinf(lw), axis=1) > 1) or np.any(np.sum(np.isneginf(lw), axis=1) > 1)):
        return False
    return True


def _lnlike(lwls, fl, sigma, gp, mu_GP):
    """Error behaviour of the reference on degenerate input, pinned by running the reference itself
    (tests/golden/make_golden_conventions.py -> golden_conventions_v1.json):

    * a negative hyper-parameter -> ``-inf`` before anything else (covariance.py:317,339,362);
    * a length scale of exactly 0 -> ``ZeroDivisionError`` (``-0.5 * c_kms2 / (l*l)`` in the Cython fills,
      matrix_functions.pyx:29,111-112,165-167: Cython checks float division);
    * anything non-finite in the covariance matrix -> ``ValueError``: ``lnlike_f`` / ``lnlike_f_g_h`` factor with
      scipy's ``check_finite`` (:325,370); ``lnlike_f_g`` factors unchecked (:348), but OpenBLAS's ``dpotrf`` does not
      stop at a NaN pivot, the factor comes back non-finite and the ``cho_solve`` that follows refuses it (:354);
    * a finite matrix that is not positive definite -> ``-inf`` (``LinAlgError`` caught, :327,350,372);
    * non-finite ``fl`` / ``mu_GP`` (the right-hand side of ``cho_solve``, :331,354,376) -> ``ValueError``.

    Behind the C ABI the device keeps its own convention (NaN in, NaN or -inf out, never an exception): include/psoap_gp.h."""
    gp = [float(g) for g in gp]
    if any(g < 0.0 for g in gp):
        return -np.inf
    if any(l == 0.0 for l in gp[1::2]):
        raise ZeroDivisionError("float division")
    lw = np.stack([as_f64(w) for w in lwls])
    if not _matrix_is_finite(lw, sigma, gp):
        raise ValueError(_NONFINITE)
    h = _chunk_for(fl, sigma)
    out = np.float64(h.lnlike(lw, gp, mu_GP))
    if np.isneginf(out):
        return out
    if not (np.all(np.isfinite(np.asarray(fl, dtype=np.float64))) and np.isfinite(mu_GP)):
        raise ValueError(_NONFINITE)
    return out


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
    status = ctypes.c_int(0)
    if isinstance(want_sigma, str):
        if want_sigma != "diag":
            raise ValueError('want_sigma must be True, False or "diag"')
        var = np.empty(R)
        check(_lib.load().psoap_predictor_run_var(_predictor(), mode, c, N, M, dptr(lwls), dptr(fl), dptr(sigma),
                                                  dptr(pred), dptr(mu_c), dptr(gp), dptr(mu), dptr(var),
                                                  ctypes.byref(status)), "psoap_predictor_run_var")
        if status.value != 0:
            raise np.linalg.LinAlgError("data covariance matrix is not positive definite")
        return mu, var
    Sigma = np.empty((R, R)) if want_sigma else None
    check(_lib.load().psoap_predictor_run(_predictor(), mode, c, N, M, dptr(lwls), dptr(fl), dptr(sigma),
                                          dptr(pred), dptr(mu_c), dptr(gp), dptr(mu),
                                          None if Sigma is None else dptr(Sigma), ctypes.byref(status)),
          "psoap_predictor_run")
    if status.value != 0:
        # cho_factor raises here in the reference (covariance.py:113,182,222,292)
        raise np.linalg.LinAlgError("data covariance matrix is not positive definite")
    return (mu, Sigma) if want_sigma else mu


def predict_components_var(lwls, fl, sigma, lwls_predict, mus, gp):
    """Not in the reference: ``(mu, diag(Sigma))`` of ``predict_f`` / ``predict_f_g`` / ``predict_f_g_h`` (by the number
    of components) without forming ``Sigma`` -- all the retrieve scripts use of it is ``sqrt(diag(Sigma))``
    (scripts/psoap_retrieve_ST3.py:111-119).  ``lwls`` (c, N), ``lwls_predict`` (c, M), ``mus`` (c,), ``gp`` (2c,)."""
    return _predict(2 if len(lwls) == 1 else 0, lwls, fl, sigma, lwls_predict, mus, gp, want_sigma="diag")


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


def optimize_GP_f(wl_known, fl_known, sigma_known, amp_f, l_f, mu_GP=1.0):
    """Nelder-Mead fit of the single-component hyper-parameters to one slice of data, starting from
    ``(amp_f, l_f)`` (covariance.py:405-422); every likelihood evaluation runs on the device."""
    from scipy.optimize import minimize
    V11 = None          # the reference passes an N x N scratch matrix; the device path ignores it

    def func(x):
        a, l = x
        return -lnlike_f(V11, wl_known, fl_known, sigma_known, a, l, mu_GP)

    return minimize(func, np.array([amp_f, l_f]), method="Nelder-Mead")["x"]


# ---- calibration ---------------------------------------------------------------
_CAL_FAIL = {1: "reference-epoch covariance B", 2: "conditional covariance C'", 3: "Chebyshev normal equations"}


def _cal_result(status, fl_cor, X):
    if status != 0:
        # cho_factor raises in the reference (covariance.py:597-601, :607, :613)
        print("Failed to solve matrix inverse. Calibration not valid.")
        raise np.linalg.LinAlgError(f"{_CAL_FAIL[status]} is not positive definite")
    return fl_cor, X


def optimize_calibration(lwl0, lwl1, lwl_cal, fl_cal, fl_fixed, A, B, C, order=1, mu_GP=1.0):
    """Chebyshev calibration of one epoch with caller-filled covariance blocks (covariance.py:560-624).

    ``A`` (M,M) and ``B`` (N,N) carry the squared uncertainties on their diagonals, ``C`` (M,N) is the
    cross block.  Returns ``(fl_cor (M,), X (order+1,))``; the two Cholesky passes run on the device.
    """
    lwl_cal = as_f64(lwl_cal)
    M = lwl_cal.shape[0]
    fl_cal = as_f64(fl_cal, (M,))
    fl_fixed = as_f64(np.asarray(fl_fixed).flatten())
    N = fl_fixed.shape[0]
    A, B, C = as_f64(A, (M, M)), as_f64(B, (N, N)), as_f64(C, (M, N))
    fl_cor, X = np.empty(M), np.empty(order + 1)
    status = ctypes.c_int(0)
    check(_lib.load().psoap_calibrate_explicit(_lib.default_device(), M, N, int(order), float(lwl0), float(lwl1),
                                               dptr(lwl_cal), dptr(fl_cal), dptr(fl_fixed), dptr(A), dptr(B), dptr(C),
                                               float(mu_GP), dptr(fl_cor), dptr(X), ctypes.byref(status)),
          "psoap_calibrate_explicit")
    return _cal_result(status.value, fl_cor, X)


def optimize_calibration_components(lwl0, lwl1, lwl_cal, lwls_cal, fl_cal, sigma_cal, lwls_fixed, fl_fixed, sigma_fixed,
                                    gp, order=1, mu_GP=1.0):
    """The per-epoch body of scripts/psoap_process_calibration_ST3.py:147-183 with the covariance blocks
    evaluated on the device: ``lwls_cal`` (c,M) / ``lwls_fixed`` (c,N) are the rest-frame grids of the epoch
    and of the reference epochs, ``gp = (amp_f, l_f[, amp_g, l_g[, amp_h, l_h]])``; ``lwl_cal`` (M,) is the
    observed-frame abscissa of the Chebyshev polynomials on [lwl0, lwl1]."""
    lwls_cal = np.ascontiguousarray(np.atleast_2d(np.asarray(lwls_cal, dtype=np.float64)))
    lwls_fixed = np.ascontiguousarray(np.atleast_2d(np.asarray(lwls_fixed, dtype=np.float64)))
    c, M = lwls_cal.shape
    N = lwls_fixed.shape[1]
    assert lwls_fixed.shape[0] == c, "epoch and reference grids need the same number of components"
    lwl_cal, fl_cal, sigma_cal = as_f64(lwl_cal, (M,)), as_f64(fl_cal, (M,)), as_f64(sigma_cal, (M,))
    fl_fixed, sigma_fixed = as_f64(np.asarray(fl_fixed).flatten(), (N,)), as_f64(np.asarray(sigma_fixed).flatten(), (N,))
    gp = as_f64(gp, (2 * c,))
    fl_cor, X = np.empty(M), np.empty(order + 1)
    status = ctypes.c_int(0)
    check(_lib.load().psoap_calibrate(_lib.default_device(), c, M, N, int(order), float(lwl0), float(lwl1), dptr(lwl_cal),
                                      dptr(lwls_cal), dptr(fl_cal), dptr(sigma_cal), dptr(lwls_fixed), dptr(fl_fixed),
                                      dptr(sigma_fixed), dptr(gp), float(mu_GP), dptr(fl_cor), dptr(X),
                                      ctypes.byref(status)),
          "psoap_calibrate")
    return _cal_result(status.value, fl_cor, X)


def optimize_calibration_static(wl0, wl1, wl_cal, fl_cal, sigma_cal, wl_fixed, fl_fixed, sigma_fixed, amp, l_f, order=1,
                                mu_GP=1.0):
    """Single-component calibration with zero relative velocities (covariance.py:628-707): the same
    vectors serve as kernel abscissa and as Chebyshev abscissa."""
    return optimize_calibration_components(wl0, wl1, wl_cal, [wl_cal], fl_cal, sigma_cal,
                                           [np.asarray(wl_fixed).flatten()], fl_fixed, sigma_fixed, [amp, l_f],
                                           order=order, mu_GP=mu_GP)


def cycle_calibration(wl, fl, sigma, amp_f, l_f, ncycles, order=1, limit_array=3, mu_GP=1.0, soften=1.0):
    """Cycle the calibration over all epochs of an (n_epochs, n_pix) block (covariance.py:710-745).

    The reference body calls ``optimize_calibration`` with ``optimize_calibration_static``'s argument
    list (:740) and cannot run; this implements that evident intent."""
    wl, sigma = np.asarray(wl, dtype=np.float64), soften * np.asarray(sigma, dtype=np.float64)
    wl0, wl1 = np.min(wl), np.max(wl)
    fl_out = np.array(fl, dtype=np.float64)
    for _ in range(ncycles):
        for i in range(len(wl)):
            wl_remain = np.delete(wl, i, axis=0)[0:limit_array]
            fl_remain = np.delete(fl_out, i, axis=0)[0:limit_array]
            sigma_remain = np.delete(sigma, i, axis=0)[0:limit_array]
            fl_out[i], _X = optimize_calibration_static(wl0, wl1, wl[i], fl_out[i], sigma[i], wl_remain.flatten(),
                                                        fl_remain.flatten(), sigma_remain.flatten(), amp_f, l_f,
                                                        order=order, mu_GP=mu_GP)
    return fl_out


def cycle_calibration_chunk(chunk, amp_f, l_f, n_cycles, order=1, limit_array=3, mu_GP=1.0, soften=1.0):
    """Mask-aware cycle over a 2-D ``Chunk``; overwrites ``chunk.fl`` (covariance.py:747-800, same stale
    call at :776).  Masked pixels are left out of the solve and corrected with the fitted polynomial.
    As in the reference, ``soften`` is computed but the un-softened ``chunk.sigma`` enters the solve."""
    from numpy.polynomial import chebyshev as npcheb
    wl0, wl1 = np.min(chunk.wl), np.max(chunk.wl)
    fl_out = np.array(chunk.fl, dtype=np.float64)
    for _ in range(n_cycles):
        for i in range(chunk.n_epochs):
            mt = chunk.mask[i]
            wl_remain = np.delete(chunk.wl, i, axis=0)[0:limit_array]
            fl_remain = np.delete(fl_out, i, axis=0)[0:limit_array]
            sigma_remain = np.delete(chunk.sigma, i, axis=0)[0:limit_array]
            mr = np.delete(chunk.mask, i, axis=0)[0:limit_array]
            _fl, X = optimize_calibration_static(wl0, wl1, chunk.wl[i][mt], fl_out[i][mt], chunk.sigma[i][mt],
                                                 wl_remain[mr], fl_remain[mr], sigma_remain[mr], amp_f, l_f,
                                                 order=order, mu_GP=mu_GP)
            u = (2.0 * chunk.wl[i] - (wl0 + wl1)) / (wl1 - wl0)
            fl_out[i] = fl_out[i] * npcheb.chebval(u, X)          # D X re-evaluated on every pixel (:781-797)
    chunk.fl[:] = fl_out
