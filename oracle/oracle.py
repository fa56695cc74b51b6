"""CPU oracle for the GP-likelihood path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  The product package (``psoap_amd``) never does.

Two layers:

* ``psoap_oracle.c`` (plain C, built by ``oracle/Makefile``): the kernel-matrix
  fills, an unblocked upper Cholesky, the two triangular solves and the scalar
  log-likelihood -- a LAPACK-free restatement.
* this module: a NumPy/SciPy restatement that uses the C fills and then the
  *same third-party calls the reference makes* (``scipy.linalg.cho_factor`` /
  ``cho_solve``, ``numpy.dot``; SciPy and NumPy are un-pinned dependencies of
  the reference -- /root/reference/requirements.txt:1-7 -- here SciPy 1.15.3 /
  NumPy 2.2.6 / OpenBLAS 0.3.29).  This is the layer that is fast enough to
  check N = 6000..8192 and the one timed as ``cpu_baseline`` (kind "port").

Parity pinning: the reference's own tests hold nothing for this path, so both
layers are pinned by golden vectors produced by importing the reference in the
build container (``tests/golden/make_golden.py``); ``tests/test_oracle_golden.py``
checks them.

References are to files under /root/reference.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np
from scipy.linalg import cho_factor, cho_solve

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libpsoap_oracle.so")
_lib = None

_dp = ctypes.POINTER(ctypes.c_double)


def build(force: bool = False) -> str:
    """Compile psoap_oracle.c with gcc (idempotent)."""
    src = os.path.join(_HERE, "psoap_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or \
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libpsoap_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        L.oracle_fill_sym.argtypes = [_dp, ctypes.c_int, ctypes.c_int, _dp, _dp]
        L.oracle_fill_sym.restype = None
        L.oracle_fill_cross.argtypes = [_dp, ctypes.c_int, ctypes.c_int, _dp, _dp,
                                        ctypes.c_double, ctypes.c_double]
        L.oracle_fill_cross.restype = None
        L.oracle_cholesky_upper.argtypes = [_dp, ctypes.c_int]
        L.oracle_cholesky_upper.restype = ctypes.c_int
        L.oracle_lnlike.argtypes = [_dp, ctypes.c_int, ctypes.c_int, _dp, _dp, _dp, _dp,
                                    ctypes.c_double]
        L.oracle_lnlike.restype = ctypes.c_double
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(_dp)


def _vec(a):
    return np.ascontiguousarray(a, dtype=np.float64)


# --------------------------------------------------------------------------- fills
def fill_sym(mat, lwls, gp):
    """In-place symmetric fill; psoap/matrix_functions.pyx:19-59,99-146,149-201."""
    lwls = _vec(np.atleast_2d(lwls))
    gp = _vec(gp)
    c, N = lwls.shape
    assert mat.shape == (N, N) and mat.flags.c_contiguous and mat.dtype == np.float64
    lib().oracle_fill_sym(_p(mat), N, c, _p(lwls), _p(gp))


def fill_V11_f(mat, lwl_f, amp_f, l_f):
    fill_sym(mat, [lwl_f], [amp_f, l_f])


def fill_V11_f_g(mat, lwl_f, lwl_g, amp_f, l_f, amp_g, l_g):
    fill_sym(mat, [lwl_f, lwl_g], [amp_f, l_f, amp_g, l_g])


def fill_V11_f_g_h(mat, lwl_f, lwl_g, lwl_h, amp_f, l_f, amp_g, l_g, amp_h, l_h):
    fill_sym(mat, [lwl_f, lwl_g, lwl_h], [amp_f, l_f, amp_g, l_g, amp_h, l_h])


def fill_V12_f(mat, lwl_f, lwl_predict, amp_f, l_f):
    """In-place rectangular fill; psoap/matrix_functions.pyx:61-96."""
    lwl_f, lwl_predict = _vec(lwl_f), _vec(lwl_predict)
    M, N = lwl_f.shape[0], lwl_predict.shape[0]
    assert mat.shape == (M, N) and mat.flags.c_contiguous and mat.dtype == np.float64
    lib().oracle_fill_cross(_p(mat), M, N, _p(lwl_f), _p(lwl_predict), amp_f, l_f)


# --------------------------------------------------------------------------- lnlike
def lnlike_c(lwls, fl, sigma, gp, mu_GP=1.0):
    """Pure-C lnlike (psoap/covariance.py:299-376); LAPACK-free, O(N^3) scalar."""
    lwls = _vec(np.atleast_2d(lwls))
    c, N = lwls.shape
    scratch = np.empty((N, N))
    return lib().oracle_lnlike(_p(scratch), N, c, _p(lwls), _p(_vec(fl)), _p(_vec(sigma)),
                               _p(_vec(gp)), mu_GP)


def lnlike(lwls, fl, sigma, gp, mu_GP=1.0, V11=None):
    """NumPy/SciPy lnlike: C fill + the reference's own LAPACK calls.

    psoap/covariance.py:317-331 (c=1), :339-354 (c=2), :362-376 (c=3).
    """
    lwls = _vec(np.atleast_2d(lwls))
    gp = _vec(gp)
    if np.any(gp < 0.0):
        return -np.inf
    c, N = lwls.shape
    if V11 is None:
        V11 = np.empty((N, N))
    fill_sym(V11, lwls, gp)
    V11[np.diag_indices_from(V11)] += _vec(sigma) ** 2
    try:
        factor, flag = cho_factor(V11, overwrite_a=True, lower=False, check_finite=False)
    except np.linalg.LinAlgError:
        return -np.inf
    logdet = np.sum(2 * np.log(np.diag(factor)))
    r = _vec(fl) - mu_GP
    return -0.5 * (np.dot(r, cho_solve((factor, flag), r)) + logdet)


# --------------------------------------------------------------------------- predict
def _data_cov(lwls, sigma, gp):
    c, N = lwls.shape
    B = np.zeros((N, N))
    tmp = np.empty((N, N))
    for k in range(c):
        fill_V11_f(tmp, lwls[k], gp[2 * k], gp[2 * k + 1])
        B = tmp.copy() if k == 0 else B + tmp
    B[np.diag_indices_from(B)] += sigma ** 2
    return B


def predict_components(lwls, fl, sigma, lwls_predict, mus, gp, get_Sigma=True):
    """Joint conditional of the c components (predict_f_g: psoap/covariance.py:81-148,
    predict_f_g_h: :190-251).

    Component matrices are filled *separately* and summed (:109,:219), the mean
    offset is the hard-coded ``fl - 1.0`` (:140,:248), the prior ``A`` is
    block-diagonal (:125,:236) and ``C`` stacks the per-component cross fills
    (:136,:246).
    """
    lwls = _vec(np.atleast_2d(lwls))
    lwls_predict = _vec(np.atleast_2d(lwls_predict))
    gp = _vec(gp)
    fl, sigma = _vec(fl), _vec(sigma)
    c, N = lwls.shape
    M = lwls_predict.shape[1]
    factor = cho_factor(_data_cov(lwls, sigma, gp))
    A = np.zeros((c * M, c * M))
    C = np.empty((c * M, N))
    for k in range(c):
        blk = np.empty((M, M))
        fill_V11_f(blk, lwls_predict[k], gp[2 * k], gp[2 * k + 1])
        A[k * M:(k + 1) * M, k * M:(k + 1) * M] = blk
        cross = np.empty((M, N))
        fill_V12_f(cross, lwls_predict[k], lwls[k], gp[2 * k], gp[2 * k + 1])
        C[k * M:(k + 1) * M] = cross
    mu_cat = np.concatenate([np.full(M, float(m)) for m in mus])
    mu = mu_cat + np.dot(C, cho_solve(factor, fl - 1.0))
    if not get_Sigma:
        return mu
    Sigma = A - np.dot(C, cho_solve(factor, C.T))
    return mu, Sigma


def predict_sum(lwls, fl, sigma, lwls_predict, mu_sum, gp):
    """Conditional of the *sum* of the components.

    c=2: predict_f_g_sum (psoap/covariance.py:151-187): 1e-8 nugget on the prior
    (:165), mean offset ``fl - 1.0`` (:184).
    c=3: predict_f_g_h_sum (:253-297): no nugget (:271), mean offset
    ``fl - mu_fgh`` and ``V12.T`` in the mean (:294) -- valid only for M == N,
    which is the only way the reference calls it.
    """
    lwls = _vec(np.atleast_2d(lwls))
    lwls_predict = _vec(np.atleast_2d(lwls_predict))
    gp = _vec(gp)
    fl, sigma = _vec(fl), _vec(sigma)
    c, N = lwls.shape
    M = lwls_predict.shape[1]
    V11 = np.zeros((M, M))
    V12 = np.zeros((M, N))
    for k in range(c):
        blk = np.empty((M, M))
        fill_V11_f(blk, lwls_predict[k], gp[2 * k], gp[2 * k + 1])
        V11 = blk if k == 0 else V11 + blk
        cross = np.empty((M, N))
        fill_V12_f(cross, lwls_predict[k], lwls[k], gp[2 * k], gp[2 * k + 1])
        V12 = cross if k == 0 else V12 + cross
    if c == 2:
        V11[np.diag_indices_from(V11)] += 1e-8
    factor = cho_factor(_data_cov(lwls, sigma, gp))
    if c == 2:
        mu = mu_sum + np.dot(V12, cho_solve(factor, fl - 1.0))
    else:
        assert M == N, "predict_f_g_h_sum is only defined for M == N in the reference"
        mu = mu_sum + np.dot(V12.T, cho_solve(factor, fl - mu_sum))
    Sigma = V11 - np.dot(V12, cho_solve(factor, V12.T))
    return mu, Sigma


# ---- calibration (test infrastructure, like everything in this file) --------------------------------
def _cheb_design(lwl0, lwl1, lwl_cal, fl_cal, order):
    """D = fl_cal[:, None] * T_k(lwl_cal), k = 0..order, Chebyshev domain [lwl0, lwl1] (covariance.py:584-593)."""
    from numpy.polynomial import Chebyshev
    T = np.array([Chebyshev([0] * k + [1], domain=[lwl0, lwl1])(lwl_cal) for k in range(order + 1)])
    return np.asarray(fl_cal)[:, None] * T.T


def optimize_calibration(lwl0, lwl1, lwl_cal, fl_cal, fl_fixed, A, B, C, order=1, mu_GP=1.0):
    """covariance.py:560-624 restated with SciPy's cho_factor / cho_solve."""
    D = _cheb_design(lwl0, lwl1, lwl_cal, fl_cal, order)
    Bc = cho_factor(B)
    fl_prime = mu_GP + C @ cho_solve(Bc, np.asarray(fl_fixed).flatten() - mu_GP)
    C_prime = A - C @ cho_solve(Bc, C.T)
    Cc = cho_factor(C_prime)
    left = D.T @ cho_solve(Cc, D)
    right = D.T @ cho_solve(Cc, fl_prime)
    X = cho_solve(cho_factor(left), right)
    return D @ X, X


def calibration_blocks(lwls_cal, sigma_cal, lwls_fixed, sigma_fixed, gp):
    """A, B, C as scripts/psoap_process_calibration_ST3.py:147-176 builds them (per-component fills summed)."""
    lwls_cal, lwls_fixed = np.atleast_2d(lwls_cal), np.atleast_2d(lwls_fixed)
    c, M = lwls_cal.shape
    N = lwls_fixed.shape[1]
    A, B, C = np.zeros((M, M)), np.zeros((N, N)), np.zeros((M, N))
    for k in range(c):
        a, l = gp[2 * k], gp[2 * k + 1]
        t = np.empty((M, M)); fill_V11_f(t, lwls_cal[k], a, l); A = A + t
        t = np.empty((N, N)); fill_V11_f(t, lwls_fixed[k], a, l); B = B + t
        t = np.empty((M, N)); fill_V12_f(t, lwls_cal[k], lwls_fixed[k], a, l); C = C + t
    A[np.diag_indices_from(A)] += np.asarray(sigma_cal) ** 2
    B[np.diag_indices_from(B)] += np.asarray(sigma_fixed) ** 2
    return A, B, C
