"""Drop-in for ``psoap.matrix_functions`` (the reference's only native module).

Same names, positional signatures and in-place/return-None contract as
/root/reference/psoap/matrix_functions.pyx:
``fill_V11_f`` :19-59, ``fill_V12_f`` :61-96, ``fill_V11_f_g`` :99-146,
``fill_V11_f_g_h`` :149-201.  The matrix is produced by the gfx950 fill kernel
and copied into the caller's array.
"""
from __future__ import annotations

import numpy as np

from . import _lib
from ._lib import as_f64, check, dptr


def _target(mat, shape):
    if not isinstance(mat, np.ndarray) or mat.dtype != np.float64 or mat.ndim != 2:
        raise ValueError("Buffer dtype mismatch, expected a 2-D float64 ndarray")
    if mat.shape != tuple(shape):
        raise ValueError(f"matrix has shape {mat.shape}, expected {tuple(shape)}")
    return mat if mat.flags.c_contiguous else np.empty(shape, dtype=np.float64)


def _fill_sym(mat, lwls, gp, sigma=None, device=None):
    lwls = as_f64(lwls)
    c, n_vec = lwls.shape
    N = len(mat)                      # pyx:23  N = len(mat)
    if n_vec < N:
        raise ValueError("wavelength vector shorter than the matrix")
    lwls = as_f64(lwls[:, :N])
    gp = as_f64(gp, (2 * c,))
    out = _target(mat, (N, N))
    sig = None if sigma is None else as_f64(sigma, (N,))
    dev = _lib.default_device() if device is None else device
    check(_lib.load().psoap_fill_sym(dev, c, N, dptr(lwls), dptr(gp), None if sig is None else dptr(sig),
                                     dptr(out)), "psoap_fill_sym")
    if out is not mat:
        mat[...] = out


def fill_V11_f(mat, lwl_f, amp_f, l_f):
    _fill_sym(mat, np.stack([as_f64(lwl_f)]), [amp_f, l_f])


def fill_V11_f_g(mat, lwl_f, lwl_g, amp_f, l_f, amp_g, l_g):
    _fill_sym(mat, np.stack([as_f64(lwl_f), as_f64(lwl_g)]), [amp_f, l_f, amp_g, l_g])


def fill_V11_f_g_h(mat, lwl_f, lwl_g, lwl_h, amp_f, l_f, amp_g, l_g, amp_h, l_h):
    _fill_sym(mat, np.stack([as_f64(lwl_f), as_f64(lwl_g), as_f64(lwl_h)]),
              [amp_f, l_f, amp_g, l_g, amp_h, l_h])


def fill_V12_f(mat, lwl_f, lwl_predict, amp_f, l_f):
    lwl_f = as_f64(lwl_f)
    lwl_predict = as_f64(lwl_predict)
    M, N = len(lwl_f), len(lwl_predict)   # pyx:65-66
    out = _target(mat, (M, N))
    check(_lib.load().psoap_fill_cross(_lib.default_device(), M, N, dptr(lwl_f), dptr(lwl_predict),
                                       float(amp_f), float(l_f), dptr(out)), "psoap_fill_cross")
    if out is not mat:
        mat[...] = out
