"""Persistent per-chunk device state and batched likelihood evaluation.

Host-side counterpart of ``Worker.initialize`` / ``Worker.lnprob``
(/root/reference/psoap/sample_parallel.py:126-198): ``fl`` and ``sigma`` stay
resident in HBM, ``max_batch`` scratch matrices replace the single ``V11``
(:163), and a whole batch of proposals is evaluated per call.
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import _lib
from ._lib import K_NAMES, Timings, as_f64, check, dptr


class ChunkHandle:
    def __init__(self, fl, sigma, max_batch: int = 1, device: int | None = None):
        self._L = _lib.load()
        self.fl = as_f64(fl)
        self.sigma = as_f64(sigma, self.fl.shape)
        if self.fl.ndim != 1:
            raise ValueError("fl and sigma must be 1-D")
        self.N = int(self.fl.shape[0])
        self.max_batch = int(max_batch)
        self.device = _lib.default_device() if device is None else int(device)
        self._h = ctypes.c_void_p()
        check(self._L.psoap_chunk_create(ctypes.byref(self._h), self.device, self.N, dptr(self.fl),
                                         dptr(self.sigma), self.max_batch), "psoap_chunk_create")
        self._B = 0
        self.n_epochs = 0

    # -- lifetime -----------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._L.psoap_chunk_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- configuration --------------------------------------------------------------
    def set_data(self, fl, sigma):
        self.fl = as_f64(fl, (self.N,))
        self.sigma = as_f64(sigma, (self.N,))
        check(self._L.psoap_chunk_set_data(self._h, dptr(self.fl), dptr(self.sigma)), "psoap_chunk_set_data")

    def set_grid(self, lwl, epoch_index, n_epochs: int):
        """Observed-frame ln-wavelengths + epoch of every pixel, for device-side Doppler shifts."""
        lwl = as_f64(lwl, (self.N,))
        ep = np.ascontiguousarray(epoch_index, dtype=np.int32)
        if ep.shape != (self.N,):
            raise ValueError("epoch_index must have shape (N,)")
        check(self._L.psoap_chunk_set_grid(self._h, dptr(lwl), ep.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                                           int(n_epochs)), "psoap_chunk_set_grid")
        self.n_epochs = int(n_epochs)

    def set_stream_groups(self, groups: int):
        check(self._L.psoap_chunk_set_stream_groups(self._h, int(groups)), "psoap_chunk_set_stream_groups")

    def set_mode(self, mode: str | int):
        """"dag" (default): one persistent dependency-graph kernel; "staged": three kernels per panel."""
        m = {"staged": 0, "dag": 1}.get(mode, mode)
        check(self._L.psoap_chunk_set_mode(self._h, int(m)), "psoap_chunk_set_mode")

    def set_profiling(self, enabled: bool):
        check(self._L.psoap_chunk_set_profiling(self._h, int(bool(enabled))), "psoap_chunk_set_profiling")

    # -- evaluation -----------------------------------------------------------------
    def lnlike(self, lwls, gp, mu_GP: float = 1.0) -> float:
        lwls = as_f64(np.atleast_2d(lwls))
        c = lwls.shape[0]
        lwls = as_f64(lwls, (c, self.N))
        gp = as_f64(gp, (2 * c,))
        out = np.empty(1)
        check(self._L.psoap_lnlike(self._h, c, dptr(lwls), dptr(gp), float(mu_GP), dptr(out)), "psoap_lnlike")
        return float(out[0])

    def lnlike_batch(self, lwls, gps, mu_GP: float = 1.0) -> np.ndarray:
        lwls = as_f64(lwls)
        if lwls.ndim != 3 or lwls.shape[2] != self.N:
            raise ValueError("lwls must have shape (B, c, N)")
        B, c, _ = lwls.shape
        gps = as_f64(gps, (B, 2 * c))
        out = np.empty(B)
        check(self._L.psoap_lnlike_batch(self._h, B, c, dptr(lwls), dptr(gps), float(mu_GP), dptr(out)),
              "psoap_lnlike_batch")
        return out

    def upload(self, lwls, gps, mu_GP: float = 1.0):
        lwls = as_f64(lwls)
        B, c, _ = lwls.shape
        lwls = as_f64(lwls, (B, c, self.N))
        gps = as_f64(gps, (B, 2 * c))
        check(self._L.psoap_batch_upload(self._h, B, c, dptr(lwls), dptr(gps), float(mu_GP)), "psoap_batch_upload")
        self._B = B

    def upload_velocities(self, velocities, gps, mu_GP: float = 1.0):
        vel = as_f64(velocities)
        if vel.ndim != 3 or vel.shape[2] != self.n_epochs:
            raise ValueError("velocities must have shape (B, c, n_epochs); call set_grid first")
        B, c, _ = vel.shape
        gps = as_f64(gps, (B, 2 * c))
        check(self._L.psoap_batch_upload_velocities(self._h, B, c, dptr(vel), dptr(gps), float(mu_GP)),
              "psoap_batch_upload_velocities")
        self._B = B

    def eval(self):
        check(self._L.psoap_batch_eval(self._h), "psoap_batch_eval")

    def fetch(self) -> np.ndarray:
        out = np.empty(self._B)
        check(self._L.psoap_batch_fetch(self._h, dptr(out)), "psoap_batch_fetch")
        return out

    def sync(self):
        check(self._L.psoap_chunk_sync(self._h), "psoap_chunk_sync")

    def predict(self, mode: int, lwls, lwls_predict, mu_c, gp, want_sigma=True):
        """``predict_*`` on this chunk's resident ``fl`` / ``sigma`` (include/psoap_gp.h: psoap_chunk_predict;
        mode 0 components, 1 sum, 2 predict_f).  The workspace stays with the handle.
        ``want_sigma``: True -> (mu, Sigma); "diag" -> (mu, diag(Sigma)) without ever forming Sigma
        (psoap_chunk_predict_var); False -> mu."""
        lwls = as_f64(np.atleast_2d(lwls))
        pred = as_f64(np.atleast_2d(lwls_predict))
        c = lwls.shape[0]
        lwls = as_f64(lwls, (c, self.N))
        M = pred.shape[1]
        pred = as_f64(pred, (c, M))
        mu_c = as_f64(mu_c)
        gp = as_f64(gp, (2 * c,))
        R = c * M if mode == 0 else M
        mu = np.empty(R)
        status = ctypes.c_int(0)
        if isinstance(want_sigma, str):
            if want_sigma != "diag":
                raise ValueError('want_sigma must be True, False or "diag"')
            var = np.empty(R)
            check(self._L.psoap_chunk_predict_var(self._h, int(mode), c, M, dptr(lwls), dptr(pred), dptr(mu_c), dptr(gp),
                                                  dptr(mu), dptr(var), ctypes.byref(status)), "psoap_chunk_predict_var")
            if status.value != 0:
                raise np.linalg.LinAlgError("data covariance matrix is not positive definite")
            return mu, var
        Sigma = np.empty((R, R)) if want_sigma else None
        check(self._L.psoap_chunk_predict(self._h, int(mode), c, M, dptr(lwls), dptr(pred), dptr(mu_c), dptr(gp),
                                          dptr(mu), None if Sigma is None else dptr(Sigma), ctypes.byref(status)),
              "psoap_chunk_predict")
        if status.value != 0:
            raise np.linalg.LinAlgError("data covariance matrix is not positive definite")
        return (mu, Sigma) if want_sigma else mu

    def predict_timings(self) -> dict:
        t = _lib.PredictTimings()
        check(self._L.psoap_chunk_predict_timings(self._h, ctypes.byref(t)), "psoap_chunk_predict_timings")
        return t.as_dict()

    def timings(self) -> dict:
        t = Timings()
        check(self._L.psoap_chunk_get_timings(self._h, ctypes.byref(t)), "psoap_chunk_get_timings")
        out = {"total_ms": t.total_ms}
        for k, name in enumerate(K_NAMES):
            out[name] = {"ms": t.ms[k], "launches": int(t.launches[k]), "flops": t.flops[k], "bytes": t.bytes[k]}
        return out


class ChunkGroup:
    """Several chunk handles (one device, one component count; sizes may differ) evaluated by ONE launch
    of the persistent kernel over the heterogeneous batch -- the one-GPU form of the reference's one
    worker process per chunk (sample_parallel.py:258-278).  ``upload*`` on every member, ``eval()``,
    ``fetch()`` on every member.  Small chunks cannot fill the device one at a time; together the
    matrices of all chunks hide each other's dependency chains."""

    def __init__(self, handles):
        self.handles = list(handles)
        if not self.handles:
            raise ValueError("a group needs at least one chunk handle")
        self._L = _lib.load()
        self._g = ctypes.c_void_p()
        arr = (ctypes.c_void_p * len(self.handles))(*[h._h for h in self.handles])
        check(self._L.psoap_group_create(ctypes.byref(self._g), arr, len(self.handles)), "psoap_group_create")

    def eval(self):
        check(self._L.psoap_group_eval(self._g), "psoap_group_eval")

    def stats(self) -> dict:
        """task-list builds (batch sizes changed) and record refreshes (a member changed its proposal slot) so far"""
        a, b = ctypes.c_longlong(0), ctypes.c_longlong(0)
        check(self._L.psoap_group_stats(self._g, ctypes.byref(a), ctypes.byref(b)), "psoap_group_stats")
        return {"plan_builds": a.value, "record_refreshes": b.value}

    def close(self):
        if getattr(self, "_g", None) is not None and self._g:
            self._L.psoap_group_destroy(self._g)
            self._g = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def microbench(device: int | None = None) -> dict:
    """Measured ceilings of the device (libpsoap_bench.so, include/psoap_bench.h): never part of the product path."""
    L = _lib.load_bench()
    check = _lib.check_bench
    dev = _lib.default_device() if device is None else device
    tf = ctypes.c_double()
    w = ctypes.c_double()
    c = ctypes.c_double()
    check(L.psoap_microbench_mfma_f64(dev, ctypes.byref(tf)), "psoap_microbench_mfma_f64")
    check(L.psoap_microbench_hbm(dev, ctypes.byref(w), ctypes.byref(c)), "psoap_microbench_hbm")
    t1 = ctypes.c_double()
    t0 = ctypes.c_double()
    # variants 9 / 8: the production engine (LDS-DMA staging) on L2-resident / HBM-streamed operands
    check(L.psoap_microbench_tile_engine(dev, 9, ctypes.byref(t1)), "psoap_microbench_tile_engine")
    check(L.psoap_microbench_tile_engine(dev, 8, ctypes.byref(t0)), "psoap_microbench_tile_engine")
    return {"mfma_f64_tflops": tf.value, "hbm_write_gbs": w.value, "hbm_copy_gbs": c.value,
            "tile_engine_l2_tflops": t1.value, "tile_engine_hbm_tflops": t0.value}
