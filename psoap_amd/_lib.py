"""ctypes binding of include/psoap_gp.h.  No CPU fallback: if the HIP library is
missing or a call fails, an exception is raised."""
from __future__ import annotations

import ctypes
import os

import numpy as np

from .build import LIB_PATH

_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int)
_i32p = ctypes.POINTER(ctypes.c_int32)
_vp = ctypes.c_void_p

K_CLASSES = 6
K_NAMES = ("fill", "panel_update", "potrf", "trsm", "misc", "dag")


class Timings(ctypes.Structure):
    _fields_ = [("ms", ctypes.c_double * K_CLASSES),
                ("launches", ctypes.c_int64 * K_CLASSES),
                ("flops", ctypes.c_double * K_CLASSES),
                ("bytes", ctypes.c_double * K_CLASSES),
                ("total_ms", ctypes.c_double)]


class PredictTimings(ctypes.Structure):
    _fields_ = [(k, ctypes.c_double) for k in ("device_ms", "factor_ms", "sigma_ms", "download_ms", "total_ms", "flops")]

    def as_dict(self) -> dict:
        return {k: getattr(self, k) for k, _ in self._fields_}


class PsoapError(RuntimeError):
    pass


# name -> (restype, argtypes); this table is also what tests check against the header
SIGNATURES = {
    "psoap_version": (ctypes.c_int, []),
    "psoap_last_error": (ctypes.c_char_p, []),
    "psoap_device_count": (ctypes.c_int, [_ip]),
    "psoap_share_stats": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(ctypes.c_longlong), ctypes.c_int]),
    "psoap_chunk_create": (ctypes.c_int, [ctypes.POINTER(_vp), ctypes.c_int, ctypes.c_int, _dp, _dp, ctypes.c_int]),
    "psoap_chunk_destroy": (ctypes.c_int, [_vp]),
    "psoap_chunk_set_data": (ctypes.c_int, [_vp, _dp, _dp]),
    "psoap_chunk_set_grid": (ctypes.c_int, [_vp, _dp, _i32p, ctypes.c_int]),
    "psoap_lnlike": (ctypes.c_int, [_vp, ctypes.c_int, _dp, _dp, ctypes.c_double, _dp]),
    "psoap_lnlike_batch": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, _dp, _dp, ctypes.c_double, _dp]),
    "psoap_batch_upload": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, _dp, _dp, ctypes.c_double]),
    "psoap_batch_upload_velocities": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, _dp, _dp, ctypes.c_double]),
    "psoap_chunk_set_dates": (ctypes.c_int, [_vp, _dp, ctypes.c_int]),
    "psoap_batch_upload_orbits": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, _dp, _dp, ctypes.c_double]),
    "psoap_orbit_velocities": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, _dp, ctypes.c_int, _dp, _dp]),
    "psoap_batch_eval": (ctypes.c_int, [_vp]),
    "psoap_batch_fetch": (ctypes.c_int, [_vp, _dp]),
    "psoap_chunk_sync": (ctypes.c_int, [_vp]),
    "psoap_fill_sym": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, _dp, _dp, _dp, _dp]),
    "psoap_fill_cross": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, _dp, _dp, ctypes.c_double,
                                        ctypes.c_double, _dp]),
    "psoap_predict": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _dp, _dp,
                                     _dp, _dp, _dp, _dp, _dp, _dp, _ip]),
    "psoap_chunk_predict": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _dp, _dp, _dp, _dp, _dp, _dp,
                                           _ip]),
    "psoap_chunk_predict_var": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _dp, _dp, _dp, _dp, _dp, _dp,
                                               _ip]),
    "psoap_chunk_predict_release": (ctypes.c_int, [_vp]),
    "psoap_chunk_predict_timings": (ctypes.c_int, [_vp, ctypes.POINTER(PredictTimings)]),
    "psoap_predictor_create": (ctypes.c_int, [ctypes.POINTER(_vp), ctypes.c_int]),
    "psoap_predictor_run": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _dp, _dp, _dp,
                                           _dp, _dp, _dp, _dp, _dp, _ip]),
    "psoap_predictor_run_var": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _dp, _dp, _dp,
                                               _dp, _dp, _dp, _dp, _dp, _ip]),
    "psoap_predictor_timings": (ctypes.c_int, [_vp, ctypes.POINTER(PredictTimings)]),
    "psoap_predictor_destroy": (ctypes.c_int, [_vp]),
    "psoap_calibrate": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                       ctypes.c_double, ctypes.c_double, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp,
                                       ctypes.c_double, _dp, _dp, _ip]),
    "psoap_calibrate_explicit": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double,
                                                ctypes.c_double, _dp, _dp, _dp, _dp, _dp, _dp, ctypes.c_double, _dp, _dp,
                                                _ip]),
    "psoap_group_create": (ctypes.c_int, [ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.c_int]),
    "psoap_group_eval": (ctypes.c_int, [_vp]),
    "psoap_group_destroy": (ctypes.c_int, [_vp]),
    "psoap_group_stats": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_longlong), ctypes.POINTER(ctypes.c_longlong)]),
    "psoap_stream_open": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "psoap_stream_submit": (ctypes.c_int, [_vp, ctypes.c_int, _dp, _dp, ctypes.c_double, ctypes.POINTER(ctypes.c_longlong)]),
    "psoap_stream_submit_velocities": (ctypes.c_int, [_vp, ctypes.c_int, _dp, _dp, ctypes.c_double, ctypes.POINTER(ctypes.c_longlong)]),
    "psoap_stream_submit_orbits": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, _dp, _dp, ctypes.c_double,
                                                  ctypes.POINTER(ctypes.c_longlong)]),
    "psoap_stream_fetch": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.POINTER(ctypes.c_longlong), _dp]),
    "psoap_stream_wait_any": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.POINTER(ctypes.c_longlong), _ip]),
    "psoap_stream_ready": (ctypes.c_int, [_vp, ctypes.c_longlong, _ip]),
    "psoap_stream_close": (ctypes.c_int, [_vp]),
    "psoap_stream_pause": (ctypes.c_int, [_vp]),
    "psoap_stream_last_launch": (ctypes.c_int, [_vp, _dp, ctypes.POINTER(ctypes.c_longlong)]),
    "psoap_stream_stats": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_longlong), ctypes.POINTER(ctypes.c_longlong),
                                          ctypes.POINTER(ctypes.c_longlong), _ip, ctypes.POINTER(ctypes.c_longlong)]),
    "psoap_stream_plan": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp, ctypes.c_longlong,
                                         ctypes.POINTER(ctypes.c_longlong), ctypes.POINTER(ctypes.c_longlong),
                                         ctypes.POINTER(ctypes.c_longlong), _ip]),
    "psoap_stream_tasklog": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.POINTER(ctypes.c_uint64), ctypes.c_longlong]),
    "psoap_stream_tasks": (ctypes.c_int, [_vp, _vp, ctypes.c_longlong, ctypes.POINTER(ctypes.c_longlong)]),
    "psoap_chunk_set_profiling": (ctypes.c_int, [_vp, ctypes.c_int]),
    "psoap_chunk_get_timings": (ctypes.c_int, [_vp, ctypes.POINTER(Timings)]),
    "psoap_chunk_set_stream_groups": (ctypes.c_int, [_vp, ctypes.c_int]),
    "psoap_chunk_set_mode": (ctypes.c_int, [_vp, ctypes.c_int]),
    "psoap_chunk_dag_tasklog": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_uint64), ctypes.c_longlong]),
    "psoap_chunk_dag_tasks": (ctypes.c_int, [_vp, _vp, ctypes.c_longlong, ctypes.POINTER(ctypes.c_longlong)]),
    "psoap_dag_plan": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp, ctypes.c_longlong,
                                      ctypes.POINTER(ctypes.c_longlong), ctypes.POINTER(ctypes.c_longlong),
                                      ctypes.POINTER(ctypes.c_longlong), ctypes.POINTER(ctypes.c_uint32)]),
    "psoap_dag_plan_multi": (ctypes.c_int, [ctypes.c_int, _ip, ctypes.c_int, _vp, ctypes.c_longlong,
                                      ctypes.POINTER(ctypes.c_longlong), ctypes.POINTER(ctypes.c_longlong),
                                      ctypes.POINTER(ctypes.c_longlong), ctypes.POINTER(ctypes.c_uint32)]),
    "psoap_dag_plan_aug": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp,
                                          ctypes.c_longlong, ctypes.POINTER(ctypes.c_longlong),
                                          ctypes.POINTER(ctypes.c_longlong), ctypes.POINTER(ctypes.c_longlong),
                                          ctypes.POINTER(ctypes.c_uint32)]),
    "psoap_dag_pick_workers": (ctypes.c_int, [ctypes.c_int, _ip, ctypes.c_int, ctypes.c_int, ctypes.c_int, _ip]),
    "psoap_dag_plan_pool": (ctypes.c_int, [ctypes.c_int, _ip, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp,
                                           ctypes.c_longlong, ctypes.POINTER(ctypes.c_longlong), ctypes.POINTER(ctypes.c_uint32),
                                           ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32),
                                           ctypes.POINTER(ctypes.c_uint32), _ip, ctypes.POINTER(ctypes.c_longlong)]),
}

# include/psoap_bench.h (libpsoap_bench.so): measurement kernels, loaded by bench.py / tools / one GPU test only
BENCH_SIGNATURES = {
    "psoap_bench_last_error": (ctypes.c_char_p, []),
    "psoap_microbench_exp_check": (ctypes.c_int, [ctypes.c_int, ctypes.c_longlong, _dp, ctypes.POINTER(ctypes.c_longlong)]),
    "psoap_microbench_mfma_f64": (ctypes.c_int, [ctypes.c_int, _dp]),
    "psoap_microbench_tile_engine": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, _dp]),
    "psoap_microbench_potrf": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, _dp]),
    "psoap_microbench_hbm": (ctypes.c_int, [ctypes.c_int, _dp, _dp]),
    "psoap_microbench_mix": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _dp]),
    "psoap_litmus_l2": (ctypes.c_int, [ctypes.c_int] * 7 + [ctypes.POINTER(ctypes.c_ulonglong)]),
    "psoap_litmus_writeback": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_ulonglong)]),
}

_lib = None
_bench = None


def load():
    """Load libpsoap_gp.so (built by psoap_amd.build).  Raises if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("PSOAP_GP_LIB", LIB_PATH)   # override: A/B runs of two builds of the HIP library
    if not os.path.exists(path):
        raise PsoapError(
            f"{path} not found: build it with `python -m psoap_amd.build` "
            "(hipcc, gfx950).  There is no CPU fallback.")
    L = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(L, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def load_bench():
    """Load libpsoap_bench.so (psoap_amd.build.build_bench): the micro-benchmarks behind the measured peaks."""
    global _bench
    if _bench is not None:
        return _bench
    from .build import BENCH_LIB_PATH
    path = os.environ.get("PSOAP_BENCH_LIB", BENCH_LIB_PATH)
    if not os.path.exists(path):
        raise PsoapError(f"{path} not found: build it with `python -m psoap_amd.build` (hipcc, gfx950)")
    L = ctypes.CDLL(path)
    for name, (res, args) in BENCH_SIGNATURES.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _bench = L
    return L


def check_bench(rc: int, what: str = ""):
    if rc != 0:
        msg = load_bench().psoap_bench_last_error()
        raise PsoapError(f"{what}: {msg.decode() if msg else 'error'} (rc={rc})")


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().psoap_last_error()
        raise PsoapError(f"{what}: {msg.decode() if msg else 'error'} (rc={rc})")


def dptr(a: np.ndarray):
    return a.ctypes.data_as(_dp)


def as_f64(a, shape=None) -> np.ndarray:
    out = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None and out.shape != tuple(shape):
        raise ValueError(f"expected shape {tuple(shape)}, got {out.shape}")
    return out


SHARE_NAMES = ("procs", "dag_launches", "tainted", "retries", "staged_fallbacks", "staged_policy", "moved_tasks", "moved_xcd",
               "lock_acquisitions", "lock_wait_us", "stream_resubmits", "lock_enabled")


def share_stats(device: int | None = None) -> dict:
    """What sharing the GPU with other processes has cost this process so far (include/psoap_gp.h: PSOAP_SHARE_*):
    persistent launches, how many were disturbed by the device's scheduler (``tainted``) and run again (``retries``) or
    sent down the staged path (``staged_fallbacks``; ``staged_policy``: from the start), time spent waiting for the device."""
    out = (ctypes.c_longlong * len(SHARE_NAMES))()
    check(load().psoap_share_stats(default_device() if device is None else int(device), out, len(SHARE_NAMES)),
          "psoap_share_stats")
    return dict(zip(SHARE_NAMES, (int(v) for v in out)))


def default_device() -> int:
    """PSOAP_DEVICE, else LOCAL_RANK (one process per GPU), else 0."""
    for key in ("PSOAP_DEVICE", "LOCAL_RANK"):
        if key in os.environ:
            return int(os.environ[key])
    return 0
