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

    # -- streamed evaluation (include/psoap_gp.h: psoap_stream_*) ----------------------------------------
    def stream_open(self, c: int, lanes: int | None = None, scheme: int = -1):
        """Keep ONE launch of the persistent kernel resident: ``stream_submit`` hands proposals to free lanes,
        ``stream_fetch`` returns their lnprob while the other lanes keep the device busy (the back-to-back iterations
        of /root/reference/psoap/sample_parallel.py:434-438).  Batch calls are refused until ``stream_close``."""
        lanes = self.max_batch if lanes is None else int(lanes)
        check(self._L.psoap_stream_open(self._h, int(c), lanes, int(scheme)), "psoap_stream_open")
        self._stream_c = int(c)
        self.stream_lanes = lanes

    def stream_submit(self, lwls, gps, mu_GP: float = 1.0) -> np.ndarray:
        """``lwls`` (n, c, N), ``gps`` (n, 2c) -> tickets (n,) int64"""
        lwls = as_f64(lwls)
        if lwls.ndim != 3 or lwls.shape[1:] != (self._stream_c, self.N):
            raise ValueError(f"lwls must have shape (n, {self._stream_c}, {self.N})")
        n = lwls.shape[0]
        gps = as_f64(gps, (n, 2 * self._stream_c))
        tickets = np.empty(n, dtype=np.int64)
        check(self._L.psoap_stream_submit(self._h, n, dptr(lwls), dptr(gps), float(mu_GP),
                                          tickets.ctypes.data_as(ctypes.POINTER(ctypes.c_longlong))), "psoap_stream_submit")
        return tickets

    def stream_submit_velocities(self, velocities, gps, mu_GP: float = 1.0) -> np.ndarray:
        """``velocities`` (n, c, n_epochs): the resident launch shifts the chunk's grid itself (``set_grid`` before
        ``stream_open``)"""
        vel = as_f64(velocities)
        if vel.ndim != 3 or vel.shape[1:] != (self._stream_c, self.n_epochs):
            raise ValueError(f"velocities must have shape (n, {self._stream_c}, {self.n_epochs}); call set_grid first")
        n = vel.shape[0]
        gps = as_f64(gps, (n, 2 * self._stream_c))
        tickets = np.empty(n, dtype=np.int64)
        check(self._L.psoap_stream_submit_velocities(self._h, n, dptr(vel), dptr(gps), float(mu_GP),
                                                     tickets.ctypes.data_as(ctypes.POINTER(ctypes.c_longlong))),
              "psoap_stream_submit_velocities")
        return tickets

    def stream_submit_orbits(self, model_id: int, p_orb, gps, mu_GP: float = 1.0) -> np.ndarray:
        """orbital parameters (n, n_orb): Kepler solve, |v| >= c rule and Doppler shift inside the resident launch"""
        p_orb = as_f64(np.atleast_2d(p_orb))
        n = p_orb.shape[0]
        from .utils import MODEL_ID, n_params_orb
        width = {MODEL_ID[m]: n_params_orb[m] for m in MODEL_ID}.get(int(model_id))
        if width is None or p_orb.shape[1] != width:       # (the library copies exactly that many doubles per proposal)
            raise ValueError(f"p_orb must have shape (n, {width}) for orbit model {model_id}")
        gps = as_f64(gps, (n, 2 * self._stream_c))
        tickets = np.empty(n, dtype=np.int64)
        check(self._L.psoap_stream_submit_orbits(self._h, n, int(model_id), dptr(p_orb), dptr(gps), float(mu_GP),
                                                 tickets.ctypes.data_as(ctypes.POINTER(ctypes.c_longlong))),
              "psoap_stream_submit_orbits")
        return tickets

    def stream_fetch(self, tickets) -> np.ndarray:
        tickets = np.ascontiguousarray(tickets, dtype=np.int64)
        out = np.empty(tickets.shape[0])
        check(self._L.psoap_stream_fetch(self._h, tickets.shape[0],
                                         tickets.ctypes.data_as(ctypes.POINTER(ctypes.c_longlong)), dptr(out)),
              "psoap_stream_fetch")
        return out

    def stream_wait_any(self, tickets) -> int:
        """index of a ticket whose result has arrived (blocks until one has)"""
        tickets = np.ascontiguousarray(tickets, dtype=np.int64)
        w = ctypes.c_int(-1)
        check(self._L.psoap_stream_wait_any(self._h, tickets.shape[0],
                                            tickets.ctypes.data_as(ctypes.POINTER(ctypes.c_longlong)), ctypes.byref(w)),
              "psoap_stream_wait_any")
        return int(w.value)

    def stream_ready(self, ticket: int) -> bool:
        r = ctypes.c_int(0)
        check(self._L.psoap_stream_ready(self._h, int(ticket), ctypes.byref(r)), "psoap_stream_ready")
        return bool(r.value)

    def stream_stats(self) -> dict:
        a, b, c_, t = (ctypes.c_longlong(0) for _ in range(4))
        sch = ctypes.c_int(0)
        check(self._L.psoap_stream_stats(self._h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c_), ctypes.byref(sch),
                                         ctypes.byref(t)), "psoap_stream_stats")
        return {"launches": a.value, "submitted": b.value, "completed": c_.value, "scheme": sch.value,
                "tasks_per_matrix": t.value}

    def stream_pause(self):
        """The resident launch leaves now (after what is in flight) and the device is free; the next submit relaunches."""
        check(self._L.psoap_stream_pause(self._h), "psoap_stream_pause")

    def stream_last_launch(self) -> dict:
        """the resident launch ``stream_pause`` ended last: duration by HIP events on its stream, matrices completed"""
        ms, n = ctypes.c_double(0.0), ctypes.c_longlong(0)
        check(self._L.psoap_stream_last_launch(self._h, ctypes.byref(ms), ctypes.byref(n)), "psoap_stream_last_launch")
        return {"ms": ms.value, "matrices": n.value}

    def stream_close(self):
        check(self._L.psoap_stream_close(self._h), "psoap_stream_close")

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


class StreamPipeline:
    """``groups`` sub-ensembles of one walker ensemble in flight through a stream (``ChunkHandle.stream_open``): step k of
    group g is fetched and its successor submitted while the other groups keep the device busy -- neither group's
    proposals depend on another group's accept / reject (red / black halves of an ensemble; independent chains), so
    the iteration order of /root/reference/psoap/sample_parallel.py:434-438 is kept per group.

        pipe = StreamPipeline(h, c, walkers=32, groups=2)
        pipe.start(lwls, gps)                    # all groups submitted (group g delayed by g / groups of a period)
        for k in range(steps):
            lnp = pipe.step(next_lwls, next_gps) # per group: fetch, then submit the same rows of the next proposals
        last = pipe.drain()

    ``step`` returns the lnprob of the proposals submitted one call earlier, in walker order."""

    def __init__(self, handle: ChunkHandle, c: int, walkers: int, groups: int = 2, scheme: int = -1, submit=None):
        """``submit``: what turns the rows of a group into tickets -- default ``handle.stream_submit(lwls, gps, mu_GP)``;
        e.g. ``handle.stream_submit_velocities`` or ``ChunkWorker.stream_submit`` (orbital parameters in): ``start`` /
        ``step`` then take that callable's array arguments, each sliced by the group's rows."""
        if walkers % groups:
            raise ValueError("walkers must be a multiple of groups")
        self.h, self.c, self.walkers, self.groups = handle, int(c), int(walkers), int(groups)
        self.rows = [slice(g * walkers // groups, (g + 1) * walkers // groups) for g in range(groups)]
        handle.stream_open(c, walkers, scheme)
        self._submit = submit if submit is not None else handle.stream_submit
        self.tickets = [None] * groups
        self.period = None             # seconds per ensemble step, measured by calibrate()

    def _go(self, g, arrays, mu_GP):
        r = self.rows[g]
        self.tickets[g] = self._submit(*[a[r] for a in arrays], mu_GP)

    def calibrate(self, *arrays, mu_GP: float = 1.0, steps: int = 2) -> float:
        """Seconds per ensemble step in the steady state (a few untimed steps): what the start-up stagger is set from."""
        import time
        self.start(*arrays, mu_GP=mu_GP, stagger=0.0)
        self.step(*arrays, mu_GP=mu_GP)
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step(*arrays, mu_GP=mu_GP)
        self.period = (time.perf_counter() - t0) / steps
        self.drain()
        return self.period

    def start(self, *arrays, mu_GP: float = 1.0, stagger: float | None = None):
        """Submit every group, group g a little later than group g - 1, so that the groups end up ``1 / groups`` of a
        period apart and their first block rows and tails never coincide.  While only k groups are in flight they have
        the whole device and advance ``groups / k`` times as fast as in the steady state: the k-th interval is
        ``k * period / groups**2`` long (two groups: the second a QUARTER of a period after the first, which is then
        half way through).  ``stagger``: that unit, ``period / groups**2``, in seconds (default from ``calibrate``)."""
        import time
        if stagger is None:
            stagger = (self.period or 0.0) / self.groups ** 2
        t0 = time.perf_counter()
        for g in range(self.groups):
            while time.perf_counter() - t0 < 0.5 * g * (g + 1) * stagger:
                pass
            self._go(g, arrays, mu_GP)

    def step(self, *arrays, mu_GP: float = 1.0, between=None) -> np.ndarray:
        """``between(g, rows, lnp_rows)``: called for every group after its results are in and BEFORE its successor is
        submitted -- where a sampler decides accept / reject, and where several ranks exchange the group's lnprobs
        (the gather of /root/reference/psoap/sample_parallel.py:378-387: the next proposals depend on the chunk sum)."""
        out = np.empty(self.walkers)
        for g, r in enumerate(self.rows):
            out[r] = self.h.stream_fetch(self.tickets[g])
            if between is not None:
                between(g, r, out[r])
            self._go(g, arrays, mu_GP)
        return out

    def step_any_order(self, *arrays, mu_GP: float = 1.0) -> np.ndarray:
        """The same step with the groups taken in the order in which they COMPLETE (``groups == walkers``: every chain
        resubmitted the moment its result is there -- independent chains): no lane waits for the slowest matrix of a
        group.  Every group is fetched and resubmitted exactly once per call."""
        out = np.empty(self.walkers)
        left = list(range(self.groups))
        while left:
            # (a group is complete when its last ticket is: wait on one ticket per group)
            k = self.h.stream_wait_any([self.tickets[g][-1] for g in left]) if len(left) > 1 else 0
            g = left.pop(k)
            r = self.rows[g]
            out[r] = self.h.stream_fetch(self.tickets[g])
            self._go(g, arrays, mu_GP)
        return out

    def drain(self, between=None) -> np.ndarray:
        out = np.empty(self.walkers)
        for g, r in enumerate(self.rows):
            out[r] = self.h.stream_fetch(self.tickets[g])
            if between is not None:
                between(g, r, out[r])
            self.tickets[g] = None
        return out

    def close(self):
        self.h.stream_close()


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
