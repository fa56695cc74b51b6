"""The per-chunk ``lnprob(p)`` of the parallel sampler with everything after the parameter
plumbing on the device.

``ChunkWorker.lnprob(p)`` reproduces ``Worker.lnprob`` (/root/reference/psoap/sample_parallel.py:168-198):
``convert_vector`` -> orbit velocities -> |v| >= c_kms -> -inf -> ``replicate_wls`` -> ``lnlike[model]``.
Orbit evaluation (batched Kepler solve), Doppler shift and likelihood run in the HIP library; per proposal
the host ships ``n_orb + 2c`` doubles and receives one.  ``lnprob_batch`` evaluates a whole ensemble.
"""
from __future__ import annotations

import numpy as np

from . import _lib
from ._lib import as_f64, check, dptr
from .chunk import ChunkHandle
from .utils import MODEL_ID, N_COMPONENTS, convert_vectors


class ChunkWorker:
    def __init__(self, model: str, lwl, fl, sigma, epoch_index, dates, fix_params=(), defaults=None,
                 max_batch: int = 1, device: int | None = None, soften: float = 1.0):
        """lwl/fl/sigma: the chunk's masked, flattened arrays (N,); epoch_index (N,): epoch of each pixel;
        dates (n_epochs,): ``date1D``; ``soften`` scales sigma (sample_parallel.py:141)."""
        self.model = model
        self.fix_params = list(fix_params)
        self.defaults = dict(defaults or {})
        self.handle = ChunkHandle(fl, as_f64(sigma) * soften, max_batch=max_batch, device=device)
        dates = as_f64(dates)
        self.handle.set_grid(lwl, epoch_index, dates.shape[0])
        check(self.handle._L.psoap_chunk_set_dates(self.handle._h, dptr(dates), dates.shape[0]), "psoap_chunk_set_dates")

    def close(self):
        self.handle.close()

    def upload_proposals(self, ps, mu_GP: float = 1.0) -> None:
        """Ship fitted parameter vectors (B, n_fit); the orbit solve and the Doppler shift are queued on the
        chunk's stream.  Follow with ``handle.eval()`` (or a ``ChunkGroup.eval()``) and ``handle.fetch()``."""
        p_orb, p_gp = convert_vectors(np.atleast_2d(ps), self.model, self.fix_params, **self.defaults)
        B = p_orb.shape[0]
        h = self.handle
        check(h._L.psoap_batch_upload_orbits(h._h, B, MODEL_ID[self.model], dptr(p_orb), dptr(p_gp), float(mu_GP)),
              "psoap_batch_upload_orbits")
        h._B = B

    # -- streamed form: ONE resident launch across sampler iterations (include/psoap_gp.h: psoap_stream_*) ----------
    def stream_open(self, lanes: int | None = None, scheme: int = -1):
        self.handle.stream_open(N_COMPONENTS[self.model], lanes, scheme)

    def stream_submit(self, ps, mu_GP: float = 1.0) -> np.ndarray:
        """fitted parameter vectors (n, n_fit) -> tickets; Kepler solve, |v| >= c rule, Doppler shift and likelihood all
        inside the resident launch (``Worker.lnprob`` for n proposals, sample_parallel.py:168-198)"""
        p_orb, p_gp = convert_vectors(np.atleast_2d(ps), self.model, self.fix_params, **self.defaults)
        return self.handle.stream_submit_orbits(MODEL_ID[self.model], p_orb, p_gp, mu_GP)

    def stream_fetch(self, tickets) -> np.ndarray:
        return self.handle.stream_fetch(tickets)

    def stream_close(self):
        self.handle.stream_close()

    def lnprob_batch(self, ps, mu_GP: float = 1.0) -> np.ndarray:
        self.upload_proposals(ps, mu_GP)
        self.handle.eval()
        return self.handle.fetch()

    def lnprob(self, p, mu_GP: float = 1.0) -> float:
        return float(self.lnprob_batch(np.atleast_2d(p), mu_GP)[0])
