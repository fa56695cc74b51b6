"""Chunk-sharded ensemble evaluation: one process per GPU, one small gather.

The reference runs one worker process per spectral chunk and the master sums the
per-chunk likelihoods (/root/reference/psoap/sample_parallel.py:258-278 fork,
:378-387 gather + ``np.sum``).  Here rank ``r`` of ``G`` owns chunks
``{k : k mod G == r}``, keeps them resident on its GPU, evaluates every walker of the
ensemble for them, and the per-(chunk, walker) log-probabilities are exchanged with a
single ``all_gather`` (RCCL over xGMI when the tensors are on the GPU, gloo in the CPU
tests).  Every rank then sums over chunks in the fixed order k = 0..n_chunks-1, so the
walker log-probabilities are bit-identical on every rank.  (Across world sizes they agree to a
few ulp: how a rank's matrices are scheduled -- and with it the order of the partial sums inside
the factorisation -- depends on how many of them share a launch.)

There is no other communication on the path: chunks are independent.
"""
from __future__ import annotations

from typing import Callable, Sequence

import numpy as np


def owned_chunks(n_chunks: int, world: int, rank: int) -> list[int]:
    """Chunk-major round-robin partition."""
    return [k for k in range(n_chunks) if k % world == rank]


def _dist():
    import torch.distributed as dist
    return dist


class _GatherBuffers:
    """Send / receive tensors of one (n_chunks, B, world, backend, device) gather: allocated once, reused every step
    (round 3 allocated two tensors and went numpy -> torch -> device -> .cpu() per step)."""

    def __init__(self, per: int, B: int, world: int, dev):
        import torch
        self.send = torch.zeros((per, B), dtype=torch.float64, device=dev)
        self.recv = torch.empty((world * per, B), dtype=torch.float64, device=dev)
        # pinned staging for the device case: one H2D and one D2H of a few KB, no pageable copies
        self.cuda = dev.type == "cuda"
        if self.cuda:
            self.h_send = torch.zeros((per, B), dtype=torch.float64).pin_memory()
            self.h_recv = torch.empty((world * per, B), dtype=torch.float64).pin_memory()


_buffers: dict = {}
_host_group = {"pg": None}


def host_group():
    """A gloo process group beside the default (RCCL) one, created on first use -- collectively: every rank reaches its
    first host-side gather at the same point.  What a gather uses WHILE a resident stream launch holds the device: the
    lnprobs are in pinned host memory already, and a device collective would wait for the launch to leave (measured:
    90.6 ms per 32-walker step against 37.9, whatever number of workgroup slots the launch leaves free --
    profiles/r5_gather_beside_stream.txt)."""
    dist = _dist()
    if _host_group["pg"] is None:
        _host_group["pg"] = dist.new_group(backend="gloo")
    return _host_group["pg"]


def gather_chunk_lnprobs(local: np.ndarray, n_chunks: int, world: int, rank: int,
                         device_index: int | None = None, force_collective: bool = False,
                         on_host: bool = False) -> np.ndarray:
    """all_gather the (n_local, B) block of every rank into the (n_chunks, B) table.

    ``local[i]`` is the lnprob vector of chunk ``owned_chunks(...)[i]``.  Ranks with fewer
    chunks are padded so the collective has equal counts.  ``force_collective``: go through the
    process group even when ``world == 1`` (a one-rank RCCL communicator exercises the same code).
    ``on_host``: with an RCCL default group, exchange through the gloo side group instead (``host_group``) -- for gathers
    issued while a resident stream launch is on the device.
    """
    local = np.ascontiguousarray(local, dtype=np.float64)
    if local.ndim != 2:
        raise ValueError("local must have shape (n_local, B)")
    B = local.shape[1]
    mine = owned_chunks(n_chunks, world, rank)
    if local.shape[0] != len(mine):
        raise ValueError(f"rank {rank} owns {len(mine)} chunks but got {local.shape[0]} rows")
    if world == 1 and not force_collective:
        return local.copy()
    import torch
    dist = _dist()
    per = -(-n_chunks // world)
    use_cuda = dist.get_backend() == "nccl" and not on_host
    group = host_group() if (on_host and dist.get_backend() == "nccl") else None
    dev = torch.device("cuda", device_index if device_index is not None else torch.cuda.current_device()) \
        if use_cuda else torch.device("cpu")
    key = (per, B, world, dist.get_backend() if group is None else "gloo-side", str(dev))
    buf = _buffers.get(key)
    if buf is None:
        buf = _buffers[key] = _GatherBuffers(per, B, world, dev)
    if buf.cuda:
        buf.h_send.zero_()
        buf.h_send[:len(mine)] = torch.from_numpy(local)
        buf.send.copy_(buf.h_send, non_blocking=True)
        dist.all_gather_into_tensor(buf.recv, buf.send)
        buf.h_recv.copy_(buf.recv, non_blocking=True)
        torch.cuda.current_stream(dev).synchronize()
        table = buf.h_recv.numpy().reshape(world, per, B)
    else:
        buf.send.zero_()
        buf.send[:len(mine)] = torch.from_numpy(local)
        dist.all_gather_into_tensor(buf.recv, buf.send, group=group)
        table = buf.recv.numpy().reshape(world, per, B)
    out = np.empty((n_chunks, B))
    for k in range(n_chunks):
        out[k] = table[k % world, k // world]
    return out


def release_gather_buffers():
    """drop the cached tensors and the host-side group (before ``destroy_process_group``)"""
    _buffers.clear()
    _host_group["pg"] = None


def sum_over_chunks(table: np.ndarray) -> np.ndarray:
    """Fixed-order sum k = 0..n_chunks-1 (sample_parallel.py:387 sums the chunk lnprobs)."""
    total = np.zeros(table.shape[1])
    for k in range(table.shape[0]):
        total = total + table[k]
    return total


def gather_and_sum(lnp: np.ndarray, world: int, device_index: int | None = None) -> np.ndarray:
    """One chunk per rank: gather the (B,) vectors of all ranks and sum them in rank order."""
    lnp = np.ascontiguousarray(lnp, dtype=np.float64)
    if world == 1:
        return lnp.copy()
    rank = _dist().get_rank()
    table = gather_chunk_lnprobs(lnp[None, :], world, world, rank, device_index)
    return sum_over_chunks(table)


class SharedDeviceLock:
    """Dry runs only: several ranks on ONE GPU (``--backend gloo`` on a one-GPU box).  The device then time-slices the
    ranks' persistent kernels (compute wave save / restore), and a workgroup that is suspended between its stores and the
    release that publishes them is not covered by the kernel's hand-off protocol -- measured with 8 ranks on one MI355X:
    a wrong (chunk, walker) value in about every second run, with the round-3 library as well; never with one process
    per GPU, which is the production layout.  This lock (``flock`` on a file named after the job) lets one
    rank at a time use the device: take it around everything from a launch to the fetch of its results, never around
    a collective."""

    def __init__(self, device_index: int = 0, job: str | None = None):
        import os
        import tempfile
        job = job or os.environ.get("TORCHELASTIC_RUN_ID") or os.environ.get("MASTER_PORT") or "solo"
        self.path = os.path.join(tempfile.gettempdir(), f"psoap_shared_gpu{device_index}_{job}.lock")
        self._fh = None

    def __enter__(self):
        import fcntl
        self._fh = open(self.path, "w")
        fcntl.flock(self._fh, fcntl.LOCK_EX)
        return self

    def __exit__(self, *exc):
        import fcntl
        fcntl.flock(self._fh, fcntl.LOCK_UN)
        self._fh.close()
        self._fh = None


class _NoLock:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        pass


class EnsembleEvaluator:
    """Evaluates ``lnprob(walker) = sum_k lnlike(chunk k, walker)`` for a walker ensemble.

    ``evaluate(k, proposals) -> (B,)`` computes the likelihood of every proposal for
    chunk ``k``; in production it is ``ChunkHandle.lnlike_batch`` of the chunk resident on
    this rank's GPU (see ``from_chunks``).
    """

    def __init__(self, n_chunks: int, evaluate: Callable[[int, object], np.ndarray],
                 world: int = 1, rank: int = 0, device_index: int | None = None):
        self.n_chunks = int(n_chunks)
        self.world = int(world)
        self.rank = int(rank)
        self.device_index = device_index
        self.evaluate = evaluate
        self.mine = owned_chunks(self.n_chunks, self.world, self.rank)
        self.device_lock = _NoLock()       # SharedDeviceLock for dry runs with several ranks on one GPU

    @classmethod
    def from_chunks(cls, chunks: Sequence, max_batch: int, world: int = 1, rank: int = 0,
                    device_index: int | None = None) -> "EnsembleEvaluator":
        """``chunks``: objects with ``fl``, ``sigma`` (all n_chunks of them; only the owned
        ones are uploaded).  Proposals are ``(lwls[k] (B,c,N_k), gps (B,2c))`` per chunk."""
        from .chunk import ChunkHandle
        mine = owned_chunks(len(chunks), world, rank)
        handles = {k: ChunkHandle(chunks[k].fl, chunks[k].sigma, max_batch=max_batch, device=device_index)
                   for k in mine}

        def evaluate(k, proposals):
            lwls, gps = proposals[k]
            return handles[k].lnlike_batch(lwls, gps)

        ev = cls(len(chunks), evaluate, world, rank, device_index)
        ev.handles = handles
        if len(mine) > 1:
            from .chunk import ChunkGroup
            ev.group = ChunkGroup([handles[k] for k in mine])     # several chunks on this GPU: one launch
        return ev

    def lnprob(self, proposals) -> np.ndarray:
        if getattr(self, "handles", None):
            with self.device_lock:
                self.upload(proposals)
                self.launch()
                local = self.fetch_local()
            return self._gather_and_sum(local)
        local = [np.asarray(self.evaluate(k, proposals), dtype=np.float64) for k in self.mine]
        return self._gather_and_sum(local)

    # split-phase form (device-resident chunks only): the proposals of the NEXT ensemble step may be uploaded
    # between launch() and collect() -- every chunk handle keeps two proposal batches, so the copy runs
    # while the current step is being factored
    def upload(self, proposals):
        for k in self.mine:
            lwls, gps = proposals[k]
            self.handles[k].upload(lwls, gps)

    def launch(self):
        group = getattr(self, "group", None)
        if group is not None:
            group.eval()
        else:
            for k in self.mine:
                self.handles[k].eval()

    def fetch_local(self) -> list:
        """the results of this rank's chunks (blocks until the launch is through); no collective"""
        return [np.asarray(self.handles[k].fetch(), dtype=np.float64) for k in self.mine]

    def collect(self) -> np.ndarray:
        return self._gather_and_sum(self.fetch_local())

    def _gather_and_sum(self, local) -> np.ndarray:
        B = local[0].shape[0] if local else 0
        if self.world > 1 and not local:
            raise ValueError("every rank must own at least one chunk (n_chunks >= world)")
        block = np.stack(local) if local else np.zeros((0, B))
        self.table = gather_chunk_lnprobs(block, self.n_chunks, self.world, self.rank, self.device_index)
        return sum_over_chunks(self.table)

    def close(self):
        if getattr(self, "group", None) is not None:
            self.group.close()
        for h in getattr(self, "handles", {}).values():
            h.close()
