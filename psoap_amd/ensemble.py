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


def gather_chunk_lnprobs(local: np.ndarray, n_chunks: int, world: int, rank: int,
                         device_index: int | None = None) -> np.ndarray:
    """all_gather the (n_local, B) block of every rank into the (n_chunks, B) table.

    ``local[i]`` is the lnprob vector of chunk ``owned_chunks(...)[i]``.  Ranks with fewer
    chunks are padded so the collective has equal counts.
    """
    local = np.ascontiguousarray(local, dtype=np.float64)
    if local.ndim != 2:
        raise ValueError("local must have shape (n_local, B)")
    B = local.shape[1]
    mine = owned_chunks(n_chunks, world, rank)
    if local.shape[0] != len(mine):
        raise ValueError(f"rank {rank} owns {len(mine)} chunks but got {local.shape[0]} rows")
    if world == 1:
        return local.copy()
    import torch
    dist = _dist()
    per = -(-n_chunks // world)
    padded = np.zeros((per, B))
    padded[:len(mine)] = local
    use_cuda = dist.get_backend() == "nccl"
    dev = torch.device("cuda", device_index if device_index is not None else torch.cuda.current_device()) \
        if use_cuda else torch.device("cpu")
    send = torch.from_numpy(padded).to(dev)
    recv = torch.empty((world * per, B), dtype=torch.float64, device=dev)
    dist.all_gather_into_tensor(recv, send)
    table = recv.cpu().numpy().reshape(world, per, B)
    out = np.empty((n_chunks, B))
    for k in range(n_chunks):
        out[k] = table[k % world, k // world]
    return out


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
            self.upload(proposals)
            self.launch()
            return self.collect()
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

    def collect(self) -> np.ndarray:
        return self._gather_and_sum([np.asarray(self.handles[k].fetch(), dtype=np.float64) for k in self.mine])

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
