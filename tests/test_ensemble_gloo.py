"""world_size-2 gloo test of the chunk-sharded ensemble (CPU).  The per-chunk evaluator is
the CPU oracle here -- the test covers partitioning, the gather and the fixed-order sum."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    """Kept as the third argument of the rank functions; the rendezvous itself goes through a file store in the test's
    temporary directory (no port to probe and lose between probe and bind)."""
    return 0


def _worker(rank, world, port, n_chunks, outdir):
    os.environ["GLOO_SOCKET_IFNAME"] = "lo"
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch.distributed as dist
    import oracle
    from psoap_amd import synthetic as syn
    from psoap_amd.ensemble import EnsembleEvaluator
    dist.init_process_group("gloo", init_method="file://" + os.path.join(outdir, "rendezvous"), rank=rank, world_size=world)
    chunks = [syn.make_chunk(2, 3, 30, seed=50 + k) for k in range(n_chunks)]
    gps = syn.make_walkers(2, 5, seed=9)

    def evaluate(k, proposals):
        return np.array([oracle.lnlike(chunks[k].lwls, chunks[k].fl, chunks[k].sigma, g) for g in proposals])

    ev = EnsembleEvaluator(n_chunks, evaluate, world, rank)
    total = ev.lnprob(gps)
    np.save(os.path.join(outdir, f"rank{rank}.npy"), total)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_chunks", [2, 3, 8])
def test_two_ranks_match_single_process(tmp_path, n_chunks, oracle):
    import torch.multiprocessing as mp
    from psoap_amd import synthetic as syn
    from psoap_amd.ensemble import EnsembleEvaluator
    port = _free_port()
    mp.spawn(_worker, args=(2, port, n_chunks, str(tmp_path)), nprocs=2, join=True)
    r0 = np.load(tmp_path / "rank0.npy")
    r1 = np.load(tmp_path / "rank1.npy")
    chunks = [syn.make_chunk(2, 3, 30, seed=50 + k) for k in range(n_chunks)]
    gps = syn.make_walkers(2, 5, seed=9)

    def evaluate(k, proposals):
        return np.array([oracle.lnlike(chunks[k].lwls, chunks[k].fl, chunks[k].sigma, g) for g in proposals])

    single = EnsembleEvaluator(n_chunks, evaluate).lnprob(gps)
    # bit-identical on every rank and for every world size (fixed-order sum)
    assert np.array_equal(r0, r1)
    assert np.array_equal(r0, single)
