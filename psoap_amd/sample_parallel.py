"""Multi-chunk, multi-chain sampling driver on the device lnprob(p) boundary.

What /root/reference/psoap/sample_parallel.py does with one forked process per chunk, a pipe protocol
(:206-255, :258-278) and one proposal per iteration (:371-390, :434-438), restated for GPUs:

* every chunk of ``chunks.dat`` is loaded, masked (:66-72) and kept resident on a GPU by a
  ``ChunkWorker`` (orbit solve + Doppler shift + GP likelihood on the device);
* with G processes (one per GPU, ``torch.distributed``), rank r owns chunks ``k = r (mod G)``; the
  per-(chunk, chain) log-likelihoods are exchanged by one small all_gather (RCCL) and summed in chunk
  order on every rank (:382-390), so all ranks see bit-identical posteriors and take identical
  accept/reject decisions without broadcasting proposals (same seeds everywhere);
* B independent Metropolis-Hastings chains advance in lock-step (``MultiChainMHSampler``), chain b
  writing what reference run ``run_index + b`` would: ``<outdir>/run{NN}/flatchain.npy`` and
  ``lnprob.npy`` (:440-443) plus a copy of the configuration (:52).

    python -m psoap_amd.sample_parallel 0 --chains 32 [--seed 1]
    python -m torch.distributed.run --nproc-per-node 8 -m psoap_amd.sample_parallel 0 --chains 32
"""
from __future__ import annotations

import argparse
import json
import os
import shutil

import numpy as np

from . import data as pdata
from . import priors as ppriors
from . import utils
from .ensemble import _NoLock, gather_chunk_lnprobs, owned_chunks, sum_over_chunks
from .samplers import MultiChainMHSampler


def load_config(path="config.yaml"):
    """config.yaml keys: model, chunk_file, mask_file, epoch_limit, soften, parameters, jumps, fix_params,
    samples, opt_jump, outdir (/root/reference/psoap/data/config.SB2.yaml; loaded at sample_parallel.py:11-14)."""
    import yaml
    try:
        with open(path) as f:
            return yaml.safe_load(f)
    except FileNotFoundError:
        print("You need to copy a config.yaml file to this directory, and then edit the values to your particular case.")
        raise


def load_chunks(config, prefix=""):
    """Open and mask every chunk of ``config['chunk_file']`` (sample_parallel.py:58-72)."""
    chunks = []
    for order, wl0, wl1 in pdata.read_chunk_table(config["chunk_file"]):
        ch = pdata.Chunk.open(order, wl0, wl1, limit=config["epoch_limit"], prefix=prefix)
        ch.apply_mask()
        chunks.append(ch)
    return chunks


def proposal_covariance(config, model, dim):
    """``opt_jump`` covariance if the file loads, else the diagonal of squared hand-specified jumps (:425-431)."""
    try:
        cov = np.load(config["opt_jump"])
        print("using optimal jumps")
    except Exception:
        print("using hand-specified jumps")
        cov = utils.convert_dict(model, config["fix_params"], **config["jumps"]) ** 2 * np.eye(dim)
    return cov


class Posterior:
    """``lnprob_batch(P)`` = prior + sum over chunks of the per-chunk GP log-likelihood (:371-390).

    ``chunks``: masked ``Chunk``-like objects (``lwl, fl, sigma, epoch_index, date1D``), ALL chunks on
    every rank; only the owned ones are uploaded.  ``make_worker`` is injectable for CPU tests.
    """

    def __init__(self, model, chunks, fix_params=(), parameters=None, soften=1.0, max_batch=1, world=1, rank=0,
                 device_index=None, prior=None, make_worker=None):
        self.model = model
        self.fix_params = list(fix_params)
        self.parameters = dict(parameters or {})
        self.n_chunks = len(chunks)
        self.world, self.rank, self.device_index = int(world), int(rank), device_index
        self.max_batch = int(max_batch)
        self.mine = owned_chunks(self.n_chunks, self.world, self.rank)
        if self.world > 1 and not self.mine:
            raise ValueError("every rank must own at least one chunk (n_chunks >= world)")
        self.prior = prior if prior is not None else ppriors.make_prior(model, self.fix_params, **self.parameters)
        if make_worker is None:
            from .lnprob import ChunkWorker

            def make_worker(ch):
                return ChunkWorker(model, ch.lwl, ch.fl, ch.sigma, ch.epoch_index, ch.date1D,
                                   fix_params=self.fix_params, defaults=self.parameters, max_batch=self.max_batch,
                                   device=device_index, soften=soften)
        self.workers = {k: make_worker(chunks[k]) for k in self.mine}
        self.device_lock = _NoLock()      # ensemble.SharedDeviceLock in dry runs with several ranks on one GPU
        # several chunks on this GPU: ONE launch of the persistent kernel over all of them per evaluation
        self.group = None
        if len(self.mine) > 1 and all(hasattr(w, "upload_proposals") and hasattr(w, "handle") for w in self.workers.values()):
            from .chunk import ChunkGroup
            self.group = ChunkGroup([self.workers[k].handle for k in self.mine])

    def close(self):
        if self.group is not None:
            self.group.close()
        for w in self.workers.values():
            if hasattr(w, "close"):
                w.close()

    def lnprob_batch(self, P):
        P = np.atleast_2d(np.asarray(P, dtype=np.float64))
        B = P.shape[0]
        lnprior = np.asarray(self.prior(P), dtype=np.float64)
        ok = np.isfinite(lnprior)                 # -inf prior: the reference never evaluates those (:373-375)
        out = np.full(B, -np.inf)
        # every rank sees the same P, hence the same `ok`: the collective below stays matched
        if ok.any():
            # The device batch keeps its size: a prior-rejected slot is evaluated on a stand-in (the first
            # admissible proposal) and overwritten with -inf afterwards.  A batch whose size followed the
            # number of rejections would rebuild and re-upload the task list on every change, and -- the
            # split factors depend on the batch size -- give the same proposal different last bits
            # depending on how many OTHER chains were rejected.
            Pev = P.copy()
            Pev[~ok] = P[ok][0]
            block = np.empty((len(self.mine), B))
            with self.device_lock:
                for s in range(0, B, self.max_batch):
                    piece = Pev[s:s + self.max_batch]
                    if self.group is not None:
                        for k in self.mine:
                            self.workers[k].upload_proposals(piece)
                        self.group.eval()
                        for i, k in enumerate(self.mine):
                            block[i, s:s + self.max_batch] = self.workers[k].handle.fetch()
                    else:
                        for i, k in enumerate(self.mine):
                            block[i, s:s + self.max_batch] = self.workers[k].lnprob_batch(piece)
            table = gather_chunk_lnprobs(block, self.n_chunks, self.world, self.rank, self.device_index)
            out[ok] = (sum_over_chunks(table) + lnprior)[ok]
        return out

    def lnprob(self, p):
        return float(self.lnprob_batch(np.atleast_2d(p))[0])

    # -- streamed form: the sampler's iterations through ONE resident launch per GPU (samplers.sample_streamed) ----------
    def can_stream(self) -> bool:
        """One chunk on this rank, and a worker that has the stream entry points: two resident launches cannot share a
        device (each takes every compute unit), so a rank with several chunks keeps the launch-per-step group path."""
        return len(self.mine) == 1 and all(hasattr(w, "stream_submit") for w in self.workers.values())

    def stream_open(self, scheme: int = -1):
        if not self.can_stream():
            raise RuntimeError("Posterior.stream_open: needs exactly one chunk on this rank (see can_stream)")
        self.workers[self.mine[0]].stream_open(self.max_batch, scheme)
        self._streaming = True

    def stream_submit(self, P, group: int = 0):
        """proposals (n, dim) of one sub-ensemble -> token; every rank submits the same rows (same seeds everywhere)"""
        P = np.atleast_2d(np.asarray(P, dtype=np.float64))
        lnprior = np.asarray(self.prior(P), dtype=np.float64)
        ok = np.isfinite(lnprior)
        tickets = None
        if ok.any():
            Pev = P.copy()
            Pev[~ok] = P[ok][0]          # (a prior-rejected slot is evaluated on a stand-in: lnprob_batch explains why)
            tickets = self.workers[self.mine[0]].stream_submit(Pev)
        return (tickets, lnprior, ok)

    def stream_fetch(self, token) -> np.ndarray:
        """lnprob of the rows of ``token``: the chunk's result, ONE gather over the ranks, fixed-order sum, prior"""
        tickets, lnprior, ok = token
        out = np.full(lnprior.shape[0], -np.inf)
        if tickets is not None:
            block = np.asarray(self.workers[self.mine[0]].stream_fetch(tickets), dtype=np.float64)[None, :]
            # (on the host: a device collective would wait for the resident launch to leave -- ensemble.host_group)
            table = gather_chunk_lnprobs(block, self.n_chunks, self.world, self.rank, self.device_index, on_host=True)
            out[ok] = (sum_over_chunks(table) + lnprior)[ok]
        return out

    def stream_close(self):
        if getattr(self, "_streaming", False):
            self.workers[self.mine[0]].stream_close()
            self._streaming = False


# Round 6: the automatic rule never streams.  The resident launch ties with one launch per step since the batch kernels'
# round-5 gains (N = 6000: 0.99-1.005 x, N = 8192: 1.002 x, 1-7 % slower below N = 5000: profiles/r5_stream_table.jsonl,
# r6_*), cannot run beside a device collective, and holds the device while it has results outstanding: it has to be asked
# for (`--stream on`, `run(stream=True)`).  bench.py measures it beside the headline in every default run (`stream_beside`).
STREAM_MIN_N = 5000            # (where a forced stream is at least level: below, the launch-per-step path is 1-7 % ahead)
STREAM_MIN_CHAINS = 16


def want_stream(post, chunks, n_chains) -> bool:
    """the automatic rule of ``run(stream=None)``: False (round 6; see above).  ``PSOAP_STREAM_AUTO=1`` brings round 5's rule
    back for experiments: ONE rank with one chunk, an even number of >= 16 chains, chunks of >= 5000 pixels, one process on
    the GPU."""
    if os.environ.get("PSOAP_STREAM_AUTO", "0") != "1":
        return False
    if not post.can_stream() or n_chains < STREAM_MIN_CHAINS or n_chains % 2:
        return False
    if post.world > 1:
        return False
    if not isinstance(post.device_lock, _NoLock):
        return False
    return min(int(np.asarray(ch.fl).shape[0]) for ch in chunks) >= STREAM_MIN_N


def run(config, chunks, run_index=0, n_chains=1, seed=None, world=1, rank=0, device_index=None, iterations=None,
        config_path=None, make_worker=None, prior=None, verbose=True, overwrite=False, device_lock=None, stream=None):
    """Sample ``n_chains`` chains for ``config['samples']`` iterations; returns the sampler.

    Chain b uses ``RandomState(seed + b)`` when ``seed`` is given (fresh entropy otherwise -- then every
    rank must be handed the same ``seed``: pass one explicitly under torch.distributed).
    Chain b is written to ``run{run_index + b}``; existing output directories are only replaced with
    ``overwrite=True`` (the reference replaces its single ``run{idx}``, sample_parallel.py:24-30 -- with
    ``n_chains`` directories per invocation a silent replace would wipe earlier runs).
    """
    model = config["model"]
    pars = config["parameters"]
    fix = config["fix_params"]
    dim = len(utils.registered_params[model]) - len(fix)
    p0 = utils.convert_dict(model, fix, **pars)
    if world > 1 and seed is None:
        raise ValueError("multi-rank sampling needs an explicit seed so that all ranks draw the same proposals")
    routdirs = [os.path.join(config["outdir"], "run{:0>2}".format(run_index + b)) + "/" for b in range(n_chains)]
    # Only rank 0 writes, so only rank 0 looks -- and every rank raises or none does: a rank that raised alone
    # (output directory on a non-shared file system, or created between the ranks' checks) would leave the others
    # hanging in the first gather.
    taken = [d for d in routdirs if os.path.exists(d)] if rank == 0 else []
    if world > 1:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            box = [taken]
            dist.broadcast_object_list(box, src=0)
            taken = box[0]
    if taken and not overwrite:
        raise FileExistsError("output directories exist (pass overwrite=True / --overwrite to replace them): "
                              + ", ".join(taken))
    post = Posterior(model, chunks, fix, pars, soften=config.get("soften", 1.0), max_batch=n_chains, world=world,
                     rank=rank, device_index=device_index, prior=prior, make_worker=make_worker)
    if device_lock is not None:
        post.device_lock = device_lock
    try:
        if verbose and rank == 0:
            print("Trying first evaluation")
        lnp0 = post.lnprob(p0)
        if lnp0 == -np.inf:
            raise RuntimeError("Starting position for Markov Chain evaluates to -np.inf")     # :405-419
        if verbose and rank == 0:
            print("Starting position good. lnp: {}".format(lnp0))
        cov = proposal_covariance(config, model, dim)
        seeds = [None if seed is None else seed + b for b in range(n_chains)]
        sampler = MultiChainMHSampler(cov, dim, post.lnprob_batch, n_chains, seeds)
        n_iter = int(config["samples"] if iterations is None else iterations)
        # stream: None = automatic (want_stream), True / False force it.  Streamed, the chains' iterations go through one
        # resident launch per GPU in two halves (samplers.sample_streamed: the loop of sample_parallel.py:434-438 with the
        # gather of :378-387 per half).  The two modes draw the same proposals and apply the same accept rule, so they give
        # the same chains GIVEN THE SAME lnprob values -- and a stream lane sums a matrix in another order than a batch
        # launch does: the values agree to ~1e-14 relative, not bit for bit, so a fixed-seed run can part ways with the
        # other mode's at the first accept decision that falls inside that margin.  Which mode a run used is recorded in its
        # output directory (evaluation.json); PSOAP_FIXED_PLAN=1 makes the batch path sum like a lane (bit-identical modes).
        use_stream = want_stream(post, chunks, n_chains) if stream is None else bool(stream)
        sampler.streamed = use_stream
        if use_stream:
            post.stream_open()
            # (the starting value through the stream as well: a lane's result is bit-identical whatever else is in flight,
            # the batch path's B = 1 plan sums in another order)
            steps = sampler.sample_streamed(p0, post.stream_submit, post.stream_fetch, groups=2, iterations=n_iter)
        else:
            steps = sampler.sample(p0, lnprob0=lnp0, iterations=n_iter)
        for i, _ in enumerate(steps):
            if verbose and rank == 0 and (i + 1) % 20 == 0:
                print("Iteration", i + 1)
    finally:
        # (a failing stream_close must neither hide the exception that brought us here nor skip post.close())
        try:
            post.stream_close()
        finally:
            post.close()
    if rank == 0:
        if verbose:
            print("Acceptance fraction", sampler.acceptance_fraction)
        for b, routdir in enumerate(routdirs):
            if os.path.exists(routdir):
                shutil.rmtree(routdir)
            os.makedirs(routdir)
            if config_path is not None and os.path.exists(config_path):
                shutil.copy(config_path, routdir + "config.yaml")
            with open(routdir + "evaluation.json", "w") as fh:
                json.dump({"stream": bool(use_stream), "world": int(world), "chains": int(n_chains),
                           "fixed_plan": os.environ.get("PSOAP_FIXED_PLAN", "0") == "1"}, fh)
            np.save(routdir + "lnprob.npy", sampler.lnprobability[b])
            np.save(routdir + "flatchain.npy", sampler.chain[b])
    return sampler


def main(argv=None):
    parser = argparse.ArgumentParser(description="Sample the distribution across multiple chunks.")
    parser.add_argument("run_index", type=int, nargs="?", default=0,
                        help="First output subdirectory; chain b is written to run{run_index + b}.")
    parser.add_argument("--chains", type=int, default=32, help="Independent MH chains evaluated per batched step.")
    parser.add_argument("--seed", type=int, default=None)
    parser.add_argument("--config", default="config.yaml")
    parser.add_argument("--prefix", default="", help="Directory prefix of the chunk files.")
    parser.add_argument("--overwrite", action="store_true", help="Replace existing run directories.")
    parser.add_argument("--stream", choices=("auto", "on", "off"), default="auto",
                        help="Iterations through one resident launch per GPU (auto = off since round 6: it ties with one launch per step).")
    args = parser.parse_args(argv)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    seed = args.seed
    if world > 1:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local)
        dist.init_process_group("nccl")
        if seed is None:
            seed = 0
    config = load_config(args.config)
    chunks = load_chunks(config, args.prefix)
    if rank == 0:
        print("Sampling {} chunks on {} GPU(s), {} chains per step.".format(len(chunks), world, args.chains))
    prior = ppriors.load_user_prior(".")
    if rank == 0:
        print("Loaded user defined prior." if prior is not None else "Using default prior.")
    try:
        run(config, chunks, args.run_index, args.chains, seed, world, rank, local if world > 1 else None,
            config_path=args.config, prior=prior, overwrite=args.overwrite,
            stream={"auto": None, "on": True, "off": False}[args.stream])
    finally:
        if world > 1:
            import torch.distributed as dist
            dist.destroy_process_group()


if __name__ == "__main__":
    main()
