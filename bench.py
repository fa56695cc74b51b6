#!/usr/bin/env python3
"""Headline benchmark: GP lnprob evaluations per second, SB2 chunk at N = 6000.

One "step" = one ensemble step of the hot path on every rank: the rank's SB2 chunk (20 epochs x 300 px,
N = 6000; BASELINE.json configs[2]/[3]) is evaluated for 32 walkers (fill -> +sigma^2 -> Cholesky -> solve ->
logdet -> lnprob), the proposals (32 walkers x 2 components x 6000 ln-wavelengths = 3 MB) come from host
memory over PCIe and the 32 lnprobs go back to it, and the per-(walker, chunk) lnprobs are gathered over
RCCL and summed in fixed chunk order (the gather-and-sum of psoap/sample_parallel.py:378-387).
So `value` INCLUDES the per-step H2D of c*N doubles per proposal and the D2H of the results (BASELINE.md
section 4).

Default `--mode dag` (rounds 1-3, and again from round 6 on): one launch of the persistent kernel per step, the next
step's proposals uploaded under it.  `--mode stream` (round 4) runs the K timed steps through ONE resident launch
(psoap_stream_*: include/psoap_gp.h), the ensemble as two half-ensembles in flight -- a half is fetched and its successor
submitted while the other half keeps the device busy (the back-to-back iterations of psoap/sample_parallel.py:434-438); the
timed region then starts with nothing in flight and no launch resident and ends when the last result is back and the
launch has left.  Whichever mode is `value`, the other is measured beside it in the same run (`stream_beside` /
`launch_per_step`): since round 5 the two tie (N = 6000: 0.99-1.005 x), so the simple one is the default.
Chunks are independent, so per-GPU work is fixed as N grows ("weak"); at 8 GPUs a step is exactly
the 32-walker x 8-chunk ensemble of configs[3].  The same run also times configs[3] AS NAMED -- the fixed
8-chunk x 32-walker ensemble, chunk k on rank k mod G, 256 evaluations per step at every G -- and reports
it as `cfg4_strong` (strong scaling; never `value`).

    python bench.py                      # 1 GPU
    python bench.py --gpus 8             # spawns 8 ranks itself (torch.distributed.run, RCCL)
    python -m torch.distributed.run --nproc-per-node 8 ... bench.py --gpus 8     # or under a launcher
    python bench.py --gpus 2 --backend gloo    # dry run of the N>1 path on a 1-GPU box

Rank 0 prints ONE JSON line.  Every reported lnprob is checked against the committed golden values
of the reference (tests/golden/*.npz) and, at N = 1, against the CPU baseline timed beside it.
"""
import argparse
import glob
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from psoap_amd import synthetic as syn  # noqa: E402

PEAK_FP64_TFLOPS = 78.6   # MI355X spec, fp64 matrix == fp64 vector
N_WALKERS = 32
PARITY_RTOL = 1e-10       # |dlnp| <= 1e-10 max(1, |lnp|), the contract of SURVEY.md section 8(c)


def flops_eval(N: int) -> float:
    """Algorithmic flops of one lnprob evaluation: N^3/3 + 2 N^2."""
    return N ** 3 / 3.0 + 2.0 * N ** 2


def flops_panel_update(N: int, nb: int = 128) -> float:
    """Algorithmic share of N^3/3 done by the left-looking panel update: every upper-triangle
    element (i, j >= i) receives one multiply-add from each of the 128*floor(i/128) finished rows."""
    i = np.arange(N, dtype=np.float64)
    k0 = nb * np.floor(i / nb)
    return float(np.sum(2.0 * k0 * (N - i)))


def close(a, b, rtol=PARITY_RTOL) -> bool:
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return bool(np.all(np.abs(a - b) <= rtol * np.maximum(1.0, np.abs(b))))


def require(ok: bool, what: str):
    if not ok:
        raise SystemExit(f"bench.py: PARITY FAILURE: {what}")


def golden(name: str):
    return np.load(os.path.join(ROOT, "tests", "golden", name), allow_pickle=False)


_CPU_CHILD = r"""
import json, os, sys, time
import numpy as np
root, chunk_index, threads, n_evals = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "oracle"))
from threadpoolctl import threadpool_limits
import oracle
from psoap_amd import synthetic as syn
ch = syn.make_config_chunk(4, chunk_index)
gp = syn.GP_BASE[2]
V11 = np.empty((ch.N, ch.N))
with threadpool_limits(limits=threads, user_api="blas"):
    oracle.lnlike(ch.lwls, ch.fl, ch.sigma, gp, V11=V11)          # warm-up
    print("READY", flush=True)
    sys.stdin.readline()                                           # all workers start their timed evaluations together
    ts = []
    for _ in range(n_evals):
        t0 = time.perf_counter()
        val = oracle.lnlike(ch.lwls, ch.fl, ch.sigma, gp, V11=V11)
        ts.append(time.perf_counter() - t0)
print(json.dumps({"chunk": chunk_index, "s_per_eval": float(np.median(ts)), "wall": float(sum(ts)), "lnprob": float(val)}), flush=True)
"""


def cpu_baseline_cfg4(n_evals: int = 3, also_threads: int | None = None):
    """The process model of SURVEY.md section 8(d) (threads = cores / 8 per worker) and, when the single-process sweep
    found a smaller thread count to be faster, that count as well; the faster of the two is the reported baseline."""
    cores = os.cpu_count() or 8
    runs = [_cpu_cfg4_run(max(1, cores // 8), n_evals)]
    if also_threads and 0 < also_threads < cores // 8:
        runs.append(_cpu_cfg4_run(also_threads, n_evals))
    best = max(runs, key=lambda r: r["value"])
    best = dict(best)
    best["runs"] = [{"threads_per_process": r["threads_per_process"], "evals_per_s": r["value"]} for r in runs]
    return best


def _cpu_cfg4_run(threads: int, n_evals: int):
    """BASELINE configs[3] in the reference's own process model (psoap/sample_parallel.py:258-278): one worker
    process per chunk, all eight evaluating at once, BLAS threads = cores / 8 each.
    Children are separate programs (nothing of this process's GPU state in them); they warm up, then start their
    timed evaluations together."""
    cores = os.cpu_count() or 8
    procs = [subprocess.Popen([sys.executable, "-c", _CPU_CHILD, ROOT, str(k), str(threads), str(n_evals)],
                              stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True,
                              env=dict(os.environ, OMP_NUM_THREADS=str(threads), OPENBLAS_NUM_THREADS=str(threads)))
             for k in range(8)]
    try:
        for p in procs:
            line = p.stdout.readline()
            if "READY" not in line:
                raise RuntimeError(f"cpu_baseline_cfg4: worker did not come up: {line!r}")
        t0 = time.perf_counter()
        for p in procs:
            p.stdin.write("go\n")
            p.stdin.flush()
        recs = [json.loads(p.stdout.readline()) for p in procs]
        wall = time.perf_counter() - t0
    finally:
        for p in procs:
            try:
                p.stdin.close()
            except Exception:
                pass
            p.wait(timeout=60)
    for r in recs:
        require(bool(np.isfinite(r["lnprob"])), f"cpu_baseline_cfg4: chunk {r['chunk']} lnprob not finite")
    return {"value": 8 * n_evals / wall, "unit": "evals/s", "processes": 8, "threads_per_process": threads,
            "cores": 8 * threads, "kind": "port",
            "sample": f"8 worker processes (one per configs[3] chunk, N=6000) x {n_evals} evals each after 1 warm-up, started "
                      f"together; {wall:.2f} s wall; per-process median {np.median([r['s_per_eval'] for r in recs]):.3f} s/eval; "
                      f"host has {cores} cores",
            "s_per_eval_per_process": [round(r["s_per_eval"], 4) for r in recs]}


def cpu_baseline(chunk, sweep=(8, 16, 32, 64, 128), n_sweep: int = 3, n_best: int = 6):
    """The CPU oracle (C fill + SciPy cho_factor/cho_solve, i.e. the reference's own library calls) timed on this
    box's host cores: a sweep over BLAS thread counts, the BEST of which is the reported baseline (so that the
    stated baseline is the reference's best on this host).  Reported, never the target."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle
    oracle.build()
    from threadpoolctl import threadpool_info, threadpool_limits
    cores = os.cpu_count() or 1
    max_threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    counts = sorted({t for t in sweep if t <= max_threads} | {max_threads})
    gp = syn.GP_BASE[chunk.n_components]
    V11 = np.empty((chunk.N, chunk.N))

    def timed(threads, n):
        with threadpool_limits(limits=threads, user_api="blas"):
            oracle.lnlike(chunk.lwls, chunk.fl, chunk.sigma, gp, V11=V11)     # warm-up at this thread count
            ts = []
            for _ in range(n):
                t0 = time.perf_counter()
                v = oracle.lnlike(chunk.lwls, chunk.fl, chunk.sigma, gp, V11=V11)
                ts.append(time.perf_counter() - t0)
        return float(np.median(ts)), float(v)

    table = {}
    for t in counts:
        table[t], val = timed(t, n_sweep)
    best = min(table, key=table.get)
    med, val = timed(best, n_best)
    med = min(med, table[best])
    # the Cython-equivalent fill alone (matrix_functions.pyx:99-146), for the fill drop-in comparison
    tf = []
    for _ in range(3):
        t0 = time.perf_counter()
        oracle.fill_V11_f_g(V11, chunk.lwls[0], chunk.lwls[1], *gp)
        tf.append(time.perf_counter() - t0)
    return {"value": 1.0 / med, "unit": "evals/s", "cores": int(best), "kind": "port",
            "sample": f"lnlike_f_g on the same N={chunk.N} SB2 chunk: BLAS thread sweep "
                      + ", ".join(f"{t}: {1.0 / table[t]:.2f}/s" for t in counts)
                      + f" ({n_sweep} evals each after 1 warm-up, median), then {n_best} evals at the best count ({best} threads): "
                      f"{med:.3f} s/eval; C fill (1 thread) + SciPy/OpenBLAS dpotrf/dpotrs; host has {cores} cores",
            "thread_sweep_evals_per_s": {str(t): 1.0 / table[t] for t in counts},
            "lnprob": float(val), "fill_ms": 1e3 * float(np.median(tf))}


def launch_ranks(args) -> int:
    """`python bench.py --gpus N` outside a launcher: start the N ranks as children (one process per
    GPU under torch.distributed.run) BEFORE this process touches the GPU, relay rank 0's JSON line."""
    # --standalone: the launcher binds its own rendezvous port on the loopback address (no probe-then-bind race)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           f"--nproc-per-node={args.gpus}", os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if line is not None:
        print(line, flush=True)
    return proc.returncode if proc.returncode != 0 else (0 if line is not None else 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--walkers", type=int, default=N_WALKERS)
    ap.add_argument("--groups", type=int, default=2, help="concurrent stream groups per batch (staged mode)")
    ap.add_argument("--mode", default=None, choices=["stream", "dag", "staged"],
                    help="stream: the timed steps through ONE resident launch (two half-ensembles in flight); dag / staged: "
                         "one launch (three per panel) per step.  Default: stream on one GPU; dag on several -- the RCCL "
                         "gather of a step then runs between two launches (beside a resident launch a device collective "
                         "waits for the launch to leave: profiles/r5_gather_beside_stream.txt; --mode stream on several "
                         "ranks gathers every half-ensemble through the gloo side group on the host instead)")
    ap.add_argument("--stream-groups", type=int, default=2, help="sub-ensembles in flight in --mode stream")
    ap.add_argument("--allow-fallback", action="store_true",
                    help="print `value` even if the library was built from the fallback flag rung (psoap_amd/build.py)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-strong", action="store_true", help="skip the configs[3] strong-scaling leg (cfg4_strong)")
    ap.add_argument("--no-extras", action="store_true",
                    help="only the timed loop + roofline (profiling runs: keeps every k_chol_dag dispatch alike)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend (nccl = RCCL; gloo only for single-GPU dry runs of the N>1 path)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args))

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.mode is None:
        # Round 6: one launch per step IS the headline path on every number of GPUs.  The resident launch (round 4) ties
        # with it since the batch kernels' round-5 gains (N = 6000: 0.99-1.005 x, N = 8192: 1.002 x, below N = 5000 slower:
        # profiles/r5_stream_table.jsonl, r6_stream_table.jsonl), needs ~1,500 lines of protocol, and cannot run beside a
        # device collective; it stays an opt-in (--mode stream) and is measured BESIDE the headline in every default run
        # (`stream_beside`), so that the choice is re-made from data each round.
        args.mode = "dag"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    if args.backend == "gloo":       # dry run: several ranks may share one GPU
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    os.environ.setdefault("PSOAP_DEVICE", str(local_rank))
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend="gloo")

    from psoap_amd import build as _build
    from psoap_amd.chunk import ChunkHandle, StreamPipeline, microbench
    from psoap_amd.ensemble import SharedDeviceLock, _NoLock, gather_chunk_lnprobs, sum_over_chunks

    # Dry runs (--backend gloo: several ranks on one GPU): one rank at a time on the device -- its time-slicing of several
    # processes' persistent kernels is not covered by the kernels' hand-off protocol (psoap_amd/ensemble.py).  Around
    # everything from a launch to the fetch of its results, never around a collective.
    shared_gpu = world > 1 and args.backend == "gloo"
    gpu = SharedDeviceLock(local_rank) if shared_gpu else _NoLock()
    if shared_gpu:
        # (a dry run puts up to 8 ranks + whatever started them on ONE device: more process contexts than the device keeps
        # mapped, where the library would send every evaluation down the staged path and refuse streams.  The dry run exists
        # to exercise the persistent-kernel paths of the N > 1 layout: keep them -- a launch the scheduler disturbs is still
        # detected and run again -- and say so in the line.)
        os.environ.setdefault("PSOAP_SHARE_DAG_MAX", "64")

    # which binary runs: its hash, what it was built from and by, and the rung of the build's flag ladder
    library = _build.provenance()
    if library.get("fallback_rung") not in (0, None) and not args.allow_fallback:
        raise SystemExit(f"bench.py: libpsoap_gp.so was built from fallback rung {library['fallback_rung']} of the flag ladder "
                         "(no following scheme, 3.3-3.4 ms single evaluations); pass --allow-fallback to measure it anyway")

    # ---- workload.  One GPU: the configs[2] chunk (seed 3000).  N GPUs: rank r owns chunk r of the
    # configs[3] ensemble (seeds 4000 + r); walkers identical on every rank by seeding.
    cfg = 3 if world == 1 else 4
    chunk = syn.make_config_chunk(cfg, chunk_index=rank)
    c, N, B = chunk.n_components, chunk.N, args.walkers
    gps = syn.make_walkers(c, B, seed=1000 * cfg + 500)
    vels = syn.make_walker_velocities(chunk, B, seed=1000 * cfg + 501 + rank)
    lwls_a = syn.walker_lwls(chunk, vels)
    # a second proposal set (walkers rotated by one) so that consecutive steps upload different bytes
    lwls_b, gps_b = np.roll(lwls_a, 1, axis=0).copy(), np.roll(gps, 1, axis=0).copy()
    sets = [(lwls_a, gps), (lwls_b, gps_b)]

    h = ChunkHandle(chunk.fl, chunk.sigma, max_batch=B, device=local_rank)
    h.set_stream_groups(args.groups)
    batch_mode = "dag" if args.mode == "stream" else args.mode
    h.set_mode(batch_mode)

    collectives = {"n": 0}

    def gather(lnp):
        table = gather_chunk_lnprobs(lnp[None, :], world, world, rank, local_rank)   # (n_chunks, B)
        if world > 1:
            collectives["n"] += 1
        return table, sum_over_chunks(table)

    state = {"k": 0}

    def step():
        """eval(k) || upload(k+1), fetch(k), gather.  One H2D, one evaluation, one D2H per step."""
        with gpu:
            h.eval()
            state["k"] += 1
            h.upload(*sets[state["k"] & 1])
            lnp = h.fetch()
            if shared_gpu:
                h.sync()                   # (the upload too: nothing of this rank is left on the device)
        return gather(lnp)

    def fence_nohandle():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def fence():
        fence_nohandle()
        h.sync()

    def run_launch_per_step(n_warm, n_steps):
        """rounds 1-3: one launch per step, the next step's proposals uploaded under it"""
        state["k"] = 0
        with gpu:
            h.upload(*sets[0])
            h.sync()
        for _ in range(n_warm):
            table, total = step()
        fence()
        t0 = time.perf_counter()
        for _ in range(n_steps):
            table, total = step()
        fence()
        return time.perf_counter() - t0, table, total, (state["k"] - 1) & 1

    stream_info = None

    def measure_stream(n_warm, n_steps):
        """n_steps ensemble steps through ONE resident launch, from nothing in flight to nothing in flight and no launch
        resident: (seconds, table, total, last proposal set, the launch's own account)"""
        if B % args.stream_groups:
            raise SystemExit("--walkers must be a multiple of --stream-groups")
        pipe = StreamPipeline(h, c, B, args.stream_groups)
        half = {"table": np.zeros((world, B)), "total": np.zeros(B)}

        def gather_half(g, rows, lnp_rows):
            """a sub-ensemble's lnprobs over the ranks BEFORE its successor is submitted: a sampler's next proposals
            depend on the chunk sum (sample_parallel.py:378-390)"""
            # (on the host -- the gloo side group of ensemble.host_group -- while the launch is resident: a device collective
            # would wait for it to leave, 2.4 x the step time: profiles/r5_gather_beside_stream.txt)
            t = gather_chunk_lnprobs(lnp_rows[None, :], world, world, rank, local_rank, on_host=True)
            if world > 1:
                collectives["host"] = collectives.get("host", 0) + 1
            half["table"][:, rows] = t
            half["total"][rows] = sum_over_chunks(t)

        def run_stream_shared(n):
            """the dry-run form: a step's sub-ensembles go through the stream together and the resident launch leaves
            before the gather -- the same entry points, no overlap (the device belongs to one rank at a time)"""
            for k in range(n):
                with gpu:
                    pipe.start(*sets[k & 1], stagger=0.0)
                    lnp = pipe.drain()
                    h.stream_pause()
                table, total = gather(lnp)
            return table, total, (n - 1) & 1

        def run_stream(n):
            if shared_gpu:
                return run_stream_shared(n)
            pipe.start(*sets[0])
            for k in range(1, n):
                # per sub-ensemble: results of step k - 1 -> gather over the ranks -> proposals of step k submitted
                pipe.step(*sets[k & 1], between=gather_half)
            pipe.drain(between=gather_half)
            h.stream_pause()                           # the resident launch leaves: the device is free again
            return half["table"].copy(), half["total"].copy(), (n - 1) & 1

        if n_warm > 0:
            # the warm-up steps also give the period the timed region's start-up stagger is set from (they start in
            # lock-step themselves: the pipeline knows no period yet)
            tw = time.perf_counter()
            run_stream(n_warm)
            pipe.period = (time.perf_counter() - tw) / n_warm
        fence_nohandle()
        t0 = time.perf_counter()
        table, total, last_set = run_stream(n_steps)
        fence_nohandle()
        dt = time.perf_counter() - t0
        info = dict(h.stream_last_launch(), **h.stream_stats())
        with gpu:
            pipe.close()
        return dt, table, total, last_set, info

    if args.mode == "stream":
        dt, table, total, last_set, stream_info = measure_stream(args.warmup, args.steps)
    else:
        dt, table, total, last_set = run_launch_per_step(args.warmup, args.steps)
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    evals = world * B * args.steps
    value = evals / dt
    if last_set == 1:                        # back to walker order
        table, total = np.roll(table, -1, axis=1), np.roll(total, -1)

    # ---- parity gate on what the timed loop produced (reference goldens, no tolerance games)
    parity = {}
    if world == 1:
        g1 = golden("golden_v1.npz")
        want0 = float(g1["lnlike_vals"][list(g1["lnlike_names"]).index("cfg3_sb2_n6000")])
        require(close(total[0], want0), f"walker 0 {total[0]!r} vs reference golden {want0!r}")
        parity["golden_cfg3_walker0"] = want0
        # the reference's 4-walker batch on this chunk is a prefix of the ensemble (same seeds, drawn in order)
        nw = min(B, len(g1["walkers_cfg3"]))
        require(close(total[:nw], g1["walkers_cfg3"][:nw]), f"walkers {total[:nw]} vs reference goldens")
        parity["golden_cfg3_walkers"] = nw
    else:
        gf = golden("golden_full_v1.npz")["cfg4_lnlike"]             # (8 chunks, 4 walkers)
        nk, nw = min(world, gf.shape[0]), min(B, gf.shape[1])
        require(close(table[:nk, :nw], gf[:nk, :nw]),
                f"gathered (chunk, walker) table differs from the reference goldens: {table[:nk, :nw]} vs {gf[:nk, :nw]}")
        parity["golden_cfg4_table"] = [nk, nw]

    # ---- the launch-per-step path beside the streamed headline (same run, same box; never `value` in stream mode)
    per_step = None
    if args.mode == "stream":
        dps, tps, totps, lsps = run_launch_per_step(min(args.warmup, 2), args.steps)
        if world > 1:
            tmax = torch.tensor([dps], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dps = float(tmax.item())
        if lsps == 1:
            totps = np.roll(totps, -1)
        # the same proposals through both paths: equal to the parity tolerance (the stream's task list is that of one
        # matrix, the batch's that of 32 in lock-step: another order of summation)
        if not close(totps, total):
            bad = np.where(np.abs(totps - total) > PARITY_RTOL * np.maximum(1.0, np.abs(total)))[0]
            raise SystemExit("bench.py: PARITY FAILURE: launch-per-step path vs streamed path on the same proposals: walkers "
                             f"{bad.tolist()}: launch-per-step {totps[bad].tolist()} streamed {total[bad].tolist()}; per-chunk "
                             f"tables of those walkers: launch-per-step {tps[:, bad].tolist() if lsps == 0 else np.roll(tps, -1, axis=1)[:, bad].tolist()} "
                             f"streamed {table[:, bad].tolist()} (rank {rank})")
        per_step = {"evals_per_s": evals / dps, "ms_per_step": 1e3 * dps / args.steps,
                    "what": "rounds 1-3: one launch of the persistent kernel per step, next step's proposals uploaded under it"}

    # ---- the resident launch beside the launch-per-step headline (same run, same box; never `value` in dag mode): what
    # keeps the default an informed one (verdict of round 5: the stream has to earn its place with >= 1.02 x)
    stream_beside = None
    if args.mode == "dag" and world == 1 and not shared_gpu and not args.no_extras:
        dss, tss, totss, lsss, info_s = measure_stream(min(args.warmup, 2), args.steps)
        if lsss == 1:
            totss = np.roll(totss, -1)
        require(close(totss, total), "the resident launch vs the launch-per-step headline on the same proposals")
        stream_beside = {"evals_per_s": evals / dss, "ms_per_step": 1e3 * dss / args.steps,
                         "ratio_to_value": (evals / dss) / value, "scheme": info_s.get("scheme"),
                         "launch_ms": info_s.get("ms"), "matrices": info_s.get("matrices"),
                         "what": f"the same {args.steps} steps through ONE resident launch, {args.stream_groups} sub-ensembles in flight (--mode stream)"}

    # ---- proposals-resident rate (the round-1 headline; never `value`): eval + fetch + gather only
    with gpu:
        h.upload(*sets[0])
        h.eval(); h.fetch()
    fence()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        with gpu:
            h.eval()
            lnp_r = h.fetch()
        gather(lnp_r)
    fence()
    dt_res = time.perf_counter() - t1
    if world > 1:
        tmax = torch.tensor([dt_res], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt_res = float(tmax.item())
    resident_value = evals / dt_res

    # ---- live per-kernel timing (HIP events on the launch stream) for the roofline object
    with gpu:
        h.set_profiling(True)
        h.eval()
        h.fetch()
        tm = h.timings()
        h.set_profiling(False)
    mode = "dag" if tm["dag"]["launches"] > 0 else "staged"
    if args.mode == "stream":
        # the dominant kernel is the RESIDENT launch of the timed region: HIP events around it on its stream
        # (psoap_stream_last_launch), algorithmic flops = the matrices it completed x F(N)
        mode = "stream"
        dom = {"ms": stream_info["ms"], "launches": 1, "flops": float("nan")}
        dom_name = ("k_chol_dag<%d, false, false, stream> (ONE resident launch over the %d timed steps: persistent tile DAG, "
                    "v_mfma_f64_16x16x4_f64)" % (c, args.steps))
        alg_per_launch = stream_info["matrices"] * flops_eval(N)
        require(shared_gpu or (stream_info["matrices"] == B * args.steps and stream_info["launches"] >= 1),
                f"the timed region's resident launch completed {stream_info['matrices']} matrices, expected {B * args.steps}")
        if shared_gpu:       # dry run: one launch per step -- the line's roofline object describes the last one
            alg_per_launch = stream_info["matrices"] * flops_eval(N)
    elif mode == "dag":
        # one persistent launch does the whole factorisation + solve of the batch:
        # algorithmic flops per launch = B x F(N)  
        dom, dom_name = tm["dag"], "k_chol_dag (persistent tile DAG, v_mfma_f64_16x16x4_f64)"
        alg_per_launch = B * flops_eval(N) / max(1, dom["launches"])
    else:
        dom, dom_name = tm["panel_update"], "k_panel_update (v_mfma_f64_16x16x4_f64)"
        alg_per_launch = B * flops_panel_update(N) / max(1, dom["launches"])
    avg_ms = dom["ms"] / max(1, dom["launches"])
    achieved = alg_per_launch / (avg_ms * 1e-3) / 1e12

    def allreduce_max(x):
        if world == 1:
            return float(x)
        t = torch.tensor([x], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    out = None
    extras = {}
    if world == 1 and not args.no_extras:
        extras = run_extras(args, h, chunk, gps, lwls_a, local_rank, batch_mode)
    with gpu:
        h.close()
    # the configs[3] curve: the fixed 8-chunk ensemble over G ranks, at every G (after the headline handle is gone: at
    # G = 1 the eight chunks' 256 matrices take 74 GB)
    strong = None if args.no_strong else strong_leg(args, world, rank, local_rank, fence_nohandle, allreduce_max, gpu)

    if rank == 0:
        mb = microbench(local_rank)
        traffic, traffic_src = None, None
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")), reverse=True):
            tj = json.load(open(path))
            w = tj.get("workload", {})
            if (w.get("N"), w.get("components"), w.get("walkers"), w.get("mode")) == (N, c, B, mode):
                if mode == "stream":
                    # a resident launch's traffic scales with the matrices it completes: the PMC passes measured it
                    # per evaluation (launches of 160 evaluations), this launch completed stream_info["matrices"]
                    traffic = tj["hbm_bytes_per_evaluation"] * stream_info["matrices"]
                else:
                    traffic = tj["hbm_bytes_per_launch"]
                traffic_src = os.path.relpath(path, ROOT)
                break
        out = {
            "metric": "GP lnprob evals/sec (SB2, N=6000)",
            "value": value, "unit": "evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            # ranks of the RCCL communicator the gathers of this run actually went through (0: none ran -- one GPU, or gloo)
            "rccl_ranks": dist.get_world_size() if (world > 1 and args.backend == "nccl" and collectives["n"] > 0) else 0,
            "collectives_completed": collectives["n"],
            "host_gathers_completed": collectives.get("host", 0),
            "backend": args.backend if world > 1 else None,
            "config": {"workload": f"SB2 chunk 20 epochs x 300 px (N={N}), {B} walkers per step per GPU, "
                                   f"one chunk per GPU (BASELINE.json configs[2]; configs[3] at 8 GPUs); " +
                                   (f"the {args.steps} timed steps through ONE resident launch, {args.stream_groups} sub-ensembles "
                                    f"of {B // args.stream_groups} walkers in flight (proposals pulled from pinned host memory, "
                                    f"{B} lnprobs per step written to it), every sub-ensemble gathered over the ranks before its successor is submitted"
                                    if args.mode == "stream" else
                                    f"timed step = H2D of next proposals || eval, D2H of {B} lnprobs, gather"),
                       "N": N, "components": c, "walkers": B, "chunks_per_gpu": 1, "mode": args.mode,
                       "stream_groups": args.stream_groups if args.mode == "stream" else args.groups,
                       "parallelism": f"chunk-sharded x{world}, RCCL all_gather of walker lnprobs"},
            "timing_boundary": ("pcie_inclusive (every proposal is pulled from pinned host memory by the resident launch, "
                                "every result written to it; the timed region starts and ends with nothing in flight and "
                                "no launch resident)" if args.mode == "stream" else
                                "pcie_inclusive (per-step H2D of B*c*N doubles double-buffered under the previous eval)"),
            "library": library,
            "launch_per_step": per_step,
            "stream": stream_info,
            "stream_beside": stream_beside,
            "resident_evals_per_s": resident_value,
            "inclusive_over_resident": value / resident_value,
            "roofline": {"bound": "mfma", "kernel": dom_name,
                         "achieved": achieved, "peak": PEAK_FP64_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP64_TFLOPS, "traffic": traffic, "traffic_unit": "bytes/launch",
                         "traffic_source": traffic_src,
                         "launches_per_step": (1.0 / args.steps) if args.mode == "stream" else dom["launches"],
                         "avg_launch_ms": avg_ms,
                         "units_per_launch": (f"{stream_info['matrices']} evaluations = {args.steps} steps x {B} walkers"
                                              if args.mode == "stream" else f"{B} evaluations"),
                         "algorithmic_flops_per_launch": alg_per_launch,
                         "executed_tflops": (dom["flops"] / (dom["ms"] * 1e-3) / 1e12
                                             if dom["ms"] > 0 and dom["flops"] == dom["flops"] else None),
                         "measured_peak": mb["mfma_f64_tflops"]},
            "roofline_eval": {"bound": "mfma", "achieved": value / world * flops_eval(N) / 1e12,
                              "peak": PEAK_FP64_TFLOPS, "unit": "TFLOP/s",
                              "frac": value / world * flops_eval(N) / 1e12 / PEAK_FP64_TFLOPS,
                              "flops_per_eval": flops_eval(N)},
            "kernel_ms_profiled_step": {k: round(tm[k]["ms"], 3) for k in
                                        ("fill", "panel_update", "potrf", "trsm", "misc", "dag")},
            "profiled_step_total_ms": tm["total_ms"],
            "lnprob_walker0": float(total[0]),
            "parity_checked": True, "parity": parity,
        }
        out.update(extras)
        if strong is not None:
            out["cfg4_strong"] = strong
            if world == 1:       # the names of rounds 1-2 for the same measurement
                out["cfg4_one_gpu_evals_per_s"] = strong["evals_per_s"]
                out["cfg4_one_gpu_tflops"] = strong["tflops_per_gpu"]
                out["cfg4_one_gpu_frac"] = strong["frac_of_peak_per_gpu"]
        if "roofline_fill" in extras:
            out["roofline_fill"]["measured_write_peak"] = mb["hbm_write_gbs"]
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(chunk)
            require(close(total[0], cb["lnprob"]), f"walker 0 {total[0]!r} vs CPU baseline {cb['lnprob']!r}")
            out["cpu_baseline"] = cb
            out["speedup_vs_cpu"] = value / cb["value"]
            if strong is not None:
                out["cpu_baseline_cfg4"] = cpu_baseline_cfg4(also_threads=cb["cores"])
                out["cfg4_speedup_vs_cpu"] = strong["evals_per_s"] / out["cpu_baseline_cfg4"]["value"]
            if "fill_dropin_ms" in out:
                out["fill_dropin_vs_cpu_fill"] = cb["fill_ms"] / out["fill_dropin_ms"]
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def strong_leg(args, world, rank, dev, fence, allreduce_max, gpu):
    """BASELINE configs[3] as named: the FIXED 8-chunk x 32-walker ensemble (seeds 4000..4007), chunk k on rank
    k mod G (psoap/sample_parallel.py:258-278 fans the chunks out, :378-387 gathers and sums), all chunks of a rank
    in ONE launch of the persistent kernel (ChunkGroup) -- 256 evaluations per step whatever G is: strong scaling.
    Same timing protocol as the headline (barrier + synchronize on both sides, max over ranks, next step's
    proposals uploaded under the evaluation); every (chunk, walker) value of the first walkers is checked against
    the reference's goldens on every rank."""
    from psoap_amd.ensemble import EnsembleEvaluator, SharedDeviceLock
    shared = isinstance(gpu, SharedDeviceLock)
    B = args.walkers
    n_chunks = 8
    chunks8 = [syn.make_config_chunk(4, k) for k in range(n_chunks)]
    gps4 = syn.make_walkers(2, B, seed=4500)
    with gpu:
        ev8 = EnsembleEvaluator.from_chunks(chunks8, max_batch=B, world=world, rank=rank, device_index=dev)
    ev8.device_lock = gpu
    props8 = {k: (syn.walker_lwls(chunks8[k], syn.make_walker_velocities(chunks8[k], B, seed=4501 + k)), gps4)
              for k in ev8.mine}
    N = chunks8[0].N
    tot8 = ev8.lnprob(props8)                # warm-up: plans, workspaces, the collective
    with gpu:
        ev8.upload(props8)
        for hk in ev8.handles.values():
            hk.sync()
    n4 = max(2, min(args.steps, 4))
    fence()
    t4 = time.perf_counter()
    for _ in range(n4):                      # same boundary as the headline: H2D of the next step under the evaluation
        with gpu:
            ev8.launch()
            ev8.upload(props8)
            local8 = ev8.fetch_local()
            if shared:                       # dry run: nothing of this rank may be left on the device outside the lock
                for hk in ev8.handles.values():
                    hk.sync()
        tot8 = ev8._gather_and_sum(local8)
    fence()
    dt4 = allreduce_max(time.perf_counter() - t4)
    rate = n4 * n_chunks * B / dt4
    gf = golden("golden_full_v1.npz")["cfg4_lnlike"]
    nw = min(B, gf.shape[1])
    require(close(ev8.table[:, :nw], gf[:, :nw]), "configs[3] strong leg: (chunk, walker) table vs reference goldens")
    want = np.zeros(nw)
    for k in range(n_chunks):
        want = want + gf[k, :nw]
    require(close(tot8[:nw], want), "configs[3] strong leg: walker sums vs reference goldens")
    stats = ev8.group.stats() if getattr(ev8, "group", None) is not None else None
    with gpu:
        ev8.close()
    return {"workload": f"BASELINE configs[3]: {n_chunks} SB2 chunks (N={N}) x {B} walkers = {n_chunks * B} evals per step, "
                        f"chunk k on rank k mod {world}, one launch per rank",
            "evals_per_s": rate, "ms_per_step": 1e3 * dt4 / n4, "steps": n4, "n_gpus": world, "scaling": "strong",
            "chunks_per_rank": [len([k for k in range(n_chunks) if k % world == r]) for r in range(world)],
            "tflops_per_gpu": rate / world * flops_eval(N) / 1e12,
            "frac_of_peak_per_gpu": rate / world * flops_eval(N) / 1e12 / PEAK_FP64_TFLOPS,
            "group_plan_builds": None if stats is None else stats["plan_builds"],
            "parity_checked": True, "parity_table": [n_chunks, nw]}


def run_extras(args, h, chunk, gps, lwls, dev, mode):
    """Side measurements of a single-GPU run (never `value`): the stand-alone fill kernel and its drop-in,
    the lnprob(p) and sampler boundaries, configs[3] on one GPU, configs[4]'s reconstruction."""
    from psoap_amd import matrix_functions
    from psoap_amd.chunk import ChunkHandle
    from psoap_amd.ensemble import EnsembleEvaluator
    from psoap_amd.lnprob import ChunkWorker
    from psoap_amd.samplers import MultiChainMHSampler
    c, N, B = chunk.n_components, chunk.N, args.walkers
    ex = {}

    # the HBM-bound stand-alone fill kernel (fill_V11_* drop-in, staged path, predict) by one
    # event-profiled staged step
    h.set_mode("staged")
    h.set_profiling(True)
    h.eval()
    h.fetch()
    fill = h.timings()["fill"]
    h.set_profiling(False)
    h.set_mode(mode)
    gbs = fill["bytes"] / (fill["ms"] * 1e-3) / 1e9 if fill["ms"] > 0 else None
    ex["roofline_fill"] = {"bound": "hbm", "kernel": f"k_fill_sym<{c}> (upper tiles)", "achieved": gbs, "peak": 8000.0,
                           "unit": "GB/s", "frac": gbs / 8000.0 if gbs else None}
    # the drop-in itself, end to end: fill_V11_f_g(mat, ...) into the caller's host matrix (288 MB over PCIe)
    mat = np.empty((N, N))
    matrix_functions.fill_V11_f_g(mat, chunk.lwls[0], chunk.lwls[1], *syn.GP_BASE[2])
    tfs = []
    for _ in range(3):
        t0 = time.perf_counter()
        matrix_functions.fill_V11_f_g(mat, chunk.lwls[0], chunk.lwls[1], *syn.GP_BASE[2])
        tfs.append(time.perf_counter() - t0)
    ex["fill_dropin_ms"] = 1e3 * float(np.median(tfs))
    require(bool(np.array_equal(mat, mat.T)), "fill_V11_f_g drop-in: matrix not symmetric")
    del mat

    # the drop-in call itself: what an unchanged Worker.lnprob issues per proposal (psoap/sample_parallel.py:193) -- ONE
    # evaluation through the reference's signature, host arrays in, a float out
    from psoap_amd import covariance as cov
    lnlike = {1: cov.lnlike_f, 2: cov.lnlike_f_g, 3: cov.lnlike_f_g_h}[c]
    call = (None, *[lwls[0][k] for k in range(c)], chunk.fl, chunk.sigma, *gps[0])
    v0 = lnlike(*call)
    require(close(v0, float(h.lnlike_batch(lwls[:1], gps[:1])[0])), "drop-in lnlike vs the batched path (one walker)")
    tds = []
    for _ in range(15):
        t0 = time.perf_counter()
        lnlike(*call)
        tds.append(time.perf_counter() - t0)
    t4 = []
    for _ in range(8):
        t0 = time.perf_counter()
        h.lnlike_batch(lwls[:4], gps[:4])
        t4.append(time.perf_counter() - t0)
    h.lnlike_batch(lwls, gps)                  # leave the headline batch in the slot
    tf1 = flops_eval(N) / float(np.median(tds)) / 1e12
    ex["dropin_eval"] = {"call": f"covariance.{lnlike.__name__}(V11, *lwls, fl, sigma, *p_GP), N={N}: upload + one evaluation + fetch",
                         "ms": 1e3 * float(np.median(tds)), "min_ms": 1e3 * float(min(tds)),
                         "tflops": tf1,
                         # the reference's own calling pattern (psoap/sample_parallel.py:193), host call to host result
                         "roofline": {"bound": "mfma", "achieved": tf1, "peak": PEAK_FP64_TFLOPS, "unit": "TFLOP/s",
                                      "frac": tf1 / PEAK_FP64_TFLOPS, "algorithmic_flops": flops_eval(N),
                                      "note": "per call, PCIe and Python included; one matrix: bound by the row-to-row chain"},
                         "batch4_ms": 1e3 * float(np.median(t4)), "lnprob": float(v0)}

    # the lnprob(p) boundary: orbital parameters in, lnprob out
    model = {1: "SB1", 2: "SB2", 3: "ST3"}[c]
    worker = ChunkWorker(model, chunk.lwl, chunk.fl, chunk.sigma, chunk.epoch_index, chunk.dates, max_batch=B,
                         device=dev)
    pfit = np.hstack([syn.make_orbit_proposals(model, B, seed=3502), gps])
    worker.lnprob_batch(pfit)
    t2 = time.perf_counter()
    for _ in range(3):
        worker.lnprob_batch(pfit)
    ex["lnprob_of_p_evals_per_s"] = 3 * B / (time.perf_counter() - t2)
    # ... and the same boundary through ONE resident launch: orbital parameters to the lanes, Kepler solve, |v| >= c rule
    # and Doppler shift inside the launch (psoap_stream_submit_orbits), two half-ensembles in flight
    from psoap_amd.chunk import StreamPipeline
    lnp_batch = worker.lnprob_batch(pfit)
    spipe = StreamPipeline(worker.handle, c, B, 2, submit=worker.stream_submit)
    spipe.calibrate(pfit)
    t2s = time.perf_counter()
    spipe.start(pfit)
    for _ in range(5):
        lnp_s = spipe.step(pfit)
    lnp_s = spipe.drain()
    ex["lnprob_of_p_streamed_evals_per_s"] = 6 * B / (time.perf_counter() - t2s)
    spipe.close()
    require(close(lnp_s, lnp_batch), "lnprob(p) through the stream vs the batch path")
    # the sampler boundary (8(f) f-2): B Metropolis-Hastings chains in lock-step on that worker
    cov_mh = 1e-6 * np.eye(pfit.shape[1])
    mh = MultiChainMHSampler(cov_mh, pfit.shape[1], worker.lnprob_batch, B, seeds=[7000 + b for b in range(B)])
    t3 = time.perf_counter()
    mh.run_mcmc(pfit, 3)                              # 1 starting + 3 proposal evaluations of B chains
    ex["mh_sampler_lockstep_evals_per_s"] = 4 * B / (time.perf_counter() - t3)
    # ... and through ONE resident launch (round 5: MultiChainMHSampler.sample_streamed): the chains in two halves, a half's
    # accept / reject and next proposals drawn while the other half is being factored -- the loop of sample_parallel.py:434-438
    # as a consumer of the stream.  Same seeds: the chains must be the lock-step sampler's (decisions identical; lnprob to the
    # parity tolerance -- a lane's plan sums in another order than a batch of 32).
    n_it = 30            # (long enough for the start and the drain not to show: tools/sampler_stream_bench.py)
    worker.stream_open(B)
    mhs = MultiChainMHSampler(cov_mh, pfit.shape[1], None, B, seeds=[7000 + b for b in range(B)])
    list(mhs.sample_streamed(pfit, lambda P, g: worker.stream_submit(P), worker.stream_fetch, groups=2, iterations=1))   # warm
    mhs = MultiChainMHSampler(cov_mh, pfit.shape[1], None, B, seeds=[7000 + b for b in range(B)])
    t3s = time.perf_counter()
    list(mhs.sample_streamed(pfit, lambda P, g: worker.stream_submit(P), worker.stream_fetch, groups=2, iterations=n_it))
    ex["mh_sampler_evals_per_s"] = (n_it + 1) * B / (time.perf_counter() - t3s)      # start + n_it iterations of B chains
    worker.stream_close()
    require(np.array_equal(mhs.chain[:, :3], mh.chain),
            "streamed sampler: the first 3 iterations differ from the lock-step sampler's chains")
    require(close(mhs.lnprobability[:, :3], mh.lnprobability), "streamed sampler: lnprob vs the lock-step sampler")
    ex["mh_sampler_what"] = (f"{B} Metropolis-Hastings chains through one resident launch, two halves in flight, {n_it} iterations "
                             "(+ the starting evaluation); chains equal to the lock-step sampler's")
    worker.close()

    # BASELINE configs[4]: predict_f_g_h at the retrieve shape (N = 8192, M = 2 n_pix = 1024), handle-resident
    ch5 = syn.make_config_chunk(5)
    M5 = 2 * ch5.n_pix
    pred = np.linspace(np.min(ch5.lwls[0]), np.max(ch5.lwls[0]), num=M5)
    h5 = ChunkHandle(ch5.fl, ch5.sigma, max_batch=1, device=dev)
    args5 = (0, ch5.lwls, np.stack([pred] * 3), np.zeros(3), syn.GP_BASE[3])
    mu5, Sig5 = h5.predict(*args5)          # first call: allocates the workspace
    t_first = h5.predict_timings()
    best = None
    for _ in range(3):
        mu5, Sig5 = h5.predict(*args5)
        t = h5.predict_timings()
        if best is None or t["total_ms"] < best["total_ms"]:
            best = t
    gf5 = golden("golden_full_v1.npz")
    require(float(np.max(np.abs(mu5 - gf5["cfg5_pred_mu"]))) <= 1e-10, "predict cfg5: mu vs reference golden")
    require(float(np.max(np.abs(np.diag(Sig5) - gf5["cfg5_pred_diag"]))) <= 1e-9, "predict cfg5: diag(Sigma)")
    require(float(np.max(np.abs(Sig5[gf5["cfg5_pred_row_index"]] - gf5["cfg5_pred_rows"]))) <= 1e-9,
            "predict cfg5: Sigma rows")
    tf = best["flops"] / (best["device_ms"] * 1e-3) / 1e12
    ex["predict_cfg5"] = {
        "workload": f"predict_f_g_h, N={ch5.N}, M={M5} (R = {3 * M5} prediction columns), handle-resident workspace",
        "device_ms": best["device_ms"], "factor_ms": best["factor_ms"], "mean_and_sigma_ms": best["sigma_ms"],
        "sigma_download_ms": best["download_ms"], "total_ms": best["total_ms"], "first_call_total_ms": t_first["total_ms"],
        "roofline": {"bound": "mfma", "achieved": tf, "peak": PEAK_FP64_TFLOPS, "unit": "TFLOP/s",
                     "frac": tf / PEAK_FP64_TFLOPS, "algorithmic_flops": best["flops"]},
        "parity_checked": True}
    h5.close()
    return ex


if __name__ == "__main__":
    main()
