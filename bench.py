#!/usr/bin/env python3
"""Headline benchmark: GP lnprob evaluations per second, SB2 chunk at N = 6000.

One "step" = one ensemble step of the hot path on every rank: the rank's own SB2
chunk (20 epochs x 300 px, N = 6000; BASELINE.json configs[2]/[3]) is evaluated for a
batch of 32 walkers (fill -> +sigma^2 -> Cholesky -> solve -> logdet -> lnprob), then
the per-(walker, chunk) lnprobs are gathered over RCCL and summed in fixed chunk order
(the gather-and-sum of psoap/sample_parallel.py:378-387).  Chunks are independent, so
per-GPU work is fixed as N grows ("weak"); at 8 GPUs a step is exactly the 32-walker x
8-chunk ensemble of configs[3].

    python bench.py --gpus 1 --steps 10 --warmup 2
    python -m torch.distributed.run --nproc-per-node 8 ... bench.py --gpus 8 ...

Rank 0 prints ONE JSON line.  Proposals are resident in HBM before the timed region.
"""
import argparse
import glob
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from psoap_amd import synthetic as syn  # noqa: E402

PEAK_FP64_TFLOPS = 78.6   # MI355X spec, fp64 matrix == fp64 vector
N_WALKERS = 32


def flops_eval(N: int) -> float:
    """Algorithmic flops of one lnprob evaluation: N^3/3 + 2 N^2."""
    return N ** 3 / 3.0 + 2.0 * N ** 2


def flops_panel_update(N: int, nb: int = 128) -> float:
    """Algorithmic share of N^3/3 done by the left-looking panel update: every upper-triangle
    element (i, j >= i) receives one multiply-add from each of the 128*floor(i/128) finished rows."""
    i = np.arange(N, dtype=np.float64)
    k0 = nb * np.floor(i / nb)
    return float(np.sum(2.0 * k0 * (N - i)))


def cpu_baseline(chunk, n_evals: int = 14):
    """The CPU oracle (C fill + SciPy cho_factor/cho_solve, i.e. the reference's own
    library calls) timed on this box's host cores.  Reported, never the target."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle
    oracle.build()
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        threads = os.cpu_count() or 1
    gp = syn.GP_BASE[chunk.n_components]
    V11 = np.empty((chunk.N, chunk.N))
    oracle.lnlike(chunk.lwls, chunk.fl, chunk.sigma, gp, V11=V11)     # warm-up
    ts = []
    val = None
    for _ in range(n_evals):
        t0 = time.perf_counter()
        val = oracle.lnlike(chunk.lwls, chunk.fl, chunk.sigma, gp, V11=V11)
        ts.append(time.perf_counter() - t0)
    med = float(np.median(ts))
    return {"value": 1.0 / med, "unit": "evals/s", "cores": int(threads), "kind": "port",
            "sample": f"{n_evals} evals of lnlike_f_g on the same N={chunk.N} SB2 chunk after 1 warm-up, "
                      f"median {med:.3f} s/eval; C fill (1 thread) + SciPy/OpenBLAS dpotrf/dpotrs "
                      f"({threads} threads); host has {os.cpu_count()} cores",
            "lnprob": float(val)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--walkers", type=int, default=N_WALKERS)
    ap.add_argument("--config", type=int, default=3, help="BASELINE.json config shape (3 = SB2 N=6000)")
    ap.add_argument("--groups", type=int, default=2, help="concurrent stream groups per batch (staged mode)")
    ap.add_argument("--mode", default="dag", choices=["dag", "staged"], help="execution mode of the batch eval")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cfg4", action="store_true",
                    help="skip the 8-chunk ensemble on one GPU (profiling runs: keeps every k_chol_dag dispatch alike)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend (nccl = RCCL; gloo only for single-GPU dry runs of the N>1 path)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
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

    from psoap_amd.chunk import ChunkHandle, microbench
    from psoap_amd.ensemble import gather_and_sum

    # ---- workload: this rank's chunk + the walker ensemble (identical on every rank by seeding)
    cfg = args.config
    chunk = syn.make_config_chunk(cfg, chunk_index=rank)
    c, N, B = chunk.n_components, chunk.N, args.walkers
    gps = syn.make_walkers(c, B, seed=1000 * cfg + 500)
    vels = syn.make_walker_velocities(chunk, B, seed=1000 * cfg + 501 + rank)
    lwls = syn.walker_lwls(chunk, vels)

    h = ChunkHandle(chunk.fl, chunk.sigma, max_batch=B, device=local_rank)
    h.set_stream_groups(args.groups)
    h.set_mode(args.mode)
    h.upload(lwls, gps)          # proposals resident in HBM before the timed region
    h.sync()

    def step():
        h.eval()
        lnp = h.fetch()                                  # (B,) this chunk's lnprob per walker
        return gather_and_sum(lnp, world, local_rank)    # (B,) summed over chunks, fixed order

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        total = step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        total = step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    evals = world * B * args.steps
    value = evals / dt

    # ---- live per-kernel timing (HIP events on the launch stream) for the roofline object
    h.set_profiling(True)
    h.eval()
    h.fetch()
    tm = h.timings()
    h.set_profiling(False)
    mode = "dag" if tm["dag"]["launches"] > 0 else "staged"
    if mode == "dag":
        # one persistent launch does the whole factorisation + solve of the batch:
        # algorithmic flops per launch = B x F(N)  
        dom, dom_name = tm["dag"], "k_chol_dag (persistent tile DAG, v_mfma_f64_16x16x4_f64)"
        alg_per_launch = B * flops_eval(N) / max(1, dom["launches"])
    else:
        dom, dom_name = tm["panel_update"], "k_panel_update (v_mfma_f64_16x16x4_f64)"
        alg_per_launch = B * flops_panel_update(N) / max(1, dom["launches"])
    avg_ms = dom["ms"] / max(1, dom["launches"])
    achieved = alg_per_launch / (avg_ms * 1e-3) / 1e12
    fill = tm["fill"]
    if mode == "dag":
        # the persistent kernel evaluates K on the fly; time the standalone HBM-bound fill kernel
        # (the fill_V11_* drop-in and the staged path use it) with one event-profiled staged step
        h.set_mode("staged")
        h.set_profiling(True)
        h.eval()
        h.fetch()
        fill = h.timings()["fill"]
        h.set_profiling(False)
        h.set_mode("dag")

    # ---- PCIe-inclusive rate (never `value`): proposals uploaded from host memory every step
    t1 = time.perf_counter()
    for _ in range(3):
        h.upload(lwls, gps)
        h.eval()
        h.fetch()
    pcie_value = 3 * B / (time.perf_counter() - t1)

    # ---- the lnprob(p) boundary: orbital parameters in, lnprob out; Kepler solve,
    # Doppler shift and likelihood all on the device.  Reported beside `value`, never as `value`.
    from psoap_amd.lnprob import ChunkWorker
    h.close()
    model = {1: "SB1", 2: "SB2", 3: "ST3"}[c]
    worker = ChunkWorker(model, chunk.lwl, chunk.fl, chunk.sigma, chunk.epoch_index, chunk.dates, max_batch=B,
                         device=local_rank)
    pfit = np.hstack([syn.make_orbit_proposals(model, B, seed=1000 * cfg + 502), gps])
    worker.lnprob_batch(pfit)
    t2 = time.perf_counter()
    for _ in range(3):
        worker.lnprob_batch(pfit)
    lnprob_p_value = 3 * B / (time.perf_counter() - t2)

    # ---- the sampler boundary (8(f) f-2): B Metropolis-Hastings chains in lock-step on that worker,
    # proposals drawn on the host, one batched device evaluation per iteration.
    from psoap_amd.samplers import MultiChainMHSampler
    mh = MultiChainMHSampler(1e-6 * np.eye(pfit.shape[1]), pfit.shape[1], worker.lnprob_batch, B,
                             seeds=[7000 + b for b in range(B)])
    t3 = time.perf_counter()
    mh.run_mcmc(pfit, 3)                              # 1 starting + 3 proposal evaluations of B chains
    mh_value = 4 * B / (time.perf_counter() - t3)
    worker.close()

    # ---- BASELINE configs[3] on ONE GPU (rank 0, single-GPU runs only): 32 walkers x 8 chunks, all eight
    # chunks factored by one launch of the persistent kernel over the heterogeneous batch (ChunkGroup).
    # Reported beside `value`, never as `value`.
    cfg4_value = None
    if world == 1 and cfg == 3 and not args.no_cfg4:
        from psoap_amd.ensemble import EnsembleEvaluator
        chunks8 = [syn.make_config_chunk(cfg, k) for k in range(8)]
        props8 = {k: (np.repeat(chunks8[k].lwls[None], B, axis=0), gps) for k in range(8)}
        ev8 = EnsembleEvaluator.from_chunks(chunks8, max_batch=B, device_index=local_rank)
        ev8.lnprob(props8)
        t4 = time.perf_counter()
        for _ in range(2):
            ev8.lnprob(props8)
        cfg4_value = 2 * 8 * B / (time.perf_counter() - t4)
        ev8.close()

    out = None
    if rank == 0:
        mb = microbench(local_rank)
        traffic, traffic_src = None, None
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")), reverse=True):
            tj = json.load(open(path))
            w = tj.get("workload", {})
            if (w.get("N"), w.get("components"), w.get("walkers"), w.get("mode")) == (N, c, B, mode):
                traffic, traffic_src = tj["hbm_bytes_per_launch"], os.path.relpath(path, ROOT)
                break
        out = {
            "metric": "GP lnprob evals/sec (SB2, N=6000)",
            "value": value, "unit": "evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"SB2 chunk 20 epochs x 300 px (N={N}), {B} walkers per step per GPU, "
                                   f"one chunk per GPU (BASELINE.json configs[2]; configs[3] at 8 GPUs)",
                       "N": N, "components": c, "walkers": B, "chunks_per_gpu": 1, "mode": args.mode, "stream_groups": args.groups,
                       "parallelism": f"chunk-sharded x{world}, RCCL all_gather of walker lnprobs"},
            "roofline": {"bound": "mfma", "kernel": dom_name,
                         "achieved": achieved, "peak": PEAK_FP64_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP64_TFLOPS, "traffic": traffic, "traffic_unit": "bytes/launch",
                         "traffic_source": traffic_src,
                         "launches_per_step": dom["launches"], "avg_launch_ms": avg_ms,
                         "algorithmic_flops_per_launch": alg_per_launch,
                         "executed_tflops": dom["flops"] / (dom["ms"] * 1e-3) / 1e12 if dom["ms"] > 0 else None,
                         "measured_peak": mb["mfma_f64_tflops"]},
            "roofline_eval": {"bound": "mfma", "achieved": value / world * flops_eval(N) / 1e12,
                              "peak": PEAK_FP64_TFLOPS, "unit": "TFLOP/s",
                              "frac": value / world * flops_eval(N) / 1e12 / PEAK_FP64_TFLOPS,
                              "flops_per_eval": flops_eval(N)},
            "roofline_fill": {"bound": "hbm", "kernel": "k_fill_sym<2> (upper tiles)",
                              "achieved": fill["bytes"] / (fill["ms"] * 1e-3) / 1e9 if fill["ms"] > 0 else None,
                              "peak": 8000.0, "unit": "GB/s",
                              "frac": fill["bytes"] / (fill["ms"] * 1e-3) / 1e9 / 8000.0 if fill["ms"] > 0 else None,
                              "measured_write_peak": mb["hbm_write_gbs"]},
            "kernel_ms_profiled_step": {k: round(tm[k]["ms"], 3) for k in
                                        ("fill", "panel_update", "potrf", "trsm", "misc", "dag")},
            "profiled_step_total_ms": tm["total_ms"],
            "pcie_inclusive_evals_per_s": pcie_value,
            "lnprob_of_p_evals_per_s": lnprob_p_value,
            "mh_sampler_evals_per_s": mh_value,
            "cfg4_one_gpu_evals_per_s": cfg4_value,
            "lnprob_walker0": float(total[0]),
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(chunk)
            out["speedup_vs_cpu"] = value / out["cpu_baseline"]["value"]
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
