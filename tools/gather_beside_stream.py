"""Does an RCCL all_gather per sub-ensemble run BESIDE a resident stream launch, or behind it?  One rank, backend nccl,
device 0 (a scaling node is not available on this pool; what can starve a collective -- a resident grid that holds every
register file -- is the same with one rank as with eight).  Per step and half:  fetch(half) -> gather(half, through the
process group) -> submit(half's next proposals)  -- the dependence of /root/reference/psoap/sample_parallel.py:378-390.

    python -m torch.distributed.run --standalone --local-addr 127.0.0.1 --nnodes=1 --nproc-per-node=1 \
        tools/gather_beside_stream.py [steps] [device,host,per-step]

Prints one JSON line per way of gathering -- "device": all_gather_into_tensor over RCCL on device tensors beside the resident
launch (it waits for the launch to leave: 2.4 x the step time); "host": the gloo side group of psoap_amd.ensemble.host_group
on the pinned results (the half waits ~2 ms for the exchange with only the other half in flight: 1.08-1.12 x); "per-step":
the launch-per-step path with the RCCL gather between two launches (what several ranks run by default) --: ms per step
without any collective, with the gather, their ratio.  profiles/r5_gather_beside_stream.txt."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

from psoap_amd import synthetic as syn
from psoap_amd.chunk import ChunkHandle, StreamPipeline
from psoap_amd.ensemble import gather_chunk_lnprobs, release_gather_buffers

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
ways = (sys.argv[2] if len(sys.argv) > 2 else "device,host,per-step").split(",")
rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))
B = 32
ch = syn.make_config_chunk(3)
c = ch.n_components
gps = syn.make_walkers(c, B, seed=3500)
lw = syn.walker_lwls(ch, syn.make_walker_velocities(ch, B, seed=3501))
sets = [(lw, gps), (np.roll(lw, 1, axis=0).copy(), np.roll(gps, 1, axis=0).copy())]
n_coll = {"n": 0}
table = np.zeros((1, B))


on_host = {"v": False}


def gather_half(g, rows, lnp_rows):
    table[:, rows] = gather_chunk_lnprobs(lnp_rows[None, :], 1, world, rank, 0, force_collective=True, on_host=on_host["v"])
    n_coll["n"] += 1


def run(pipe, between, n):
    pipe.start(*sets[0])
    t0 = time.perf_counter()
    out = None
    for k in range(1, n + 1):
        out = pipe.step(*sets[k & 1], between=between)
    dt = time.perf_counter() - t0
    pipe.drain(between=between)
    return 1e3 * dt / n, out


def run_per_step(h, with_gather, n):
    """the launch-per-step path: eval(k) || upload(k+1), fetch(k), gather(k) over RCCL between two launches"""
    h.upload(*sets[0])
    h.sync()
    out = None
    t0 = time.perf_counter()
    for k in range(1, n + 1):
        h.eval()
        h.upload(*sets[k & 1])
        out = h.fetch()
        if with_gather:
            gather_half(0, slice(0, B), out)
    dt = time.perf_counter() - t0
    h.sync()
    return 1e3 * dt / n, out


for way in ways:
    on_host["v"] = way == "host"
    if way == "per-step":
        table = np.zeros((1, B))
        with ChunkHandle(ch.fl, ch.sigma, max_batch=B, device=0) as h:
            run_per_step(h, True, 3)
            ms_plain, ref = run_per_step(h, False, steps)
            ms_coll, got = run_per_step(h, True, steps)
            ms_plain2, _ = run_per_step(h, False, steps)
        print("RESULT " + json.dumps({"N": ch.N, "walkers": B, "gather": way, "steps": steps,
                                      "ms_per_step_no_collective": [round(ms_plain, 3), round(ms_plain2, 3)],
                                      "ms_per_step_with_gather_per_step": round(ms_coll, 3),
                                      "ratio": round(ms_coll / min(ms_plain, ms_plain2), 4), "collectives": n_coll["n"],
                                      "values_equal": bool(np.array_equal(ref, got)), "backend": dist.get_backend(),
                                      "ranks": dist.get_world_size()}), flush=True)
        continue
    with ChunkHandle(ch.fl, ch.sigma, max_batch=B, device=0) as h:
        pipe = StreamPipeline(h, c, B, 2)
        pipe.calibrate(*sets[0])
        gather_half(0, slice(0, B // 2), np.zeros(B // 2))           # communicator and buffers come up outside the timing
        ms_plain, ref = run(pipe, None, steps)
        ms_coll, got = run(pipe, gather_half, steps)
        ms_plain2, _ = run(pipe, None, steps)
        same = bool(np.array_equal(ref, got))
        pipe.close()
    best_plain = min(ms_plain, ms_plain2)
    print("RESULT " + json.dumps({"N": ch.N, "walkers": B, "gather": way, "steps": steps,
                                  "ms_per_step_no_collective": [round(ms_plain, 3), round(ms_plain2, 3)],
                                  "ms_per_step_with_gather_per_half": round(ms_coll, 3),
                                  "ratio": round(ms_coll / best_plain, 4), "collectives": n_coll["n"],
                                  "values_equal": same, "backend": dist.get_backend(), "ranks": dist.get_world_size()}), flush=True)
release_gather_buffers()
dist.barrier()
dist.destroy_process_group()
