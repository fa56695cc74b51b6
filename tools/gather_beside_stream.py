"""Does an RCCL all_gather per sub-ensemble run BESIDE a resident stream launch, or behind it?  One rank, backend nccl,
device 0 (a scaling node is not available on this pool; what can starve a collective -- a resident grid that holds every
register file -- is the same with one rank as with eight).  Per step and half:  fetch(half) -> gather(half, through the
process group) -> submit(half's next proposals)  -- the dependence of /root/reference/psoap/sample_parallel.py:378-390.

    python -m torch.distributed.run --standalone --local-addr 127.0.0.1 --nnodes=1 --nproc-per-node=1 \
        tools/gather_beside_stream.py [steps] [reserve,reserve,...]

Prints one JSON line per reserve value: ms per step without any collective, with the gather, their ratio."""
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
reserves = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "0,8").split(",")]
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


def gather_half(g, rows, lnp_rows):
    table[:, rows] = gather_chunk_lnprobs(lnp_rows[None, :], 1, world, rank, 0, force_collective=True)
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


for reserve in reserves:
    with ChunkHandle(ch.fl, ch.sigma, max_batch=B, device=0) as h:
        h.set_stream_reserve(reserve)
        pipe = StreamPipeline(h, c, B, 2)
        pipe.calibrate(*sets[0])
        gather_half(0, slice(0, B // 2), np.zeros(B // 2))           # communicator and buffers come up outside the timing
        ms_plain, ref = run(pipe, None, steps)
        ms_coll, got = run(pipe, gather_half, steps)
        ms_plain2, _ = run(pipe, None, steps)
        same = bool(np.array_equal(ref, got))
        pipe.close()
    best_plain = min(ms_plain, ms_plain2)
    print("RESULT " + json.dumps({"N": ch.N, "walkers": B, "reserve": reserve, "steps": steps,
                                  "ms_per_step_no_collective": [round(ms_plain, 3), round(ms_plain2, 3)],
                                  "ms_per_step_with_gather_per_half": round(ms_coll, 3),
                                  "ratio": round(ms_coll / best_plain, 4), "collectives": n_coll["n"],
                                  "values_equal": same, "backend": dist.get_backend(), "ranks": dist.get_world_size()}), flush=True)
release_gather_buffers()
dist.barrier()
dist.destroy_process_group()
