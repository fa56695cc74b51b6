"""Several chunks per GPU: one launch per chunk vs ONE launch over all chunks (ChunkGroup).
python tools/multichunk_bench.py [config] [n_chunks] [walkers]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from psoap_amd import synthetic as syn
from psoap_amd.ensemble import EnsembleEvaluator

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n_chunks = int(sys.argv[2]) if len(sys.argv) > 2 else 8
B = int(sys.argv[3]) if len(sys.argv) > 3 else 32
chunks = [syn.make_config_chunk(cfg, k) for k in range(n_chunks)]
c = chunks[0].n_components
gps = syn.make_walkers(c, B, seed=7)
props = {k: (np.repeat(chunks[k].lwls[None], B, axis=0), gps) for k in range(n_chunks)}
ev = EnsembleEvaluator.from_chunks(chunks, max_batch=B)
for name in ("one launch", "per chunk"):
    if name == "per chunk":
        run = lambda: np.sum([ev.handles[k].lnlike_batch(*props[k]) for k in range(n_chunks)], axis=0)
    else:
        run = lambda: ev.lnprob(props)
    out = run(); run()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        out = run()
    dt = (time.perf_counter() - t0) / n
    print(f"N={chunks[0].N} x {n_chunks} chunks x {B} walkers, {name:10s}: {1e3*dt:8.2f} ms per ensemble step  ({n_chunks*B/dt:9.1f} evals/s)  lnprob[0]={out[0]:.6f}")
ev.close()
