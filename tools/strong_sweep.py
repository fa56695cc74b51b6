"""configs[3] on one GPU (8 chunks x 32 walkers in one launch) under the environment's scheduler knobs: ms per step."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from psoap_amd import synthetic as syn
from psoap_amd.ensemble import EnsembleEvaluator
B = 32
chunks = [syn.make_config_chunk(4, k) for k in range(8)]
gps = syn.make_walkers(2, B, seed=4500)
props = {k: (syn.walker_lwls(chunks[k], syn.make_walker_velocities(chunks[k], B, seed=4501 + k)), gps) for k in range(8)}
ev = EnsembleEvaluator.from_chunks(chunks, max_batch=B)
ev.lnprob(props); ev.upload(props)
t0 = time.perf_counter(); n = 4
for _ in range(n):
    ev.launch(); ev.upload(props); tot = ev.collect()
dt = (time.perf_counter() - t0) / n
print(f"{1e3 * dt:8.2f} ms per step, {256 / dt:7.1f} evals/s, frac {256 / dt * 7.2072e10 / 78.6e12:.4f}")
ev.close()
