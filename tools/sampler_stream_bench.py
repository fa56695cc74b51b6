"""The streamed Metropolis-Hastings sampler (samplers.sample_streamed) on the bench chunk: evals/s over n iterations for 2 and 4
sub-ensembles, beside the lock-step sampler and the raw stream rate.    python tools/sampler_stream_bench.py [iterations]"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from psoap_amd import synthetic as syn
from psoap_amd.lnprob import ChunkWorker
from psoap_amd.samplers import MultiChainMHSampler
from psoap_amd.chunk import StreamPipeline
n_it = int(sys.argv[1]) if len(sys.argv) > 1 else 30
B = 32
chunk = syn.make_config_chunk(3)
c = chunk.n_components
gps = syn.make_walkers(c, B, seed=3500)
worker = ChunkWorker("SB2", chunk.lwl, chunk.fl, chunk.sigma, chunk.epoch_index, chunk.dates, max_batch=B, device=0)
pfit = np.hstack([syn.make_orbit_proposals("SB2", B, seed=3502), gps])
cov = 1e-6 * np.eye(pfit.shape[1])
seeds = [7000 + b for b in range(B)]
worker.lnprob_batch(pfit)
mh = MultiChainMHSampler(cov, pfit.shape[1], worker.lnprob_batch, B, seeds=seeds)
t0 = time.perf_counter(); mh.run_mcmc(pfit, n_it); dt = time.perf_counter() - t0
out = {"iterations": n_it, "lockstep_evals_per_s": round((n_it + 1) * B / dt, 1)}
spipe = StreamPipeline(worker.handle, c, B, 2, submit=worker.stream_submit)
spipe.calibrate(pfit)
t0 = time.perf_counter(); spipe.start(pfit)
for _ in range(n_it - 1):
    spipe.step(pfit)
spipe.drain(); dt = time.perf_counter() - t0
out["raw_stream_evals_per_s"] = round(n_it * B / dt, 1)
spipe.close()
for groups in (2, 4):
    worker.stream_open(B)
    s = MultiChainMHSampler(cov, pfit.shape[1], None, B, seeds=seeds)
    list(s.sample_streamed(pfit, lambda P, g: worker.stream_submit(P), worker.stream_fetch, groups=groups, iterations=2))
    s = MultiChainMHSampler(cov, pfit.shape[1], None, B, seeds=seeds)
    host = {"t": 0.0}
    t0 = time.perf_counter()
    list(s.sample_streamed(pfit, lambda P, g: worker.stream_submit(P), worker.stream_fetch, groups=groups, iterations=n_it))
    dt = time.perf_counter() - t0
    worker.stream_close()
    out[f"streamed_{groups}_groups_evals_per_s"] = round((n_it + 1) * B / dt, 1)
    out[f"chains_equal_lockstep_{groups}"] = bool(np.array_equal(s.chain, mh.chain))
worker.close()
print(json.dumps(out))
