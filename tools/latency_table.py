"""Throughput / latency of lnlike for the BASELINE config shapes at several batch sizes."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from psoap_amd import synthetic as syn
from psoap_amd.chunk import ChunkHandle

def flops_eval(N): return N**3 / 3.0 + 2.0 * N**2
rows = []
for cfg in (1, 2, 3, 5):
    ch = syn.make_config_chunk(cfg)
    c = ch.n_components
    for B in (1, 4, 32):
        gps = syn.make_walkers(c, B, seed=1)
        lw = np.repeat(ch.lwls[None], B, axis=0)
        with ChunkHandle(ch.fl, ch.sigma, max_batch=B) as h:
            for mode in ("dag", "staged"):
                h.set_mode(mode)
                h.upload(lw, gps)
                for _ in range(2):
                    h.eval(); h.fetch()
                n = 5
                t0 = time.perf_counter()
                for _ in range(n):
                    h.eval(); h.fetch()
                dt = (time.perf_counter() - t0) / n
                rows.append(dict(cfg=cfg, N=ch.N, c=c, B=B, mode=mode, ms_per_batch=1e3 * dt, evals_per_s=B / dt,
                                 tflops=B * flops_eval(ch.N) / dt / 1e12))
                print(json.dumps(rows[-1]), flush=True)
