"""Scheme comparison of the persistent DAG kernel (PSOAP_DAG_SCHEME=0 throughput / 1 latency / 2 following, and the automatic
rule) over batch sizes: is dag_auto_scheme still where the crossovers are?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from psoap_amd import synthetic as syn
from psoap_amd.chunk import ChunkHandle

for cfg in (1, 2, 3, 5):
    ch = syn.make_config_chunk(cfg)
    c = ch.n_components
    for B in (1, 2, 4, 8, 12, 16, 24, 32):
        gps = syn.make_walkers(c, B, seed=1)
        lw = np.repeat(ch.lwls[None], B, axis=0)
        res = {}
        for scheme in (0, 1, 2, -1):
            if scheme >= 0:
                os.environ["PSOAP_DAG_SCHEME"] = str(scheme)
            else:
                os.environ.pop("PSOAP_DAG_SCHEME", None)
            with ChunkHandle(ch.fl, ch.sigma, max_batch=B) as h:
                h.upload(lw, gps)
                for _ in range(2):
                    h.eval(); out = h.fetch()
                n = 4
                t0 = time.perf_counter()
                for _ in range(n):
                    h.eval(); h.fetch()
                res[scheme] = (1e3 * (time.perf_counter() - t0) / n, out[0])
        assert all(abs(res[0][1] - res[k][1]) <= 1e-10 * abs(res[0][1]) for k in res), res
        best = min((0, 1, 2), key=lambda k: res[k][0])
        print(f"N={ch.N:5d} B={B:2d}: scheme 0 {res[0][0]:8.2f} ms   1 {res[1][0]:8.2f} ms   2 {res[2][0]:8.2f} ms   auto {res[-1][0]:8.2f} ms"
              f"   best {best}   auto/best {res[-1][0] / res[best][0]:.3f}", flush=True)
