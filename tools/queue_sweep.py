"""Ticket queues in use (PSOAP_DAG_QUEUES=8 | 4 | 2 | 1 against the automatic rule, dag_kernel.hpp: dag_queue_count) over batch
sizes that are not multiples of 8."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from psoap_amd import synthetic as syn
from psoap_amd.chunk import ChunkHandle

cfgs = [int(a) for a in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1, 2, 3, 5]
Bs = [int(a) for a in sys.argv[2].split(",")] if len(sys.argv) > 2 else (9, 10, 12, 13, 16, 17, 20, 24, 25, 28, 32)
for cfg in cfgs:
    ch = syn.make_config_chunk(cfg)
    c = ch.n_components
    for B in Bs:
        gps = syn.make_walkers(c, B, seed=1)
        lw = np.repeat(ch.lwls[None], B, axis=0)
        res = {}
        for nq in (8, 4, 2, 1, 0):
            if nq:
                os.environ["PSOAP_DAG_QUEUES"] = str(nq)
            else:
                os.environ.pop("PSOAP_DAG_QUEUES", None)
            with ChunkHandle(ch.fl, ch.sigma, max_batch=B) as h:
                h.upload(lw, gps)
                for _ in range(2):
                    h.eval(); out = h.fetch()
                n = 4
                t0 = time.perf_counter()
                for _ in range(n):
                    h.eval(); h.fetch()
                res[nq] = (1e3 * (time.perf_counter() - t0) / n, out.copy())
        assert all(np.allclose(res[8][1], res[k][1], rtol=1e-11, atol=0) for k in res), res
        print(f"N={ch.N:5d} B={B:2d}: 8 queues {res[8][0]:8.2f} ms   4 {res[4][0]:8.2f} ms   2 {res[2][0]:8.2f} ms   1 {res[1][0]:8.2f} ms   auto {res[0][0]:8.2f} ms",
              flush=True)
