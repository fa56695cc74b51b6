import os, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from psoap_amd import synthetic as syn
from psoap_amd.chunk import ChunkHandle
for cfg in (1, 2, 3, 5):
    ch = syn.make_config_chunk(cfg); c = ch.n_components
    for B in (16, 32):
        gps = syn.make_walkers(c, B, seed=1)
        lw = syn.walker_lwls(ch, syn.make_walker_velocities(ch, B, seed=3))
        res = {}
        for pf in ("0", "1"):
            os.environ["PSOAP_DAG_PREFILL"] = pf
            os.environ["PSOAP_DAG_SCHEME"] = "0"
            with ChunkHandle(ch.fl, ch.sigma, max_batch=B) as h:
                h.upload(lw, gps)
                for _ in range(2):
                    h.eval(); out = h.fetch()
                ts = []
                for _ in range(6):
                    t0 = time.perf_counter(); h.eval(); h.fetch(); ts.append(time.perf_counter() - t0)
                res[pf] = (1e3 * float(np.median(ts)), out.copy())
        eq = np.array_equal(res["0"][1], res["1"][1])
        dev = float(np.max(np.abs(res["0"][1] - res["1"][1]) / np.abs(res["0"][1])))
        print(f"N={ch.N} B={B}: fused {res['0'][0]:.3f} ms  prefilled {res['1'][0]:.3f} ms  ratio {res['1'][0]/res['0'][0]:.4f}  equal {eq} maxrel {dev:.2e}", flush=True)
