"""Quick latency table of the persistent kernel (dag mode only): ms per batch for the BASELINE shapes at
B = 1, 4, 32, plus the predict call at the retrieve shape.  One JSON line per row."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from psoap_amd import synthetic as syn
from psoap_amd.chunk import ChunkHandle

def flops_eval(N): return N**3 / 3.0 + 2.0 * N**2
cfgs = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1, 2, 3, 5]
Bs = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 4, 32]
for cfg in cfgs:
    ch = syn.make_config_chunk(cfg)
    c = ch.n_components
    for B in Bs:
        gps = syn.make_walkers(c, B, seed=1)
        lw = np.repeat(ch.lwls[None], B, axis=0)
        with ChunkHandle(ch.fl, ch.sigma, max_batch=B) as h:
            h.upload(lw, gps)
            for _ in range(3):
                h.eval(); out = h.fetch()
            n = 10 if B < 32 else 5
            ts = []
            for _ in range(n):
                t0 = time.perf_counter()
                h.eval(); h.fetch()
                ts.append(time.perf_counter() - t0)
            dt = float(np.median(ts))
            print(json.dumps(dict(cfg=cfg, N=ch.N, c=c, B=B, ms=round(1e3 * dt, 3), min_ms=round(1e3 * min(ts), 3),
                                  evals_per_s=round(B / dt, 1), tflops=round(B * flops_eval(ch.N) / dt / 1e12, 2),
                                  frac=round(B * flops_eval(ch.N) / dt / 78.6e12, 3), lnp0=float(out[0]))), flush=True)
if "nopredict" not in sys.argv:
    ch = syn.make_config_chunk(5)
    M = 2 * ch.n_pix
    pred = np.linspace(ch.lwls[0].min(), ch.lwls[0].max(), M)
    with ChunkHandle(ch.fl, ch.sigma, max_batch=1) as h:
        best = None
        for _ in range(4):
            h.predict(0, ch.lwls, np.stack([pred] * 3), np.zeros(3), syn.GP_BASE[3])
            t = h.predict_timings()
            if best is None or t["device_ms"] < best["device_ms"]:
                best = t
        best["tflops"] = best["flops"] / best["device_ms"] / 1e9
        print(json.dumps({"predict_cfg5": {k: round(v, 3) for k, v in best.items()}}), flush=True)
