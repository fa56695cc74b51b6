"""Stand-alone symmetric fill kernel (k_fill_sym, upper tiles) at the BASELINE shapes: GB/s of algorithmic
bytes (4 N (N+1) + 8 (c+1) N per matrix), from one event-profiled staged step with 32 matrices."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from psoap_amd import synthetic as syn
from psoap_amd.chunk import ChunkHandle, microbench
for cfg in (2, 3, 5):
    ch = syn.make_config_chunk(cfg)
    c, B = ch.n_components, 32 if cfg != 5 else 16
    gps = syn.make_walkers(c, B, seed=1)
    lw = np.repeat(ch.lwls[None], B, axis=0)
    with ChunkHandle(ch.fl, ch.sigma, max_batch=B) as h:
        h.set_mode("staged"); h.set_stream_groups(1)
        h.upload(lw, gps); h.eval(); h.fetch()
        best = None
        for _ in range(3):
            h.set_profiling(True); h.eval(); h.fetch(); f = h.timings()["fill"]; h.set_profiling(False)
            gbs = f["bytes"] / (f["ms"] * 1e-3) / 1e9
            best = gbs if best is None or gbs > best else best
        print(json.dumps(dict(cfg=cfg, N=ch.N, c=c, B=B, fill_ms=round(f["ms"], 4), fill_gbs=round(best, 1))), flush=True)
if "ceil" in sys.argv:
    print(json.dumps(microbench(0)))
