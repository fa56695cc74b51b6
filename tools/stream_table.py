"""Streamed evaluation against one launch per step over the BASELINE shapes at 32 walkers (profiles/rN_stream_table.jsonl):
ms per ensemble step in the steady state and over a whole run of `steps` steps (start-up stagger and drain included), as
fractions of the fp64 peak, on one box in one call.  One JSON line per shape."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from psoap_amd import synthetic as syn  # noqa: E402
from psoap_amd.chunk import ChunkHandle, StreamPipeline  # noqa: E402

PEAK = 78.6e12
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for cfg in (1, 2, 3, 5):
    ch = syn.make_config_chunk(cfg)
    c, N, B = ch.n_components, ch.N, 32
    F = N ** 3 / 3.0 + 2.0 * N ** 2
    gps = syn.make_walkers(c, B, seed=cfg)
    lw = syn.walker_lwls(ch, syn.make_walker_velocities(ch, B, seed=cfg + 10))
    with ChunkHandle(ch.fl, ch.sigma, max_batch=B) as h:
        ref = h.lnlike_batch(lw, gps)
        h.upload(lw, gps)
        ts = []
        for _ in range(max(5, steps // 2)):
            t0 = time.perf_counter()
            h.eval(); h.upload(lw, gps); h.fetch()
            ts.append(time.perf_counter() - t0)
        per_step = float(np.median(ts))
        row = dict(cfg=cfg, N=N, c=c, B=B, launch_per_step_ms=round(1e3 * per_step, 3), launch_per_step_frac=round(B * F / per_step / PEAK, 4))
        for groups in (2, 4):
            pipe = StreamPipeline(h, c, B, groups)
            pipe.calibrate(lw, gps)
            best = None
            for _ in range(2):
                t0 = time.perf_counter()
                pipe.start(lw, gps)
                marks = []
                for _ in range(steps - 1):
                    out = pipe.step(lw, gps)
                    marks.append(time.perf_counter())
                out = pipe.drain()
                h.stream_pause()
                whole = (time.perf_counter() - t0) / steps
                steady = (marks[-1] - marks[1]) / (len(marks) - 2)
                if best is None or whole < best[0]:
                    best = (whole, steady)
            st = h.stream_stats()
            pipe.close()
            dev = float(np.max(np.abs(out - ref) / np.maximum(1.0, np.abs(ref))))
            row.update({f"stream{groups}_ms": round(1e3 * best[0], 3), f"stream{groups}_frac": round(B * F / best[0] / PEAK, 4),
                        f"stream{groups}_steady_ms": round(1e3 * best[1], 3), f"stream{groups}_steady_frac": round(B * F / best[1] / PEAK, 4),
                        "scheme": st["scheme"], "tasks_per_matrix": st["tasks_per_matrix"], "max_rel_dev_vs_batch": dev})
        row["steps"] = steps
        print(json.dumps(row), flush=True)
