"""Streamed evaluation against the plain launch-per-step path on the same proposals (tools; not a test).

    python tools/stream_bench.py [N_epochs n_pix walkers groups steps scheme]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from psoap_amd import synthetic as syn  # noqa: E402
from psoap_amd.chunk import ChunkHandle, StreamPipeline  # noqa: E402

ne, npx, B, G, steps, scheme, anyorder = (int(a) for a in (sys.argv[1:8] + ["20", "300", "32", "2", "10", "-1", "0"][len(sys.argv) - 1:]))
ch = syn.make_chunk(2, ne, npx, seed=3000)
c, N = ch.n_components, ch.N
gps = syn.make_walkers(c, B, seed=3500)
lw = syn.walker_lwls(ch, syn.make_walker_velocities(ch, B, seed=3501))
F = N ** 3 / 3.0 + 2.0 * N ** 2

h = ChunkHandle(ch.fl, ch.sigma, max_batch=B)
ref = h.lnlike_batch(lw, gps)
t0 = time.perf_counter()
for _ in range(steps):
    h.upload(lw, gps); h.eval(); h.fetch()
dt = (time.perf_counter() - t0) / steps
print(f"N={N} B={B}: plain launch per step {1e3 * dt:8.3f} ms  {B / dt:8.1f} evals/s  {B * F / dt / 1e12 / 78.6:.3f} of peak")

pipe = StreamPipeline(h, c, B, G, scheme)
print("stream:", h.stream_stats())
per = pipe.calibrate(lw, gps)
print(f"calibrated period {1e3 * per:.3f} ms")
for rep in range(3):
    t0 = time.perf_counter()
    pipe.start(lw, gps)
    marks = []
    for _ in range(steps - 1):
        out = pipe.step_any_order(lw, gps) if anyorder else pipe.step(lw, gps)
        marks.append(time.perf_counter())
    out = pipe.drain()
    dt = (time.perf_counter() - t0) / steps
    if len(marks) > 4:      # the steady state: from the end of the second step to the end of the last one before the drain
        ss = (marks[-1] - marks[1]) / (len(marks) - 2)
        print(f"  steady state {1e3 * ss:8.3f} ms/step  {B / ss:8.1f} evals/s  {B * F / ss / 1e12 / 78.6:.3f} of peak", end="")
    dev = np.max(np.abs(out - ref) / np.maximum(1.0, np.abs(ref)))
    print(f"  streamed, {G} groups{' (completion order)' if anyorder else ''}, {steps} steps incl. start + drain: {1e3 * dt:8.3f} ms/step  {B / dt:8.1f} evals/s  "
          f"{B * F / dt / 1e12 / 78.6:.3f} of peak   max rel dev vs plain {dev:.2e}")
print("stream:", h.stream_stats())
pipe.close()
h.close()
