"""Per-task timeline of a STREAMED run (debug / analysis aid; profiles/r4_timeline.txt): delivered TFLOP/s and tasks in
flight per time slice over `steps` ensemble steps through one resident launch, and where the tasks' time goes.

    python tools/stream_timeline.py [cfg walkers groups steps scheme]
"""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from psoap_amd import synthetic as syn  # noqa: E402
from psoap_amd.chunk import ChunkHandle, StreamPipeline  # noqa: E402

cfg, B, G, steps, scheme, anyorder = (int(a) for a in (sys.argv[1:7] + ["3", "32", "2", "6", "-1", "0"][len(sys.argv) - 1:]))
ch = syn.make_config_chunk(cfg)
c, N = ch.n_components, ch.N
gps = syn.make_walkers(c, B, seed=1)
lw = syn.walker_lwls(ch, syn.make_walker_velocities(ch, B, seed=2))
task_dt = np.dtype([("type", "u1"), ("q", "u1"), ("j", "u1"), ("S", "u1"), ("b", "<u2"), ("pa", "u1"), ("pb", "u1"),
                    ("slot", "<u4"), ("ctr", "<u4")])
F = N ** 3 / 3.0 + 2.0 * N ** 2
h = ChunkHandle(ch.fl, ch.sigma, max_batch=B)
# calibration in a stream of its own (the task log must be allocated before the first submission of the logged one)
pipe = StreamPipeline(h, c, B, G, scheme)
period = pipe.calibrate(lw, gps)
pipe.close()
pipe = StreamPipeline(h, c, B, G, scheme)
pipe.period = period
nsub = B * steps
L = h._L
L.psoap_stream_tasklog(h._h, nsub, None, 0)
n = ctypes.c_longlong(0)
L.psoap_stream_tasks(h._h, None, 0, ctypes.byref(n))
nt = n.value
tasks = np.zeros(nt, dtype=task_dt)
L.psoap_stream_tasks(h._h, tasks.ctypes.data_as(ctypes.c_void_p), nt, ctypes.byref(n))
t0 = time.perf_counter()
pipe.start(lw, gps)
for _ in range(steps - 1):
    pipe.step_any_order(lw, gps) if anyorder else pipe.step(lw, gps)
pipe.drain()
wall = time.perf_counter() - t0
st = h.stream_stats()
# (reading the log needs the resident launch gone: it leaves after the idle time-out)
log = np.zeros(nsub * nt * 8, dtype=np.uint64)
L.psoap_stream_tasklog(h._h, nsub, log.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), log.size)
pipe.close()
h.close()
log = log.reshape(nsub, nt, 8)
waits = (log[:, :, 7] >> np.uint64(40)).astype(np.float64) / 100.0      # us spent in the update's dependency waits
log = log.astype(np.float64) / 100.0       # us
assert (log[:, :, 0] > 0).all(), "tasks without a start stamp"
base = log[:, :, 0].min()
log = np.where(log > 0, log - base, 0.0)
span = log[:, :, 3].max()
ty = np.broadcast_to(tasks["type"] & 0x0F, (nsub, nt))
q_of = np.broadcast_to(tasks["q"].astype(int), (nsub, nt))
pa = np.broadcast_to(tasks["pa"].astype(float), (nsub, nt))
pb = np.broadcast_to(tasks["pb"].astype(float), (nsub, nt))
part, diag, off = ty == 0, ty == 1, ty == 2
dur = log[:, :, 3] - log[:, :, 0]
print(f"N={N} {B} walkers in {G} groups, {steps} steps = {nsub} matrices through one resident launch (scheme {st['scheme']}, "
      f"{nt} tasks per matrix, launches {st['launches']}); wall {1e3 * wall:.2f} ms = {1e3 * wall / steps:.3f} ms per step = "
      f"{nsub / wall:.1f} evals/s = {nsub * F / wall / 1e12 / 78.6:.3f} of peak; device span {span / 1e3:.2f} ms")
print(f"sum(task time) {dur.sum() / 1e3:.1f} ms -> average concurrency {dur.sum() / span:.1f}")
d01 = log[:, :, 1] - log[:, :, 0]
d12 = log[:, :, 2] - log[:, :, 1]
d23 = log[:, :, 3] - log[:, :, 2]
print(f"PART: total {dur[part].sum() / 1e3:9.1f} ms")
print(f"OFF : update+store {d01[off].sum() / 1e3:9.1f} ms | wait potrf {d12[off].sum() / 1e3:8.1f} ms | trsm+publish {d23[off].sum() / 1e3:8.1f} ms")
print(f"DIAG: update+store {d01[diag].sum() / 1e3:9.1f} ms | potrf      {d12[diag].sum() / 1e3:8.1f} ms | publish      {d23[diag].sum() / 1e3:8.1f} ms")
# the wait for the block row above inside the update: the last panel starts at stamp 4; its own length is [5] - [4]; the
# panels before it are assumed to run at that pace
have4 = (log[:, :, 4] > 0) & (log[:, :, 5] > 0) & off
lastp = (log[:, :, 5] - log[:, :, 4])
est_wait = np.where(have4, (log[:, :, 4] - log[:, :, 0]) - (pb - pa - 1) * lastp, 0.0).clip(0)
print(f"OFF : estimated wait for the row above inside the update {est_wait.sum() / 1e3:8.1f} ms "
      f"({100 * est_wait.sum() / dur.sum():.1f} % of all task time); last-panel length median {np.median(lastp[have4]):.1f} us")
print(f"measured dependency waits inside updates: {waits.sum() / 1e3:8.1f} ms ({100 * waits.sum() / dur.sum():.1f} % of all task time): "
      f"PART {waits[part].sum() / 1e3:.1f}, OFF {waits[off].sum() / 1e3:.1f}, DIAG {waits[diag].sum() / 1e3:.1f} ms; "
      f"tasks waiting > 20 us: {100 * (waits > 20).mean():.1f} %, > 200 us: {100 * (waits > 200).mean():.1f} %")
for q in (1, 5, 10, 20, 30, 40, 46):
    m = (q_of == q) & off
    if m.any():
        print(f"  q={q:2d}: OFF update+store median {np.median(d01[m]):7.1f} us, wait-potrf median {np.median(d12[m]):6.1f}, "
              f"trsm median {np.median(d23[m]):6.1f}, measured wait mean {np.mean(waits[m]):6.1f} / p90 {np.percentile(waits[m], 90):6.1f} us")
# per matrix: submission -> completion
m_start = log[:, :, 0].min(axis=1)
m_end = log[:, :, 3].max(axis=1)
print(f"matrix in flight: median {np.median(m_end - m_start) / 1e3:.2f} ms (min {np.min(m_end - m_start) / 1e3:.2f}, max {np.max(m_end - m_start) / 1e3:.2f})")
# per lane: the time between the end of a matrix and the first task of its successor (fetch of the whole group, host
# turnaround, dispatcher) -- lanes are handed out lowest first, so submission s sits in lane (s mod B)
gaps = [m_start[s + B] - m_end[s] for s in range(nsub - B)]
if gaps:
    print(f"lane idle between a matrix and its successor: median {np.median(gaps):.0f} us (min {np.min(gaps):.0f}, max {np.max(gaps):.0f}); "
          f"group completion spread (last - first end) median {np.median([m_end[s:s + B // G].max() - m_end[s:s + B // G].min() for s in range(0, nsub, B // G)]):.0f} us; "
          f"group start spread median {np.median([m_start[s:s + B // G].max() - m_start[s:s + B // G].min() for s in range(0, nsub, B // G)]):.0f} us")
gs = B // G
print("groups (start .. end, ms):", "  ".join(f"[{m_start[s:s + gs].min() / 1e3:.1f} .. {m_end[s:s + gs].max() / 1e3:.1f}]" for s in range(0, min(nsub, 8 * gs), gs)))
dg = waits[diag].reshape(nsub, -1)
print("DIAG wait by block row (mean us):", " ".join(f"{dg[:, q].mean():.0f}" for q in range(dg.shape[1])))
nsl = 40 * max(1, steps // 2)
edges = np.linspace(0, span, nsl + 1)
fl0, fl3, k_end = log[:, :, 0].ravel(), log[:, :, 3].ravel(), np.where(log[:, :, 5] > 0, log[:, :, 5], log[:, :, 1]).ravel()
fl2 = log[:, :, 2].ravel()
upd = (2.0 * 128 ** 3 * (pb - pa)).ravel()
offr = off.ravel()
occ, rate = [], []
for e0, e1 in zip(edges[:-1], edges[1:]):
    occ.append((np.minimum(fl3, e1) - np.maximum(fl0, e0)).clip(0).sum() / (e1 - e0))
    ov = (np.minimum(k_end, e1) - np.maximum(fl0, e0)).clip(0)
    f = (upd * ov / (k_end - fl0).clip(1e-3)).sum()
    ov2 = (np.minimum(fl3, e1) - np.maximum(fl2, e0)).clip(0)
    f += (np.where(offr, 128.0 ** 3 * 1.125, 0.0) * ov2 / (fl3 - fl2).clip(1e-3)).sum()
    rate.append(f / ((e1 - e0) * 1e-6) / 1e12)
print(f"tasks in flight per slice of {span / nsl / 1e3:.2f} ms:", " ".join(f"{o:.0f}" for o in occ))
print(f"TFLOP/s delivered per slice of {span / nsl / 1e3:.2f} ms:", " ".join(f"{r:.0f}" for r in rate))
