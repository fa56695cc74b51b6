"""Per-task timeline of the persistent DAG kernel (debug/analysis aid)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from psoap_amd import synthetic as syn
from psoap_amd.chunk import ChunkHandle
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
ch = syn.make_config_chunk(cfg)
c = ch.n_components
gps = syn.make_walkers(c, B, seed=1)
lw = np.repeat(ch.lwls[None], B, axis=0)
with ChunkHandle(ch.fl, ch.sigma, max_batch=B) as h:
    P = (ch.N + 127) // 128
    ntask = B * P * (P + 1) // 2
    h._L.psoap_chunk_dag_tasklog(h._h, None, 0)
    h.lnlike_batch(lw, gps)
    h.lnlike_batch(lw, gps)
    log = np.zeros(ntask * 4, dtype=np.uint64)
    h._L.psoap_chunk_dag_tasklog(h._h, log.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), ntask)
log = log.reshape(ntask, 4).astype(np.float64) / 100.0   # us
base = log[:, 0].min()
log -= base
span = log[:, 3].max()
# task metadata
q_of = np.empty(ntask, int); diag = np.zeros(ntask, bool)
t = 0
for q in range(P):
    n = B * (P - q)
    q_of[t:t + n] = q
    diag[t:t + B] = True
    t += n
d01 = log[:, 1] - log[:, 0]; d12 = log[:, 2] - log[:, 1]; d23 = log[:, 3] - log[:, 2]
busy = (log[:, 3] - log[:, 0]).sum()
print(f"N={ch.N} B={B} tasks={ntask} span={span/1e3:.2f} ms  sum(task time)={busy/1e3:.1f} ms  -> avg concurrency {busy/span:.1f} of 512")
print(f"OFF : update+store {d01[~diag].sum()/1e3:8.1f} ms | wait potrf {d12[~diag].sum()/1e3:8.1f} ms | trsm+publish {d23[~diag].sum()/1e3:8.1f} ms")
print(f"DIAG: update+store {d01[diag].sum()/1e3:8.1f} ms | potrf      {d12[diag].sum()/1e3:8.1f} ms | publish      {d23[diag].sum()/1e3:8.1f} ms")
print(f"potrf median {np.median(d12[diag]):.1f} us; OFF trsm median {np.median(d23[~diag]):.1f} us; OFF wait-potrf median {np.median(d12[~diag]):.1f} us, mean {d12[~diag].mean():.1f}")
# ideal K-loop time per task at 64 cyc/MFMA: chunks * 64 MFMA * 64 cyc / clock
for q in (1, 5, 10, 20, 30, 40, 46):
    if q < P:
        m = (q_of == q) & ~diag
        if m.any():
            print(f"  q={q:2d}: OFF update+store median {np.median(d01[m]):7.1f} us (K={128*q}: {128*q/16*64*64/2.4e3:7.1f} us at 2.4 GHz MFMA peak), row start {log[m,0].min()/1e3:.2f} ms end {log[m,3].max()/1e3:.2f} ms")
# occupancy over time
edges = np.linspace(0, span, 41)
occ = [(np.minimum(log[:, 3], e1) - np.maximum(log[:, 0], e0)).clip(0).sum() / (e1 - e0) for e0, e1 in zip(edges[:-1], edges[1:])]
print("tasks in flight per 2.5% time slice:", " ".join(f"{o:.0f}" for o in occ))
