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
task_dt = np.dtype([("type", "u1"), ("q", "u1"), ("j", "u1"), ("S", "u1"), ("b", "<u2"), ("pa", "u1"), ("pb", "u1"),
                    ("slot", "<u4"), ("ctr", "<u4")])
with ChunkHandle(ch.fl, ch.sigma, max_batch=B) as h:
    P = (ch.N + 127) // 128
    h._L.psoap_chunk_dag_tasklog(h._h, None, 0)
    h.lnlike_batch(lw, gps)
    h.lnlike_batch(lw, gps)
    n = ctypes.c_longlong(0)
    h._L.psoap_chunk_dag_tasks(h._h, None, 0, ctypes.byref(n))
    ntask = n.value
    tasks = np.zeros(ntask, dtype=task_dt)
    h._L.psoap_chunk_dag_tasks(h._h, tasks.ctypes.data_as(ctypes.c_void_p), ntask, ctypes.byref(n))
    log = np.zeros(ntask * 8, dtype=np.uint64)
    h._L.psoap_chunk_dag_tasklog(h._h, log.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), ntask)
tasks["type"] &= 0x0F   # strip the chain flag
waits = (log.reshape(ntask, 8)[:, 7] >> np.uint64(40)).astype(np.float64) / 100.0     # us in the updates' dependency waits
log = log.reshape(ntask, 8).astype(np.float64) / 100.0   # us
base = log[:, 0].min()
log -= base
span = log[:, 3].max()
part = tasks["type"] == 0; diag = tasks["type"] == 1; off = tasks["type"] == 2
q_of = tasks["q"].astype(int)
d01 = log[:, 1] - log[:, 0]; d12 = log[:, 2] - log[:, 1]; d23 = log[:, 3] - log[:, 2]
busy = (log[:, 3] - log[:, 0]).sum()
print(f"N={ch.N} B={B} tasks={ntask} (PART {part.sum()}, DIAG {diag.sum()}, OFF {off.sum()}) span={span/1e3:.2f} ms  "
      f"sum(task time)={busy/1e3:.1f} ms  -> avg concurrency {busy/span:.1f}")
print(f"measured dependency waits inside updates: {waits.sum()/1e3:.1f} ms ({100*waits.sum()/busy:.1f} % of all task time)")
print(f"PART: total {(log[part,3]-log[part,0]).sum()/1e3:8.1f} ms")
print(f"OFF : update+store {d01[off].sum()/1e3:8.1f} ms | wait potrf {d12[off].sum()/1e3:8.1f} ms | trsm+publish {d23[off].sum()/1e3:8.1f} ms")
print(f"DIAG: update+store {d01[diag].sum()/1e3:8.1f} ms | potrf      {d12[diag].sum()/1e3:8.1f} ms | publish      {d23[diag].sum()/1e3:8.1f} ms")
print(f"potrf median {np.median(d12[diag]):.1f} us; OFF trsm median {np.median(d23[off]):.1f} us; OFF wait-potrf mean {d12[off].mean():.1f} us; DIAG update+store median {np.median(d01[diag]):.1f} us")
for q in (1, 5, 10, 20, 30, 40, 46):
    if q < P:
        m = (q_of == q) & off
        if m.any():
            print(f"  q={q:2d}: OFF update+store median {np.median(d01[m]):7.1f} us, S={tasks['S'][m][0]}, row start {log[m,0].min()/1e3:.2f} ms end {log[m,3].max()/1e3:.2f} ms")
edges = np.linspace(0, span, 41)
occ = [(np.minimum(log[:, 3], e1) - np.maximum(log[:, 0], e0)).clip(0).sum() / (e1 - e0) for e0, e1 in zip(edges[:-1], edges[1:])]
print("tasks in flight per 2.5% time slice:", " ".join(f"{o:.0f}" for o in occ))
# MFMA work delivered per time slice (a task's update flops spread evenly over its K-loop interval [0]..[5],
# its strip solve over [2]..[3]): where in the launch the rate falls below the steady state
upd = 2.0 * 128 * 128 * 128 * (tasks["pb"].astype(float) - tasks["pa"].astype(float))
have5 = log[:, 5] > 0
k_end = np.where(have5, log[:, 5], log[:, 1])
rate = []
for e0, e1 in zip(edges[:-1], edges[1:]):
    ov = (np.minimum(k_end, e1) - np.maximum(log[:, 0], e0)).clip(0)
    dur = (k_end - log[:, 0]).clip(1e-3)
    f = (upd * ov / dur).sum()
    ov2 = (np.minimum(log[:, 3], e1) - np.maximum(log[:, 2], e0)).clip(0)
    dur2 = (log[:, 3] - log[:, 2]).clip(1e-3)
    f += (np.where(off, 128.0 ** 3 * 1.125, 0.0) * ov2 / dur2).sum()
    rate.append(f / ((e1 - e0) * 1e-6) / 1e12)
print("TFLOP/s delivered per 2.5% time slice:", " ".join(f"{r:.0f}" for r in rate))
