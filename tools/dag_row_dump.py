"""Per-task stamps of a few block rows of the persistent DAG kernel (debug aid): python tools/dag_row_dump.py cfg B q0 q1"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from psoap_amd import synthetic as syn
from psoap_amd.chunk import ChunkHandle
cfg, B, q0, q1 = (int(x) for x in sys.argv[1:5])
ch = syn.make_config_chunk(cfg)
c = ch.n_components
gps = syn.make_walkers(c, B, seed=1)
lw = np.repeat(ch.lwls[None], B, axis=0)
task_dt = np.dtype([("type", "u1"), ("q", "u1"), ("j", "u1"), ("S", "u1"), ("b", "<u2"), ("pa", "u1"), ("pb", "u1"),
                    ("slot", "<u4"), ("ctr", "<u4")])
with ChunkHandle(ch.fl, ch.sigma, max_batch=B) as h:
    h._L.psoap_chunk_dag_tasklog(h._h, None, 0)
    h.lnlike_batch(lw, gps); h.lnlike_batch(lw, gps)
    n = ctypes.c_longlong(0)
    h._L.psoap_chunk_dag_tasks(h._h, None, 0, ctypes.byref(n)); nt = n.value
    tasks = np.zeros(nt, dtype=task_dt)
    h._L.psoap_chunk_dag_tasks(h._h, tasks.ctypes.data_as(ctypes.c_void_p), nt, ctypes.byref(n))
    log = np.zeros(nt * 8, dtype=np.uint64)
    h._L.psoap_chunk_dag_tasklog(h._h, log.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), nt)
tasks["type"] &= 0x0F   # strip the chain flag
raw = log.reshape(nt, 8).astype(np.float64)
log = raw / 100.0
log[:, 7] = 0.0
log -= log[:, 0].min()
dd = np.where((tasks["type"] & 0x0F) == 1)[0]
dd = dd[np.argsort(raw[dd, 0])]
names = {0: "PART", 1: "DIAG", 2: "OFF "}
print(f"span {log[:,3].max()/1e3:.2f} ms, tasks {nt}")
for q in range(q0, q1 + 1):
    idx = np.where((tasks["q"] == q) & (tasks["b"] == 0))[0]
    idx = idx[np.argsort(log[idx, 3])]
    print(f"--- row {q}: {len(idx)} tasks; first start {log[idx,0].min():.0f} us, last end {log[idx,3].max():.0f} us")
    for i in idx[-6:]:
        t = tasks[i]
        st = log[i]
        print(f"  ticket {i:6d} {names[int(t['type'])]} j={t['j']:2d} S={t['S']} panels [{t['pa']:2d},{t['pb']:2d})  start {st[0]:8.0f}  dep {st[4]:8.0f} gemm {st[5]:8.0f} chain {st[6]:8.0f} upd {st[1]:8.0f}  +wait/potrf {st[2]-st[1]:6.0f}  +trsm {st[3]-st[2]:6.0f}  end {st[3]:8.0f}")
    d = idx[tasks["type"][idx] == 1]
    for i in d:
        st = log[i]; t = tasks[i]
        print(f"  DIAG ticket {i} panels [{t['pa']},{t['pb']}) start {st[0]:.0f} dep-ready {st[4]:.0f} gemm-end {st[5]:.0f} chain-ready/steps-end {st[6]:.0f} upd-end/potrf-out {st[1]:.0f} potrf-end(drained) {st[2]:.0f} end {st[3]:.0f}")
