"""Phase stamps of the fused diagonal tasks (matrix 0, block rows Q0..Q1, default 18..26): python tools/diag_phases.py CFG B [Q0 Q1]"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from psoap_amd import synthetic as syn
from psoap_amd.chunk import ChunkHandle
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
Q0 = int(sys.argv[3]) if len(sys.argv) > 3 else 18
Q1 = int(sys.argv[4]) if len(sys.argv) > 4 else 26
ch = syn.make_config_chunk(cfg)
gps = syn.make_walkers(ch.n_components, B, seed=1); lw = np.repeat(ch.lwls[None], B, axis=0)
task_dt = np.dtype([("type", "u1"), ("q", "u1"), ("j", "u1"), ("S", "u1"), ("b", "<u2"), ("pa", "u1"), ("pb", "u1"), ("slot", "<u4"), ("ctr", "<u4")])
with ChunkHandle(ch.fl, ch.sigma, max_batch=B) as h:
    h._L.psoap_chunk_dag_tasklog(h._h, None, 0)
    h.lnlike_batch(lw, gps); h.lnlike_batch(lw, gps)
    n = ctypes.c_longlong(0)
    h._L.psoap_chunk_dag_tasks(h._h, None, 0, ctypes.byref(n)); nt = n.value
    tasks = np.zeros(nt, dtype=task_dt)
    h._L.psoap_chunk_dag_tasks(h._h, tasks.ctypes.data_as(ctypes.c_void_p), nt, ctypes.byref(n))
    log = np.zeros(nt * 8, dtype=np.uint64)
    h._L.psoap_chunk_dag_tasklog(h._h, log.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), nt)
log = log.reshape(nt, 8).astype(np.float64) / 100.0
d = np.where(((tasks["type"] & 0x0F) == 1) & (tasks["q"] >= Q0) & (tasks["q"] <= Q1) & (tasks["b"] == 0))[0]
d = d[np.argsort(tasks["q"][d])]
base = log[d, 4].min()
prev_end = None
for i in d:
    st = log[i] - base
    gap = (st[4] - prev_end) if prev_end is not None else 0.0
    print(f"q={tasks['q'][i]} dep-ready {st[4]:7.1f} (gap after previous end {gap:5.1f}) | update {st[5]-st[4]:5.1f} | 8 steps {st[6]-st[5]:5.1f} | outputs {st[1]-st[6]:4.1f} | solve+publish {st[3]-st[1]:5.1f} | total {st[3]-st[4]:5.1f}")
    prev_end = st[3]
print("block-row period %.1f us" % ((log[d[-1], 3] - log[d[0], 3]) / (len(d) - 1)))
