"""End of every block row's last final (us from the start of the launch) and the row-to-row period (debug aid):
    python tools/row_periods.py cfg B"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from psoap_amd import synthetic as syn
from psoap_amd.chunk import ChunkHandle
cfg, B = int(sys.argv[1]), int(sys.argv[2])
ch = syn.make_config_chunk(cfg)
gps = syn.make_walkers(ch.n_components, B, seed=1)
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
ty = tasks["type"] & 0x0F
raw = log.reshape(nt, 8).astype(np.float64)
t0 = raw[:, 0].min()
P = int(tasks["q"].max()) + 1
ends, dends = [], []
for q in range(P):
    idx = np.where((tasks["q"] == q) & (tasks["b"] == 0) & (ty != 0) & (ty != 3))[0]
    ends.append((raw[idx, 3].max() - t0) / 100.0)
    d = np.where((tasks["q"] == q) & (tasks["b"] == 0) & (ty == 1))[0]
    dends.append((raw[d, 3].max() - t0) / 100.0)
print("span %.0f us" % ((raw[:, 3].max() - t0) / 100.0))
print("row end  :", " ".join(f"{e:.0f}" for e in ends))
print("period   :", " ".join(f"{b - a:.0f}" for a, b in zip(ends[:-1], ends[1:])))
print("diag end :", " ".join(f"{e:.0f}" for e in dends))
