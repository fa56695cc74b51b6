import sys, os, ctypes
sys.path.insert(0, '/root/repo')
import numpy as np
from psoap_amd import synthetic as syn
from psoap_amd.chunk import ChunkHandle
ch = syn.make_config_chunk(3); B = 1
gps = syn.make_walkers(2, B, seed=1); lw = np.repeat(ch.lwls[None], B, axis=0)
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
d = np.where(((tasks["type"] & 0x0F) == 1) & (tasks["q"] >= 18) & (tasks["q"] <= 24))[0]
base = log[d, 4].min()
for i in d:
    st = log[i] - base
    print(f"q={tasks['q'][i]} dep {st[4]:7.1f} | steps-end +{st[6]-st[4]:5.1f} | potrf-out +{st[1]-st[6]:4.1f} | drained+pub +{st[2]-st[1]:4.1f} | off1-waited +{st[0]-st[2]:4.1f} | trsm +{st[7]-st[0]:5.1f} | drain +{st[5]-st[7]:4.1f} | release+pub +{st[3]-st[5]:4.1f} | total {st[3]-st[4]:5.1f}")
