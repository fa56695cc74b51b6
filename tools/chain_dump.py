"""The PART chain of one tile next to the tasks it waits for (debug aid for the latency schemes):
    python tools/chain_dump.py cfg B q j        -- stamps in us from the start of the launch"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from psoap_amd import synthetic as syn
from psoap_amd.chunk import ChunkHandle
cfg, B, q, j = (int(x) for x in sys.argv[1:5])
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
us = lambda x: (x - t0) / 100.0
names = {0: "PART", 1: "DIAG", 2: "OFF ", 3: "SCHUR"}
rows_end = {}
for r in range(max(0, q - 4), q + 1):
    idx = np.where((tasks["q"] == r) & (tasks["b"] == 0) & (ty != 0))[0]
    rows_end[r] = us(raw[idx, 3].max())
print("last end of the finals of rows:", {r: round(v) for r, v in rows_end.items()})
idx = np.where((tasks["q"] == q) & (tasks["j"] == j) & (tasks["b"] == 0))[0]
for i in idx:
    t = tasks[i]
    st = raw[i]
    print(f"ticket {i:6d} {names[int(ty[i])]} S={t['S']:2d} panels [{t['pa']:2d},{t['pb']:2d}) start {us(st[0]):7.0f} s1 {us(st[1]):7.0f} s2 {us(st[2]):7.0f} dep {us(st[4]):7.0f} "
          f"gemm-end {us(st[5]):7.0f} stamp6 {us(st[6]):7.0f} end {us(st[3]):7.0f}   xcd {int(log.reshape(nt, 8)[i, 7] >> 32)}")
