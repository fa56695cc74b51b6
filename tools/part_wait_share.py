"""How much of the time PART tasks (and finals) hold a workgroup is spent in dependency waits?  The in-kernel stamps add up
the two waits of a task's update (dag_update: bits 40.. of log word 7) and mark the end of the wait for the predecessor's tile
(word 6 - word 5).     python tools/part_wait_share.py cfg B"""
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
log = log.reshape(nt, 8)
ty = tasks["type"] & 0x0F
start, end = log[:, 0].astype(np.float64), log[:, 3].astype(np.float64)
ok = (end > start) & (start > 0)
upd_wait = (log[:, 7] >> np.uint64(40)).astype(np.float64)           # ticks (10 ns) in the update's two waits
pred_wait = np.where((log[:, 6] > log[:, 5]) & (log[:, 5] > 0), (log[:, 6] - log[:, 5]).astype(np.float64), 0.0)
span = (end[ok].max() - start[ok].min()) / 100.0
for name, sel in (("PART", ok & (ty == 0)), ("finals (in-kernel path)", ok & (ty != 0))):
    hold = (end[sel] - start[sel]).sum() / 100.0
    print(f"N={ch.N} B={B} {name}: {sel.sum()} tasks hold workgroups for {hold:.0f} us in all (span {span:.0f} us): "
          f"{100 * upd_wait[sel].sum() / 100.0 / hold:.1f} % in the update's waits for block rows, "
          f"{100 * pred_wait[sel].sum() / 100.0 / hold:.1f} % waiting for the predecessor's / the parts' tiles")
