"""Worker-time split of the persistent kernel from its per-task stamps: K-loops (incl. dependency waits inside the
update), partial-sum gather, store routine (epilogue: on-the-fly covariance + stores + drain), strip solve, in-block
Cholesky.   python tools/dag_phase_share.py CFG B"""
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
    h._L.psoap_chunk_dag_tasklog(h._h, None, 0)
    h.lnlike_batch(lw, gps); h.lnlike_batch(lw, gps)
    n = ctypes.c_longlong(0)
    h._L.psoap_chunk_dag_tasks(h._h, None, 0, ctypes.byref(n)); nt = n.value
    tasks = np.zeros(nt, dtype=task_dt)
    h._L.psoap_chunk_dag_tasks(h._h, tasks.ctypes.data_as(ctypes.c_void_p), nt, ctypes.byref(n))
    log = np.zeros(nt * 8, dtype=np.uint64)
    h._L.psoap_chunk_dag_tasklog(h._h, log.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), nt)
ty = tasks["type"] & 0x0F
flags = tasks["type"]
log = log.reshape(nt, 8).astype(np.float64) / 100.0
span = log[:, 3].max() - log[:, 0].min()
ok = (log[:, 5] > 0) & (log[:, 6] > 0) & (log[:, 3] > 0)      # tasks of the generic path (the fused diagonal path stamps differently)
part, diag, off = (ty == 0) & ok, (ty == 1) & ok, (ty == 2) & ok
fin = (diag | off)
tot = (log[:, 3] - log[:, 0])[ok | (ty == 1)].sum()
kloop = (log[:, 5] - log[:, 0])[ok].sum()
gather = (log[:, 6] - log[:, 5])[ok].sum()
store_fin = (log[:, 1] - log[:, 6])[fin].sum()
store_part = (log[:, 3] - log[:, 6])[part].sum()
solve = (log[:, 3] - log[:, 1])[off].sum()
potrf = (log[:, 3] - log[:, 1])[diag].sum()
print(f"N={ch.N} B={B}: span {span/1e3:.2f} ms, {nt} tasks; worker time {tot/1e3:.0f} ms = {tot/span:.0f} workers busy on average")
for name, v in (("update K-loops + dependency waits", kloop), ("wait for / fold partial tiles", gather),
                ("store routine + drain, finals (covariance on the fly)", store_fin), ("store + publish, PARTs", store_part),
                ("wait potrf + strip solve + publish (OFF)", solve), ("in-block Cholesky + publish (DIAG)", potrf)):
    print(f"  {name:58s} {v/1e3:9.1f} ms  {100*v/tot:5.1f} %")
n_fin = fin.sum()
print(f"  per final tile: store routine {store_fin/n_fin:.1f} us, strip solve phase {solve/max(1,off.sum()):.1f} us; per PART: store {store_part/max(1,part.sum()):.1f} us")
# the finals' store phase in two pieces (stamp 4 = the tile's last store issued by wave 0; 1 = drained + barrier)
issued = (log[:, 4] > log[:, 6]) & fin
if issued.any():
    print(f"  finals: covariance + 64 stores issued {(log[:, 4] - log[:, 6])[issued].mean():.1f} us, drain + barrier {(log[:, 1] - log[:, 4])[issued].mean():.1f} us"
          f" (of {int(issued.sum())} tasks)")
