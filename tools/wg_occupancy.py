"""What the persistent workgroups do over time (debug aid for the latency schemes): per time bucket, the average number
of workgroups that hold a task but wait for its dependencies (start -> dep stamp), that work on one (dep -> end), and
that hold none; split by task kind.      python tools/wg_occupancy.py cfg B [bucket_us]"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from psoap_amd import synthetic as syn
from psoap_amd.chunk import ChunkHandle
cfg, B = int(sys.argv[1]), int(sys.argv[2])
bucket = float(sys.argv[3]) if len(sys.argv) > 3 else 100.0
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
start, dep, end = (raw[:, 0] - t0) / 100.0, (raw[:, 4] - t0) / 100.0, (raw[:, 3] - t0) / 100.0
dep = np.clip(dep, start, end)          # tasks without a dep stamp count as working
span = end.max()
nb = int(np.ceil(span / bucket))
def cover(a, b, sel):
    out = np.zeros(nb)
    for x, y in zip(a[sel], b[sel]):
        i0, i1 = int(x // bucket), int(min(y, span - 1e-9) // bucket)
        for i in range(i0, i1 + 1):
            out[i] += max(0.0, min(y, (i + 1) * bucket) - max(x, i * bucket))
    return out / bucket
part, fin = ty == 0, ty != 0
print(f"span {span:.0f} us, {nt} tasks; columns: bucket start | PART waiting, working | finals waiting, working | rows finished")
rows_end = np.array([end[(tasks["q"] == q) & fin & (tasks["b"] == 0)].max() for q in range(int(tasks["q"].max()) + 1)])
pw, pk, fw, fk = cover(start, dep, part), cover(dep, end, part), cover(start, dep, fin), cover(dep, end, fin)
for i in range(nb):
    print(f"{i * bucket:6.0f} | {pw[i]:6.1f} {pk[i]:6.1f} | {fw[i]:6.1f} {fk[i]:6.1f} | {int((rows_end <= (i + 1) * bucket).sum()):3d}")
