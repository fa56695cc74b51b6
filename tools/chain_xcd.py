"""Single evaluation: does a block row take longer when its diagonal task runs on another XCD than the one before it, or than
the strip solve of the tile above it?  Per block row: period, XCD of the diagonal task, of the strip solve of tile (q-1, q).
    python tools/chain_xcd.py cfg [reps]"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from psoap_amd import synthetic as syn
from psoap_amd.chunk import ChunkHandle
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
ch = syn.make_config_chunk(cfg)
gps = syn.make_walkers(ch.n_components, 1, seed=1)
lw = ch.lwls[None]
task_dt = np.dtype([("type", "u1"), ("q", "u1"), ("j", "u1"), ("S", "u1"), ("b", "<u2"), ("pa", "u1"), ("pb", "u1"),
                    ("slot", "<u4"), ("ctr", "<u4")])
same, hop = [], []
with ChunkHandle(ch.fl, ch.sigma, max_batch=1) as h:
    h._L.psoap_chunk_dag_tasklog(h._h, None, 0)
    h.lnlike_batch(lw, gps)
    n = ctypes.c_longlong(0)
    h._L.psoap_chunk_dag_tasks(h._h, None, 0, ctypes.byref(n)); nt = n.value
    tasks = np.zeros(nt, dtype=task_dt)
    h._L.psoap_chunk_dag_tasks(h._h, tasks.ctypes.data_as(ctypes.c_void_p), nt, ctypes.byref(n))
    ty = tasks["type"] & 0x0F
    P = int(tasks["q"].max()) + 1
    for rep in range(reps):
        h.lnlike_batch(lw, gps)
        log = np.zeros(nt * 8, dtype=np.uint64)
        h._L.psoap_chunk_dag_tasklog(h._h, log.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), nt)
        log = log.reshape(nt, 8)
        end = log[:, 3].astype(np.float64) / 100.0
        xcd = (log[:, 7] >> np.uint64(32)).astype(int)
        d_end, d_x, o_x = np.zeros(P), np.zeros(P, int), np.full(P, -1)
        for q in range(P):
            i = np.where((ty == 1) & (tasks["q"] == q))[0][-1]
            d_end[q], d_x[q] = end[i], xcd[i]
            if q > 0:
                k = np.where((ty == 2) & (tasks["q"] == q - 1) & (tasks["j"] == q))[0]
                if len(k):
                    o_x[q] = xcd[k[-1]]
        per = np.diff(d_end)
        for q in range(2, P):
            (same if (d_x[q] == d_x[q - 1] and (o_x[q] < 0 or o_x[q] == d_x[q])) else hop).append(per[q - 1])
        if rep == 0:
            print("row period us :", " ".join(f"{p:3.0f}" for p in per))
            print("diag task xcd :", " ".join(f"{x:3d}" for x in d_x))
            print("strip(q-1,q)  :", " ".join(f"{x:3d}" for x in o_x))
print(f"rows whose diagonal task, its predecessor and the strip solve above ran on ONE xcd: {len(same)}, period mean {np.mean(same) if same else 0:.1f} median {np.median(same) if same else 0:.1f} us")
print(f"rows with a hop between XCDs on the chain: {len(hop)}, period mean {np.mean(hop):.1f} median {np.median(hop):.1f} us")
