"""Do the two workgroups that share a compute unit run their non-MFMA phases (epilogue + strip solve) at the same
time?  Per-task stamps + HW_REG_HW_ID of the persistent kernel.   python tools/dag_cu_overlap.py CFG B"""
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
with ChunkHandle(ch.fl, ch.sigma, max_batch=B) as h:
    h._L.psoap_chunk_dag_tasklog(h._h, None, 0)
    h.lnlike_batch(lw, gps); h.lnlike_batch(lw, gps)
    n = ctypes.c_longlong(0)
    h._L.psoap_chunk_dag_tasks(h._h, None, 0, ctypes.byref(n)); nt = n.value
    log = np.zeros(nt * 8, dtype=np.uint64)
    h._L.psoap_chunk_dag_tasklog(h._h, log.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), nt)
log = log.reshape(nt, 8)
hw = log[:, 7]
xcc = (hw >> np.uint64(32)).astype(np.int64)
hwid = (hw & np.uint64(0xffffffff)).astype(np.int64)
cu = (hwid >> 8) & 0xf; sh = (hwid >> 12) & 1; se = (hwid >> 13) & 7
cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
t = log[:, [0, 5, 3]].astype(np.float64) / 100.0     # start, K-loop end, task end (us)
ok = (log[:, 5] > 0) & (log[:, 3] > 0)
t0 = t[ok, 0].min()
print(f"N={ch.N} B={B}: {nt} tasks on {len(np.unique(cuid))} distinct CU ids")
tot_valu = tot_both = tot_span = 0.0
for cid in np.unique(cuid):
    m = ok & (cuid == cid)
    if m.sum() < 4: continue
    a, b = t[m, 1] - t0, t[m, 2] - t0            # non-MFMA phase intervals of all tasks on this CU
    lo, hi = t[m, 0].min() - t0, b.max()
    grid = np.arange(lo, hi, 1.0)                 # 1 us resolution
    depth = np.zeros(len(grid))
    for x, y in zip(a, b):
        depth[int(max(0, x - lo)):int(max(0, y - lo))] += 1
    tot_span += hi - lo
    tot_valu += (depth >= 1).sum()
    tot_both += (depth >= 2).sum()
f1 = tot_valu / tot_span; f2 = tot_both / tot_span
# if the two workgroups were independent with each in its non-MFMA phase a fraction p of the time: f1 = 2p - p^2, f2 = p^2
p = 1 - np.sqrt(1 - f1 - f2 + f2) if False else (f1 + f2) / 2
print(f"time with at least one workgroup of the CU outside its K-loop: {100*f1:.1f} %;  with both: {100*f2:.1f} %")
print(f"per-workgroup share outside the K-loop p = {100*p:.1f} %  ->  independent phases would give both-outside {100*p*p:.1f} %, "
      f"locked phases {100*p:.1f} %")
if "--fields" in sys.argv:
    for name, sh_, w in (("wave", 0, 4), ("simd", 4, 2), ("pipe", 6, 2), ("cu", 8, 4), ("sh", 12, 1), ("se", 13, 3), ("tg", 16, 4),
                         ("vm", 20, 4), ("queue", 24, 3), ("state", 27, 3), ("me", 30, 2)):
        v = (hwid >> sh_) & ((1 << w) - 1)
        print(name, np.unique(v, return_counts=True))
    print("xcc", np.unique(xcc, return_counts=True))

# K-loop speed against the neighbour's state: microseconds per 16-row stage of a task's update as a function of the
# fraction of that interval during which the other workgroup of the CU was in a K-loop as well
task_dt = np.dtype([("type", "u1"), ("q", "u1"), ("j", "u1"), ("S", "u1"), ("b", "<u2"), ("pa", "u1"), ("pb", "u1"),
                    ("slot", "<u4"), ("ctr", "<u4")])
with ChunkHandle(ch.fl, ch.sigma, max_batch=B) as h2:
    h2.upload(lw, gps); h2.eval(); h2.fetch()
    n = ctypes.c_longlong(0)
    h2._L.psoap_chunk_dag_tasks(h2._h, None, 0, ctypes.byref(n))
    tasks = np.zeros(n.value, dtype=task_dt)
    h2._L.psoap_chunk_dag_tasks(h2._h, tasks.ctypes.data_as(ctypes.c_void_p), n.value, ctypes.byref(n))
if len(tasks) == nt:
    stages = 8.0 * (tasks["pb"].astype(float) - tasks["pa"].astype(float))
    slot = hwid & 0xf                                    # wave slot of wave 0: tells the two workgroups of a CU apart
    fr, us = [], []
    for cid in np.unique(cuid):
        m = np.where(ok & (cuid == cid))[0]
        if len(m) < 4: continue
        for s_ in np.unique(slot[m]):
            mine = m[slot[m] == s_]; other = m[slot[m] != s_]
            if len(other) == 0: continue
            oa, ob = t[other, 0], t[other, 1]            # the neighbour's K-loop intervals
            for i in mine:
                if stages[i] < 80: continue              # long K-loops only
                a0, b0 = t[i, 0], t[i, 1]
                ov = (np.minimum(ob, b0) - np.maximum(oa, a0)).clip(0).sum()
                fr.append(ov / (b0 - a0)); us.append((b0 - a0) / stages[i])
    fr, us = np.array(fr), np.array(us)
    A = np.vstack([np.ones_like(fr), fr]).T
    coef = np.linalg.lstsq(A, us, rcond=None)[0]
    print(f"{len(fr)} long K-loops: mean neighbour-in-K-loop fraction {fr.mean():.2f}; us per stage = {coef[0]:.2f} + {coef[1]:.2f} x fraction "
          f"(alone {coef[0]:.2f}, shared {coef[0]+coef[1]:.2f}; 100 % of the MFMA peak alone would be 1.74 at 2.35 GHz)")
    for lo_, hi_ in ((0, .5), (.5, .7), (.7, .8), (.8, .9), (.9, 1.01)):
        mm = (fr >= lo_) & (fr < hi_)
        if mm.any(): print(f"   fraction {lo_:.1f}-{hi_:.1f}: {mm.sum():6d} K-loops, {us[mm].mean():.2f} us per stage")
