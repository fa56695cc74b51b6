"""Where the time of predict's persistent launch goes (N = 8192, M = 1024: the retrieve shape): per-task stamps written by the
library when PSOAP_PREDICT_TLOG names a file.      python tools/predict_timeline.py [bucket_us]"""
import sys, os, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
path = os.path.join(tempfile.gettempdir(), "psoap_predict_tlog.bin")
os.environ["PSOAP_PREDICT_TLOG"] = path
from psoap_amd import synthetic as syn
from psoap_amd.chunk import ChunkHandle
bucket = float(sys.argv[1]) if len(sys.argv) > 1 else 500.0
ch = syn.make_config_chunk(5)
M = 2 * ch.n_pix
pred = np.linspace(ch.lwls[0].min(), ch.lwls[0].max(), M)
with ChunkHandle(ch.fl, ch.sigma, max_batch=1) as h:
    for _ in range(3):
        h.predict(0, ch.lwls, np.stack([pred] * 3), np.zeros(3), syn.GP_BASE[3])
    t = h.predict_timings()
raw = open(path, "rb").read()
nt, P, Mt, scheme = np.frombuffer(raw[:32], dtype=np.uint64).astype(int)
task_dt = np.dtype([("type", "u1"), ("q", "u1"), ("j", "u1"), ("S", "u1"), ("b", "<u2"), ("pa", "u1"), ("pb", "u1"),
                    ("slot", "<u4"), ("ctr", "<u4")])
tasks = np.frombuffer(raw[32:32 + 16 * nt], dtype=task_dt)
log = np.frombuffer(raw[32 + 16 * nt:], dtype=np.uint64).reshape(nt, 8)
ty = tasks["type"] & 0x0F
start, end = log[:, 0].astype(np.float64) / 100.0, log[:, 3].astype(np.float64) / 100.0
ok = (end > start) & (start > 0)
t0 = start[ok].min()
start, end = start - t0, end - t0
span = end[ok].max()
hwid = (log[:, 7] & np.uint64(0xffffffff)).astype(np.int64)
print(f"predict N={ch.N} R={3 * M}: device {t['device_ms']:.2f} ms, launch span {span / 1e3:.2f} ms, {nt} tasks (scheme {scheme}), "
      f"{len(np.unique(((log[:, 7] >> np.uint64(32)).astype(np.int64) * 65536 + ((hwid >> 8) & 0xff))[ok]))} compute units seen")
data = tasks["j"] < P          # tiles of B;   appended columns: j >= P, q < P;   Schur: q >= P
kinds = (("B: diagonal finals", ok & (ty == 1)), ("B: off-diagonal finals", ok & (ty == 2) & data),
         ("appended-column finals", ok & (ty == 2) & ~data), ("Schur finals", ok & (ty == 3)),
         ("PARTs of B tiles", ok & (ty == 0) & data & (tasks["q"] < P)), ("PARTs of appended columns", ok & (ty == 0) & ~data & (tasks["q"] < P)),
         ("PARTs of Schur tiles", ok & (ty == 0) & (tasks["q"] >= P)))
upd_wait = (log[:, 7] >> np.uint64(40)).astype(np.float64) / 100.0
tot = 0.0
for name, sel in kinds:
    hold = (end[sel] - start[sel]).sum()
    tot += hold
    print(f"  {name:28s} {int(sel.sum()):6d} tasks, {hold / 1e3:8.1f} ms of workgroup time, mean {hold / max(1, sel.sum()):6.1f} us, "
          f"{100 * upd_wait[sel].sum() / max(hold, 1):4.1f} % in the update's row waits")
print(f"  all: {tot / 1e3:.1f} ms = {tot / span:.0f} workgroups busy on average")
# phases of the PART tasks (stamps: 5 = update done, 6 = predecessor's tile waited for, 3 = stored and published)
k5, k6 = log[:, 5].astype(np.float64) / 100.0 - t0, log[:, 6].astype(np.float64) / 100.0 - t0
parts = ok & (ty == 0) & (log[:, 5] > 0) & (log[:, 6] > 0)
panels = (tasks["pb"].astype(int) - tasks["pa"].astype(int))
for name, sel in (("PARTs of B tiles", parts & data & (tasks["q"] < P)), ("PARTs of appended columns", parts & ~data & (tasks["q"] < P)),
                  ("PARTs of Schur tiles", parts & (tasks["q"] >= P))):
    n = max(1, int(sel.sum()))
    kl, fold, st = (k5[sel] - start[sel]).sum() / n, (k6[sel] - k5[sel]).sum() / n, (end[sel] - k6[sel]).sum() / n
    pn = panels[sel].mean() if sel.any() else 0.0
    print(f"  {name:28s} mean {pn:4.1f} panels: update {kl:6.1f} us ({kl / max(pn, 1e-9):5.1f} us per panel; 27.3 at this workgroup's share "
          f"of the peak), predecessor's tile {fold:5.1f} us, fold + store + publish {st:5.1f} us")
# the hand-over in four pieces (stamps 1 = predecessor's tile folded in, 2 = stores issued, 4 = stores drained + barrier)
k1, k2, k4 = (log[:, i].astype(np.float64) / 100.0 - t0 for i in (1, 2, 4))
sub = parts & (log[:, 1] > 0) & (log[:, 2] > 0) & (log[:, 4] > 0)
for name, sel in (("PARTs with a predecessor", sub & (tasks["S"] > 0)), ("first PARTs of a chain", sub & (tasks["S"] == 0))):
    n = max(1, int(sel.sum()))
    print(f"  {name:28s} {int(sel.sum()):6d}: fold {(k1[sel] - k6[sel]).sum() / n:5.1f} us, stores issued {(k2[sel] - k1[sel]).sum() / n:5.1f}, "
          f"drained {(k4[sel] - k2[sel]).sum() / n:5.1f}, release + arrive {(end[sel] - k4[sel]).sum() / n:5.1f}")
dend = np.array([end[ok & (ty == 1) & (tasks["q"] == q)].max() for q in range(P)])
print("diag end (us):", " ".join(f"{x:.0f}" for x in dend))
print("row period  :", " ".join(f"{b - a:.0f}" for a, b in zip(dend[:-1], dend[1:])))
nb = int(np.ceil(span / bucket))
print("bucket | busy workgroups by kind:", ", ".join(k for k, _ in kinds))
for i in range(nb):
    lo, hi = i * bucket, (i + 1) * bucket
    row = []
    for _, sel in kinds:
        ov = np.clip(np.minimum(end[sel], hi) - np.maximum(start[sel], lo), 0, None).sum() / bucket
        row.append(f"{ov:6.1f}")
    print(f"{lo:7.0f} | " + " ".join(row))
