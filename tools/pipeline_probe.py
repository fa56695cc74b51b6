"""Two half-ensembles in flight: two handles on the same chunk (16 walkers each, own workspaces and streams), their
persistent launches staggered by half a step, against one 32-walker handle.  python tools/pipeline_probe.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from psoap_amd import synthetic as syn
from psoap_amd.chunk import ChunkHandle

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
ch = syn.make_config_chunk(3)
B = 32
gps = syn.make_walkers(2, B, seed=3500)
lw = syn.walker_lwls(ch, syn.make_walker_velocities(ch, B, seed=3501))
with ChunkHandle(ch.fl, ch.sigma, max_batch=B) as h:
    h.upload(lw, gps); h.eval(); ref = h.fetch()
    t0 = time.perf_counter()
    for _ in range(steps):
        h.eval(); h.upload(lw, gps); h.fetch()
    dt = time.perf_counter() - t0
    print(f"one handle, B=32: {1e3 * dt / steps:.2f} ms per 32 evals, {B * steps / dt:.1f} evals/s")
H = B // 2
a = ChunkHandle(ch.fl, ch.sigma, max_batch=H)
b = ChunkHandle(ch.fl, ch.sigma, max_batch=H)
pa, pb = (lw[:H], gps[:H]), (lw[H:], gps[H:])
a.upload(*pa); b.upload(*pb)
a.eval(); b.eval(); ra = a.fetch(); rb = b.fetch()
assert np.allclose(np.concatenate([ra, rb]), ref, rtol=1e-10, atol=0), "halves differ from the 32-walker batch"
for label, stagger in (("in phase", False), ("staggered", True)):
    a.upload(*pa); b.upload(*pb)
    a.eval()
    if not stagger:
        b.eval()
    t0 = time.perf_counter()
    for _ in range(steps):
        if stagger:
            b.eval()                       # B starts while A is in flight
            ra = a.fetch(); a.upload(*pa); a.eval()
            rb = b.fetch(); b.upload(*pb)
        else:
            ra = a.fetch(); rb = b.fetch()
            a.upload(*pa); b.upload(*pb); a.eval(); b.eval()
    a.fetch()
    if not stagger:
        b.fetch()
    dt = time.perf_counter() - t0
    print(f"two handles of 16, {label}: {1e3 * dt / steps:.2f} ms per 32 evals, {B * steps / dt:.1f} evals/s")
a.close(); b.close()

# two FULL 32-walker steps in flight (two handles, each with its own 9.3 GB of matrices): step k+1 is enqueued before
# step k is fetched, so the tail of k and the first block rows of k+1 share the device
a = ChunkHandle(ch.fl, ch.sigma, max_batch=B)
b = ChunkHandle(ch.fl, ch.sigma, max_batch=B)
a.upload(lw, gps); b.upload(lw, gps)
a.eval(); b.eval(); assert np.array_equal(a.fetch(), ref) and np.array_equal(b.fetch(), ref)
a.upload(lw, gps); b.upload(lw, gps)
a.eval()
t0 = time.perf_counter()
for _ in range(steps // 2):
    b.eval()
    ra = a.fetch(); a.upload(lw, gps); a.eval()
    rb = b.fetch(); b.upload(lw, gps)
a.fetch()
dt = time.perf_counter() - t0
n = 2 * (steps // 2)
assert np.array_equal(ra, ref) and np.array_equal(rb, ref)
print(f"two handles of 32, step k+1 enqueued before step k is fetched: {1e3 * dt / n:.2f} ms per 32 evals, {B * n / dt:.1f} evals/s")
a.close(); b.close()
