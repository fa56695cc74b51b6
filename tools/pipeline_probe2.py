"""Only the last experiment of pipeline_probe.py (two full 32-walker steps in flight), for a kernel trace."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from psoap_amd import synthetic as syn
from psoap_amd.chunk import ChunkHandle
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ch = syn.make_config_chunk(3)
B = 32
gps = syn.make_walkers(2, B, seed=3500)
lw = syn.walker_lwls(ch, syn.make_walker_velocities(ch, B, seed=3501))
a = ChunkHandle(ch.fl, ch.sigma, max_batch=B)
b = ChunkHandle(ch.fl, ch.sigma, max_batch=B)
a.upload(lw, gps); b.upload(lw, gps)
a.eval(); b.eval(); ref = a.fetch(); b.fetch()
a.upload(lw, gps); b.upload(lw, gps)
a.eval()
t0 = time.perf_counter()
for _ in range(steps // 2):
    b.eval()
    ra = a.fetch(); a.upload(lw, gps); a.eval()
    rb = b.fetch(); b.upload(lw, gps)
a.fetch()
dt = time.perf_counter() - t0
print(f"{1e3 * dt / (2 * (steps // 2)):.2f} ms per 32 evals")
a.close(); b.close()
