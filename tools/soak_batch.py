"""The launch-per-step part of tools/soak.py for ONE shape, for a given time, with the details of every mismatch:
    python tools/soak_batch.py CFG B SECONDS"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from psoap_amd import synthetic as syn
from psoap_amd.chunk import ChunkHandle

cfg, B, budget = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
ch = syn.make_config_chunk(cfg)
c = ch.n_components
gps = syn.make_walkers(c, B, seed=cfg)
lw = syn.walker_lwls(ch, syn.make_walker_velocities(ch, B, seed=cfg + 10))
t_end = time.time() + budget
n, bad = 0, []
with ChunkHandle(ch.fl, ch.sigma, max_batch=B) as h:
    h.upload(lw, gps)
    h.eval()
    ref = h.fetch().copy()
    while time.time() < t_end:
        for _ in range(50):
            h.eval()
            out = h.fetch()
            n += B
            if not np.array_equal(out, ref):
                w = np.flatnonzero(out != ref)
                bad.append((n, w.tolist(), out[w].tolist(), ref[w].tolist()))
print(f"cfg {cfg} B {B} (launch per step): {n} matrices, {len(bad)} mismatching launches")
for b in bad[:20]:
    print("  ", b)
