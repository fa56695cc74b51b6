"""Launch-per-step soak with a DIFFERENT proposal in every workspace at every launch:
    python tools/soak_batch_perm.py CFG B SECONDS [SCHEME]
tools/soak_batch.py (rounds 1-5) uploads its B proposals once and evaluates them over and over, so every matrix workspace
holds, from the launch before, exactly the bits the next launch is going to write: a task that reads a tile, a mailbox block
or an accumulator AHEAD of its producer gets the right answer by accident and the soak cannot see it.  Here the proposals
are permuted (and, for B = 1, cycled) before every launch, so what a workspace holds from the launch before belongs to
another proposal; with PSOAP_DEBUG_POISON=15 the library additionally fills the matrices, mailboxes, partial tiles and
accumulator records with NaN patterns before every launch."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from psoap_amd import synthetic as syn
from psoap_amd.chunk import ChunkHandle

cfg, B, budget = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
if len(sys.argv) > 4:
    os.environ["PSOAP_DAG_SCHEME"] = sys.argv[4]
ch = syn.make_config_chunk(cfg)
c = ch.n_components
NP = max(B, 8)                       # pool of proposals the launches draw from
gps = syn.make_walkers(c, NP, seed=cfg)
lw = syn.walker_lwls(ch, syn.make_walker_velocities(ch, NP, seed=cfg + 10))
rng = np.random.default_rng(cfg)
t_end = time.time() + budget
n, bad = 0, []
with ChunkHandle(ch.fl, ch.sigma, max_batch=B) as h:
    ref = np.empty(NP)
    for i0 in range(0, NP, B):          # reference values: the pool in order, B at a time
        idx = np.arange(i0, min(i0 + B, NP))
        idx = np.concatenate([idx, np.arange(B - len(idx))]) if len(idx) < B else idx
        h.upload(lw[idx], gps[idx]); h.eval()
        ref[idx] = h.fetch()
    while time.time() < t_end:
        for _ in range(50):
            idx = rng.permutation(NP)[:B]
            h.upload(lw[idx], gps[idx]); h.eval()
            out = h.fetch()
            n += B
            if not np.array_equal(out, ref[idx]):
                w = np.flatnonzero(out != ref[idx])
                bad.append((n, idx[w].tolist(), out[w].tolist(), ref[idx][w].tolist()))
print(f"cfg {cfg} B {B} (launch per step, permuted proposals, poison {os.environ.get('PSOAP_DEBUG_POISON', '0')}, "
      f"scheme {os.environ.get('PSOAP_DAG_SCHEME', 'auto')}): {n} matrices, {len(bad)} mismatching launches")
for b in bad[:20]:
    print("  ", b)
