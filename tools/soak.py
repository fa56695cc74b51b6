"""Soak test of the persistent kernel's dependency protocol: many back-to-back launches over several shapes
and batch sizes (both split schemes, group launches), every result compared bit for bit with the first."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from psoap_amd import synthetic as syn
from psoap_amd.chunk import ChunkGroup, ChunkHandle

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
t_end = time.time() + budget
cases = [(1, 32), (1, 5), (2, 32), (3, 32), (3, 3), (3, 12), (5, 8), (5, 1)]
n_launch = 0
n_stream = 0
while time.time() < t_end:
    for cfg, B in cases:
        ch = syn.make_config_chunk(cfg)
        c = ch.n_components
        gps = syn.make_walkers(c, B, seed=cfg)
        lw = syn.walker_lwls(ch, syn.make_walker_velocities(ch, B, seed=cfg + 10))
        with ChunkHandle(ch.fl, ch.sigma, max_batch=B) as h:
            h.upload(lw, gps)
            h.eval()
            ref = h.fetch()
            reps = 200 if ch.N <= 2000 else (40 if ch.N <= 4096 else 15)
            for _ in range(reps):
                h.eval()
                out = h.fetch()
                assert np.array_equal(out, ref), (cfg, B)
                n_launch += 1
    # streamed evaluation: many matrices through the lanes of one resident launch, two submissions in flight, changing
    # batch sizes -- every result bit-identical to the first of its kind
    for cfg, B, scheme in ((1, 32, -1), (3, 16, 0), (2, 8, -1), (1, 8, 1), (5, 8, -1)):
        ch = syn.make_config_chunk(cfg)
        c = ch.n_components
        gps = syn.make_walkers(c, B, seed=cfg)
        lw = syn.walker_lwls(ch, syn.make_walker_velocities(ch, B, seed=cfg + 10))
        rng = np.random.default_rng(cfg)
        with ChunkHandle(ch.fl, ch.sigma, max_batch=B) as h:
            h.stream_open(c, B, scheme)
            ref = h.stream_fetch(h.stream_submit(lw, gps))
            pending = []
            for _ in range(60 if ch.N <= 2000 else 12):
                idx = rng.permutation(B)[: int(rng.integers(1, B + 1))]
                while pending and (len(pending) == 2 or sum(len(p[1]) for p in pending) + len(idx) > B):
                    t, i = pending.pop(0)
                    assert np.array_equal(h.stream_fetch(t), ref[i]), ("stream", cfg, B)
                    n_stream += len(i)
                pending.append((h.stream_submit(lw[idx], gps[idx]), idx))
            for t, i in pending:
                assert np.array_equal(h.stream_fetch(t), ref[i]), ("stream", cfg, B)
                n_stream += len(i)
            h.stream_close()
    # group launches over mixed sizes
    chunks = [syn.make_chunk(2, 3 + k, 90 + 17 * k, seed=50 + k) for k in range(5)]
    hs = [ChunkHandle(c_.fl, c_.sigma, max_batch=6) for c_ in chunks]
    gp6 = syn.make_walkers(2, 6, seed=9)
    with ChunkGroup(hs) as g:
        ref = None
        for _ in range(100):
            for h, c_ in zip(hs, chunks):
                h.upload(np.repeat(c_.lwls[None], 6, axis=0), gp6)
            g.eval()
            out = np.stack([h.fetch() for h in hs])
            if ref is None:
                ref = out
            assert np.array_equal(out, ref)
            n_launch += 1
    for h in hs:
        h.close()
print(f"soak ok: {n_launch} launches and {n_stream} matrices through resident (stream) launches, all results bit-identical "
      "to the first of their kind")
