"""The stream part of tools/soak.py for ONE shape, for a given time, with the details of every mismatch (a repro tool):
    python tools/soak_stream.py CFG B SCHEME SECONDS"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from psoap_amd import synthetic as syn
from psoap_amd.chunk import ChunkHandle

cfg, B, scheme, budget = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])
ch = syn.make_config_chunk(cfg)
c = ch.n_components
gps = syn.make_walkers(c, B, seed=cfg)
lw = syn.walker_lwls(ch, syn.make_walker_velocities(ch, B, seed=cfg + 10))
t_end = time.time() + budget
n, bad, opens = 0, [], 0
while time.time() < t_end:
    rng = np.random.default_rng(cfg + opens)
    opens += 1
    with ChunkHandle(ch.fl, ch.sigma, max_batch=B) as h:
        h.stream_open(c, B, scheme)
        ref = h.stream_fetch(h.stream_submit(lw, gps))
        if opens == 1:
            ref0 = ref.copy()
            # PSOAP_SOAK_REF=<file.npy>: the first run (the product library) leaves its values there, a later run of ANOTHER
            # build of the same task lists (the chaos builds) must reproduce them bit for bit
            refp = os.environ.get("PSOAP_SOAK_REF")
            if refp:
                refp = f"{refp}.cfg{cfg}.B{B}.s{scheme}.npy"
                if os.path.exists(refp):
                    want = np.load(refp)
                    if not np.array_equal(want, ref0):
                        bad.append(("first values differ from the reference library's", np.flatnonzero(want != ref0).tolist(),
                                    ref0[want != ref0].tolist(), want[want != ref0].tolist()))
                        ref0 = want
                else:
                    np.save(refp, ref0)
        elif not np.array_equal(ref, ref0):
            bad.append(("first submission of open %d" % opens, np.flatnonzero(ref != ref0).tolist(), (ref - ref0)[ref != ref0].tolist()))
        pending = []

        def check(t, i):
            global n
            out = h.stream_fetch(t)
            n += len(i)
            if not np.array_equal(out, ref0[i]):
                w = np.flatnonzero(out != ref0[i])
                bad.append(("open %d" % opens, i[w].tolist(), out[w].tolist(), ref0[i][w].tolist()))

        for _ in range(60 if ch.N <= 2000 else 12):
            idx = rng.permutation(B)[: int(rng.integers(1, B + 1))]
            while pending and (len(pending) == 2 or sum(len(p[1]) for p in pending) + len(idx) > B):
                check(*pending.pop(0))
            pending.append((h.stream_submit(lw[idx], gps[idx]), idx))
        for t, i in pending:
            check(t, i)
        h.stream_close()
print(f"cfg {cfg} B {B} scheme {scheme}: {n} matrices through {opens} streams, {len(bad)} mismatching fetches")
for b in bad[:20]:
    print("  ", b)
