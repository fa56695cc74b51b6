"""Many small matrices in ONE group launch: the dependency graph against one workgroup per matrix (solo_kernel.hpp).
    python tools/small_bench.py C EPOCHS PIX CHUNKS WALKERS [MODE]
MODE: 0 graph (PSOAP_SOLO=0), 1 solo (PSOAP_SOLO=1), ab (default: both, each in a process of its own -- the switch is read
once).  One JSON line per mode: ms per ensemble step, evals/s, fraction of the fp64 peak (F(N) = N^3/3 + 2 N^2, 78.6
TFLOP/s), and the largest relative difference of the first chunk's walker lnprobs from the oracle (CPU: SciPy)."""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

C, E, PIX, NCH, B = [int(x) for x in sys.argv[1:6]]
mode = sys.argv[6] if len(sys.argv) > 6 else "ab"
if mode == "ab":
    for m in ("0", "1"):
        subprocess.run([sys.executable, os.path.abspath(__file__)] + sys.argv[1:6] + [m], env=dict(os.environ, PSOAP_SOLO=m))
    sys.exit(0)
os.environ["PSOAP_SOLO"] = mode
import numpy as np
from psoap_amd import synthetic as syn
from psoap_amd.ensemble import EnsembleEvaluator

chunks = [syn.make_chunk(C, E, PIX, seed=100 + k) for k in range(NCH)]
N = chunks[0].N
gps = syn.make_walkers(C, B, seed=7)
props = {k: (syn.walker_lwls(chunks[k], syn.make_walker_velocities(chunks[k], B, seed=20 + k)), gps) for k in range(NCH)}
ev = EnsembleEvaluator.from_chunks(chunks, max_batch=B)
out = ev.lnprob(props); ev.lnprob(props)
reps = 10
t0 = time.perf_counter()
for _ in range(reps):
    out = ev.lnprob(props)
dt = (time.perf_counter() - t0) / reps
# parity: chunk 0's walkers through the handle alone (same kernels, its own launch) against the oracle
import oracle
h0 = ev.handles[0]
got = h0.lnlike_batch(*props[0])
nchk = min(B, 4)
want = np.array([oracle.lnlike(props[0][0][w], chunks[0].fl, chunks[0].sigma, list(gps[w])) for w in range(nchk)])
rel = float(np.max(np.abs(got[:nchk] - want) / np.maximum(1.0, np.abs(want))))
flops = N ** 3 / 3.0 + 2.0 * N ** 2
print(json.dumps({"N": N, "C": C, "chunks": NCH, "walkers": B, "mode": "solo" if mode == "1" else "graph", "ms_per_step": round(1e3 * dt, 3),
                  "evals_per_s": round(NCH * B / dt, 1), "frac_peak": round(NCH * B * flops / dt / 78.6e12, 4),
                  "max_rel_vs_oracle": rel, "sum0": float(out[0])}), flush=True)
ev.close()
