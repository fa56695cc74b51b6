"""How safe is ONE GPU under several processes?  W worker processes (the reference's fork-then-INIT model, sample_parallel.py:
258-278), each with its own chunk, evaluate TWO proposals in turn, over and over, through the drop-in call (a stale tile of
the previous evaluation then belongs to the other proposal: a repeated single proposal would hide it); every value is
compared with the process's first of that proposal.

    python tools/shared_gpu_probe.py [workers reps cfg lock [policy]]

cfg: 3 = N 6000 (SB2), 1 = N 2000 (SB1), 5 = N 8192 (ST3); lock: 0 = PSOAP_DEVICE_LOCK=0, 1 = ensemble.SharedDeviceLock around
each call and the library's own lock off, 2 = the library's device lock (the default); policy: auto (default) | dag | staged
(PSOAP_SHARE_POLICY).  PSOAP_GPU_SERVER=auto in the environment: the workers become clients of ONE process that owns the GPU
(psoap_amd/server.py) and evaluates the requests that arrive together in one group launch.  PSOAP_SHARE_DETECT_ONLY=1 in the environment: disturbed launches are counted but their values kept --
how many of the wrong values belong to a launch that reported a moved workgroup ("wrong_in_disturbed_launch")?  Per worker: WRONG values (beyond the 1e-10 parity contract: what must never happen), values that
differ in the last bits only (a retry that ended on the staged path), -1 = the series ended in an error; and the library's
own account (psoap_share_stats): disturbed launches, retries, staged evaluations, time spent waiting for the device."""
import json
import multiprocessing as mp
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
argv = sys.argv[1:]
W, reps, cfg, lock = (int(a) for a in (argv[:4] + ["8", "100", "3", "2"][len(argv[:4]):]))
policy = argv[4] if len(argv) > 4 else "auto"
os.environ["PSOAP_DEVICE_LOCK"] = "1" if lock == 2 else "0"
if policy != "auto":
    os.environ["PSOAP_SHARE_POLICY"] = policy


def worker(k, q, go):
    import numpy as np
    from psoap_amd import _lib, covariance, synthetic as syn
    from psoap_amd.ensemble import SharedDeviceLock, _NoLock
    ch = syn.make_config_chunk(4 if cfg == 3 else cfg, k) if cfg >= 3 else syn.make_chunk(cfg, 10, 200, seed=100 + k)
    gp = np.asarray(syn.GP_BASE[ch.n_components], dtype=float)
    gps = (gp, gp * (1.0 + 0.03 * (1 + np.arange(gp.size) % 2)))
    fn = {1: covariance.lnlike_f, 2: covariance.lnlike_f_g, 3: covariance.lnlike_f_g_h}[ch.n_components]
    calls = [(None, *ch.lwls, ch.fl, ch.sigma, *g) for g in gps]
    guard = SharedDeviceLock(0, "probe") if lock == 1 else _NoLock()
    err = ""
    wrong = bits = wrong_tainted = 0
    detect_only = os.environ.get("PSOAP_SHARE_DETECT_ONLY") == "1"
    t0 = time.time()
    first = [None, None]
    details = []
    try:
        with guard:
            first = [fn(*c) for c in calls]          # (the first pair comes up while the others still start: HIP init)
        try:
            go.wait(timeout=900)
        except Exception:                            # (another worker failed before the start line: go on alone)
            pass
        t0 = time.time()
        for r in range(reps):
            before = _lib.share_stats(0) if not os.environ.get("PSOAP_GPU_SERVER") else {"tainted": 0, "retries": 0, "staged_fallbacks": 0, "staged_policy": 0}
            t_before = before["tainted"]
            with guard:
                v = fn(*calls[r & 1])
            f = first[r & 1]
            if v != f:
                if abs(v - f) > 1e-10 * max(1.0, abs(f)):
                    wrong += 1
                    if detect_only and _lib.share_stats(0)["tainted"] > t_before:
                        wrong_tainted += 1
                    if len(details) < 4 and not os.environ.get("PSOAP_GPU_SERVER"):   # what came back instead, and what the library knew
                        after = _lib.share_stats(0)
                        details.append({"call": r, "got": repr(v), "want": repr(f), "other_proposal": repr(first[1 - (r & 1)]),
                                        "tainted_during": after["tainted"] - before["tainted"],
                                        "retries_during": after["retries"] - before["retries"],
                                        "staged_during": after["staged_fallbacks"] + after["staged_policy"]
                                        - before["staged_fallbacks"] - before["staged_policy"], "procs": after["procs"]})
                else:
                    bits += 1
    except Exception as e:          # (a dependency wait that timed out, a lock that was never released: the series ends)
        err = str(e)[:200]
        wrong = -1
        try:
            go.abort()
        except Exception:
            pass
    dt = time.time() - t0
    try:
        if os.environ.get("PSOAP_GPU_SERVER"):         # (the workers hold no GPU context: what the SERVER did, from worker 0)
            stats = {"server_" + key: v for key, v in next(iter(covariance._handles.values())).server_stats().items()} if k == 0 else {}
        else:
            stats = _lib.share_stats(0)
    except Exception as e:
        stats = {"error": str(e)[:80]}
    stats["wrong_in_disturbed_launch"] = int(wrong_tainted)
    stats["wrong_details"] = details
    q.put((k, int(wrong), int(bits), first, dt, err, stats))


if __name__ == "__main__":
    ctx = mp.get_context("fork")
    q = ctx.Queue()
    go = ctx.Barrier(W)
    ps = [ctx.Process(target=worker, args=(k, q, go)) for k in range(W)]
    for p in ps:
        p.start()
    res = sorted(q.get() for _ in ps)
    for p in ps:
        p.join()
    tot = {}
    for r in res:
        for key, v in r[6].items():
            if isinstance(v, int) and key not in ("procs", "lock_enabled"):
                tot[key] = tot.get(key, 0) + v
    secs = max(r[4] for r in res)
    n_eval = W * reps
    print(f"{W} workers x {reps} evaluations (cfg {cfg}, lock {lock}, policy {policy}): WRONG per worker {[r[1] for r in res]}, "
          f"last-bit differences {[r[2] for r in res]}, {secs:.1f} s = {1e3 * secs / max(reps, 1):.2f} ms per call per worker, "
          f"{n_eval / secs:.0f} evals/s in all")
    print("  library account (all workers): " + json.dumps(tot) + f"; procs seen {[r[6].get('procs') for r in res]}")
    for r in res:
        if r[5]:
            print(f"  worker {r[0]} ended with: {r[5]}")
        for d in r[6].get("wrong_details", []):
            print(f"  worker {r[0]} WRONG: {json.dumps(d)}")
    sys.exit(1 if any(r[1] for r in res) else 0)
