"""How safe is ONE GPU under several processes?  W worker processes (the reference's fork-then-INIT model, sample_parallel.py:
258-278), each with its own chunk, evaluate the same proposal over and over through the drop-in call; every value is compared
with the process's first.  Prints mismatches per worker (-1: the worker's series ended in an error).      python tools/shared_gpu_probe.py [workers reps cfg lock]
lock: 0 = none (PSOAP_DEVICE_LOCK=0: the hazard itself), 1 = psoap_amd.ensemble.SharedDeviceLock around each call and the
library's own lock off, 2 = the library's device lock (the default behaviour of libpsoap_gp.so)."""
import multiprocessing as mp
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
W, reps, cfg, lock = (int(a) for a in (sys.argv[1:5] + ["8", "100", "3", "2"][len(sys.argv) - 1:]))
os.environ["PSOAP_DEVICE_LOCK"] = "1" if lock == 2 else "0"


def worker(k, q):
    import numpy as np
    from psoap_amd import covariance, synthetic as syn
    from psoap_amd.ensemble import SharedDeviceLock, _NoLock
    ch = syn.make_config_chunk(4 if cfg == 3 else cfg, k) if cfg >= 3 else syn.make_chunk(2, 10, 140, seed=100 + k)
    gp = syn.GP_BASE[ch.n_components]
    call = (None, *ch.lwls, ch.fl, ch.sigma, *gp)
    fn = {1: covariance.lnlike_f, 2: covariance.lnlike_f_g, 3: covariance.lnlike_f_g_h}[ch.n_components]
    guard = SharedDeviceLock(0, "probe") if lock == 1 else _NoLock()
    with guard:
        first = fn(*call)
    bad = 0
    t0 = time.time()
    err = ""
    try:
        for _ in range(reps):
            with guard:
                v = fn(*call)
            bad += (v != first)
    except Exception as e:          # (without a lock: a dependency wait that timed out ends the worker's series)
        err = str(e)[:160]
        bad = -1
    q.put((k, int(bad), first, time.time() - t0, err))


if __name__ == "__main__":
    ctx = mp.get_context("fork")
    q = ctx.Queue()
    ps = [ctx.Process(target=worker, args=(k, q)) for k in range(W)]
    for p in ps:
        p.start()
    res = sorted(q.get() for _ in ps)
    for p in ps:
        p.join()
    print(f"{W} workers x {reps} evaluations (cfg {cfg}, lock {lock}): mismatches per worker {[r[1] for r in res]}, "
          f"seconds {max(r[3] for r in res):.1f}")
    for r in res:
        if r[4]:
            print(f"  worker {r[0]} ended with: {r[4]}")
    sys.exit(1 if lock and any(r[1] for r in res) else 0)
