"""Wall time of ONE reference-signature call, as an unchanged psoap.sample_parallel worker makes it per Metropolis
step (sample_parallel.py:193): psoap_amd.covariance.lnlike_f_g(V11, wl_f, wl_g, fl, sigma, amp_f, l_f, amp_g, l_g)
with new wavelength vectors every call, against the device time of the launch."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from psoap_amd import synthetic as syn, covariance as cov
for cfg in (1, 3, 5):
    ch = syn.make_config_chunk(cfg)
    c = ch.n_components
    fn = {1: cov.lnlike_f, 2: cov.lnlike_f_g, 3: cov.lnlike_f_g_h}[c]
    V11 = np.empty((1, 1))
    vel = syn.make_walker_velocities(ch, 16, seed=3)
    lws = syn.walker_lwls(ch, vel)                       # 16 different proposals
    for _ in range(3):
        fn(V11, *lws[0], ch.fl, ch.sigma, *syn.GP_BASE[c])
    ts = []
    for k in range(16):
        t0 = time.perf_counter()
        lnp = fn(V11, *lws[k], ch.fl, ch.sigma, *syn.GP_BASE[c])
        ts.append(time.perf_counter() - t0)
    print(f"cfg{cfg} N={ch.N} c={c}: {1e3*np.median(ts):.3f} ms per call (min {1e3*min(ts):.3f}), lnp {lnp:.6f}")
cov.release_handles()
