"""Wall time of the predict family at the retrieve-script shape (cfg5: N = 8192, c = 3, M = 1024) and of
one calibration solve at the calibration-script shape (M = 512, N = 1536, c = 3)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from psoap_amd import synthetic as syn
from psoap_amd import covariance as cov

ch = syn.make_config_chunk(5)
M = 1024
pred = np.linspace(ch.lwls[0].min(), ch.lwls[0].max(), M)
gp = syn.GP_BASE[3]
args = (ch.lwls[0], ch.lwls[1], ch.lwls[2], ch.fl, ch.sigma, pred, pred, pred, 0.0, 0.0, 0.0, *gp)
N = ch.N
flops = N**3 / 3 + N**2 * 3 * M + N * (3 * M) ** 2 + 2 * N * 3 * M
for rep in range(3):
    t0 = time.perf_counter()
    mu, Sigma = cov.predict_f_g_h(*args)
    dt = time.perf_counter() - t0
    print(f"predict_f_g_h N={N} M={M}: {1e3*dt:8.1f} ms  ({flops/dt/1e12:5.1f} TFLOP/s algorithmic)  mu[0]={mu[0]:.6f}")
chs = syn.make_chunk(3, 4, 512, seed=77)
ep = chs.epoch_index
cal, fix = ep == 0, ep >= 1
for rep in range(3):
    t0 = time.perf_counter()
    fl_cor, X = cov.optimize_calibration_components(chs.lwl.min(), chs.lwl.max(), chs.lwl[cal], chs.lwls[:, cal], chs.fl[cal],
                                                    chs.sigma[cal], chs.lwls[:, fix], chs.fl[fix], chs.sigma[fix], gp, order=1)
    dt = time.perf_counter() - t0
    print(f"calibration M={cal.sum()} N={fix.sum()}: {1e3*dt:8.2f} ms  X={X}")
