#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ by IMPORTING the reference.

Runs only in the build container (needs /root/reference, Cython, gcc).  The
reference never travels: this script compiles psoap/matrix_functions.pyx into a
temporary directory, imports psoap.covariance / psoap.data straight from
/root/reference with two import shims (an empty ``h5py`` stub because
psoap/data.py:4 imports it at module top, and ``np.float = float`` because
predict_* use the removed alias, e.g. psoap/covariance.py:102), calls the
reference functions on seeded synthetic inputs and stores *inputs' seeds and
outputs only* as small .npz files.

    python tests/golden/make_golden.py

Inputs are regenerated from the seeds by psoap_amd.synthetic on any machine.
"""
import glob
import importlib.machinery
import importlib.util
import os
import subprocess
import sys
import sysconfig
import tempfile
import types
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

from psoap_amd import synthetic as syn  # noqa: E402


def import_reference():
    warnings.simplefilter("ignore")
    np.float = float
    sys.modules["h5py"] = types.ModuleType("h5py")
    sys.path.insert(0, REF)
    build = tempfile.mkdtemp(prefix="psoap_ref_build_")
    csrc = os.path.join(build, "matrix_functions.c")
    subprocess.check_call([sys.executable, "-m", "cython", "-3",
                           os.path.join(REF, "psoap", "matrix_functions.pyx"), "-o", csrc])
    so = os.path.join(build, "matrix_functions" + sysconfig.get_config_var("EXT_SUFFIX"))
    subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-w",
                           "-I", sysconfig.get_paths()["include"], "-I", np.get_include(),
                           csrc, "-o", so, "-lm"])
    import psoap
    loader = importlib.machinery.ExtensionFileLoader("psoap.matrix_functions", so)
    spec = importlib.util.spec_from_loader("psoap.matrix_functions", loader)
    mod = importlib.util.module_from_spec(spec)
    loader.exec_module(mod)
    sys.modules["psoap.matrix_functions"] = mod
    psoap.matrix_functions = mod
    from psoap import covariance, data
    return mod, covariance, data


def main():
    mf, cov, data = import_reference()
    out = {}

    # ---------------------------------------------------------------- fills (small, full matrices)
    ch = syn.make_chunk(3, 4, 24, seed=11)           # N = 96
    N = ch.N
    gp = syn.GP_BASE[3]
    m = np.empty((N, N)); mf.fill_V11_f(m, ch.lwls[0], *gp[:2]); out["fill_f"] = m.copy()
    m = np.empty((N, N)); mf.fill_V11_f_g(m, ch.lwls[0], ch.lwls[1], *gp[:4]); out["fill_f_g"] = m.copy()
    m = np.empty((N, N)); mf.fill_V11_f_g_h(m, *ch.lwls, *gp); out["fill_f_g_h"] = m.copy()
    pred = np.linspace(ch.lwls[0].min(), ch.lwls[0].max(), 40)
    m = np.empty((40, N)); mf.fill_V12_f(m, pred, ch.lwls[0], *gp[:2]); out["fill_cross_40xN"] = m.copy()
    m = np.empty((N, 40)); mf.fill_V12_f(m, ch.lwls[1], pred, *gp[2:4]); out["fill_cross_Nx40"] = m.copy()
    out["fill_meta"] = np.array([3, 4, 24, 11, 40])

    # ---------------------------------------------------------------- replicate_wls (harness convention)
    chm = syn.make_chunk(2, 7, 150, seed=5, masked_fraction=0.2)
    out["replicate_masked"] = data.replicate_wls(chm.lwl, chm.velocities, chm.mask)
    out["replicate_meta"] = np.array([2, 7, 150, 5])

    # ---------------------------------------------------------------- lnlike scalars
    # (name, c, n_epochs, n_pix, seed, masked_fraction)
    cases = [
        ("sb1_n64", 1, 4, 16, 101, 0.0),
        ("sb2_n64", 2, 4, 16, 102, 0.0),
        ("st3_n64", 3, 4, 16, 103, 0.0),
        ("sb1_n256", 1, 8, 32, 104, 0.0),
        ("sb2_n256", 2, 8, 32, 105, 0.0),
        ("st3_n256", 3, 8, 32, 106, 0.0),
        ("sb2_ragged", 2, 7, 150, 5, 0.2),           # N = 826, ragged epochs
        ("sb2_n129", 2, 3, 43, 107, 0.0),            # one past a 128 tile edge
        ("st3_n1000", 3, 10, 100, 108, 0.0),
        ("cfg1_sb1_n2000", 1, 10, 200, 1000, 0.0),
        ("cfg2_sb1_n4096", 1, 16, 256, 2000, 0.0),
        ("cfg3_sb2_n6000", 2, 20, 300, 3000, 0.0),
        ("cfg5_st3_n8192", 3, 16, 512, 5000, 0.0),
    ]
    names, meta, vals = [], [], []
    for name, c, ne, npx, seed, mf_ in cases:
        chk = syn.make_chunk(c, ne, npx, seed=seed, masked_fraction=mf_)
        V = np.empty((chk.N, chk.N))
        fn = {1: cov.lnlike_f, 2: cov.lnlike_f_g, 3: cov.lnlike_f_g_h}[c]
        v = fn(V, *chk.lwls, chk.fl, chk.sigma, *syn.GP_BASE[c])
        names.append(name); meta.append([c, ne, npx, seed, int(round(mf_ * 100)), chk.N]); vals.append(v)
        print(name, chk.N, repr(v))
    out["lnlike_names"] = np.array(names)
    out["lnlike_meta"] = np.array(meta)
    out["lnlike_vals"] = np.array(vals)

    # non-default mu_GP
    chk = syn.make_chunk(2, 8, 32, seed=105)
    V = np.empty((chk.N, chk.N))
    out["lnlike_mu0p9"] = np.array(cov.lnlike_f_g(V, *chk.lwls, chk.fl, chk.sigma, *syn.GP_BASE[2], mu_GP=0.9))

    # -inf conventions: negative amp, negative l, non positive-definite
    infs = []
    infs.append(cov.lnlike_f(V, chk.lwls[0], chk.fl, chk.sigma, -0.2, 5.0))
    infs.append(cov.lnlike_f_g(V, *chk.lwls, chk.fl, chk.sigma, 0.2, 5.0, 0.1, -7.0))
    infs.append(cov.lnlike_f_g_h(V, *chk.lwls, chk.lwls[0], chk.fl, chk.sigma, 0.2, 5.0, 0.1, 7.0, -0.05, 6.0))
    # singular: two identical pixels with zero noise -> LinAlgError -> -inf (covariance.py:349-350)
    lw = chk.lwls.copy(); lw[:, 1] = lw[:, 0]
    sig0 = np.zeros_like(chk.sigma)
    infs.append(cov.lnlike_f_g(V, *lw, chk.fl, sig0, *syn.GP_BASE[2]))
    out["lnlike_infs"] = np.array(infs)
    print("infs", infs)

    # zero amplitude is legal (only <0 is rejected): K = diag(sigma^2)
    out["lnlike_zero_amp"] = np.array(cov.lnlike_f(V, chk.lwls[0], chk.fl, chk.sigma, 0.0, 5.0))

    # ---------------------------------------------------------------- walker batch on the cfg3 chunk
    ch3 = syn.make_config_chunk(3)
    nw = 4
    gps = syn.make_walkers(2, nw, seed=3500)
    vels = syn.make_walker_velocities(ch3, nw, seed=3501)
    lw = syn.walker_lwls(ch3, vels)
    ref_lw = np.stack([data.replicate_wls(ch3.lwl, v, ch3.mask) for v in vels])
    assert np.array_equal(ref_lw, lw), "synthetic.replicate_wls differs from the reference"
    V = np.empty((ch3.N, ch3.N))
    out["walkers_cfg3"] = np.array([cov.lnlike_f_g(V, *lw[w], ch3.fl, ch3.sigma, *gps[w]) for w in range(nw)])
    print("walkers", out["walkers_cfg3"])

    # ---------------------------------------------------------------- predict (small: full outputs)
    chp = syn.make_chunk(3, 5, 60, seed=21)          # N = 300
    M = 48
    predg = np.linspace(chp.lwls[0].min(), chp.lwls[0].max(), M)
    mu, Sig = cov.predict_f_g(chp.lwls[0], chp.lwls[1], chp.fl, chp.sigma, predg, predg,
                              0.0, 0.2, 5.0, 0.0, 0.1, 7.0)
    out["pred_fg_mu"], out["pred_fg_Sigma"] = mu, Sig
    out["pred_fg_mu_only"] = cov.predict_f_g(chp.lwls[0], chp.lwls[1], chp.fl, chp.sigma, predg, predg + 1e-5,
                                             0.3, 0.2, 5.0, 0.7, 0.1, 7.0, get_Sigma=False)
    mu, Sig = cov.predict_f_g_h(*chp.lwls, chp.fl, chp.sigma, predg, predg, predg,
                                0.0, 0.0, 0.0, *syn.GP_BASE[3])
    out["pred_fgh_mu"], out["pred_fgh_Sigma"] = mu, Sig
    mu, Sig = cov.predict_f_g_sum(chp.lwls[0], chp.lwls[1], chp.fl, chp.sigma, predg, predg,
                                  1.0, 0.2, 5.0, 0.1, 7.0)
    out["pred_fg_sum_mu"], out["pred_fg_sum_Sigma"] = mu, Sig
    # 3-component sum: only valid for M == N (covariance.py:294); predict on the data grid
    chq = syn.make_chunk(3, 4, 30, seed=22)          # N = 120
    mu, Sig = cov.predict_f_g_h_sum(*chq.lwls, chq.fl, chq.sigma, *chq.lwls, 1.0, *syn.GP_BASE[3])
    out["pred_fgh_sum_mu"], out["pred_fgh_sum_Sigma"] = mu, Sig
    out["pred_meta"] = np.array([3, 5, 60, 21, M, 3, 4, 30, 22])

    # ---------------------------------------------------------------- predict (retrieve-script shape, summaries)
    chr_ = syn.make_chunk(3, 10, 200, seed=23)       # N = 2000, M = 2*n_pix = 400 (psoap_retrieve_ST3.py:97)
    Mr = 400
    predr = np.linspace(chr_.lwls[0].min(), chr_.lwls[0].max(), Mr)
    mu, Sig = cov.predict_f_g_h(*chr_.lwls, chr_.fl, chr_.sigma, predr, predr, predr,
                                0.0, 0.0, 0.0, *syn.GP_BASE[3])
    out["predL_fgh_mu"] = mu
    out["predL_fgh_diag"] = np.diag(Sig).copy()
    out["predL_fgh_rows"] = Sig[[0, 399, 400, 777, 1199]].copy()
    mu, Sig = cov.predict_f_g(chr_.lwls[0], chr_.lwls[1], chr_.fl, chr_.sigma, predr, predr,
                              0.0, 0.2, 5.0, 0.0, 0.1, 7.0)
    out["predL_fg_mu"] = mu
    out["predL_fg_diag"] = np.diag(Sig).copy()
    out["predL_meta"] = np.array([3, 10, 200, 23, Mr])

    path = os.path.join(HERE, "golden_v1.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
