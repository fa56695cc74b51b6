"""Inputs of the non-finite-input convention cases shared by tests/golden/make_golden_conventions.py (which records what
the REFERENCE does with them) and tests/test_gpu_parity.py (which checks the shim against that record).  Test support
only; nothing in the product imports it."""
import numpy as np

from . import synthetic as syn


def _poisoned(a, i, v=np.nan):
    b = a.copy()
    b[i] = v
    return b


def cases():
    """-> [(name, function name in covariance, positional args after V11, kwargs)]"""
    ch1 = syn.make_chunk(1, 3, 50, seed=79)
    ch2 = syn.make_chunk(2, 3, 50, seed=77)
    ch3 = syn.make_chunk(3, 3, 50, seed=78)
    g1, g2, g3 = (list(syn.GP_BASE[c]) for c in (1, 2, 3))
    P = _poisoned
    dup = ch1.lwls[0].copy()
    dup[8] = dup[7]
    out = []
    for tag, bad in (("nan", np.nan), ("inf", np.inf)):
        out += [
            (f"f_fl_{tag}", "lnlike_f", [ch1.lwls[0], P(ch1.fl, 3, bad), ch1.sigma, *g1], {}),
            (f"fg_fl_{tag}", "lnlike_f_g", [ch2.lwls[0], ch2.lwls[1], P(ch2.fl, 17, bad), ch2.sigma, *g2], {}),
            (f"fgh_fl_{tag}", "lnlike_f_g_h", [*ch3.lwls, P(ch3.fl, 9, bad), ch3.sigma, *g3], {}),
            (f"f_sigma_{tag}", "lnlike_f", [ch1.lwls[0], ch1.fl, P(ch1.sigma, 5, bad), *g1], {}),
            (f"fg_sigma_{tag}", "lnlike_f_g", [ch2.lwls[0], ch2.lwls[1], ch2.fl, P(ch2.sigma, 5, bad), *g2], {}),
            (f"fgh_sigma_{tag}", "lnlike_f_g_h", [*ch3.lwls, ch3.fl, P(ch3.sigma, 5, bad), *g3], {}),
            (f"f_wl_{tag}", "lnlike_f", [P(ch1.lwls[0], 11, bad), ch1.fl, ch1.sigma, *g1], {}),
            (f"fg_wl_{tag}", "lnlike_f_g", [P(ch2.lwls[0], 11, bad), ch2.lwls[1], ch2.fl, ch2.sigma, *g2], {}),
            (f"fgh_wl_{tag}", "lnlike_f_g_h", [ch3.lwls[0], P(ch3.lwls[1], 2, bad), ch3.lwls[2], ch3.fl, ch3.sigma, *g3], {}),
            (f"f_amp_{tag}", "lnlike_f", [ch1.lwls[0], ch1.fl, ch1.sigma, bad, 5.0], {}),
            (f"fg_amp_{tag}", "lnlike_f_g", [ch2.lwls[0], ch2.lwls[1], ch2.fl, ch2.sigma, 0.2, 5.0, bad, 7.0], {}),
            (f"fgh_l_{tag}", "lnlike_f_g_h", [*ch3.lwls, ch3.fl, ch3.sigma, 0.2, 5.0, 0.1, bad, 0.05, 6.0], {}),
            (f"fg_mu_{tag}", "lnlike_f_g", [ch2.lwls[0], ch2.lwls[1], ch2.fl, ch2.sigma, *g2], {"mu_GP": bad}),
        ]
    out += [
        ("f_l0_coincident", "lnlike_f", [dup, ch1.fl, ch1.sigma, 0.2, 0.0], {}),
        ("f_l0_distinct", "lnlike_f", [ch1.lwls[0], ch1.fl, ch1.sigma, 0.2, 0.0], {}),
        ("fg_l0_coincident", "lnlike_f_g", [P(ch2.lwls[0], 8, ch2.lwls[0][7]), ch2.lwls[1], ch2.fl, ch2.sigma, 0.2, 0.0, 0.1, 7.0], {}),
        ("f_negative_amp_wins", "lnlike_f", [P(ch1.lwls[0], 1), P(ch1.fl, 2), ch1.sigma, -0.2, 5.0], {}),
        ("fg_negative_l_wins", "lnlike_f_g", [ch2.lwls[0], ch2.lwls[1], P(ch2.fl, 2), ch2.sigma, 0.2, 5.0, 0.1, -7.0], {}),
        ("fg_clean", "lnlike_f_g", [ch2.lwls[0], ch2.lwls[1], ch2.fl, ch2.sigma, *g2], {}),
    ]
    return out


def outcome(fn, args, kwargs, V11):
    """What a call does, as a record: {"kind": "ValueError" | "ZeroDivisionError" | "-inf" | "nan" | "finite", "value": ...}"""
    try:
        v = fn(V11, *args, **kwargs)
    except ValueError:
        return {"kind": "ValueError", "value": None}
    except ZeroDivisionError:
        return {"kind": "ZeroDivisionError", "value": None}
    v = float(v)
    if np.isneginf(v):
        return {"kind": "-inf", "value": None}
    if np.isnan(v):
        return {"kind": "nan", "value": None}
    return {"kind": "finite", "value": v}
