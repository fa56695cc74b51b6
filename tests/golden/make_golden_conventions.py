"""What the REFERENCE's lnlike_f / lnlike_f_g / lnlike_f_g_h do with non-finite inputs (ValueError from scipy's
check_finite, -inf from a failed factorisation, or a number): recorded by running the reference itself in the build
container -> tests/golden/golden_conventions_v1.json (outcomes only).  Same import recipe as make_golden.py.

    python tests/golden/make_golden_conventions.py
"""
import json
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import make_golden  # noqa: E402
from psoap_amd import _convention_cases as cc  # noqa: E402


def main():
    mf, cov, data = make_golden.import_reference()
    warnings.simplefilter("ignore")
    rec = {}
    with np.errstate(all="ignore"):
        for name, fname, args, kwargs in cc.cases():
            N = len(args[0])
            rec[name] = dict(cc.outcome(getattr(cov, fname), args, kwargs, np.empty((N, N))), function=fname)
    with open(os.path.join(HERE, "golden_conventions_v1.json"), "w") as fh:
        json.dump(rec, fh, indent=1, sort_keys=True)
    for k, v in rec.items():
        print(f"{k:24s} {v['function']:14s} {v['kind']}")


if __name__ == "__main__":
    main()
