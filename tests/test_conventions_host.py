"""CPU: the shim's handling of degenerate inputs (psoap_amd/covariance.py::_lnlike) against the outcomes recorded from
the reference (tests/golden/golden_conventions_v1.json), with the oracle standing in for the device call -- the host
logic decides ValueError / ZeroDivisionError / -inf; the GPU version of this test is in tests/test_gpu_parity.py."""
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shim_conventions_match_the_reference_record(oracle, monkeypatch):
    from psoap_amd import _convention_cases as cc
    from psoap_amd import covariance as cov

    class OracleHandle:
        def __init__(self, fl, sigma):
            self.fl, self.sigma = np.asarray(fl, float), np.asarray(sigma, float)

        def lnlike(self, lw, gp, mu):
            with np.errstate(all="ignore"):
                return oracle.lnlike(lw, self.fl, self.sigma, gp, mu)

    monkeypatch.setattr(cov, "_chunk_for", lambda fl, sigma: OracleHandle(fl, sigma))
    with open(os.path.join(ROOT, "tests", "golden", "golden_conventions_v1.json")) as fh:
        want = json.load(fh)
    cases = cc.cases()
    assert sorted(want) == sorted(c[0] for c in cases)
    for name, fname, args, kwargs in cases:
        got = cc.outcome(getattr(cov, fname), args, kwargs, None)
        assert got["kind"] == want[name]["kind"], (name, got, want[name])
        if got["kind"] == "finite":
            assert abs(got["value"] - want[name]["value"]) <= 1e-10 * max(1.0, abs(want[name]["value"])), name
