"""CPU oracle for the sampling loop -- TEST INFRASTRUCTURE ONLY (see oracle/oracle.py for the rules).

``mh_chain`` is a plain restatement of the Metropolis-Hastings loop the reference carries in
/root/reference/psoap/samplers.py:103-159 (its copy of emcee 2.x's ``MHSampler.sample``; emcee itself is
not vendored in the reference and not installed here, so parity with emcee is UNPINNED -- the loop is
pinned only against this restatement of the reference's own source).  ``lnprob`` restates
``Worker.lnprob`` + the master's sum (/root/reference/psoap/sample_parallel.py:168-198, :371-390) with
the orbit and likelihood oracles, which ARE pinned by golden vectors.
"""
import numpy as np

import oracle
import orbit_oracle

C_KMS = 2.99792458e5


def mh_chain(lnprob, p0, cov, iterations, rng):
    """Returns (chain (iterations, dim), lnprobs (iterations,), n_accepted)."""
    p = np.array(p0, dtype=np.float64)
    lp = lnprob(p)
    chain = np.zeros((iterations, len(p)))
    lps = np.zeros(iterations)
    acc = 0
    for i in range(iterations):
        q = rng.multivariate_normal(p, cov)
        new = lnprob(q)
        d = new - lp
        if d < 0:
            d = np.exp(d) - rng.rand()
        if d > 0:
            p, lp = q, new
            acc += 1
        chain[i] = p
        lps[i] = lp
    return chain, lps, acc


def chunk_lnprob(model, p_orb, p_gp, lwl, fl, sigma, epoch_index, dates):
    """One chunk's lnprob for full orbital / GP parameter vectors (sample_parallel.py:179-193)."""
    v = orbit_oracle.velocities(model, p_orb, dates)
    if np.any(np.abs(v) >= C_KMS):
        return -np.inf
    lwls = lwl[None, :] - v[:, epoch_index] / C_KMS
    return oracle.lnlike(lwls, fl, sigma, p_gp)
