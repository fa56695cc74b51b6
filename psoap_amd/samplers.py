"""Metropolis-Hastings drivers for the lnprob(p) boundary.

The reference samples with emcee's ``MHSampler`` (``sampler = MHSampler(cov, dim, lnprob)``,
/root/reference/psoap/sample_parallel.py:395,434-438; /root/reference/psoap/sample.py:243) and carries
its own hook-enabled copy of the same loop, ``StateSampler`` (/root/reference/psoap/samplers.py:13-159).
emcee (requirements.txt:6, unpinned; the ``MHSampler`` class exists in emcee 2.x only) is not in this
image, so the algorithm is restated here from the reference's own copy of the loop
(samplers.py:103-159): propose ``q ~ N(p, cov)``, accept with probability ``min(1, exp(dlnp))`` where
the uniform deviate is drawn only when ``dlnp < 0``.  Parity with emcee itself is unpinned; parity with
a plain restatement of samplers.py is tested (tests/test_samplers.py, oracle/sampler_oracle.py).

One proposal per iteration cannot fill a GPU (a single N = 6000 evaluation takes 7 ms, 32 batched take
39 ms), and the reference's own workflow already runs several independent chains side by side
(``run_index`` directories, /root/reference/scripts/psoap_gelman_rubin.py).  ``MultiChainMHSampler``
advances B such chains in lock-step: every iteration draws one proposal per chain and evaluates all B
with ONE ``lnprob_batch`` call.  Each chain owns its random stream, so chain b takes the draws and -- given the same
log-probabilities -- the decisions of a scalar ``MHSampler`` run with that stream: the batching changes throughput, not
statistics.  (A batched device evaluation agrees with a single one to a few ulp, not bit for bit: the persistent kernel
schedules 1-4 matrices differently from 32, so its sums round differently; results are bit-reproducible for a given
batch size.)
"""
from __future__ import annotations

import numpy as np


class MHSampler:
    """Single-chain Metropolis-Hastings with emcee 2.x's ``MHSampler`` interface.

    ``MHSampler(cov, dim, lnprobfn, args=[], kwargs={})``; ``sample`` is a generator yielding
    ``(p, lnprob, random_state)`` per iteration; ``chain``/``flatchain`` (iterations, dim),
    ``lnprobability`` (iterations,), ``acceptance_fraction``, ``run_mcmc``, ``reset``.
    Optional ``rejectfn`` / ``acceptfn`` callbacks as in ``StateSampler`` (samplers.py:133-147).
    """

    def __init__(self, cov, dim, lnprobfn, args=(), kwargs=None, rejectfn=None, acceptfn=None):
        self.cov = np.asarray(cov, dtype=np.float64)
        self.dim = int(dim)
        self.lnprobfn = lnprobfn
        self.args = tuple(args)
        self.kwargs = dict(kwargs or {})
        self.rejectfn = rejectfn
        self.acceptfn = acceptfn
        self._random = np.random.mtrand.RandomState()
        self.reset()

    # -- emcee.sampler.Sampler surface -------------------------------------------------------------
    @property
    def random_state(self):
        return self._random.get_state()

    @random_state.setter
    def random_state(self, state):
        try:
            self._random.set_state(state)
        except Exception:
            pass

    def reset(self):
        self.iterations = 0
        self.naccepted = 0
        self._chain = np.empty((0, self.dim))
        self._lnprob = np.empty(0)

    clear_chain = reset

    @property
    def chain(self):
        return self._chain

    @property
    def flatchain(self):
        return self._chain

    @property
    def lnprobability(self):
        return self._lnprob

    @property
    def acceptance_fraction(self):
        return self.naccepted / self.iterations

    def get_lnprob(self, p):
        return self.lnprobfn(p, *self.args, **self.kwargs)

    def _propose(self, p):
        if self.dim == 1 and self.cov.ndim < 2:       # samplers.py:116-117
            return self._random.normal(loc=p[0], scale=np.ravel(self.cov)[0], size=(1,))
        return self._random.multivariate_normal(p, self.cov)

    def sample(self, p0, lnprob0=None, randomstate=None, thin=1, storechain=True, iterations=1, incremental_save=0,
               backup="chain_backup.npy"):
        """``incremental_save = k > 0`` writes the chain so far to ``backup`` every k iterations
        (``StateSampler.sample``, samplers.py:153-156)."""
        self.random_state = randomstate
        p = np.array(p0, dtype=np.float64)
        lnprob = self.get_lnprob(p) if lnprob0 is None else lnprob0
        if storechain:
            n = int(iterations / thin)
            self._chain = np.concatenate((self._chain, np.zeros((n, self.dim))), axis=0)
            self._lnprob = np.append(self._lnprob, np.zeros(n))
        i0 = self.iterations
        for i in range(int(iterations)):
            self.iterations += 1
            q = self._propose(p)
            newlnprob = self.get_lnprob(q)
            diff = newlnprob - lnprob
            if diff < 0:
                diff = np.exp(diff) - self._random.rand()
                if diff < 0 and self.rejectfn is not None:
                    self.rejectfn()
            if diff > 0:
                p = q
                lnprob = newlnprob
                self.naccepted += 1
                if self.acceptfn is not None:
                    self.acceptfn()
            if storechain and i % thin == 0:
                ind = i0 + int(i / thin)
                self._chain[ind, :] = p
                self._lnprob[ind] = lnprob
            if incremental_save and (i + 1) % incremental_save == 0 and i > 0:
                np.save(backup, self._chain)
            yield p, lnprob, self.random_state

    def run_mcmc(self, pos0, N, rstate0=None, lnprob0=None, **kwargs):
        results = None
        for results in self.sample(pos0, lnprob0, rstate0, iterations=N, **kwargs):
            pass
        return results


class MultiChainMHSampler:
    """B independent Metropolis-Hastings chains advanced in lock-step over a batched log-posterior.

    ``lnprob_batch(P)`` maps (B, dim) positions to (B,) log-probabilities (``Posterior.lnprob_batch``).
    ``seeds``: one ``RandomState`` seed (or instance) per chain; chain b consumes its stream exactly as
    ``MHSampler`` does (one multivariate normal per iteration, one uniform only when dlnp < 0).
    ``chain`` is (B, iterations, dim), ``lnprobability`` (B, iterations), ``naccepted`` (B,).
    """

    def __init__(self, cov, dim, lnprob_batch, n_chains, seeds=None):
        self.cov = np.asarray(cov, dtype=np.float64)
        self.dim = int(dim)
        self.lnprob_batch = lnprob_batch
        self.n_chains = int(n_chains)
        if seeds is None:
            seeds = [None] * self.n_chains
        if len(seeds) != self.n_chains:
            raise ValueError("need one seed per chain")
        self._random = [s if isinstance(s, np.random.mtrand.RandomState) else np.random.mtrand.RandomState(s)
                        for s in seeds]
        self._mvn_factor = None
        self.reset()

    def reset(self):
        self.iterations = 0
        self.naccepted = np.zeros(self.n_chains, dtype=np.int64)
        self._chain = np.empty((self.n_chains, 0, self.dim))
        self._lnprob = np.empty((self.n_chains, 0))

    @property
    def chain(self):
        return self._chain

    @property
    def flatchain(self):
        """(B * iterations, dim), chain-major -- ``chain[b]`` is what one reference run saves as flatchain.npy."""
        return self._chain.reshape(-1, self.dim)

    @property
    def lnprobability(self):
        return self._lnprob

    @property
    def acceptance_fraction(self):
        return self.naccepted / self.iterations

    def _eval(self, P):
        out = np.asarray(self.lnprob_batch(P), dtype=np.float64)
        if out.shape != (self.n_chains,):
            raise ValueError(f"lnprob_batch returned shape {out.shape}, expected ({self.n_chains},)")
        return out

    def sample(self, p0, lnprob0=None, thin=1, storechain=True, iterations=1, incremental_save=0,
               backup="chain_backup.npy"):
        """``p0``: (dim,) -- every chain starts there, as B reference runs would -- or (B, dim).
        ``incremental_save = k > 0`` writes the (B, iterations, dim) chain so far to ``backup`` every k
        iterations (samplers.py:153-156)."""
        p = np.array(np.broadcast_to(np.asarray(p0, dtype=np.float64), (self.n_chains, self.dim)))
        lnprob = self._eval(p) if lnprob0 is None else np.array(np.broadcast_to(lnprob0, (self.n_chains,)), dtype=np.float64)
        if storechain:
            n = int(iterations / thin)
            self._chain = np.concatenate((self._chain, np.zeros((self.n_chains, n, self.dim))), axis=1)
            self._lnprob = np.concatenate((self._lnprob, np.zeros((self.n_chains, n))), axis=1)
        i0 = self.iterations
        one_d = self.dim == 1 and self.cov.ndim < 2
        for i in range(int(iterations)):
            self.iterations += 1
            if one_d:
                q = np.stack([r.normal(loc=p[b, 0], scale=np.ravel(self.cov)[0], size=(1,))
                              for b, r in enumerate(self._random)])
            else:
                # RandomState.multivariate_normal draws standard normals, then maps them through
                # sqrt(s) * v of the covariance's SVD -- recomputed on every call.  The factor is the same
                # for every chain and iteration, so it is computed once; the draws stay bit-identical.
                if self._mvn_factor is None:
                    _u, sv, v = np.linalg.svd(self.cov)
                    self._mvn_factor = np.sqrt(sv)[:, None] * v
                q = np.stack([np.dot(r.standard_normal(self.dim).reshape(-1, self.dim), self._mvn_factor)[0] + p[b]
                              for b, r in enumerate(self._random)])
            newlnprob = self._eval(q)
            for b in range(self.n_chains):
                diff = newlnprob[b] - lnprob[b]
                if diff < 0:
                    diff = np.exp(diff) - self._random[b].rand()
                if diff > 0:
                    p[b] = q[b]
                    lnprob[b] = newlnprob[b]
                    self.naccepted[b] += 1
            if storechain and i % thin == 0:
                ind = i0 + int(i / thin)
                self._chain[:, ind, :] = p
                self._lnprob[:, ind] = lnprob
            if incremental_save and (i + 1) % incremental_save == 0 and i > 0:
                np.save(backup, self._chain)
            yield p.copy(), lnprob.copy()

    def run_mcmc(self, p0, N, lnprob0=None, **kwargs):
        results = None
        for results in self.sample(p0, lnprob0, iterations=N, **kwargs):
            pass
        return results

    # -- the same chains through a split-phase evaluator ------------------------------------------------------------
    def _propose_rows(self, p, rows):
        """one proposal per chain of ``rows`` (a range of chain indices), each from its chain's own stream -- the draws of
        ``sample`` above, chain by chain"""
        one_d = self.dim == 1 and self.cov.ndim < 2
        if one_d:
            return np.stack([self._random[b].normal(loc=p[b, 0], scale=np.ravel(self.cov)[0], size=(1,)) for b in rows])
        if self._mvn_factor is None:
            _u, sv, v = np.linalg.svd(self.cov)
            self._mvn_factor = np.sqrt(sv)[:, None] * v
        return np.stack([np.dot(self._random[b].standard_normal(self.dim).reshape(-1, self.dim), self._mvn_factor)[0] + p[b]
                         for b in rows])

    def sample_streamed(self, p0, submit, fetch, groups=2, lnprob0=None, thin=1, storechain=True, iterations=1):
        """The chains of ``sample`` with the evaluation split into ``submit(P_rows, g) -> token`` and
        ``fetch(token) -> lnprob_rows``: the B chains form ``groups`` sub-ensembles (red / black halves for two), and while
        one group's accept / reject is decided and its next proposals are drawn on the host, the other groups' matrices
        keep the device busy -- the back-to-back iterations of /root/reference/psoap/sample_parallel.py:434-438 through ONE
        resident launch (``ChunkHandle.stream_*``, ``Posterior.stream_*``).

        No chain's proposal depends on another chain, and every chain still consumes its own random stream in the order
        propose, (uniform when dlnp < 0), propose, ...: given the same log-probabilities the chains are those of ``sample``
        and of B scalar ``MHSampler`` runs, bit for bit.  Yields ``(p, lnprob)`` per iteration like ``sample``."""
        B = self.n_chains
        groups = int(groups)
        if groups < 1 or B % groups:
            raise ValueError("n_chains must be a multiple of groups")
        rows = [range(g * B // groups, (g + 1) * B // groups) for g in range(groups)]
        p = np.array(np.broadcast_to(np.asarray(p0, dtype=np.float64), (B, self.dim)))
        if lnprob0 is None:
            tok = [submit(p[r.start:r.stop], g) for g, r in enumerate(rows)]
            lnprob = np.concatenate([np.asarray(fetch(t), dtype=np.float64) for t in tok])
        else:
            lnprob = np.array(np.broadcast_to(lnprob0, (B,)), dtype=np.float64)
        if storechain:
            n = int(iterations / thin)
            self._chain = np.concatenate((self._chain, np.zeros((B, n, self.dim))), axis=1)
            self._lnprob = np.concatenate((self._lnprob, np.zeros((B, n))), axis=1)
        i0 = self.iterations
        n_iter = int(iterations)
        if n_iter < 1:
            return
        q = np.empty_like(p)
        tokens = [None] * groups
        for g, r in enumerate(rows):                       # iteration 0 of every group goes out
            q[r.start:r.stop] = self._propose_rows(p, r)
            tokens[g] = submit(q[r.start:r.stop], g)
        for i in range(n_iter):
            self.iterations += 1
            for g, r in enumerate(rows):
                new = np.asarray(fetch(tokens[g]), dtype=np.float64)
                if new.shape != (len(r),):
                    raise ValueError(f"fetch returned shape {new.shape}, expected ({len(r)},)")
                for j, b in enumerate(r):
                    diff = new[j] - lnprob[b]
                    if diff < 0:
                        diff = np.exp(diff) - self._random[b].rand()
                    if diff > 0:
                        p[b] = q[b]
                        lnprob[b] = new[j]
                        self.naccepted[b] += 1
                if i + 1 < n_iter:                         # the group's next proposals leave before the next group is looked at
                    q[r.start:r.stop] = self._propose_rows(p, r)
                    tokens[g] = submit(q[r.start:r.stop], g)
            if storechain and i % thin == 0:
                ind = i0 + int(i / thin)
                self._chain[:, ind, :] = p
                self._lnprob[:, ind] = lnprob
            yield p.copy(), lnprob.copy()


def gelman_rubin(samplelist):
    """Split-chain Gelman-Rubin statistics of several flatchains (BDA3 p. 284), as
    /root/reference/psoap/utils.py:98-160 and scripts/psoap_gelman_rubin.py:20-80 compute them.

    Returns ``(mean, std_hat, R_hat)``, each (n_params,).  Chains must share an even length.
    """
    full_iterations = len(samplelist[0])
    assert full_iterations % 2 == 0, "Number of iterations must be even. Try cutting off a different number of burn in samples."
    shape = samplelist[0].shape
    for flatchain in samplelist:
        assert len(flatchain) == full_iterations, "Not all chains have the same number of iterations!"
        assert flatchain.shape == shape, "Not all flatchains have the same shape!"
    n = full_iterations // 2
    m = 2 * len(samplelist)
    halves = []
    for flatchain in samplelist:
        halves += [np.asarray(flatchain[:n], dtype=np.float64), np.asarray(flatchain[n:], dtype=np.float64)]
    chains = np.stack(halves, axis=1)                       # (n, m, n_params)
    chain_mean = chains.mean(axis=0)
    grand_mean = chains.mean(axis=(0, 1))
    between = n / (m - 1.0) * np.sum((chain_mean - grand_mean) ** 2, axis=0)
    within = np.mean(np.sum((chains - chain_mean) ** 2, axis=0) / (n - 1.0), axis=0)
    var_hat = (n - 1.0) / n * within + between / n
    return grand_mean, np.sqrt(var_hat), np.sqrt(var_hat / within)
