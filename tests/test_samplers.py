"""CPU: the Metropolis-Hastings drivers against a plain restatement of the reference's loop
(oracle/sampler_oracle.py, /root/reference/psoap/samplers.py:103-159), priors, Gelman-Rubin and the
multi-chunk driver with the CPU oracle standing in for the per-chunk device worker."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

from psoap_amd import priors, samplers, utils  # noqa: E402
from psoap_amd import synthetic as syn  # noqa: E402


def _gauss(icov):
    return lambda p: -0.5 * float(p @ icov @ p)


def test_mh_sampler_matches_oracle_loop():
    import sampler_oracle
    cov_t = np.array([[1.0, 0.6], [0.6, 2.0]])
    lnp = _gauss(np.linalg.inv(cov_t))
    jump = 0.8 * np.eye(2)
    s = samplers.MHSampler(jump, 2, lnp)
    rs = np.random.mtrand.RandomState(11)
    last = s.run_mcmc(np.zeros(2), 400, rstate0=rs.get_state())
    chain, lps, acc = sampler_oracle.mh_chain(lnp, np.zeros(2), jump, 400, np.random.mtrand.RandomState(11))
    assert np.array_equal(s.chain, chain) and np.array_equal(s.lnprobability, lps)
    assert s.naccepted == acc and s.iterations == 400 and 0.2 < s.acceptance_fraction < 0.9
    assert np.array_equal(last[0], chain[-1]) and last[1] == lps[-1]
    assert s.flatchain is s.chain
    # continuing a run appends (emcee semantics), thinning stores every thin-th state
    s.run_mcmc(last[0], 100, lnprob0=last[1], thin=5)
    assert s.chain.shape == (420, 2) and s.iterations == 500
    s.reset()
    assert s.chain.shape == (0, 2) and s.iterations == 0 and s.naccepted == 0


def test_mh_sampler_callbacks_and_inf():
    calls = {"acc": 0, "rej": 0}
    lnp = lambda p: -np.inf if p[0] < 0 else -0.5 * float(p @ p)       # noqa: E731
    s = samplers.MHSampler(np.eye(1) * 4.0, 1, lnp, acceptfn=lambda: calls.__setitem__("acc", calls["acc"] + 1),
                           rejectfn=lambda: calls.__setitem__("rej", calls["rej"] + 1))
    s.run_mcmc(np.array([1.0]), 300, rstate0=np.random.mtrand.RandomState(5).get_state())
    assert (s.chain >= 0).all()                      # -inf proposals are never accepted
    assert calls["acc"] == s.naccepted and calls["acc"] + calls["rej"] == 300
    # both -inf: nan difference -> reject, chain stays put
    s2 = samplers.MHSampler(np.eye(2), 2, lambda p: -np.inf)
    s2.run_mcmc(np.zeros(2), 10)
    assert s2.naccepted == 0 and np.array_equal(s2.chain, np.zeros((10, 2)))


def test_incremental_save(tmp_path):
    s = samplers.MHSampler(np.eye(2), 2, lambda p: -0.5 * float(p @ p))
    bk = str(tmp_path / "chain_backup.npy")
    list(s.sample(np.zeros(2), iterations=25, incremental_save=10, backup=bk))
    saved = np.load(bk)
    assert saved.shape == (25, 2) and np.array_equal(saved[:20], s.chain[:20])     # last write at iteration 20
    m = samplers.MultiChainMHSampler(np.eye(2), 2, lambda P: -0.5 * np.sum(P * P, axis=1), 3, seeds=[1, 2, 3])
    list(m.sample(np.zeros(2), iterations=12, incremental_save=5, backup=bk))
    assert np.load(bk).shape == (3, 12, 2)


def test_mh_statistics_gaussian_target():
    cov_t = np.array([[1.0, -0.4], [-0.4, 0.5]])
    s = samplers.MHSampler(1.2 * cov_t, 2, _gauss(np.linalg.inv(cov_t)))
    s.run_mcmc(np.zeros(2), 40000, rstate0=np.random.mtrand.RandomState(2).get_state())
    c = s.chain[2000:]
    assert np.all(np.abs(c.mean(axis=0)) < 0.06)
    assert np.allclose(np.cov(c.T), cov_t, atol=0.07)


def test_multichain_equals_independent_scalar_chains():
    cov_t = np.diag([1.0, 4.0, 0.25])
    icov = np.linalg.inv(cov_t)
    calls = []

    def batch(P):
        calls.append(P.shape)
        return np.array([-0.5 * p @ icov @ p for p in P])

    B, n = 5, 150
    jump = 0.5 * cov_t
    m = samplers.MultiChainMHSampler(jump, 3, batch, B, seeds=[100 + b for b in range(B)])
    p0 = np.array([0.1, -0.2, 0.3])
    m.run_mcmc(p0, n)
    assert m.chain.shape == (B, n, 3) and m.lnprobability.shape == (B, n)
    assert calls == [(B, 3)] * (n + 1)               # ONE batched evaluation per iteration (+ the start)
    for b in range(B):
        s = samplers.MHSampler(jump, 3, _gauss(icov))
        s.run_mcmc(p0, n, rstate0=np.random.mtrand.RandomState(100 + b).get_state())
        assert np.array_equal(m.chain[b], s.chain), b
        assert np.allclose(m.lnprobability[b], s.lnprobability, rtol=0, atol=1e-13)
        assert m.naccepted[b] == s.naccepted
    assert m.flatchain.shape == (B * n, 3) and np.array_equal(m.flatchain[:n], m.chain[0])
    assert not np.array_equal(m.chain[0], m.chain[1])
    with pytest.raises(ValueError):
        samplers.MultiChainMHSampler(jump, 3, lambda P: np.zeros(2), B).run_mcmc(p0, 1)
    with pytest.raises(ValueError):
        samplers.MultiChainMHSampler(jump, 3, batch, B, seeds=[1, 2])


def test_gelman_rubin():
    rng = np.random.RandomState(8)
    chains = [rng.standard_normal((1000, 3)) * [1.0, 2.0, 0.5] + [0.0, 5.0, -1.0] for _ in range(4)]
    mean, std, R = samplers.gelman_rubin(chains)
    assert np.allclose(mean, [0.0, 5.0, -1.0], atol=0.1) and np.allclose(std, [1.0, 2.0, 0.5], rtol=0.06)
    assert np.all(np.abs(R - 1.0) < 0.01)
    # explicit BDA3 formulas on the split chains
    n, m = 500, 8
    split = np.stack([h for c in chains for h in (c[:n], c[n:])], axis=1)
    W = split.var(axis=0, ddof=1).mean(axis=0)
    Bv = n * split.mean(axis=0).var(axis=0, ddof=1)
    assert np.allclose(R, np.sqrt(((n - 1) / n * W + Bv / n) / W), rtol=1e-12)
    chains[2] = chains[2] + [3.0, 0.0, 0.0]          # one chain stuck elsewhere
    assert samplers.gelman_rubin(chains)[2][0] > 1.3
    with pytest.raises(AssertionError):
        samplers.gelman_rubin([c[:999] for c in chains])


def test_estimate_covariance_round_trip(tmp_path):
    # the reference's resume workflow: flatchain -> opt_jump.npy -> proposal covariance of the next run
    rng = np.random.RandomState(3)
    true = np.array([[1.0, 0.3], [0.3, 0.5]])
    chain = rng.multivariate_normal([0, 0], true, size=20000)
    cov = utils.estimate_covariance(chain)
    assert np.allclose(cov, 2.38 ** 2 / 2 * true, rtol=0.05, atol=0.02)
    assert np.array_equal(utils.estimate_covariance(chain, ndim=4), 2.38 ** 2 / 4 * np.cov(chain, rowvar=0))


def test_default_priors():
    # sample_parallel.py:330-358: strict inequalities, bounds themselves allowed
    p = dict(q=0.5, K=10.0, e=0.0, omega=-90.0, P=5.0, T0=-3.0, gamma=-20.0, amp_f=0.0, l_f=1.0, amp_g=0.1, l_g=2.0)
    full = np.array([[p[n] for n in utils.registered_params["SB2"]]])
    assert priors.box_prior_full("SB2", full)[0] == 0.0
    for name, bad in (("q", -0.1), ("K", -1.0), ("e", -1e-9), ("e", 1.0000001), ("P", -2.0), ("omega", -90.1),
                      ("omega", 450.5), ("amp_f", -0.1), ("l_f", -1.0), ("amp_g", -0.1), ("l_g", -3.0)):
        x = full.copy()
        x[0, utils.registered_params["SB2"].index(name)] = bad
        assert priors.box_prior_full("SB2", x)[0] == -np.inf, name
    for name, okv in (("e", 1.0), ("omega", 450.0), ("T0", -1e9), ("gamma", -1e4)):
        x = full.copy()
        x[0, utils.registered_params["SB2"].index(name)] = okv
        assert priors.box_prior_full("SB2", x)[0] == 0.0, name
    reg3 = utils.registered_params["ST3"]
    f3 = np.tile([0.5, 5.0, 0.1, 10.0, 3.0, 0.0, 0.4, 2.0, 0.2, 20.0, 100.0, 1.0, 0.0, 0.1, 5.0, 0.1, 5.0, 0.1, 5.0], (3, 1))
    f3[1, reg3.index("e_out")] = 1.2
    f3[2, reg3.index("q_out")] = -0.2
    assert priors.box_prior_full("ST3", f3).tolist() == [0.0, -np.inf, -np.inf]
    # fitted-vector form with fixed parameters filled from the configuration
    pr = priors.make_prior("SB2", ["gamma"], **p)
    fit = np.array([[p[n] for n in utils.registered_params["SB2"] if n != "gamma"]] * 2)
    fit[1, 0] = -1.0
    assert pr(fit).tolist() == [0.0, -np.inf]
    assert priors.make_prior("SB2", ["gamma", "e"], **dict(p, e=1.5))(np.delete(fit, 1, axis=1)[:1])[0] == -np.inf


def test_user_prior_file(tmp_path):
    assert priors.load_user_prior(str(tmp_path)) is None
    (tmp_path / "prior.py").write_text("import numpy as np\ndef prior(p):\n    return -np.inf if p[0] > 1 else -p[0]\n")
    pr = priors.load_user_prior(str(tmp_path))
    assert pr(np.array([[0.5, 0.0], [2.0, 0.0]])).tolist() == [-0.5, -np.inf]


# ---------------------------------------------------------------- the multi-chunk driver ------------
CONFIG = dict(model="SB2", epoch_limit=20, soften=1.0, samples=6, opt_jump="missing.npy", fix_params=["gamma"],
              parameters=dict(q=0.6, K=25.0, e=0.1, omega=30.0, P=12.0, T0=2455010.0, gamma=3.0,
                              amp_f=0.2, l_f=6.0, amp_g=0.1, l_g=8.0),
              jumps=dict(q=0.01, K=0.2, e=0.01, omega=0.5, P=0.01, T0=0.05, gamma=0.01,
                         amp_f=0.01, l_f=0.3, amp_g=0.01, l_g=0.3))


class _Chunk:
    def __init__(self, s):
        self.lwl, self.fl, self.sigma, self.epoch_index, self.date1D = s.lwl, s.fl, s.sigma, s.epoch_index, s.dates


def _chunks(n=3):
    return [_Chunk(syn.make_chunk(2, 5, 12, seed=700 + k, masked_fraction=0.1 if k else 0.0)) for k in range(n)]


class _OracleWorker:
    """Per-chunk lnprob(p) on the CPU oracle, standing in for the device ChunkWorker."""

    def __init__(self, ch, config):
        self.ch, self.config = ch, config

    def lnprob_batch(self, P):
        import sampler_oracle
        orb, gp = utils.convert_vectors(P, self.config["model"], self.config["fix_params"], **self.config["parameters"])
        c = self.ch
        return np.array([sampler_oracle.chunk_lnprob(self.config["model"], orb[i], gp[i], c.lwl, c.fl,
                                                     c.sigma * self.config["soften"], c.epoch_index, c.date1D)
                         for i in range(len(P))])


def _scalar_posterior(chunks, config):
    pr = priors.make_prior(config["model"], config["fix_params"], **config["parameters"])
    workers = [_OracleWorker(c, config) for c in chunks]

    def lnprob(p):
        lp = pr(p[None, :])[0]
        if lp == -np.inf:
            return -np.inf
        tot = 0.0
        for w in workers:
            tot = tot + w.lnprob_batch(p[None, :])[0]
        return tot + lp
    return lnprob


def test_driver_single_process(tmp_path):
    import sampler_oracle
    from psoap_amd import sample_parallel as sp
    config = dict(CONFIG, outdir=str(tmp_path / "output"))
    chunks = _chunks()
    s = sp.run(config, chunks, run_index=3, n_chains=4, seed=40, make_worker=lambda ch: _OracleWorker(ch, config),
               verbose=False)
    assert s.chain.shape == (4, 6, 10)
    lnprob = _scalar_posterior(chunks, config)
    p0 = utils.convert_dict("SB2", ["gamma"], **config["parameters"])
    cov = utils.convert_dict("SB2", ["gamma"], **config["jumps"]) ** 2 * np.eye(10)
    for b in range(4):
        chain, lps, acc = sampler_oracle.mh_chain(lnprob, p0, cov, 6, np.random.mtrand.RandomState(40 + b))
        assert np.array_equal(s.chain[b], chain) and np.allclose(s.lnprobability[b], lps, rtol=1e-13, atol=0)
        d = str(tmp_path / "output" / "run{:02d}".format(3 + b))
        assert np.array_equal(np.load(d + "/flatchain.npy"), chain)
        assert np.array_equal(np.load(d + "/lnprob.npy"), s.lnprobability[b])
    assert s.naccepted.sum() > 0
    # opt_jump covariance file is honoured (sample_parallel.py:425-427)
    np.save(str(tmp_path / "opt.npy"), 1e-6 * np.eye(10))
    cfg2 = dict(config, opt_jump=str(tmp_path / "opt.npy"))
    s2 = sp.run(cfg2, chunks, n_chains=2, seed=1, make_worker=lambda ch: _OracleWorker(ch, cfg2), verbose=False)
    assert np.abs(s2.chain - p0).max() < 0.05
    # run00 / run01 exist now: a second invocation must not silently wipe them
    with pytest.raises(FileExistsError, match="run00"):
        sp.run(cfg2, chunks, n_chains=2, seed=1, make_worker=lambda ch: _OracleWorker(ch, cfg2), verbose=False)
    keep = np.load(str(tmp_path / "output" / "run03" / "flatchain.npy"))
    sp.run(cfg2, chunks, n_chains=2, seed=2, make_worker=lambda ch: _OracleWorker(ch, cfg2), verbose=False, overwrite=True)
    assert np.array_equal(np.load(str(tmp_path / "output" / "run03" / "flatchain.npy")), keep)   # untouched
    # a starting point outside the prior aborts (sample_parallel.py:405-419)
    bad = dict(config, parameters=dict(config["parameters"], K=-1.0))
    with pytest.raises(RuntimeError, match="-np.inf"):
        sp.run(bad, chunks, n_chains=2, seed=1, make_worker=lambda ch: _OracleWorker(ch, bad), verbose=False,
               overwrite=True)


def test_posterior_keeps_batch_size_and_masks_rows_outside_prior():
    """Prior-rejected rows are -inf without their parameters ever reaching a worker, and the device batch
    keeps its size (a stand-in proposal fills the slot), so the task list and the split factors -- hence the
    last bits of the other chains -- do not depend on how many rows were rejected."""
    from psoap_amd import sample_parallel as sp
    seen = []

    class W:
        def lnprob_batch(self, P):
            seen.append(len(P))
            assert np.all(P[:, 1] > 0.0) and np.all(P[:, 2] < 1.0)      # only admissible proposals arrive
            return np.full(len(P), -1.5)

    post = sp.Posterior("SB2", _chunks(2), ["gamma"], CONFIG["parameters"], max_batch=2, make_worker=lambda ch: W())
    P = np.tile(utils.convert_dict("SB2", ["gamma"], **CONFIG["parameters"]), (5, 1))
    P[1, 1] = -3.0                                    # K < 0
    P[4, 2] = 1.5                                     # e > 1
    out = post.lnprob_batch(P)
    assert out.tolist() == [-3.0, -np.inf, -3.0, -3.0, -np.inf]
    assert seen == [2, 2, 2, 2, 1, 1]                 # all 5 slots per chunk, in max_batch pieces (piece-major)
    seen.clear()
    assert post.lnprob_batch(P[[1, 4]]).tolist() == [-np.inf, -np.inf] and seen == []


def _free_port():
    """Kept as the third argument of the rank functions; the rendezvous itself goes through a file store in the test's
    temporary directory (no port to probe and lose between probe and bind)."""
    return 0


def _rank_main(rank, world, port, outdir):
    os.environ["GLOO_SOCKET_IFNAME"] = "lo"
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import test_samplers as me
    from psoap_amd import sample_parallel as sp
    dist.init_process_group("gloo", init_method="file://" + os.path.join(outdir, "rendezvous"), rank=rank, world_size=world)
    try:
        config = dict(me.CONFIG, outdir=os.path.join(outdir, "output"))
        built = []

        def make(ch):
            built.append(1)
            return me._OracleWorker(ch, config)

        s = sp.run(config, me._chunks(3), n_chains=3, seed=9, world=world, rank=rank, make_worker=make, verbose=False)
        np.save(os.path.join(outdir, f"chain_{rank}.npy"), s.chain)
        np.save(os.path.join(outdir, f"built_{rank}.npy"), np.array(len(built)))
    finally:
        dist.destroy_process_group()


def test_driver_two_ranks_gloo(tmp_path):
    import torch.multiprocessing as mp
    from psoap_amd import sample_parallel as sp
    port = _free_port()
    mp.spawn(_rank_main, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    c0, c1 = np.load(tmp_path / "chain_0.npy"), np.load(tmp_path / "chain_1.npy")
    assert np.array_equal(c0, c1)                      # same proposals, same decisions on every rank
    assert int(np.load(tmp_path / "built_0.npy")) == 2 and int(np.load(tmp_path / "built_1.npy")) == 1
    config = dict(CONFIG, outdir=str(tmp_path / "single"))
    s = sp.run(config, _chunks(3), n_chains=3, seed=9, make_worker=lambda ch: _OracleWorker(ch, config), verbose=False)
    assert np.array_equal(s.chain, c0)                 # and identical to the single-process run
    assert os.path.exists(tmp_path / "output" / "run00" / "flatchain.npy")
    with pytest.raises(ValueError, match="seed"):
        sp.run(config, _chunks(3), n_chains=2, world=2, rank=0, make_worker=lambda ch: None, verbose=False)


def _rank_outdir_clash(rank, world, port, outdir):
    os.environ["GLOO_SOCKET_IFNAME"] = "lo"
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import test_samplers as me
    from psoap_amd import sample_parallel as sp
    dist.init_process_group("gloo", init_method="file://" + os.path.join(outdir, "rendezvous"), rank=rank, world_size=world)
    try:
        # the run directory exists on rank 0's side only: rank 1 looks at a private, empty output tree
        config = dict(me.CONFIG, outdir=os.path.join(outdir, "output" if rank == 0 else "elsewhere"))
        verdict = "ran"
        try:
            sp.run(config, me._chunks(3), n_chains=1, seed=3, world=world, rank=rank, iterations=2,
                   make_worker=lambda ch: me._OracleWorker(ch, config), verbose=False)
        except FileExistsError:
            verdict = "FileExistsError"
        with open(os.path.join(outdir, f"verdict_{rank}.txt"), "w") as fh:
            fh.write(verdict)
    finally:
        dist.destroy_process_group()


def test_existing_output_directory_raises_on_every_rank_together(tmp_path):
    """the check used to run on each rank on its own, before any collective -- a rank that raised alone left
    the others hanging in the first gather.  Rank 0 decides, the verdict is broadcast."""
    import torch.multiprocessing as mp
    os.makedirs(tmp_path / "output" / "run00")
    mp.spawn(_rank_outdir_clash, args=(2, 0, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "verdict_0.txt").read_text() == "FileExistsError"
    assert (tmp_path / "verdict_1.txt").read_text() == "FileExistsError"


# ---------------------------------------------------------------- streamed chains (round 5) ------------
def test_streamed_sampler_equals_lockstep_and_scalar_chains():
    """samplers.sample_streamed: the B chains in two halves through a split-phase evaluator -- same draws, same decisions,
    same chains as the lock-step loop and as B scalar MHSampler runs; and the half that was just decided is resubmitted
    BEFORE the other half is looked at (that is what keeps a resident launch busy)."""
    cov_t = np.diag([1.0, 4.0, 0.25])
    icov = np.linalg.inv(cov_t)
    log = []

    def submit(P, g):
        log.append(("submit", g, len(P)))
        return np.array(P)

    def fetch(tok):
        log.append(("fetch", len(tok)))
        return np.array([-0.5 * p @ icov @ p for p in tok])

    B, n = 6, 120
    jump = 0.5 * cov_t
    p0 = np.array([0.1, -0.2, 0.3])
    seeds = [300 + b for b in range(B)]
    m = samplers.MultiChainMHSampler(jump, 3, None, B, seeds=seeds)
    out = list(m.sample_streamed(p0, submit, fetch, groups=2, iterations=n))
    assert len(out) == n and m.iterations == n and m.chain.shape == (B, n, 3)
    ref = samplers.MultiChainMHSampler(jump, 3, lambda P: np.array([-0.5 * p @ icov @ p for p in P]), B, seeds=seeds)
    ref.run_mcmc(p0, n)
    assert np.array_equal(m.chain, ref.chain) and np.array_equal(m.lnprobability, ref.lnprobability)
    assert np.array_equal(m.naccepted, ref.naccepted)
    for b in (0, B - 1):
        s = samplers.MHSampler(jump, 3, _gauss(icov))
        s.run_mcmc(p0, n, rstate0=np.random.mtrand.RandomState(seeds[b]).get_state())
        assert np.array_equal(m.chain[b], s.chain)
    # order of calls: start (2 submits, 2 fetches), prime (2 submits), then per iteration fetch g, submit g, fetch g', ...
    assert log[:6] == [("submit", 0, 3), ("submit", 1, 3), ("fetch", 3), ("fetch", 3), ("submit", 0, 3), ("submit", 1, 3)]
    body = log[6:]
    assert body[:4] == [("fetch", 3), ("submit", 0, 3), ("fetch", 3), ("submit", 1, 3)]
    assert body[-2:] == [("fetch", 3), ("fetch", 3)]                    # the last iteration resubmits nothing
    assert sum(1 for e in log if e[0] == "submit") == 2 * (n + 1) and sum(1 for e in log if e[0] == "fetch") == 2 * (n + 1)
    # continuing appends; three groups; a given starting value skips the first evaluation
    more = samplers.MultiChainMHSampler(jump, 3, None, B, seeds=seeds)
    list(more.sample_streamed(p0, submit, fetch, groups=3, lnprob0=-0.5 * p0 @ icov @ p0, iterations=n))
    assert np.array_equal(more.chain, ref.chain)
    with pytest.raises(ValueError):
        list(samplers.MultiChainMHSampler(jump, 3, None, 5, seeds=[1] * 5).sample_streamed(p0, submit, fetch, groups=2))


class _StreamOracleWorker(_OracleWorker):
    """the stream entry points of ChunkWorker on the CPU oracle (tickets = the proposals themselves)"""
    opened = 0

    def stream_open(self, lanes=None, scheme=-1):
        type(self).opened += 1

    def stream_submit(self, P, mu_GP=1.0):
        return np.array(P)

    def stream_fetch(self, tickets):
        return self.lnprob_batch(tickets)

    def stream_close(self):
        type(self).opened -= 1


def test_driver_streamed_equals_lockstep(tmp_path):
    """run(stream=True): same chains as the lock-step driver and as the oracle's scalar loop; prior-rejected rows never
    reach the worker; several chunks on a rank refuse to stream."""
    import sampler_oracle
    from psoap_amd import sample_parallel as sp
    config = dict(CONFIG, outdir=str(tmp_path / "a"), samples=8)
    chunks = _chunks(1)
    s = sp.run(config, chunks, n_chains=4, seed=40, make_worker=lambda ch: _StreamOracleWorker(ch, config), verbose=False,
               stream=True)
    assert s.streamed and _StreamOracleWorker.opened == 0
    lnprob = _scalar_posterior(chunks, config)
    p0 = utils.convert_dict("SB2", ["gamma"], **config["parameters"])
    cov = utils.convert_dict("SB2", ["gamma"], **config["jumps"]) ** 2 * np.eye(10)
    for b in range(4):
        chain, lps, _ = sampler_oracle.mh_chain(lnprob, p0, cov, 8, np.random.mtrand.RandomState(40 + b))
        assert np.array_equal(s.chain[b], chain) and np.allclose(s.lnprobability[b], lps, rtol=1e-13, atol=0)
    config_b = dict(config, outdir=str(tmp_path / "b"))
    s2 = sp.run(config_b, chunks, n_chains=4, seed=40, make_worker=lambda ch: _StreamOracleWorker(ch, config_b), verbose=False,
                stream=False)
    assert not s2.streamed and np.array_equal(s2.chain, s.chain)
    # the automatic rule: small chunks / few chains do not stream
    config_c = dict(config, outdir=str(tmp_path / "c"))
    assert not sp.run(config_c, chunks, n_chains=4, seed=40, make_worker=lambda ch: _StreamOracleWorker(ch, config_c),
                      verbose=False).streamed
    post = sp.Posterior("SB2", _chunks(2), ["gamma"], CONFIG["parameters"], max_batch=2,
                        make_worker=lambda ch: _StreamOracleWorker(ch, config))
    assert not post.can_stream()
    with pytest.raises(RuntimeError, match="one chunk"):
        post.stream_open()
    # prior-rejected rows: -inf, a stand-in goes to the worker
    post1 = sp.Posterior("SB2", chunks, ["gamma"], CONFIG["parameters"], max_batch=4,
                         make_worker=lambda ch: _StreamOracleWorker(ch, config))
    post1.stream_open()
    P = np.tile(p0, (3, 1))
    P[1, 1] = -3.0
    tok = post1.stream_submit(P, 0)
    assert np.all(tok[0][:, 1] > 0.0)
    out = post1.stream_fetch(tok)
    assert out[1] == -np.inf and out[0] == out[2] == lnprob(p0)
    assert post1.stream_fetch(post1.stream_submit(P[[1]], 0)).tolist() == [-np.inf]
    post1.stream_close()


def _rank_streamed(rank, world, port, outdir):
    os.environ["GLOO_SOCKET_IFNAME"] = "lo"
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import test_samplers as me
    from psoap_amd import sample_parallel as sp
    dist.init_process_group("gloo", init_method="file://" + os.path.join(outdir, "rendezvous"), rank=rank, world_size=world)
    try:
        config = dict(me.CONFIG, outdir=os.path.join(outdir, "output"), samples=5)
        s = sp.run(config, me._chunks(2), n_chains=4, seed=9, world=world, rank=rank,
                   make_worker=lambda ch: me._StreamOracleWorker(ch, config), verbose=False, stream=True)
        np.save(os.path.join(outdir, f"schain_{rank}.npy"), s.chain)
    finally:
        dist.destroy_process_group()


def test_driver_streamed_two_ranks_gloo(tmp_path):
    """one chunk per rank, every half gathered before it is resubmitted (sample_parallel.py:378-390): the chains of both
    ranks and of the single-process lock-step run are identical"""
    import torch.multiprocessing as mp
    from psoap_amd import sample_parallel as sp
    mp.spawn(_rank_streamed, args=(2, 0, str(tmp_path)), nprocs=2, join=True)
    c0, c1 = np.load(tmp_path / "schain_0.npy"), np.load(tmp_path / "schain_1.npy")
    assert np.array_equal(c0, c1)
    config = dict(CONFIG, outdir=str(tmp_path / "single"), samples=5)
    s = sp.run(config, _chunks(2), n_chains=4, seed=9, make_worker=lambda ch: _OracleWorker(ch, config), verbose=False)
    assert np.array_equal(s.chain, c0)
