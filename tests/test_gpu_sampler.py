"""GPU: the multi-chain Metropolis-Hastings driver on the device lnprob(p) boundary against the scalar
loop of the reference restated on the CPU oracle (oracle/sampler_oracle.py)."""
import os
import sys

import numpy as np
import pytest

from psoap_amd import synthetic as syn
from psoap_amd import utils

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
LNP_RTOL = 1e-8


def _write_dataset(tmp, n_chunks=2):
    """chunk_*.npz + chunks.dat + config.yaml, the files the reference's driver reads from its CWD."""
    import yaml
    from psoap_amd import data as pdata
    from test_samplers import CONFIG
    rows = []
    for k in range(n_chunks):
        s = syn.make_chunk(2, 8, 60, seed=800 + k, masked_fraction=0.15)
        full = lambda v, fill: np.where(s.mask, 0.0, fill) + _scatter(s.mask, v)      # noqa: E731
        wl = np.exp(full(s.lwl, 8.5))
        date = np.broadcast_to(s.dates[:, None], s.mask.shape).copy()
        pdata.Chunk(wl, full(s.fl, 1.0), full(s.sigma, 1.0), date, s.mask).save(20 + k, 5100.0 + k, 5110.0 + k,
                                                                                 prefix=str(tmp) + "/")
        rows.append((20 + k, 5100.0 + k, 5110.0 + k))
    pdata.write_chunk_table(str(tmp / "chunks.dat"), rows)
    config = dict(CONFIG, chunk_file=str(tmp / "chunks.dat"), outdir=str(tmp / "output"), samples=10)
    with open(tmp / "config.yaml", "w") as f:
        yaml.safe_dump(config, f)
    return config


def _scatter(mask, v):
    out = np.zeros(mask.shape)
    out[mask] = v
    return out


def test_device_chains_equal_oracle_chains(tmp_path):
    import sampler_oracle
    from psoap_amd import sample_parallel as sp
    from test_samplers import _scalar_posterior
    config = _write_dataset(tmp_path)
    chunks = sp.load_chunks(sp.load_config(str(tmp_path / "config.yaml")), prefix=str(tmp_path) + "/")
    assert [c.N for c in chunks] == [int(syn.make_chunk(2, 8, 60, seed=800 + k, masked_fraction=0.15).N) for k in range(2)]
    s = sp.run(config, chunks, run_index=0, n_chains=4, seed=21, verbose=False)
    lnprob = _scalar_posterior(chunks, config)
    p0 = utils.convert_dict("SB2", ["gamma"], **config["parameters"])
    cov = utils.convert_dict("SB2", ["gamma"], **config["jumps"]) ** 2 * np.eye(10)
    for b in range(4):
        chain, lps, acc = sampler_oracle.mh_chain(lnprob, p0, cov, 10, np.random.mtrand.RandomState(21 + b))
        assert np.array_equal(s.chain[b], chain), b            # same accept / reject decisions
        assert np.all(np.abs(s.lnprobability[b] - lps) <= LNP_RTOL * np.maximum(1.0, np.abs(lps)))
        assert s.naccepted[b] == acc
    assert 0 < s.naccepted.sum() < 40


def test_command_line_driver(tmp_path, monkeypatch):
    from psoap_amd import sample_parallel as sp
    _write_dataset(tmp_path)
    monkeypatch.chdir(tmp_path)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    sp.main(["5", "--chains", "3", "--seed", "4", "--prefix", str(tmp_path) + "/"])
    for b in range(3):
        d = tmp_path / "output" / "run{:02d}".format(5 + b)
        assert np.load(d / "flatchain.npy").shape == (10, 10) and np.load(d / "lnprob.npy").shape == (10,)
        assert (d / "config.yaml").exists()
    from psoap_amd import samplers
    mean, std, R = samplers.gelman_rubin([np.load(tmp_path / "output" / f"run{5 + b:02d}" / "flatchain.npy") for b in range(3)])
    assert mean.shape == (10,) and np.all(np.isfinite(mean))


def test_quickstart_example(tmp_path):
    """examples/quickstart.py end to end at a tiny size: files -> sampling -> R_hat -> opt_jump -> reconstruction."""
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import quickstart
    sampler, res = quickstart.main([str(tmp_path / "work"), "--chunks", "2", "--chains", "3", "--samples", "8"])
    assert sampler.chain.shape == (3, 8, 10)
    assert (tmp_path / "work" / "opt_jump.npy").exists()
    assert (tmp_path / "work" / "plots_chunk_20_5100_5110" / "f.npy").exists()
    assert np.all(np.isfinite(res["mu"]))


def test_streamed_device_chains_equal_oracle_and_lockstep_chains(tmp_path):
    """Round 5: run(stream=True) -- the sampler's iterations through ONE resident launch, the chains in two halves
    (samplers.sample_streamed, Posterior.stream_*; /root/reference/psoap/sample_parallel.py:434-438 with the gather of :378-387
    per half).  One chunk on the rank; same seeds: the chains of the oracle's scalar loop and of the lock-step driver."""
    import sampler_oracle
    from psoap_amd import sample_parallel as sp
    from test_samplers import _scalar_posterior
    config = _write_dataset(tmp_path, n_chunks=1)
    chunks = sp.load_chunks(sp.load_config(str(tmp_path / "config.yaml")), prefix=str(tmp_path) + "/")
    s = sp.run(config, chunks, run_index=0, n_chains=6, seed=31, verbose=False, stream=True)
    assert s.streamed
    lock = sp.run(config, chunks, run_index=0, n_chains=6, seed=31, verbose=False, stream=False, overwrite=True)
    assert not lock.streamed and np.array_equal(s.chain, lock.chain) and np.array_equal(s.naccepted, lock.naccepted)
    assert np.all(np.abs(s.lnprobability - lock.lnprobability) <= 1e-10 * np.maximum(1.0, np.abs(lock.lnprobability)))
    lnprob = _scalar_posterior(chunks, config)
    p0 = utils.convert_dict("SB2", ["gamma"], **config["parameters"])
    cov = utils.convert_dict("SB2", ["gamma"], **config["jumps"]) ** 2 * np.eye(10)
    for b in range(6):
        chain, lps, acc = sampler_oracle.mh_chain(lnprob, p0, cov, 10, np.random.mtrand.RandomState(31 + b))
        assert np.array_equal(s.chain[b], chain), b
        assert np.all(np.abs(s.lnprobability[b] - lps) <= LNP_RTOL * np.maximum(1.0, np.abs(lps)))
    # the automatic rule leaves small chunks / few chains on the launch-per-step path
    assert not sp.run(config, chunks, run_index=0, n_chains=6, seed=31, verbose=False, overwrite=True).streamed
    # a row outside the prior inside a streamed half: -inf, never accepted
    post = sp.Posterior("SB2", chunks, ["gamma"], config["parameters"], max_batch=4)
    post.stream_open()
    P = np.tile(p0, (4, 1))
    P[2, 1] = -1.0
    out = post.stream_fetch(post.stream_submit(P))
    assert out[2] == -np.inf and out[0] == out[1] == out[3] and np.isfinite(out[0])
    post.stream_close()
    post.close()
