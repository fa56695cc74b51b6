"""GPU: one workgroup per matrix (psoap_amd/csrc/solo_kernel.hpp, PSOAP_SOLO=1) -- the path for many small matrices --
against the reference's golden values, the oracle at edge sizes, the graph kernel on the same inputs, heterogeneous group
launches, bit-reproducibility and the -inf conventions."""
import numpy as np
import pytest

from psoap_amd import synthetic as syn

pytestmark = pytest.mark.gpu
LNP_RTOL = 1e-10


def lnp_close(got, want):
    return np.all(np.abs(np.asarray(got) - np.asarray(want)) <= LNP_RTOL * np.maximum(1.0, np.abs(want)))


@pytest.fixture()
def solo(monkeypatch):
    monkeypatch.setenv("PSOAP_SOLO", "1")


def test_solo_reference_goldens(golden, solo):
    """cfg1 (N = 2000), cfg2 (N = 4096) and the small golden cases: values recorded from the reference itself"""
    from psoap_amd.chunk import ChunkHandle
    names = [str(n) for n in golden["lnlike_names"]]
    meta = golden["lnlike_meta"]
    vals = golden["lnlike_vals"]
    seen = 0
    for name, (c, ne, npx, seed, mfrac10, N), val in zip(names, meta, vals):
        if int(N) > 4096:
            continue
        ch = syn.make_chunk(int(c), int(ne), int(npx), seed=int(seed), masked_fraction=int(mfrac10) / 100.0)
        with ChunkHandle(ch.fl, ch.sigma, max_batch=1) as h:
            got = h.lnlike(ch.lwls, syn.GP_BASE[int(c)])
        assert lnp_close(got, val), (name, got, val)
        seen += 1
    assert seen >= 2


@pytest.mark.parametrize("N", [1, 2, 63, 127, 128, 129, 255, 257, 640, 1008])
def test_solo_vs_oracle_edge_sizes(oracle, solo, N):
    from psoap_amd.chunk import ChunkHandle
    ch = syn.make_chunk(2, 1, N, seed=4000 + N)
    with ChunkHandle(ch.fl, ch.sigma, max_batch=1) as h:
        got = h.lnlike(ch.lwls, syn.GP_BASE[2])
    want = oracle.lnlike(ch.lwls, ch.fl, ch.sigma, syn.GP_BASE[2])
    assert lnp_close(got, want), (N, got, want)


def test_solo_batch_group_and_graph_agree(oracle, monkeypatch):
    """a heterogeneous group launch (five chunk sizes x six walkers, more matrices than ... one launch) through both
    kernels: equal within the contract, each bit-identical to itself run after run, -inf where the reference says so"""
    from psoap_amd.chunk import ChunkGroup, ChunkHandle
    chunks = [syn.make_chunk(2, 3 + k, 90 + 17 * k, seed=50 + k) for k in range(5)]
    gp6 = syn.make_walkers(2, 6, seed=9)
    gp6[2, 0] = -0.1                       # a negative amplitude: -inf (covariance.py:336-337)
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("PSOAP_SOLO", mode)
        hs = [ChunkHandle(c_.fl, c_.sigma, max_batch=6) for c_ in chunks]
        with ChunkGroup(hs) as g:
            outs = []
            for _ in range(3):
                for h, c_ in zip(hs, chunks):
                    h.upload(np.repeat(c_.lwls[None], 6, axis=0), gp6)
                g.eval()
                outs.append(np.stack([h.fetch() for h in hs]))
        for h in hs:
            h.close()
        assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
        res[mode] = outs[0]
    assert np.all(np.isneginf(res["1"][:, 2])) and np.all(np.isneginf(res["0"][:, 2]))
    ok = np.ones(6, bool)
    ok[2] = False
    assert lnp_close(res["1"][:, ok], res["0"][:, ok])
    for k, c_ in enumerate(chunks):
        want = oracle.lnlike(c_.lwls, c_.fl, c_.sigma, list(gp6[0]))
        assert lnp_close(res["1"][k, 0], want), (k, res["1"][k, 0], want)


def test_solo_not_positive_definite_and_many_matrices(oracle, solo):
    """more matrices than workgroup slots (the ticket counter hands the rest out), one of them not positive definite"""
    from psoap_amd.chunk import ChunkHandle
    ch = syn.make_chunk(1, 2, 100, seed=77)          # N = 200
    B = 1100
    gps = syn.make_walkers(1, B, seed=3)
    lw = np.repeat(ch.lwls[None], B, axis=0).copy()
    with ChunkHandle(ch.fl, np.zeros_like(ch.sigma), max_batch=B) as h0:
        lw_bad = lw[:4].copy()
        lw_bad[:, :, 1] = lw_bad[:, :, 0]            # two identical pixels and no noise: singular
        assert np.all(np.isneginf(h0.lnlike_batch(lw_bad, gps[:4])))
    with ChunkHandle(ch.fl, ch.sigma, max_batch=B) as h:
        got = h.lnlike_batch(lw, gps)
        again = h.lnlike_batch(lw, gps)
    assert np.array_equal(got, again)
    for w in (0, 513, B - 1):
        assert lnp_close(got[w], oracle.lnlike(lw[w], ch.fl, ch.sigma, list(gps[w])))
