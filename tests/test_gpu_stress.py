"""Randomised agreement of the two execution modes (persistent DAG kernel vs staged kernels) and the
oracle over many shapes and batch sizes; guards the in-kernel dependency protocol."""
import numpy as np
import pytest

from psoap_amd import synthetic as syn

pytestmark = pytest.mark.gpu
LNP_RTOL = 1e-10


def close(a, b):
    return abs(a - b) <= LNP_RTOL * max(1.0, abs(b))


def test_random_shapes_dag_vs_staged_vs_oracle(oracle):
    from psoap_amd.chunk import ChunkHandle
    rng = np.random.default_rng(2024)
    for it in range(36):
        c = int(rng.integers(1, 4))
        n_epochs = int(rng.integers(1, 6))
        n_pix = int(rng.integers(1, 330))
        B = int(rng.integers(1, 10))
        ch = syn.make_chunk(c, n_epochs, n_pix, seed=7000 + it, masked_fraction=0.15 if n_pix > 20 else 0.0)
        gps = syn.make_walkers(c, B, seed=8000 + it)
        lw = syn.walker_lwls(ch, syn.make_walker_velocities(ch, B, seed=9000 + it))
        with ChunkHandle(ch.fl, ch.sigma, max_batch=B) as h:
            h.set_mode("dag")
            dag = h.lnlike_batch(lw, gps)
            assert np.array_equal(h.lnlike_batch(lw, gps), dag)       # bit-reproducible
            h.set_mode("staged")
            staged = h.lnlike_batch(lw, gps)
        for w in range(B):
            assert close(dag[w], staged[w]), (it, ch.N, B, w, dag[w], staged[w])
        w = int(rng.integers(0, B))
        want = oracle.lnlike(lw[w], ch.fl, ch.sigma, gps[w])
        assert close(dag[w], want), (it, ch.N, B, w, dag[w], want)


def test_many_back_to_back_batches_same_handle():
    """Re-launching the persistent kernel on one handle many times (state re-zeroed every launch),
    alternating batch sizes so the task list is rebuilt."""
    from psoap_amd.chunk import ChunkHandle
    ch = syn.make_chunk(2, 4, 200, seed=99)          # N = 800
    gps = syn.make_walkers(2, 8, seed=3)
    lw = syn.walker_lwls(ch, syn.make_walker_velocities(ch, 8, seed=4))
    with ChunkHandle(ch.fl, ch.sigma, max_batch=8) as h:
        ref = h.lnlike_batch(lw, gps)
        first = {}
        for rep in range(40):
            B = 1 + rep % 8
            got = h.lnlike_batch(lw[:B], gps[:B])
            # bit for bit the same whenever the same batch size comes round again; across batch sizes the schedule
            # (throughput / latency / following scheme) and with it the order of the sums may differ: a few ulp
            assert np.array_equal(got, first.setdefault(B, got)), rep
            assert np.all(np.abs(got - ref[:B]) <= 1e-13 * np.abs(ref[:B])), rep


def test_not_positive_definite_inside_batch():
    """One singular matrix in a batch: that slot is -inf, the others are unaffected."""
    from psoap_amd.chunk import ChunkHandle
    ch = syn.make_chunk(2, 3, 100, seed=5)
    B = 4
    gps = np.tile(np.array(syn.GP_BASE[2]), (B, 1))
    lw = np.repeat(ch.lwls[None], B, axis=0).copy()
    lw[2, :, 1] = lw[2, :, 0]                        # duplicate pixel in proposal 2
    sigma = ch.sigma.copy()
    sigma[:2] = 0.0                                  # ... with zero noise there -> singular
    for mode in ("dag", "staged"):
        with ChunkHandle(ch.fl, sigma, max_batch=B) as h:
            h.set_mode(mode)
            got = h.lnlike_batch(lw, gps)
        assert got[2] == -np.inf
        assert np.isfinite(got[[0, 1, 3]]).all() and got[0] == got[1] == got[3]


def test_both_split_schemes_agree_with_oracle(oracle, monkeypatch):
    """The persistent kernel's two ways of cutting a tile's update (gathered partial tiles for
    throughput, chained partial sums + last-panel finals for latency) pinned one after the other."""
    from psoap_amd.chunk import ChunkHandle
    for c, ne, npx, B, seed in ((2, 5, 300, 3, 31), (3, 4, 260, 1, 32), (1, 9, 200, 11, 33)):
        ch = syn.make_chunk(c, ne, npx, seed=seed, masked_fraction=0.1)
        gps = syn.make_walkers(c, B, seed=seed + 100)
        lw = syn.walker_lwls(ch, syn.make_walker_velocities(ch, B, seed=seed + 200))
        got = {}
        for scheme in ("0", "1"):
            monkeypatch.setenv("PSOAP_DAG_SCHEME", scheme)
            with ChunkHandle(ch.fl, ch.sigma, max_batch=B) as h:
                got[scheme] = h.lnlike_batch(lw, gps)
                assert np.array_equal(h.lnlike_batch(lw, gps), got[scheme])
        for w in range(B):
            assert close(got["0"][w], got["1"][w]), (ch.N, w)
        want = oracle.lnlike(lw[B - 1], ch.fl, ch.sigma, gps[B - 1])
        assert close(got["0"][B - 1], want) and close(got["1"][B - 1], want), (ch.N, got, want)


def test_large_matrix_against_oracle(oracle):
    """N = 12288 (96 block rows, 1.2 GB per matrix): beyond every BASELINE shape, single and paired."""
    from psoap_amd.chunk import ChunkHandle
    ch = syn.make_chunk(2, 32, 384, seed=4242)
    assert ch.N == 12288
    gps = syn.make_walkers(2, 2, seed=4243)
    lw = syn.walker_lwls(ch, syn.make_walker_velocities(ch, 2, seed=4244))
    with ChunkHandle(ch.fl, ch.sigma, max_batch=2) as h:
        both = h.lnlike_batch(lw, gps)
        one = h.lnlike_batch(lw[:1], gps[:1])
    want = oracle.lnlike(lw[0], ch.fl, ch.sigma, gps[0])
    assert close(both[0], want) and close(one[0], want), (both, one, want)
    assert np.isfinite(both[1])


def test_smallest_matrices_one_to_four_block_rows(oracle):
    """N from 100 to 385 -- one, two, three and four 128-row blocks, with and without padding: the first block rows of the
    following scheme (a diagonal task that starts from a covariance-only PART and has no tile above it, a second one that
    follows the first row's strip solve) are all there is."""
    from psoap_amd.chunk import ChunkHandle
    for c, ne, npx in ((1, 1, 100), (1, 1, 128), (2, 1, 129), (2, 2, 100), (3, 3, 85), (2, 3, 128), (1, 5, 77)):
        ch = syn.make_chunk(c, ne, npx, seed=900 + npx)
        for B in (1, 3, 5):
            gps = syn.make_walkers(c, B, seed=11)
            lw = syn.walker_lwls(ch, syn.make_walker_velocities(ch, B, seed=12))
            with ChunkHandle(ch.fl, ch.sigma, max_batch=B) as h:
                h.set_mode("staged")
                staged = h.lnlike_batch(lw, gps)
                h.set_mode("dag")
                got = h.lnlike_batch(lw, gps)
                assert np.array_equal(h.lnlike_batch(lw, gps), got)
            for w in range(B):
                assert close(got[w], staged[w]), (ch.N, B, w)
            assert close(got[0], oracle.lnlike(lw[0], ch.fl, ch.sigma, gps[0])), (ch.N, B)
