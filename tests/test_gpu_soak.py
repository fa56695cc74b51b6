"""Short soak of the persistent kernel's dependency protocol (the long version is tools/soak.py): repeated launches
over both split schemes, single evaluations, a heterogeneous group launch and the predict path -- every result
bit-identical to the first of its kind.  A protocol race or a miscompiled hand-off shows up here as a changed
last bit long before it shows up as a wrong digit."""
import numpy as np
import pytest

from psoap_amd import synthetic as syn
from psoap_amd.chunk import ChunkGroup, ChunkHandle

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cfg,B,reps", [(1, 32, 60), (1, 5, 60), (2, 32, 20), (3, 32, 8), (3, 4, 15), (3, 1, 20), (5, 2, 8)])
def test_repeated_launches_are_bit_identical(cfg, B, reps):
    ch = syn.make_config_chunk(cfg)
    c = ch.n_components
    gps = syn.make_walkers(c, B, seed=cfg)
    lw = syn.walker_lwls(ch, syn.make_walker_velocities(ch, B, seed=cfg + 10))
    with ChunkHandle(ch.fl, ch.sigma, max_batch=B) as h:
        h.upload(lw, gps)
        h.eval()
        ref = h.fetch()
        assert np.all(np.isfinite(ref))
        for _ in range(reps):
            h.eval()
            assert np.array_equal(h.fetch(), ref)


def test_repeated_group_launches_are_bit_identical():
    chunks = [syn.make_chunk(2, 3 + k, 90 + 17 * k, seed=50 + k) for k in range(5)]
    hs = [ChunkHandle(c_.fl, c_.sigma, max_batch=6) for c_ in chunks]
    gp6 = syn.make_walkers(2, 6, seed=9)
    try:
        with ChunkGroup(hs) as g:
            ref = None
            for _ in range(40):
                for h, c_ in zip(hs, chunks):
                    h.upload(np.repeat(c_.lwls[None], 6, axis=0), gp6)
                g.eval()
                out = np.stack([h.fetch() for h in hs])
                if ref is None:
                    ref = out
                assert np.array_equal(out, ref)
    finally:
        for h in hs:
            h.close()


@pytest.mark.parametrize("M,reps", [(160, 10), (130, 60), (43, 60)])
def test_repeated_predict_is_bit_identical(M, reps):
    """Also the host side of the Sigma download: a few threads pre-fault and fill the (fresh) output array of every call;
    small shapes finish on the device before those threads are even running (a thread that touched rows it does not
    fill itself zeroed words of finished rows in round 3's first version)."""
    ch = syn.make_chunk(3, 5, 120, seed=91)          # N = 600
    pred = np.linspace(ch.lwls[0].min(), ch.lwls[0].max(), M)
    with ChunkHandle(ch.fl, ch.sigma, max_batch=1) as h:
        mu0, S0 = h.predict(0, ch.lwls, np.stack([pred] * 3), np.zeros(3), syn.GP_BASE[3])
        assert np.array_equal(S0, S0.T)
        for _ in range(reps):
            mu, S = h.predict(0, ch.lwls, np.stack([pred] * 3), np.zeros(3), syn.GP_BASE[3])
            assert np.array_equal(mu, mu0) and np.array_equal(S, S0)
