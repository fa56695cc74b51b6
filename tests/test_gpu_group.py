"""GPU: several chunks of different sizes factored by ONE launch of the persistent kernel (ChunkGroup)
against the same chunks evaluated one at a time and against the oracle."""
import numpy as np
import pytest

from psoap_amd import synthetic as syn

pytestmark = pytest.mark.gpu
LNP_RTOL = 1e-10


def close(a, b):
    return abs(a - b) <= LNP_RTOL * max(1.0, abs(b))


@pytest.mark.parametrize("c,shapes,Bs", [
    (2, [(4, 150), (7, 118), (10, 130), (1, 77)], [5, 3, 5, 1]),     # N = 600, 826 (ragged), 1300, 77
    (1, [(3, 43), (3, 43), (9, 200)], [2, 2, 9]),                    # two equal chunks and a larger one
    (3, [(5, 100)], [4]),                                            # a group of one
])
def test_group_launch_matches_single_launches_and_oracle(oracle, c, shapes, Bs):
    from psoap_amd.chunk import ChunkGroup, ChunkHandle
    chunks = [syn.make_chunk(c, ne, npx, seed=9100 + i, masked_fraction=0.1 if npx > 100 else 0.0)
              for i, (ne, npx) in enumerate(shapes)]
    props = []
    for i, (ch, B) in enumerate(zip(chunks, Bs)):
        gps = syn.make_walkers(c, B, seed=9200 + i)
        lw = syn.walker_lwls(ch, syn.make_walker_velocities(ch, B, seed=9300 + i))
        props.append((lw, gps))
    props[0][1][-1, 1] = -1.0                    # a rejected proposal (l < 0) inside the first chunk's batch
    handles = [ChunkHandle(ch.fl, ch.sigma, max_batch=max(Bs)) for ch in chunks]
    try:
        single = [h.lnlike_batch(*p) for h, p in zip(handles, props)]
        with ChunkGroup(handles) as g:
            for rep in range(2):                 # the second pass reuses the cached task list
                for h, p in zip(handles, props):
                    h.upload(*p)
                g.eval()
                grouped = [h.fetch() for h in handles]
                for k in range(len(chunks)):
                    assert grouped[k].shape == (Bs[k],)
                    for w in range(Bs[k]):
                        if np.isneginf(single[k][w]):
                            assert np.isneginf(grouped[k][w])
                        else:
                            assert close(grouped[k][w], single[k][w]), (k, w, grouped[k][w], single[k][w])
            # batch sizes change: the task list is rebuilt
            handles[0].upload(props[0][0][:1], props[0][1][:1])
            for h, p in zip(handles[1:], props[1:]):
                h.upload(*p)
            g.eval()
            assert close(handles[0].fetch()[0], single[0][0])
            assert close(handles[-1].fetch()[0], single[-1][0])
        assert np.isneginf(single[0][-1])
        for k, ch in enumerate(chunks):
            want = oracle.lnlike(props[k][0][0], ch.fl, ch.sigma, props[k][1][0])
            assert close(grouped[k][0], want), (k, grouped[k][0], want)
        # the handles still work on their own after the group is gone
        last = len(handles) - 1
        assert close(handles[last].lnlike_batch(*props[last])[0], single[last][0])
    finally:
        for h in handles:
            h.close()


def test_group_rejects_mixed_component_counts():
    from psoap_amd._lib import PsoapError
    from psoap_amd.chunk import ChunkGroup, ChunkHandle
    a, b = syn.make_chunk(1, 2, 40, seed=1), syn.make_chunk(2, 2, 40, seed=2)
    with ChunkHandle(a.fl, a.sigma, 2) as ha, ChunkHandle(b.fl, b.sigma, 2) as hb, ChunkGroup([ha, hb]) as g:
        ha.upload(a.lwls[None], np.array([syn.GP_BASE[1]]))
        hb.upload(b.lwls[None], np.array([syn.GP_BASE[2]]))
        with pytest.raises(PsoapError, match="same number of components"):
            g.eval()


def test_ensemble_evaluator_uses_one_launch_for_its_chunks(oracle):
    from psoap_amd.ensemble import EnsembleEvaluator
    chunks = [syn.make_chunk(2, 3, 90 + 10 * k, seed=9400 + k) for k in range(3)]
    B = 4
    gps = syn.make_walkers(2, B, seed=9500)
    props = {k: (syn.walker_lwls(chunks[k], syn.make_walker_velocities(chunks[k], B, seed=9600 + k)), gps)
             for k in range(3)}
    ev = EnsembleEvaluator.from_chunks(chunks, max_batch=B)
    try:
        assert getattr(ev, "group", None) is not None
        got = ev.lnprob(props)
    finally:
        ev.close()
    want = sum(oracle.lnlike(props[k][0][1], chunks[k].fl, chunks[k].sigma, gps[1]) for k in range(3))
    assert close(got[1], want)


def test_group_plan_is_built_once_across_an_upload_eval_loop():
    """upload / eval / fetch steps alternate every member's proposal slot: the task list must be built once (it depends
    on the batch sizes only), the matrix records are refreshed on the device, and every step's values stay those of
    single launches -- with the uploads of step k+1 queued while step k is being evaluated."""
    from psoap_amd.chunk import ChunkGroup, ChunkHandle
    chunks = [syn.make_chunk(2, 3 + k, 100 + 9 * k, seed=9700 + k) for k in range(3)]
    B = 4
    steps = 6
    props = [[(syn.walker_lwls(ch, syn.make_walker_velocities(ch, B, seed=9800 + 10 * st + k)),
               syn.make_walkers(2, B, seed=9900 + st)) for k, ch in enumerate(chunks)] for st in range(steps)]
    handles = [ChunkHandle(ch.fl, ch.sigma, max_batch=B) for ch in chunks]
    try:
        want = [[h.lnlike_batch(*props[st][k]) for k, h in enumerate(handles)] for st in range(steps)]
        with ChunkGroup(handles) as g:
            for k, h in enumerate(handles):
                h.upload(*props[0][k])
            for st in range(steps):
                g.eval()
                if st + 1 < steps:
                    for k, h in enumerate(handles):          # next step's proposals go to the other slot meanwhile
                        h.upload(*props[st + 1][k])
                for k, h in enumerate(handles):
                    got = h.fetch()
                    assert np.allclose(got, want[st][k], rtol=LNP_RTOL, atol=0.0), (st, k)
            stats = g.stats()
            assert stats["plan_builds"] == 1, stats
            assert stats["record_refreshes"] == steps, stats      # one device-side refresh per slot flip
            # a different batch size: one more build
            for k, h in enumerate(handles):
                h.upload(props[0][k][0][:2], props[0][k][1][:2])
            g.eval()
            assert np.allclose(handles[1].fetch(), want[0][1][:2], rtol=LNP_RTOL, atol=0.0)
            assert g.stats()["plan_builds"] == 2
    finally:
        for h in handles:
            h.close()


def test_staged_mode_back_to_back_evals_with_changing_batch_size():
    """eval, upload, eval with no fetch in between and a different B: the stream-group boundaries of the staged mode
    move, so every group stream has to wait for the handle's previous evaluation before it touches the shared
    workspaces (round 2 waited on the slot's own event, i.e. on the evaluation two steps back)."""
    from psoap_amd.chunk import ChunkHandle
    ch = syn.make_chunk(2, 6, 200, seed=9950)               # N = 1200
    big, small = 6, 2
    gps = syn.make_walkers(2, big, seed=9951)
    lw = syn.walker_lwls(ch, syn.make_walker_velocities(ch, big, seed=9952))
    with ChunkHandle(ch.fl, ch.sigma, max_batch=big) as h:
        want_big = h.lnlike_batch(lw, gps)
        want_small = h.lnlike_batch(lw[3:5], gps[3:5])
        h.set_mode("staged")
        h.set_stream_groups(3)
        for _ in range(5):
            h.upload(lw, gps)
            h.eval()
            h.upload(lw[3:5], gps[3:5])
            h.eval()                                        # B = 2 right behind B = 6, nothing fetched in between
            got_small = h.fetch()
            h.upload(lw, gps)
            h.eval()
            got_big = h.fetch()
            assert np.allclose(got_small, want_small, rtol=LNP_RTOL, atol=0.0)
            assert np.allclose(got_big, want_big, rtol=LNP_RTOL, atol=0.0)


def test_group_launch_at_the_reference_chunk_sizes(oracle):
    """The reference's own regime: chunks of ~80 pixels x 12 epochs (scripts/psoap_generate_chunks.py:8-9, 73-96 -> N ~ 1000),
    many chunks x the walkers of one ensemble step in ONE launch (256 matrices: the throughput scheme, 32 per queue): every
    chunk's first and last walker against the oracle, the whole table bit-identical on a second launch."""
    from psoap_amd.ensemble import EnsembleEvaluator
    n_chunks, B = 8, 32
    chunks = [syn.make_chunk(2, 12, 84, seed=7400 + k) for k in range(n_chunks)]        # N = 1008
    gps = syn.make_walkers(2, B, seed=7401)
    props = {k: (syn.walker_lwls(chunks[k], syn.make_walker_velocities(chunks[k], B, seed=7410 + k)), gps) for k in range(n_chunks)}
    ev = EnsembleEvaluator.from_chunks(chunks, max_batch=B)
    try:
        total = ev.lnprob(props)
        assert np.array_equal(total, ev.lnprob(props))
        want = np.zeros(B)
        for w in (0, B - 1):
            want[w] = sum(oracle.lnlike(props[k][0][w], chunks[k].fl, chunks[k].sigma, list(gps[w])) for k in range(n_chunks))
            assert abs(total[w] - want[w]) <= LNP_RTOL * max(1.0, abs(want[w])), (w, total[w], want[w])
    finally:
        ev.close()
