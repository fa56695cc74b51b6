"""GPU: the two-slot upload pipeline of a chunk handle (include/psoap_gp.h: psoap_batch_upload / eval / fetch).
upload(k+1) while eval(k) runs must never disturb the results of k, and must deliver k+1 afterwards."""
import numpy as np
import pytest

from psoap_amd import synthetic as syn

pytestmark = pytest.mark.gpu


def _props(ch, B, seed):
    gps = syn.make_walkers(ch.n_components, B, seed=seed)
    lw = syn.walker_lwls(ch, syn.make_walker_velocities(ch, B, seed=seed + 1))
    return lw, gps


@pytest.mark.parametrize("mode", ["dag", "staged"])
def test_upload_overlaps_eval_without_mixing_batches(mode):
    from psoap_amd.chunk import ChunkHandle
    ch = syn.make_chunk(2, 10, 140, seed=6100)               # N = 1400
    B = 6
    sets = [_props(ch, B, 6200 + 10 * i) for i in range(4)]
    sets[2][1][3, 0] = -0.1                                   # a rejected proposal in batch 2 only
    with ChunkHandle(ch.fl, ch.sigma, max_batch=B) as h:
        h.set_mode(mode)
        serial = [h.lnlike_batch(*s) for s in sets]          # upload, eval, fetch one after the other
        assert np.isneginf(serial[2][3]) and np.all(np.isfinite(serial[1]))
        # pipelined: the next upload is issued while the evaluation is in flight
        h.upload(*sets[0])
        got = []
        for k in range(len(sets)):
            h.eval()
            if k + 1 < len(sets):
                h.upload(*sets[k + 1])
            got.append(h.fetch())
        for k in range(len(sets)):
            assert np.array_equal(got[k], serial[k]), (mode, k, got[k], serial[k])
        # eval without a new upload re-evaluates the current batch; a double upload keeps the last one
        h.eval()
        assert np.array_equal(h.fetch(), serial[-1])
        h.upload(*sets[0])
        h.upload(*sets[1])
        h.eval()
        assert np.array_equal(h.fetch(), serial[1])
        # smaller batch after a larger one, and the velocity upload path through the same slots (two matrices are
        # scheduled differently from six -- the following scheme -- so the sums round differently: a few ulp)
        h.upload(sets[3][0][:2], sets[3][1][:2])
        h.eval()
        two = h.fetch()
        assert np.all(np.abs(two - serial[3][:2]) <= 1e-13 * np.abs(serial[3][:2]))
        h.eval()
        assert np.array_equal(h.fetch(), two)


def test_pipeline_with_device_side_doppler_shift():
    from psoap_amd.chunk import ChunkHandle
    ch = syn.make_chunk(2, 8, 100, seed=6300)                # N = 800
    B = 4
    gps = syn.make_walkers(2, B, seed=6301)
    vels = [syn.make_walker_velocities(ch, B, seed=6302 + i) for i in range(3)]
    with ChunkHandle(ch.fl, ch.sigma, max_batch=B) as h:
        h.set_grid(ch.lwl, ch.epoch_index, ch.n_epochs)
        serial = [h.lnlike_batch(syn.walker_lwls(ch, v), gps) for v in vels]
        h.upload_velocities(vels[0], gps)
        for k in range(3):
            h.eval()
            if k + 1 < 3:
                h.upload_velocities(vels[k + 1], gps)
            got = h.fetch()
            assert np.allclose(got, serial[k], rtol=1e-12, atol=0.0), (k, got, serial[k])
