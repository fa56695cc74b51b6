"""GPU: several processes on one device (the reference's worker-per-chunk model, /root/reference/psoap/sample_parallel.py:
258-278) -- no silently wrong value.  The library's three lines of defence, each driven on purpose:

* a persistent launch that reports a workgroup the scheduler MOVED between compute units (DagCtl::pad[3], dag_where in
  dag_kernel.hpp) is issued again, then evaluated by the staged path -- forced here with PSOAP_TEST_TAINT_EVERY, which
  makes the host treat every k-th launch as disturbed (the device-side detection itself is exercised by the many-worker
  probes below and by tools/r5_share_experiments.sh);
* the staged path, selected up front (PSOAP_SHARE_POLICY=staged is what several processes without the lock get), meets the
  same goldens;
* the inter-process lock times out with an error that names the holder instead of hanging."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

_CODE = r'''
import json, os, sys
import numpy as np
sys.path.insert(0, %(root)r)
from psoap_amd import _lib, covariance, synthetic as syn
from psoap_amd.chunk import ChunkHandle, ChunkGroup, StreamPipeline
out = {}
ch = syn.make_chunk(2, 8, 150, seed=8100)                       # N = 1200
B = 6
gps = syn.make_walkers(2, B, seed=8101)
lw = syn.walker_lwls(ch, syn.make_walker_velocities(ch, B, seed=8102))
with ChunkHandle(ch.fl, ch.sigma, max_batch=B, device=0) as h:
    vals = [h.lnlike_batch(lw, gps) for _ in range(7)]
    out["batch"] = [[float(x).hex() for x in v] for v in vals]
    out["single"] = [float(h.lnlike(lw[0], gps[0])).hex() for _ in range(5)]
    # pipelined: upload(next) between eval and fetch -- a retry must evaluate the ACTIVE batch, not the pending one
    h.upload(lw, gps)
    seq = []
    for k in range(5):
        h.eval()
        h.upload(lw[::-1].copy() if k %% 2 == 0 else lw, gps[::-1].copy() if k %% 2 == 0 else gps)
        seq.append([float(x).hex() for x in h.fetch()])
    h.sync()
    out["pipelined"] = seq
    # stream: a disturbed matrix is submitted again under the caller's ticket
    h.stream_open(2, B)
    st = []
    for k in range(4):
        t = h.stream_submit(lw, gps)
        st.append([float(x).hex() for x in h.stream_fetch(t)])
    out["stream"] = st
    h.stream_close()
    mu, Sigma = h.predict(0, ch.lwls, np.stack([np.linspace(ch.lwls[0].min(), ch.lwls[0].max(), 130)] * 2), [1.0, 0.0],
                          syn.GP_BASE[2])
    mu2, Sigma2 = h.predict(0, ch.lwls, np.stack([np.linspace(ch.lwls[0].min(), ch.lwls[0].max(), 130)] * 2), [1.0, 0.0],
                            syn.GP_BASE[2])
    out["predict_mu"] = [float(x) for x in mu]
    out["predict_mu2"] = [float(x) for x in mu2]
    out["predict_sig"] = [float(np.abs(Sigma - Sigma2).max()), float(np.abs(Sigma).max())]
ch2 = syn.make_chunk(2, 5, 100, seed=8103)
lw2 = syn.walker_lwls(ch2, syn.make_walker_velocities(ch2, 3, seed=8104))
with ChunkHandle(ch.fl, ch.sigma, max_batch=B, device=0) as a, ChunkHandle(ch2.fl, ch2.sigma, max_batch=3, device=0) as b:
    g = ChunkGroup([a, b])
    grp = []
    for k in range(4):
        a.upload(lw, gps); b.upload(lw2, gps[:3])
        g.eval()
        grp.append([float(x).hex() for x in a.fetch()] + [float(x).hex() for x in b.fetch()])
    out["group"] = grp
    g.close()
out["stats"] = _lib.share_stats(0)
print("RESULT " + json.dumps(out), flush=True)
'''


def _run(env_extra, tmp_path):
    prog = tmp_path / "share_prog.py"
    prog.write_text(_CODE % {"root": ROOT})
    env = dict(os.environ, PSOAP_LOCK_DIR=str(tmp_path / "locks"), **env_extra)
    res = subprocess.run([sys.executable, str(prog)], capture_output=True, text=True, timeout=900, env=env)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    return json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("RESULT ")][-1][len("RESULT "):])


def _f(x):
    return np.array([float.fromhex(v) for v in np.ravel(x)])


def test_disturbed_launches_are_evaluated_again_and_nothing_changes(tmp_path):
    clean = _run({}, tmp_path)
    assert clean["stats"]["tainted"] == 0 and clean["stats"]["retries"] == 0 and clean["stats"]["staged_fallbacks"] == 0
    # (procs: this child, plus the test process itself where it holds a GPU context -- the driver's list counts every process
    # with a queue on the device, library or not)
    assert clean["stats"]["dag_launches"] > 20 and 1 <= clean["stats"]["procs"] <= 3 and clean["stats"]["lock_enabled"] == 1
    # every 3rd launch "disturbed": retried with the same task list -> the very same bits everywhere
    retried = _run({"PSOAP_TEST_TAINT_EVERY": "3"}, tmp_path)
    for key in ("batch", "single", "pipelined", "stream", "group"):
        assert retried[key] == clean[key], key
    assert retried["predict_mu"] == clean["predict_mu"] and retried["predict_mu2"] == clean["predict_mu2"]
    s = retried["stats"]
    assert s["tainted"] >= 10 and s["retries"] >= 8 and s["stream_resubmits"] >= 4 and s["staged_fallbacks"] == 0
    # no retries allowed: a disturbed launch goes down the staged path -- other order of summation, same values to the
    # parity contract (|d lnp| <= 1e-10 max(1, |lnp|)); a stream has no staged path and keeps resubmitting
    staged = _run({"PSOAP_TEST_TAINT_EVERY": "3", "PSOAP_SHARE_RETRIES": "0"}, tmp_path)
    s = staged["stats"]
    assert s["staged_fallbacks"] >= 8 and s["retries"] == 0
    for key in ("batch", "single", "pipelined", "group"):
        a, b = _f(staged[key]), _f(clean[key])
        assert np.all(np.abs(a - b) <= 1e-10 * np.maximum(1.0, np.abs(b))), key
    assert staged["stream"] == clean["stream"]
    assert np.allclose(staged["predict_mu"], clean["predict_mu"], rtol=0, atol=1e-10)
    # every value of a series is the series' first (repeatability inside each run)
    for run in (clean, retried):
        assert all(v == run["batch"][0] for v in run["batch"]) and all(v == run["single"][0] for v in run["single"])


def test_staged_policy_meets_the_goldens(tmp_path):
    """PSOAP_SHARE_POLICY=staged (what several processes WITHOUT the device lock get automatically): the reference goldens
    through the drop-in calls, lnlike and predict."""
    env = dict(os.environ, PSOAP_SHARE_POLICY="staged", PSOAP_LOCK_DIR=str(tmp_path / "locks"))
    res = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-q", "-m", "gpu", "-x",
                          "-k", "lnlike_golden or edge_sizes or predict_golden or predict_edge or walker_batch or conventions",
                          "-p", "no:cacheprovider"], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-2000:]
    assert " passed" in res.stdout


_HOLD = r'''
import fcntl, glob, os, sys, time
d = sys.argv[1]
for _ in range(600):
    fs = glob.glob(os.path.join(d, "gpu_*.lock"))
    if fs:
        break
    time.sleep(0.05)
fh = open(fs[0], "r+")
fcntl.flock(fh, fcntl.LOCK_EX)
fh.seek(0); fh.truncate(); fh.write(str(os.getpid()) + "\n"); fh.flush()
print("HELD", flush=True)
time.sleep(float(sys.argv[2]))
'''


def test_lock_wait_times_out_with_an_error_that_names_the_holder(tmp_path):
    locks = tmp_path / "locks"
    code = r'''
import os, sys, subprocess, time
sys.path.insert(0, %r)
from psoap_amd import _lib, synthetic as syn
from psoap_amd.chunk import ChunkHandle
ch = syn.make_chunk(1, 4, 100, seed=8200)
with ChunkHandle(ch.fl, ch.sigma, device=0) as h:
    first = h.lnlike(ch.lwls, syn.GP_BASE[1])                      # creates the lock file
    holder = subprocess.Popen([sys.executable, %r, %r, "6"], stdout=subprocess.PIPE, text=True)
    assert holder.stdout.readline().strip() == "HELD"
    t0 = time.time()
    try:
        h.lnlike(ch.lwls, syn.GP_BASE[1])
        print("RESULT no error")
    except _lib.PsoapError as e:
        print("RESULT %%.1f %%d %%s" %% (time.time() - t0, holder.pid, str(e)))
    holder.wait()
    again = h.lnlike(ch.lwls, syn.GP_BASE[1])                      # the device is free again: same value
    print("AGAIN", first == again)
'''
    hold = tmp_path / "hold.py"
    hold.write_text(_HOLD)
    prog = tmp_path / "prog.py"
    prog.write_text(code % (ROOT, str(hold), str(locks)))
    env = dict(os.environ, PSOAP_LOCK_DIR=str(locks), PSOAP_DEVICE_LOCK_TIMEOUT_S="1.5")
    res = subprocess.run([sys.executable, str(prog)], capture_output=True, text=True, timeout=300, env=env)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    line = [ln for ln in res.stdout.splitlines() if ln.startswith("RESULT ")][-1].split(" ", 3)
    waited, pid, msg = float(line[1]), line[2], line[3]
    assert 1.4 <= waited < 5.0, line
    assert "was not released within" in msg and ("pid " + pid) in msg, msg
    assert "AGAIN True" in res.stdout
    st = os.stat(str(locks))
    assert (st.st_mode & 0o777) == 0o700
    assert all((os.stat(os.path.join(str(locks), f)).st_mode & 0o777) == 0o600 for f in os.listdir(str(locks)))


@pytest.mark.parametrize("workers,lock", [(12, 2), (6, 0)])
def test_many_worker_processes_never_get_a_wrong_value(workers, lock):
    """12 forked workers with the device lock on (more than the eight process contexts the device keeps mapped: its scheduler
    then suspends and moves running workgroups -- 3-8 silently wrong lnprobs in 24,000 before round 5; the library sends them
    down the staged path, without the lock), and 6 WITHOUT the lock (persistent launches of several processes would starve
    each other: the staged path as well):
    two proposals in turn, every value equal to the worker's first to the parity contract, no time-out, and the library's
    account says what it did."""
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "shared_gpu_probe.py"), str(workers), "120", "3", str(lock)],
                         capture_output=True, text=True, timeout=1200)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-2000:]
    assert f"WRONG per worker {[0] * workers}" in res.stdout, res.stdout
    acct = json.loads(res.stdout.split("library account (all workers): ")[1].split("; procs seen")[0])
    # both regimes end on the staged path, without the lock (12 processes are more than PSOAP_SHARE_DAG_MAX; without the lock
    # two are): only what ran before the other workers had their slots was a persistent launch under the lock
    assert acct["staged_policy"] >= workers * 100
    if lock == 0:
        assert acct["lock_acquisitions"] == 0


_TWO_STREAMS = r'''
import json, os, sys, time
import numpy as np
sys.path.insert(0, %(root)r)
from psoap_amd import _lib, synthetic as syn
from psoap_amd.chunk import ChunkHandle
k = int(sys.argv[1])
ch = syn.make_chunk(2, 10, 200, seed=8300 + k)                  # N = 2000
B = 8
gps = syn.make_walkers(2, B, seed=8301)
lw = syn.walker_lwls(ch, syn.make_walker_velocities(ch, B, seed=8302 + k))
with ChunkHandle(ch.fl, ch.sigma, max_batch=B, device=0) as h:
    ref = h.lnlike_batch(lw, gps)
    h.stream_open(2, B)
    first = None
    bad = 0
    for step in range(40):
        out = h.stream_fetch(h.stream_submit(lw, gps))          # the stream stays OPEN between the steps
        first = out if first is None else first
        bad += int(not np.array_equal(out, first))
        time.sleep(0.002)                                        # (the other process gets the device in between)
    st = h.stream_stats()
    h.stream_close()
ok = bool(np.all(np.abs(first - ref) <= 1e-10 * np.maximum(1.0, np.abs(ref))))
print("RESULT " + json.dumps({"bad": bad, "ok": ok, "launches": st["launches"], "stats": _lib.share_stats(0)}), flush=True)
'''


def test_two_processes_with_open_streams_take_the_device_in_turn(tmp_path):
    """a stream whose last ticket has been fetched gave the device lock back while its resident launch
    still held every compute unit (idle time-out 20 ms) -- another process could start ITS persistent launch beside it.  Now the
    launch leaves before the lock does whenever other processes use the device.  Two processes, each with a stream left open
    between 40 steps: every step's values are the first step's, nothing times out, and the streams relaunch per step."""
    prog = tmp_path / "two_streams.py"
    prog.write_text(_TWO_STREAMS % {"root": ROOT})
    env = dict(os.environ, PSOAP_LOCK_DIR=str(tmp_path / "locks"))
    ps = [subprocess.Popen([sys.executable, str(prog), str(k)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
          for k in range(2)]
    outs = [p.communicate(timeout=600) for p in ps]
    for p, (so, se) in zip(ps, outs):
        assert p.returncode == 0, so[-1500:] + se[-3000:]
    recs = [json.loads([ln for ln in so.splitlines() if ln.startswith("RESULT ")][-1][len("RESULT "):]) for so, _ in outs]
    for r in recs:
        assert r["bad"] == 0 and r["ok"], recs
        assert 2 <= r["stats"]["procs"] <= 4 and r["stats"]["lock_acquisitions"] >= 40
        assert r["launches"] >= 30             # shared device: the resident launch left after (nearly) every step


def test_stream_submissions_are_checked_against_the_lane_buffers():
    """orbital parameters of the wrong width and more epochs than the dispatcher's LDS staging holds are
    refused instead of read / written past a lane's buffer."""
    from psoap_amd import _lib, synthetic as syn
    from psoap_amd.lnprob import ChunkWorker
    from psoap_amd.chunk import ChunkHandle
    ch = syn.make_chunk(2, 6, 50, seed=8400)
    w = ChunkWorker("SB2", ch.lwl, ch.fl, ch.sigma, ch.epoch_index, ch.dates, max_batch=2, device=0)
    w.stream_open(2)
    with pytest.raises(ValueError, match="p_orb must have shape"):
        w.handle.stream_submit_orbits(1, np.zeros((1, 5)), np.tile(syn.GP_BASE[2], (1, 1)))
    w.stream_close()
    w.close()
    big = syn.make_chunk(1, 3100, 2, seed=8401)                  # 3100 epochs: 3 x 3100 + 16 doubles > the 72 KB of LDS
    with ChunkHandle(big.fl, big.sigma, max_batch=2, device=0) as h:
        h.set_grid(big.lwl, big.epoch_index, 3100)
        with pytest.raises(_lib.PsoapError, match="too many epochs"):
            h.stream_open(1, 2)
