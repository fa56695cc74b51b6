"""GPU: the process that owns the device for the reference's forked workers (psoap_amd/server.py) -- the drop-in call of K
worker processes evaluated as group launches by ONE process, values equal to the in-process path's to the parity contract, and
the auto-start / idle-exit life cycle."""
import json
import os
import subprocess
import sys
import time

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

_WORKER = r'''
import json, os, sys, time
import numpy as np
sys.path.insert(0, %(root)r)
from psoap_amd import covariance, synthetic as syn
k, n_iter = int(sys.argv[1]), int(sys.argv[2])
ch = syn.make_chunk(2, 6, 100 + 10 * k, seed=9100 + k)          # chunks of different sizes, as real ones are
vals = []
for it in range(n_iter):
    gp = np.asarray(syn.GP_BASE[2]) * (1.0 + 0.01 * (it %% 3))
    vals.append(float(covariance.lnlike_f_g(None, *ch.lwls, ch.fl, ch.sigma, *gp)))
h = next(iter(covariance._handles.values()))
st = h.server_stats()
hip_loaded = any("libamdhip64" in ln for ln in open("/proc/self/maps"))
print("RESULT " + json.dumps({"k": k, "vals": vals, "stats": st, "hip_loaded": hip_loaded}), flush=True)
'''


def _direct_values(k, n_iter):
    from psoap_amd import synthetic as syn
    from psoap_amd.chunk import ChunkHandle
    ch = syn.make_chunk(2, 6, 100 + 10 * k, seed=9100 + k)
    with ChunkHandle(ch.fl, ch.sigma, device=0) as h:
        return [h.lnlike(ch.lwls, np.asarray(syn.GP_BASE[2]) * (1.0 + 0.01 * (it % 3))) for it in range(min(n_iter, 3))]


def test_workers_through_an_auto_started_server(tmp_path):
    """PSOAP_GPU_SERVER=auto: the first of 6 forked-style workers starts the server, all six evaluate through it -- group
    launches of up to 6 chunks of different sizes -- without ever loading the HIP runtime themselves; the values are the
    in-process path's to the parity contract; the server leaves once it has had no client for its idle time."""
    prog = tmp_path / "worker.py"
    prog.write_text(_WORKER % {"root": ROOT})
    sock = str(tmp_path / "locks" / "gpu_server_0.sock")
    env = dict(os.environ, PSOAP_GPU_SERVER="auto", PSOAP_LOCK_DIR=str(tmp_path / "locks"), PSOAP_GPU_SERVER_IDLE_S="3")
    K, n_iter = 6, 30
    ps = [subprocess.Popen([sys.executable, str(prog), str(k), str(n_iter)], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           text=True, env=env) for k in range(K)]
    outs = [p.communicate(timeout=900) for p in ps]
    for p, (so, se) in zip(ps, outs):
        assert p.returncode == 0, so[-1500:] + se[-3000:]
    recs = sorted((json.loads([ln for ln in so.splitlines() if ln.startswith("RESULT ")][-1][len("RESULT "):]) for so, _ in outs),
                  key=lambda r: r["k"])
    assert os.path.exists(sock)
    for r in recs:
        assert not r["hip_loaded"], "a worker loaded the HIP runtime"
        want = _direct_values(r["k"], n_iter)
        for it, v in enumerate(r["vals"]):
            w = want[it % 3]
            assert abs(v - w) <= 1e-10 * max(1.0, abs(w)), (r["k"], it, v, w)
    last = max(recs, key=lambda r: r["stats"]["requests"])["stats"]
    assert last["clients_seen"] == K and last["requests"] >= K * n_iter - K
    assert last["grouped_launches"] > 0 and last["largest_group"] >= 3 and last["launches"] < last["requests"]
    t0 = time.time()
    while os.path.exists(sock) and time.time() - t0 < 30:
        time.sleep(0.5)
    assert not os.path.exists(sock), "the server did not leave after its idle time"
    assert "psoap GPU server" in open(sock + ".log").read()


def test_reference_goldens_through_the_server(tmp_path):
    """The drop-in functions against the reference's goldens (all five BASELINE shapes) and its recorded behaviour on degenerate
    input, with the evaluation in the server's process."""
    env = dict(os.environ, PSOAP_GPU_SERVER="auto", PSOAP_LOCK_DIR=str(tmp_path / "locks"), PSOAP_GPU_SERVER_IDLE_S="3")
    res = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-q", "-m", "gpu", "-x",
                          "-k", "lnlike_golden or conventions", "-p", "no:cacheprovider"],
                         env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-2000:]
    assert " passed" in res.stdout
    assert "psoap GPU server" in open(str(tmp_path / "locks" / "gpu_server_0.sock.log")).read()
