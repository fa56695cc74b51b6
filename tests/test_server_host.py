"""CPU: the GPU-owning server of psoap_amd/server.py with the oracle standing in for the device -- wire format, batching of
the requests that arrive together into one group launch, error replies, clients that come and go, the drop-in call through
``PSOAP_GPU_SERVER``.  (The device side: tests/test_gpu_server.py.)"""
import multiprocessing as mp
import os
import sys
import threading
import time

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

from psoap_amd import server, synthetic as syn  # noqa: E402


class _Handle:
    def __init__(self, fl, sigma, log):
        self.fl, self.sigma, self.N, self.log = fl, sigma, len(fl), log
        self.up = None
        self.closed = False

    def _value(self, lw, gp, mu):
        import oracle
        if any(g < 0 for g in gp):
            return -np.inf
        return float(oracle.lnlike(lw, self.fl, self.sigma, list(gp), mu_GP=mu))

    def lnlike(self, lw, gp, mu):
        self.log.append(("single", self.N))
        return self._value(lw, gp, mu)

    def upload(self, lw, gp, mu):
        self.up = (lw[0], gp[0], mu)

    def fetch(self):
        return np.array([self._value(*self.up)])

    def close(self):
        self.closed = True


class _Group:
    def __init__(self, handles, log):
        self.handles, self.log = handles, log

    def eval(self):
        self.log.append(("group", len(self.handles)))

    def close(self):
        pass


class _Backend:
    def __init__(self):
        self.log = []
        self.handles = []

    def open(self, fl, sigma):
        h = _Handle(fl, sigma, self.log)
        self.handles.append(h)
        return h

    def group(self, handles):
        return _Group(handles, self.log)


def _start(tmp_path, **kw):
    path = str(tmp_path / "srv.sock")
    be = _Backend()
    srv = server.GpuServer(path, be, idle_exit_s=0, **kw)
    th = threading.Thread(target=srv.serve, daemon=True)
    th.start()
    return path, be, srv, th


def test_round_trips_errors_and_bookkeeping(tmp_path):
    import oracle
    path, be, srv, th = _start(tmp_path)
    ch = syn.make_chunk(2, 3, 40, seed=31)
    gp = syn.GP_BASE[2]
    rc = server.RemoteChunk(ch.fl, ch.sigma, path)
    want = oracle.lnlike(ch.lwls, ch.fl, ch.sigma, list(gp))
    assert rc.lnlike(ch.lwls, gp) == want
    assert rc.lnlike(ch.lwls, gp, 1.1) == oracle.lnlike(ch.lwls, ch.fl, ch.sigma, list(gp), mu_GP=1.1)
    assert rc.lnlike(ch.lwls, (-0.1, 5.0, 0.1, 7.0)) == -np.inf
    with pytest.raises(ValueError):
        rc.lnlike(ch.lwls[:, :-1], gp)
    # a second chunk of another size and component count on another connection
    ch1 = syn.make_chunk(1, 2, 30, seed=32)
    rc1 = server.RemoteChunk(ch1.fl, ch1.sigma, path)
    assert rc1.lnlike(ch1.lwls, syn.GP_BASE[1]) == oracle.lnlike(ch1.lwls, ch1.fl, ch1.sigma, list(syn.GP_BASE[1]))
    st = rc.server_stats()
    assert st["chunks"] == 2 and st["requests"] == 4 and st["clients_seen"] == 2
    # a request for a chunk that is not this connection's is answered with an error, the server lives on
    import struct
    server._send(rc1.sock, b"L" + struct.pack("<qqd", rc.cid, 1, 1.0) + np.zeros(2).tobytes() + np.zeros(ch.N).tobytes())
    assert server._recv(rc1.sock)[:1] == b"e"
    server._send(rc1.sock, b"X")
    assert server._recv(rc1.sock)[:1] == b"e"
    assert rc1.lnlike(ch1.lwls, syn.GP_BASE[1]) == oracle.lnlike(ch1.lwls, ch1.fl, ch1.sigma, list(syn.GP_BASE[1]))
    rc1.close()
    rc.sock.close()                       # a worker that dies without saying goodbye: its chunk is released
    rc.sock = None
    for _ in range(100):
        if all(h.closed for h in be.handles):
            break
        time.sleep(0.02)
    assert all(h.closed for h in be.handles)
    c2 = server.RemoteChunk(ch.fl, ch.sigma, path)
    server._send(c2.sock, b"Q")
    assert server._recv(c2.sock) == b"q"
    th.join(timeout=5)
    assert not th.is_alive() and not os.path.exists(path)


def _worker(k, path, n_iter, barrier, q):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    ch = syn.make_chunk(2, 3, 30 + 2 * k, seed=40 + k)
    rc = server.RemoteChunk(ch.fl, ch.sigma, path)
    vals = []
    for it in range(n_iter):
        barrier.wait()                                    # the master's proposal reaches every worker at once
        vals.append(rc.lnlike(ch.lwls, np.asarray(syn.GP_BASE[2]) * (1.0 + 0.01 * it)))
    st = rc.server_stats() if k == 0 else None
    rc.close()
    q.put((k, vals, st))


def test_requests_of_one_iteration_share_a_launch(tmp_path, monkeypatch):
    """K worker processes, each with its own chunk, ask at the same time (the reference's master sends a proposal to all its
    workers, sample_parallel.py:378-381): from the second iteration on the server evaluates them as ONE group launch."""
    import oracle
    # (a generous window and memory: what is tested is the grouping rule, not this machine's scheduling of six processes)
    monkeypatch.setattr(server, "RECENT_S", 30.0)
    path, be, srv, th = _start(tmp_path, window_s=5.0)
    K, n_iter = 5, 4
    ctx = mp.get_context("fork")
    barrier, q = ctx.Barrier(K), ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(k, path, n_iter, barrier, q)) for k in range(K)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join()
    for k, vals, _ in res:
        ch = syn.make_chunk(2, 3, 30 + 2 * k, seed=40 + k)
        for it, v in enumerate(vals):
            assert v == oracle.lnlike(ch.lwls, ch.fl, ch.sigma, list(np.asarray(syn.GP_BASE[2]) * (1.0 + 0.01 * it)))
    st = res[0][2]
    assert K * (n_iter - 1) < st["requests"] <= K * n_iter and st["largest_group"] == K, (st, be.log)   # (asked by worker 0 right after ITS last reply)
    assert ("group", K) in be.log and be.log.count(("group", K)) >= n_iter - 2
    assert st["launches"] < K * n_iter


def test_drop_in_call_goes_through_the_server(tmp_path, monkeypatch):
    """PSOAP_GPU_SERVER=on: covariance.lnlike_f_g in a worker is a round trip to the process that owns the GPU -- the worker
    loads no HIP library (this test has none to load); the shim's conventions (negative hyper-parameter, l == 0, NaN) stay
    on the worker's side."""
    import oracle
    from psoap_amd import covariance
    path, be, srv, th = _start(tmp_path)
    monkeypatch.setenv("PSOAP_GPU_SERVER", "on")
    monkeypatch.setenv("PSOAP_GPU_SERVER_SOCKET", path)
    ch = syn.make_chunk(2, 3, 40, seed=51)
    gp = syn.GP_BASE[2]
    try:
        got = covariance.lnlike["SB2"](None, *ch.lwls, ch.fl, ch.sigma, *gp)
        assert got == oracle.lnlike(ch.lwls, ch.fl, ch.sigma, list(gp))
        assert covariance.lnlike_f_g(None, *ch.lwls, ch.fl, ch.sigma, -0.2, 5.0, 0.1, 7.0) == -np.inf
        with pytest.raises(ZeroDivisionError):
            covariance.lnlike_f_g(None, *ch.lwls, ch.fl, ch.sigma, 0.2, 0.0, 0.1, 7.0)
        with pytest.raises(ValueError):
            covariance.lnlike_f_g(None, ch.lwls[0] * np.nan, ch.lwls[1], ch.fl, ch.sigma, *gp)
        assert len(be.handles) == 1                       # one resident chunk for all those calls
    finally:
        for h in list(covariance._handles.values()):
            h.close()
        covariance._handles.clear()
    monkeypatch.setenv("PSOAP_GPU_SERVER_SOCKET", str(tmp_path / "nobody.sock"))
    from psoap_amd._lib import PsoapError
    with pytest.raises(PsoapError, match="no server answers"):
        server.connect_chunk(ch.fl, ch.sigma, device=0)


def test_a_worker_that_dies_with_a_request_pending_fails_nobody_else(tmp_path):
    """Three workers are expected (a long window); one sends its request and dies before the launch: its request is
    dropped with it, the other two get their values (round 5's review: the dead worker's entry stayed in `pending` and the
    launch then failed for everybody with the same component count).  A second server on a live socket is refused."""
    import struct
    import oracle
    path, be, srv, th = _start(tmp_path, window_s=0.5)
    chs = [syn.make_chunk(2, 3, 30 + 4 * k, seed=70 + k) for k in range(3)]
    gp = syn.GP_BASE[2]
    rcs = [server.RemoteChunk(c.fl, c.sigma, path) for c in chs]
    for rc, c in zip(rcs, chs):           # everybody has asked once: all three count as "expected" from now on
        assert rc.lnlike(c.lwls, gp) == oracle.lnlike(c.lwls, c.fl, c.sigma, list(gp))
    with pytest.raises(RuntimeError):
        server.GpuServer(path, _Backend(), idle_exit_s=0)
    dead, c0 = rcs[0], chs[0]
    server._send(dead.sock, b"L" + struct.pack("<qqd", dead.cid, 2, 1.0) + np.asarray(gp, dtype="<f8").tobytes()
                 + np.ascontiguousarray(c0.lwls, dtype="<f8").tobytes())
    time.sleep(0.05)
    dead.sock.close()
    dead.sock = None
    out = {}

    def ask(k):
        out[k] = rcs[k].lnlike(chs[k].lwls, gp)

    ts = [threading.Thread(target=ask, args=(k,)) for k in (1, 2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=20)
    for k in (1, 2):
        assert out[k] == oracle.lnlike(chs[k].lwls, chs[k].fl, chs[k].sigma, list(gp))
    server._send(rcs[1].sock, b"Q")
    assert server._recv(rcs[1].sock) == b"q"
    th.join(timeout=5)
