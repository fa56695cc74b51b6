"""One process owns the GPU; the reference's forked workers become its clients.

/root/reference/psoap/sample_parallel.py forks one worker process per spectral chunk (:258-278) -- dozens per run with the
default ~80-pixel chunks (scripts/psoap_generate_chunks.py:73-96) -- and every iteration the master sends the SAME proposal to all
of them and sums what comes back (:371-390).  Dozens of processes on one GPU is what a GPU does worst: beyond eight process
contexts its scheduler swaps address spaces under running kernels (the library then falls back to the staged path,
safe, 16 workers = 155 evaluations per second at N = 6000), while ONE process evaluates the chunks of one iteration as ONE
heterogeneous launch of the persistent kernel (``ChunkGroup``: 8 chunks x 32 walkers at N = 2000 = 14,000 evaluations per
second).  This module puts that process between the workers and the device:

* ``python -m psoap_amd.server [--device D]`` (or ``PSOAP_GPU_SERVER=auto``: the first worker that finds none starts one) owns
  the device.  It keeps one resident ``ChunkHandle`` per client chunk, collects the requests that arrive together -- the K
  workers of one iteration -- and evaluates them in one group launch; a lone request is evaluated at once.
* ``covariance.lnlike_f / _f_g / _f_g_h`` in a worker -- the drop-in call, unchanged -- become a round trip over a Unix socket
  in the per-user directory of the device lock (``RemoteChunk``): the worker never initialises HIP.

Wire format: frames of ``<uint64 length><payload>``; payload = one tag byte + packed fields (little-endian doubles / int64).
The values are those of the same launches issued in-process (tests/test_gpu_server.py); which requests share a launch decides
the last bits (a batch's plan), never more.  ``predict_*`` and the fills stay in-process calls (the retrieve and
calibration scripts are single processes).
"""
from __future__ import annotations

import json
import os
import selectors
import socket
import struct
import sys
import time

import numpy as np

_HDR = struct.Struct("<Q")
WINDOW_S = 0.002          # how long a request waits for the others of its iteration (only while others are expected)
FRAME_TIMEOUT_S = 5.0       # a frame that does not arrive whole within this time: the client is dropped
RECENT_S = 2.0            # a client counts as "expected" while its last request is at most this old (an idle one costs the others WINDOW_S)


# ------------------------------------------------------------------------------------------------ framing
def _send(sock, payload: bytes):
    sock.sendall(_HDR.pack(len(payload)) + payload)


def _recv_exact(sock, n: int) -> bytes:
    buf = bytearray(n)
    view = memoryview(buf)
    got = 0
    while got < n:
        k = sock.recv_into(view[got:], n - got)
        if k == 0:
            raise ConnectionError("peer closed the connection")
        got += k
    return bytes(buf)


def _recv(sock) -> bytes:
    (n,) = _HDR.unpack(_recv_exact(sock, _HDR.size))
    if n > (1 << 34):
        raise ConnectionError("oversized frame")
    return _recv_exact(sock, n)


def socket_path(device: int = 0) -> str:
    """``$PSOAP_GPU_SERVER_SOCKET``, else ``<dir>/gpu_server_<device>.sock`` with the per-user directory of the device lock
    (``$PSOAP_LOCK_DIR``, ``$XDG_RUNTIME_DIR/psoap``, ``/tmp/psoap-<uid>``; created 0700)."""
    p = os.environ.get("PSOAP_GPU_SERVER_SOCKET")
    if p:
        return p
    d = os.environ.get("PSOAP_LOCK_DIR") or (os.path.join(os.environ["XDG_RUNTIME_DIR"], "psoap")
                                             if os.environ.get("XDG_RUNTIME_DIR") else f"/tmp/psoap-{os.geteuid()}")
    os.makedirs(d, mode=0o700, exist_ok=True)
    st = os.lstat(d)
    if not os.path.isdir(d) or os.path.islink(d) or st.st_uid != os.geteuid():
        raise RuntimeError(f"{d} is not a directory of this user")
    return os.path.join(d, f"gpu_server_{int(device)}.sock")


# ------------------------------------------------------------------------------------------------ server
class _DeviceBackend:
    """What the server evaluates with: resident chunk handles and group launches of the HIP library."""

    def __init__(self, device: int):
        from .chunk import ChunkGroup, ChunkHandle
        self._H, self._G, self.device = ChunkHandle, ChunkGroup, device

    def open(self, fl, sigma):
        return self._H(fl, sigma, max_batch=1, device=self.device)

    def group(self, handles):
        return self._G(handles)


class GpuServer:
    def __init__(self, path: str, backend, idle_exit_s: float = 60.0, window_s: float | None = None, max_groups: int = 16):
        self.path, self.backend = path, backend
        self.idle_exit_s = idle_exit_s
        self.window_s = WINDOW_S if window_s is None else window_s
        self.max_groups = max_groups
        self.chunks = {}                 # chunk id -> handle
        self.owner = {}                  # chunk id -> socket
        self.groups = {}                 # tuple of chunk ids -> group (insertion order = age)
        self.last_request = {}           # socket -> time of its last lnlike request
        self.next_id = 1
        self.stats = {"requests": 0, "launches": 0, "grouped_launches": 0, "largest_group": 0, "clients_seen": 0}
        self.pending = {}                # socket -> its request waiting for the next launch
        self.sel = selectors.DefaultSelector()
        if os.path.exists(path):
            # a socket file left by a server that died is replaced; one a LIVE server listens on is not stolen
            probe = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
            probe.settimeout(1.0)
            try:
                probe.connect(path)
            except OSError:
                os.unlink(path)
            else:
                raise RuntimeError(f"a GPU server is already listening on {path}")
            finally:
                probe.close()
        self.listener = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
        old = os.umask(0o177)
        try:
            self.listener.bind(path)
        finally:
            os.umask(old)
        self.listener.listen(128)
        self.listener.setblocking(False)
        self.sel.register(self.listener, selectors.EVENT_READ, "accept")
        self.quit = False

    # -- bookkeeping ------------------------------------------------------------------------------
    def _drop_client(self, sock):
        for cid in [c for c, s in self.owner.items() if s is sock]:
            self._close_chunk(cid)
        self.last_request.pop(sock, None)
        self.pending.pop(sock, None)     # (its request must not reach the next launch: the chunk is gone)
        try:
            self.sel.unregister(sock)
        except Exception:
            pass
        sock.close()

    def _close_chunk(self, cid):
        for key in [k for k in self.groups if cid in k]:
            self.groups.pop(key).close()
        h = self.chunks.pop(cid, None)
        self.owner.pop(cid, None)
        if h is not None:
            h.close()

    def _group_for(self, ids):
        key = tuple(ids)
        g = self.groups.get(key)
        if g is None:
            while len(self.groups) >= self.max_groups:
                self.groups.pop(next(iter(self.groups))).close()
            g = self.groups[key] = self.backend.group([self.chunks[c] for c in ids])
        return g

    # -- one message ------------------------------------------------------------------------------
    def _read(self, sock, pending):
        try:
            msg = _recv(sock)
        except (ConnectionError, OSError):
            self._drop_client(sock)
            return
        tag = msg[:1]
        try:
            if tag == b"O":
                (n,) = struct.unpack_from("<q", msg, 1)
                fl = np.frombuffer(msg, dtype="<f8", count=n, offset=9).copy()
                sigma = np.frombuffer(msg, dtype="<f8", count=n, offset=9 + 8 * n).copy()
                cid = self.next_id
                self.next_id += 1
                self.chunks[cid] = self.backend.open(fl, sigma)
                self.owner[cid] = sock
                _send(sock, b"o" + struct.pack("<q", cid))
            elif tag == b"L":
                cid, c, mu = struct.unpack_from("<qqd", msg, 1)
                if cid not in self.chunks or self.owner[cid] is not sock:
                    raise KeyError("unknown chunk id")
                n = self.chunks[cid].N
                gp = np.frombuffer(msg, dtype="<f8", count=2 * c, offset=25).copy()
                lw = np.frombuffer(msg, dtype="<f8", count=c * n, offset=25 + 16 * c).reshape(c, n).copy()
                pending[sock] = (cid, int(c), float(mu), gp, lw)
                self.last_request[sock] = time.monotonic()
                self.stats["requests"] += 1
            elif tag == b"C":
                (cid,) = struct.unpack_from("<q", msg, 1)
                if self.owner.get(cid) is sock:
                    self._close_chunk(cid)
                _send(sock, b"c")
            elif tag == b"S":
                _send(sock, b"s" + json.dumps(dict(self.stats, chunks=len(self.chunks), groups=len(self.groups))).encode())
            elif tag == b"Q":
                self.quit = True
                _send(sock, b"q")
            else:
                raise ValueError(f"unknown request {tag!r}")
        except Exception as e:                      # a bad request is answered, never fatal for the server
            try:
                _send(sock, b"e" + f"{type(e).__name__}: {e}".encode())
            except OSError:
                self._drop_client(sock)

    # -- evaluation -------------------------------------------------------------------------------
    def _evaluate(self, pending):
        by_c = {}
        for sock, req in list(pending.items()):
            if req[0] not in self.chunks:           # its client went away between the request and the launch
                continue
            by_c.setdefault(req[1], []).append((sock, req))
        for c, reqs in by_c.items():
            reqs.sort(key=lambda r: r[1][0])
            try:
                if len(reqs) == 1:
                    sock, (cid, _, mu, gp, lw) = reqs[0]
                    vals = [self.chunks[cid].lnlike(lw, gp, mu)]
                else:
                    ids = [r[1][0] for r in reqs]
                    group = self._group_for(ids)
                    for _, (cid, _, mu, gp, lw) in reqs:
                        self.chunks[cid].upload(lw[None], gp[None], mu)
                    group.eval()
                    vals = [float(self.chunks[cid].fetch()[0]) for cid in ids]
                    self.stats["grouped_launches"] += 1
                self.stats["launches"] += 1
                self.stats["largest_group"] = max(self.stats["largest_group"], len(reqs))
                for (sock, _), v in zip(reqs, vals):
                    try:
                        _send(sock, b"l" + struct.pack("<d", float(v)))
                    except OSError:
                        self._drop_client(sock)
            except Exception as e:
                for sock, _ in reqs:
                    try:
                        _send(sock, b"e" + f"{type(e).__name__}: {e}".encode())
                    except OSError:
                        self._drop_client(sock)

    # -- main loop --------------------------------------------------------------------------------
    def serve(self):
        idle_since = time.monotonic()
        pending = self.pending
        first = 0.0
        try:
            while not self.quit:
                if pending:
                    now = time.monotonic()
                    expected = sum(1 for t in self.last_request.values() if now - t <= RECENT_S)
                    left = first + self.window_s - now
                    if len(pending) >= expected or left <= 0.0:
                        self._evaluate(pending)
                        pending.clear()
                        continue
                    timeout = left
                else:
                    timeout = 1.0
                for key, _ in self.sel.select(timeout):
                    if key.data == "accept":
                        conn, _ = self.listener.accept()
                        # (frames are read whole inside the one-threaded loop: a client that stops in the middle of one
                        # holds everybody up for at most this long, then it is dropped)
                        conn.settimeout(FRAME_TIMEOUT_S)
                        self.sel.register(conn, selectors.EVENT_READ, "client")
                        self.stats["clients_seen"] += 1
                    else:
                        had = bool(pending)
                        self._read(key.fileobj, pending)
                        if pending and not had:
                            first = time.monotonic()
                n_clients = len(self.sel.get_map()) - 1
                if n_clients > 0 or pending:
                    idle_since = time.monotonic()
                elif self.idle_exit_s > 0 and time.monotonic() - idle_since > self.idle_exit_s:
                    break
        finally:
            for cid in list(self.chunks):
                self._close_chunk(cid)
            for key in list(self.sel.get_map().values()):
                if key.data == "client":
                    key.fileobj.close()
            self.listener.close()
            try:
                os.unlink(self.path)
            except OSError:
                pass


# ------------------------------------------------------------------------------------------------ client
class RemoteChunk:
    """The ``ChunkHandle`` of the drop-in path on the other side of the socket: ``lnlike(lwls, gp, mu_GP)``, ``close()``."""

    def __init__(self, fl, sigma, path: str):
        from ._lib import PsoapError
        self._err = PsoapError
        self.fl = np.ascontiguousarray(fl, dtype=np.float64)
        self.sigma = np.ascontiguousarray(sigma, dtype=np.float64)
        self.N = int(self.fl.shape[0])
        self.sock = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
        self.sock.connect(path)
        _send(self.sock, b"O" + struct.pack("<q", self.N) + self.fl.astype("<f8").tobytes() + self.sigma.astype("<f8").tobytes())
        rep = _recv(self.sock)
        if rep[:1] != b"o":
            raise self._err("GPU server: " + rep[1:].decode(errors="replace"))
        (self.cid,) = struct.unpack_from("<q", rep, 1)

    def lnlike(self, lwls, gp, mu_GP: float = 1.0) -> float:
        lw = np.ascontiguousarray(np.atleast_2d(lwls), dtype="<f8")
        c = lw.shape[0]
        if lw.shape != (c, self.N):
            raise ValueError(f"expected shape ({c}, {self.N}), got {lw.shape}")
        gp = np.ascontiguousarray(gp, dtype="<f8")
        if gp.shape != (2 * c,):
            raise ValueError(f"expected shape ({2 * c},), got {gp.shape}")
        try:
            _send(self.sock, b"L" + struct.pack("<qqd", self.cid, c, float(mu_GP)) + gp.tobytes() + lw.tobytes())
            rep = _recv(self.sock)
        except (ConnectionError, OSError) as e:       # the server is gone: loud, like every other failure of the path
            raise self._err(f"GPU server: connection lost ({e})") from e
        if rep[:1] == b"l":
            return struct.unpack_from("<d", rep, 1)[0]
        raise self._err("GPU server: " + rep[1:].decode(errors="replace"))

    def server_stats(self) -> dict:
        _send(self.sock, b"S")
        return json.loads(_recv(self.sock)[1:].decode())

    def close(self):
        if getattr(self, "sock", None) is not None:
            try:
                _send(self.sock, b"C" + struct.pack("<q", self.cid))
                _recv(self.sock)
            except Exception:
                pass
            self.sock.close()
            self.sock = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def wanted() -> str | None:
    """``PSOAP_GPU_SERVER``: unset / ``0`` -> None (evaluate in this process); ``auto`` -> start a server for the default device
    when none answers; ``1`` / ``on`` -> a server must be running already."""
    v = os.environ.get("PSOAP_GPU_SERVER", "").strip().lower()
    return None if v in ("", "0", "off", "no") else ("auto" if v == "auto" else "on")


def connect_chunk(fl, sigma, device: int | None = None) -> RemoteChunk:
    """A ``RemoteChunk`` on the server of ``device``; with ``PSOAP_GPU_SERVER=auto`` the server is started when absent (one
    starter at a time: ``flock`` on ``<socket>.start``), as a detached child that outlives this worker and exits once it has
    been without clients for ``PSOAP_GPU_SERVER_IDLE_S`` seconds (default 60)."""
    from . import _lib
    dev = _lib.default_device() if device is None else int(device)
    path = socket_path(dev)
    try:
        return RemoteChunk(fl, sigma, path)
    except (FileNotFoundError, ConnectionRefusedError):
        if wanted() != "auto":
            raise _lib.PsoapError(f"PSOAP_GPU_SERVER is set but no server answers on {path} "
                                  "(start one with `python -m psoap_amd.server`, or use PSOAP_GPU_SERVER=auto)")
    import fcntl
    import subprocess
    with open(path + ".start", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            return RemoteChunk(fl, sigma, path)          # somebody else started it while we waited
        except (FileNotFoundError, ConnectionRefusedError):
            pass
        log = open(path + ".log", "ab")
        subprocess.Popen([sys.executable, "-m", "psoap_amd.server", "--device", str(dev), "--socket", path,
                          "--idle-exit", os.environ.get("PSOAP_GPU_SERVER_IDLE_S", "60")],
                         stdin=subprocess.DEVNULL, stdout=log, stderr=log, start_new_session=True,
                         cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        deadline = time.monotonic() + 180.0             # (the first `import torch`-free HIP start-up of a fresh box takes a while)
        while time.monotonic() < deadline:
            try:
                return RemoteChunk(fl, sigma, path)
            except (FileNotFoundError, ConnectionRefusedError):
                time.sleep(0.05)
    raise _lib.PsoapError(f"the GPU server did not come up on {path} (see {path}.log)")


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(description="Own one GPU and evaluate the likelihood calls of PSOAP's worker processes on it.")
    ap.add_argument("--device", type=int, default=None)
    ap.add_argument("--socket", default=None)
    ap.add_argument("--idle-exit", type=float, default=60.0, help="seconds without clients after which the server exits (0: never)")
    ap.add_argument("--window-us", type=float, default=1e6 * WINDOW_S, help="how long a request waits for the others of its iteration")
    args = ap.parse_args(argv)
    from . import _lib
    dev = _lib.default_device() if args.device is None else args.device
    path = args.socket or socket_path(dev)
    srv = GpuServer(path, _DeviceBackend(dev), idle_exit_s=args.idle_exit, window_s=1e-6 * args.window_us)
    print(f"psoap GPU server: device {dev}, socket {path}", flush=True)
    srv.serve()


if __name__ == "__main__":
    main()
