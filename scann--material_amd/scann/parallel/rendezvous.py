"""Host-side rendezvous of the ranks of ONE node over a loopback TCP socket -- no torch, no MPI.

The ranks of a data-parallel job (one process per GPU) need three host-side exchanges and nothing else: the 128-byte
``ncclUniqueId`` from rank 0 to everybody before ``scann_comm_init``, a barrier, and small reductions of Python numbers
(the max of a wall time in ``bench.py``).  Everything on the data path goes over RCCL inside ``libscann_hip.so``.

Rank 0 listens on ``MASTER_ADDR`` when that is a loopback address (the ranks of one node: the normal case) -- otherwise on that
address too, so that a multi-node ``torch.distributed.run`` job meets as well (``SCANN_RDZV_BIND`` overrides where rank 0 binds;
the other ranks always CONNECT to ``MASTER_ADDR``), and then the per-job secret must come from the environment
(``SCANN_RDZV_SECRET``, identical on every node): the 0600 key file of the single-node case lives in one host's /tmp.  The port is
the first free one of a fixed candidate list
derived from ``MASTER_PORT`` (the port itself belongs to the launcher: ``torch.distributed.run`` keeps its own store there);
the other ranks walk the same list until a server answers the handshake token of THIS job.  The token is a hash over the
job's coordinates AND a random per-job secret: ``SCANN_RDZV_SECRET`` when the launcher provides one
(``scann.parallel.launch.spawn_ranks`` does), otherwise a file of mode 0600 that rank 0 writes and the other ranks of the
same user read.  Messages are length-prefixed JSON (numbers, lists, None, bytes as hex) -- nothing received from the
socket is ever unpickled or evaluated.  Works under ``torch.distributed.run`` (RANK / WORLD_SIZE / MASTER_ADDR /
MASTER_PORT in the environment) and under ``spawn_ranks`` (same variables, set by the parent).
"""
from __future__ import annotations

import hashlib
import hmac
import json
import os
import secrets
import socket
import struct
import tempfile
import time

_MAGIC = b"SCANNRDZ"


def _candidates(master_port):
    base = 20000 + (int(master_port) * 7 + 1009) % 20000
    return [base + 13 * k for k in range(24)]


def _token(addr, port, world, secret):
    run = os.environ.get("TORCHELASTIC_RUN_ID", "") + "|" + os.environ.get("SCANN_RDZV_ID", "")
    return hashlib.sha256(("%s|%s|%d|%s|%s" % (addr, port, world, run, secret)).encode()).digest()[:16]


def _secret_path(addr, port, world):
    run = os.environ.get("TORCHELASTIC_RUN_ID", "") + "|" + os.environ.get("SCANN_RDZV_ID", "")
    tag = hashlib.sha256(("%s|%s|%d|%s" % (addr, port, world, run)).encode()).hexdigest()[:16]
    return os.path.join(tempfile.gettempdir(), "scann_rdzv_%d_%s.key" % (os.getuid(), tag))


def _write_secret(path):
    """Rank 0: a fresh random secret in a file only this user can read (replaces a stale file of an earlier job)."""
    value = secrets.token_hex(32)
    try:
        os.unlink(path)
    except FileNotFoundError:
        pass
    fd = os.open(path, os.O_WRONLY | os.O_CREAT | os.O_EXCL, 0o600)
    with os.fdopen(fd, "w") as f:
        f.write(value)
    return value


def _read_secret(path):
    try:
        st = os.stat(path)
        if st.st_uid != os.getuid() or (st.st_mode & 0o077):
            return None  # not ours, or readable by others: never trust it
        with open(path) as f:
            value = f.read().strip()
        return value if len(value) == 64 else None
    except OSError:
        return None


def _enc(obj):
    if isinstance(obj, (bytes, bytearray)):
        return {"__bytes__": bytes(obj).hex()}
    if isinstance(obj, (list, tuple)):
        return [_enc(x) for x in obj]
    if obj is None or isinstance(obj, (bool, int, float, str)):
        return obj
    if hasattr(obj, "item") and getattr(obj, "shape", None) == ():  # NumPy scalar
        return obj.item()
    if hasattr(obj, "tolist"):
        return _enc(obj.tolist())
    raise TypeError("rendezvous: cannot send a %s (numbers, strings, None, bytes and lists of them only)" % type(obj).__name__)


def _dec(obj):
    if isinstance(obj, dict):
        if set(obj) != {"__bytes__"} or not isinstance(obj["__bytes__"], str):
            raise ValueError("rendezvous: malformed message")
        return bytes.fromhex(obj["__bytes__"])
    if isinstance(obj, list):
        return [_dec(x) for x in obj]
    return obj


_MAX_MESSAGE = 64 << 20


def _send(sock, obj):
    data = json.dumps(_enc(obj), allow_nan=True).encode()
    sock.sendall(struct.pack("<Q", len(data)) + data)


def _recv_exact(sock, n):
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(n - len(buf))
        if not chunk:
            raise ConnectionError("rendezvous peer closed the connection")
        buf += chunk
    return bytes(buf)


def _recv(sock):
    (n,) = struct.unpack("<Q", _recv_exact(sock, 8))
    if n > _MAX_MESSAGE:
        raise ConnectionError("rendezvous: oversized message")
    return _dec(json.loads(_recv_exact(sock, n).decode()))


class Rendezvous:
    """``Rendezvous()`` reads RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT; world 1 needs no socket at all."""

    def __init__(self, rank=None, world=None, addr=None, port=None, timeout=120.0):
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else int(rank)
        self.world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else int(world)
        self.addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
        self.port = int(port if port is not None else os.environ.get("MASTER_PORT", "29500"))
        self._peers = []   # rank 0: sockets of ranks 1..world-1, by rank
        self._sock = None  # other ranks: socket to rank 0
        self._secret_file = None
        if self.world <= 1:
            return
        loop = self.addr in ("localhost", "ip6-localhost") or self.addr.startswith("127.") or self.addr == "::1"
        # rank 0 binds where the others will look for it: MASTER_ADDR (loopback for the ranks of one node)
        bind = os.environ.get("SCANN_RDZV_BIND", "127.0.0.1" if loop else self.addr)
        connect = "127.0.0.1" if loop else self.addr
        env_secret = os.environ.get("SCANN_RDZV_SECRET")
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(self.world)) or self.world)
        if not env_secret and (self.world > local_world or not loop):
            raise RuntimeError("rendezvous: a job that spans nodes (WORLD_SIZE %d > LOCAL_WORLD_SIZE %d, or MASTER_ADDR %s is not "
                               "loopback) needs SCANN_RDZV_SECRET exported identically on every node -- the fallback key file is "
                               "local to one host" % (self.world, local_world, self.addr))
        spath = _secret_path(self.addr, self.port, self.world)
        deadline = time.time() + timeout
        if self.rank == 0:
            tok = _token(self.addr, self.port, self.world, env_secret if env_secret else _write_secret(spath))
            self._secret_file = None if env_secret else spath
            srv = None
            for p in _candidates(self.port):
                s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                try:
                    s.bind((bind, p))
                    srv = s
                    break
                except OSError:
                    s.close()
            if srv is None:
                raise RuntimeError("rendezvous: no free port among the candidates of MASTER_PORT %d" % self.port)
            srv.listen(self.world)
            peers = {}
            while len(peers) < self.world - 1:
                srv.settimeout(max(0.1, deadline - time.time()))
                try:
                    c, _ = srv.accept()
                except socket.timeout:
                    raise TimeoutError("rendezvous: %d of %d ranks arrived" % (len(peers) + 1, self.world))
                c.settimeout(10.0)
                try:
                    hello = _recv_exact(c, len(_MAGIC) + 16 + 4)
                except (OSError, ConnectionError):
                    c.close()
                    continue
                r = struct.unpack("<i", hello[-4:])[0]
                if hello[:len(_MAGIC)] != _MAGIC or not hmac.compare_digest(hello[len(_MAGIC):-4], tok) or not 0 < r < self.world or r in peers:
                    c.close()  # not a rank of this job
                    continue
                c.sendall(b"OK")
                c.settimeout(None)
                c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                peers[r] = c
            srv.close()
            self._peers = [peers[r] for r in range(1, self.world)]
        else:
            while self._sock is None:
                secret = env_secret if env_secret else _read_secret(spath)  # re-read every round: rank 0 may not have written it yet
                if secret is None:
                    if time.time() > deadline:
                        raise TimeoutError("rendezvous: rank %d found no job secret (%s)" % (self.rank, spath))
                    time.sleep(0.05)
                    continue
                hello = _MAGIC + _token(self.addr, self.port, self.world, secret) + struct.pack("<i", self.rank)
                for p in _candidates(self.port):
                    try:
                        s = socket.create_connection((connect, p), timeout=2.0)
                        s.sendall(hello)
                        if _recv_exact(s, 2) == b"OK":
                            s.settimeout(None)
                            s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                            self._sock = s
                            break
                        s.close()
                    except (OSError, ConnectionError):
                        pass
                if self._sock is None:
                    if time.time() > deadline:
                        raise TimeoutError("rendezvous: rank %d found no rank-0 server" % self.rank)
                    time.sleep(0.05)

    # -- collectives on small Python objects (rank 0 is the hub) ---------------------------------------------------
    def gather(self, obj):
        """-> list of every rank's object on rank 0, None elsewhere."""
        if self.world <= 1:
            return [obj]
        if self.rank == 0:
            return [obj] + [_recv(s) for s in self._peers]
        _send(self._sock, obj)
        return None

    def broadcast(self, obj):
        """rank 0's object -> every rank."""
        if self.world <= 1:
            return obj
        if self.rank == 0:
            for s in self._peers:
                _send(s, obj)
            return obj
        return _recv(self._sock)

    def allgather(self, obj):
        return self.broadcast(self.gather(obj))

    def barrier(self):
        self.allgather(None)

    def allreduce_max(self, x):
        return max(self.allgather(x))

    def allreduce_sum(self, x):
        return sum(self.allgather(x))

    def close(self):
        for s in self._peers + ([self._sock] if self._sock else []):
            try:
                s.close()
            except OSError:
                pass
        self._peers, self._sock = [], None
        if self._secret_file:
            try:
                os.unlink(self._secret_file)
            except OSError:
                pass
            self._secret_file = None
