"""Host-side rendezvous of the ranks of ONE node over a loopback TCP socket -- no torch, no MPI.

The ranks of a data-parallel job (one process per GPU) need three host-side exchanges and nothing else: the 128-byte
``ncclUniqueId`` from rank 0 to everybody before ``scann_comm_init``, a barrier, and small reductions of Python numbers
(the max of a wall time in ``bench.py``).  Everything on the data path goes over RCCL inside ``libscann_hip.so``.

Rank 0 listens on the first free port of a fixed candidate list derived from ``MASTER_PORT`` (the port itself belongs to
the launcher: ``torch.distributed.run`` keeps its own store there); the other ranks walk the same list until a server
answers the handshake token of THIS job.  Works under ``torch.distributed.run`` (RANK / WORLD_SIZE / MASTER_ADDR /
MASTER_PORT in the environment) and under ``scann.parallel.launch.spawn_ranks`` (same variables, set by the parent).
"""
from __future__ import annotations

import hashlib
import os
import pickle
import socket
import struct
import time

_MAGIC = b"SCANNRDZ"


def _candidates(master_port):
    base = 20000 + (int(master_port) * 7 + 1009) % 20000
    return [base + 13 * k for k in range(24)]


def _token(addr, port, world):
    run = os.environ.get("TORCHELASTIC_RUN_ID", "") + "|" + os.environ.get("SCANN_RDZV_ID", "")
    return hashlib.sha256(("%s|%s|%d|%s" % (addr, port, world, run)).encode()).digest()[:16]


def _send(sock, obj):
    data = pickle.dumps(obj, protocol=4)
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
    return pickle.loads(_recv_exact(sock, n))


class Rendezvous:
    """``Rendezvous()`` reads RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT; world 1 needs no socket at all."""

    def __init__(self, rank=None, world=None, addr=None, port=None, timeout=120.0):
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else int(rank)
        self.world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else int(world)
        self.addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
        self.port = int(port if port is not None else os.environ.get("MASTER_PORT", "29500"))
        self._peers = []   # rank 0: sockets of ranks 1..world-1, by rank
        self._sock = None  # other ranks: socket to rank 0
        if self.world <= 1:
            return
        tok = _token(self.addr, self.port, self.world)
        deadline = time.time() + timeout
        if self.rank == 0:
            srv = None
            for p in _candidates(self.port):
                s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                try:
                    s.bind((self.addr, p))
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
                if hello[:len(_MAGIC)] != _MAGIC or hello[len(_MAGIC):-4] != tok or not 0 < r < self.world or r in peers:
                    c.close()  # not a rank of this job
                    continue
                c.sendall(b"OK")
                c.settimeout(None)
                c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                peers[r] = c
            srv.close()
            self._peers = [peers[r] for r in range(1, self.world)]
        else:
            hello = _MAGIC + tok + struct.pack("<i", self.rank)
            while self._sock is None:
                for p in _candidates(self.port):
                    try:
                        s = socket.create_connection((self.addr, p), timeout=2.0)
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
