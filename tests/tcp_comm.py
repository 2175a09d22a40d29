"""A host-side communicator with vireo_amd.dist's three-method interface (allgather, bcast,
barrier) over plain TCP sockets -- test infrastructure for running the restart shard of
``vireo_wrap`` with TWO OR MORE RANKS ON ONE GPU.

RCCL refuses two ranks on the same device, and the 1-GPU test box has one; what the shard needs
from a communicator is an all-gather of n_init doubles and a broadcast of the winner's state,
so the ranks (one process each, all on device 0, every fit on the real kernels) talk through
rank 0 here: star topology, length-prefixed float64 frames.  No PyTorch: torch next to
libvireo_hip.so would bring a second HIP runtime into the process (vireo_amd/dist.py).
"""
import socket
import struct
import time

import numpy as np

_HELLO = b"VRXTCP1"


def _send(sock, arr):
    raw = np.ascontiguousarray(arr, dtype=np.float64).tobytes()
    sock.sendall(struct.pack("<q", len(raw)) + raw)


def _recv_exact(sock, n):
    buf = bytearray()
    while len(buf) < n:
        part = sock.recv(min(1 << 20, n - len(buf)))
        if not part:
            raise ConnectionError("peer closed the connection")
        buf += part
    return bytes(buf)


def _recv(sock):
    (n,) = struct.unpack("<q", _recv_exact(sock, 8))
    return np.frombuffer(_recv_exact(sock, n), dtype=np.float64).copy()


class TcpComm:
    """rank 0 listens on (addr, port); ranks 1 .. world-1 connect and stay connected."""

    def __init__(self, rank, world, port, addr="127.0.0.1", timeout=300.0):
        self.rank, self.world = int(rank), int(world)
        self._peers = {}          # rank 0: {rank: socket}; others: {0: socket}
        if self.world == 1:
            return
        if self.rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((addr, port))
            srv.listen(self.world)
            srv.settimeout(timeout)
            try:
                while len(self._peers) < self.world - 1:
                    conn, _ = srv.accept()
                    conn.settimeout(timeout)
                    msg = _recv_exact(conn, len(_HELLO) + 4)
                    peer = int.from_bytes(msg[len(_HELLO):], "little")
                    if msg[:len(_HELLO)] != _HELLO or not 0 < peer < self.world or peer in self._peers:
                        conn.close()
                        continue
                    conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    self._peers[peer] = conn
            finally:
                srv.close()
        else:
            deadline = time.time() + timeout
            while True:
                try:
                    s = socket.create_connection((addr, port), timeout=5.0)
                    break
                except OSError:
                    if time.time() > deadline:
                        raise TimeoutError("rank 0 never listened on %s:%d" % (addr, port))
                    time.sleep(0.1)
            s.settimeout(timeout)
            s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            s.sendall(_HELLO + self.rank.to_bytes(4, "little"))
            self._peers[0] = s

    def allgather(self, local):
        local = np.ascontiguousarray(local, dtype=np.float64).ravel()
        if self.world == 1:
            return local.copy()
        if self.rank == 0:
            parts = [local] + [_recv(self._peers[r]) for r in range(1, self.world)]
            out = np.concatenate(parts)
            for r in range(1, self.world):
                _send(self._peers[r], out)
            return out
        _send(self._peers[0], local)
        return _recv(self._peers[0])

    def bcast(self, arr, root):
        a = np.ascontiguousarray(arr, dtype=np.float64)
        if self.world == 1:
            return a.copy()
        root = int(root)
        if self.rank == 0:
            flat = a.ravel() if root == 0 else _recv(self._peers[root])
            for r in range(1, self.world):
                if r != root:
                    _send(self._peers[r], flat)
            return flat.reshape(a.shape).copy()
        if self.rank == root:
            _send(self._peers[0], a.ravel())
            return a.copy()
        return _recv(self._peers[0]).reshape(a.shape)

    def barrier(self):
        self.allgather(np.zeros(1))

    def close(self):
        for s in self._peers.values():
            try:
                s.close()
            except OSError:
                pass
        self._peers = {}
