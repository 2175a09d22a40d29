"""Restart sharding across the GPUs of one node.

The reference farms the n_init random restarts of vireo_wrap to a
``multiprocessing.Pool`` and keeps ``argmax(ELBO_[-1])`` (vireoSNP/utils/vireo_wrap.py:74-91).
Restarts are independent, so here they are sharded one process per GPU -- restart i runs
on rank i % world -- and the ONLY exchange on the path is an all-gather of the per-restart
ELBOs (a few hundred bytes over RCCL/xGMI) plus a broadcast of the winner's state.

Backends
  RcclComm   libvireo_hip.so's RCCL communicator (one rank per GPU; the production path)
  TcpComm    the same three calls over host sockets (star through rank 0) for ranks that SHARE a
             device, which RCCL refuses: ``VIREO_COMM=tcp`` (the one-GPU test box runs the shard
             at world 2 and 8 this way, every fit on the real kernels).  A ONE-DEVICE TEST HARNESS,
             never a production backend: whatever it produces is labelled ``backend: "tcp"``
             (``comm_record``), and bench.py refuses it for a multi-GPU headline unless ``--comm tcp``
             was given on its own command line.
  LocalComm  world size 1
Every communicator carries ``backend`` ("rccl" | "tcp" | "local"); ``comm_record`` is the
self-description bench.py prints (which backend, which ranks on which physical GPUs).
``make_comm`` picks one from the environment a launcher left (torch.distributed.run or
vireo_amd/launch.py).  (The CPU-only tests of the sharding logic bring their own gloo
communicator with the same three methods: tests/gloo_comm.py)
"""
import ctypes as C
import os
import socket
import struct
import sys
import time

import numpy as np

from . import _lib


def env_rank_world():
    """(rank, world, device).  The device is the one ``counts.default_device`` gives every
    problem and model of this process (VIREO_DEVICE, else LOCAL_RANK, else 0), so that the RCCL
    communicator always lives on the GPU that holds the fits."""
    from .counts import default_device
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            default_device())


class LocalComm:
    rank, world = 0, 1
    backend = "local"

    def allgather(self, local):
        return np.asarray(local, dtype=np.float64).copy()

    def bcast(self, arr, root):
        return arr

    def barrier(self):
        pass


def _tail(path, n=30):
    try:
        with open(path, errors="replace") as f:
            return "".join(f.readlines()[-n:])
    except OSError:
        return ""


class RcclComm:
    """RCCL communicator behind the C ABI.  ``exchange(bytes_or_None) -> bytes`` must hand
    rank 0's 128-byte unique id to every rank (any out-of-band channel: torch.distributed
    store, a file, a socket).

    First contact with a new node must fail FAST and say why: librccl logs at NCCL_DEBUG=WARN into
    a per-rank file (unless the caller configured NCCL_DEBUG / NCCL_DEBUG_FILE), a failing
    ``ncclCommInitRank`` raises with RCCL's error string, its detail text and the tail of that
    file, the unique-id exchange gives up after VIREO_RDZV_TIMEOUT (default 120 s), and a watchdog
    ends a rank whose ``ncclCommInitRank`` neither returns nor fails within ``init_timeout`` seconds
    (VIREO_COMM_INIT_TIMEOUT, default 120; exit code 3) -- a peer that died behind the rendezvous
    would otherwise leave the others inside that call for good."""
    backend = "rccl"

    def __init__(self, rank, world, device, exchange, init_timeout=None):
        import threading
        self.rank, self.world, self.device = rank, world, device
        self.timing = {}
        self._h = C.c_void_p()
        self._log = None
        if "NCCL_DEBUG" not in os.environ and "NCCL_DEBUG_FILE" not in os.environ:
            import tempfile
            self._log = os.path.join(tempfile.gettempdir(), "vireo_rccl_rank%d_%d.log" % (rank, os.getpid()))
            os.environ["NCCL_DEBUG"] = "WARN"
            os.environ["NCCL_DEBUG_FILE"] = self._log
        if init_timeout is None:
            init_timeout = float(os.environ.get("VIREO_COMM_INIT_TIMEOUT", "120"))

        def give_up():
            sys.stderr.write("[vireo_amd.dist] rank %d of %d: ncclCommInitRank had not returned after %.0f s; "
                             "giving up (exit code 3).\n%s"
                             % (rank, world, init_timeout, self._debug_tail()))
            sys.stderr.flush()
            os._exit(3)

        t0 = time.perf_counter()
        uid = (C.c_uint8 * _lib.UNIQUE_ID_BYTES)()
        if rank == 0:
            rc = _lib.lib().vrx_comm_unique_id(uid)
            if rc != 0:
                raise _lib.VrxError("rank 0 of %d: %s (code %d)\n%s" % (
                    world, _lib.lib().vrx_last_error().decode(), rc, self._debug_tail()))
            raw = exchange(bytes(uid))      # (raises TimeoutError after VIREO_RDZV_TIMEOUT)
        else:
            raw = exchange(None)
        uid = (C.c_uint8 * _lib.UNIQUE_ID_BYTES).from_buffer_copy(raw)
        t1 = time.perf_counter()
        dog = threading.Timer(init_timeout, give_up)
        dog.daemon = True
        dog.start()
        try:
            # librccl prints a version banner on the C stdout of rank 0 when a communicator is
            # made; callers that emit machine-readable output there (bench.py: ONE JSON line) must
            # not see it, so stdout points at stderr for the duration of the call
            sys.stdout.flush()
            keep = os.dup(1)
            os.dup2(2, 1)
            try:
                rc = _lib.lib().vrx_comm_create(device, rank, world, uid, C.byref(self._h))
            finally:
                C.CDLL(None).fflush(None)
                os.dup2(keep, 1)
                os.close(keep)
            if rc != 0:
                raise _lib.VrxError("rank %d of %d: %s (code %d)\n%s" % (
                    rank, world, _lib.lib().vrx_last_error().decode(), rc, self._debug_tail()))
            self.timing = {"unique_id_exchange_ms": (t1 - t0) * 1e3,
                           "comm_init_rank_ms": (time.perf_counter() - t1) * 1e3}
        finally:
            dog.cancel()

    def _debug_tail(self):
        path = self._log or os.environ.get("NCCL_DEBUG_FILE", "")
        text = _tail(path) if path else ""
        return ("---- NCCL_DEBUG=%s tail (%s) ----\n%s" % (os.environ.get("NCCL_DEBUG"), path, text)
                if text else "(no NCCL debug output%s)\n" % (" in " + path if path else ""))

    def info(self):
        a = (C.c_int32 * 4)()
        _lib.check(_lib.lib().vrx_comm_info(self._h, a))
        v = int(a[3])
        return dict(rank=int(a[0]), world=int(a[1]), device=int(a[2]), rccl_version_code=v,
                    rccl_version="%d.%d.%d" % (v // 10000, v // 100 % 100, v % 100) if v else None)

    def allgather(self, local):
        local = _lib.f64(local).ravel()
        out = np.empty(local.size * self.world)
        _lib.check(_lib.lib().vrx_comm_allgather_f64(self._h, _lib.dptr(local), local.size,
                                                     _lib.dptr(out)))
        return out

    def bcast(self, arr, root):
        buf = _lib.f64(arr).copy()
        flat = buf.reshape(-1)
        _lib.check(_lib.lib().vrx_comm_bcast_f64(self._h, _lib.dptr(flat), flat.size, int(root)))
        return buf

    def bcast_model(self, dm, root):
        """the variational state of device model ``dm`` <- rank ``root``'s, DEVICE TO DEVICE
        (vrx_comm_bcast_model: one RCCL group call on the models' own HBM buffers)"""
        _lib.check(_lib.lib().vrx_comm_bcast_model(self._h, dm._h, int(root)))

    def barrier(self):
        _lib.check(_lib.lib().vrx_comm_barrier(self._h))

    def close(self):
        if self._h:
            _lib.lib().vrx_comm_destroy(self._h)
            self._h = C.c_void_p()
        if self._log:
            try:
                os.remove(self._log)
            except OSError:
                pass
            self._log = None


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


_MAX_MESSAGE = 1 << 33      # 8 GiB of doubles: far beyond any state this path exchanges


def _recv(sock):
    (n,) = struct.unpack("<q", _recv_exact(sock, 8))
    if n < 0 or n > _MAX_MESSAGE or n % 8:
        raise ConnectionError("peer announced a message of %d bytes" % n)
    return np.frombuffer(_recv_exact(sock, n), dtype=np.float64).copy()


class TcpComm:
    """rank 0 listens on (addr, port); ranks 1 .. world-1 connect and stay connected."""

    backend = "tcp"

    def __init__(self, rank, world, port, addr="127.0.0.1", timeout=600.0, rdzv_timeout=None):
        """``rdzv_timeout`` bounds the rendezvous (VIREO_RDZV_TIMEOUT, default 120 s: the ranks of one
        launch arrive within seconds of each other); ``timeout`` every later receive -- a collective
        waits for the slowest rank's fits."""
        self.rank, self.world = int(rank), int(world)
        if rdzv_timeout is None:
            rdzv_timeout = float(os.environ.get("VIREO_RDZV_TIMEOUT", "120"))
        self._peers = {}          # rank 0: {rank: socket}; others: {0: socket}
        if self.world == 1:
            return
        if self.rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((addr, port))
            srv.listen(self.world)
            deadline = time.time() + rdzv_timeout
            try:
                while len(self._peers) < self.world - 1:
                    srv.settimeout(max(0.1, deadline - time.time()))
                    try:
                        conn, _ = srv.accept()
                    except socket.timeout:
                        raise TimeoutError("ranks %s never connected to %s:%d" % (
                            sorted(set(range(1, self.world)) - set(self._peers)), addr, port))
                    # anything that is not a rank saying hello (a port probe, a stale peer, a
                    # connection that closes or stays silent) is dropped without aborting rank 0
                    try:
                        conn.settimeout(5.0)
                        msg = _recv_exact(conn, len(_HELLO) + 4)
                    except (OSError, ConnectionError):
                        conn.close()
                        continue
                    peer = int.from_bytes(msg[len(_HELLO):], "little")
                    if msg[:len(_HELLO)] != _HELLO or not 0 < peer < self.world or peer in self._peers:
                        conn.close()
                        continue
                    conn.settimeout(timeout)
                    conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    self._peers[peer] = conn
            finally:
                srv.close()
        else:
            deadline = time.time() + rdzv_timeout
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


def make_comm(rank=None, world=None, device=None, force_rccl=False):
    """The communicator the environment asks for: LocalComm at world 1 (unless ``force_rccl``
    or VIREO_FORCE_RCCL=1: the RCCL path on a 1-GPU box), TcpComm with VIREO_COMM=tcp (rendezvous on
    MASTER_ADDR : VIREO_TCP_PORT or MASTER_PORT + 2), else RcclComm (unique id over MASTER_PORT + 1)."""
    r, w, d = env_rank_world()
    rank = r if rank is None else rank
    world = w if world is None else world
    device = d if device is None else device
    kind = os.environ.get("VIREO_COMM", "rccl").lower()
    if kind not in ("rccl", "tcp"):
        raise _lib.VrxError("VIREO_COMM must be 'rccl' or 'tcp', not %r" % kind)
    if kind == "tcp" and world > 1:
        port = int(os.environ.get("VIREO_TCP_PORT", int(os.environ.get("MASTER_PORT", "29500")) + 2))
        return TcpComm(rank, world, port, addr=os.environ.get("MASTER_ADDR", "127.0.0.1"))
    if world > 1 or force_rccl or os.environ.get("VIREO_FORCE_RCCL") == "1":
        return RcclComm(rank, world, device, socket_exchange(rank, world))
    return LocalComm()


def socket_exchange(rank, world, addr=None, port=None, timeout=None):
    """unique-id exchange over a plain TCP socket: rank 0 serves the id on
    (MASTER_ADDR, VIREO_RDZV_PORT or MASTER_PORT + 1) -- bound to that address only -- and
    every other rank fetches it after announcing its rank.
    No PyTorch involved -- importing torch next to libvireo_hip.so puts a second HIP runtime
    (and a second librccl) into the process, and RCCL initialisation then fails."""
    if timeout is None:     # the ranks of one launch arrive within seconds of each other
        timeout = float(os.environ.get("VIREO_RDZV_TIMEOUT", "120"))
    addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
    if port is None:
        port = int(os.environ.get("VIREO_RDZV_PORT",
                                  int(os.environ.get("MASTER_PORT", "29500")) + 1))

    hello = b"VRXID1"

    def exchange(raw):
        if world == 1:
            return raw
        if rank == 0:
            # serve the id once to every distinct rank that says hello; anything else that
            # connects (a port probe, a stale peer) is dropped without using up a slot
            with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as srv:
                srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                srv.bind((addr, port))
                srv.listen(world)
                served, deadline = set(), time.time() + timeout
                while len(served) < world - 1:
                    srv.settimeout(max(0.1, deadline - time.time()))
                    try:
                        conn, _peer = srv.accept()
                    except socket.timeout:
                        raise TimeoutError("ranks %s never asked for the RCCL unique id"
                                           % sorted(set(range(1, world)) - served))
                    with conn:
                        try:
                            conn.settimeout(5.0)
                            msg = b""
                            while len(msg) < len(hello) + 4:
                                part = conn.recv(len(hello) + 4 - len(msg))
                                if not part:
                                    break
                                msg += part
                            peer = int.from_bytes(msg[len(hello):], "little") if len(msg) == len(hello) + 4 else -1
                            if msg[:len(hello)] == hello and 0 < peer < world and peer not in served:
                                conn.sendall(raw)
                                served.add(peer)
                        except OSError:
                            pass
            return raw
        deadline = time.time() + timeout
        while True:
            try:
                with socket.create_connection((addr, port), timeout=10.0) as s:
                    s.sendall(hello + int(rank).to_bytes(4, "little"))
                    buf = b""
                    while len(buf) < _lib.UNIQUE_ID_BYTES:
                        chunk = s.recv(_lib.UNIQUE_ID_BYTES - len(buf))
                        if not chunk:
                            break
                        buf += chunk
                if len(buf) == _lib.UNIQUE_ID_BYTES:
                    return buf
            except OSError:
                pass
            if time.time() > deadline:
                raise TimeoutError("no RCCL unique id from rank 0 at %s:%d" % (addr, port))
            time.sleep(0.2)
    return exchange


# ---- the sharding arithmetic (pure host logic, unit-tested on CPU) ---------------------
def my_restarts(n_init, rank, world):
    """restart indices fitted by ``rank``: i % world == rank."""
    return list(range(rank, n_init, world))


def gather_restart_elbos(comm, n_init, local_elbos):
    """local_elbos: {restart index: ELBO_[-1]} of this rank -> full (n_init,) array, the
    same on every rank, in restart order (so np.argmax's first-max rule matches the
    reference's, vireo_wrap.py:90-91)."""
    per = (n_init + comm.world - 1) // comm.world
    send = np.full(per, -np.inf)
    for j, i in enumerate(my_restarts(n_init, comm.rank, comm.world)):
        send[j] = local_elbos[i]
    got = np.asarray(comm.allgather(send)).reshape(comm.world, per)
    out = np.empty(n_init)
    for i in range(n_init):
        out[i] = got[i % comm.world, i // comm.world]
    return out


def first_record(elbos):
    """The initialisation ``BinomMixtureVB.fit`` keeps (bmm_model.py:248-252): the LAST i with
    ``i == 0 or elbo[i] > np.max(elbo[:i])`` -- the first maximum, and with NumPy's NaN semantics
    (a NaN before i makes ``np.max`` NaN and every later comparison False)."""
    elbos = np.asarray(elbos, dtype=np.float64)
    best = 0
    for i in range(1, len(elbos)):
        if elbos[i] > np.max(elbos[:i]):
            best = i
    return best


def comm_record(comm, device, n_init=32, repeats=5):
    """What a multi-GPU result must say about itself (bench.py's ``comm`` block): the backend,
    the world size, RCCL's version, every rank's device and PHYSICAL GPU (PCI bus id, all-gathered
    through the communicator itself), the time of the unique-id exchange / ncclCommInitRank, and one
    timed all-gather of ``n_init`` doubles -- the only exchange on the restart shard's path.
    Collective: every rank calls it; every rank gets the same record."""
    pci = _lib.device_pci_bus_id(device)
    try:                                   # "0000:c1:00.0" -> domain, bus, device, function
        dom, bus, rest = pci.split(":")
        dv, fn = rest.split(".")
        nums = [int(dom, 16), int(bus, 16), int(dv, 16), int(fn, 16)]
    except ValueError:
        nums = [-1, -1, -1, -1]
    import resource
    rss_mb = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024.0     # (peak host memory of this rank so far)
    mine = np.array([comm.rank, device, os.getpid()] + nums + [rss_mb], dtype=np.float64)
    rows = np.asarray(comm.allgather(mine)).reshape(comm.world, -1)
    ranks = []
    for r in rows:
        d = [int(x) for x in r]
        ranks.append(dict(rank=d[0], device=d[1], pid=d[2],
                          pci_bus_id="%04x:%02x:%02x.%x" % tuple(d[3:7]) if d[3] >= 0 else None,
                          host_peak_rss_mb=d[7]))
    per = -(-n_init // comm.world)
    us = []
    for _ in range(repeats):
        comm.barrier()
        t0 = time.perf_counter()
        comm.allgather(np.zeros(per))
        us.append((time.perf_counter() - t0) * 1e6)
    rec = dict(backend=comm.backend, world=comm.world, ranks=ranks,
               distinct_gpus=len({r["pci_bus_id"] for r in ranks}),
               allgather_us=dict(doubles_per_rank=per, n_init=n_init, median=float(np.median(us)),
                                 min=float(min(us)), runs=[round(x, 1) for x in us]))
    if isinstance(comm, RcclComm):
        inf = comm.info()
        rec.update(rccl_version=inf["rccl_version"], rccl_version_code=inf["rccl_version_code"],
                   **{k: round(v, 2) for k, v in comm.timing.items()})
    return rec
