"""One process per GPU without an external launcher.

The reference takes its parallelism as an ARGUMENT (``nproc`` -> multiprocessing.Pool over the
restarts, vireoSNP/utils/vireo_wrap.py:74-91; ``-p`` on the command line, vireo.py:82-83); the
counterpart here is ``--nGPU N`` / ``bench.py --gpus N``: the command re-executes itself N times
-- rank r on GPU r, ``RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT`` set the way
``torch.distributed.run`` would set them, so a command started by that launcher and one started by
this module take the same code path (vireo_amd/dist.py: the RCCL unique id travels over a plain
socket on MASTER_PORT + 1).  Rank 0's stdout is this process's stdout (one JSON line of bench.py,
the prints of ``vireo``); the other ranks' stdout goes to stderr.  The return code is non-zero if
any rank fails; when one fails the others are stopped instead of waiting for it in a collective.
"""
import os
import socket
import subprocess
import sys
import time

LAUNCH_ENV = "VIREO_LAUNCHED"        # set in every spawned rank (so that a rank never re-spawns)


def launched_externally():
    """True inside a rank of torch.distributed.run / spawn_ranks / any launcher that set the
    rendezvous variables"""
    return "WORLD_SIZE" in os.environ and "RANK" in os.environ


def free_port(addr="127.0.0.1", span=3):
    """a port p such that p, p + 1 and p + 2 are free now: MASTER_PORT, the RCCL unique-id port
    (dist.socket_exchange: p + 1) and the rendezvous of the one-device TcpComm harness (p + 2)"""
    for _ in range(64):
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
            s.bind((addr, 0))
            p = s.getsockname()[1]
        if p + span - 1 > 65535:
            continue
        try:
            for q in range(p + 1, p + span):
                with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s2:
                    s2.bind((addr, q))
            return p
        except OSError:
            continue
    raise OSError("no %d adjacent free ports on %s" % (span, addr))


def rank_env(rank, world, port, addr="127.0.0.1", base=None, devices=None):
    """the environment of rank ``rank``; ``devices`` maps rank -> GPU (default: rank r on GPU r)"""
    env = dict(os.environ if base is None else base)
    dev = rank if devices is None else devices[rank]
    env.update(RANK=str(rank), LOCAL_RANK=str(dev), WORLD_SIZE=str(world),
               LOCAL_WORLD_SIZE=str(world), MASTER_ADDR=addr, MASTER_PORT=str(port))
    env[LAUNCH_ENV] = "1"
    # dmabuf IPC (the image exports this already; kept for environments that were built by hand:
    # without it RCCL's hipIpcGetMemHandle fails on hosts whose driver only supports dmabuf IPC)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.pop("VIREO_RDZV_PORT", None)
    return env


def spawn_ranks(argv, world, devices=None, poll=0.05, grace=5.0, env=None):
    """Run ``argv`` (a full command line: [sys.executable, script, ...]) once per rank and wait.
    -> 0 when every rank returned 0, else the first non-zero return code (the other ranks are
    terminated: they would wait in a collective for a peer that is gone)."""
    port = free_port()
    procs = []
    for r in range(world):
        out = None if r == 0 else sys.stderr
        procs.append(subprocess.Popen(argv, env=rank_env(r, world, port, base=env, devices=devices),
                                      stdout=out))
    rc = 0
    try:
        live = set(range(world))
        while live:
            for r in sorted(live):
                code = procs[r].poll()
                if code is None:
                    continue
                live.discard(r)
                if code != 0 and rc == 0:
                    rc = code
                    sys.stderr.write("[vireo_amd.launch] rank %d exited with code %d; stopping "
                                     "the other ranks\n" % (r, code))
                    for q in live:
                        procs[q].terminate()
            if live:
                time.sleep(poll)
    finally:
        # (reached with live ranks only when this process is being torn down -- KeyboardInterrupt,
        #  an exception above: ask them to stop, then insist)
        for p in procs:
            if p.poll() is None:
                p.terminate()
        deadline = time.time() + grace
        for p in procs:
            if p.poll() is None:
                try:
                    p.wait(timeout=max(0.0, deadline - time.time()))
                except subprocess.TimeoutExpired:
                    p.kill()
                    p.wait()
    return rc


def relaunch_self(world, argv=None, module=None):
    """Re-execute the running command once per rank (``python script args`` or, with ``module``,
    ``python -m module args``) and exit with the ranks' return code."""
    argv = sys.argv[1:] if argv is None else list(argv)
    cmd = [sys.executable] + (["-m", module] if module else [os.path.abspath(sys.argv[0])]) + argv
    sys.stdout.flush()
    sys.exit(spawn_ranks(cmd, world))
