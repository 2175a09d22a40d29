"""The restart shard of ``vireo_wrap`` (vireoSNP/utils/vireo_wrap.py:64-94) with TWO RANKS AND
REAL DEVICE FITS on the one GPU of the test box.

The CPU tests (tests/test_host_cpu.py) run the sharded control flow with the fits replaced by
the oracle; the RCCL test (tests/test_gpu_fullsize.py::test_restart_shard_over_rccl_world2)
needs two devices.  Here every rank is its own process on device 0 and drives the real
kernels -- ``LegacyStream.skip`` / the generator jump over the other rank's restarts,
``DeviceRestarts(n_owned=...)``, the restart batches, the owner's refinement, ``_bcast_model``
-- and the ranks exchange the n_init ELBOs and the winner's state through a host-side TCP
communicator with vireo_amd.dist's three-method interface (tests/tcp_comm.py).  What only real
RCCL at world > 1 would add is ncclCommInitRank / the xGMI transport of those two calls
(covered at world 1 by test_rccl_communicator_world1).
"""
import contextlib
import io
import os
import pickle
import socket
import subprocess
import sys

import numpy as np
import pytest

from tests import gold

pytestmark = pytest.mark.gpu
RTOL = 1e-5
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def va():
    import vireo_amd
    from vireo_amd import _lib
    _lib.require_gpu()
    return vireo_amd


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_ranks(tmp_path, world, case, n_init, env_extra=None):
    port = _free_port()
    outs = [str(tmp_path / ("%s_%d_rank%d.pkl" % (case, n_init, r))) for r in range(world)]
    procs = []
    for r in range(world):
        env = dict(os.environ, PYTHONPATH=ROOT, VIREO_DEVICE="0")
        env.pop("RANK", None), env.pop("WORLD_SIZE", None), env.pop("LOCAL_RANK", None)
        env.update(env_extra or {})
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_tcp_worker.py"),
                                       str(r), str(world), str(port), outs[r], case, str(n_init)], env=env))
    for p in procs:
        assert p.wait(timeout=900) == 0
    return [pickle.load(open(o, "rb")) for o in outs]


STATE = ("ID_prob", "GT_prob", "theta_mean", "theta_sum", "LB_list", "doublet_prob", "doublet_LLR")


def _same(a, b):
    for k in STATE:
        assert np.array_equal(a[k], b[k]), k
    assert a["LB_doublet"] == b["LB_doublet"]


def test_two_ranks_one_gpu_demo_data_n_init4(va, tmp_path, monkeypatch):
    """c1 data, n_init = 4, one restart per device model (so that world 1 and world 2 run the
    same kernel instances): both ranks return the SAME result, bitwise equal to the world-1 run
    in this process, and it is the reference's (golden c1_wrap_seed2_init4)."""
    monkeypatch.setenv("VIREO_RESTART_BATCH", "1")
    rvs = _run_ranks(tmp_path, 2, "c1", 4, {"VIREO_RESTART_BATCH": "1"})
    AD, DP = gold.c1()
    np.random.seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        one = va.vireo_wrap(AD, DP, n_donor=4, n_init=4, random_seed=2)
    rng_one = np.random.get_state()
    for rv in rvs:
        _same(rv, one)
        # the global legacy stream ends where the reference's would, on every rank
        assert np.array_equal(rv["rng_after"][0], rng_one[1][:8]) and rv["rng_after"][1] == int(rng_one[2])
    # restarts 0, 2 on rank 0 and 1, 3 on rank 1; the winner's owner refined it
    assert [rv["search"]["restarts"] for rv in rvs] == [2, 2]
    best = rvs[0]["search"]["best"]
    assert rvs[0]["search"]["owner"] == best % 2 == rvs[1]["search"]["owner"]
    assert best == int(np.argmax(one["LB_list"]))
    assert rvs[best % 2]["search"]["final_iterations"] > 0 and rvs[1 - best % 2]["search"]["final_iterations"] == 0
    assert "skip" in rvs[0]["phases"] and "skip" in rvs[1]["phases"]       # both jumped over the other's draws
    g = gold.load("c1_wrap_seed2_init4")
    np.testing.assert_allclose(one["LB_list"], g["LB_list"], rtol=RTOL)
    np.testing.assert_allclose(one["ID_prob"], g["ID_prob"], rtol=RTOL, atol=1e-290)
    np.testing.assert_allclose(one["GT_prob"], g["GT_prob"], rtol=RTOL, atol=1e-290)
    np.testing.assert_allclose(one["doublet_prob"], g["doublet_prob"], rtol=RTOL, atol=1e-290)
    assert one["LB_doublet"] == pytest.approx(g["LB_doublet"], rel=RTOL)
    assert np.array_equal(one["ID_prob"].argmax(1), g["ID_prob"].argmax(1))


def test_two_ranks_one_gpu_rank_without_restarts(va, tmp_path, monkeypatch):
    """n_init = 1 on two ranks: rank 1 owns nothing, still joins the gather and leaves with the
    winner's state (golden c1_wrap_seed2_init1)."""
    rvs = _run_ranks(tmp_path, 2, "c1", 1)
    assert [rv["search"]["restarts"] for rv in rvs] == [1, 0]
    _same(rvs[0], rvs[1])
    g = gold.load("c1_wrap_seed2_init1")
    np.testing.assert_allclose(rvs[1]["LB_list"], g["LB_list"], rtol=RTOL)
    np.testing.assert_allclose(rvs[1]["ID_prob"], g["ID_prob"], rtol=RTOL, atol=1e-290)
    np.testing.assert_allclose(rvs[1]["GT_prob"], g["GT_prob"], rtol=RTOL, atol=1e-290)
    assert rvs[1]["LB_doublet"] == pytest.approx(g["LB_doublet"], rel=RTOL)
    assert np.array_equal(rvs[1]["ID_prob"].argmax(1), g["ID_prob"].argmax(1))


def test_two_ranks_one_gpu_config2_with_restart_batches(va, tmp_path, monkeypatch):
    """BASELINE.json configs[1] size (N = 10k x M = 5k, K = 4), n_init = 6, the LDS-resident
    passes forced so that a column's sums do not depend on its batch neighbours: rank 0 packs
    restarts 0, 2, 4 into one device model, rank 1 restarts 1, 3, 5 -- the world-1 run packs all
    six differently and must still give the same bits."""
    from vireo_amd import synth
    from vireo_amd.counts import DeviceCounts
    env = {"VIREO_LDS": "1"}
    rvs = _run_ranks(tmp_path, 2, "c2", 6, env)
    assert [rv["search"]["restarts"] for rv in rvs] == [3, 3]
    assert rvs[0]["search"]["batch"] > 1
    _same(rvs[0], rvs[1])
    monkeypatch.setenv("VIREO_LDS", "1")
    N, M, K, dens = synth.CONFIGS["c2"]
    w = synth.donor_workload(N, M, K, dens, seed=0)
    counts = DeviceCounts.from_merged(w["shape"], w["colptr"], w["rowidx"], w["ad"], w["dp"])
    with contextlib.redirect_stdout(io.StringIO()):
        one = va.vireo_wrap(counts, None, n_donor=K, n_init=6, random_seed=5, check_doublet=False)
    _same(rvs[0], one)
    assert np.all(np.isfinite(one["LB_list"])) and len(set(one["LB_list"])) > 1
