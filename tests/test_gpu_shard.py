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


@pytest.mark.parametrize("n_init", [32, 13])
def test_eight_ranks_one_gpu_config2(va, tmp_path, monkeypatch, n_init):
    """north_star's world: EIGHT ranks (all on device 0), BASELINE.json configs[1] size, n_init =
    32 (4 restarts per rank, every rank jumps the generator over 28 foreign ones -- configs[3]'s
    arithmetic) and n_init = 13 (ranks 0-4 own two restarts, ranks 5-7 one: the gathered ELBO
    array is padded; owner != 0 is likely), restart batches on, LDS-resident passes forced so
    that a column's sums do not depend on its batch neighbours: every rank returns the bits of
    the world-1 run (vireo_wrap.py:64-94)."""
    from vireo_amd import synth
    from vireo_amd.counts import DeviceCounts
    from vireo_amd.dist import my_restarts
    rvs = _run_ranks(tmp_path, 8, "c2", n_init, {"VIREO_LDS": "1"})
    assert [rv["search"]["restarts"] for rv in rvs] == [len(my_restarts(n_init, r, 8)) for r in range(8)]
    assert rvs[0]["search"]["batch"] > 1
    for rv in rvs[1:]:
        _same(rvs[0], rv)
        assert np.array_equal(rv["rng_after"][0], rvs[0]["rng_after"][0]) and rv["rng_after"][1] == rvs[0]["rng_after"][1]
    monkeypatch.setenv("VIREO_LDS", "1")
    N, M, K, dens = synth.CONFIGS["c2"]
    w = synth.donor_workload(N, M, K, dens, seed=0)
    counts = DeviceCounts.from_merged(w["shape"], w["colptr"], w["rowidx"], w["ad"], w["dp"])
    with contextlib.redirect_stdout(io.StringIO()):
        one = va.vireo_wrap(counts, None, n_donor=K, n_init=n_init, random_seed=5, check_doublet=False)
    rng_one = np.random.get_state()
    _same(rvs[0], one)
    assert np.array_equal(rvs[0]["rng_after"][0], rng_one[1][:8]) and rvs[0]["rng_after"][1] == int(rng_one[2])
    best = int(np.argmax(one["LB_list"]))
    assert len(one["LB_list"]) == n_init and len(set(one["LB_list"])) > 1
    for r, rv in enumerate(rvs):
        assert rv["search"]["best"] == best and rv["search"]["owner"] == best % 8
        assert (rv["search"]["final_iterations"] > 0) == (r == best % 8)      # only the owner refined
        assert "skip" in rv["phases"]


def test_clone_mode_inits_sharded_two_ranks(va, tmp_path, monkeypatch):
    """``BinomMixtureVB.fit(comm=)`` ("BMM: same restart shard";
    bmm_model.py:242-254): initialisation i on rank i % 2, the ELBOs all-gathered, the owner of
    the first maximum re-fits and broadcasts.  One initialisation per device model: both ranks
    return the bits of the world-1 call in this process, which is the reference's
    (golden mito_bmm_k3_seed1, the notebook's known answer); with the initialisations packed
    into batches (each rank packs its 25 differently from the 50 of world 1; gather kernels pick
    their lane layout from the column count): the same to 1e-9."""
    AD, DP = gold.mito()
    g = gold.load("mito_bmm_k3_seed1")
    for batch_env, exact in (({"VIREO_RESTART_BATCH": "1"}, True), ({}, False)):
        tag = tmp_path / ("b%d" % exact)
        tag.mkdir()
        rvs = _run_ranks(tag, 2, "bmm", 50, batch_env)
        for k, v in batch_env.items():
            monkeypatch.setenv(k, v)
        b = va.BinomMixtureVB(n_var=AD.shape[0], n_cell=AD.shape[1], n_donor=3)
        b.fit(AD, DP, min_iter=30, n_init=50, random_seed=1, verbose=False)
        rng_one = np.random.get_state()
        for k in batch_env:
            monkeypatch.delenv(k)
        for rv in rvs:
            for name in ("ID_prob", "beta_mu", "beta_sum", "ELBO_iters", "ELBO_inits"):
                if exact:
                    assert np.array_equal(rv[name], getattr(b, name)), name
                else:
                    np.testing.assert_allclose(rv[name], getattr(b, name), rtol=1e-9, atol=1e-300)
                assert np.array_equal(rv[name], rvs[0][name])       # every rank leaves with the same state
            assert np.array_equal(rv["rng_after"][0], rng_one[1][:8]) and rv["rng_after"][1] == int(rng_one[2])
            assert len(rv["ELBO_iters"]) == len(g["ELBO_iters"])
            np.testing.assert_allclose(rv["ELBO_iters"], g["ELBO_iters"], rtol=RTOL)
            np.testing.assert_allclose(rv["ELBO_inits"], g["ELBO_inits"], rtol=RTOL)
            np.testing.assert_allclose(rv["ID_prob"], g["ID_prob"], rtol=RTOL, atol=1e-290)
            assert rv["ELBO_iters"][-1] == pytest.approx(-190779.74335041404, rel=RTOL)
            assert np.array_equal(rv["ID_prob"].argmax(1), g["ID_prob"].argmax(1))


def test_bench_launches_its_own_ranks(va):
    """``python bench.py --gpus 2`` with no launcher and no WORLD_SIZE in the environment (how the
    driver starts ``--gpus 1``): the command spawns its two ranks (vireo_amd/launch.py), rank 0
    prints the ONE JSON line, restart r runs on rank r.  Both ranks on device 0 over the
    host-socket communicator (RCCL refuses two ranks on one device)."""
    import json
    env = dict(os.environ, VIREO_DEVICE="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "VIREO_COMM", "VIREO_FORCE_RCCL"):
        env.pop(k, None)
    lines = {}
    for n in (2, 1):
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--config", "c2",
                            "--steps", "20", "--warmup", "5", "--only-headline", "--comm", "tcp"],
                           env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
        assert p.returncode == 0, p.stderr[-2000:]
        out = [ln for ln in p.stdout.splitlines() if ln.strip()]
        assert len(out) == 1, p.stdout[-2000:]
        lines[n] = json.loads(out[0])
    two, one = lines[2], lines[1]
    # ranks that share a device over host sockets rehearse the plumbing: the line says so
    assert two["n_gpus"] == 2 and two["scaling"] == "plumbing-only" and two["value"] > 0
    assert one["scaling"] == "weak" and one["comm"]["backend"] == "local" and one["comm"]["world"] == 1
    c = two["comm"]
    assert c["backend"] == "tcp" and c["world"] == 2 and c["distinct_gpus"] == 1
    assert [r["rank"] for r in c["ranks"]] == [0, 1] and {r["device"] for r in c["ranks"]} == {0}
    assert c["ranks"][0]["pci_bus_id"] == c["ranks"][1]["pci_bus_id"] and c["ranks"][0]["pid"] != c["ranks"][1]["pid"]
    assert c["allgather_us"]["median"] > 0 and c["winner_broadcast"]["host_staged_us"]["min"] > 0
    assert "device_to_device_us" not in c["winner_broadcast"]          # (no device path over sockets)
    assert len(two["config"]["restart_protocol_elbos"]) == 2
    # rank 0 iterates restart 0 (= the world-1 run's), rank 1 the second constructor's draws
    assert two["config"]["restart_protocol_elbos"][0] == one["config"]["restart_protocol_elbos"][0]
    assert two["config"]["restart_protocol_elbos"][1] != two["config"]["restart_protocol_elbos"][0]
    assert np.all(np.isfinite(two["config"]["restart_elbos"]))


def test_bench_line_names_its_communicator_rccl_world1(va):
    """the JSON line says which communicator produced it.  VIREO_FORCE_RCCL=1 takes
    the RCCL path at world 1 on this one-GPU box: backend rccl, RCCL's version, the rank's device and
    PCI bus id as all-gathered over RCCL, the unique-id / ncclCommInitRank times, a timed all-gather,
    and the winner's state broadcast device to device beside the host-staged route."""
    import json
    import re
    env = dict(os.environ, VIREO_FORCE_RCCL="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "VIREO_COMM", "VIREO_DEVICE"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "c2", "--steps", "20",
                        "--warmup", "5", "--only-headline"], env=env, capture_output=True, text=True,
                       timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    out = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(out) == 1, p.stdout[-2000:]
    line = json.loads(out[0])
    c = line["comm"]
    assert c["backend"] == "rccl" and c["world"] == 1 and line["scaling"] == "weak"
    assert re.fullmatch(r"\d+\.\d+\.\d+", c["rccl_version"]) and c["rccl_version_code"] >= 20000
    assert len(c["ranks"]) == 1 and c["ranks"][0]["rank"] == 0 and c["ranks"][0]["device"] == 0
    assert re.fullmatch(r"[0-9a-f]{4}:[0-9a-f]{2}:[0-9a-f]{2}\.[0-9a-f]", c["ranks"][0]["pci_bus_id"])
    assert c["distinct_gpus"] == 1 and c["comm_init_rank_ms"] > 0 and c["unique_id_exchange_ms"] >= 0
    assert c["allgather_us"]["n_init"] == 32 and c["allgather_us"]["median"] > 0
    wb = c["winner_broadcast"]
    assert wb["device_to_device_us"]["min"] > 0 and wb["host_staged_us"]["min"] > 0 and wb["state_bytes"] > 0
