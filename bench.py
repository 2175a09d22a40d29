#!/usr/bin/env python
"""bench.py -- EM iterations/sec of the Vireo VB hot path on MI355X (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config c3|mid|c2] [--no-cpu]

A step is ONE full coordinate-ascent iteration (theta update, GT update, ID update, ELBO:
vireoSNP/utils/vireo_model.py:257-264) over the synthetic AD/DP of SURVEY.md 8(d), inputs
and state already resident in HBM.  Default workload = BASELINE.json configs[2]
(N=100k variants x M=50k cells, K=16, ~2 % nnz), the configuration the metric is quoted on.
W warm-up iterations run without the theta update (the protocol's delay_fit_theta=3), the K
timed ones with it.  N > 1 (launched by torch.distributed.run, one rank per GPU): every
rank holds the problem and iterates its own restart (vireo_wrap's restart shard, weak
scaling); the per-restart ELBOs are all-gathered over RCCL.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md


def algorithmic_bytes(N, M, K, T, nnz):
    """bytes one EM iteration must move if every operand is touched once.
    Split per kernel class."""
    variant = 12 * nnz + 4 * (N + 1) + 8 * M * K + 16 * N * K       # stream + ID read + S write
    cell = 12 * nnz + 4 * (M + 1) + 16 * N * K + 8 * M * K          # stream + W read + LID write
    dense = 8 * (2 * N * K * T + 2 * N * K + 2 * N * K + 2 * M * K)  # GT r/w, S read, W write,
    return dict(variant=variant, cell=cell, dense=dense,            # LID read + ID write
                total=variant + cell + dense)


def cpu_baseline_leg(w, K, seed):
    """ONE full-size iteration of the oracle (the reference's SciPy op sequence, 1 core)."""
    from oracle import vireo_oracle as O
    from vireo_amd.synth import as_scipy
    AD, DP = as_scipy(w)
    N, M = w["shape"]
    np.random.seed(seed)
    st = O.vireo_new(M, N, K)
    t0 = time.perf_counter()
    O.vireo_theta_step(st, AD, DP)
    O.vireo_gt_step(st, AD, DP)
    L = O.vireo_id_step(st, AD, DP)
    elbo = O.vireo_elbo(st, L)
    dt = time.perf_counter() - t0
    return dt, elbo, st


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="c3")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with "
                         "python -m torch.distributed.run --nproc-per-node N)" % (args.gpus, world))

    import __graft_entry__ as entry
    entry.build()
    from vireo_amd import _lib
    from vireo_amd.counts import DeviceCounts
    from vireo_amd.engine import DeviceModel
    from vireo_amd.vireo_model import Vireo
    from vireo_amd import synth, dist as vdist
    _lib.require_gpu()

    comm = vdist.LocalComm()
    # One process per GPU (torch.distributed.run only LAUNCHES the ranks and sets RANK /
    # WORLD_SIZE / MASTER_*); the ranks talk over RCCL through libvireo_hip.so, and the RCCL
    # unique id travels over a plain socket -- torch is never imported here: its bundled HIP
    # runtime and librccl would collide with the library's.  VIREO_BENCH_FORCE_RCCL=1 takes
    # this path at world 1 too, so a 1-GPU box can exercise it.
    if world > 1 or os.environ.get("VIREO_BENCH_FORCE_RCCL") == "1":
        comm = vdist.RcclComm(rank, world, local, vdist.socket_exchange(rank, world))

    N, M, K, dens = synth.CONFIGS[args.config]
    T = 3
    t_gen = time.perf_counter()
    w = synth.donor_workload(N, M, K, dens, seed=0)
    t_gen = time.perf_counter() - t_gen
    nnz = int(w["rowidx"].size)
    t_up = time.perf_counter()
    counts = DeviceCounts.from_merged(w["shape"], w["colptr"], w["rowidx"], w["ad"], w["dp"],
                                      device=local)
    t_up = time.perf_counter() - t_up

    # model init of the timing protocol: one np.random.seed, then sequential constructor
    # draws (vireo_wrap.py:53-71); rank r iterates restart r.
    np.random.seed(1)
    for _ in range(rank + 1):
        host = Vireo(n_var=N, n_cell=M, n_donor=K)
    dm = DeviceModel(counts, _lib.KIND_VIREO, K, n_gt=T)
    dm.set_state(host.ID_prob, host.GT_prob, host.beta_mu, host.beta_sum)
    dm.set_prior(host.ID_prior, host.GT_prior, host.theta_s1_prior, host.theta_s2_prior)

    # parity probe (rank 0): the very first iteration against the oracle, same init
    parity = None
    cpu = None
    if rank == 0:
        first, _ = dm.run_iters(1, theta_from_iter=0)
        ID1 = dm.get_state(want_GT=False)[0]
        dm.set_state(host.ID_prob, host.GT_prob, host.beta_mu, host.beta_sum)
        if not args.no_cpu:
            dt, elbo_cpu, st = cpu_baseline_leg(w, K, seed=1)
            # after ONE iteration from a random start the posteriors are still uniform to
            # ~3e-7, so a few cells have a top-2 gap below fp64 summation-order noise; they
            # are counted separately
            srt = np.sort(st.ID_prob, axis=1)
            near_tie = (srt[:, -1] - srt[:, -2]) <= 1e-9 * srt[:, -1]
            differ = ID1.argmax(1) != st.ID_prob.argmax(1)
            parity = dict(elbo_gpu=float(first[0]), elbo_cpu=float(elbo_cpu),
                          elbo_rel_err=float(abs(first[0] - elbo_cpu) / abs(elbo_cpu)),
                          id_prob_max_rel_err=float(np.max(
                              np.abs(ID1 - st.ID_prob) / np.maximum(st.ID_prob, 1e-300))),
                          assignment_mismatches=int(differ.sum()),
                          near_tie_cells=int(near_tie.sum()),
                          assignments_identical_outside_near_ties=bool(
                              not np.any(differ & ~near_tie)))
            cpu = dict(value=1.0 / dt, unit="EM iterations/s", cores=1, kind="port",
                       sample="1 full-size EM iteration (theta+GT+ID+ELBO) of the NumPy/SciPy "
                              "oracle (the reference's 13 SpMM + 2 sparse-subtract op sequence, "
                              "single-threaded like scipy.sparse) on the same %s inputs; %.1f s; "
                              "box has %d cores" % (args.config, dt, os.cpu_count()))
            del st
    del w

    if args.warmup > 0:
        dm.run_iters(args.warmup, theta_from_iter=10 ** 9)
    comm.barrier()
    t0 = time.perf_counter()
    trace, ms_dev = dm.run_iters(args.steps, theta_from_iter=0)   # syncs the stream
    t1 = time.perf_counter()
    comm.barrier()
    wall = t1 - t0
    walls = comm.allgather(np.array([wall]))
    wall_max = float(np.max(walls))
    last_elbos = comm.allgather(np.array([trace[-1]]))            # the restart-shard exchange

    # roofline leg: the same K iterations again with every pass bracketed by HIP events on
    # the library's stream
    dm.profile(True)
    dm.run_iters(args.steps, theta_from_iter=0)
    ms, n = dm.profile_read()
    dm.profile(False)

    if rank == 0:
        kinfo = dm.info()
        B = algorithmic_bytes(N, M, K, T, nnz)
        avg_v = ms[_lib.KERN_VARIANT_PASS] / max(n[_lib.KERN_VARIANT_PASS], 1)
        avg_c = ms[_lib.KERN_CELL_PASS] / max(n[_lib.KERN_CELL_PASS], 1)
        avg_d = ms[_lib.KERN_DENSE] / max(args.steps, 1)
        dom = "cell" if avg_c >= avg_v else "variant"
        avg = max(avg_c, avg_v)
        ach = B[dom] / (avg * 1e-3) / 1e9
        info = _lib.device_info(local)
        # HBM traffic of the dominant kernel: PMC counters cannot be read from inside this
        # process; the figure comes from the committed rocprofv3 --pmc run of this same
        # command (profiles/traffic_<config>.json) and is null when none matches.
        traffic, traffic_src = None, None
        kname = ("vrx_spmm_lds<4,%d>" % (dom == "cell")) if kinfo["lds_" + dom] else None
        tpath = os.path.join(ROOT, "profiles", "traffic_%s.json" % args.config)
        if kname and os.path.exists(tpath):
            rec = json.load(open(tpath)).get("kernels", {}).get(kname)
            if rec:
                traffic = rec["traffic_bytes"]
                traffic_src = "profiles/traffic_%s.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE)" % args.config
        out = {
            "metric": "EM iterations/sec", "value": world * args.steps / wall_max,
            "unit": "EM iterations/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": wall_max / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s: synthetic sparse AD/DP N=%d variants x M=%d cells, K=%d "
                                   "donors, nnz=%d (SURVEY.md 8d generator, seed 0); one restart "
                                   "per GPU" % (args.config, N, M, K, nnz),
                       "N": N, "M": M, "K": K, "nnz": nnz, "device": info["name"],
                       "restart_elbos": [float(x) for x in np.ravel(last_elbos)],
                       "best_restart": int(np.argmax(last_elbos)),
                       "host_setup_s": {"generate": round(t_gen, 1), "upload+transpose": round(t_up, 1)}},
            "roofline": {"bound": "hbm",
                         "kernel": "%s (%s pass)" % (
                             "vrx_spmm_lds<4,%d>" % (dom == "cell") if kinfo["lds_" + dom]
                             else "vrx_spmm<..,%d,fmt%d>" % (dom == "cell", kinfo["fmt_" + dom]), dom),
                         "kernel_info": kinfo,
                         "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": B[dom],
                         "avg_launch_ms": avg,
                         "per_iteration_ms": {"variant_pass": avg_v, "cell_pass": avg_c,
                                              "dense_kernels": avg_d},
                         "whole_iteration": {"algorithmic_bytes": B["total"],
                                             "achieved_GBs": B["total"] / (wall_max / args.steps) / 1e9,
                                             "frac": B["total"] / (wall_max / args.steps) / 1e9 / HBM_PEAK_GBS}},
            "cpu_baseline": cpu,
            "parity": parity,
        }
        if cpu:
            out["speedup_vs_cpu_1core"] = out["value"] / cpu["value"]
        print(json.dumps(out))
    if isinstance(comm, vdist.RcclComm):
        comm.barrier()
        comm.close()


if __name__ == "__main__":
    main()
