#!/usr/bin/env python
"""bench.py -- EM iterations/sec of the Vireo VB hot path on MI355X (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K=200] [--warmup W=50] [--config c3|mid|c2] [--no-cpu] [--no-c4]
                  [--comm rccl|tcp]

A step is ONE full coordinate-ascent iteration (theta update, GT update, ID update, ELBO:
vireoSNP/utils/vireo_model.py:257-264) over the synthetic AD/DP of SURVEY.md 8(d), inputs
and state already resident in HBM.  Default workload = BASELINE.json configs[2]
(N=100k variants x M=50k cells, K=16, ~2 % nnz), the configuration the metric is quoted on.
The W warm-up iterations are the protocol's start (delay_fit_theta=3: the first three without the
theta update, so every kernel of the timed iterations has been launched once), the K timed ones
all run with it.  The GPU legs this run executes anyway (c2, c5's GPU half, the c4 restart
search, and last -- directly before the warm-up, nothing but Python call overhead in between --
every rank's own whole-protocol fit of its restart, at N = 1 also the perturbed copy of the
parity check) come BEFORE the timed region and are listed in `preceded_by`: the chip used to
idle through ~20 s of host-side input generation right before a 16-ms timed window (driver flags
--steps 20 --warmup 5), which then ran 10-15 % slower than its own repeats (a 5-ms pause is
enough: scratch/gap_probe.py); the CPU-oracle legs and the heavy-tailed c3_skew leg run after it.
`ms_per_step_repeats` shows the spread.
N > 1 (launched by torch.distributed.run, one rank per GPU): every
rank holds the problem and iterates its own restart (vireo_wrap's restart shard, weak
scaling); the per-restart ELBOs are all-gathered over RCCL.  Rank 0 prints ONE JSON line.
The line says which communicator produced it (`comm`: backend rccl | tcp | local, world, RCCL
version, every rank's device and PCI bus id as all-gathered through that communicator, the timed
exchanges).  `--comm tcp` (host sockets, ranks sharing one device: a plumbing rehearsal on a 1-GPU
box) must be asked for on THIS command line: a VIREO_COMM=tcp merely inherited from the environment
is refused for N > 1, and a line it produces carries "scaling": "plumbing-only", never "weak".

Besides the headline value the line carries
  roofline      the dominant sparse pass against the HBM roofline (HIP events on the library's
                stream; PMC traffic from profiles/traffic_<config>.json when it was collected
                on the same kernel sources),
  cpu_baseline  the CPU oracle (the reference's NumPy/SciPy op sequence, 1 core) running the
                WHOLE timing protocol fit(min_iter=5, max_iter=20, delay_fit_theta=3) on the
                same inputs,
  parity        that run against the same fit on the GPU: iteration count, whole ELBO trace,
                final assignments (the metric is "EM iterations/sec + ELBO-match"),
  c4            BASELINE.json configs[3]: vireo_wrap(n_init=32) on the same data with the
                restarts sharded over the N ranks (strong scaling: the 32 restarts are the
                fixed job), restart-iterations/s, wall time and the per-phase split.
  c2, c5        BASELINE.json configs[1] (launch-bound small problem) and configs[4]
                (BinomMixtureVB clone mode) on the driver's clock: us / ms per iteration,
                c5 with its roofline fraction and the first iterations against the oracle.
  c3_skew       c3's shape with heavy-tailed (log-normal) coverage / depth -- real-data-shaped
                input -- on the same kernels: ms per iteration, padding, row pieces, imbalance.
  comm          the communicator's self-description (vireo_amd/dist.py comm_record) and the
                winner's-state broadcast timed over it: device to device (vrx_comm_bcast_model)
                against the host-staged route.
  c3_flags      the flag paths at headline size (ASE_mode, fixed GT from a donor prior, learned GT
                with a non-uniform prior, fix_beta_sum): ms per iteration and their own
                algorithmic bytes.
  ms_per_step_repeats   the timed K iterations repeated four more times (min / median / max).
"""
import argparse
import contextlib
import hashlib
import io
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md
PROTOCOL = dict(min_iter=5, max_iter=20, delay_fit_theta=3)


def algorithmic_bytes(N, M, K, T, nnz):
    """bytes one EM iteration must move if every operand is touched once.
    Split per kernel class."""
    variant = 12 * nnz + 4 * (N + 1) + 8 * M * K + 16 * N * K       # stream + ID read + S write
    cell = 12 * nnz + 4 * (M + 1) + 16 * N * K + 8 * M * K          # stream + W read + LID write
    dense = 8 * (2 * N * K * T + 2 * N * K + 2 * N * K + 2 * M * K)  # GT r/w, S read, W write,
    return dict(variant=variant, cell=cell, dense=dense,            # LID read + ID write
                total=variant + cell + dense)


def kernel_source_hash():
    """what a PMC traffic record must have been collected on to be quoted"""
    h = hashlib.sha256()
    for f in ("vrx_kernels.h", "vrx_engine.hip", "vrx_common.h"):
        h.update(open(os.path.join(ROOT, "vireo_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def stream_choice():
    """the headline problem's build: balanced slabs unless VIREO_BALANCE=0 (a traffic record names its own)"""
    return "balanced slabs" if os.environ.get("VIREO_BALANCE", "1") != "0" else "default build"


def cpu_protocol_leg(w, K, seed):
    """the whole timing protocol on the oracle (1 core), from the same initial state"""
    from oracle import vireo_oracle as O
    from vireo_amd.synth import as_scipy
    AD, DP = as_scipy(w)
    N, M = w["shape"]
    np.random.seed(seed)
    st = O.vireo_new(M, N, K)
    t0 = time.perf_counter()
    trace, it = O.vireo_fit_vb(st, AD, DP, **PROTOCOL)     # (the trace without the constant)
    dt = time.perf_counter() - t0
    return dt, st, trace, it


def c4_leg(counts, K, comm, n_init=32):
    """BASELINE.json configs[3]: the n_init=32 restart search of vireo_wrap, sharded."""
    from vireo_amd import restarts
    W = sys.modules["vireo_amd.vireo_wrap"]
    restarts.PHASES = {}
    comm.barrier()
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        rv = W.vireo_wrap(counts, None, n_donor=K, n_init=n_init, max_iter_init=20,
                          random_seed=1, check_doublet=False, comm=comm)
    wall = time.perf_counter() - t0
    comm.barrier()
    phases, restarts.PHASES = restarts.PHASES, None
    stats = dict(W.LAST_SEARCH)
    # per-rank numbers -> rank-major arrays on every rank
    keys = ["search_wall", "draw", "skip", "stage", "upload+normalise", "fit", "snapshot", "gather",
            "final_fit", "download", "broadcast", "template", "device_models"]
    mine = np.array([wall, stats["restart_iterations"], stats["final_iterations"]] +
                    [phases.get(k, 0.0) for k in keys])
    allr = np.asarray(comm.allgather(mine)).reshape(comm.world, -1)
    wall_max = float(allr[:, 0].max())
    restart_its = int(allr[:, 1].sum())
    final_its = int(allr[:, 2].sum())
    # the restart-shard phase ends when the slowest rank has fitted its share (wall clock of
    # the restart loop: the random draws and their staged upload ("stage") run one restart ahead
    # on a helper thread, so the draw / skip / stage and the fit phases below overlap)
    shard = allr[:, 3]
    return dict(
        workload="c4: vireo_wrap(n_init=%d, max_iter_init=20, random_seed=1, no doublets) on the "
                 "c3 data, restart i on rank i %% %d" % (n_init, comm.world),
        n_init=n_init, wall_s=wall_max, restart_iterations=restart_its,
        final_fit_iterations=final_its,
        restart_iterations_per_s=restart_its / float(shard.max()),
        whole_job_iterations_per_s=(restart_its + final_its) / wall_max,
        restart_shard_phase_s=float(shard.max()),
        phases_s_max_over_ranks={k: float(allr[:, 3 + i].max()) for i, k in enumerate(keys)},
        LB_list_head=[float(x) for x in rv["LB_list"][:4]], best_restart=stats["best"]), rv


def doublet_leg(counts, K, rv, repeats=3):
    """The DEFAULT post-step of vireo_wrap (check_doublet=True, vireo_wrap.py:151-156 ->
    vireo_doublet.py:11-82) on the winner the c4 leg returned: K + K(K-1)/2 columns through the
    cell pass in sweeps of 16, the pair genotype table (653 MB in the reference at c3) formed
    inside the W kernel, then the reference's side effects (ID_prob <- singlet block,
    update_GT_prob).  Seconds per call, host transfers of GT_prob (38 MB up) and the two result
    tables (54 MB down) included -- what a `vireo_wrap` caller pays on top of the c4 job."""
    from vireo_amd.vireo_doublet import predict_doublet
    from vireo_amd.vireo_model import Vireo
    N, M = counts.shape
    m = Vireo(n_var=N, n_cell=M, n_donor=K, ID_prob_init=rv["ID_prob"], GT_prob_init=rv["GT_prob"],
              beta_mu_init=rv["theta_mean"], beta_sum_init=rv["theta_sum"])
    secs = []
    for _ in range(repeats):
        m.ID_prob, m.GT_prob = rv["ID_prob"], rv["GT_prob"]      # (the step overwrites both)
        t0 = time.perf_counter()
        dbl, sing, llr = predict_doublet(m, counts, None)
        secs.append(time.perf_counter() - t0)
    cols = K + K * (K - 1) // 2
    return dict(workload="predict_doublet(update_GT=True, update_ID=True) on the c4 winner: %d cells x "
                         "%d columns (%d donors + %d pairs), 6 genotype classes per pair" % (M, cols, K, cols - K),
                doublet_s=min(secs), doublet_s_runs=[round(x, 4) for x in secs], columns=cols,
                cell_pass_sweeps=-(-cols // 16),
                cells_called_doublet=int(np.sum(dbl.sum(1) >= 0.9)),
                max_doublet_logLikRatio=float(np.max(llr)))


def c2_leg(device, steps=200):
    """BASELINE.json configs[1] (N=10k x M=5k, K=4): a launch-bound problem.  One restart per
    model against 16 restarts in one model (vrx_model_cfg.n_batch; vireo_wrap packs its
    restarts like this when n_donor leaves columns of the 16-wide passes idle)."""
    from vireo_amd import _lib, synth
    from vireo_amd.counts import DeviceCounts
    from vireo_amd.engine import DeviceBatch
    N, M, K, dens = synth.CONFIGS["c2"]
    w = synth.donor_workload(N, M, K, dens, seed=0)
    counts = DeviceCounts.from_merged(w["shape"], w["colptr"], w["rowidx"], w["ad"], w["dp"],
                                      device=device)
    rng = np.random.default_rng(0)
    mu, sm = np.linspace(0.01, 0.99, 3)[None, :], np.full((1, 3), 50.0)
    out = dict(workload="c2: N=%d x M=%d, K=%d, nnz=%d; %d iterations with theta" % (
        N, M, K, int(w["rowidx"].size), steps), us_per_restart_iteration={})
    for R in (1, 4, 16):
        db = DeviceBatch(counts, _lib.KIND_VIREO, K, R)
        # (the reference's default priors: uniform ID / GT, theta Beta(0.3, 29.7), (3, 3), (29.7, 0.3))
        db.set_prior(np.full((1, K), 1.0 / K), np.full((1, K, 3), 1.0 / 3),
                     np.array([[0.3, 3.0, 29.7]]), np.array([[29.7, 3.0, 0.3]]))
        for r in range(R):
            db.set_restart(r, rng.random((M, K)), rng.random((N, K, 3)), mu, sm, raw=True)
        db.run_iters(10)
        tr, ms = db.run_iters(steps)
        if not np.all(np.isfinite(tr)):
            raise RuntimeError("c2 leg: a non-finite ELBO in the timed iterations")
        out["us_per_restart_iteration"]["n_batch=%d" % R] = round(ms / steps / R * 1e3, 2)
        db.close()
    return out


def c3_skew_leg(device, K, uniform_ms, steps=100):
    """c3's shape with heavy-tailed coverage / depth (synth.C3_SKEW, log-normal sigma 1.0 / 0.7:
    what real cellSNP matrices look like, io_utils.py:42-59) on this build's kernels: ms per
    iteration in steady state (after 50 warm-up iterations) beside the uniform generator's, the
    stream padding, the pieces long rows are cut into, the wave imbalance."""
    from vireo_amd import _lib, synth
    from vireo_amd.counts import DeviceCounts
    from vireo_amd.engine import DeviceModel
    from vireo_amd.vireo_model import Vireo
    N, M, _, dens = synth.CONFIGS["c3"]
    w = synth.donor_workload(N, M, K, dens, seed=0, skew=synth.C3_SKEW)
    nnz = int(w["rowidx"].size)
    rows = np.bincount(w["rowidx"], minlength=N)
    cols = np.diff(w["colptr"])
    max_count = int(w["dp"].max())
    np.random.seed(1)
    host = Vireo(n_var=N, n_cell=M, n_donor=K)

    def measure(balance):
        t_b = time.perf_counter()
        counts = DeviceCounts.from_merged(w["shape"], w["colptr"], w["rowidx"], w["ad"], w["dp"], device=device,
                                          balance=balance)
        build_s = time.perf_counter() - t_b
        binfo = counts.build_info()
        dm = DeviceModel(counts, _lib.KIND_VIREO, K, n_gt=3)
        dm.set_state(host.ID_prob, host.GT_prob, host.beta_mu, host.beta_sum)
        dm.set_prior(host.ID_prior, host.GT_prior, host.theta_s1_prior, host.theta_s2_prior)
        dm.run_iters(50, theta_from_iter=PROTOCOL["delay_fit_theta"])
        t0 = time.perf_counter()
        tr, _ = dm.run_iters(steps, theta_from_iter=0)
        ms = (time.perf_counter() - t0) / steps * 1e3
        if not np.all(np.isfinite(tr)):
            raise RuntimeError("c3_skew leg: a non-finite ELBO in the timed iterations")
        dm.profile(True)
        dm.run_iters(steps, theta_from_iter=0)
        pm_, pn_ = dm.profile_read()
        info_ = dm.info()
        dm.close()
        counts.close()
        return ms, pm_, pn_, info_, binfo, build_s, np.asarray(tr)

    # the default build first (rows cut into pieces, contiguous slabs), then the headline's stream choice
    d_ms, d_pm, d_pn, d_info, _, d_build_s, d_tr = measure(False)
    balanced = os.environ.get("VIREO_BALANCE", "1") != "0"
    if balanced:
        ms_it, pm, pn, info, binfo, build_s, tr = measure(True)
    else:
        ms_it, pm, pn, info, binfo, build_s, tr = d_ms, d_pm, d_pn, d_info, None, d_build_s, d_tr
    del w
    default_build = dict(ms_per_iteration=d_ms, pad_variant=d_info["pad_variant"], pad_cell=d_info["pad_cell"],
                         passes_ms={"variant_pass": d_pm[0] / max(d_pn[0], 1), "cell_pass": d_pm[1] / max(d_pn[1], 1),
                                    "dense_kernels": d_pm[2] / steps},
                         build_s=round(d_build_s, 3),
                         elbo_rel_diff_to_balanced=float(np.max(np.abs(tr - d_tr) / np.abs(d_tr))))
    B = algorithmic_bytes(N, M, K, 3, nnz)
    return dict(workload="c3 shape, log-normal coverage / depth (sigma %.1f / %.1f): nnz=%d, entries per "
                         "variant %d..%d (median %d), per cell %d..%d (median %d), largest count %d; "
                         "%d iterations after 50" % (synth.C3_SKEW + (nnz, rows.min(), rows.max(), np.median(rows),
                                                     cols.min(), cols.max(), np.median(cols), max_count, steps)),
                ms_per_iteration=ms_it, iterations_per_s=1e3 / ms_it,
                ms_per_iteration_uniform_c3=uniform_ms,
                ratio_to_uniform_per_nonzero=(ms_it / nnz) / (uniform_ms[0] / uniform_ms[1]),
                passes_ms={"variant_pass": pm[0] / max(pn[0], 1), "cell_pass": pm[1] / max(pn[1], 1),
                           "dense_kernels": pm[2] / steps},
                whole_iteration_roofline_frac=B["total"] / (ms_it * 1e-3) / 1e9 / HBM_PEAK_GBS,
                pad_variant=info["pad_variant"], pad_cell=info["pad_cell"],
                extra_pieces_variant=info["extra_pieces_variant"], extra_pieces_cell=info["extra_pieces_cell"],
                imbalance_variant=info["imbalance_variant"], imbalance_cell=info["imbalance_cell"],
                lds_passes=bool(info["lds_variant"] and info["lds_cell"]), kernel_info=info,
                stream="balanced slabs, the pieces of long rows as the unit" if balanced and binfo and binfo["balanced_cell"]
                else "default build",
                build_info=binfo, build_s=round(build_s, 3), default_build=default_build)


FLAG_PATHS = ("ase", "fixedGT", "priorGT", "fixsum")


def flag_model(flag, N, M, K, GT_true):
    """the host ``Vireo`` of one flag path at the timing protocol's start (np.random.seed(1))"""
    from vireo_amd import synth
    from vireo_amd.vireo_model import Vireo
    kw = dict(ase=dict(ASE_mode=True), fixedGT=dict(learn_GT=False), priorGT=dict(),
              fixsum=dict(fix_beta_sum=True))[flag]
    prior = None
    if flag == "fixedGT":           # the donors' genotypes are known: `vireo -d donors.vcf` (mode 2)
        prior = synth.planted_gt_prior(GT_true, 1.0)
    elif flag == "priorGT":         # known for 90 % of the calls and learned on (mode 4, --forceLearnGT)
        rng = np.random.default_rng(5)
        noisy = np.where(rng.random(GT_true.shape) < 0.9, GT_true, rng.integers(0, 3, GT_true.shape))
        prior = synth.planted_gt_prior(noisy, 0.8)
    np.random.seed(1)
    m = Vireo(n_var=N, n_cell=M, n_donor=K, **kw, **(dict(GT_prob_init=prior.copy()) if prior is not None else {}))
    if prior is not None:
        m.set_prior(GT_prior=prior)
    return m


def flag_bytes(flag, N, M, K, T, nnz):
    """SURVEY.md 8(d) for the flag paths: theta as N x T in ASE mode (posterior written, the three
    digamma tables written and read: 8 N T doubles); a fixed GT is read once and never written;
    a non-uniform GT_prior adds its N K T table."""
    B = algorithmic_bytes(N, M, K, T, nnz)
    extra = {"ase": 8 * 8 * N * T, "fixedGT": -8 * N * K * T, "priorGT": 8 * N * K * T, "fixsum": 0}[flag]
    return B["total"] + extra


def c3_flags_leg(counts, GT_true, K, uniform_ms, steps=50):
    """The flag paths at headline size -- they take kernels the headline never runs (vrx_theta_ase
    and per-variant theta tables; W from a fixed genotype table; the GT_prior table inside
    vrx_gt_update and KL_GT; the fixed-sum theta update): ms per iteration in steady state (20
    protocol iterations first, theta from the third on) beside the default path's, the per-pass
    split, and the whole-iteration roofline on the path's own algorithmic bytes."""
    N, M = counts.shape
    out = {"workload": "c3 data, one model per flag path from np.random.seed(1); %d iterations after 20; "
                       "fixedGT: the planted genotypes as GT_prob_init = GT_prior, learn_GT=False (mode 2); "
                       "priorGT: a prior right for 90 %% of the calls, sharpness 0.8, learn_GT=True (mode 4)" % steps,
           "default_ms_per_iteration": uniform_ms}
    for flag in FLAG_PATHS:
        host = flag_model(flag, N, M, K, GT_true)
        dm, _ = host._device_model(counts, None)
        del host
        dm.run_iters(20, theta_from_iter=PROTOCOL["delay_fit_theta"])
        t0 = time.perf_counter()
        tr, _ = dm.run_iters(steps, theta_from_iter=0)
        ms_it = (time.perf_counter() - t0) / steps * 1e3
        if not np.all(np.isfinite(tr)):
            raise RuntimeError("c3_flags leg (%s): a non-finite ELBO in the timed iterations" % flag)
        dm.profile(True)
        dm.run_iters(steps, theta_from_iter=0)
        pm, pn = dm.profile_read()
        dm.close()
        nb = flag_bytes(flag, N, M, K, 3, counts.nnz)
        out[flag] = dict(ms_per_iteration=ms_it, iterations_per_s=1e3 / ms_it,
                         ratio_to_default=ms_it / uniform_ms,
                         passes_ms={"variant_pass": pm[0] / max(pn[0], 1), "cell_pass": pm[1] / max(pn[1], 1),
                                    "dense_kernels": pm[2] / steps},
                         algorithmic_bytes_per_iteration=nb,
                         whole_iteration_roofline_frac=nb / (ms_it * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         elbo_last=float(tr[-1]))
    return out


def e2e_leg(w, K, n_init=8):
    """The `vireo` COMMAND end to end at the headline size (vireoSNP/vireo.py:109-242): a cellSNP
    folder on disk (two MatrixMarket files of ~1e8 entries, a VCF of the variants, the barcodes)
    -> loaders -> device problem -> vireo_wrap(n_init restarts, doublets) -> donor_ids.tsv,
    summary.tsv, prob_*.tsv.gz, GT_donors.vireo.vcf.gz.  Wall seconds of the command and of its
    phases (wrappers around the command's own calls); the folder is written beforehand (not part
    of the command) and removed afterwards."""
    import shutil
    import tempfile
    from vireo_amd import synth
    from vireo_amd import vireo as cli
    root = tempfile.mkdtemp(prefix="vireo_e2e_")
    phases = {}

    def timed(mod, name, key):
        fn = getattr(mod, name)

        def wrapper(*a, **k):
            t0 = time.perf_counter()
            try:
                return fn(*a, **k)
            finally:
                phases[key] = phases.get(key, 0.0) + time.perf_counter() - t0
        setattr(mod, name, wrapper)
        return fn

    saved = []
    try:
        t0 = time.perf_counter()
        nbytes = synth.write_cellsnp_folder(w, root + "/cells")
        t_write = time.perf_counter() - t0
        for name, key in (("load_cells", "load (VCF + 2 MatrixMarket files -> CSC)"),
                          ("device_counts", "device problem (merge + upload + both tiled streams)"),
                          ("vireo_wrap", "vireo_wrap (restarts, final fit, doublets)"),
                          ("write_donor_id", "write donor_ids / summary / prob tables"),
                          ("write_VCF", "write GT_donors.vireo.vcf.gz")):
            saved.append((name, timed(cli, name, key)))
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()) as out:
            cli.main(["-c", root + "/cells", "-N", str(K), "-o", root + "/out", "--randSeed", "1",
                      "-M", str(n_init), "--noPlot"])
        wall = time.perf_counter() - t0
        files = sorted(os.listdir(root + "/out"))
        summary = open(root + "/out/summary.tsv").read().split("\n")[1:-1]
    finally:
        for name, fn in saved:
            setattr(cli, name, fn)
        shutil.rmtree(root, ignore_errors=True)
    N, M = w["shape"]
    return dict(workload="`vireo -c <cellSNP folder> -N %d -M %d --randSeed 1` on the headline data written to "
                         "disk: N=%d x M=%d, %.2f GB of input files" % (K, n_init, N, M, nbytes / 1e9),
                wall_s=wall, phases_s={k: round(v, 3) for k, v in phases.items()},
                other_s=round(wall - sum(phases.values()), 3), folder_written_in_s=round(t_write, 2),
                output_files=files, summary=summary, log_tail=out.getvalue().strip().split("\n")[-1])


def c5_gpu_leg(device, steps=50):
    """BASELINE.json configs[4]: BinomMixtureVB clone mode, N=200 variants x M=200k cells, K=8
    clones (bmm_model.py:178-201: one iteration = theta update, E[log lik], ID update, ELBO).
    Iterations/s with inputs and state resident and the HBM roofline on SURVEY.md 8(d)'s 0.89 GB
    per iteration.  (The GPU half; `c5_cpu_leg` checks its first iterations against the oracle
    after the timed region of the headline.)"""
    from vireo_amd import _lib, synth
    from vireo_amd.bmm_model import BinomMixtureVB
    from vireo_amd.counts import DeviceCounts
    from vireo_amd.engine import DeviceModel
    N, M, K = 200, 200000, 8
    AD, DP = synth.clone_workload(N, M, K, seed=0)
    counts = DeviceCounts(AD, DP, device=device)
    nnz = int(counts.nnz)
    np.random.seed(1)
    host = BinomMixtureVB(n_var=N, n_cell=M, n_donor=K)
    dm = DeviceModel(counts, _lib.KIND_BMM, K)
    host._push(dm)
    first, _ = dm.run_iters(3)                       # the first iterations: checked in c5_cpu_leg
    t0 = time.perf_counter()
    dm.run_iters(steps)
    wall = time.perf_counter() - t0
    dm.profile(True)
    dm.run_iters(steps)
    pm, pn = dm.profile_read()
    info = dm.info()
    dm.close()
    counts.close()
    # 2 x 12 B per entry + the dense operands once per producer / consumer
    bytes_it = 2 * 12 * nnz + 8 * (2 * M * K + 2 * N * K * 3)
    ms_it = wall / steps * 1e3
    out = dict(workload="c5: BinomMixtureVB N=%d x M=%d, K=%d, nnz=%d (SURVEY.md 8d generator, "
                        "seed 0); %d iterations" % (N, M, K, nnz, steps),
               ms_per_iteration=ms_it, iterations_per_s=1e3 / ms_it,
               passes_ms={"variant_pass": pm[0] / max(pn[0], 1), "cell_pass": pm[1] / max(pn[1], 1),
                          "dense_kernels": pm[2] / steps},
               algorithmic_bytes_per_iteration=bytes_it,
               roofline_frac=bytes_it / (ms_it * 1e-3) / 1e9 / HBM_PEAK_GBS,
               kernel_info=info)
    # VERDICT r5 (weak 7): the fraction above is on SURVEY.md 8(d)'s 12 B per entry and orientation; this
    # path stores ONE 4-B pair word per entry, so the bytes it really moves are far fewer -- say so
    tpath = os.path.join(ROOT, "profiles", "traffic_c5.json")
    if os.path.exists(tpath):
        doc = json.load(open(tpath))
        if doc.get("kernel_source_hash") == kernel_source_hash():
            moved = sum(k["traffic_bytes"] for k in doc["kernels"].values())
            out["passes_measured_traffic"] = dict(
                bytes_per_iteration=moved, source="profiles/traffic_c5.json (rocprofv3 --pmc, same kernel sources)",
                passes_GBs_on_measured_traffic=moved / ((out["passes_ms"]["variant_pass"] + out["passes_ms"]["cell_pass"]) * 1e-3) / 1e9,
                note="the roofline_frac above is on the 8(d) algorithmic bytes (12 B per entry and orientation); the pair-word "
                     "stream is 4 B per entry, so the passes move this much and are bound on-chip like c3's")
    return out, (AD, DP, first)


def c5_cpu_leg(out, data):
    """the oracle's first three clone-mode iterations (1 core) beside the GPU's"""
    from oracle import vireo_oracle as O
    AD, DP, first = data
    N, M = AD.shape
    K = 8
    np.random.seed(1)
    ref = O.bmm_new(M, N, K)
    ref_elbo = []
    t0 = time.perf_counter()
    for _ in range(3):
        O.bmm_theta_step(ref, AD, DP)
        L = O.bmm_cell_loglik(ref, AD, DP)
        O.bmm_id_step(ref, L)
        ref_elbo.append(O.bmm_elbo(ref, L))
    t_cpu = (time.perf_counter() - t0) / 3
    out["elbo_rel_err_first_iterations"] = [float("%.3g" % (abs(a - b) / abs(b)))
                                            for a, b in zip(first, ref_elbo)]
    out["cpu_oracle_s_per_iteration"] = t_cpu
    out["cpu_cores"] = 1
    return out


def comm_leg(comm, device, dm, host):
    """The communicator's self-description (dist.comm_record: backend, world, RCCL version, every
    rank's device / PCI bus id all-gathered THROUGH it, the unique-id exchange, a timed all-gather of
    32 ELBOs) and the winner's-state broadcast of vireo_wrap (vireo_wrap.py:90-94) timed both ways
    on the timed model's state: device to device (vrx_comm_bcast_model, RCCL only) and the
    host-staged route (four vrx_comm_bcast_f64 / socket broadcasts of host arrays)."""
    from vireo_amd import dist as vdist
    rec = vdist.comm_record(comm, device)
    if comm.backend == "local":
        rec["winner_broadcast"] = None
        return rec
    arrays = [host.ID_prob, host.GT_prob, np.ascontiguousarray(host.beta_mu), np.ascontiguousarray(host.beta_sum)]
    nbytes = int(sum(a.nbytes for a in arrays))
    out = dict(state_bytes=nbytes, root=0)
    if hasattr(comm, "bcast_model"):
        us = []
        for _ in range(3):
            comm.barrier()
            t0 = time.perf_counter()
            comm.bcast_model(dm, 0)
            us.append((time.perf_counter() - t0) * 1e6)
        out["device_to_device_us"] = dict(min=round(min(us), 1), runs=[round(x, 1) for x in us])
    us = []
    for _ in range(2):
        comm.barrier()
        t0 = time.perf_counter()
        for a in arrays:
            comm.bcast(a, 0)
        us.append((time.perf_counter() - t0) * 1e6)
    out["host_staged_us"] = dict(min=round(min(us), 1), runs=[round(x, 1) for x in us])
    rec["winner_broadcast"] = out
    return rec


def side_leg(name, fn, *a, **k):
    """a side leg must never cost the run its headline: an exception becomes {"error": ...}"""
    try:
        return fn(*a, **k)
    except BaseException as e:      # noqa: BLE001
        if isinstance(e, KeyboardInterrupt):
            raise
        sys.stderr.write("bench.py: the %s leg failed: %r\n" % (name, e))
        return {"error": "%s leg failed: %r" % (name, e)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--config", default="c3")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline / parity leg")
    ap.add_argument("--no-c4", action="store_true", help="skip the n_init=32 restart-shard leg")
    ap.add_argument("--no-side-legs", action="store_true",
                    help="skip the c2 / c5 / c3_skew / doublet legs (N = 1 only anyway)")
    ap.add_argument("--only-headline", action="store_true",
                    help="= --no-cpu --no-c4 --no-side-legs (A/B runs, profiler runs)")
    ap.add_argument("--comm", choices=("rccl", "tcp"), default=None,
                    help="tcp: host sockets instead of RCCL, for ranks that share ONE device "
                         "(a plumbing rehearsal; the line is marked plumbing-only)")
    args = ap.parse_args()
    if args.only_headline:
        args.no_cpu = args.no_c4 = args.no_side_legs = True
    # The communicator of a multi-GPU line is chosen on THIS command line.  VIREO_COMM=tcp left
    # behind in an environment would otherwise turn the driver's `bench.py --gpus 8` into a
    # host-socket run that looks like an RCCL one.
    inherited = os.environ.get("VIREO_COMM", "").lower()
    if args.comm is not None:
        os.environ["VIREO_COMM"] = args.comm        # (the ranks this process spawns inherit it)
    elif inherited == "tcp" and args.gpus > 1:
        sys.stderr.write("bench.py: VIREO_COMM=tcp is set in the environment but --comm tcp was not given: "
                         "refusing to produce a %d-GPU line over host sockets.  Unset VIREO_COMM for the "
                         "RCCL run, or pass --comm tcp for a one-device plumbing rehearsal.\n" % args.gpus)
        sys.exit(2)

    import __graft_entry__ as entry
    from vireo_amd import launch
    # `python bench.py --gpus N` with no launcher around it: this process becomes the launcher --
    # one copy of this command per GPU with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set the way
    # torch.distributed.run sets them (vireo_amd/launch.py; the reference takes its parallelism
    # as an argument too: nproc, vireo_wrap.py:74-91).  Rank 0's JSON line is this process's
    # stdout; the exit code is non-zero if any rank fails.
    if args.gpus > 1 and not launch.launched_externally():
        entry.build()                       # once, not N times at once
        launch.relaunch_self(args.gpus)

    # stdout carries exactly ONE JSON line: libraries that print on the C stdout (librccl's
    # version banner sits in the stdio buffer until exit) get stderr instead
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    from vireo_amd import dist as vdist
    rank, world, local = vdist.env_rank_world()
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    entry.build()
    from vireo_amd import _lib
    from vireo_amd.counts import DeviceCounts
    from vireo_amd.engine import DeviceModel
    from vireo_amd.vireo_model import Vireo
    from vireo_amd import synth
    _lib.require_gpu()

    # One process per GPU (a launcher -- torch.distributed.run or vireo_amd/launch.py -- only
    # STARTS the ranks and sets RANK / WORLD_SIZE / MASTER_*); the ranks talk over RCCL through
    # libvireo_hip.so, and the RCCL unique id travels over a plain socket -- torch is never
    # imported here: its bundled HIP runtime and librccl would collide with the library's.
    # VIREO_FORCE_RCCL=1 (or the older VIREO_BENCH_FORCE_RCCL=1) takes this path at world 1 too,
    # so a 1-GPU box can exercise it; VIREO_COMM=tcp replaces RCCL by host sockets for ranks that
    # share one device (RCCL refuses that): `VIREO_COMM=tcp VIREO_DEVICE=0 python bench.py --gpus 8`.
    comm = vdist.make_comm(rank, world, local,
                           force_rccl=os.environ.get("VIREO_BENCH_FORCE_RCCL") == "1")

    N, M, K, dens = synth.CONFIGS[args.config]
    T = 3
    t_gen = time.perf_counter()
    w = synth.donor_workload(N, M, K, dens, seed=0)
    t_gen = time.perf_counter() - t_gen
    nnz = int(w["rowidx"].size)
    # The headline runs on BALANCED SLABS (vrx_problem_create2, VRX_PROBLEM_BALANCED: which contracted rows
    # share a slab is chosen per row tile): a one-off effort at problem build, reported
    # in `host_setup_s` / `config.stream`, that `vireo_wrap` asks for by itself from VIREO_BALANCE_MIN_ITERS
    # expected iterations on.  The c4 leg below runs on the stream `vireo_wrap`'s own policy picks for its job
    # (n_init = 32: the default build), and `unbalanced_stream` times the headline's iterations on that one too.
    from vireo_amd.counts import balance_policy
    t_up = time.perf_counter()
    counts = DeviceCounts.from_merged(w["shape"], w["colptr"], w["rowidx"], w["ad"], w["dp"],
                                      device=local, balance=os.environ.get("VIREO_BALANCE", "1") != "0")
    t_up = time.perf_counter() - t_up
    binfo = counts.build_info()
    c4_iters = -(-32 // world) * 20 + 200
    # The default build of the same problem beside it: the c4 job runs on it when `vireo_wrap`'s policy would
    # (announced iterations below VIREO_BALANCE_MIN_ITERS), `unbalanced_stream` times the headline's iterations
    # on it, and a SECOND balanced build (the first one in a process also pays the library's one-time costs)
    # gives the wall-clock price of balancing: balanced build - default build, both warm.
    counts_default, t_up_default, t_up_again = None, None, None
    if (binfo["balanced_cell"] or binfo["balanced_variant"]) and args.config == "c3" \
            and not (args.no_c4 and args.no_side_legs):
        t_up_default = time.perf_counter()
        counts_default = DeviceCounts.from_merged(w["shape"], w["colptr"], w["rowidx"], w["ad"], w["dp"],
                                                  device=local, balance=False)
        t_up_default = time.perf_counter() - t_up_default
        t_up_again = time.perf_counter()
        again = DeviceCounts.from_merged(w["shape"], w["colptr"], w["rowidx"], w["ad"], w["dp"],
                                         device=local, balance=True)
        t_up_again = time.perf_counter() - t_up_again
        again.close()
    counts_job = counts if counts_default is None or balance_policy(None, c4_iters) else counts_default

    # ---- GPU legs that the run executes anyway, BEFORE the timed region ---------------------
    # (the timed K iterations used to open on a chip that had idled through ~20 s of
    # host-side input generation; the driver's --steps 20 --warmup 5 window is 16 ms, far less
    # than the clocks need to come back.  Nothing is added to the warm-up: `warmup` stays what
    # the flag says.  The legs are named in the JSON line, `preceded_by`.)
    preceded_by = []
    c4 = c2 = c5 = None
    c5_data = None
    parity_gpu = None
    solo = rank == 0 and world == 1
    if solo and not args.no_side_legs and args.config == "c3":
        c2 = side_leg("c2", c2_leg, local)
        preceded_by.append("c2 leg (3 x 210 iterations of the N=10k x M=5k problem)")
        got = side_leg("c5", c5_gpu_leg, local)
        c5, c5_data = got if isinstance(got, tuple) else (got, None)
        preceded_by.append("c5 leg, GPU half (103 clone-mode iterations)")
    # model init of the timing protocol: one np.random.seed, then sequential constructor
    # draws (vireo_wrap.py:53-71); rank r iterates restart r.
    np.random.seed(1)
    for _ in range(rank + 1):
        host = Vireo(n_var=N, n_cell=M, n_donor=K)
    dm = DeviceModel(counts, _lib.KIND_VIREO, K, n_gt=T)
    dm.set_state(host.ID_prob, host.GT_prob, host.beta_mu, host.beta_sum)
    dm.set_prior(host.ID_prior, host.GT_prior, host.theta_s1_prior, host.theta_s2_prior)
    # (the timed model is resident before the c4 leg: at N > 1, where no parity fits follow, only
    #  the tail of that leg -- the winner's download / broadcast -- separates its fits from the warm-up)
    if not args.no_c4 and args.config == "c3":
        c4, c4_rv = c4_leg(counts_job, K, comm)
        c4["stream"] = ("balanced slabs" if counts_job is counts and binfo["balanced_cell"] else "default build") + \
            " (what vireo_wrap's policy picks for %d expected iterations)" % c4_iters
        preceded_by.append("c4 leg (vireo_wrap n_init=32 on the same data: ~0.6 s of fits)")
        if solo and not args.no_side_legs:
            c4["doublet"] = side_leg("doublet", doublet_leg, counts_job, K, c4_rv)
            c4["doublet_s"] = c4["doublet"].get("doublet_s")
            preceded_by.append("doublet step on the c4 winner (3 x predict_doublet, 136 columns)")
        del c4_rv

    # The whole timing protocol _fit_VB(min_iter=5, max_iter=20, delay_fit_theta=3) of this rank's
    # restart, on a second device model from the same constructor draws: what vireo_wrap runs per
    # restart (vireo_wrap.py:84-87) and, on rank 0 at N = 1, the GPU half of the parity check (the
    # CPU oracle runs the same fit after the timed region).  It is Vireo._fit_VB taken apart --
    # device model + upload first, the fit here, the download after the timed region -- so that the
    # LAST thing the chip does before the warm-up is 20 (rank 0, N = 1: 2 x 20) iterations of
    # this very workload with no host work in between: on this part a 5-ms pause (one 45-MB
    # upload) is enough for the clocks to drop, and the next ~15 ms then run ~10 % slow
    # (scratch/gap_probe.py, DESIGN_HISTORY.md section 6).
    fit_args = (PROTOCOL["max_iter"], PROTOCOL["min_iter"], 1e-2, PROTOCOL["delay_fit_theta"])
    proto_dm, _ = host._device_model(counts, None)
    pert_dm = None
    if solo and not args.no_cpu:     # (N = 1 only: the other ranks would wait for the oracle)
        # the same fit from an initial ID_prob perturbed by 1e-13 relative (the size of the GPU/CPU
        # difference over the first iterations): the first iterations leave a symmetric, unstable
        # state (posteriors uniform to ~3e-7) and rounding-order differences are amplified while
        # the clusters form; no two implementations can agree better mid-trace than this
        np.random.seed(1)
        pert = Vireo(n_var=N, n_cell=M, n_donor=K)
        pert.ID_prob = pert.ID_prob * (1.0 + 1e-13 * np.random.default_rng(7).standard_normal(pert.ID_prob.shape))
        pert_dm, _ = pert._device_model(counts, None)
        del pert
    tg = time.perf_counter()
    gfull, git, _ = proto_dm.fit(*fit_args)
    tg = time.perf_counter() - tg
    n_pre = 1
    if pert_dm is not None:
        pfull, pit, _ = pert_dm.fit(*fit_args)
        parity_gpu = (host, proto_dm, gfull[:git], pert_dm, pfull[:pit], tg)
        n_pre = 2
    preceded_by.append("whole-protocol fit of this rank's restart on the GPU (%d x %d iterations%s)"
                       % (n_pre, git + 1, "; the parity check downloads their results after the timed region"
                          if pert_dm is not None else ""))

    if args.warmup > 0:
        dm.run_iters(args.warmup, theta_from_iter=PROTOCOL["delay_fit_theta"])
    comm.barrier()
    t0 = time.perf_counter()
    trace, ms_dev = dm.run_iters(args.steps, theta_from_iter=0)   # syncs the stream
    t1 = time.perf_counter()
    comm.barrier()
    wall = t1 - t0
    walls = comm.allgather(np.array([wall]))
    wall_max = float(np.max(walls))
    last_elbos = comm.allgather(np.array([trace[-1]]))            # the restart-shard exchange
    proto_elbos = comm.allgather(np.array([gfull[git - 1]]))      # ELBO[:it][-1] of every rank's protocol fit
    # spread of the number above: the same K iterations four more times (not part of `value`)
    repeats = [wall / args.steps * 1e3]
    for _ in range(4):
        t0 = time.perf_counter()
        dm.run_iters(args.steps, theta_from_iter=0)
        repeats.append((time.perf_counter() - t0) / args.steps * 1e3)

    # roofline leg: the same K iterations again with every pass bracketed by HIP events on
    # the library's stream
    # (three times; the line reports the run with the median pass time and lists all three: the
    #  boxes of the pool show a sporadic slow stretch -- one run of K iterations 15-30 % over its
    #  neighbours, `ms_per_step_repeats` catches them too -- and one such stretch under this leg
    #  used to BE the roofline figure)
    prof_runs = []
    for _ in range(3):
        dm.profile(True)
        dm.run_iters(args.steps, theta_from_iter=0)
        prof_runs.append(dm.profile_read())
    dm.profile(False)
    prof_runs.sort(key=lambda r: r[0][_lib.KERN_VARIANT_PASS] + r[0][_lib.KERN_CELL_PASS])
    ms, n = prof_runs[1]
    kinfo = dm.info()
    # (collective: every rank.  At N > 1 an exception in it is NOT swallowed like a side leg's: a rank
    #  that stayed alive after a failed collective would leave the others waiting for it -- it exits,
    #  the launcher stops the rest, the reason is on stderr)
    comm_rec = comm_leg(comm, local, dm, host) if world > 1 else side_leg("comm", comm_leg, comm, local, dm, host)
    dm.close()
    if parity_gpu is None:
        proto_dm.close()

    unbalanced = None
    if solo and counts_default is not None:
        def _unbalanced():
            np.random.seed(1)
            h = Vireo(n_var=N, n_cell=M, n_donor=K)
            d2, _ = h._device_model(counts_default, None)
            d2.run_iters(20, theta_from_iter=PROTOCOL["delay_fit_theta"])
            t0 = time.perf_counter()
            d2.run_iters(100, theta_from_iter=0)
            ms_it = (time.perf_counter() - t0) / 100 * 1e3
            d2.profile(True)
            d2.run_iters(50, theta_from_iter=0)
            pm, pn = d2.profile_read()
            i2 = d2.info()
            d2.close()
            gain_ms = ms_it - float(np.median(repeats))
            added = t_up_again - t_up_default
            return dict(ms_per_iteration=ms_it, iterations_per_s=1e3 / ms_it,
                        passes_ms={"variant_pass": pm[0] / max(pn[0], 1), "cell_pass": pm[1] / max(pn[1], 1),
                                   "dense_kernels": pm[2] / 50},
                        pad_variant=i2["pad_variant"], pad_cell=i2["pad_cell"], build_s=round(t_up_default, 3),
                        balanced_build_s=round(t_up_again, 3), balanced_first_build_in_process_s=round(t_up, 3),
                        balancing_added_s=round(added, 3),
                        balance_seconds_on_the_build_thread=round(binfo["balance_seconds"], 3),
                        break_even_iterations=int(added / max(gain_ms, 1e-6) * 1e3) if gain_ms > 0 else None,
                        note="the same iterations on the default build of the same problem: 100 after 20; "
                             "balancing_added_s = wall of a balanced build - wall of the default build, both "
                             "after the process's first build")
        unbalanced = side_leg("unbalanced_stream", _unbalanced)

    c3_skew = None
    if solo and not args.no_side_legs and args.config == "c3":
        c3_skew = side_leg("c3_skew", c3_skew_leg, local, K, (float(np.median(repeats)), nnz))

    c3_flags = None
    if solo and not args.no_side_legs and args.config == "c3":
        c3_flags = side_leg("c3_flags", c3_flags_leg, counts, w["GT"], K, float(np.median(repeats)))

    e2e = None
    if solo and not args.no_side_legs and args.config == "c3":
        e2e = side_leg("e2e", e2e_leg, w, K)

    # ---- CPU legs (rank 0): the oracle beside the GPU results formed above --------------------
    if c5 is not None and c5_data is not None:
        c5 = side_leg("c5 (oracle half)", c5_cpu_leg, c5, c5_data)
    c5_data = None
    parity = None
    cpu = None
    if parity_gpu is not None:
        dev, dev_dm, gtrace, pert_dm, ptrace, tg = parity_gpu
        dev._pull(dev_dm, want_GT=True)
        dev_dm.close()
        pert_dm.close()
        self_rel = (np.abs(ptrace - gtrace) / np.abs(gtrace) if len(ptrace) == len(gtrace) else None)
        dt, st, ctrace, it_cpu = cpu_protocol_leg(w, K, seed=1)
        n_cpu = len(ctrace)
        same_len = len(gtrace) == n_cpu
        rel_it = np.abs(gtrace - ctrace) / np.abs(ctrace) if same_len else None
        rel = None if rel_it is None else np.max(rel_it)
        differ = dev.ID_prob.argmax(1) != st.ID_prob.argmax(1)
        # the arbiter: the same protocol in 80-bit extended precision (tests/golden/
        # make_c3_arbiter.py, run once in the build container) -- which float64 trace is closer to
        # the mathematics where rounding differences are amplified (iterations 6-10)
        arb = None
        apath = os.path.join(ROOT, "tests", "golden", "%s_protocol_longdouble.npz" % args.config)
        if os.path.exists(apath) and same_len:
            g = np.load(apath)
            if int(g["n_iter"]) == n_cpu and int(g["nnz"]) == nnz:
                exact = g["elbo_hi"].astype(np.longdouble) + g["elbo_lo"].astype(np.longdouble)
                e_gpu = np.abs((gtrace.astype(np.longdouble) - exact) / exact).astype(float)
                e_cpu = np.abs((ctrace.astype(np.longdouble) - exact) / exact).astype(float)
                arb = dict(source="tests/golden/%s_protocol_longdouble.npz (np.longdouble, eps 1.1e-19)" % args.config,
                           gpu_vs_exact_per_iteration=[float("%.3g" % x) for x in e_gpu],
                           oracle_vs_exact_per_iteration=[float("%.3g" % x) for x in e_cpu],
                           gpu_vs_exact_max=float(e_gpu.max()), oracle_vs_exact_max=float(e_cpu.max()),
                           assignments_equal_exact=bool(np.array_equal(dev.ID_prob.argmax(1), g["assign"])))
                if "ID_prob" in g.files:    # (round 4) the end-state posteriors of the protocol
                    xi, xg = g["ID_prob"], g["GT_prob_sample"]
                    sl = slice(None, None, int(g["GT_stride"]))
                    arb["end_state"] = dict(
                        id_prob_max_abs_gpu_vs_exact=float(np.max(np.abs(dev.ID_prob - xi))),
                        id_prob_max_abs_oracle_vs_exact=float(np.max(np.abs(st.ID_prob - xi))),
                        gt_prob_sample_max_abs_gpu_vs_exact=float(np.max(np.abs(dev.GT_prob[sl] - xg))),
                        gt_prob_sample_max_abs_oracle_vs_exact=float(np.max(np.abs(st.GT_prob[sl] - xg))),
                        gt_sample="every %d-th variant" % int(g["GT_stride"]))
        parity = dict(protocol="_fit_VB(min_iter=5, max_iter=20, delay_fit_theta=3) from "
                               "np.random.seed(1): ELBO[:it] without the binomial constant",
                      iterations_gpu=len(gtrace), iterations_cpu=n_cpu,
                      elbo_trace_max_rel_err=None if rel is None else float(rel),
                      elbo_trace_rel_err_per_iteration=None if rel_it is None else
                      [float("%.3g" % x) for x in rel_it],
                      elbo_final_rel_err=None if rel_it is None else float(rel_it[-1]),
                      gpu_self_sensitivity_per_iteration=None if self_rel is None else
                      [float("%.3g" % x) for x in self_rel],   # (to a 1e-13 perturbation)
                      elbo_final_gpu=float(gtrace[-1]), elbo_final_cpu=float(ctrace[-1]),
                      id_prob_max_abs_err=float(np.max(np.abs(dev.ID_prob - st.ID_prob))),
                      assignment_mismatches=int(differ.sum()),
                      identical_assignments=bool(not differ.any()),
                      extended_precision_arbiter=arb,
                      gpu_fit_wall_s=round(tg, 3))
        # the oracle executes one more iteration than it keeps (ELBO[:it], vireo_model.py:276)
        cpu = dict(value=(n_cpu + 1) / dt, unit="EM iterations/s", cores=1, kind="port",
                   sample="the whole timing protocol on the NumPy/SciPy oracle (the reference's 13 "
                          "SpMM + 2 sparse-subtract op sequence per iteration, single-threaded "
                          "like scipy.sparse): %d iterations of the %s inputs in %.1f s; box has "
                          "%d cores" % (n_cpu + 1, args.config, dt, os.cpu_count()))
        del st
    del w

    if rank == 0:
        B = algorithmic_bytes(N, M, K, T, nnz)
        avg_v = ms[_lib.KERN_VARIANT_PASS] / max(n[_lib.KERN_VARIANT_PASS], 1)
        avg_c = ms[_lib.KERN_CELL_PASS] / max(n[_lib.KERN_CELL_PASS], 1)
        avg_d = ms[_lib.KERN_DENSE] / max(args.steps, 1)
        dom = "cell" if avg_c >= avg_v else "variant"
        avg = max(avg_c, avg_v)
        ach = B[dom] / (avg * 1e-3) / 1e9
        info = _lib.device_info(local)
        # HBM traffic of the dominant kernel: PMC counters cannot be read from inside this
        # process; the figure comes from the committed rocprofv3 --pmc run of this same command
        # (profiles/traffic_<config>.json, scratch/collect_traffic.py) and is quoted only when
        # that run used the kernel sources this process was built from.
        traffic, traffic_src = None, None
        kname = ("vrx_spmm_lds<%d>" % (dom == "cell")) if kinfo["lds_" + dom] else None
        tpath = os.path.join(ROOT, "profiles", "traffic_%s.json" % args.config)
        if kname and os.path.exists(tpath):
            doc = json.load(open(tpath))
            rec = doc.get("kernels", {}).get(kname)
            same_stream = args.config != "c3" or doc.get("stream") == stream_choice()
            if rec and doc.get("kernel_source_hash") == kernel_source_hash() and same_stream:
                traffic = rec["traffic_bytes"]
                traffic_src = ("profiles/traffic_%s.json (rocprofv3 --pmc FETCH_SIZE x2 + "
                               "WRITE_SIZE, kernel sources %s%s)"
                               % (args.config, kernel_source_hash(),
                                  ", " + doc["stream"] if doc.get("stream") else ""))
            elif rec and not same_stream:
                traffic_src = ("profiles/traffic_%s.json was collected on the other stream build (%s): "
                               "not quoted" % (args.config, doc.get("stream")))
            elif rec:
                traffic_src = ("profiles/traffic_%s.json was collected on other kernel sources "
                               "(%s, now %s): not quoted" % (args.config, doc.get("kernel_source_hash"),
                                                             kernel_source_hash()))
        out = {
            "metric": "EM iterations/sec", "value": world * args.steps / wall_max,
            "unit": "EM iterations/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": wall_max / args.steps * 1e3,
            "higher_is_better": True,
            # one restart per rank whatever N is: weak scaling -- when the ranks are GPUs talking
            # over RCCL.  Ranks that share a device over host sockets only rehearse the plumbing.
            "scaling": "weak" if world == 1 or comm.backend == "rccl" else "plumbing-only",
            "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s: synthetic sparse AD/DP N=%d variants x M=%d cells, K=%d "
                                   "donors, nnz=%d (SURVEY.md 8d generator, seed 0); one restart "
                                   "per GPU" % (args.config, N, M, K, nnz),
                       "N": N, "M": M, "K": K, "nnz": nnz, "device": info["name"],
                       "restart_elbos": [float(x) for x in np.ravel(last_elbos)],
                       "best_restart": int(np.argmax(last_elbos)),
                       "restart_protocol_elbos": [float(x) for x in np.ravel(proto_elbos)],
                       "stream": ("balanced slabs (per-tile choice of the contracted rows of a slab; +%.2f s at problem "
                                  "build)" % binfo["balance_seconds"]) if binfo["balanced_cell"] else "default build",
                       "build_info": binfo,
                       "host_setup_s": {"generate": round(t_gen, 1), "upload+transpose": round(t_up, 2)}},
            "roofline": {"bound": "hbm",
                         "kernel": "%s (%s pass)" % (
                             "vrx_spmm_lds<MODE %d, FORM %d>" % (dom == "cell", kinfo["cell_form"] if dom == "cell" else kinfo["var_form"])
                             if kinfo["lds_" + dom]
                             else "vrx_spmm<..,%d,fmt%d>" % (dom == "cell", kinfo["fmt_" + dom]), dom),
                         "kernel_info": kinfo,
                         "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": B[dom],
                         "avg_launch_ms": avg,
                         "per_iteration_ms": {"variant_pass": avg_v, "cell_pass": avg_c,
                                              "dense_kernels": avg_d},
                         "event_runs_ms": [{"variant_pass": r[0][_lib.KERN_VARIANT_PASS] / max(r[1][_lib.KERN_VARIANT_PASS], 1),
                                            "cell_pass": r[0][_lib.KERN_CELL_PASS] / max(r[1][_lib.KERN_CELL_PASS], 1),
                                            "dense_kernels": r[0][_lib.KERN_DENSE] / max(args.steps, 1)}
                                           for r in prof_runs],
                         "event_runs_note": "the K iterations three times with HIP events around every pass; "
                                            "`avg_launch_ms` / `per_iteration_ms` are the run with the median pass time",
                         "whole_iteration": {"algorithmic_bytes": B["total"],
                                             "achieved_GBs": B["total"] / (wall_max / args.steps) / 1e9,
                                             "frac": B["total"] / (wall_max / args.steps) / 1e9 / HBM_PEAK_GBS}},
            "comm": comm_rec,
            "preceded_by": preceded_by,
            # what precedes the timed window differs with the flags, so `value` is
            # like-for-like only between lines that carry the same tag
            "protocol": {"version": "r6", "warmup": args.warmup, "steps": args.steps,
                         "legs_before_timed_region": len(preceded_by),
                         "flags": {"no_cpu": bool(args.no_cpu), "no_c4": bool(args.no_c4),
                                   "no_side_legs": bool(args.no_side_legs)},
                         "tag": "r6:w%d:k%d:%s" % (args.warmup, args.steps, "+".join(
                             x for x, on in (("c2c5", c2 is not None), ("c4", c4 is not None),
                                             ("parity", parity_gpu is not None)) if on) or "bare")},
            "cpu_baseline": cpu,
            "parity": parity,
            "c4": c4,
            "c2": c2,
            "c5": c5,
            "unbalanced_stream": unbalanced,
            "c3_skew": c3_skew,
            "c3_flags": c3_flags,
            "e2e": e2e,
            "ms_per_step_repeats": {"runs": [round(x, 4) for x in repeats],
                                    "min": round(min(repeats), 4), "median": round(float(np.median(repeats)), 4),
                                    "max": round(max(repeats), 4),
                                    "note": "run 0 is the timed run `value` comes from; runs 1-4 "
                                            "repeat the same K iterations on this rank"},
        }
        if cpu:
            out["speedup_vs_cpu_1core"] = out["value"] / cpu["value"]
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if not isinstance(comm, vdist.LocalComm):
        comm.barrier()
        comm.close()


if __name__ == "__main__":
    main()
