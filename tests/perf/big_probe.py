"""Size ceiling: problems far beyond c3 on one MI355X -- build time, HBM in use,
ms per iteration, whole-iteration roofline -- and parity where the oracle cannot iterate (the
column-subset trick, tests/subset_parity.py).

    python tests/perf/big_probe.py 16x      N = 400k x M = 200k, K = 16, density 0.02: 1.6e9 entries (16x c3)
    python tests/perf/big_probe.py 2g       N = 480k x M = 230k: 2.2e9 entries, past 2^31 (device build to 2^32 - 1)
    python tests/perf/big_probe.py 4x       N = 200k x M = 100k: the size of the -m gpu test

One JSON line per run (kept in profiles/r06_big_probe_*.txt).  Lives under tests/ because it uses the
CPU oracle as checker.
"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np                                              # noqa: E402

SIZES = {"4x": (200000, 100000), "16x": (400000, 200000), "2g": (480000, 230000)}


def vram_used():
    out = subprocess.run(["rocm-smi", "--showmeminfo", "vram"], capture_output=True, text=True).stdout
    for line in out.splitlines():
        if "Total Used" in line:
            return int(line.split(":")[-1])
    return None


def main(size, check_theta=True):
    from vireo_amd import _lib, synth
    from vireo_amd.counts import DeviceCounts
    from vireo_amd.vireo_model import Vireo
    from tests.subset_parity import one_iteration_subset_check
    N, M = SIZES[size]
    K, dens = 16, 0.02
    t0 = time.time()
    w = synth.big_workload(N, M, K, dens, seed=0, threads=int(os.environ.get("BIG_PROBE_THREADS", "48")))
    t_gen = time.time() - t0
    nnz = int(w["rowidx"].size)
    rss_gen = __import__("resource").getrusage(0).ru_maxrss / 1048576.0
    v0 = vram_used()
    t0 = time.time()
    counts = DeviceCounts.from_merged(w["shape"], w["colptr"], w["rowidx"], w["ad"], w["dp"], device=0)
    t_build = time.time() - t0
    v_problem = vram_used()
    np.random.seed(1)
    m = Vireo(n_var=N, n_cell=M, n_donor=K)
    t0 = time.time()
    m.fit(counts, None, min_iter=5, max_iter=20, delay_fit_theta=3, verbose=False)
    t_fit = time.time() - t0
    dm, _ = m._device_model(counts, None)
    info = dm.info()
    dm.run_iters(10, theta_from_iter=0)
    t0 = time.perf_counter()
    tr, _ = dm.run_iters(30, theta_from_iter=0)
    ms = (time.perf_counter() - t0) / 30 * 1e3
    dm.profile(True)
    dm.run_iters(20)
    pm, pn = dm.profile_read()
    v_model = vram_used()
    dm.close()
    lab = m.ID_prob.argmax(1)
    conf = np.zeros((K, K), int)
    np.add.at(conf, (w["z"], lab), 1)
    purity = conf.max(1).sum() / M
    t0 = time.time()
    par = one_iteration_subset_check(m, counts, w, n_sub=2000, check_theta=check_theta)
    t_par = time.time() - t0
    B = 2 * 12 * nnz + 4 * (N + M + 2) + 8 * (2 * M * K + 2 * N * K * 3 + 8 * N * K)
    print(json.dumps(dict(
        size=size, N=N, M=M, K=K, nnz=nnz, nnz_over_2_31=round(nnz / 2.0 ** 31, 3), generate_s=round(t_gen, 1),
        host_peak_rss_gb_after_generation=round(rss_gen, 1), build_s=round(t_build, 2),
        device_built=bool(info["lds_variant"] and info["lds_cell"] and info["tiles_variant"] == 1),
        protocol_fit_s=round(t_fit, 3), protocol_elbo_entries=len(m.ELBO_), purity=round(float(purity), 5),
        ms_per_iteration=ms, iterations_per_s=1e3 / ms,
        passes_ms=dict(variant=pm[0] / max(pn[0], 1), cell=pm[1] / max(pn[1], 1), dense=pm[2] / 20),
        elbo_finite=bool(np.all(np.isfinite(tr))), algorithmic_bytes_per_iteration=B,
        whole_iteration_roofline_frac=B / (ms * 1e-3) / 8e12,
        hbm_bytes_in_use=dict(before=v0, problem=v_problem, with_model=v_model),
        subset_parity=dict(par, seconds=round(t_par, 1), theta_checked=check_theta,
                           note="one GPU iteration from the fitted state vs the oracle on 2 000 variants / 2 000 cells "
                                "(exact for those rows) and, for theta, the oracle's whole-matrix sums; rtol 1e-5"),
        info=info, device=_lib.device_info(0)["name"])), flush=True)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "16x", check_theta=os.environ.get("BIG_PROBE_THETA", "1") == "1")
