"""The misses of a `fuzz_sweep.py` run (its OUTDIR/seed_<n>.npz device results) against the 80-bit
arbiter of tests/golden/make_bmm_arbiter.py: is the device or the oracle further from exact?  CPU only.

    python tests/perf/fuzz_arbiter_check.py gpurun_out/r5_12/miss > profiles/r05_fuzz_sweep_1000_2999_arbiter.txt
"""
import glob
import importlib.util
import multiprocessing as mp
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np                                              # noqa: E402

spec = importlib.util.spec_from_file_location("arb", os.path.join(ROOT, "tests", "golden", "make_bmm_arbiter.py"))
arb = importlib.util.module_from_spec(spec)
spec.loader.exec_module(arb)


def same_calls(x, exact):
    """argmax equal, or the exact row is a tie to 1e-9 (cells without information: a uniform row)"""
    a, b = x.argmax(1), exact.argmax(1)
    top = np.sort(exact, axis=1)
    tie = top[:, -1] - top[:, -2] < 1e-9 * top[:, -1]
    return bool(np.all((a == b) | tie))


def rel(x, exact):
    m = exact > 1e-290
    return float(np.max(np.abs(x[m] - exact[m]) / exact[m])) if m.any() else 0.0


def one(job):
    seed, path = job
    from tests.test_gpu_fuzz import draw_case
    from oracle import vireo_oracle as O
    gpu = np.load(path)
    AD, DP, K, rng = draw_case(seed)
    N, M = AD.shape
    if seed % 4 == 3:
        _, out, _ = arb.exact_bmm(seed)
        K = max(K, 2)
        np.random.seed(seed)
        init = np.random.rand(M, K)
        ref = O.bmm_new(M, N, K, ID_prob_init=init.copy())
        O.bmm_fit_vb(ref, AD, DP, min_iter=2, max_iter=4)
        ex = out["s%d_ID_prob" % seed]
        return (seed, "bmm", N, M, K, int(DP.max()), rel(gpu["gpu_ID_prob"], ex), rel(ref.ID_prob, ex), None, None,
                same_calls(gpu["gpu_ID_prob"], ex))
    _, out, _ = arb.exact_vireo(seed)
    rng.choice([1, 16, 1024])
    flags = dict(ASE_mode=bool(rng.random() < 0.2), fix_beta_sum=bool(rng.random() < 0.2),
                 learn_theta=bool(rng.random() < 0.85))
    np.random.seed(seed)
    ref = O.vireo_new(M, N, K, **flags)
    O.vireo_fit(ref, AD, DP, min_iter=2, max_iter=5, delay_fit_theta=1)
    ex, rows, exg = out["s%d_ID_prob" % seed], out["s%d_GT_rows" % seed], out["s%d_GT_prob" % seed]
    return (seed, "vireo", N, M, K, int(DP.max()), rel(gpu["gpu_ID_prob"], ex), rel(ref.ID_prob, ex),
            rel(gpu["gpu_GT_prob"][rows], exg), rel(ref.GT_prob[rows], exg),
            same_calls(gpu["gpu_ID_prob"], ex))


def main():
    d = sys.argv[1]
    jobs = sorted((int(re.search(r"seed_(\d+)", p).group(1)), p) for p in glob.glob(os.path.join(d, "seed_*.npz")))
    print("worst relative distance from the 80-bit result (elements > 1e-290); GT_prob on the variants where the oracle is > 1e-6 off + every 8th")
    print("%5s %-6s %-26s | ID_prob: %9s %9s | GT_prob: %9s %9s | %s" % ("seed", "kind", "case", "GPU", "oracle", "GPU", "oracle", "assignments = exact (ties aside)"))
    worse = 0
    with mp.Pool(min(8, os.cpu_count() or 1)) as pool:
        for r in pool.imap(one, jobs):
            seed, kind, N, M, K, top, gi, oi, gg, og, same = r
            f = lambda x: "%9.1e" % x if x is not None else "        -"      # noqa: E731
            gpu_worst, orc_worst = max(gi, gg or 0.0), max(oi, og or 0.0)
            worse += gpu_worst > max(orc_worst, 1e-5)
            print("%5d %-6s %-26s |          %s %s |          %s %s | %s%s" % (
                seed, kind, "N=%d M=%d K=%d top=%d" % (N, M, K, top), f(gi), f(oi), f(gg), f(og), same,
                "   <-- GPU further than the oracle" if gpu_worst > max(orc_worst, 1e-5) else ""), flush=True)
    print("%d cases; the device is further from exact than max(1e-5, the oracle) in %d of them" % (len(jobs), worse))


if __name__ == "__main__":
    main()
