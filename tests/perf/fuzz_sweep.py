"""The random-case comparison of tests/test_gpu_fuzz.py for any seed range (MI355X).

    python tests/perf/fuzz_sweep.py LO HI [OUTDIR [KINDS]]      KINDS: keep the device results of these kinds only (vireo,bmm)
    python tests/perf/fuzz_sweep.py @FILE 0 [OUTDIR [KINDS]]    the seeds listed in FILE (whitespace / comma separated)

One line per case on stdout (kind, shape, K, largest count, iterations, worst relative error per
compared array), `MISS` where an array misses rtol 1e-5 against the oracle; for every miss the
device results are kept in OUTDIR/seed_<n>.npz so that the build container can set them beside
the 80-bit arbiter (tests/golden/make_bmm_arbiter.py).  The log of the round's sweep is committed
under profiles/ (r05_fuzz_sweep_*.log)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np                                              # noqa: E402

from tests.test_gpu_fuzz import RTOL, draw_case                 # noqa: E402
from oracle import vireo_oracle as O                            # noqa: E402   (the checker)
import vireo_amd as va                                          # noqa: E402
from vireo_amd.counts import DeviceCounts                       # noqa: E402


def worst(a, b):
    """largest |a - b| / |b| over the elements the suite's atol = 1e-300 does not excuse"""
    a, b = np.asarray(a, dtype=float).ravel(), np.asarray(b, dtype=float).ravel()
    d = np.abs(a - b)
    m = d > 1e-300
    if not m.any():
        return 0.0, 0
    r = d[m] / np.maximum(np.abs(b[m]), 1e-320)
    return float(r.max()), int((r > RTOL).sum())


def run_case(seed):
    AD, DP, K, rng = draw_case(seed)
    N, M = AD.shape
    os.environ["VIREO_LDS"] = "1" if seed % 2 else "0"
    os.environ["VIREO_LDS_BLOCKS"] = str(int(rng.choice([1, 16, 1024])))
    counts = DeviceCounts(AD, DP)
    keep = {}
    if seed % 4 == 3:
        kind = "bmm"
        K = max(K, 2)
        np.random.seed(seed)
        init = np.random.rand(M, K)
        ref = O.bmm_new(M, N, K, ID_prob_init=init.copy())
        dev = va.BinomMixtureVB(n_cell=M, n_var=N, n_donor=K, ID_prob_init=init.copy())
        O.bmm_fit_vb(ref, AD, DP, min_iter=2, max_iter=4)
        dev._fit_BV(AD, DP, min_iter=2, max_iter=4, verbose=False)
        its = (len(dev.ELBO_iters), len(ref.ELBO_iters))
        pairs = dict(ELBO=(dev.ELBO_iters, ref.ELBO_iters), ID_prob=(dev.ID_prob, ref.ID_prob),
                     beta_mu=(dev.beta_mu, ref.beta_mu), beta_sum=(dev.beta_sum, ref.beta_sum))
    else:
        kind = "vireo"
        flags = dict(ASE_mode=bool(rng.random() < 0.2), fix_beta_sum=bool(rng.random() < 0.2),
                     learn_theta=bool(rng.random() < 0.85))
        np.random.seed(seed)
        ref = O.vireo_new(M, N, K, **flags)
        np.random.seed(seed)
        dev = va.Vireo(n_cell=M, n_var=N, n_donor=K, **flags)
        O.vireo_fit(ref, AD, DP, min_iter=2, max_iter=5, delay_fit_theta=1)
        dev.fit(counts, None, min_iter=2, max_iter=5, delay_fit_theta=1, verbose=False)
        its = (len(dev.ELBO_), len(ref.ELBO_))
        pairs = dict(ELBO=(dev.ELBO_, ref.ELBO_), ID_prob=(dev.ID_prob, ref.ID_prob),
                     GT_prob=(dev.GT_prob, ref.GT_prob), beta_mu=(dev.beta_mu, ref.beta_mu),
                     beta_sum=(dev.beta_sum, ref.beta_sum))
    counts.close()
    errs = {}
    miss = its[0] != its[1]
    for name, (a, b) in pairs.items():
        if its[0] != its[1] and name == "ELBO":
            continue
        errs[name] = worst(a, b)
        miss = miss or errs[name][1] > 0
        keep["gpu_" + name] = np.asarray(a)
    return dict(kind=kind, N=N, M=M, K=K, top=int(DP.max()), its=its, errs=errs, miss=miss), keep


def main():
    if sys.argv[1].startswith("@"):     # a seed list (round 6: the misses of an earlier sweep, for the arbiter)
        seeds = [int(x) for x in open(sys.argv[1][1:]).read().replace(",", " ").split()]
        lo, hi = min(seeds), max(seeds) + 1
    else:
        lo, hi = int(sys.argv[1]), int(sys.argv[2])
        seeds = range(lo, hi)
    outdir = sys.argv[3] if len(sys.argv) > 3 else None
    keep_kinds = sys.argv[4].split(",") if len(sys.argv) > 4 else ("vireo", "bmm")
    if outdir:
        os.makedirs(outdir, exist_ok=True)
    n = {"vireo": 0, "bmm": 0}
    missed = {"vireo": [], "bmm": []}
    for seed in seeds:
        try:
            r, keep = run_case(seed)
        except Exception as e:      # noqa: BLE001 -- a crash is a finding of the sweep too
            print("seed %d ERROR %r" % (seed, e), flush=True)
            missed.setdefault("error", []).append(seed)
            continue
        n[r["kind"]] += 1
        print("seed %d %s %s N=%d M=%d K=%d top=%d its=%d/%d  %s" % (
            seed, "MISS" if r["miss"] else "ok  ", r["kind"], r["N"], r["M"], r["K"], r["top"],
            r["its"][0], r["its"][1],
            "  ".join("%s %.2e(%d)" % (k, v[0], v[1]) for k, v in r["errs"].items())), flush=True)
        if r["miss"]:
            missed[r["kind"]].append(seed)
            if outdir and r["kind"] in keep_kinds:
                np.savez(os.path.join(outdir, "seed_%d.npz" % seed), **keep)
    print("SUMMARY seeds %d..%d: %d vireo cases, %d miss %s; %d bmm cases, %d miss %s; errors %s"
          % (lo, hi - 1, n["vireo"], len(missed["vireo"]), missed["vireo"], n["bmm"],
             len(missed["bmm"]), missed["bmm"], missed.get("error", [])))


if __name__ == "__main__":
    main()
