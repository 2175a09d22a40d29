"""Would a REFERENCE-GROUPED clone-mode cell pass bring the HIP path within 1e-5 of the oracle on the
sweep's 22 deviation cases?  A float64 emulation on the CPU -- the question is
one of arithmetic, not of kernels.

    python tests/perf/bmm_grouping_study.py [dir with the sweep's seed_<n>.npz device results]

The reference forms  L = AD'psi(s1) + BD'psi(s2) - DP'psi(s1 + s2)  (bmm_model.py:125-129): three
SciPy products, each a sequential sum in increasing variant order, rounded separately, then
combined.  The kernels accumulate  ad (psi1 - psis) + bd (psi2 - psis).  Each case's WHOLE fit
(_fit_BV(min_iter=2, max_iter=4): four iterations) is repeated in float64 with the cell log
likelihood formed three ways, everything else being the oracle's code:

  ref      the reference's three products (= the oracle, bit for bit)
  grp-blk  the reference's GROUPING (three separately rounded sums, combined the same way), each
           summed in blocks of 64 variants whose partial sums are then added -- what any tiled /
           multi-lane GPU accumulation does: same grouping, different order
  fused    ad (psi1 - psis) + bd (psi2 - psis), sequential  (the kernels' grouping)

and the worst relative error of the END-STATE posteriors (elements > 1e-290) is reported against
`ref` and against the 80-bit run (tests/golden/fuzz_arbiter.npz), next to the device's.  The column
"1 step" is grp-blk against ref when only the LAST cell update differs (same state before it): the
size of one update's rounding difference before the next theta step amplifies it.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np                                              # noqa: E402
from scipy.sparse import csr_matrix                             # noqa: E402
from scipy.special import digamma                               # noqa: E402

from tests.test_gpu_fuzz import draw_case                       # noqa: E402
from oracle import vireo_oracle as O                            # noqa: E402


def worst(a, b):
    m = b > 1e-290
    return float(np.max(np.abs(a[m] - b[m]) / b[m]))


def fit(AD, DP, K, init, n_exec, loglik):
    N, M = AD.shape
    st = O.bmm_new(M, N, K, ID_prob_init=init.copy())
    for _ in range(n_exec):
        O.bmm_theta_step(st, AD, DP)
        s1, s2 = st.beta_mu * st.beta_sum, (1 - st.beta_mu) * st.beta_sum
        O.bmm_id_step(st, loglik(digamma(s1), digamma(s2), digamma(s1 + s2)))
    return st


def main():
    gdir = sys.argv[1] if len(sys.argv) > 1 else None
    g = np.load(os.path.join(ROOT, "tests", "golden", "fuzz_arbiter.npz"))
    print("%5s  %-28s | end state vs ref: %8s %8s %8s  %8s | vs exact: %8s %8s %8s %8s" % (
        "seed", "case", "grp-blk", "fused", "GPU", "(1 step)", "ref", "grp-blk", "fused", "GPU"))
    for seed in [int(x) for x in g["bmm_seeds"]]:
        AD, DP, K, _ = draw_case(seed)
        N, M = AD.shape
        K = max(K, 2)
        np.random.seed(seed)
        init = np.random.rand(M, K)
        n_exec = int(g["s%d_n_exec" % seed])
        BD = DP - AD
        At, Bt, Dt = csr_matrix(AD.T), csr_matrix(BD.T), csr_matrix(DP.T)

        def ref_L(p1, p2, ps):
            return AD.T @ p1 + BD.T @ p2 - DP.T @ ps

        def blk_L(p1, p2, ps):
            acc = [np.zeros((M, K)) for _ in range(3)]
            for lo in range(0, N, 64):
                sl = slice(lo, min(lo + 64, N))
                acc[0] += At[:, sl] @ p1[sl]
                acc[1] += Bt[:, sl] @ p2[sl]
                acc[2] += Dt[:, sl] @ ps[sl]
            return acc[0] + acc[1] - acc[2]

        def fused_L(p1, p2, ps):
            return AD.T @ (p1 - ps) + BD.T @ (p2 - ps)

        ref = fit(AD, DP, K, init, n_exec, ref_L).ID_prob
        blk = fit(AD, DP, K, init, n_exec, blk_L).ID_prob
        fus = fit(AD, DP, K, init, n_exec, fused_L).ID_prob
        # one update's difference: the oracle's state before its last cell update, grp-blk there
        st = fit(AD, DP, K, init, n_exec - 1, ref_L)
        O.bmm_theta_step(st, AD, DP)
        s1, s2 = st.beta_mu * st.beta_sum, (1 - st.beta_mu) * st.beta_sum
        O.bmm_id_step(st, blk_L(digamma(s1), digamma(s2), digamma(s1 + s2)))
        one = st.ID_prob
        exact = g["s%d_ID_prob" % seed]
        gpu = None
        if gdir and os.path.exists(os.path.join(gdir, "seed_%d.npz" % seed)):
            gpu = np.load(os.path.join(gdir, "seed_%d.npz" % seed))["gpu_ID_prob"]
        f = lambda x: "%8.1e" % x                                # noqa: E731
        nog = "       -"
        print("%5d  %-28s |                   %s %s %s  %s |           %s %s %s %s" % (
            seed, "N=%d M=%d K=%d top=%d" % (N, M, K, DP.max()),
            f(worst(blk, ref)), f(worst(fus, ref)), f(worst(gpu, ref)) if gpu is not None else nog,
            f(worst(one, ref)),
            f(worst(ref, exact)), f(worst(blk, exact)), f(worst(fus, exact)),
            f(worst(gpu, exact)) if gpu is not None else nog), flush=True)


if __name__ == "__main__":
    main()
