"""Extended-precision arbiter for the random cases of tests/test_gpu_fuzz.py::draw_case on which
the HIP path and the oracle (= the reference's float64 arithmetic) differ by more than 1e-5
relative.

The round-5 sweep of seeds 48 .. 999 (tests/perf/fuzz_sweep.py, log: profiles/r05_fuzz_sweep_48_999.log)
leaves 23 such cases (the fixture also holds five of the 56 a second sweep over 1000 .. 2999 added): 22 clone-mode (``BinomMixtureVB``) cases with deep counts (up to 5000 per
entry) whose SMALL posteriors (1e-6 ... 1e-279) differ by 1e-5 ... 4e-4 relative, and one ``Vireo``
case (ASE mode) with seven ``GT_prob`` entries of ~1e-199 off by 1.7e-5.  The reference forms the
cell log likelihood as three separately rounded sums of ~1e6-1e9 that cancel
(vireoSNP/utils/bmm_model.py:125-129: AD'psi(s1) + BD'psi(s2) - DP'psi(s1+s2)); the kernels
accumulate ad (psi1 - psis) + bd (psi2 - psis), the same quantity without the cancellation.  This
script settles, case by case, which float64 result is closer to the mathematics: the same
iterations in 80-bit extended precision (np.longdouble, eps 1.1e-19; SciPy's sparse products
instantiated for long double; digamma from mpmath at 40 digits).

Run in the build container (no GPU; ~25 minutes on 8 cores from scratch, ARBITER_REBUILD=1; by default only
the seeds the fixture does not hold yet are computed):  python tests/golden/make_bmm_arbiter.py
Output: tests/golden/fuzz_arbiter.npz -- per seed the END STATE of the exact run rounded to
float64: ``s<seed>_ID_prob`` (clone mode: the whole (M, K) posterior); for the Vireo case
``s<seed>_ID_prob``, and ``s<seed>_GT_rows`` / ``s<seed>_GT_prob``: the variants kept (every 8th,
and every variant on which the oracle is further than 1e-6 relative from exact) and their
(rows, K, T) posteriors; ``s<seed>_n_exec`` the iterations executed.  The GPU test
(tests/test_gpu_fuzz.py::test_known_deviation_cases_vs_arbiter) holds |GPU - exact| against
|oracle - exact| element by element.  The update equations are the reference's (cited inline).
"""
import multiprocessing as mp
import os
import sys
import time

import mpmath
import numpy as np
from scipy.sparse import csc_matrix

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, "..", "..")
sys.path.insert(0, ROOT)

LD = np.longdouble
mpmath.mp.dps = 40

# the sweep's misses (profiles/r05_fuzz_sweep_48_999.log, SUMMARY line)
BMM_SEEDS = [75, 155, 171, 211, 239, 343, 367, 463, 563, 595, 611, 643, 695, 719, 755, 803, 811,
             855, 887, 895, 951, 987]
VIREO_SEEDS = [537]
# ... and the worst offenders of the second sweep, seeds 1000 .. 2999 (profiles/r05_fuzz_sweep_1000_2999.log:
# 56 more misses of the same class; the oracle up to 3.4e-3 from exact)
BMM_SEEDS += [1003, 1263, 2907]
VIREO_SEEDS += [1648, 2260]
# ... and (round 6) every case of the three sweeps on which the DEVICE itself is further than 1e-5
# from exact (18 of the first three sweeps' 199 misses, all clone mode; profiles/r06_fuzz_deviation_class.txt): 563, 1263
# and 2907 are above, these are the other fifteen.  The GPU test pins the device's distance from exact
# as an upper bound per seed (tests/test_gpu_fuzz.py::DEVICE_BEYOND_RTOL).
BMM_SEEDS += [1583, 1919, 2547, 2615, 2967, 3811, 4691, 4895, 6887, 7179, 7383, 7583, 7695, 7867, 7911]
# ... and the five of round 6's regression sweep (seeds 8000 .. 9999, profiles/r06_fuzz_sweep_8000_9999_arbiter.txt)
BMM_SEEDS += [8767, 9107, 9215, 9327, 9503]
GT_EVERY = 8
THETA_SEEDS = [1263, 9327]     # cases whose THETA misses 1e-5 too (one beta_sum entry at 1.2e-5): the exact theta is kept


def psi(v):
    """digamma of a long-double array through mpmath (str() of a longdouble round-trips)"""
    flat = np.asarray(v, dtype=LD).ravel()
    return np.array([LD(mpmath.nstr(mpmath.digamma(mpmath.mpf(str(x))), 30)) for x in flat],
                    dtype=LD).reshape(np.shape(v))


def softmax_rows(Z, axis):
    Z = Z - Z.max(axis, keepdims=True)          # loglik_amplify, vireo_base.py:62-74
    P = np.exp(Z)
    return P / P.sum(axis, keepdims=True)       # normalize, vireo_base.py:25-45


def as_ld(X):
    X = csc_matrix(X)
    return csc_matrix((X.data.astype(LD), X.indices, X.indptr), shape=X.shape)


def exact_bmm(seed):
    """tests/test_gpu_fuzz.py's clone-mode branch: _fit_BV(min_iter=2, max_iter=4) from
    ID_prob_init = rand(M, K) under np.random.seed(seed) (bmm_model.py:178-201)"""
    from tests.test_gpu_fuzz import draw_case
    from oracle import vireo_oracle as O
    AD, DP, K, _rng = draw_case(seed)
    N, M = AD.shape
    K = max(K, 2)
    np.random.seed(seed)
    init = np.random.rand(M, K)
    ref = O.bmm_new(M, N, K, ID_prob_init=init.copy())
    O.bmm_fit_vb(ref, AD, DP, min_iter=2, max_iter=4)
    n_exec = len(ref.ELBO_iters) + 1                 # the loop executes one more than it keeps
    A, B = as_ld(AD), as_ld(DP - AD)
    ID = O.unit_sum(init.copy(), axis=1).astype(LD)  # the float64 state both implementations start from
    for _ in range(n_exec):
        t1 = A @ ID + LD(1)                          # bmm_model.py:133-144, prior Beta(1, 1) (:92-98)
        t2 = B @ ID + LD(1)
        s1, s2 = t1, t2                              # mu * sum, (1 - mu) * sum with sum = t1 + t2
        d1, d2, ds = psi(s1), psi(s2), psi(s1 + s2)
        L = A.T @ (d1 - ds) + B.T @ (d2 - ds)        # bmm_model.py:118-130, regrouped (exact in LD to 1e-10)
        ID = softmax_rows(L - np.log(LD(K)), 1)      # bmm_model.py:147-154, uniform ID prior
    out = {"s%d_ID_prob" % seed: ID.astype(np.float64), "s%d_n_exec" % seed: np.int64(n_exec)}
    if seed in THETA_SEEDS:                          # (theta of the last iteration: bmm_model.py:141-144)
        out["s%d_beta_mu" % seed] = (t1 / (t1 + t2)).astype(np.float64)
        out["s%d_beta_sum" % seed] = (t1 + t2).astype(np.float64)
    ex = ID.astype(np.float64)
    m = ex > 1e-290
    dev = float(np.max(np.abs(ref.ID_prob[m] - ex[m]) / ex[m]))
    return seed, out, "bmm N=%d M=%d K=%d top=%d: oracle vs exact %.2e" % (N, M, K, DP.max(), dev)


def exact_vireo(seed):
    """tests/test_gpu_fuzz.py's Vireo branch: fit(min_iter=2, max_iter=5, delay_fit_theta=1) with
    the case's drawn flags (vireo_model.py:251-276)"""
    from tests.test_gpu_fuzz import draw_case
    from oracle import vireo_oracle as O
    AD, DP, K, rng = draw_case(seed)
    N, M = AD.shape
    T = 3
    rng.choice([1, 16, 1024])                        # (the test draws VIREO_LDS_BLOCKS here)
    flags = dict(ASE_mode=bool(rng.random() < 0.2), fix_beta_sum=bool(rng.random() < 0.2),
                 learn_theta=bool(rng.random() < 0.85))
    np.random.seed(seed)
    ref = O.vireo_new(M, N, K, **flags)
    ID = ref.ID_prob.astype(LD)                      # the float64 initial state (vireo_model.py:98,103)
    GT = ref.GT_prob.astype(LD)
    O.vireo_fit(ref, AD, DP, min_iter=2, max_iter=5, delay_fit_theta=1)
    n_exec = len(ref.ELBO_) + 1
    rows = N if flags["ASE_mode"] else 1
    mu0 = np.linspace(0.01, 0.99, T)
    p1 = np.broadcast_to((mu0 * 50.0).astype(LD), (rows, T)).copy()          # vireo_model.py:114-120
    p2 = np.broadcast_to(((1 - mu0) * 50.0).astype(LD), (rows, T)).copy()
    mu = np.broadcast_to(mu0.astype(LD), (rows, T)).copy()                    # :84-92
    sm = np.full((rows, T), 50.0, dtype=LD)
    A, D = as_ld(AD), as_ld(DP)
    B = as_ld(DP - AD)
    for it in range(n_exec):
        S1, SS = A @ ID, D @ ID                      # vireo_model.py:169-170, :207-209
        S2 = SS - S1
        if flags["learn_theta"] and it >= 1:         # update_theta_size, :165-185
            t1, t2 = p1.copy(), p2.copy()
            for g in range(T):
                if flags["ASE_mode"]:
                    t1[:, g] += np.sum(S1 * GT[:, :, g], axis=1)
                    t2[:, g] += np.sum(S2 * GT[:, :, g], axis=1)
                else:
                    t1[:, g] += np.sum(S1 * GT[:, :, g])
                    t2[:, g] += np.sum(S2 * GT[:, :, g])
            mu = t1 / (t1 + t2)
            if not flags["fix_beta_sum"]:
                sm = t1 + t2
        s1, s2 = mu * sm, (1 - mu) * sm
        d1, d2, ds = psi(s1), psi(s2), psi(s1 + s2)  # (rows, T)
        L = np.empty((N, K, T), dtype=LD)            # update_GT_prob, :204-219
        for g in range(T):
            L[:, :, g] = S1 * d1[:, g:g + 1] + S2 * d2[:, g:g + 1] - SS * ds[:, g:g + 1]
        GT = softmax_rows(L - np.log(LD(T)), 2)
        Wa = sum(GT[:, :, g] * (d1[:, g:g + 1] - ds[:, g:g + 1]) for g in range(T))   # update_ID_prob, :187-201
        Wb = sum(GT[:, :, g] * (d2[:, g:g + 1] - ds[:, g:g + 1]) for g in range(T))
        ID = softmax_rows(A.T @ Wa + B.T @ Wb - np.log(LD(K)), 1)
    exg = GT.astype(np.float64)
    m = exg > 1e-290
    rel = np.zeros(exg.shape)
    rel[m] = np.abs(ref.GT_prob[m] - exg[m]) / exg[m]
    keep = np.zeros(N, dtype=bool)
    keep[::GT_EVERY] = True
    keep |= rel.reshape(N, -1).max(1) > 1e-6
    idx = np.flatnonzero(keep)
    out = {"s%d_ID_prob" % seed: ID.astype(np.float64), "s%d_GT_rows" % seed: idx.astype(np.int32),
           "s%d_GT_prob" % seed: exg[idx], "s%d_n_exec" % seed: np.int64(n_exec)}
    return seed, out, "vireo N=%d M=%d K=%d %s: oracle vs exact on GT_prob %.2e, %d rows kept" % (
        N, M, K, flags, float(rel.max()), idx.size)


def _run(job):
    kind, seed = job
    t0 = time.time()
    seed, out, note = (exact_bmm if kind == "bmm" else exact_vireo)(seed)
    return seed, out, "seed %d %s (%.0f s)" % (seed, note, time.time() - t0)


def main():
    jobs = [("vireo", s) for s in VIREO_SEEDS] + [("bmm", s) for s in BMM_SEEDS]
    only = [int(a) for a in sys.argv[1:]]
    if only:
        jobs = [j for j in jobs if j[1] in only]
    store = {}
    path = os.path.join(HERE, "fuzz_arbiter.npz")
    if not only and os.path.exists(path) and os.environ.get("ARBITER_REBUILD") != "1":
        # extend the fixture: seeds it already holds are kept as they are (ARBITER_REBUILD=1: all again)
        old = np.load(path)
        store.update({k: old[k] for k in old.files if k not in ("bmm_seeds", "vireo_seeds")})
        all_jobs = jobs
        jobs = [j for j in jobs if "s%d_n_exec" % j[1] not in store]
        print("fixture holds %d of %d cases; computing %d" % (len(all_jobs) - len(jobs), len(all_jobs), len(jobs)))
    else:
        all_jobs = jobs
    with mp.Pool(min(8, os.cpu_count() or 1)) as pool:
        for seed, out, note in pool.imap_unordered(_run, jobs):
            print(note, flush=True)
            store.update(out)
    store["bmm_seeds"] = np.array([s for k, s in all_jobs if k == "bmm"], dtype=np.int32)
    store["vireo_seeds"] = np.array([s for k, s in all_jobs if k == "vireo"], dtype=np.int32)
    if not only:
        np.savez_compressed(path, **store)
        print("saved fuzz_arbiter.npz: %d arrays" % len(store))


if __name__ == "__main__":
    main()
