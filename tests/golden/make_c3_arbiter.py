"""Extended-precision arbiter for the config-3 timing protocol.

The protocol -- ``np.random.seed(1); Vireo(n_var=100000, n_cell=50000, n_donor=16)`` on the
SURVEY.md 8(d) synthetic matrix, ``_fit_VB(min_iter=5, max_iter=20, delay_fit_theta=3)``
(vireoSNP/utils/vireo_model.py:251-276) -- leaves a nearly symmetric state around iterations 6-9
from which the clusters break out; rounding differences between two float64 implementations
are amplified ~1000x per iteration there, so the GPU trace and the NumPy/SciPy oracle's differ
by ~1e-5 relative at iteration 8 although both are "right".  This script settles which one is
closer to the mathematics: the same iteration in 80-bit extended precision (np.longdouble
arrays, eps 1.1e-19; scipy's sparse products are instantiated for long double; the nine
digamma / three betaln values per iteration come from mpmath at 40 digits).  Its own rounding
noise is ~1e-3 of float64's, i.e. ~1e-8 relative on the ELBO at the worst iteration.

Run in the build container (about 15 minutes, ~12 GB):  python tests/golden/make_c3_arbiter.py
Output: tests/golden/c3_protocol_longdouble.npz -- the ELBO trace (without the binomial
constant, like ``_fit_VB``'s), the iteration count, the final assignments, and (round 4) the
END-STATE POSTERIORS the protocol leaves (vireo_model.py:198-199, :218-219): the whole final
``ID_prob`` and every 97th variant's ``GT_prob`` rows, rounded to float64.  The GPU test
(tests/test_gpu_fullsize.py) and bench.py compare |GPU - exact| with |oracle - exact|.

The update equations are those of the reference (cited inline); only the arithmetic differs, and
the regrouping  AD'(GT psi1) + BD'(GT psi2) - DP'(GT psis) = AD' Wa + BD' Wb  with
Wa = sum_t GT_t (psi1_t - psis_t), Wb = sum_t GT_t (psi2_t - psis_t)  (2 instead of 9 products).
"""
import os
import sys
import time

import mpmath
import numpy as np
from scipy.sparse import csc_matrix

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from vireo_amd import synth                     # noqa: E402  (the generator only: host NumPy)

LD = np.longdouble
GT_STRIDE = 97          # GT_prob rows kept: variants 0, 97, 194, ... (1031 x K x T doubles at c3)
mpmath.mp.dps = 40


def ld(x):
    return LD(mpmath.nstr(x, 30))


def psi(v):
    return np.array([ld(mpmath.digamma(mpmath.mpf(str(x)))) for x in v.ravel()], dtype=LD).reshape(v.shape)


def betaln(a, b):
    return ld(mpmath.log(mpmath.beta(mpmath.mpf(str(a)), mpmath.mpf(str(b)))))


def rel_entr_sum(p, q_const):
    """sum p (log p - log q) with 0 log 0 = 0 (scipy.stats.entropy, vireo_model.py:237-238)"""
    out = LD(0)
    m = p > 0
    out = np.sum(p[m] * (np.log(p[m]) - q_const))
    return out


def main(config="c3", out_name="c3_protocol_longdouble"):
    N, M, K, d = synth.CONFIGS[config]
    T = 3
    t0 = time.time()
    w = synth.donor_workload(N, M, K, d, seed=0)
    shape = tuple(int(x) for x in w["shape"])
    AD = csc_matrix((w["ad"].astype(LD), w["rowidx"], w["colptr"]), shape=shape)
    BD = csc_matrix(((w["dp"] - w["ad"]).astype(LD), w["rowidx"], w["colptr"]), shape=shape)
    print("matrix %s nnz %d (%.0f s)" % (shape, AD.nnz, time.time() - t0), flush=True)
    # the reference's constructor draws (vireo_model.py:98,103) and default priors (:107-137)
    np.random.seed(1)
    ID = np.random.rand(M, K).astype(LD)
    ID /= ID.sum(1, keepdims=True)
    GT = np.random.rand(N, K, T).astype(LD)
    GT /= GT.sum(2, keepdims=True)
    mu0 = np.linspace(0.01, 0.99, T)
    p1 = (mu0 * 50.0).astype(LD)            # theta_s1_prior
    p2 = ((1 - mu0) * 50.0).astype(LD)      # theta_s2_prior
    mu, sm = mu0.astype(LD).copy(), np.full(T, 50.0, dtype=LD)
    log_id_prior, log_gt_prior = -np.log(LD(K)), -np.log(LD(T))
    max_iter, min_iter, eps, delay = 20, 5, 1e-2, 3
    trace = np.zeros(max_iter, dtype=LD)
    it = 0
    for it in range(max_iter):
        t1 = time.time()
        A = AD @ ID                          # vireo_model.py:169-170, 207-208
        B = BD @ ID
        if it >= delay:                      # update_theta_size, vireo_model.py:165-185
            s1 = p1 + np.array([np.sum(A * GT[:, :, g]) for g in range(T)], dtype=LD)
            s2 = p2 + np.array([np.sum(B * GT[:, :, g]) for g in range(T)], dtype=LD)
            mu, sm = s1 / (s1 + s2), s1 + s2
        s1, s2 = mu * sm, (1 - mu) * sm
        d1, d2, ds = psi(s1), psi(s2), psi(s1 + s2)
        # update_GT_prob, vireo_model.py:204-219
        L = np.empty((N, K, T), dtype=LD)
        for g in range(T):
            L[:, :, g] = A * d1[g] + B * d2[g] - (A + B) * ds[g]
        L += log_gt_prior
        L -= L.max(2, keepdims=True)
        GT = np.exp(L)
        GT /= GT.sum(2, keepdims=True)
        del L
        # update_ID_prob, vireo_model.py:187-201
        Wa = sum(GT[:, :, g] * (d1[g] - ds[g]) for g in range(T))
        Wb = sum(GT[:, :, g] * (d2[g] - ds[g]) for g in range(T))
        LID = AD.T @ Wa + BD.T @ Wb
        Z = LID + log_id_prior
        Z -= Z.max(1, keepdims=True)
        ID = np.exp(Z)
        ID /= ID.sum(1, keepdims=True)
        # get_ELBO, vireo_model.py:222-248
        LB_p = np.sum(LID * ID)
        KL_ID = rel_entr_sum(ID, log_id_prior)
        KL_GT = rel_entr_sum(GT, log_gt_prior)
        KL_th = LD(0)
        for g in range(T):                   # beta_entropy, vireo_base.py:77-127
            cq = betaln(p1[g], p2[g]) - (p1[g] - 1) * d1[g] - (p2[g] - 1) * d2[g] + (p1[g] + p2[g] - 2) * ds[g]
            cp = betaln(s1[g], s2[g]) - (s1[g] - 1) * d1[g] - (s2[g] - 1) * d2[g] + (s1[g] + s2[g] - 2) * ds[g]
            KL_th += cq - cp
        trace[it] = LB_p - KL_ID - KL_GT - KL_th
        print("it %2d ELBO %s (%.0f s)" % (it, np.format_float_positional(trace[it], precision=12), time.time() - t1),
              flush=True)
        if it > min_iter:                    # vireo_model.py:266-274
            if trace[it] < trace[it - 1] - 1e-6:
                print("lower bound decreases")
            elif it == max_iter - 1:
                pass
            elif trace[it] - trace[it - 1] < eps:
                break
    kept = trace[:it]
    hi = kept.astype(np.float64)
    lo = (kept - hi.astype(LD)).astype(np.float64)
    np.savez_compressed(os.path.join(HERE, out_name + ".npz"), elbo_hi=hi, elbo_lo=lo, n_iter=np.int64(it),
                        assign=ID.argmax(1).astype(np.int8), theta_mu=mu.astype(np.float64),
                        theta_sum=sm.astype(np.float64), nnz=np.int64(AD.nnz),
                        ID_prob=ID.astype(np.float64), GT_stride=np.int64(GT_STRIDE),
                        GT_prob_sample=GT[::GT_STRIDE].astype(np.float64))
    print("saved %s: %d kept iterations, %.0f s" % (out_name, it, time.time() - t0))


if __name__ == "__main__":
    main(*sys.argv[1:])
