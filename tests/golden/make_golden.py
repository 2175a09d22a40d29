"""Generate the golden fixtures in this directory from the REAL reference.

Run ONLY in the build container, where the read-only reference checkout lives
at /root/reference (it never travels to the GPU box):

    python tests/golden/make_golden.py

The fixtures are pure data: the reference's bundled inputs (data/cellSNP_mat,
data/mitoDNA -- its own demo/"test" data, examples/demo.sh) re-encoded as CSC
arrays, seeded synthetic inputs, and the arrays/scalars the reference returns
for them.  No reference source text is stored.
"""
import io
import os
import sys
import contextlib

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(HERE, "..", ".."))

import vireoSNP                                              # noqa: E402
from vireoSNP import Vireo, BinomMixtureVB, vireo_wrap       # noqa: E402
from vireoSNP.utils.vireo_base import get_binom_coeff        # noqa: E402
from vireoSNP.utils.vireo_doublet import predict_doublet     # noqa: E402
from scipy.io import mmread                                  # noqa: E402
from scipy.sparse import csc_matrix                          # noqa: E402

from oracle.vireo_oracle import synth_donor                  # noqa: E402


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrs)
    print("%-28s %8.1f KB" % (name, os.path.getsize(path) / 1024))


def pack(AD, DP):
    AD = csc_matrix(AD)
    DP = csc_matrix(DP)
    AD.sort_indices()
    DP.sort_indices()
    return dict(shape=np.array(DP.shape, np.int64),
                AD_indptr=AD.indptr.astype(np.int64), AD_indices=AD.indices.astype(np.int32),
                AD_data=AD.data.astype(np.int64),
                DP_indptr=DP.indptr.astype(np.int64), DP_indices=DP.indices.astype(np.int32),
                DP_data=DP.data.astype(np.int64))


def state(m, pre=""):
    return {pre + "ID_prob": m.ID_prob, pre + "GT_prob": m.GT_prob,
            pre + "beta_mu": m.beta_mu, pre + "beta_sum": m.beta_sum}


def main():
    assert vireoSNP.__version__ == "0.5.9", vireoSNP.__version__
    AD = mmread(REF + "/data/cellSNP_mat/cellSNP.tag.AD.mtx").tocsc()
    DP = mmread(REF + "/data/cellSNP_mat/cellSNP.tag.DP.mtx").tocsc()
    N, M = AD.shape
    save("c1_data", **pack(AD, DP))

    mAD = mmread(REF + "/data/mitoDNA/cellSNP.tag.AD.mtx").tocsc()
    mDP = mmread(REF + "/data/mitoDNA/cellSNP.tag.DP.mtx").tocsc()
    save("mito_data", **pack(mAD, mDP))

    # ---- binomial-coefficient constant -------------------------------------
    save("binom_const",
         c1=np.sum(get_binom_coeff(AD, DP)), c1_terms=np.asarray(get_binom_coeff(AD, DP)).ravel(),
         mito=np.sum(get_binom_coeff(mAD, mDP)),
         mito_terms=np.asarray(get_binom_coeff(mAD, mDP)).ravel())

    # ---- one-step fixtures: every kernel in isolation ----------------------
    np.random.seed(2)
    m = Vireo(n_var=N, n_cell=M, n_donor=4)
    quiet(m.fit, AD, DP, max_iter=6, verbose=False)
    d = state(m, "s0_")
    m.update_theta_size(AD, DP)
    d.update(state(m, "s1_"))
    m.update_GT_prob(AD, DP)
    d.update(state(m, "s2_"))
    L = m.update_ID_prob(AD, DP)
    d.update(state(m, "s3_"))
    d["logLik_ID"] = L
    d["ELBO"] = np.float64(m.get_ELBO(L))
    d["ELBO_recompute"] = np.float64(m.get_ELBO(None, AD, DP))
    save("c1_onestep", **d)

    # ---- full traces (fit, then warm-restart fit) --------------------------
    np.random.seed(2)
    m = Vireo(n_var=N, n_cell=M, n_donor=4)
    init = state(m, "init_")
    quiet(m.fit, AD, DP, min_iter=5, max_iter=20, delay_fit_theta=3, verbose=False)
    n_first = len(m.ELBO_)
    mid = state(m, "mid_")
    quiet(m.fit, AD, DP, min_iter=5, verbose=False)
    save("c1_trace_seed2", ELBO_=m.ELBO_, n_first=np.int64(n_first),
         **init, **mid, **state(m, "end_"))

    # ---- flags --------------------------------------------------------------
    for tag, kw in [("ase", dict(ASE_mode=True)), ("fixsum", dict(fix_beta_sum=True)),
                    ("notheta", dict(learn_theta=False))]:
        np.random.seed(2)
        m = Vireo(n_var=N, n_cell=M, n_donor=4, **kw)
        init = state(m, "init_")
        quiet(m.fit, AD, DP, max_iter=12, verbose=False)
        save("c1_flag_" + tag, ELBO_=m.ELBO_, **init, **state(m, "end_"))

    # learn_GT=False with an informative (fixed) genotype prior: take the GT of
    # a converged fit, blur it, and use it both as GT_prob_init and GT_prior,
    # the way vireo_wrap does for a donor VCF (vireo_wrap.py:66-71).
    np.random.seed(3)
    m0 = Vireo(n_var=N, n_cell=M, n_donor=4)
    quiet(m0.fit, AD, DP, verbose=False)
    GTp = 0.9 * m0.GT_prob + 0.1 / 3
    np.random.seed(2)
    m = Vireo(n_var=N, n_cell=M, n_donor=4, learn_GT=False, GT_prob_init=GTp.copy())
    m.set_prior(GT_prior=GTp.copy())
    init = state(m, "init_")
    quiet(m.fit, AD, DP, max_iter=12, verbose=False)
    save("c1_flag_fixedGT", ELBO_=m.ELBO_, GT_prior_in=GTp, GT_prior=m.GT_prior,
         **init, **state(m, "end_"))
    # the same prior but learn_GT=True (non-uniform GT_prior in the GT step)
    np.random.seed(2)
    m = Vireo(n_var=N, n_cell=M, n_donor=4, learn_GT=True, GT_prob_init=GTp.copy())
    m.set_prior(GT_prior=GTp.copy())
    init = state(m, "init_")
    quiet(m.fit, AD, DP, max_iter=12, verbose=False)
    save("c1_flag_priorGT", ELBO_=m.ELBO_, GT_prior_in=GTp, **init, **state(m, "end_"))

    # ---- vireo_wrap ---------------------------------------------------------
    def wrap_case(name, **kw):
        rv = quiet(vireo_wrap, AD, DP, nproc=1, **kw)
        save(name, ID_prob=rv["ID_prob"], GT_prob=rv["GT_prob"],
             doublet_prob=rv["doublet_prob"], doublet_LLR=rv["doublet_LLR"],
             theta_shapes=rv["theta_shapes"], theta_mean=rv["theta_mean"],
             theta_sum=rv["theta_sum"], LB_list=rv["LB_list"],
             LB_doublet=np.float64(rv["LB_doublet"]))
    wrap_case("c1_wrap_seed2_init1", n_donor=4, n_init=1, random_seed=2)
    wrap_case("c1_wrap_seed2_init4", n_donor=4, n_init=4, random_seed=2)
    wrap_case("c1_wrap_seed1_init50", n_donor=4, n_init=50, random_seed=1,
              learn_GT=True, n_extra_donor=0, check_doublet=True)
    wrap_case("c1_wrap_seed2_nodoublet", n_donor=3, n_init=2, random_seed=2,
              check_doublet=False)
    # extra-donor search (donor_select, vireo_wrap.py:95-105) and ASE mode through the wrap
    wrap_case("c1_wrap_extra1_dist", n_donor=4, n_init=3, random_seed=2, n_extra_donor=1)
    wrap_case("c1_wrap_extra2_size", n_donor=3, n_init=2, random_seed=5, n_extra_donor=2,
              extra_donor_mode="size")
    wrap_case("c1_wrap_ase", n_donor=4, n_init=2, random_seed=2, ASE_mode=True)

    # ---- BinomMixtureVB on mitoDNA (notebook known answer) ------------------
    b = BinomMixtureVB(n_var=mAD.shape[0], n_cell=mAD.shape[1], n_donor=3)
    quiet(b.fit, mAD, mDP, min_iter=30, n_init=50, random_seed=1, verbose=False)
    assert b.ELBO_iters[-1] == -190779.74335041404, b.ELBO_iters[-1]
    save("mito_bmm_k3_seed1", ELBO_iters=b.ELBO_iters, ELBO_inits=b.ELBO_inits,
         ID_prob=b.ID_prob, beta_mu=b.beta_mu, beta_sum=b.beta_sum)
    # single init, short: a step-by-step trace
    np.random.seed(5)
    b = BinomMixtureVB(n_var=mAD.shape[0], n_cell=mAD.shape[1], n_donor=4)
    ID0 = b.ID_prob.copy()
    b._fit_BV(mAD, mDP, max_iter=15, min_iter=5, verbose=False)
    save("mito_bmm_k4_trace", ID_prob_init=ID0, ELBO_iters=b.ELBO_iters,
         ID_prob=b.ID_prob, beta_mu=b.beta_mu, beta_sum=b.beta_sum)

    # ---- synthetic: generator parity + other donor counts -------------------
    for tag, (n, mm, k, dens) in {"k3": (300, 200, 3, 0.05), "k16": (1500, 800, 16, 0.05),
                                  "k5": (800, 500, 5, 0.04)}.items():
        sAD, sDP = synth_donor(n, mm, k, dens, seed=0)
        np.random.seed(1)
        m = Vireo(n_var=n, n_cell=mm, n_donor=k)
        init = state(m, "init_")
        quiet(m.fit, sAD, sDP, min_iter=5, max_iter=20, delay_fit_theta=3, verbose=False)
        mid = state(m, "end_")
        mm_ = m
        dbl, sing, llr = quiet(predict_doublet, mm_, sAD, sDP)
        save("synth_" + tag, **pack(sAD, sDP), ELBO_=m.ELBO_, **init, **mid,
             doublet_prob=dbl, singlet_prob=sing, doublet_LLR=llr,
             GT_prob_after_doublet=mm_.GT_prob)


if __name__ == "__main__":
    main()
