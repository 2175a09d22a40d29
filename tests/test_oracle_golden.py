"""Pin the CPU oracle against the golden vectors captured from the real
reference (tests/golden/make_golden.py).  Bit-exact: the oracle calls the same
SciPy routines in the same order.  CPU only."""
import numpy as np
import pytest

from oracle import vireo_oracle as O
from tests import gold


def eq(a, b):
    np.testing.assert_array_equal(np.asarray(a), np.asarray(b))


def test_binom_const_c1_and_mito():
    g = gold.load("binom_const")
    AD, DP = gold.c1()
    eq(np.asarray(O.binom_const_terms(AD, DP)).ravel(), g["c1_terms"])
    assert O.binom_const(AD, DP) == g["c1"]
    assert O.binom_const(AD, DP).dtype == np.float32
    mAD, mDP = gold.mito()
    eq(np.asarray(O.binom_const_terms(mAD, mDP)).ravel(), g["mito_terms"])
    assert O.binom_const(mAD, mDP) == g["mito"]
    assert (g["mito_terms"] == 700).any()          # the clamp IS exercised


def _state_from(g, pre, **kw):
    st = O.vireo_new(g[pre + "ID_prob"].shape[0], g[pre + "GT_prob"].shape[0],
                     g[pre + "ID_prob"].shape[1], ID_prob_init=g[pre + "ID_prob"],
                     GT_prob_init=g[pre + "GT_prob"], beta_mu_init=g[pre + "beta_mu"],
                     beta_sum_init=g[pre + "beta_sum"], **kw)
    # the ctor re-normalises; restore the exact arrays
    st.ID_prob = g[pre + "ID_prob"].copy()
    st.GT_prob = g[pre + "GT_prob"].copy()
    return st


def _check_state(st, g, pre):
    for k in ("ID_prob", "GT_prob", "beta_mu", "beta_sum"):
        eq(getattr(st, k), g[pre + k])


def test_onestep_each_kernel():
    g = gold.load("c1_onestep")
    AD, DP = gold.c1()
    st = _state_from(g, "s0_")
    O.vireo_theta_step(st, AD, DP)
    _check_state(st, g, "s1_")
    O.vireo_gt_step(st, AD, DP)
    _check_state(st, g, "s2_")
    L = O.vireo_id_step(st, AD, DP)
    _check_state(st, g, "s3_")
    eq(L, g["logLik_ID"])
    assert O.vireo_elbo(st, L) == g["ELBO"]
    assert O.vireo_elbo(st, None, AD, DP) == g["ELBO_recompute"]
    # anchor from SURVEY.md 8(c)
    assert abs(g["ELBO"] - (-50752.464183086)) < 1e-6


def test_trace_and_warm_restart():
    g = gold.load("c1_trace_seed2")
    AD, DP = gold.c1()
    np.random.seed(2)
    st = O.vireo_new(AD.shape[1], AD.shape[0], 4)
    _check_state(st, g, "init_")
    O.vireo_fit(st, AD, DP, min_iter=5, max_iter=20, delay_fit_theta=3)
    assert len(st.ELBO_) == int(g["n_first"]) == 19
    _check_state(st, g, "mid_")
    O.vireo_fit(st, AD, DP, min_iter=5)
    eq(st.ELBO_, g["ELBO_"])
    assert len(st.ELBO_) == 79
    _check_state(st, g, "end_")


@pytest.mark.parametrize("tag,kw", [("ase", dict(ASE_mode=True)),
                                    ("fixsum", dict(fix_beta_sum=True)),
                                    ("notheta", dict(learn_theta=False))])
def test_flags(tag, kw):
    g = gold.load("c1_flag_" + tag)
    AD, DP = gold.c1()
    np.random.seed(2)
    st = O.vireo_new(AD.shape[1], AD.shape[0], 4, **kw)
    _check_state(st, g, "init_")
    O.vireo_fit(st, AD, DP, max_iter=12)
    eq(st.ELBO_, g["ELBO_"])
    _check_state(st, g, "end_")


@pytest.mark.parametrize("tag,learn", [("fixedGT", False), ("priorGT", True)])
def test_gt_prior(tag, learn):
    g = gold.load("c1_flag_" + tag)
    AD, DP = gold.c1()
    np.random.seed(2)
    st = O.vireo_new(AD.shape[1], AD.shape[0], 4, learn_GT=learn,
                     GT_prob_init=g["GT_prior_in"].copy())
    O.vireo_prior(st, GT_prior=g["GT_prior_in"].copy())
    _check_state(st, g, "init_")
    O.vireo_fit(st, AD, DP, max_iter=12)
    eq(st.ELBO_, g["ELBO_"])
    _check_state(st, g, "end_")


@pytest.mark.parametrize("name,kw", [
    ("c1_wrap_seed2_init1", dict(n_donor=4, n_init=1, random_seed=2)),
    ("c1_wrap_seed2_init4", dict(n_donor=4, n_init=4, random_seed=2)),
    ("c1_wrap_seed2_nodoublet", dict(n_donor=3, n_init=2, random_seed=2,
                                     check_doublet=False)),
])
def test_wrap(name, kw):
    g = gold.load(name)
    AD, DP = gold.c1()
    rv = O.vireo_wrap_oracle(AD, DP, **kw)
    for k in ("ID_prob", "GT_prob", "doublet_prob", "doublet_LLR", "theta_shapes",
              "theta_mean", "theta_sum", "LB_list"):
        eq(rv[k], g[k])
    assert rv["LB_doublet"] == g["LB_doublet"]


def test_wrap_anchor_values():
    g = gold.load("c1_wrap_seed2_init1")
    assert abs(g["LB_list"][0] - (-47838.73209372)) < 1e-6
    assert abs(g["LB_doublet"] - (-46212.44402477946)) < 1e-9
    g = gold.load("c1_wrap_seed1_init50")          # notebook known answer
    ids = np.argmax(g["ID_prob"], axis=1)
    assert len(g["LB_list"]) == 50
    assert abs(g["LB_doublet"] - (-41672.93668438444)) < 1e-9


def test_bmm_mito_known_answer():
    g = gold.load("mito_bmm_k3_seed1")
    AD, DP = gold.mito()
    st = O.bmm_new(AD.shape[1], AD.shape[0], 3)
    O.bmm_fit(st, AD, DP, min_iter=30, n_init=50, random_seed=1)
    assert st.ELBO_iters[-1] == -190779.74335041404    # vireoSNP_clones.ipynb
    eq(st.ELBO_iters, g["ELBO_iters"])
    eq(st.ELBO_inits, g["ELBO_inits"])
    for k in ("ID_prob", "beta_mu", "beta_sum"):
        eq(getattr(st, k), g[k])


def test_bmm_trace():
    g = gold.load("mito_bmm_k4_trace")
    AD, DP = gold.mito()
    np.random.seed(5)
    st = O.bmm_new(AD.shape[1], AD.shape[0], 4)
    eq(st.ID_prob, g["ID_prob_init"])
    O.bmm_fit_vb(st, AD, DP, max_iter=15, min_iter=5)
    eq(st.ELBO_iters, g["ELBO_iters"])
    for k in ("ID_prob", "beta_mu", "beta_sum"):
        eq(getattr(st, k), g[k])


@pytest.mark.parametrize("tag,shape", [("k3", (300, 200, 3, 0.05)),
                                       ("k16", (1500, 800, 16, 0.05)),
                                       ("k5", (800, 500, 5, 0.04))])
def test_synthetic_generator_and_fit(tag, shape):
    g = gold.load("synth_" + tag)
    n, m, k, dens = shape
    AD, DP = O.synth_donor(n, m, k, dens, seed=0)
    gAD, gDP = gold.unpack(g)
    assert (AD != gAD).nnz == 0 and (DP != gDP).nnz == 0
    np.random.seed(1)
    st = O.vireo_new(m, n, k)
    _check_state(st, g, "init_")
    O.vireo_fit(st, AD, DP, min_iter=5, max_iter=20, delay_fit_theta=3)
    eq(st.ELBO_, g["ELBO_"])
    _check_state(st, g, "end_")
    dbl, sing, llr = O.vireo_doublet(st, AD, DP)
    eq(dbl, g["doublet_prob"])
    eq(sing, g["singlet_prob"])
    eq(llr, g["doublet_LLR"])
    eq(st.GT_prob, g["GT_prob_after_doublet"])


def test_fuzz_arbiter_fixture_is_the_oracles_problem():
    """tests/golden/fuzz_arbiter.npz (80-bit end states of the sweep's deviation cases,
    make_bmm_arbiter.py) against the oracle on two of its smaller cases: same problem, same
    iteration count, same assignments -- and the oracle's small posteriors ARE 1e-5 ... 1e-4
    relative away from the exact ones (the deviation class the GPU tests pin)."""
    from tests.test_gpu_fuzz import draw_case
    g = gold.load("fuzz_arbiter")
    assert len(g["bmm_seeds"]) == 45 and list(g["vireo_seeds"]) == [537, 1648, 2260]
    for seed, lo, hi in ((343, 5e-6, 5e-5), (695, 4e-5, 2e-4), (2547, 3e-5, 1.3e-4)):     # (2547: one of round 6's fifteen)
        AD, DP, K, _ = draw_case(seed)
        N, M = AD.shape
        np.random.seed(seed)
        init = np.random.rand(M, max(K, 2))
        ref = O.bmm_new(M, N, max(K, 2), ID_prob_init=init.copy())
        O.bmm_fit_vb(ref, AD, DP, min_iter=2, max_iter=4)
        exact = g["s%d_ID_prob" % seed]
        assert exact.shape == ref.ID_prob.shape and len(ref.ELBO_iters) + 1 == int(g["s%d_n_exec" % seed])
        assert np.array_equal(exact.argmax(1), ref.ID_prob.argmax(1))
        np.testing.assert_allclose(exact.sum(1), 1.0, rtol=0, atol=1e-12)
        m = exact > 1e-290
        dev = np.max(np.abs(ref.ID_prob[m] - exact[m]) / exact[m])
        assert lo < dev < hi, (seed, dev)
