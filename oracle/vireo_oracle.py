"""CPU ORACLE for the vireoSNP variational-EM hot path.  TEST INFRASTRUCTURE ONLY.

This file is a NumPy/SciPy restatement of the reference algorithm
(vireoSNP 0.5.9).  It exists so that the HIP path can be *checked*; it is never
the thing shipped or measured as the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it.  Nothing under ``vireo_amd/`` imports it.

Parity status: PINNED.  ``tests/golden/make_golden.py`` imports the real
reference from /root/reference in the build container, runs it on its bundled
data (data/cellSNP_mat, data/mitoDNA) and on seeded synthetic inputs, and
commits the inputs/outputs as fixtures under ``tests/golden/``;
``tests/test_oracle_golden.py`` requires this file to reproduce every one of
them bit-for-bit (same SciPy kernels, same operation order).

Every function cites the reference lines (relative to /root/reference) whose
arithmetic it restates.  The state is a plain ``SimpleNamespace`` and the steps
are free functions, so that a test can call one step at a time.

Third-party arithmetic on the path (not under /root/reference): scipy
(unpinned ``scipy>=1.4.0`` in setup.py:21; 1.15.3 in this image) --
``csr_matvecs``/``csc_matvecs`` (sequential accumulation in increasing
contracted index), ``scipy.special.{digamma,betaln,binom}``,
``scipy.stats.entropy``.  We call the very same routines here.
"""
from types import SimpleNamespace
import itertools

import numpy as np
from scipy.sparse import csc_matrix, issparse
from scipy.special import digamma, betaln, binom
from scipy.stats import entropy


# ----------------------------------------------------------------------------
# helpers  (vireoSNP/utils/vireo_base.py)
# ----------------------------------------------------------------------------
def unit_sum(X, axis=-1):
    """X / sum(X) along ``axis``  (vireo_base.py:44-55 ``normalize``)."""
    return X / np.sum(X, axis=axis, keepdims=True)


def shift_max(X, axis=-1):
    """X - max(X) along ``axis``  (vireo_base.py:62-74 ``loglik_amplify``)."""
    return X - np.max(X, axis=axis, keepdims=True)


def softmax_log(L, axis=-1):
    """exp(L - max) / sum  -- the composition used at vireo_model.py:198-199,
    :218-219 and bmm_model.py:153-154."""
    return unit_sum(np.exp(shift_max(L, axis=axis)), axis=axis)


def binom_const_terms(AD, DP, cap=700):
    """float32 vector of min(log C(DP,AD), cap) over entries with DP>0
    (vireo_base.py:7-22 ``get_binom_coeff``)."""
    sel = DP > 0
    a = AD[sel].astype(np.int64)
    d = DP[sel].astype(np.int64)
    with np.errstate(over="ignore", divide="ignore"):
        v = np.log(binom(d, a))
    v[v > cap] = cap
    return v.astype(np.float32)


def binom_const(AD, DP):
    """The scalar the reference adds to the ELBO trace (vireo_model.py:313,
    bmm_model.py:239): a float32 sum of the float32 terms."""
    return np.sum(binom_const_terms(AD, DP))


def beta_kl(post, prior):
    """KL( Beta(post) || Beta(prior) ) summed over all entries
    (vireo_base.py:77-127 ``beta_entropy`` with X_prior given).
    ``post``/``prior`` have shape (L, 2, T) (prior may broadcast on L)."""
    def cross(p, q):                       # -E_p[log q], vireo_base.py:96-105
        return (betaln(q[:, 0], q[:, 1])
                - (q[:, 0] - 1) * digamma(p[:, 0])
                - (q[:, 1] - 1) * digamma(p[:, 1])
                + (q.sum(axis=1) - 2) * digamma(p.sum(axis=1)))
    return np.sum(cross(post, prior) - cross(post, post))


def maybe_sparsify(AD, DP):
    """dense -> CSC when < 30 % non-zero (vireo_model.py:300-305,
    vireo_wrap.py:29-34, bmm_model.py:232-237)."""
    if type(DP) is np.ndarray and np.mean(DP > 0) < 0.3:
        AD, DP = csc_matrix(AD), csc_matrix(DP)
    return AD, DP


# ----------------------------------------------------------------------------
# Vireo model  (vireoSNP/utils/vireo_model.py)
# ----------------------------------------------------------------------------
def vireo_new(n_cell, n_var, n_donor, n_GT=3, learn_GT=True, learn_theta=True,
              ASE_mode=False, fix_beta_sum=False, beta_mu_init=None,
              beta_sum_init=None, ID_prob_init=None, GT_prob_init=None):
    """State with the reference's default initialisation and priors.
    RNG draw order (global legacy stream): rand(M,K) then rand(N,K,T)
    (vireo_model.py:78-104), priors as set_prior() with no arguments
    (vireo_model.py:107-137)."""
    st = SimpleNamespace(n_cell=n_cell, n_var=n_var, n_donor=n_donor, n_GT=n_GT,
                         learn_GT=learn_GT, learn_theta=learn_theta,
                         ASE_mode=ASE_mode, fix_beta_sum=fix_beta_sum,
                         ELBO_=np.zeros(0))
    L = n_var if ASE_mode else 1
    st.beta_mu = (beta_mu_init if beta_mu_init is not None else
                  np.ones((L, n_GT)) * np.linspace(0.01, 0.99, n_GT).reshape(1, -1))
    st.beta_sum = (beta_sum_init if beta_sum_init is not None else
                   np.ones((L, n_GT)) * 50)
    st.ID_prob = (unit_sum(ID_prob_init, axis=1) if ID_prob_init is not None
                  else unit_sum(np.random.rand(n_cell, n_donor)))
    st.GT_prob = (unit_sum(GT_prob_init) if GT_prob_init is not None
                  else unit_sum(np.random.rand(n_var, n_donor, n_GT)))
    vireo_prior(st)
    return st


def vireo_prior(st, GT_prior=None, ID_prior=None, beta_mu_prior=None,
                beta_sum_prior=None, min_GP=0.00001):
    """vireo_model.py:107-137.  NB: clips the caller's GT_prior IN PLACE like
    the reference does (:132-133)."""
    if beta_mu_prior is None:
        beta_mu_prior = np.expand_dims(
            np.linspace(0.01, 0.99, st.beta_mu.shape[1]), axis=0)
    if beta_sum_prior is None:
        beta_sum_prior = np.ones(beta_mu_prior.shape) * 50.0
    st.theta_s1_prior = beta_mu_prior * beta_sum_prior
    st.theta_s2_prior = (1 - beta_mu_prior) * beta_sum_prior
    if ID_prior is not None:
        st.ID_prior = ID_prior[None, :] if ID_prior.ndim == 1 else ID_prior
    else:
        st.ID_prior = unit_sum(np.ones(st.ID_prob.shape))
    if GT_prior is not None:
        if GT_prior.ndim == 2:
            GT_prior = GT_prior[None, :, :]
        GT_prior[GT_prior < min_GP] = min_GP
        GT_prior[GT_prior > 1 - min_GP] = 1 - min_GP
        st.GT_prior = unit_sum(GT_prior)
    else:
        st.GT_prior = unit_sum(np.ones(st.GT_prob.shape))


def _shape12(st):
    """Beta shape parameters (vireo_model.py:139-147)."""
    return st.beta_mu * st.beta_sum, (1 - st.beta_mu) * st.beta_sum


def _psi3(st):
    """psi(s1), psi(s2), psi(s1+s2) as (L,1,T) (vireo_model.py:149-162)."""
    s1, s2 = _shape12(st)
    return (digamma(s1)[:, None, :], digamma(s2)[:, None, :],
            digamma(s1 + s2)[:, None, :])


def vireo_theta_step(st, AD, DP):
    """vireo_model.py:165-185."""
    BD = DP - AD
    A = AD @ st.ID_prob
    B = BD @ st.ID_prob
    t1 = np.zeros(st.beta_mu.shape)
    t2 = np.zeros(st.beta_mu.shape)
    t1 += st.theta_s1_prior.copy()
    t2 += st.theta_s2_prior.copy()
    ax = 1 if st.ASE_mode else None
    for g in range(st.n_GT):
        t1[:, g:g + 1] += np.sum(A * st.GT_prob[:, :, g], axis=ax, keepdims=True)
        t2[:, g:g + 1] += np.sum(B * st.GT_prob[:, :, g], axis=ax, keepdims=True)
    st.beta_mu = t1 / (t1 + t2)
    if st.fix_beta_sum == False:      # noqa: E712 (mirrors the reference test)
        st.beta_sum = t1 + t2


def vireo_gt_step(st, AD, DP):
    """vireo_model.py:204-219."""
    A = AD @ st.ID_prob
    S = DP @ st.ID_prob
    B = S - A
    p1, p2, ps = _psi3(st)
    L = np.zeros(st.GT_prior.shape)
    for g in range(st.n_GT):
        L[:, :, g] = A * p1[:, :, g] + B * p2[:, :, g] - S * ps[:, :, g]
    st.GT_prob = softmax_log(L + np.log(st.GT_prior))


def _cell_loglik(GT, p1, p2, ps, AD, DP):
    """The 3*T transposed products of vireo_model.py:190-196 (also :227-234 and
    vireo_doublet.py:53-62)."""
    BD = DP - AD
    L = np.zeros((AD.shape[1], GT.shape[1]))
    for g in range(GT.shape[2]):
        a = AD.T @ (GT[:, :, g] * p1[:, :, g])
        b = BD.T @ (GT[:, :, g] * p2[:, :, g])
        s = DP.T @ (GT[:, :, g] * ps[:, :, g])
        L += (a + b - s)
    return L


def vireo_id_step(st, AD, DP):
    """vireo_model.py:187-201; returns logLik_ID."""
    L = _cell_loglik(st.GT_prob, *_psi3(st), AD, DP)
    st.ID_prob = softmax_log(L + np.log(st.ID_prior))
    return L


def vireo_elbo_parts(st, logLik_ID):
    """(LB_p, KL_ID, KL_GT, KL_theta) of vireo_model.py:236-248."""
    s1, s2 = _shape12(st)
    LB_p = np.sum(logLik_ID * st.ID_prob)
    KL_ID = np.sum(entropy(st.ID_prob, st.ID_prior, axis=-1))
    KL_GT = np.sum(entropy(st.GT_prob, st.GT_prior, axis=-1))
    KL_th = beta_kl(
        np.append(s1[:, None, :], s2[:, None, :], axis=1),
        np.append(st.theta_s1_prior[:, None, :], st.theta_s2_prior[:, None, :], axis=1))
    return LB_p, KL_ID, KL_GT, KL_th


def vireo_elbo(st, logLik_ID, AD=None, DP=None):
    """vireo_model.py:222-248."""
    if logLik_ID is None:
        logLik_ID = _cell_loglik(st.GT_prob, *_psi3(st), AD, DP)
    a, b, c, d = vireo_elbo_parts(st, logLik_ID)
    return a - b - c - d


def vireo_fit_vb(st, AD, DP, max_iter=200, min_iter=5, epsilon_conv=1e-2,
                 delay_fit_theta=0, verbose=False):
    """vireo_model.py:251-276.  Returns (ELBO[:it], it): the trace WITHOUT the
    last computed value, exactly like the reference."""
    trace = np.zeros(max_iter)
    it = 0
    for it in range(max_iter):
        if st.learn_theta and it >= delay_fit_theta:
            vireo_theta_step(st, AD, DP)
        if st.learn_GT:
            vireo_gt_step(st, AD, DP)
        L = vireo_id_step(st, AD, DP)
        trace[it] = vireo_elbo(st, L)
        if it > min_iter:
            if trace[it] < trace[it - 1] - 1e-6:
                if verbose:
                    print("Warning: Lower bound decreases!\n")
            elif it == max_iter - 1:
                if verbose:
                    print("Warning: VB did not converge!\n")
            elif trace[it] - trace[it - 1] < epsilon_conv:
                break
    return trace[:it], it


def vireo_fit(st, AD, DP, max_iter=200, min_iter=5, epsilon_conv=1e-2,
              delay_fit_theta=0, verbose=False):
    """vireo_model.py:278-315 (warm restart: continues from st, appends)."""
    AD, DP = maybe_sparsify(AD, DP)
    tr, it = vireo_fit_vb(st, AD, DP, max_iter, min_iter, epsilon_conv,
                          delay_fit_theta, verbose)
    tr = tr + binom_const(AD, DP)
    st.ELBO_ = np.append(st.ELBO_, tr)
    return it


# ----------------------------------------------------------------------------
# doublets  (vireoSNP/utils/vireo_doublet.py:11-136)
# ----------------------------------------------------------------------------
def doublet_theta(beta_mu, beta_sum):
    """vireo_doublet.py:85-102."""
    pr = np.array(list(itertools.combinations(range(beta_mu.shape[1]), 2)))
    mu2 = (beta_mu[:, pr[:, 0]] + beta_mu[:, pr[:, 1]]) / 2.0
    sm2 = np.sqrt(beta_sum[:, pr[:, 0]] * beta_sum[:, pr[:, 1]])
    return np.append(beta_mu, mu2, axis=-1), np.append(beta_sum, sm2, axis=-1)


def doublet_GT(GT):
    """vireo_doublet.py:105-136."""
    T = GT.shape[2]
    gp = np.array(list(itertools.combinations(range(T), 2)))
    sp = np.array(list(itertools.combinations(range(GT.shape[1]), 2)))
    g1, g2 = gp[:, 0], gp[:, 1]
    a, b = GT[:, sp[:, 0], :], GT[:, sp[:, 1], :]
    G2 = np.zeros((GT.shape[0], sp.shape[0], T + gp.shape[0]))
    G2[:, :, :T] = a * b
    G2[:, :, T:] = a[:, :, g1] * b[:, :, g2] + a[:, :, g2] * b[:, :, g1]
    G2 = unit_sum(G2, axis=2)
    G1 = np.append(GT, np.zeros((GT.shape[0], GT.shape[1], gp.shape[0])), axis=2)
    return np.append(G1, G2, axis=1)


def vireo_doublet(st, AD, DP, doublet_rate_prior=None):
    """vireo_doublet.py:11-82 with update_GT=update_ID=True.  Mutates st like
    the reference (ID_prob <- un-renormalised singlet block; GT step re-run)."""
    GT2 = doublet_GT(st.GT_prob)
    mu2, sm2 = doublet_theta(st.beta_mu, st.beta_sum)
    n_pair = GT2.shape[1] - st.GT_prob.shape[1]
    if doublet_rate_prior is None:
        doublet_rate_prior = min(0.5, AD.shape[1] / 100000)
    prior2 = np.append(st.ID_prior * (1 - doublet_rate_prior),
                       np.ones((st.n_cell, n_pair)) / n_pair * doublet_rate_prior,
                       axis=1)
    p1 = digamma(sm2 * mu2)[:, None, :]
    p2 = digamma(sm2 * (1 - mu2))[:, None, :]
    ps = digamma(sm2)[:, None, :]
    L = _cell_loglik(GT2, p1, p2, ps, AD, DP)
    llr = L[:, st.n_donor:].max(1) - L[:, :st.n_donor].max(1)
    P = softmax_log(L + np.log(prior2))
    st.ID_prob = P[:, :st.n_donor]
    vireo_gt_step(st, AD, DP)
    return P[:, st.n_donor:], P[:, :st.n_donor], llr


# ----------------------------------------------------------------------------
# restart driver  (vireoSNP/utils/vireo_wrap.py:22-183; main path only:
# n_extra_donor == 0, GT_prior None or with exactly n_donor donors)
# ----------------------------------------------------------------------------
def vireo_wrap_oracle(AD, DP, GT_prior=None, n_donor=None, learn_GT=True,
                      n_init=20, random_seed=None, check_doublet=True,
                      max_iter_init=20, delay_fit_theta=3, **kw):
    AD, DP = maybe_sparsify(AD, DP)
    if n_donor is None:
        n_donor = GT_prior.shape[1]
    if learn_GT is False and n_init > 1:
        n_init = 1                                     # vireo_wrap.py:48-50
    if random_seed is not None:
        np.random.seed(random_seed)                    # vireo_wrap.py:53-54
    prior_use = GT_prior.copy() if GT_prior is not None else None
    models = []
    for _ in range(n_init):                            # vireo_wrap.py:66-71
        m = vireo_new(AD.shape[1], AD.shape[0], n_donor, learn_GT=learn_GT,
                      GT_prob_init=prior_use, **kw)
        vireo_prior(m, GT_prior=prior_use)
        models.append(m)
    for m in models:                                   # vireo_wrap.py:84-87
        vireo_fit(m, AD, DP, min_iter=5, max_iter=max_iter_init,
                  delay_fit_theta=delay_fit_theta)
    elbo_all = np.array([m.ELBO_[-1] for m in models])  # vireo_wrap.py:90-94
    best = models[int(np.argmax(elbo_all))]
    vireo_fit(best, AD, DP, min_iter=5)
    if check_doublet:                                  # vireo_wrap.py:151-156
        dbl, ID_prob, llr = vireo_doublet(best, AD, DP)
    else:
        ID_prob = best.ID_prob
        dbl = np.zeros((AD.shape[1], int(n_donor * (n_donor - 1) / 2)))
        llr = np.zeros(AD.shape[1])
    return dict(ID_prob=ID_prob, GT_prob=best.GT_prob, doublet_LLR=llr,
                doublet_prob=dbl,
                theta_shapes=np.append(best.beta_mu * best.beta_sum,
                                       (1 - best.beta_mu) * best.beta_sum, axis=0),
                theta_mean=best.beta_mu, theta_sum=best.beta_sum,
                LB_list=elbo_all, LB_doublet=best.ELBO_[-1], model=best)


# ----------------------------------------------------------------------------
# Binomial mixture, clone mode  (vireoSNP/utils/bmm_model.py)
# ----------------------------------------------------------------------------
def bmm_new(n_cell, n_var, n_donor, fix_beta_sum=False, beta_mu_init=None,
            beta_sum_init=None, ID_prob_init=None):
    """bmm_model.py:24-115: priors Beta(1,1), uniform ID prior; init mu=.5,
    sum=30, ID_prob = normalised rand(M,K)."""
    st = SimpleNamespace(n_cell=n_cell, n_var=n_var, n_donor=n_donor,
                         fix_beta_sum=fix_beta_sum, beta_mu_init=beta_mu_init,
                         beta_sum_init=beta_sum_init, ID_prob_init=ID_prob_init)
    mu0 = np.ones((n_var, n_donor)) * 0.5
    sm0 = np.ones(mu0.shape) * 2.0
    st.theta_s1_prior = mu0 * sm0
    st.theta_s2_prior = (1 - mu0) * sm0
    st.ID_prior = unit_sum(np.ones((n_cell, n_donor)))
    bmm_init(st, beta_mu_init, beta_sum_init, ID_prob_init)
    return st


def bmm_init(st, beta_mu_init=None, beta_sum_init=None, ID_prob_init=None):
    """bmm_model.py:65-85."""
    st.beta_mu = (beta_mu_init if beta_mu_init is not None
                  else np.ones((st.n_var, st.n_donor)) * 0.5)
    st.beta_sum = (beta_sum_init if beta_sum_init is not None
                   else np.ones(st.beta_mu.shape) * 30)
    st.ID_prob = (unit_sum(ID_prob_init, axis=1) if ID_prob_init is not None
                  else unit_sum(np.random.rand(st.n_cell, st.n_donor)))
    st.ELBO_iters = np.array([])


def bmm_theta_step(st, AD, DP):
    """bmm_model.py:133-144."""
    BD = DP - AD
    t1 = AD @ st.ID_prob
    t2 = BD @ st.ID_prob
    t1 += st.theta_s1_prior
    t2 += st.theta_s2_prior
    st.beta_mu = t1 / (t1 + t2)
    if st.fix_beta_sum == False:      # noqa: E712
        st.beta_sum = t1 + t2


def bmm_cell_loglik(st, AD, DP):
    """bmm_model.py:118-130."""
    BD = DP - AD
    s1, s2 = _shape12(st)
    return AD.T @ digamma(s1) + BD.T @ digamma(s2) - DP.T @ digamma(s1 + s2)


def bmm_id_step(st, L):
    """bmm_model.py:147-154."""
    st.ID_prob = softmax_log(L + np.log(st.ID_prior))


def bmm_elbo(st, L):
    """bmm_model.py:157-175."""
    s1, s2 = _shape12(st)
    LB_p = np.sum(L * st.ID_prob)
    KL_ID = np.sum(entropy(st.ID_prob, st.ID_prior, axis=-1))
    KL_th = beta_kl(
        np.append(s1[:, None, :], s2[:, None, :], axis=1),
        np.append(st.theta_s1_prior[:, None, :], st.theta_s2_prior[:, None, :], axis=1))
    return LB_p - KL_ID - KL_th


def bmm_fit_vb(st, AD, DP, max_iter=200, min_iter=20, epsilon_conv=1e-2,
               verbose=False):
    """bmm_model.py:178-201 (appends ELBO[:it] to st.ELBO_iters)."""
    trace = np.zeros(max_iter)
    it = 0
    for it in range(max_iter):
        bmm_theta_step(st, AD, DP)
        L = bmm_cell_loglik(st, AD, DP)
        bmm_id_step(st, L)
        trace[it] = bmm_elbo(st, L)
        if it > min_iter:
            if trace[it] - trace[it - 1] < -1e-6:
                if verbose:
                    print("Warning: ELBO decreases %.8f to %.8f!\n"
                          % (trace[it - 1], trace[it]))
            elif it == max_iter - 1:
                if verbose:
                    print("Warning: VB did not converge!\n")
            elif trace[it] - trace[it - 1] < epsilon_conv:
                break
    st.ELBO_iters = np.append(st.ELBO_iters, trace[:it])
    return it


def bmm_fit(st, AD, DP, n_init=10, max_iter=200, max_iter_pre=100,
            random_seed=None, **kw):
    """bmm_model.py:204-263."""
    if random_seed is not None:
        np.random.seed(random_seed)
    AD, DP = maybe_sparsify(AD, DP)
    const = binom_const(AD, DP)
    st.ELBO_inits = []
    best = None
    for i in range(n_init):
        bmm_init(st, st.beta_mu_init, st.beta_sum_init, st.ID_prob_init)
        bmm_fit_vb(st, AD, DP, max_iter=max_iter_pre, **kw)
        st.ELBO_inits.append(st.ELBO_iters[-1])
        if i == 0 or st.ELBO_iters[-1] > np.max(st.ELBO_inits[:-1]):
            best = (st.ID_prob + 0, st.beta_mu + 0, st.beta_sum + 0,
                    st.ELBO_iters + 0)
    bmm_init(st, best[1], best[2], best[0])
    st.ELBO_iters = best[3]
    bmm_fit_vb(st, AD, DP, max_iter=max_iter, **kw)
    st.ELBO_iters = st.ELBO_iters + const
    st.ELBO_inits = np.array(st.ELBO_inits) + const


# ----------------------------------------------------------------------------
# synthetic inputs  (the BASELINE.json configs)
# ----------------------------------------------------------------------------
def synth_donor(N, M, K, density, seed=0):
    """Configs 2-4: draw order fixed by SURVEY.md 8(d).  Returns CSC AD, DP
    (int64) with duplicate (r,c) summed and AD's explicit zeros dropped."""
    rng = np.random.default_rng(seed)
    nnz_t = int(N * M * density)
    r = rng.integers(0, N, nnz_t)
    c = rng.integers(0, M, nnz_t)
    dp = 1 + rng.poisson(1.0, nnz_t)
    GT = rng.integers(0, 3, (N, K))
    z = rng.integers(0, K, M)
    theta = np.array([0.01, 0.5, 0.99])[GT[r, z[c]]]
    ad = rng.binomial(dp, theta)
    from scipy.sparse import coo_matrix
    DP = coo_matrix((dp, (r, c)), shape=(N, M)).tocsc()
    AD = coo_matrix((ad, (r, c)), shape=(N, M)).tocsc()
    AD.eliminate_zeros()
    DP.sum_duplicates()
    AD.sum_duplicates()
    return AD, DP


def synth_clone(N=200, M=200000, K=8, seed=0):
    """Config 5 (BinomMixtureVB clone mode), SURVEY.md 8(d)."""
    rng = np.random.default_rng(seed)
    mask = rng.random((N, M)) < 0.9
    dp = rng.poisson(50, (N, M)) * mask
    z = rng.integers(0, K, M)
    af = rng.beta(0.3, 3, (N, K))
    ad = rng.binomial(dp, af[:, z])
    return csc_matrix(ad), csc_matrix(dp)
