"""Doublet prediction with a fitted Vireo model, drop-in for
vireoSNP/utils/vireo_doublet.py:11-136 (``predict_doublet`` and its two table builders).

The K + K(K-1)/2 column cell log-likelihood -- 3*6 transposed sparse products in the
reference (:53-62) -- is one cell pass on the GPU (``vrx_problem_doublet``); the genotype
table of the donor pairs (653 MB at N=100k, K=16 in the reference) is formed on the fly in
the kernel, and the 6 pair thetas are host-side arithmetic on 3 numbers.  ``add_doublet_GT``
is kept as a public helper (and for n_GT > 3).
"""
import itertools

import numpy as np
from scipy.special import digamma

from . import _lib
from ._lib import dptr, f64
from .counts import device_counts
from .vireo_base import normalize


def add_doublet_theta(beta_mu, beta_sum):
    """theta of the mixed genotypes 0&1, 0&2, 1&2: mean of the means, geometric mean of
    the concentrations (vireo_doublet.py:85-102)."""
    pairs = np.array(list(itertools.combinations(range(beta_mu.shape[1]), 2)))
    a, b = pairs[:, 0], pairs[:, 1]
    mu_db = (beta_mu[:, a] + beta_mu[:, b]) / 2.0
    sum_db = np.sqrt(beta_sum[:, a] * beta_sum[:, b])
    return np.append(beta_mu, mu_db, axis=-1), np.append(beta_sum, sum_db, axis=-1)


def add_doublet_GT(GT_prob):
    """Genotype table of all donor pairs over T + T(T-1)/2 classes, appended to the singlet
    table padded with zero mixed classes (vireo_doublet.py:105-136)."""
    n_gt = GT_prob.shape[2]
    gt_pairs = np.array(list(itertools.combinations(range(n_gt), 2)))
    dn_pairs = np.array(list(itertools.combinations(range(GT_prob.shape[1]), 2)))
    g1, g2 = gt_pairs[:, 0], gt_pairs[:, 1]
    P = GT_prob[:, dn_pairs[:, 0], :]
    Q = GT_prob[:, dn_pairs[:, 1], :]
    both = np.zeros((GT_prob.shape[0], dn_pairs.shape[0], n_gt + gt_pairs.shape[0]))
    both[:, :, :n_gt] = P * Q
    both[:, :, n_gt:] = P[:, :, g1] * Q[:, :, g2] + P[:, :, g2] * Q[:, :, g1]
    both = normalize(both, axis=2)
    single = np.append(
        GT_prob, np.zeros((GT_prob.shape[0], GT_prob.shape[1], gt_pairs.shape[0])), axis=2)
    return np.append(single, both, axis=1)


def predict_doublet(vobj, AD, DP, update_GT=True, update_ID=True,
                    doublet_rate_prior=None):
    """-> (doublet_prob (n_cell, K(K-1)/2), singlet ID_prob (n_cell, K), logLik_ratio)
    exactly as vireo_doublet.py:11-82, including its side effects on ``vobj``
    (ID_prob <- un-renormalised singlet block, then update_GT_prob)."""
    counts = device_counts(AD, DP)
    K, T = vobj.GT_prob.shape[1], vobj.GT_prob.shape[2]
    n_pair = K * (K - 1) // 2
    C_ = K + n_pair
    beta_mu_both, beta_sum_both = add_doublet_theta(vobj.beta_mu, vobj.beta_sum)
    if doublet_rate_prior is None:
        doublet_rate_prior = min(0.5, counts.n_cell / 100000)
    ID_prior_both = np.append(
        vobj.ID_prior * (1 - doublet_rate_prior),
        np.ones((vobj.n_cell, n_pair)) / n_pair * doublet_rate_prior, axis=1)

    # T + T(T-1)/2 digamma values per theta row: O(T') host work (vireo_doublet.py:55-57)
    psi1 = f64(digamma(beta_sum_both * beta_mu_both))
    psi2 = f64(digamma(beta_sum_both * (1 - beta_mu_both)))
    psis = f64(digamma(beta_sum_both))
    logLik_ID = np.empty((counts.n_cell, C_))
    ID_prob_both = np.empty((counts.n_cell, C_))
    prior = f64(ID_prior_both)
    if T <= 3 and K >= 2:
        # the pair genotype table (add_doublet_GT) is formed inside the kernel, never in memory
        GT = f64(vobj.GT_prob)
        _lib.check(_lib.lib().vrx_problem_doublet(
            counts.handle, K, T, dptr(GT), dptr(psi1), dptr(psi2), dptr(psis), psi1.shape[0],
            dptr(prior), prior.shape[0], dptr(logLik_ID), dptr(ID_prob_both)))
    else:   # unusual n_GT: explicit table, same cell pass
        GT_both = f64(add_doublet_GT(vobj.GT_prob))
        _lib.check(_lib.lib().vrx_problem_cell_loglik(
            counts.handle, C_, GT_both.shape[2], dptr(GT_both), dptr(psi1), dptr(psi2),
            dptr(psis), psi1.shape[0], dptr(prior), prior.shape[0], dptr(logLik_ID),
            dptr(ID_prob_both)))

    logLik_ratio = (logLik_ID[:, vobj.n_donor:].max(1) -
                    logLik_ID[:, :vobj.n_donor].max(1))
    if update_ID:
        vobj.ID_prob = ID_prob_both[:, :vobj.n_donor]
    if update_GT:
        if update_ID:
            vobj.update_GT_prob(counts, None)
        else:
            print("For update_GT, please turn on update_ID.")
    return (ID_prob_both[:, vobj.n_donor:], ID_prob_both[:, :vobj.n_donor], logLik_ratio)
