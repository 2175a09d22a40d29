"""Doublet prediction with a fitted Vireo model, drop-in for
vireoSNP/utils/vireo_doublet.py:11-136 (``predict_doublet`` and its two table builders).

The K + K(K-1)/2 column cell log-likelihood -- 3*6 transposed sparse products in the
reference (:53-62) -- is one cell pass on the GPU (``vrx_problem_cell_loglik``); the
genotype/theta tables of the donor pairs are small host-side combinatorics.
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
    GT_both = add_doublet_GT(vobj.GT_prob)
    beta_mu_both, beta_sum_both = add_doublet_theta(vobj.beta_mu, vobj.beta_sum)
    n_pair = GT_both.shape[1] - vobj.GT_prob.shape[1]
    if doublet_rate_prior is None:
        doublet_rate_prior = min(0.5, counts.n_cell / 100000)
    ID_prior_both = np.append(
        vobj.ID_prior * (1 - doublet_rate_prior),
        np.ones((vobj.n_cell, n_pair)) / n_pair * doublet_rate_prior, axis=1)

    # T' digamma values per theta row: O(T') host work (vireo_doublet.py:55-57)
    psi1 = f64(digamma(beta_sum_both * beta_mu_both))
    psi2 = f64(digamma(beta_sum_both * (1 - beta_mu_both)))
    psis = f64(digamma(beta_sum_both))
    C_, G_ = GT_both.shape[1], GT_both.shape[2]
    logLik_ID = np.empty((counts.n_cell, C_))
    ID_prob_both = np.empty((counts.n_cell, C_))
    prior = f64(ID_prior_both)
    GT_both = f64(GT_both)
    _lib.check(_lib.lib().vrx_problem_cell_loglik(
        counts.handle, C_, G_, dptr(GT_both), dptr(psi1), dptr(psi2), dptr(psis),
        psi1.shape[0], dptr(prior), prior.shape[0], dptr(logLik_ID), dptr(ID_prob_both)))

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
