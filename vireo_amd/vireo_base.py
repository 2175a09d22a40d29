"""Host-side helpers with the reference's names (vireoSNP/utils/vireo_base.py).

Only what the host needs to prepare the initial state and priors lives here; the
per-iteration normalisation / amplification / Beta-KL arithmetic of the reference runs
inside the HIP kernels (vireo_amd/csrc/vrx_kernels.h).
"""
import numpy as np

from .counts import device_counts


def normalize(X, axis=-1):
    """X / X.sum(axis) (vireo_base.py:44-55).  Host NumPy: used on the freshly drawn
    random initial state, whose RNG order must match the reference's legacy stream."""
    return X / np.sum(X, axis=axis, keepdims=True)


def tensor_normalize(X, axis=1):
    """vireo_base.py:58-59."""
    return normalize(X, axis)


def loglik_amplify(X, axis=-1):
    """X - X.max(axis) (vireo_base.py:62-74)."""
    return X - np.max(X, axis=axis, keepdims=True)


def get_binom_coeff(AD, DP, max_val=700, is_log=True):
    """float32 log C(DP, AD) of every entry with DP > 0, clamped at max_val -- the per-entry
    helper the reference exports (vireo_base.py:7-22; ``is_log`` is ignored there too).  Host
    NumPy/SciPy: the fits never call it, they add the GPU-reduced sum (``binom_coeff_sum``).
    Entry order follows the reference's boolean indexing ``X[DP > 0]``: row-major (variant by
    variant) for every input format; sparse input gives a (1, n) array like the reference's
    np.matrix, dense input a 1-D array."""
    from scipy.sparse import issparse
    from scipy.special import binom
    from .counts import merge_counts
    if issparse(DP):
        _, _, _, ad, dp = merge_counts(AD.T, DP.T)      # columns of the transpose = rows
        keep = dp > 0
        ad, dp = ad[keep].astype(np.int64), dp[keep].astype(np.int64)
    else:
        keep = np.asarray(DP) > 0
        ad, dp = np.asarray(AD)[keep].astype(np.int64), np.asarray(DP)[keep].astype(np.int64)
    coeff = np.minimum(np.log(binom(dp, ad)), max_val).astype(np.float32)
    return coeff[None, :] if issparse(DP) else coeff


def beta_entropy(X, X_prior=None, axis=None):
    """Entropy of Beta(X[:, 0], X[:, 1]) distributions, or with X_prior their KL divergence
    from Beta(X_prior[:, 0], X_prior[:, 1]), summed over ``axis`` (vireo_base.py:77-127).  Host
    helper with the reference's name; inside the fits the same arithmetic is ``vrx_beta_kl``
    (vireo_amd/csrc/vrx_kernels.h)."""
    from scipy.special import betaln, digamma

    def cross(p, q):        # -E_p[log q]
        return (betaln(q[:, 0], q[:, 1]) - (q[:, 0] - 1) * digamma(p[:, 0])
                - (q[:, 1] - 1) * digamma(p[:, 1]) + (q.sum(axis=1) - 2) * digamma(p.sum(axis=1)))

    X = np.asarray(X)
    if X.ndim == 1:
        if X.shape[0] != 2:
            print("Error: unsupported shape. Make sure it's (N, 2)")
        X = X.reshape(-1, 2)
    if X_prior is None:
        return np.sum(cross(X, X), axis=axis)
    return np.sum(cross(X, np.asarray(X_prior)) - cross(X, X), axis=axis)


def binom_coeff_sum(AD, DP):
    """np.sum(get_binom_coeff(AD, DP)) of the reference (vireo_base.py:7-22 summed at
    vireo_model.py:313 / bmm_model.py:239): a float32 scalar, computed on the GPU."""
    return device_counts(AD, DP).binom_const()


# ---- small host-side K x K logic around the fits (not on the GPU path) -----------------
def match(ref_ids, new_ids, uniq_ref_only=True):
    """Index of each ref_id in new_ids (None when absent); ref_ids may repeat
    (vireo_base.py:130-184).  A sort-merge like the reference, so ties and the
    ``uniq_ref_only`` switch behave identically."""
    order_ref = np.argsort(ref_ids)
    order_new = np.argsort(new_ids)
    found = [None] * len(order_ref)
    j = 0
    for i in order_ref:
        while j < len(order_new) and new_ids[order_new[j]] < ref_ids[i]:
            j += 1
        if j < len(order_new) and new_ids[order_new[j]] == ref_ids[i]:
            found[i] = order_new[j]
            if uniq_ref_only:
                j += 1
    return np.array(found)


def optimal_match(X, Z, axis=1, return_delta=False):
    """Hungarian alignment of the slices of Z to those of X along ``axis`` by mean absolute
    difference (vireo_base.py:187-206)."""
    from scipy.optimize import linear_sum_assignment
    nx, nz = X.shape[axis], Z.shape[axis]
    delta = np.zeros((nx, nz))
    for i in range(nx):
        xi = np.take(X, i, axis=axis)
        for j in range(nz):
            delta[i, j] = np.mean(np.abs(xi - np.take(Z, j, axis=axis)))
    idx0, idx1 = linear_sum_assignment(delta)
    return (idx0, idx1, delta) if return_delta else (idx0, idx1)


def donor_select(GT_prob, ID_prob, n_donor, mode="distance"):
    """Keep n_donor of the donors found with extra donors: by size, or greedily the most
    mutually distant genotypes starting from the largest (vireo_base.py:217-254)."""
    size = np.sum(ID_prob, axis=0)
    K = GT_prob.shape[1]
    if mode == "size":
        keep = np.argsort(size)[::-1]
    else:
        dist = np.zeros((K, K))
        for i in range(K):
            for j in range(K):
                dist[i, j] = np.mean(np.abs(GT_prob[:, i, :] - GT_prob[:, j, :]))
        keep = [np.argmax(size)]
        left = np.delete(np.arange(K), keep)
        dist = np.delete(dist, keep, axis=1)
        while len(keep) < dist.shape[0]:
            nxt = np.argmax(np.min(dist[keep, :], axis=0))
            keep.append(left[nxt])
            left = np.delete(left, nxt)
            dist = np.delete(dist, nxt, axis=1)
    print("[vireo] donor size with searching extra %d donors:" % (K - n_donor))
    print("\t".join(["donor%d" % x for x in keep]))
    print("\t".join(["%.0f" % size[x] for x in keep]))
    out = ID_prob[:, keep[:n_donor]]
    out[out < 10**-10] = 10**-10
    return out
