"""Parity at sizes the oracle cannot iterate: the column-subset trick.

One coordinate-ascent iteration (vireoSNP/utils/vireo_model.py:257-264) factorises: given theta and
the old ``ID_prob``, ``update_GT_prob`` (:204-219) is independent per VARIANT; given theta and the
new ``GT_prob``, ``update_ID_prob`` (:187-201) is independent per CELL.  So the oracle run on 2 000
rows / 2 000 columns of AD, DP is EXACT for those rows of ``GT_prob`` and those rows of ``ID_prob`` /
``logLik_ID`` -- at any problem size.  Only the theta sums (``update_theta_size``, :165-185) are global:
they are checked with the oracle's own two SpMMs over the whole matrix.

Test infrastructure (imports ``oracle``): used by tests/test_gpu_fullsize.py and tests/perf/big_probe.py.
"""
import numpy as np
from scipy.sparse import csc_matrix, csr_matrix

from oracle import vireo_oracle as O


def _rows_subset(w, rows):
    """AD[rows, :], DP[rows, :] as CSR from the merged-CSC workload dict (one pass over the entries)"""
    N, M = w["shape"]
    lut = np.full(N, -1, dtype=np.int64)
    lut[rows] = np.arange(len(rows))
    sel = lut[w["rowidx"]]
    keep = np.flatnonzero(sel >= 0)
    r = sel[keep]
    c = np.searchsorted(w["colptr"], keep, side="right") - 1
    shape = (len(rows), M)
    DP = csr_matrix((w["dp"][keep].astype(np.int64), (r, c)), shape=shape)
    AD = csr_matrix((w["ad"][keep].astype(np.int64), (r, c)), shape=shape)
    AD.eliminate_zeros()
    return AD, DP


def _cols_subset(w, cols):
    """AD[:, cols], DP[:, cols] as CSC"""
    N, M = w["shape"]
    cp = w["colptr"]
    lens = (cp[cols + 1] - cp[cols]).astype(np.int64)
    ptr = np.zeros(len(cols) + 1, dtype=np.int64)
    np.cumsum(lens, out=ptr[1:])
    take = np.concatenate([np.arange(cp[c], cp[c + 1]) for c in cols]) if len(cols) else np.zeros(0, np.int64)
    shape = (N, len(cols))
    DP = csc_matrix((w["dp"][take].astype(np.int64), w["rowidx"][take], ptr), shape=shape)
    AD = csc_matrix((w["ad"][take].astype(np.int64), w["rowidx"][take].copy(), ptr.copy()), shape=shape)
    AD.eliminate_zeros()
    return AD, DP


def one_iteration_subset_check(m, counts, w, n_sub=2000, seed=0, rtol=1e-5, check_theta=True, flags=None):
    """``m``: a fitted device-side ``Vireo`` (default flags, uniform priors).  Runs ONE more iteration
    on the GPU step by step (theta, GT, ID, ELBO) and holds every output against the oracle: theta
    against the oracle's whole-matrix sums (``check_theta``), ``GT_prob`` on ``n_sub`` random variants,
    ``logLik_ID`` / ``ID_prob`` on ``n_sub`` random cells.  -> dict of the worst relative errors."""
    N, M = w["shape"]
    K = m.n_donor
    flags = flags or {}
    rng = np.random.default_rng(seed)
    V = np.sort(rng.choice(N, size=min(n_sub, N), replace=False))
    C = np.sort(rng.choice(M, size=min(n_sub, M), replace=False))
    ID0, GT0 = m.ID_prob.copy(), m.GT_prob.copy()
    mu0, sm0 = m.beta_mu.copy(), m.beta_sum.copy()
    # ---- the GPU's iteration, update by update
    m.update_theta_size(counts, None)
    m.update_GT_prob(counts, None)
    Lg = m.update_ID_prob(counts, None)
    out = {}

    def worst(a, b, name, atol=1e-290):
        a, b = np.asarray(a, float), np.asarray(b, float)
        d = np.abs(a - b)
        rel = float(np.max(d / np.maximum(np.abs(b), 1e-300) * (d > atol)))
        out[name] = rel
        np.testing.assert_allclose(a, b, rtol=rtol, atol=atol, err_msg=name)

    # ---- theta: global sums (the oracle's SpMMs over the whole matrix)
    if check_theta:
        from vireo_amd.synth import as_scipy
        AD, DP = as_scipy(w)
        st = O.vireo_new(M, N, K, ID_prob_init=ID0, GT_prob_init=GT0, beta_mu_init=mu0.copy(),
                         beta_sum_init=sm0.copy(), **flags)
        st.ID_prob, st.GT_prob = ID0, GT0
        O.vireo_theta_step(st, AD, DP)
        worst(m.beta_mu, st.beta_mu, "beta_mu")
        worst(m.beta_sum, st.beta_sum, "beta_sum")
        del AD, DP, st
    # ---- GT_prob on a variant subset: per-variant independent given theta (the GPU's) and the old ID_prob
    ADv, DPv = _rows_subset(w, V)
    sv = O.vireo_new(M, len(V), K, ID_prob_init=ID0, GT_prob_init=GT0[V], beta_mu_init=m.beta_mu.copy(),
                     beta_sum_init=m.beta_sum.copy(), **flags)
    sv.ID_prob, sv.GT_prob = ID0, GT0[V].copy()
    O.vireo_gt_step(sv, ADv, DPv)
    worst(m.GT_prob[V], sv.GT_prob, "GT_prob[%d variants]" % len(V))
    # ---- logLik_ID / ID_prob on a cell subset: per-cell independent given theta and the new GT_prob
    ADc, DPc = _cols_subset(w, C)
    sc = O.vireo_new(len(C), N, K, ID_prob_init=ID0[C], GT_prob_init=m.GT_prob, beta_mu_init=m.beta_mu.copy(),
                     beta_sum_init=m.beta_sum.copy(), **flags)
    sc.GT_prob = m.GT_prob
    Lc = O.vireo_id_step(sc, ADc, DPc)
    out["logLik_ID"] = float(np.max(np.abs(Lg[C] - Lc) / np.maximum(np.abs(Lc), 1e-300)))
    np.testing.assert_allclose(Lg[C], Lc, rtol=1e-9)
    worst(m.ID_prob[C], sc.ID_prob, "ID_prob[%d cells]" % len(C))
    assert np.array_equal(m.ID_prob[C].argmax(1), sc.ID_prob.argmax(1))
    out["assignments_equal"] = True
    return out
