"""GPU parity tests: the HIP path (through the Python mirror -> ctypes -> C ABI) against
(1) the golden vectors captured from the real reference and (2) the CPU oracle on the same
seeded inputs.  Bar (BASELINE.json north_star): bit-exact assignment indices and iteration
counts; ID_prob / GT_prob / ELBO within 1e-5 relative.  The tolerance used here is
RTOL = 1e-5 as stated; ATOL only absorbs sub-denormal noise on probabilities that underflow.
"""
import numpy as np
import pytest
from scipy.sparse import csc_matrix, csr_matrix

from oracle import vireo_oracle as O
from tests import gold

pytestmark = pytest.mark.gpu

RTOL = 1e-5
ATOL = 1e-290


@pytest.fixture(scope="module")
def va():
    import vireo_amd
    from vireo_amd import _lib
    _lib.require_gpu()          # fail loudly: there is no CPU fallback
    return vireo_amd


def close(a, b, rtol=RTOL, atol=ATOL):
    np.testing.assert_allclose(np.asarray(a), np.asarray(b), rtol=rtol, atol=atol)


def check_state(m, g, pre, gt=True):
    close(m.ID_prob, g[pre + "ID_prob"])
    if gt:
        close(m.GT_prob, g[pre + "GT_prob"])
    close(m.beta_mu, g[pre + "beta_mu"])
    close(m.beta_sum, g[pre + "beta_sum"])
    assert np.array_equal(np.argmax(m.ID_prob, 1), np.argmax(g[pre + "ID_prob"], 1))


def model_from(va, g, pre, **kw):
    m = va.Vireo(n_cell=g[pre + "ID_prob"].shape[0], n_var=g[pre + "GT_prob"].shape[0],
                 n_donor=g[pre + "ID_prob"].shape[1], ID_prob_init=g[pre + "ID_prob"],
                 GT_prob_init=g[pre + "GT_prob"], beta_mu_init=g[pre + "beta_mu"].copy(),
                 beta_sum_init=g[pre + "beta_sum"].copy(), **kw)
    m.ID_prob = g[pre + "ID_prob"].copy()
    m.GT_prob = g[pre + "GT_prob"].copy()
    return m


# ---------------------------------------------------------------- golden: single kernels
def test_binom_const(va):
    g = gold.load("binom_const")
    AD, DP = gold.c1()
    c = va.device_counts(AD, DP).binom_const()
    assert c.dtype == np.float32
    assert c == g["c1"]                         # bit-identical to np.sum(get_binom_coeff(...))
    mAD, mDP = gold.mito()                      # exercises the 700 clamp (DP up to 85197)
    assert va.device_counts(mAD, mDP).binom_const() == g["mito"]
    for fmt in (csr_matrix, lambda x: x.toarray()):          # same constant for every input format
        assert va.device_counts(fmt(AD), fmt(DP)).binom_const() == g["c1"]


@pytest.mark.parametrize("stray", [0, 3])
def test_binom_const_is_numpys_float32_sum(va, stray):
    """the constant of Vireo.fit (vireo_model.py:313) over several of NumPy's 8192-element
    buffers, summed on the device in NumPy's order: equal to the oracle's np.sum bit for bit;
    stray > 0 plants AD entries outside DP's pattern (dp == 0: skipped by the reference's mask,
    which moves the buffer boundaries)"""
    AD, DP = O.synth_donor(900, 700, 4, 0.06, seed=8)          # ~37 k entries: 4 buffers + a tail
    if stray:
        A, D = AD.toarray(), DP.toarray()
        holes = np.argwhere(D == 0)[:: max(1, (D == 0).sum() // stray)][:stray]
        for r, c in holes:
            A[r, c] = 2
        AD, DP = csc_matrix(A), csc_matrix(D)
    assert AD.nnz > 4 * 8192 or DP.nnz > 4 * 8192
    c = va.device_counts(AD, DP).binom_const()
    assert c.dtype == np.float32 and c == np.float32(O.binom_const(AD, DP))


def test_onestep_each_update(va):
    g = gold.load("c1_onestep")
    AD, DP = gold.c1()
    m = model_from(va, g, "s0_")
    m.update_theta_size(AD, DP)
    check_state(m, g, "s1_")
    m.update_GT_prob(AD, DP)
    check_state(m, g, "s2_")
    L = m.update_ID_prob(AD, DP)
    check_state(m, g, "s3_")
    close(L, g["logLik_ID"], rtol=1e-9)
    close(m.get_ELBO(L, AD, DP), g["ELBO"], rtol=1e-9)
    close(m.get_ELBO(None, AD, DP), g["ELBO_recompute"], rtol=1e-9)


# ---------------------------------------------------------------- golden: full traces
def test_trace_and_warm_restart(va):
    g = gold.load("c1_trace_seed2")
    AD, DP = gold.c1()
    np.random.seed(2)
    m = va.Vireo(n_var=AD.shape[0], n_cell=AD.shape[1], n_donor=4)
    assert np.array_equal(m.ID_prob, g["init_ID_prob"])     # RNG stream consumed identically
    assert np.array_equal(m.GT_prob, g["init_GT_prob"])
    m.fit(AD, DP, min_iter=5, max_iter=20, delay_fit_theta=3, verbose=False)
    assert len(m.ELBO_) == int(g["n_first"])
    check_state(m, g, "mid_")
    m.fit(AD, DP, min_iter=5, verbose=False)
    assert len(m.ELBO_) == len(g["ELBO_"]) == 79
    close(m.ELBO_, g["ELBO_"])
    check_state(m, g, "end_")
    print("max rel ELBO err %.3e" % np.max(np.abs(m.ELBO_ / g["ELBO_"] - 1)))


def test_fused_cell_softmax_equals_separate_kernels(va, monkeypatch):
    """small problems run the cell pass with the softmax and the cells' ELBO terms in its epilogue
    (vrx_spmm FUSE = 1): the posteriors are the separate kernels' bit for bit, the ELBO differs only
    by the grouping of its partial sums"""
    AD, DP = gold.c1()
    fits = {}
    for fuse in ("0", "1"):
        monkeypatch.setenv("VIREO_FUSE_SOFTMAX", fuse)
        np.random.seed(2)
        m = va.Vireo(n_var=AD.shape[0], n_cell=AD.shape[1], n_donor=4)
        m.fit(AD, DP, min_iter=5, max_iter=15, delay_fit_theta=3, verbose=False)
        fits[fuse] = (np.array(m.ELBO_), m.ID_prob.copy(), m.GT_prob.copy())
    assert len(fits["0"][0]) == len(fits["1"][0])
    assert np.array_equal(fits["0"][1], fits["1"][1])
    assert np.array_equal(fits["0"][2], fits["1"][2])
    close(fits["1"][0], fits["0"][0], rtol=1e-13)


@pytest.mark.parametrize("tag,kw", [("ase", dict(ASE_mode=True)),
                                    ("fixsum", dict(fix_beta_sum=True)),
                                    ("notheta", dict(learn_theta=False))])
def test_flags(va, tag, kw):
    g = gold.load("c1_flag_" + tag)
    AD, DP = gold.c1()
    np.random.seed(2)
    m = va.Vireo(n_var=AD.shape[0], n_cell=AD.shape[1], n_donor=4, **kw)
    m.fit(AD, DP, max_iter=12, verbose=False)
    assert len(m.ELBO_) == len(g["ELBO_"])
    close(m.ELBO_, g["ELBO_"])
    check_state(m, g, "end_")


@pytest.mark.parametrize("tag,learn", [("fixedGT", False), ("priorGT", True)])
def test_gt_prior(va, tag, learn):
    g = gold.load("c1_flag_" + tag)
    AD, DP = gold.c1()
    np.random.seed(2)
    m = va.Vireo(n_var=AD.shape[0], n_cell=AD.shape[1], n_donor=4, learn_GT=learn,
                 GT_prob_init=g["GT_prior_in"].copy())
    m.set_prior(GT_prior=g["GT_prior_in"].copy())
    m.fit(AD, DP, max_iter=12, verbose=False)
    assert len(m.ELBO_) == len(g["ELBO_"])
    close(m.ELBO_, g["ELBO_"])
    check_state(m, g, "end_")


# ---------------------------------------------------------------- golden: vireo_wrap
@pytest.mark.parametrize("name,kw", [
    ("c1_wrap_seed2_init1", dict(n_donor=4, n_init=1, random_seed=2)),
    ("c1_wrap_seed2_init4", dict(n_donor=4, n_init=4, random_seed=2)),
    ("c1_wrap_seed2_nodoublet", dict(n_donor=3, n_init=2, random_seed=2, check_doublet=False)),
    ("c1_wrap_seed1_init50", dict(n_donor=4, n_init=50, random_seed=1)),
    ("c1_wrap_extra1_dist", dict(n_donor=4, n_init=3, random_seed=2, n_extra_donor=1)),
    ("c1_wrap_extra2_size", dict(n_donor=3, n_init=2, random_seed=5, n_extra_donor=2,
                                 extra_donor_mode="size")),
    ("c1_wrap_ase", dict(n_donor=4, n_init=2, random_seed=2, ASE_mode=True)),
])
def test_wrap(va, name, kw, capsys):
    g = gold.load(name)
    AD, DP = gold.c1()
    rv = va.vireo_wrap(AD, DP, **kw)
    capsys.readouterr()
    close(rv["LB_list"], g["LB_list"])
    assert np.argmax(rv["LB_list"]) == np.argmax(g["LB_list"])
    close(rv["LB_doublet"], g["LB_doublet"])
    for k in ("ID_prob", "GT_prob", "doublet_prob", "theta_shapes", "theta_mean", "theta_sum"):
        close(rv[k], g[k])
    close(rv["doublet_LLR"], g["doublet_LLR"], rtol=1e-5, atol=1e-8)
    assert np.array_equal(np.argmax(rv["ID_prob"], 1), np.argmax(g["ID_prob"], 1))
    if kw.get("check_doublet", True):
        assert np.array_equal(np.argmax(rv["doublet_prob"], 1), np.argmax(g["doublet_prob"], 1))


def test_wrap_on_lds_resident_passes(va, monkeypatch, capsys):
    """the notebook-style wrap (restarts, final fit, doublets) with the LDS-resident passes
    forced on: K = 4 fits, and the doublet log-likelihoods as one K' = 4 + 6 = 10 column
    operand; against the reference's golden output (the problem cache is cleared so that
    the problem is built with the LDS streams)."""
    from vireo_amd.counts import clear_cache
    monkeypatch.setenv("VIREO_LDS", "1")
    clear_cache()
    g = gold.load("c1_wrap_seed2_init4")
    AD, DP = gold.c1()
    rv = va.vireo_wrap(AD.copy(), DP.copy(), n_donor=4, n_init=4, random_seed=2)
    clear_cache()
    capsys.readouterr()
    close(rv["LB_list"], g["LB_list"])
    close(rv["LB_doublet"], g["LB_doublet"])
    for k in ("ID_prob", "GT_prob", "doublet_prob"):
        close(rv[k], g[k])
    assert np.array_equal(np.argmax(rv["doublet_prob"], 1), np.argmax(g["doublet_prob"], 1))


# ---------------------------------------------------------------- golden: clone mode
def test_bmm_mito_known_answer(va):
    g = gold.load("mito_bmm_k3_seed1")
    AD, DP = gold.mito()
    b = va.BinomMixtureVB(n_var=AD.shape[0], n_cell=AD.shape[1], n_donor=3)
    b.fit(AD, DP, min_iter=30, n_init=50, random_seed=1, verbose=False)
    assert len(b.ELBO_iters) == len(g["ELBO_iters"])
    close(b.ELBO_iters, g["ELBO_iters"])
    close(b.ELBO_iters[-1], -190779.74335041404)        # examples/vireoSNP_clones.ipynb
    close(b.ELBO_inits, g["ELBO_inits"])
    close(b.ID_prob, g["ID_prob"])
    close(b.beta_mu, g["beta_mu"])
    close(b.beta_sum, g["beta_sum"])
    assert np.array_equal(np.argmax(b.ID_prob, 1), np.argmax(g["ID_prob"], 1))


def test_bmm_trace(va):
    g = gold.load("mito_bmm_k4_trace")
    AD, DP = gold.mito()
    np.random.seed(5)
    b = va.BinomMixtureVB(n_var=AD.shape[0], n_cell=AD.shape[1], n_donor=4)
    assert np.array_equal(b.ID_prob, g["ID_prob_init"])
    b._fit_BV(AD, DP, max_iter=15, min_iter=5, verbose=False)
    assert len(b.ELBO_iters) == len(g["ELBO_iters"])
    close(b.ELBO_iters, g["ELBO_iters"])
    close(b.ID_prob, g["ID_prob"])
    close(b.beta_mu, g["beta_mu"])
    close(b.beta_sum, g["beta_sum"])


# ---------------------------------------------------------------- golden: other K
@pytest.mark.parametrize("tag", ["k3", "k16", "k5"])
def test_synthetic_golden(va, tag):
    g = gold.load("synth_" + tag)
    AD, DP = gold.unpack(g)
    n, m_ = AD.shape
    k = g["init_ID_prob"].shape[1]
    np.random.seed(1)
    m = va.Vireo(n_var=n, n_cell=m_, n_donor=k)
    m.fit(AD, DP, min_iter=5, max_iter=20, delay_fit_theta=3, verbose=False)
    assert len(m.ELBO_) == len(g["ELBO_"])
    close(m.ELBO_, g["ELBO_"])
    check_state(m, g, "end_")
    dbl, sing, llr = va.predict_doublet(m, AD, DP)
    close(dbl, g["doublet_prob"])
    close(sing, g["singlet_prob"])
    close(llr, g["doublet_LLR"], rtol=1e-5, atol=1e-8)
    close(m.GT_prob, g["GT_prob_after_doublet"])


# ---------------------------------------------------------------- oracle: input formats
def test_input_formats_agree(va):
    AD, DP = gold.c1()
    outs = []
    for conv in (lambda X: X, lambda X: csr_matrix(X), lambda X: X.astype(np.float64),
                 lambda X: X.toarray()):
        np.random.seed(4)
        m = va.Vireo(n_var=AD.shape[0], n_cell=AD.shape[1], n_donor=3)
        m.fit(conv(AD), conv(DP), max_iter=8, verbose=False)
        outs.append((m.ELBO_.copy(), m.ID_prob.copy()))
    for e, p in outs[1:]:
        assert np.array_equal(e, outs[0][0]) and np.array_equal(p, outs[0][1])


# ---------------------------------------------------------------- oracle: edge cases
def _ragged_case(seed=0):
    """empty cells, empty variants, one variant covering > 4096 cells and one cell covering
    > 4096 variants (both take the split-segment path), AD entries without DP."""
    rng = np.random.default_rng(seed)
    N, M = 6000, 5000
    dp = (rng.random((N, M)) < 0.002) * (1 + rng.poisson(1.0, (N, M)))
    dp[7, :] = 1 + rng.poisson(2.0, M)            # long variant row (5000 entries)
    dp[:, 11] = 1 + rng.poisson(2.0, N)           # long cell column (6000 entries)
    dp[100:120, :] = 0                            # empty variants
    dp[:, 200:230] = 0                            # empty cells
    ad = rng.binomial(dp, 0.4)
    ad[5, 300] = 2 if dp[5, 300] == 0 else ad[5, 300]     # AD entry outside DP's pattern
    dp[6, 301], ad[6, 301] = 0, 3
    return csc_matrix(ad), csc_matrix(dp)


@pytest.mark.parametrize("K", [2, 7])
def test_ragged_and_split_segments(va, K):
    AD, DP = _ragged_case()
    np.random.seed(3)
    ref = O.vireo_new(AD.shape[1], AD.shape[0], K)
    np.random.seed(3)
    dev = va.Vireo(n_var=AD.shape[0], n_cell=AD.shape[1], n_donor=K)
    O.vireo_fit(ref, AD, DP, max_iter=8)
    dev.fit(AD, DP, max_iter=8, verbose=False)
    assert len(dev.ELBO_) == len(ref.ELBO_)
    close(dev.ELBO_, ref.ELBO_)
    close(dev.ID_prob, ref.ID_prob)
    close(dev.GT_prob, ref.GT_prob)
    # empty cells keep the prior: uniform posterior
    close(dev.ID_prob[200:230], 1.0 / K)


def test_bmm_long_rows_vs_oracle(va):
    """clone-mode shape: few variants, many cells, ~90 % dense => every variant row is split
    over many segments."""
    rng = np.random.default_rng(1)
    N, M, K = 12, 30000, 5
    mask = rng.random((N, M)) < 0.9
    dp = rng.poisson(30, (N, M)) * mask
    z = rng.integers(0, K, M)
    af = rng.beta(0.3, 3, (N, K))
    ad = rng.binomial(dp, af[:, z])
    AD, DP = csc_matrix(ad), csc_matrix(dp)
    np.random.seed(2)
    ref = O.bmm_new(M, N, K)
    O.bmm_fit_vb(ref, AD, DP, max_iter=12, min_iter=5)
    np.random.seed(2)
    dev = va.BinomMixtureVB(n_var=N, n_cell=M, n_donor=K)
    dev._fit_BV(AD, DP, max_iter=12, min_iter=5, verbose=False)
    assert len(dev.ELBO_iters) == len(ref.ELBO_iters)
    close(dev.ELBO_iters, ref.ELBO_iters)
    close(dev.ID_prob, ref.ID_prob)
    close(dev.beta_mu, ref.beta_mu)
    close(dev.beta_sum, ref.beta_sum)


def test_wide_donor_count_vs_oracle(va):
    """K = 70 > 64 lanes: column-chunked sparse pass and looping softmax."""
    AD, DP = O.synth_donor(400, 300, 5, 0.1, seed=2)
    K = 70
    np.random.seed(6)
    ref = O.vireo_new(300, 400, K)
    np.random.seed(6)
    dev = va.Vireo(n_var=400, n_cell=300, n_donor=K)
    O.vireo_fit(ref, AD, DP, max_iter=6)
    dev.fit(AD, DP, max_iter=6, verbose=False)
    close(dev.ELBO_, ref.ELBO_)
    close(dev.ID_prob, ref.ID_prob)
    close(dev.GT_prob, ref.GT_prob)


# ---------------------------------------------------------------- oracle: BASELINE config 2
def test_config2_vs_oracle(va):
    """synthetic N=10k x M=5k, K=4, ~1 % nnz (BASELINE.json configs[1]) with the timing
    protocol's seeds."""
    AD, DP = O.synth_donor(10000, 5000, 4, 0.01, seed=0)
    np.random.seed(1)
    ref = O.vireo_new(5000, 10000, 4)
    np.random.seed(1)
    dev = va.Vireo(n_var=10000, n_cell=5000, n_donor=4)
    O.vireo_fit(ref, AD, DP, min_iter=5, max_iter=20, delay_fit_theta=3)
    dev.fit(AD, DP, min_iter=5, max_iter=20, delay_fit_theta=3, verbose=False)
    assert len(dev.ELBO_) == len(ref.ELBO_)
    close(dev.ELBO_, ref.ELBO_)
    close(dev.ID_prob, ref.ID_prob)
    close(dev.GT_prob, ref.GT_prob)
    assert np.array_equal(np.argmax(dev.ID_prob, 1), np.argmax(ref.ID_prob, 1))


@pytest.mark.parametrize("fmt,tiles_c,tiles_v", [(0, 1, 1), (1, 1, 1), (2, 1, 1), (0, 4, 2),
                                                  (1, 16, 8), (2, 24, 16), (0, 3, 5)])
def test_entry_formats_and_tiling_vs_oracle(va, monkeypatch, fmt, tiles_c, tiles_v):
    """every entry format (4/8/12 B per non-zero) and tiled, XCD-ordered segment tables give
    the oracle's result; K=6 takes the 2-columns-per-lane variant pass, K=5 the 1-column one."""
    from vireo_amd.counts import DeviceCounts
    monkeypatch.setenv("VIREO_ENTRY_FMT", str(fmt))
    monkeypatch.setenv("VIREO_TILES_CELL", str(tiles_c))
    monkeypatch.setenv("VIREO_TILES_VAR", str(tiles_v))
    AD, DP = O.synth_donor(1500, 900, 4, 0.05, seed=3)
    counts = DeviceCounts(AD, DP)
    for K in (6, 5):
        np.random.seed(7)
        ref = O.vireo_new(900, 1500, K)
        np.random.seed(7)
        dev = va.Vireo(n_var=1500, n_cell=900, n_donor=K)
        O.vireo_fit(ref, AD, DP, max_iter=8)
        dev.fit(counts, None, max_iter=8, verbose=False)
        assert len(dev.ELBO_) == len(ref.ELBO_)
        close(dev.ELBO_, ref.ELBO_)
        close(dev.ID_prob, ref.ID_prob)
        close(dev.GT_prob, ref.GT_prob)
    close(counts.binom_const(), O.binom_const(AD, DP), rtol=1e-6)


@pytest.mark.parametrize("fmt,blocks,sort", [(0, 2048, 1), (1, 1, 1), (2, 40, 1), (0, 40, 0)])
def test_lds_resident_passes_vs_oracle(va, monkeypatch, fmt, blocks, sort):
    """the LDS-resident (two-dimensionally tiled) passes, forced on a small ragged problem:
    one or many contracted ranges, K = 16 / 12 (4 / 3 lanes per entry), K = 8 / 4 (2 / 4 entries of
    a row at once on 2 / 1 lanes each) and K = 9 / 6 / 3 / 15 / 2 (dense rows zero-padded to
    12 / 8 / 4 / 16 / 4 columns in LDS) and K = 20 / 33 (column blocks of 16).  The long
    variant / cell of the ragged case are cut into interleaved pieces (sort=1) or kept whole
    in natural row order (sort=0)."""
    from vireo_amd import _lib
    from vireo_amd.counts import DeviceCounts
    from vireo_amd.engine import DeviceModel
    monkeypatch.setenv("VIREO_LDS", "1")
    monkeypatch.setenv("VIREO_LDS_BLOCKS", str(blocks))
    monkeypatch.setenv("VIREO_LDS_SORT", str(sort))
    monkeypatch.setenv("VIREO_ENTRY_FMT", str(fmt))
    AD, DP = _ragged_case(seed=4)
    counts = DeviceCounts(AD, DP)
    info = DeviceModel(counts, _lib.KIND_VIREO, 16).info()
    assert info["lds_cell"] and info["lds_variant"]
    assert (info["extra_pieces_cell"] > 0) == (info["extra_pieces_variant"] > 0) == bool(sort)
    for K in (16, 12, 8, 4, 9, 6, 3, 15, 2, 20, 33):
        np.random.seed(11)
        ref = O.vireo_new(AD.shape[1], AD.shape[0], K)
        np.random.seed(11)
        dev = va.Vireo(n_var=AD.shape[0], n_cell=AD.shape[1], n_donor=K)
        O.vireo_fit(ref, AD, DP, max_iter=7)
        dev.fit(counts, None, max_iter=7, verbose=False)
        assert len(dev.ELBO_) == len(ref.ELBO_)
        close(dev.ELBO_, ref.ELBO_)
        close(dev.ID_prob, ref.ID_prob)
        close(dev.GT_prob, ref.GT_prob)


@pytest.mark.parametrize("top,var_form", [(2047, 0), (2048, 0), (7, 2), (2048, 2), (16383, 2),
                                          (16384, 2), (90000, 2), (7, 3), (16384, 3), (90000, 3)])
def test_lds_count_limit_and_tiny_shapes(va, monkeypatch, top, var_form):
    """The single-valued AD / BD words (cell stream, FORM 1; variant stream, FORM 2: the
    default) carry the top bits of the value's double and cut a count with more than three
    significant bits into several entries, so the LDS-resident passes take any count (7 / 2048 /
    16383 / 16384: chunk boundaries; 90000: data/mitoDNA-like depth).  The (ad, dp) pair words of
    the older variant stream (VIREO_VAR_FORM=0) hold counts < 2048: at the limit that pass is
    used, one above it silently stays on the global-gather kernel.  Shapes smaller than one
    tile / one slab (N=70 variants, M=40 cells) and a single contracted range."""
    from vireo_amd import _lib
    from vireo_amd.counts import DeviceCounts
    from vireo_amd.engine import DeviceModel
    monkeypatch.setenv("VIREO_LDS", "1")
    monkeypatch.setenv("VIREO_VAR_FORM", str(var_form))
    monkeypatch.setenv("VIREO_CELL_FORM", "1")     # (counts this deep would pick pair words)
    expect_lds = var_form >= 2 or top < 2048
    rng = np.random.default_rng(5)
    dp = (rng.random((70, 40)) < 0.3) * rng.integers(1, 60, (70, 40))
    dp[3, 7] = top
    ad = rng.binomial(dp, 0.3)
    AD, DP = csc_matrix(ad), csc_matrix(dp)
    counts = DeviceCounts(AD, DP)
    K = 8
    info = DeviceModel(counts, _lib.KIND_VIREO, K).info()
    assert info["lds_variant"] == expect_lds
    assert info["lds_cell"] and info["cell_form"] == 1
    np.random.seed(2)
    ref = O.vireo_new(40, 70, K)
    np.random.seed(2)
    dev = va.Vireo(n_var=70, n_cell=40, n_donor=K)
    O.vireo_fit(ref, AD, DP, max_iter=6)
    dev.fit(counts, None, max_iter=6, verbose=False)
    close(dev.ELBO_, ref.ELBO_)
    close(dev.ID_prob, ref.ID_prob)
    close(dev.GT_prob, ref.GT_prob)


@pytest.mark.parametrize("N,M,K,top,fill", [(50, 40000, 5, 6, "1"), (50, 40000, 5, 6, "0"),
                                             (300, 30000, 16, 90, "1"), (25000, 60, 8, 6, "1")])
def test_coarse_shapes_fill_the_cus(va, monkeypatch, N, M, K, top, fill):
    """few variants x many cells (clone-mode shapes) and the transpose: the LDS-resident passes
    shorten their slabs / spread their tiles until every CU has a (tile, slab) visit
    (VIREO_LDS_FILL_CUS; AD/BD words over virtual rows for the shallow counts, pair words for the
    deep ones) -- same results as with full slabs and tiles, and as the oracle"""
    from vireo_amd import _lib
    from vireo_amd.counts import DeviceCounts
    from vireo_amd.engine import DeviceModel
    monkeypatch.setenv("VIREO_LDS", "1")
    monkeypatch.setenv("VIREO_LDS_FILL_CUS", fill)
    rng = np.random.default_rng(N + M)
    dp = (rng.random((N, M)) < 0.25) * rng.integers(1, top + 1, (N, M))
    ad = rng.binomial(dp, rng.choice([0.03, 0.5, 0.96], (N, 1)))
    AD, DP = csc_matrix(ad), csc_matrix(dp)
    counts = DeviceCounts(AD, DP)
    info = DeviceModel(counts, _lib.KIND_VIREO, K).info()
    assert info["lds_variant"] and info["lds_cell"]
    np.random.seed(4)
    ref = O.vireo_new(M, N, K)
    np.random.seed(4)
    dev = va.Vireo(n_var=N, n_cell=M, n_donor=K)
    O.vireo_fit(ref, AD, DP, min_iter=2, max_iter=5, delay_fit_theta=1)
    dev.fit(counts, None, min_iter=2, max_iter=5, delay_fit_theta=1, verbose=False)
    assert len(dev.ELBO_) == len(ref.ELBO_)
    close(dev.ELBO_, ref.ELBO_)
    close(dev.ID_prob, ref.ID_prob)
    close(dev.GT_prob, ref.GT_prob)
    close(dev.beta_mu, ref.beta_mu)


@pytest.mark.parametrize("shape", ["row", "full"])
def test_nonuniform_id_prior_vs_oracle(va, shape):
    """ID_prior as one broadcast row (set_prior with a 1-D array, vireo_model.py:122-125) and
    as a full (n_cell, n_donor) table, un-normalised on purpose: scipy's entropy normalises
    q, np.log(prior) in the softmax does not need to."""
    AD, DP = gold.c1()
    M, K = AD.shape[1], 4
    rng = np.random.default_rng(8)
    prior = rng.random(K) + 0.2 if shape == "row" else rng.random((M, K)) + 0.2
    np.random.seed(3)
    ref = O.vireo_new(M, AD.shape[0], K)
    O.vireo_prior(ref, ID_prior=prior.copy())
    np.random.seed(3)
    dev = va.Vireo(n_var=AD.shape[0], n_cell=M, n_donor=K)
    dev.set_prior(ID_prior=prior.copy())
    O.vireo_fit(ref, AD, DP, max_iter=10)
    dev.fit(AD, DP, max_iter=10, verbose=False)
    assert len(dev.ELBO_) == len(ref.ELBO_)
    close(dev.ELBO_, ref.ELBO_)
    close(dev.ID_prob, ref.ID_prob)
    close(dev.GT_prob, ref.GT_prob)


def test_bmm_fix_beta_sum_and_custom_prior_vs_oracle(va):
    AD, DP = gold.mito()
    N, M, K = AD.shape[0], AD.shape[1], 3
    np.random.seed(4)
    ref = O.bmm_new(M, N, K, fix_beta_sum=True)
    ID0 = ref.ID_prob.copy()
    O.bmm_fit_vb(ref, AD, DP, max_iter=12, min_iter=3)
    dev = va.BinomMixtureVB(n_var=N, n_cell=M, n_donor=K, fix_beta_sum=True, ID_prob_init=ID0)
    dev._fit_BV(AD, DP, max_iter=12, min_iter=3, verbose=False)
    assert len(dev.ELBO_iters) == len(ref.ELBO_iters)
    close(dev.ELBO_iters, ref.ELBO_iters)
    close(dev.ID_prob, ref.ID_prob)
    close(dev.beta_mu, ref.beta_mu)
    close(dev.beta_sum, ref.beta_sum)      # stays at its initial 30
    # the public step methods of the clone model
    L = dev.get_E_logLik(AD, DP)
    close(L, O.bmm_cell_loglik(ref, AD, DP), rtol=1e-9)
    close(dev.get_ELBO(AD, DP, logLik_ID=L), O.bmm_elbo(ref, L), rtol=1e-9)


def test_determinism(va):
    AD, DP = O.synth_donor(3000, 2000, 16, 0.03, seed=5)
    runs = []
    for _ in range(2):
        np.random.seed(9)
        m = va.Vireo(n_var=3000, n_cell=2000, n_donor=16)
        m.fit(AD, DP, max_iter=10, verbose=False)
        runs.append((m.ELBO_.copy(), m.ID_prob.copy(), m.GT_prob.copy()))
    for a, b in zip(runs[0], runs[1]):
        assert np.array_equal(a, b)          # fixed-order reductions: bitwise reproducible


# ---------------------------------------------------------------- restart plumbing
@pytest.mark.parametrize("K,T", [(4, 3), (16, 3), (7, 2), (9, 5), (20, 3), (100, 3)])
def test_device_normalise_is_numpys(va, K, T):
    """vrx_model_set_state_raw: raw draws normalised on the device over the last axis are
    bit-identical to normalize() on the host (NumPy's pairwise row sum, vireo_base.py:44-55)"""
    from vireo_amd import _lib
    from vireo_amd.counts import DeviceCounts
    from vireo_amd.engine import DeviceModel
    AD, DP = O.synth_donor(300, 200, 3, 0.05, seed=3)
    dm = DeviceModel(DeviceCounts(AD, DP), _lib.KIND_VIREO, K, n_gt=T)
    rng = np.random.default_rng(K * 10 + T)
    ID, GT = rng.random((200, K)), rng.random((300, K, T))
    mu, sm = np.linspace(0.01, 0.99, T)[None, :], np.full((1, T), 50.0)
    dm.set_state_raw(ID, GT, mu, sm)
    gID, gGT, gmu, gsm = dm.get_state()
    assert np.array_equal(gID, va.normalize(ID))
    assert np.array_equal(gGT, va.normalize(GT))
    assert np.array_equal(gmu, mu) and np.array_equal(gsm, sm)


def test_snapshot_restore_and_restart_runner(va):
    """the best restart stays in HBM: DeviceRestarts over 4 restarts == 4 independent Vireo
    fits from the same draws, bit for bit, winner included"""
    from vireo_amd.restarts import DeviceRestarts, LegacyStream
    from vireo_amd.vireo_wrap import _template
    AD, DP = gold.c1()
    counts = va.device_counts(AD, DP)
    N, M, K = AD.shape[0], AD.shape[1], 4
    np.random.seed(2)
    singles = []
    for _ in range(4):
        m = va.Vireo(n_var=N, n_cell=M, n_donor=K)
        m.fit(counts, None, min_iter=5, max_iter=20, delay_fit_theta=3, verbose=False)
        singles.append(m)
    np.random.seed(2)
    st, run = LegacyStream(), DeviceRestarts(counts, _template(counts, K, True, None, {}))
    elbos = [run.run(i, st.rand(M, K), st.rand(N, K, 3), None, None, 20, 3) for i in range(4)]
    assert elbos == [m.ELBO_[-1] for m in singles]
    best = int(np.argmax(elbos))
    win = run.winner(best, refine=False)
    for name in ("ID_prob", "GT_prob", "beta_mu", "beta_sum", "ELBO_"):
        assert np.array_equal(getattr(win, name), getattr(singles[best], name)), name
    g = gold.load("c1_wrap_seed2_init4")
    close(np.array(elbos), g["LB_list"], rtol=1e-9)


def test_count_cache_notices_inplace_edits(va):
    """the (AD, DP) -> device cache is keyed on object identity AND buffer contents"""
    from vireo_amd import counts as C
    C.clear_cache()
    AD, DP = O.synth_donor(400, 300, 3, 0.05, seed=1)
    a = va.device_counts(AD, DP)
    assert va.device_counts(AD, DP) is a
    j = np.flatnonzero(np.diff(DP.indptr) >= 2)[0]           # a column with two entries:
    lo = DP.indptr[j]                                         # swap their values (same sum)
    DP.data[lo], DP.data[lo + 1] = DP.data[lo + 1] + 1, DP.data[lo] - 1
    AD.data[:] = np.minimum(AD.data, 0)
    b = va.device_counts(AD, DP)
    assert b is not a
    AD2, DP2 = AD.copy(), DP.copy()
    assert va.device_counts(AD2, DP2) is not b               # equal content, other objects
    C.clear_cache()


def test_bmm_prior_that_differs_per_clone(va):
    """a (1, n_donor) Beta prior broadcasts over the variants (bmm_model.py:92-98)"""
    AD, DP = gold.mito()
    N, M, K = AD.shape[0], AD.shape[1], 3
    mu = np.array([[0.2, 0.5, 0.7]])
    sm = np.array([[2.0, 4.0, 3.0]])
    np.random.seed(4)
    ref = O.bmm_new(M, N, K)
    ID0 = ref.ID_prob.copy()
    ref.theta_s1_prior, ref.theta_s2_prior = mu * sm, (1 - mu) * sm
    O.bmm_fit_vb(ref, AD, DP, max_iter=8, min_iter=3)
    dev = va.BinomMixtureVB(n_var=N, n_cell=M, n_donor=K, ID_prob_init=ID0)
    dev.set_prior(beta_mu_prior=mu, beta_sum_prior=sm)
    dev._fit_BV(AD, DP, max_iter=8, min_iter=3, verbose=False)
    close(dev.ELBO_iters, ref.ELBO_iters)
    close(dev.beta_mu, ref.beta_mu)
    close(dev.ID_prob, ref.ID_prob)


@pytest.mark.parametrize("case", ["uniform", "ragged", "wide_counts"])
def test_device_builder_equals_host_builder(va, monkeypatch, case):
    """f4: the problem built on the device (vrx_build.h: validation, radix-sort transposition,
    packing, tiled-stream construction in HIP kernels) is bit-identical to the host builder's:
    packed entries, stream words, boundaries, wave starts and row maps of both orientations;
    and a fit on it agrees with the oracle."""
    from vireo_amd.counts import DeviceCounts
    monkeypatch.setenv("VIREO_LDS", "1")
    if case == "uniform":
        AD, DP = O.synth_donor(2500, 1800, 5, 0.03, seed=11)
    elif case == "ragged":
        AD, DP = _ragged_case(seed=6)
    else:
        rng = np.random.default_rng(12)
        dp = (rng.random((700, 500)) < 0.2) * rng.integers(1, 2047, (700, 500))
        dp[5, :] = rng.integers(1, 2047, 500)
        ad = rng.binomial(dp, 0.4)
        dp[3, 7], ad[3, 7] = 100, 105               # AD outside DP's range: negative BD
        AD, DP = csc_matrix(ad), csc_matrix(dp)
    monkeypatch.setenv("VIREO_BUILD", "host")
    host = DeviceCounts(AD, DP)
    monkeypatch.setenv("VIREO_BUILD", "device")
    dev = DeviceCounts(AD, DP)
    dh, dd = host.digest(), dev.digest()
    assert dd[5] == 0 and dd[11] == 0               # device build: no gather segment tables
    for k in (0, 1, 2, 3, 4, 6, 7, 8, 9, 10):
        assert dh[k] == dd[k], "digest %d differs" % k
    assert np.array_equal(host.n_vars(), dev.n_vars())
    assert host.binom_const() == dev.binom_const()
    for K in (1, 5, 16):
        np.random.seed(21)
        ref = O.vireo_new(AD.shape[1], AD.shape[0], K)
        np.random.seed(21)
        m = va.Vireo(n_var=AD.shape[0], n_cell=AD.shape[1], n_donor=K)
        O.vireo_fit(ref, AD, DP, max_iter=6)
        m.fit(dev, None, max_iter=6, verbose=False)
        close(m.ELBO_, ref.ELBO_)
        close(m.ID_prob, ref.ID_prob)
        close(m.GT_prob, ref.GT_prob)


def test_device_builder_reports_bad_input(va, monkeypatch):
    from vireo_amd import _lib
    from vireo_amd.counts import DeviceCounts
    monkeypatch.setenv("VIREO_LDS", "1")
    monkeypatch.setenv("VIREO_BUILD", "device")
    colptr = np.array([0, 2, 3], dtype=np.int64)
    with pytest.raises(_lib.VrxError, match="not strictly increasing"):
        DeviceCounts.from_merged((4, 2), colptr, np.array([2, 1, 0], dtype=np.int32),
                                 np.ones(3, np.int32), np.ones(3, np.int32))
    with pytest.raises(_lib.VrxError, match="negative count"):
        DeviceCounts.from_merged((4, 2), colptr, np.array([1, 2, 0], dtype=np.int32),
                                 np.array([1, -1, 1], np.int32), np.ones(3, np.int32))


# ---------------------------------------------------------------- restart batches
def _independent_fits(va, counts, N, M, K, n, seed, **fit):
    np.random.seed(seed)
    out = []
    for _ in range(n):
        m = va.Vireo(n_var=N, n_cell=M, n_donor=K)
        m.fit(counts, None, verbose=False, **fit)
        out.append(m)
    return out


@pytest.mark.parametrize("K,R,lds", [(4, 4, True), (4, 3, True), (3, 5, True), (16, 2, True),
                                     (4, 4, False), (2, 16, False), (16, 3, False), (7, 5, False)])
def test_restart_batch_equals_independent_fits(va, monkeypatch, K, R, lds):
    """vrx_model_cfg.n_batch: R restarts in one device model (one sparse pass for all of them,
    per-restart theta / trace / stop rule) == R independent fits from the same draws.  On the
    LDS-resident passes every column is accumulated in the same order whatever its neighbours
    are, so the match is bitwise; the gather kernels pick their lane layout from the column
    count, which changes the order of a row's partial sums (tolerance 1e-9)."""
    from vireo_amd import _lib
    from vireo_amd.counts import DeviceCounts
    from vireo_amd.engine import DeviceBatch, DeviceModel
    monkeypatch.setenv("VIREO_LDS", "1" if lds else "0")     # read when the problem is built
    AD, DP = O.synth_donor(1500, 900, K, 0.04, seed=K + R)
    counts = DeviceCounts(AD, DP)
    N, M = AD.shape
    singles = _independent_fits(va, counts, N, M, K, R, seed=4, min_iter=5, max_iter=30,
                                delay_fit_theta=2)
    np.random.seed(4)
    db = DeviceBatch(counts, _lib.KIND_VIREO, K, R)
    singles[0]._set_device_prior(db)
    info = db.info()
    assert info["lds_variant"] == lds and info["lds_cell"] == lds
    mu, sm = np.linspace(0.01, 0.99, 3)[None, :], np.full((1, 3), 50.0)
    for r in range(R):
        db.set_restart(r, np.random.rand(M, K), np.random.rand(N, K, 3), mu, sm, raw=True)
    traces, its, flags = db.fit(30, 5, 1e-2, 2)
    one = DeviceModel(counts, _lib.KIND_VIREO, K)
    assert len({len(m.ELBO_) for m in singles}) > 1 or R < 3     # restarts stop at different times
    for r, m in enumerate(singles):
        elbo = traces[r][:its[r]] + counts.binom_const()
        db.copy_to(one, r)
        ID, GT, bmu, bsm = one.get_state()
        if lds:
            assert np.array_equal(elbo, m.ELBO_), r
            assert np.array_equal(ID, m.ID_prob) and np.array_equal(GT, m.GT_prob)
            assert np.array_equal(bmu, m.beta_mu) and np.array_equal(bsm, m.beta_sum)
        else:
            assert len(elbo) == len(m.ELBO_)
            close(elbo, m.ELBO_, rtol=1e-9)
            close(ID, m.ID_prob, rtol=1e-6, atol=1e-12)
            close(bmu, m.beta_mu, rtol=1e-9)


@pytest.mark.parametrize("lds", [True, False])
def test_restart_batch_in_ase_mode(va, monkeypatch, lds):
    """R restarts in one model with ASE_mode=True (one theta row per variant: psi is
    [R][3][N][T], the largest per-restart table -- an indexing slip between restarts would show
    here): every restart equals its independent fit, also the LAST one of the batch."""
    from vireo_amd import _lib
    from vireo_amd.counts import DeviceCounts
    from vireo_amd.engine import DeviceBatch, DeviceModel
    monkeypatch.setenv("VIREO_LDS", "1" if lds else "0")
    K, R = 4, 4
    AD, DP = O.synth_donor(1200, 700, K, 0.05, seed=9)
    counts = DeviceCounts(AD, DP)
    N, M = AD.shape
    np.random.seed(6)
    singles = []
    for _ in range(R):
        m = va.Vireo(n_var=N, n_cell=M, n_donor=K, ASE_mode=True)
        m.fit(counts, None, min_iter=5, max_iter=25, delay_fit_theta=2, verbose=False)
        singles.append(m)
    np.random.seed(6)
    db = DeviceBatch(counts, _lib.KIND_VIREO, K, R, ase_mode=True)
    singles[0]._set_device_prior(db)
    mu = np.ones((N, 3)) * np.linspace(0.01, 0.99, 3)[None, :]
    sm = np.full((N, 3), 50.0)
    for r in range(R):
        db.set_restart(r, np.random.rand(M, K), np.random.rand(N, K, 3), mu, sm, raw=True)
    traces, its, _ = db.fit(25, 5, 1e-2, 2)
    one = DeviceModel(counts, _lib.KIND_VIREO, K, ase_mode=True)
    for r, m in enumerate(singles):
        elbo = traces[r][:its[r]] + counts.binom_const()
        db.copy_to(one, r)
        ID, GT, bmu, bsm = one.get_state()
        assert bmu.shape == (N, 3) and len(elbo) == len(m.ELBO_)
        if lds:
            assert np.array_equal(elbo, m.ELBO_), r
            assert np.array_equal(ID, m.ID_prob) and np.array_equal(GT, m.GT_prob)
            assert np.array_equal(bmu, m.beta_mu) and np.array_equal(bsm, m.beta_sum)
        else:
            close(elbo, m.ELBO_, rtol=1e-9)
            close(ID, m.ID_prob, rtol=1e-6, atol=1e-12)
            close(bmu, m.beta_mu, rtol=1e-9)


@pytest.mark.parametrize("batch", [2, 3, 4])
def test_wrap_with_restart_batches(va, monkeypatch, capsys, batch):
    """vireo_wrap with its restarts fitted `batch` at a time reproduces the reference's run
    (golden c1_wrap_seed2_init4) and the one-at-a-time run"""
    import sys
    wrap_mod = sys.modules["vireo_amd.vireo_wrap"]       # (the package re-exports the function)
    AD, DP = gold.c1()
    g = gold.load("c1_wrap_seed2_init4")
    res = {}
    for b in (1, batch):
        monkeypatch.setenv("VIREO_RESTART_BATCH", str(b))
        res[b] = va.vireo_wrap(AD, DP, n_donor=4, n_init=4, random_seed=2)
        assert wrap_mod.LAST_SEARCH["batch"] == b
    capsys.readouterr()
    close(res[batch]["LB_list"], g["LB_list"], rtol=1e-9)
    close(res[batch]["ID_prob"], g["ID_prob"])
    close(res[batch]["LB_list"], res[1]["LB_list"], rtol=1e-9)
    close(res[batch]["ID_prob"], res[1]["ID_prob"], rtol=1e-6, atol=1e-12)
    assert np.array_equal(np.argmax(res[batch]["ID_prob"], 1), np.argmax(res[1]["ID_prob"], 1))


@pytest.mark.parametrize("depth,want", [(1.0, (1, 3)), (50.0, (0, 0)), (3000.0, (1, 3))])
def test_stream_form_follows_count_depth(va, monkeypatch, depth, want):
    """AD/BD words (cell form 1, variant form 3 = virtual rows) on shallow data, one (ad, dp) pair word per
    entry where the counts are deep enough to need several AD/BD words each (clone mode) but
    still fit 11 bits, AD/BD words again beyond that; every choice matches the oracle"""
    from vireo_amd import _lib
    from vireo_amd.counts import DeviceCounts
    from vireo_amd.engine import DeviceModel
    monkeypatch.setenv("VIREO_LDS", "1")
    rng = np.random.default_rng(int(depth))
    N, M, K = 600, 900, 5
    mask = rng.random((N, M)) < 0.05
    DPd = (1 + rng.poisson(depth, (N, M))) * mask
    ADd = rng.binomial(DPd, 0.3)
    AD, DP = csc_matrix(ADd), csc_matrix(DPd)
    counts = DeviceCounts(AD, DP)
    dm = DeviceModel(counts, _lib.KIND_VIREO, K)
    info = dm.info()
    assert (info["cell_form"], info["var_form"]) == want and info["lds_cell"] and info["lds_variant"]
    np.random.seed(3)
    m = va.Vireo(n_var=N, n_cell=M, n_donor=K)
    st = O.vireo_new(M, N, K, ID_prob_init=m.ID_prob.copy(), GT_prob_init=m.GT_prob.copy())
    m.fit(counts, None, min_iter=3, max_iter=8, verbose=False)
    O.vireo_fit(st, AD, DP, max_iter=8, min_iter=3)
    close(m.ELBO_, st.ELBO_, rtol=1e-9)
    close(m.ID_prob, st.ID_prob, rtol=1e-6, atol=1e-12)


@pytest.mark.parametrize("build", ["device", "host"])
def test_deep_counts_with_one_beyond_pair_words(va, monkeypatch, build):
    """deep counts make the estimate choose (ad, dp) pair words, ONE count >= 2048
    does not fit them, so the builder falls back to AD/BD words -- and must then also build the cell
    stream with the tile height of THAT form (96 rows per wave, not the pair form's 64; more than
    two slabs of variants so that the short tile is not chosen either way).  Both builders, same
    streams, the fit against the oracle."""
    from vireo_amd import _lib
    from vireo_amd.counts import DeviceCounts
    from vireo_amd.engine import DeviceModel
    monkeypatch.setenv("VIREO_LDS", "1")
    rng = np.random.default_rng(77)
    N, M, K = 1300, 700, 6
    mask = rng.random((N, M)) < 0.04
    DPd = (1 + rng.poisson(40.0, (N, M))) * mask
    DPd[1111, 5] = 2048
    DPd[17, 600] = 5000
    ADd = rng.binomial(DPd, rng.choice([0.02, 0.5, 0.97], (N, 1)))
    AD, DP = csc_matrix(ADd), csc_matrix(DPd)
    monkeypatch.setenv("VIREO_BUILD", build)
    counts = DeviceCounts(AD, DP)
    monkeypatch.setenv("VIREO_BUILD", "host")
    host = DeviceCounts(AD, DP)
    dh, dd = host.digest(), counts.digest()
    for k in (0, 1, 2, 3, 4, 6, 7, 8, 9, 10):
        assert dh[k] == dd[k], "digest %d differs" % k
    info = DeviceModel(counts, _lib.KIND_VIREO, K).info()
    assert (info["cell_form"], info["var_form"]) == (1, 3) and info["lds_cell"] and info["lds_variant"]
    np.random.seed(3)
    m = va.Vireo(n_var=N, n_cell=M, n_donor=K)
    st = O.vireo_new(M, N, K, ID_prob_init=m.ID_prob.copy(), GT_prob_init=m.GT_prob.copy())
    m.fit(counts, None, min_iter=3, max_iter=8, verbose=False)
    O.vireo_fit(st, AD, DP, max_iter=8, min_iter=3)
    close(m.ELBO_, st.ELBO_, rtol=1e-9)
    close(m.ID_prob, st.ID_prob, rtol=1e-6, atol=1e-12)
    close(m.GT_prob, st.GT_prob, rtol=1e-6, atol=1e-12)


@pytest.mark.parametrize("data", ["mito", "clone"])
def test_bmm_fit_with_batched_initialisations(va, monkeypatch, capsys, data):
    """BinomMixtureVB.fit runs its n_init short fits side by side in one device model
    (bmm_model.py:241-252 runs them one after the other): same draws, same ELBO_inits, same
    winner and final fit as one at a time"""
    from vireo_amd import restarts
    if data == "mito":
        AD, DP = gold.mito()
        K, kw = 3, dict(n_init=7, min_iter=10, max_iter_pre=40, random_seed=3)
    else:
        AD, DP = O.synth_clone(60, 3000, 5, seed=2)
        K, kw = 5, dict(n_init=6, min_iter=5, max_iter_pre=30, random_seed=4)
    fits = {}
    for batch in ("1", "0"):                   # one at a time; automatic packing
        monkeypatch.setenv("VIREO_RESTART_BATCH", batch)
        b = va.BinomMixtureVB(n_var=AD.shape[0], n_cell=AD.shape[1], n_donor=K)
        b.fit(AD, DP, **kw)
        fits[batch] = (b, capsys.readouterr().out)
    assert restarts.restart_batch(K, kw["n_init"], 10 ** 5) > 1
    one, packed = fits["1"][0], fits["0"][0]
    assert fits["1"][1] == fits["0"][1]        # the reference's warnings, in the same order
    close(packed.ELBO_inits, one.ELBO_inits, rtol=1e-9)
    assert np.argmax(packed.ELBO_inits) == np.argmax(one.ELBO_inits)
    assert len(packed.ELBO_iters) == len(one.ELBO_iters)
    close(packed.ELBO_iters, one.ELBO_iters, rtol=1e-9)
    close(packed.ID_prob, one.ID_prob, rtol=1e-6, atol=1e-12)
    close(packed.beta_mu, one.beta_mu, rtol=1e-8)


def test_staged_upload_and_pipelined_polls(va, monkeypatch):
    """Round 4 (vireo_wrap turnaround): (1) raw draws uploaded into a staging buffer -- from a
    second thread, while the model is fitting -- and normalised from there give the bits of
    vrx_model_set_state_raw; (2) vrx_model_fit with its polls pipelined (the next batch of
    iterations is enqueued before the control words of the last one are read) returns what the
    wait-then-enqueue loop returns: trace, iteration count, flags, state -- also when the stop rule
    fires in the middle of a batch."""
    import threading
    from vireo_amd import _lib
    from vireo_amd.counts import DeviceCounts
    from vireo_amd.engine import DeviceModel
    AD, DP = O.synth_donor(1800, 1100, 5, 0.04, seed=31)
    counts = DeviceCounts(AD, DP)
    N, M = AD.shape
    K = 5
    rng = np.random.default_rng(3)
    draws = [(rng.random((M, K)), rng.random((N, K, 3))) for _ in range(3)]
    mu, sm = np.linspace(0.01, 0.99, 3)[None, :], np.full((1, 3), 50.0)
    tmpl = va.Vireo(n_var=N, n_cell=M, n_donor=K, ID_prob_init=np.ones((M, K)), GT_prob_init=np.ones((N, K, 3)))

    def fitted(dm, pipeline):
        monkeypatch.setenv("VIREO_FIT_PIPELINE", pipeline)
        tr, it, fl = dm.fit(60, 5, 1e-2, 2)
        return tr, it, fl, dm.get_state()

    ref = DeviceModel(counts, _lib.KIND_VIREO, K)
    tmpl._set_device_prior(ref)
    want = []
    for ID_raw, GT_raw in draws:
        ref.set_state_raw(ID_raw, GT_raw, mu, sm)
        want.append(fitted(ref, "0"))
    assert len({w[1] for w in want}) > 1 and all(w[1] < 59 for w in want)   # stops at different iterations

    dm = DeviceModel(counts, _lib.KIND_VIREO, K)
    tmpl._set_device_prior(dm)
    dm.stage_reserve()
    dm.stage_raw(0, *draws[0])
    for i in range(3):
        dm.set_state_staged(i & 1, mu, sm)
        th = None
        if i + 1 < 3:       # the next restart's draws go up while this one fits
            th = threading.Thread(target=dm.stage_raw, args=((i + 1) & 1,) + draws[i + 1])
            th.start()
        got = fitted(dm, "1")
        if th is not None:
            th.join()
        assert got[1] == want[i][1] and got[2] == want[i][2]
        assert np.array_equal(got[0], want[i][0])
        for a, b in zip(got[3], want[i][3]):
            assert np.array_equal(a, b)


def test_max_iter_beyond_the_initial_trace_capacity(va):
    """the reference takes any max_iter (its trace is np.zeros(max_iter), vireo_model.py:251); the
    device-side trace grows on demand (it was capped at 65 536)"""
    AD, DP = gold.c1()
    np.random.seed(2)
    a = va.Vireo(n_var=AD.shape[0], n_cell=AD.shape[1], n_donor=4)
    np.random.seed(2)
    b = va.Vireo(n_var=AD.shape[0], n_cell=AD.shape[1], n_donor=4)
    a.fit(AD, DP, min_iter=5, max_iter=200, verbose=False)
    b.fit(AD, DP, min_iter=5, max_iter=70000, verbose=False)
    assert np.array_equal(a.ELBO_, b.ELBO_) and np.array_equal(a.ID_prob, b.ID_prob)


def test_failed_trace_growth_leaves_the_model_usable(va):
    """growing the device-side ELBO trace allocates into a temporary and swaps it in, so
    an allocation that fails returns an error code and leaves the old trace and its capacity in
    place.  Here: most of the HBM is taken by a blocker allocation, then a fit asks for a trace of
    2^31 - 2 iterations x 16 restarts (256 GiB) -> VrxError; the next fit with a sane max_iter
    runs and repeats the result from before."""
    import ctypes as C
    from vireo_amd import _lib
    from vireo_amd.engine import DeviceBatch
    AD, DP = gold.c1()
    counts = va.DeviceCounts(AD, DP)
    R = 16
    db = DeviceBatch(counts, _lib.KIND_VIREO, 4, R)
    db.set_prior(np.full((1, 4), 0.25), np.full((1, 4, 3), 1.0 / 3), np.array([[0.5, 25.0, 49.5]]),
                 np.array([[49.5, 25.0, 0.5]]))
    mu, sm = np.linspace(0.01, 0.99, 3)[None, :], np.full((1, 3), 50.0)

    def fill():
        rng = np.random.default_rng(0)
        for r in range(R):
            db.set_restart(r, rng.random((AD.shape[1], 4)), rng.random((AD.shape[0], 4, 3)), mu, sm, raw=True)

    fill()
    tr0, it0, _ = db.fit(30, 5, 1e-2, 0)
    hip = C.CDLL("libamdhip64.so")
    blocker = C.c_void_p()
    if hip.hipMalloc(C.byref(blocker), C.c_size_t(100 << 30)) != 0:
        pytest.skip("could not take 100 GiB of HBM as a blocker")
    try:
        with pytest.raises(_lib.VrxError):
            _huge_fit(db, R)
    finally:
        hip.hipFree(blocker)
    fill()
    tr1, it1, _ = db.fit(30, 5, 1e-2, 0)
    assert np.array_equal(it0, it1)
    for r in range(R):
        assert np.array_equal(tr0[r], tr1[r])
    db.close()


def _huge_fit(db, R):
    """vrx_model_fit with max_iter = 2^31 - 2 and a dummy output (the call fails in ensure_trace,
    before any kernel runs or anything is written)"""
    import ctypes as C
    from vireo_amd import _lib
    out = np.empty(R * 8)
    its = np.zeros(R, dtype=np.int32)
    _lib.check(_lib.lib().vrx_model_fit(db._h, 2 ** 31 - 2, 5, 1e-2, 0, _lib.dptr(out),
                                        its.ctypes.data_as(C.POINTER(C.c_int32)), None))


def test_fit_loop_and_stepwise_api_agree_on_split_rows(va, monkeypatch):
    """inside ``fit`` the terms of a row that the tiled stream cut into pieces are folded
    by ``vrx_fold_split`` (eight lanes per row and column, then a butterfly) and the consumers sum
    the partial arrays themselves; the step-wise API (``update_theta_size`` / ``update_GT_prob`` /
    ``update_ID_prob`` / ``get_ELBO`` -> vrx_model_step) resolves the same sums sequentially
    (``vrx_s_from_virtual`` / ``vrx_sum_pieces``).  Different association, same terms: on
    heavy-tailed data with rows in pieces the two must agree to rounding, iteration by iteration:
    1e-12 relative on the ELBO, 1e-10 on theta, 1e-7 on the posteriors (observed 3e-9: six iterations
    carry a 1e-11 difference of the logits into the small posteriors)."""
    from vireo_amd import _lib, synth
    from vireo_amd.counts import DeviceCounts
    from vireo_amd.engine import DeviceModel
    monkeypatch.setenv("VIREO_LDS", "1")
    w = synth.donor_workload(3000, 1500, 6, 0.04, seed=3, skew=(1.5, 1.0))
    counts = DeviceCounts.from_merged(w["shape"], w["colptr"], w["rowidx"], w["ad"], w["dp"])
    N, M = w["shape"]
    info = DeviceModel(counts, _lib.KIND_VIREO, 6).info()
    assert info["lds_cell"] and info["lds_variant"]
    assert info["extra_pieces_cell"] > 0 and info["extra_pieces_variant"] > 0
    np.random.seed(9)
    a = va.Vireo(n_var=N, n_cell=M, n_donor=6)
    np.random.seed(9)
    b = va.Vireo(n_var=N, n_cell=M, n_donor=6)
    n_it = 6
    a.fit(counts, None, min_iter=n_it + 5, max_iter=n_it, verbose=False)      # runs all n_it iterations
    const = counts.binom_const()
    steps = []
    for _ in range(n_it):
        b.update_theta_size(counts, None)
        b.update_GT_prob(counts, None)
        L = b.update_ID_prob(counts, None)
        steps.append(b.get_ELBO(L, counts, None))
    assert len(a.ELBO_) == n_it - 1                                            # ELBO[:it], vireo_model.py:276
    np.testing.assert_allclose(a.ELBO_ - const, steps[:n_it - 1], rtol=1e-12)
    np.testing.assert_allclose(a.ID_prob, b.ID_prob, rtol=1e-7, atol=1e-300)
    np.testing.assert_allclose(a.GT_prob, b.GT_prob, rtol=1e-7, atol=1e-300)
    np.testing.assert_allclose(a.beta_mu, b.beta_mu, rtol=1e-10)       # (observed 3e-12)
    np.testing.assert_allclose(a.beta_sum, b.beta_sum, rtol=1e-10)


def test_elbo_riding_in_clone_mode_changes_nothing(va, monkeypatch):
    """the same for ``BinomMixtureVB``: the rider sits in vrx_bmm_theta, which itself writes the
    KL_theta partials -- two halves of the buffer, the rider reads the one the launch does not write.
    Default: fits with min_iter >= 12 (bmm_model.py:178 runs 20) at any size.  Bitwise equal with
    VIREO_ELBO_RIDE=0 / 1: single fits that stop by the rule, at max_iter and in the middle of a poll
    batch; ``fit(n_init=...)`` with batched initialisations; the LDS-resident passes with the fused
    range sum (16 lanes per element: a different number of KL partials than the first, non-fused call)."""
    from vireo_amd import synth
    AD, DP = gold.mito()
    cAD, cDP = synth.clone_workload(60, 3000, 5, seed=2)
    out = {}
    for ride in ("1", "0"):
        monkeypatch.setenv("VIREO_ELBO_RIDE", ride)
        res = []
        for (A, D, K, lds) in ((AD, DP, 3, "0"), (cAD, cDP, 5, "0"), (cAD, cDP, 5, "1")):
            monkeypatch.setenv("VIREO_LDS", lds)
            for (mx, mn, eps) in ((200, 20, 1e-2), (25, 30, 1e-2), (40, 3, 10.0), (1, 20, 1e-2)):
                np.random.seed(3)
                b = va.BinomMixtureVB(n_var=A.shape[0], n_cell=A.shape[1], n_donor=K)
                b._fit_BV(A, D, max_iter=mx, min_iter=mn, epsilon_conv=eps, verbose=False)
                b._fit_BV(A, D, max_iter=30, min_iter=20, verbose=False)          # continues from the state
                res.append((tuple(b.ELBO_iters), b.ID_prob.tobytes(), b.beta_mu.tobytes(), b.beta_sum.tobytes()))
            b = va.BinomMixtureVB(n_var=A.shape[0], n_cell=A.shape[1], n_donor=K)
            b.fit(A, D, n_init=7, min_iter=20, max_iter_pre=40, random_seed=5, verbose=False)
            res.append((tuple(b.ELBO_iters), tuple(b.ELBO_inits), b.ID_prob.tobytes(), b.beta_mu.tobytes()))
            # the step-wise API reads the KL_theta partials of the half written last
            np.random.seed(3)
            b = va.BinomMixtureVB(n_var=A.shape[0], n_cell=A.shape[1], n_donor=K)
            b.update_theta_size(A, D)
            res.append((b.get_ELBO(A, D), b.beta_mu.tobytes()))
        out[ride] = res
    assert out["1"] == out["0"]


def test_clone_mode_state_after_a_stop_in_the_middle_of_a_poll_batch(va, monkeypatch):
    """vrx_bmm_theta writes model state (beta_mu, beta_sum, W), so an ELBO riding in
    it must never be one whose stop rule can fire -- a stop found by the rider would leave theta one
    update AHEAD of the state ``_fit_BV`` breaks out with (bmm_model.py:190-199).  The state is compared
    DIRECTLY after a single ``_fit_BV`` (no second fit that would recompute theta from ID_prob), for
    fits whose rule fires in the middle of a poll batch (iterations min_iter + 2 .. min_iter + 4 of
    VIREO_FIT_BATCH = 4: not the batch's last, whose ELBO is never deferred): bitwise equal with
    VIREO_ELBO_RIDE = 1 and 0, and equal to the oracle's state."""
    from vireo_amd import synth
    mAD, mDP = gold.mito()
    cases = [("mito", 3, 5, 2), ("mito", 4, 3, 2), ("mito", 4, 7, 5), ("mito", 5, 3, 2),      # (data, K, seed, min_iter)
             ((40, 2000, 3, 5), 8, 5, 2), ((60, 3000, 3, 2), 6, 2, 2), ((40, 4000, 6, 11), 6, 11, 3)]
    # (the oracle stops these at iterations 5, 5, 7, 10, 72, 4, 6: every one inside a poll batch)
    mid_batch = 0
    for data, k, seed, mn in cases:
        A, D = (mAD, mDP) if data == "mito" else synth.clone_workload(data[0], data[1], data[2], seed=data[3])
        n, m = A.shape
        res = {}
        for ride in ("1", "0"):
            monkeypatch.setenv("VIREO_ELBO_RIDE", ride)
            np.random.seed(seed)
            b = va.BinomMixtureVB(n_var=n, n_cell=m, n_donor=k)
            b._fit_BV(A, D, max_iter=200, min_iter=mn, epsilon_conv=1e-2, verbose=False)
            res[ride] = b
        a, b = res["1"], res["0"]
        assert len(a.ELBO_iters) == len(b.ELBO_iters)
        for name in ("ELBO_iters", "ID_prob", "beta_mu", "beta_sum"):
            assert np.array_equal(getattr(a, name), getattr(b, name)), (name, data, seed)
        it = len(a.ELBO_iters)                  # the iteration the rule fired at (ELBO[:it] is kept)
        first_end = max(mn + 2, 4) - 1          # last iteration of the first batch
        if it > first_end and (it - first_end) % 4 != 0 and it < 199:
            mid_batch += 1
        np.random.seed(seed)
        ref = O.bmm_new(m, n, k)
        O.bmm_fit_vb(ref, A, D, min_iter=mn, max_iter=200)
        assert len(ref.ELBO_iters) == it
        close(a.ELBO_iters, ref.ELBO_iters)
        close(a.beta_mu, ref.beta_mu)
        close(a.beta_sum, ref.beta_sum)
        close(a.ID_prob, ref.ID_prob)
    assert mid_batch >= 5, "the cases of this list no longer stop in the middle of a poll batch"


@pytest.mark.parametrize("case", ["c1", "c2_lds", "batch"])
def test_elbo_riding_in_the_next_theta_kernel_changes_nothing(va, monkeypatch, case):
    """Launch-bound problems: the ELBO + stop rule of an iteration are finalised by an extra block
    of the next iteration's vrx_theta_partial (VIREO_ELBO_RIDE; default on where nnz x columns <
    2^24) instead of by vrx_elbo_final between the iterations.  Same code on the same partial sums:
    traces, iteration counts (the stop firing in the middle of a poll batch, at its end, never),
    warning flags and states must be bitwise those of VIREO_ELBO_RIDE=0 -- for single models (demo
    data; c2 size on the LDS-resident passes), warm restarts, and a restart batch whose members
    stop at different iterations."""
    from vireo_amd import _lib, synth
    from vireo_amd.counts import DeviceCounts
    from vireo_amd.engine import DeviceBatch
    if case == "c1":
        AD, DP = gold.c1()
        counts, K = va.DeviceCounts(AD, DP), 4
        N, M = AD.shape
    else:
        if case == "c2_lds":
            monkeypatch.setenv("VIREO_LDS", "1")
        Nn, Mm, K, dens = synth.CONFIGS["c2"]
        w = synth.donor_workload(Nn, Mm, K, dens, seed=0)
        counts = DeviceCounts.from_merged(w["shape"], w["colptr"], w["rowidx"], w["ad"], w["dp"])
        N, M = w["shape"]
    out = {}
    for ride in ("1", "0"):
        monkeypatch.setenv("VIREO_ELBO_RIDE", ride)
        res = []
        if case == "batch":
            R = 5
            db = DeviceBatch(counts, _lib.KIND_VIREO, K, R)
            db.set_prior(np.full((1, K), 1.0 / K), np.full((1, K, 3), 1.0 / 3), np.array([[0.5, 25.0, 49.5]]),
                         np.array([[49.5, 25.0, 0.5]]))
            rng = np.random.default_rng(3)
            mu, sm = np.linspace(0.01, 0.99, 3)[None, :], np.full((1, 3), 50.0)
            for r in range(R):
                db.set_restart(r, rng.random((M, K)), rng.random((N, K, 3)), mu, sm, raw=True)
            for (mx, mn, dl) in ((40, 5, 3), (9, 5, 0)):
                tr, it, fl = db.fit(mx, mn, 1e-2, dl)
                res.append((list(map(tuple, tr)), tuple(it), tuple(fl)))
            db.close()
        else:
            for (mx, mn, dl, eps) in ((200, 5, 3, 1e-2), (20, 5, 0, 1e-2), (7, 2, 1, 1e3), (6, 9, 2, 1e-2), (1, 5, 0, 1e-2)):
                np.random.seed(6)
                m = va.Vireo(n_var=N, n_cell=M, n_donor=K)
                m.fit(counts, None, max_iter=mx, min_iter=mn, delay_fit_theta=dl, epsilon_conv=eps, verbose=False)
                m.fit(counts, None, max_iter=30, min_iter=3, verbose=False)       # warm restart
                res.append((tuple(m.ELBO_), m.ID_prob.tobytes(), m.GT_prob.tobytes(), m.beta_mu.tobytes(),
                            m.beta_sum.tobytes()))
        out[ride] = res
    assert out["1"] == out["0"]


def test_balanced_slabs_small_problem_vs_oracle_and_default_build(va, monkeypatch):
    """Round 6: *balanced slabs* (vrx_problem_create2 + VRX_PROBLEM_BALANCED) -- which contracted rows share a
    slab is chosen per row tile (host greedy), the entries are relabelled and re-sorted on the device, the
    kernel stages a slab through the tile's list.  On a problem small enough for the oracle, with the device
    builder and the LDS-resident passes forced and several slabs per orientation: both streams are
    balanced, fewer stream slots than the default build, a whole fit within 1e-5 of the oracle (identical
    iteration count and assignments), within summation-order distance of the default build, and the build
    is deterministic (two builds: the same stream checksums; two fits: the same bits)."""
    from vireo_amd.counts import DeviceCounts
    from vireo_amd.engine import DeviceModel
    from vireo_amd import _lib
    monkeypatch.setenv("VIREO_LDS", "1")
    monkeypatch.setenv("VIREO_BUILD", "device")
    monkeypatch.setenv("VIREO_LDS_SPLIT_X10", "60")        # (no row pieces here: the next test has them)
    monkeypatch.setenv("VIREO_BALANCE_CHECK", "1")         # device greedy == host greedy, or the build fails
    AD, DP = O.synth_donor(2600, 2300, 6, 0.03, seed=3)          # 6 slabs of variants, 3 of double rows of cells
    N, M = AD.shape

    def build(balance):
        return DeviceCounts(AD, DP, balance=balance)

    cb, cd = build(True), build(False)
    ib = cb.build_info()
    assert ib["balanced_variant"] and ib["balanced_cell"] and ib["device_built"] and ib["balance_seconds"] > 0
    assert not cd.build_info()["balanced_cell"]
    kb, kd = DeviceModel(cb, _lib.KIND_VIREO, 6).info(), DeviceModel(cd, _lib.KIND_VIREO, 6).info()
    assert kb["lds_variant"] and kb["lds_cell"] and (kb["cell_form"], kb["var_form"]) == (1, 3)
    assert kb["pad_cell"] < kd["pad_cell"] and kb["pad_variant"] < kd["pad_variant"]
    print("stream slots per non-zero: balanced %.3f / %.3f, default %.3f / %.3f (variant / cell)"
          % (kb["pad_variant"], kb["pad_cell"], kd["pad_variant"], kd["pad_cell"]))
    assert build(True).digest() == cb.digest()
    # the same stream by every route: the greedy on host threads (the specification; by default it runs on the
    # device, and VIREO_BALANCE_CHECK=1 -- set above -- makes every build compare the two bit for bit), with
    # the cell orientation's share beside the upload or inside build_tiled, the orientations one after the other
    monkeypatch.setenv("VIREO_BALANCE_GREEDY", "host")
    assert build(True).digest() == cb.digest()
    monkeypatch.setenv("VIREO_BALANCE_EARLY", "0")
    assert build(True).digest() == cb.digest()
    monkeypatch.delenv("VIREO_BALANCE_EARLY")
    monkeypatch.delenv("VIREO_BALANCE_GREEDY")
    monkeypatch.setenv("VIREO_BUILD_CONCURRENT", "0")
    assert build(True).digest() == cb.digest()
    monkeypatch.delenv("VIREO_BUILD_CONCURRENT")

    def fit(counts, **kw):
        np.random.seed(4)
        m = va.Vireo(n_var=N, n_cell=M, n_donor=6, **kw)
        m.fit(counts, None, min_iter=5, max_iter=25, delay_fit_theta=2, verbose=False)
        return m

    for kw in (dict(), dict(ASE_mode=True)):
        a, a2, d = fit(cb, **kw), fit(cb, **kw), fit(cd, **kw)
        for name in ("ELBO_", "ID_prob", "GT_prob", "beta_mu", "beta_sum"):
            assert np.array_equal(getattr(a, name), getattr(a2, name)), name
        np.random.seed(4)
        ref = O.vireo_new(M, N, 6, **kw)
        O.vireo_fit(ref, AD, DP, min_iter=5, max_iter=25, delay_fit_theta=2)
        assert len(a.ELBO_) == len(ref.ELBO_) == len(d.ELBO_)
        close(a.ELBO_, ref.ELBO_)
        close(a.ID_prob, ref.ID_prob)
        close(a.GT_prob, ref.GT_prob)
        close(a.beta_mu, ref.beta_mu)
        close(a.beta_sum, ref.beta_sum)
        assert np.array_equal(a.ID_prob.argmax(1), ref.ID_prob.argmax(1))
        np.testing.assert_allclose(a.ELBO_, d.ELBO_, rtol=1e-10)
        np.testing.assert_allclose(a.ID_prob, d.ID_prob, rtol=1e-7, atol=1e-290)
    # the doublet step and a 40-column operand (column blocks: the list is not used there) on the balanced problem
    m = fit(cb)
    from vireo_amd.vireo_doublet import predict_doublet
    st = O.vireo_new(M, N, 6, ID_prob_init=m.ID_prob, GT_prob_init=m.GT_prob, beta_mu_init=m.beta_mu.copy(),
                     beta_sum_init=m.beta_sum.copy())
    st.ID_prob, st.GT_prob = m.ID_prob.copy(), m.GT_prob.copy()
    dbl_ref, sing_ref, llr_ref = O.vireo_doublet(st, AD, DP)
    dbl, sing, llr = predict_doublet(m, cb, None)
    close(dbl, dbl_ref)
    close(sing, sing_ref)
    close(llr, llr_ref, atol=1e-9)


def test_balanced_slabs_with_rows_cut_into_pieces(va, monkeypatch):
    """Balanced slabs on a heavy-tailed problem: long rows are cut into pieces (the default rule), the pieces
    of one row may sit in different row tiles, and each piece is relabelled by ITS tile's permutation
    (vrx_build_pieces: the pieces become rows of their own before the relabel).  Both streams split rows and
    are balanced, a whole fit is within 1e-5 of the oracle with identical iteration count and assignments,
    within summation-order distance of the default build, and deterministic."""
    from vireo_amd.counts import DeviceCounts
    from vireo_amd.engine import DeviceModel
    from vireo_amd import _lib
    monkeypatch.setenv("VIREO_LDS", "1")
    monkeypatch.setenv("VIREO_BUILD", "device")
    monkeypatch.setenv("VIREO_BALANCE_CHECK", "1")         # device greedy == host greedy, or the build fails
    rng = np.random.default_rng(11)
    N, M, K = 2600, 2300, 6
    cov_v = np.exp(rng.normal(0.0, 1.2, N))[:, None]
    cov_c = np.exp(rng.normal(0.0, 1.0, M))[None, :]
    p = np.minimum(0.9, 0.025 * cov_v * cov_c / (cov_v.mean() * cov_c.mean()))
    dp = (rng.random((N, M)) < p) * (1 + rng.poisson(1.5, (N, M)))
    z = rng.integers(0, K, M)
    gt = rng.integers(0, 3, (N, K))
    ad = rng.binomial(dp, np.array([0.01, 0.5, 0.99])[gt][:, z])
    AD, DP = csc_matrix(ad), csc_matrix(dp)
    cb, cd = DeviceCounts(AD, DP, balance=True), DeviceCounts(AD, DP, balance=False)
    ib = cb.build_info()
    assert ib["balanced_variant"] and ib["balanced_cell"] and ib["device_built"]
    kb, kd = DeviceModel(cb, _lib.KIND_VIREO, K).info(), DeviceModel(cd, _lib.KIND_VIREO, K).info()
    assert kb["lds_variant"] and kb["lds_cell"]
    assert kb["extra_pieces_variant"] > 0 and kb["extra_pieces_cell"] > 0, kb
    assert kb["pad_cell"] < kd["pad_cell"] and kb["pad_variant"] < kd["pad_variant"]
    print("rows cut into pieces: +%d / +%d; stream slots per non-zero: balanced %.3f / %.3f, default %.3f / %.3f"
          % (kb["extra_pieces_variant"], kb["extra_pieces_cell"], kb["pad_variant"], kb["pad_cell"],
             kd["pad_variant"], kd["pad_cell"]))
    assert DeviceCounts(AD, DP, balance=True).digest() == cb.digest()
    monkeypatch.setenv("VIREO_BALANCE_SPLIT", "0")          # the knob that keeps such streams on the default build
    assert not DeviceCounts(AD, DP, balance=True).build_info()["balanced_cell"]
    monkeypatch.delenv("VIREO_BALANCE_SPLIT")

    def fit(counts, **kw):
        np.random.seed(4)
        m = va.Vireo(n_var=N, n_cell=M, n_donor=K, **kw)
        m.fit(counts, None, min_iter=5, max_iter=25, delay_fit_theta=2, verbose=False)
        return m

    for kw in (dict(), dict(ASE_mode=True)):
        a, a2, d = fit(cb, **kw), fit(cb, **kw), fit(cd, **kw)
        for name in ("ELBO_", "ID_prob", "GT_prob", "beta_mu", "beta_sum"):
            assert np.array_equal(getattr(a, name), getattr(a2, name)), name
        np.random.seed(4)
        ref = O.vireo_new(M, N, K, **kw)
        O.vireo_fit(ref, AD, DP, min_iter=5, max_iter=25, delay_fit_theta=2)
        assert len(a.ELBO_) == len(ref.ELBO_) == len(d.ELBO_)
        close(a.ELBO_, ref.ELBO_)
        close(a.ID_prob, ref.ID_prob)
        close(a.GT_prob, ref.GT_prob)
        close(a.beta_mu, ref.beta_mu)
        close(a.beta_sum, ref.beta_sum)
        sure = np.sort(ref.ID_prob, axis=1)[:, -1] - np.sort(ref.ID_prob, axis=1)[:, -2] > 1e-6   # (cells without
        assert sure.sum() > 0.9 * M                                    # reads keep the flat prior: a tie)
        assert np.array_equal(a.ID_prob.argmax(1)[sure], ref.ID_prob.argmax(1)[sure])
        np.testing.assert_allclose(a.ELBO_, d.ELBO_, rtol=1e-10)
        np.testing.assert_allclose(a.ID_prob, d.ID_prob, rtol=1e-7, atol=1e-290)
