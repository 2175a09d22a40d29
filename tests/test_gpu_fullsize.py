"""Full-size GPU checks at BASELINE.json's sizes (configs[2] and configs[4]).

The oracle cannot run whole fits at these sizes in test time, so parity is established by
(1) size-independent properties of the fitted state, (2) run-to-run bitwise determinism,
(3) recovery of the planted structure of the synthetic data, and (4) ONE oracle iteration
started from the GPU's fitted state, compared with one GPU iteration from the same state
(rtol 1e-5, identical assignments) -- at a fitted state the posteriors are well separated,
so the comparison is not dominated by near-ties as it is after a random start.
"""
import numpy as np
import pytest

from oracle import vireo_oracle as O
from tests import gold

pytestmark = pytest.mark.gpu
RTOL = 1e-5


@pytest.fixture(scope="module")
def va():
    import vireo_amd
    from vireo_amd import _lib
    _lib.require_gpu()
    return vireo_amd


def test_config3_fit_properties_and_one_step_parity(va):
    from vireo_amd import synth
    from vireo_amd.counts import DeviceCounts
    N, M, K, dens = synth.CONFIGS["c3"]
    w = synth.donor_workload(N, M, K, dens, seed=0)
    counts = DeviceCounts.from_merged(w["shape"], w["colptr"], w["rowidx"], w["ad"], w["dp"])

    def fit():
        np.random.seed(1)
        m = va.Vireo(n_var=N, n_cell=M, n_donor=K)
        m.fit(counts, None, min_iter=5, max_iter=20, delay_fit_theta=3, verbose=False)
        return m

    m = fit()
    # (1) properties
    np.testing.assert_allclose(m.ID_prob.sum(1), 1.0, rtol=0, atol=1e-12)
    np.testing.assert_allclose(m.GT_prob.sum(2), 1.0, rtol=0, atol=1e-12)
    assert np.all(np.isfinite(m.ELBO_)) and len(m.ELBO_) >= 6
    gain = np.diff(m.ELBO_)
    assert np.all(gain[3:] > -1e-6 * np.abs(m.ELBO_[4:])), "ELBO decreased after theta kicked in"
    assert np.all((m.beta_mu > 0) & (m.beta_mu < 1)) and np.all(m.beta_sum > 0)
    # (2) bitwise determinism of the whole fit
    m2 = fit()
    assert np.array_equal(m.ELBO_, m2.ELBO_) and np.array_equal(m.ID_prob, m2.ID_prob)
    # (3) the planted donors are recovered: the generator's z (SURVEY.md 8d draw order)
    rng = np.random.default_rng(0)
    nnz_t = int(N * M * dens)
    rng.integers(0, N, nnz_t); rng.integers(0, M, nnz_t); rng.poisson(1.0, nnz_t)
    rng.integers(0, 3, (N, K))
    z = rng.integers(0, K, M)
    lab = m.ID_prob.argmax(1)
    conf = np.zeros((K, K), int)
    np.add.at(conf, (z, lab), 1)
    purity = conf.max(1).sum() / M
    print("c3 fit: %d ELBO entries, purity %.4f" % (len(m.ELBO_), purity))
    assert purity > 0.99
    # (4) one oracle iteration from the fitted state vs one GPU iteration from the same state
    AD, DP = synth.as_scipy(w)
    st = O.vireo_new(M, N, K, ID_prob_init=m.ID_prob, GT_prob_init=m.GT_prob,
                     beta_mu_init=m.beta_mu.copy(), beta_sum_init=m.beta_sum.copy())
    st.ID_prob, st.GT_prob = m.ID_prob.copy(), m.GT_prob.copy()
    O.vireo_theta_step(st, AD, DP)
    O.vireo_gt_step(st, AD, DP)
    L = O.vireo_id_step(st, AD, DP)
    elbo_ref = O.vireo_elbo(st, L)
    m.update_theta_size(counts, None)
    m.update_GT_prob(counts, None)
    Lg = m.update_ID_prob(counts, None)
    elbo_gpu = m.get_ELBO(Lg, counts, None)
    np.testing.assert_allclose(m.beta_mu, st.beta_mu, rtol=RTOL)
    np.testing.assert_allclose(m.beta_sum, st.beta_sum, rtol=RTOL)
    np.testing.assert_allclose(m.GT_prob, st.GT_prob, rtol=RTOL, atol=1e-290)
    np.testing.assert_allclose(m.ID_prob, st.ID_prob, rtol=RTOL, atol=1e-290)
    np.testing.assert_allclose(Lg, L, rtol=1e-9)
    np.testing.assert_allclose(elbo_gpu, elbo_ref, rtol=RTOL)
    assert np.array_equal(m.ID_prob.argmax(1), st.ID_prob.argmax(1))


# ---------------------------------------------------------------- the flag paths at headline size
_C3 = {}


def _c3_problem():
    """the c3 workload, its device problem and its SciPy form, built once for the flag tests"""
    if not _C3:
        from vireo_amd import synth
        from vireo_amd.counts import DeviceCounts
        N, M, K, dens = synth.CONFIGS["c3"]
        w = synth.donor_workload(N, M, K, dens, seed=0)
        _C3["counts"] = DeviceCounts.from_merged(w["shape"], w["colptr"], w["rowidx"], w["ad"], w["dp"])
        _C3["scipy"] = synth.as_scipy(w)
        _C3["GT"], _C3["z"] = w["GT"], w["z"]      # the planted genotypes and donors
    return _C3


@pytest.mark.parametrize("flag", ["ase", "fixedGT", "priorGT", "fixsum"])
def test_config3_flag_paths(va, flag):
    """the flag paths take other kernels than the headline measures (vrx_theta_ase and
    the per-variant theta tables; W from a fixed GT; the GT_prior table in vrx_gt_update and KL_GT;
    the fixed-sum theta update) and had no evidence above 2 600 x 2 200.  At c3 (N=100k x M=50k, K=16):
      ase      ASE_mode=True                                       (vireo_model.py:82, :177)
      fixedGT  learn_GT=False, GT_prob_init = GT_prior = the donors' genotypes: the command's mode 2,
               `vireo -d donors.vcf` (vireo.py:149-205, vireo_model.py:129-137)
      priorGT  learn_GT=True with a non-uniform GT_prior (mode 4, --forceLearnGT)
      fixsum   fix_beta_sum=True                                   (vireo_model.py:184)
    each: the whole timing protocol twice (bitwise repeatable), properties of the fitted state, and ONE
    oracle iteration from the fitted state against one GPU iteration from it (rtol 1e-5, identical
    assignments)."""
    from vireo_amd.synth import planted_gt_prior
    c3 = _c3_problem()
    counts, (AD, DP), GT_true, z = c3["counts"], c3["scipy"], c3["GT"], c3["z"]
    N, M = counts.shape
    K = 16
    kw = dict(ase=dict(ASE_mode=True), fixedGT=dict(learn_GT=False), priorGT=dict(), fixsum=dict(fix_beta_sum=True))[flag]
    prior = None
    if flag == "fixedGT":
        prior = planted_gt_prior(GT_true, 1.0)            # (set_prior clips to [1e-5, 1 - 1e-5])
    elif flag == "priorGT":
        # a donor VCF that is right for 90 % of the (variant, donor) calls
        rng = np.random.default_rng(5)
        noisy = np.where(rng.random(GT_true.shape) < 0.9, GT_true, rng.integers(0, 3, GT_true.shape))
        prior = planted_gt_prior(noisy, 0.8)

    def fit():
        np.random.seed(1)
        init = dict(GT_prob_init=prior.copy()) if prior is not None else {}
        m = va.Vireo(n_var=N, n_cell=M, n_donor=K, **kw, **init)
        if prior is not None:
            m.set_prior(GT_prior=prior.copy())
        m.fit(counts, None, min_iter=5, max_iter=20, delay_fit_theta=3, verbose=False)
        return m

    m = fit()
    m2 = fit()
    for name in ("ELBO_", "ID_prob", "GT_prob", "beta_mu", "beta_sum"):
        assert np.array_equal(getattr(m, name), getattr(m2, name)), name
    assert np.all(np.isfinite(m.ELBO_)) and len(m.ELBO_) >= 5
    np.testing.assert_allclose(m.ID_prob.sum(1), 1.0, rtol=0, atol=1e-12)
    np.testing.assert_allclose(m.GT_prob.sum(2), 1.0, rtol=0, atol=1e-12)
    assert m.beta_mu.shape == ((N, 3) if flag == "ase" else (1, 3))
    assert np.all((m.beta_mu > 0) & (m.beta_mu < 1)) and np.all(m.beta_sum > 0)
    if flag == "fixsum":
        assert np.all(m.beta_sum == 50.0)
    if flag == "fixedGT":       # the genotypes never move, and with them known the donors are the planted ones
        m0 = va.Vireo(n_var=N, n_cell=M, n_donor=K, learn_GT=False, GT_prob_init=prior.copy())
        assert np.array_equal(m.GT_prob, m0.GT_prob)
        assert np.mean(m.ID_prob.argmax(1) == z) > 0.999
    lab = m.ID_prob.argmax(1)
    conf = np.zeros((K, K), int)
    np.add.at(conf, (z, lab), 1)
    purity = conf.max(1).sum() / M
    print("c3 %s: %d ELBO entries, last %.6f, purity %.4f" % (flag, len(m.ELBO_), m.ELBO_[-1], purity))
    assert purity > (0.9 if flag == "ase" else 0.99)
    # ONE oracle iteration from the fitted state vs one GPU iteration from the same state
    st = O.vireo_new(M, N, K, ID_prob_init=m.ID_prob, GT_prob_init=m.GT_prob, beta_mu_init=m.beta_mu.copy(),
                     beta_sum_init=m.beta_sum.copy(), learn_GT=kw.get("learn_GT", True),
                     ASE_mode=kw.get("ASE_mode", False), fix_beta_sum=kw.get("fix_beta_sum", False))
    st.ID_prob, st.GT_prob = m.ID_prob.copy(), m.GT_prob.copy()
    if prior is not None:
        O.vireo_prior(st, GT_prior=prior.copy())
    O.vireo_theta_step(st, AD, DP)
    if st.learn_GT:
        O.vireo_gt_step(st, AD, DP)
    L = O.vireo_id_step(st, AD, DP)
    elbo_ref = O.vireo_elbo(st, L)
    m.update_theta_size(counts, None)
    if m.learn_GT:
        m.update_GT_prob(counts, None)
    Lg = m.update_ID_prob(counts, None)
    elbo_gpu = m.get_ELBO(Lg, counts, None)
    np.testing.assert_allclose(m.beta_mu, st.beta_mu, rtol=RTOL)
    np.testing.assert_allclose(m.beta_sum, st.beta_sum, rtol=RTOL)
    np.testing.assert_allclose(m.GT_prob, st.GT_prob, rtol=RTOL, atol=1e-290)
    np.testing.assert_allclose(m.ID_prob, st.ID_prob, rtol=RTOL, atol=1e-290)
    np.testing.assert_allclose(Lg, L, rtol=1e-9)
    np.testing.assert_allclose(elbo_gpu, elbo_ref, rtol=RTOL)
    assert np.array_equal(m.ID_prob.argmax(1), st.ID_prob.argmax(1))


def test_config3_balanced_slabs(va, monkeypatch):
    """The headline configuration on BALANCED SLABS (what bench.py's headline runs on): both streams
    balanced, a quarter fewer stream slots, the whole protocol bitwise repeatable, within summation-order
    distance of the default build's (same iteration count, same assignments), planted donors recovered, and
    ONE iteration from the fitted state against the oracle (theta on the whole matrix, GT / ID on 2 000
    rows / columns: tests/subset_parity.py)."""
    from vireo_amd import _lib, synth
    from vireo_amd.counts import DeviceCounts
    from vireo_amd.engine import DeviceModel
    from tests.subset_parity import one_iteration_subset_check
    N, M, K, dens = synth.CONFIGS["c3"]
    w = synth.donor_workload(N, M, K, dens, seed=0)
    # (the first build also runs the host's greedy -- the specification -- and fails unless the device's agrees
    #  with it bit for bit: 3.3 M (tile, contracted row) assignments per orientation)
    monkeypatch.setenv("VIREO_BALANCE_CHECK", "1")
    cb = DeviceCounts.from_merged(w["shape"], w["colptr"], w["rowidx"], w["ad"], w["dp"], balance=True)
    monkeypatch.delenv("VIREO_BALANCE_CHECK")
    assert DeviceCounts.from_merged(w["shape"], w["colptr"], w["rowidx"], w["ad"], w["dp"], balance=True).digest() \
        == cb.digest()
    cd = DeviceCounts.from_merged(w["shape"], w["colptr"], w["rowidx"], w["ad"], w["dp"], balance=False)
    ib = cb.build_info()
    assert ib["balanced_variant"] and ib["balanced_cell"] and ib["device_built"]
    kb, kd = DeviceModel(cb, _lib.KIND_VIREO, K).info(), DeviceModel(cd, _lib.KIND_VIREO, K).info()
    print("c3 balanced slabs: +%.2f s at build; stream slots per non-zero %.3f / %.3f against %.3f / %.3f"
          % (ib["balance_seconds"], kb["pad_variant"], kb["pad_cell"], kd["pad_variant"], kd["pad_cell"]))
    assert kb["pad_cell"] < 0.85 * kd["pad_cell"] and kb["pad_variant"] < 0.85 * kd["pad_variant"]

    def fit(counts):
        np.random.seed(1)
        m = va.Vireo(n_var=N, n_cell=M, n_donor=K)
        m.fit(counts, None, min_iter=5, max_iter=20, delay_fit_theta=3, verbose=False)
        return m

    m, m2, d = fit(cb), fit(cb), fit(cd)
    for name in ("ELBO_", "ID_prob", "GT_prob", "beta_mu", "beta_sum"):
        assert np.array_equal(getattr(m, name), getattr(m2, name)), name
    assert len(m.ELBO_) == len(d.ELBO_) and np.array_equal(m.ID_prob.argmax(1), d.ID_prob.argmax(1))
    # (two summation orders of one unstable trajectory: like GPU against oracle, they part by ~1e-5 in the
    #  middle of the trace and meet again at its end)
    rel = np.abs(m.ELBO_ - d.ELBO_) / np.abs(d.ELBO_)
    assert rel[:6].max() < 1e-9 and rel[-1] < 1e-7 and rel.max() < 5e-5, rel
    lab = m.ID_prob.argmax(1)
    conf = np.zeros((K, K), int)
    np.add.at(conf, (w["z"], lab), 1)
    assert conf.max(1).sum() / M > 0.99
    par = one_iteration_subset_check(m, cb, w, n_sub=2000)
    print("c3 balanced slabs, one iteration vs the oracle:", par)
    cb.close()
    cd.close()


def test_four_times_config3_subset_parity(va):
    """parity at a size the oracle cannot iterate -- N = 200k x M = 100k, K = 16, 4e8
    entries (4x c3; tests/perf/big_probe.py runs the same check at 1.6e9 and 2.2e9 entries,
    profiles/r06_big_probe_*.txt).  The whole protocol on the GPU (bitwise repeatable, planted donors
    recovered), then ONE iteration from the fitted state against the oracle by the column-subset trick
    (tests/subset_parity.py): theta against the oracle's whole-matrix sums, ``GT_prob`` on 2 000 variants
    and ``logLik_ID`` / ``ID_prob`` on 2 000 cells, where the oracle on the sub-matrix is exact."""
    from vireo_amd import synth
    from vireo_amd.counts import DeviceCounts
    from tests.subset_parity import one_iteration_subset_check
    N, M, K = 200000, 100000, 16
    w = synth.big_workload(N, M, K, 0.02, seed=0)
    counts = DeviceCounts.from_merged(w["shape"], w["colptr"], w["rowidx"], w["ad"], w["dp"], balance=True)
    assert counts.build_info()["balanced_cell"] and counts.build_info()["balanced_variant"]

    def fit():
        np.random.seed(1)
        m = va.Vireo(n_var=N, n_cell=M, n_donor=K)
        m.fit(counts, None, min_iter=5, max_iter=20, delay_fit_theta=3, verbose=False)
        return m

    m, m2 = fit(), fit()
    assert np.array_equal(m.ELBO_, m2.ELBO_) and np.array_equal(m.ID_prob, m2.ID_prob)
    del m2
    assert np.all(np.isfinite(m.ELBO_)) and len(m.ELBO_) >= 6
    lab = m.ID_prob.argmax(1)
    conf = np.zeros((K, K), int)
    np.add.at(conf, (w["z"], lab), 1)
    assert conf.max(1).sum() / M > 0.99
    par = one_iteration_subset_check(m, counts, w, n_sub=2000)
    print("4x c3 (nnz %d): %s" % (w["rowidx"].size, par))
    counts.close()


def test_config3_heavy_tailed_data(va):
    """c3's shape with log-normal coverage / depth (synth.C3_SKEW: what real cellSNP matrices look
    like, io_utils.py:42-59; the uniform SURVEY.md 8(d) generator is the kernels' best case): the
    LDS-resident passes are taken (long rows cut into pieces), a whole fit is bitwise repeatable,
    and ONE oracle iteration from the fitted state equals one GPU iteration from it."""
    from vireo_amd import _lib, synth
    from vireo_amd.counts import DeviceCounts
    from vireo_amd.engine import DeviceModel
    N, M, K, dens = synth.CONFIGS["c3"]
    w = synth.donor_workload(N, M, K, dens, seed=0, skew=synth.C3_SKEW)
    rows = np.bincount(w["rowidx"], minlength=N)
    cols = np.diff(w["colptr"])
    assert rows.max() > 20 * np.median(rows) and cols.max() > 5 * np.median(cols)    # heavy tails
    counts = DeviceCounts.from_merged(w["shape"], w["colptr"], w["rowidx"], w["ad"], w["dp"])
    info = DeviceModel(counts, _lib.KIND_VIREO, K).info()
    print("c3 skew: nnz %d, rows %d..%d (median %d), cells %d..%d (median %d), max count %d; %s"
          % (w["rowidx"].size, rows.min(), rows.max(), np.median(rows), cols.min(), cols.max(),
             np.median(cols), w["dp"].max(), info))
    assert info["lds_variant"] and info["lds_cell"]
    assert (info["cell_form"], info["var_form"]) == (1, 3)
    assert info["extra_pieces_variant"] > 0 and info["extra_pieces_cell"] > 0

    def fit():
        np.random.seed(1)
        m = va.Vireo(n_var=N, n_cell=M, n_donor=K)
        m.fit(counts, None, min_iter=5, max_iter=20, delay_fit_theta=3, verbose=False)
        return m

    m = fit()
    m2 = fit()
    assert np.array_equal(m.ELBO_, m2.ELBO_) and np.array_equal(m.ID_prob, m2.ID_prob)
    assert np.array_equal(m.GT_prob, m2.GT_prob)
    assert np.all(np.isfinite(m.ELBO_)) and len(m.ELBO_) >= 6
    np.testing.assert_allclose(m.ID_prob.sum(1), 1.0, rtol=0, atol=1e-12)
    AD, DP = synth.as_scipy(w)
    st = O.vireo_new(M, N, K, ID_prob_init=m.ID_prob, GT_prob_init=m.GT_prob,
                     beta_mu_init=m.beta_mu.copy(), beta_sum_init=m.beta_sum.copy())
    st.ID_prob, st.GT_prob = m.ID_prob.copy(), m.GT_prob.copy()
    O.vireo_theta_step(st, AD, DP)
    O.vireo_gt_step(st, AD, DP)
    L = O.vireo_id_step(st, AD, DP)
    elbo_ref = O.vireo_elbo(st, L)
    m.update_theta_size(counts, None)
    m.update_GT_prob(counts, None)
    Lg = m.update_ID_prob(counts, None)
    elbo_gpu = m.get_ELBO(Lg, counts, None)
    np.testing.assert_allclose(m.beta_mu, st.beta_mu, rtol=RTOL)
    np.testing.assert_allclose(m.beta_sum, st.beta_sum, rtol=RTOL)
    np.testing.assert_allclose(m.GT_prob, st.GT_prob, rtol=RTOL, atol=1e-290)
    np.testing.assert_allclose(m.ID_prob, st.ID_prob, rtol=RTOL, atol=1e-290)
    np.testing.assert_allclose(Lg, L, rtol=1e-9)
    np.testing.assert_allclose(elbo_gpu, elbo_ref, rtol=RTOL)
    assert np.array_equal(m.ID_prob.argmax(1), st.ID_prob.argmax(1))


def test_config5_clone_mode_vs_oracle(va):
    """BinomMixtureVB at N=200 x M=200k, K=8 (BASELINE.json configs[4]): three iterations
    from the same seeded start against the oracle."""
    AD, DP = O.synth_clone(200, 200000, 8, seed=0)
    np.random.seed(1)
    ref = O.bmm_new(200000, 200, 8)
    ID0 = ref.ID_prob.copy()
    O.bmm_fit_vb(ref, AD, DP, max_iter=3, min_iter=1)
    dev = va.BinomMixtureVB(n_var=200, n_cell=200000, n_donor=8, ID_prob_init=ID0)
    dev._fit_BV(AD, DP, max_iter=3, min_iter=1, verbose=False)
    assert len(dev.ELBO_iters) == len(ref.ELBO_iters) == 2
    np.testing.assert_allclose(dev.ELBO_iters, ref.ELBO_iters, rtol=RTOL)
    np.testing.assert_allclose(dev.beta_mu, ref.beta_mu, rtol=RTOL)
    np.testing.assert_allclose(dev.beta_sum, ref.beta_sum, rtol=RTOL)
    np.testing.assert_allclose(dev.ID_prob, ref.ID_prob, rtol=RTOL, atol=1e-290)
    assert np.array_equal(dev.ID_prob.argmax(1), ref.ID_prob.argmax(1))
    np.testing.assert_allclose(va.device_counts(AD, DP).binom_const(), O.binom_const(AD, DP),
                               rtol=1e-6)


def test_rccl_communicator_world1(va):
    """the RCCL path of the restart shard (libvireo_hip's vrx_comm_*) on the one GPU of this
    box: unique id, ncclCommInitRank, all-gather, broadcast, barrier."""
    from vireo_amd.dist import RcclComm, gather_restart_elbos
    comm = RcclComm(0, 1, 0, lambda raw: raw)
    out = comm.allgather(np.array([1.5, -2.0, 3.25]))
    assert np.array_equal(out, [1.5, -2.0, 3.25])
    x = np.arange(12.0).reshape(3, 4)
    assert np.array_equal(comm.bcast(x, 0), x)
    comm.barrier()
    assert np.array_equal(gather_restart_elbos(comm, 3, {0: 1.0, 1: 7.0, 2: 7.0}), [1, 7, 7])
    comm.close()


def test_winner_broadcast_device_to_device_equals_host_route(va):
    """``vrx_comm_bcast_model`` (ncclBroadcast straight from / into the model's HBM
    buffers) against the host-staged route of ``_bcast_model`` -- over RCCL at world 1, the most a
    one-GPU box can run: both leave every rank's model with the root's state, bit for bit, and the
    device model itself is untouched by the in-place broadcast.  (At world 2 the sharded vireo_wrap of
    test_restart_shard_over_rccl_world2 takes the device route.)"""
    import sys
    import vireo_amd.vireo_wrap            # noqa: F401   (the package exports the function under this name)
    from vireo_amd.dist import RcclComm, comm_record
    W = sys.modules["vireo_amd.vireo_wrap"]
    AD, DP = gold.c1()
    comm = RcclComm(0, 1, 0, lambda raw: raw)
    try:
        rec = comm_record(comm, 0)
        assert rec["backend"] == "rccl" and rec["world"] == 1 and rec["distinct_gpus"] == 1
        assert rec["ranks"][0]["pci_bus_id"] and rec["rccl_version_code"] >= 20000
        for flags in (dict(), dict(ASE_mode=True)):
            np.random.seed(4)
            a = va.Vireo(n_var=AD.shape[0], n_cell=AD.shape[1], n_donor=4, **flags)
            a.fit(AD, DP, max_iter=6, verbose=False)
            dm, _ = a._device_model(AD, DP)
            before = [x.copy() for x in dm.get_state()]
            via_dev = va.Vireo(n_var=AD.shape[0], n_cell=AD.shape[1], n_donor=4, **flags)
            via_host = va.Vireo(n_var=AD.shape[0], n_cell=AD.shape[1], n_donor=4, **flags)
            for m in (via_dev, via_host):
                m.ID_prob, m.GT_prob, m.beta_mu, m.beta_sum = a.ID_prob, a.GT_prob, a.beta_mu, a.beta_sum
                m.ELBO_ = a.ELBO_
            W._bcast_model(comm, via_dev, 0, dm=dm, force=True)
            W._bcast_model(comm, via_host, 0, dm=None, force=True)
            after = dm.get_state()
            for x, y in zip(before, after):
                assert np.array_equal(x, y)
            for name in ("ID_prob", "GT_prob", "beta_mu", "beta_sum", "ELBO_"):
                assert np.array_equal(getattr(via_dev, name), getattr(via_host, name)), name
                assert np.array_equal(getattr(via_dev, name), getattr(a, name)), name
            dm.close()
        # clone mode: no genotype layer (three arrays)
        from vireo_amd import _lib
        from vireo_amd.engine import DeviceModel
        mAD, mDP = gold.mito()
        np.random.seed(2)
        b = va.BinomMixtureVB(n_var=mAD.shape[0], n_cell=mAD.shape[1], n_donor=3)
        bdm = b._device_model(mAD, mDP)
        st = [None if x is None else x.copy() for x in bdm.get_state()]
        comm.bcast_model(bdm, 0)
        for x, y in zip(st, bdm.get_state()):
            assert (x is None and y is None) or np.array_equal(x, y)
        bdm.close()
        with pytest.raises(_lib.VrxError):
            comm.bcast_model(DeviceModel(va.device_counts(mAD, mDP), _lib.KIND_BMM, 3), 1)   # root outside the world
    finally:
        comm.close()


def test_config3_whole_protocol_trace_vs_oracle(va):
    """The metric is "EM iterations/sec + ELBO-match": the WHOLE timing protocol
    _fit_VB(min_iter=5, max_iter=20, delay_fit_theta=3) (vireo_model.py:251-276) at
    BASELINE.json configs[2], on the GPU and on the oracle from the same seeded start:
    identical iteration count, every ELBO of the trace within 1e-5 relative, identical
    assignments at the end.  (~2 min of single-core oracle time.)"""
    from vireo_amd import synth
    from vireo_amd.counts import DeviceCounts
    N, M, K, dens = synth.CONFIGS["c3"]
    w = synth.donor_workload(N, M, K, dens, seed=0)
    counts = DeviceCounts.from_merged(w["shape"], w["colptr"], w["rowidx"], w["ad"], w["dp"])
    np.random.seed(1)
    dev = va.Vireo(n_var=N, n_cell=M, n_donor=K)
    gtrace = dev._fit_VB(counts, None, min_iter=5, max_iter=20, delay_fit_theta=3, verbose=False)
    AD, DP = synth.as_scipy(w)
    np.random.seed(1)
    st = O.vireo_new(M, N, K)
    ctrace, it = O.vireo_fit_vb(st, AD, DP, min_iter=5, max_iter=20, delay_fit_theta=3)
    print("c3 protocol: %d kept iterations; trace rel err %.2e; ID_prob abs err %.2e"
          % (len(ctrace), np.max(np.abs(gtrace - ctrace) / np.abs(ctrace)),
             np.max(np.abs(dev.ID_prob - st.ID_prob))))
    assert len(gtrace) == len(ctrace)
    # The first iterations leave a symmetric, unstable state (posteriors uniform to ~3e-7).
    # While the donor clusters form, rounding-order differences between ANY two implementations
    # grow ~1000x per iteration (observed: 2e-13 until iteration 5, 3e-8 at 7, 1.2e-5 at 8),
    # then the iteration contracts again (4.5e-9 at the end).  The same fit on the GPU from an
    # initial state perturbed by 1e-12 shows that this is the trajectory's own sensitivity:
    np.random.seed(1)
    pert = va.Vireo(n_var=N, n_cell=M, n_donor=K)
    pert.ID_prob = pert.ID_prob * (1.0 + 1e-12 * np.random.default_rng(7).standard_normal(pert.ID_prob.shape))
    ptrace = pert._fit_VB(counts, None, min_iter=5, max_iter=20, delay_fit_theta=3, verbose=False)
    assert len(ptrace) == len(gtrace)
    own = np.abs(ptrace - gtrace) / np.abs(gtrace)
    err = np.abs(gtrace - ctrace) / np.abs(ctrace)
    print("per-iteration |gpu - cpu| / |cpu|:", " ".join("%.1e" % x for x in err))
    print("per-iteration GPU self-sensitivity (1e-12 perturbation):", " ".join("%.1e" % x for x in own))
    assert np.all(err[:6] <= 1e-9)                       # before the unstable phase: rounding only
    assert err[-1] <= 1e-7                               # after it: the same fixed point
    # In between, the arbiter decides: the same protocol in 80-bit extended precision
    # (tests/golden/make_c3_arbiter.py -> c3_protocol_longdouble.npz; its own rounding noise is
    # ~1e-3 of float64's).  The reference's float64 arithmetic (the oracle) is itself only
    # e_cpu-close to the mathematics there; the GPU has to be as close -- up to a factor for the
    # luck of the rounding draw: both errors are the SAME amplified noise process, sampled twice.
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "c3_protocol_longdouble.npz"))
    assert int(g["n_iter"]) == len(gtrace) and int(g["nnz"]) == int(w["rowidx"].size)
    exact = g["elbo_hi"].astype(np.longdouble) + g["elbo_lo"].astype(np.longdouble)
    e_gpu = np.abs((gtrace.astype(np.longdouble) - exact) / exact).astype(float)
    e_cpu = np.abs((ctrace.astype(np.longdouble) - exact) / exact).astype(float)
    print("per-iteration |gpu - exact| / |exact|:   ", " ".join("%.1e" % x for x in e_gpu))
    print("per-iteration |oracle - exact| / |exact|:", " ".join("%.1e" % x for x in e_cpu))
    assert np.all(e_gpu[:6] <= 1e-9) and e_gpu[-1] <= 1e-7
    assert np.all(e_gpu <= np.maximum(RTOL, 2.0 * e_cpu.max()))    # (observed: 6.0e-6 against the oracle's 1.3e-5)
    assert np.array_equal(dev.ID_prob.argmax(1), g["assign"])
    np.testing.assert_allclose(dev.beta_mu, st.beta_mu, rtol=RTOL)
    np.testing.assert_allclose(dev.beta_sum, st.beta_sum, rtol=RTOL)
    # END-STATE POSTERIORS (vireo_model.py:198-199, :218-219), pinned by the arbiter as well (round
    # 4): the protocol stops after 20 iterations, before the remnant of the amplified difference has
    # died out, so the GPU's and the oracle's final states each sit a little off the exact one.  An
    # element of the GPU's posteriors must be within 1e-5 relative of the exact value OR within
    # twice the oracle's own largest absolute deviation from it -- the HIP path may not be further
    # from the mathematics than the reference's float64 arithmetic is.  No absolute allowance
    # beyond that, none on the genotype calls.
    sl = slice(None, None, int(g["GT_stride"]))
    for name, gpu, cpu, exact_a in (("ID_prob", dev.ID_prob, st.ID_prob, g["ID_prob"]),
                                    ("GT_prob[::%d]" % int(g["GT_stride"]), dev.GT_prob[sl], st.GT_prob[sl],
                                     g["GT_prob_sample"])):
        assert gpu.shape == exact_a.shape
        d_gpu, d_cpu = np.abs(gpu - exact_a), np.abs(cpu - exact_a)
        tol = np.maximum(RTOL * np.abs(exact_a), 2.0 * d_cpu.max())
        print("%s: max |gpu - exact| %.2e, max |oracle - exact| %.2e; worst element at %.2f of its tolerance"
              % (name, d_gpu.max(), d_cpu.max(), np.max(d_gpu / np.maximum(tol, 1e-300))))
        assert np.all(d_gpu <= tol), name
    assert np.array_equal(dev.ID_prob.argmax(1), st.ID_prob.argmax(1))
    # genotype calls: identical to the exact ones wherever the exact posterior is not a tie
    # (variants without reads under a donor keep the uniform prior: argmax of three equal numbers)
    xg = g["GT_prob_sample"]
    top2 = np.sort(xg, axis=2)
    decided = top2[:, :, 2] - top2[:, :, 1] > 4.0 * np.abs(st.GT_prob[sl] - xg).max()
    # (a donor column without cells keeps the uniform genotype prior -- an exact tie; when the
    #  protocol stops after 20 iterations, 6 of the 16 columns are still empty)
    used = np.bincount(g["assign"], minlength=K) > 0
    assert used.sum() >= 2 and decided[:, used].mean() > 0.9 and not decided[:, ~used].any()
    assert np.array_equal(dev.GT_prob[sl].argmax(2)[decided], xg.argmax(2)[decided])
    # ... and over ALL variants the GPU's calls equal the oracle's wherever the oracle's own
    # posterior is decided by more than that margin
    t2 = np.sort(st.GT_prob, axis=2)
    dec_all = t2[:, :, 2] - t2[:, :, 1] > 4.0 * np.abs(st.GT_prob[sl] - xg).max()
    assert np.array_equal(dev.GT_prob.argmax(2)[dec_all], st.GT_prob.argmax(2)[dec_all])


def test_config4_restart_search_n_init32(va):
    """BASELINE.json configs[3] on one GPU: vireo_wrap(n_init=32, max_iter_init=20, random_seed=1,
    check_doublet=False) on the c3 data (vireo_wrap.py:64-94).  Every restart must be the fit the
    reference would run from the i-th constructor's draws: checked against independent
    ``Vireo.fit`` calls from those draws (bitwise: same kernels, same order), for restart 0 (the
    single-restart timing protocol), three others and the winner -- and those three others meet
    the oracle for one iteration from their fitted state; the winner is the FIRST
    maximum of LB_list, and the returned state is that restart refined by
    ``fit(min_iter=5)`` (vireo_wrap.py:93)."""
    import contextlib
    import io
    from vireo_amd import synth
    from vireo_amd.counts import DeviceCounts
    N, M, K, dens = synth.CONFIGS["c3"]
    w = synth.donor_workload(N, M, K, dens, seed=0)
    counts = DeviceCounts.from_merged(w["shape"], w["colptr"], w["rowidx"], w["ad"], w["dp"])
    n_init = 32
    with contextlib.redirect_stdout(io.StringIO()):
        rv = va.vireo_wrap(counts, None, n_donor=K, n_init=n_init, max_iter_init=20, random_seed=1,
                           check_doublet=False)
    LB = np.asarray(rv["LB_list"])
    assert LB.shape == (n_init,) and np.all(np.isfinite(LB))
    best = int(np.argmax(LB))                               # np.argmax: the first maximum
    check = sorted({0, 7, 19, 31, best})
    np.random.seed(1)                                       # the one seeding, vireo_wrap.py:53-54
    fitted = {}
    for i in range(n_init):                                 # sequential constructors, :66-71
        m = va.Vireo(n_var=N, n_cell=M, n_donor=K)
        if i in check:
            m.fit(counts, None, min_iter=5, max_iter=20, delay_fit_theta=3, verbose=False)
            fitted[i] = m
        else:
            del m
    for i in check:
        assert fitted[i].ELBO_[-1] == LB[i], (i, fitted[i].ELBO_[-1], LB[i])
    # restarts 7 / 19 / 31 meet the ORACLE too: one oracle iteration from the state the
    # restart's 20-iteration fit left against one GPU iteration from the same state
    AD, DP = synth.as_scipy(w)
    for i in (7, 19, 31):
        m = fitted[i]
        st = O.vireo_new(M, N, K, ID_prob_init=m.ID_prob, GT_prob_init=m.GT_prob,
                         beta_mu_init=m.beta_mu.copy(), beta_sum_init=m.beta_sum.copy())
        st.ID_prob, st.GT_prob = m.ID_prob.copy(), m.GT_prob.copy()
        O.vireo_theta_step(st, AD, DP)
        O.vireo_gt_step(st, AD, DP)
        L = O.vireo_id_step(st, AD, DP)
        elbo_ref = O.vireo_elbo(st, L)
        g = va.Vireo(n_var=N, n_cell=M, n_donor=K, ID_prob_init=m.ID_prob.copy(), GT_prob_init=m.GT_prob.copy(),
                     beta_mu_init=m.beta_mu.copy(), beta_sum_init=m.beta_sum.copy())
        g.ID_prob, g.GT_prob = m.ID_prob.copy(), m.GT_prob.copy()      # (exactly the fitted state)
        g.update_theta_size(counts, None)
        g.update_GT_prob(counts, None)
        Lg = g.update_ID_prob(counts, None)
        elbo_gpu = g.get_ELBO(Lg, counts, None)
        np.testing.assert_allclose(g.beta_mu, st.beta_mu, rtol=RTOL)
        np.testing.assert_allclose(g.beta_sum, st.beta_sum, rtol=RTOL)
        np.testing.assert_allclose(g.GT_prob, st.GT_prob, rtol=RTOL, atol=1e-290)
        np.testing.assert_allclose(g.ID_prob, st.ID_prob, rtol=RTOL, atol=1e-290)
        np.testing.assert_allclose(Lg, L, rtol=1e-9)
        np.testing.assert_allclose(elbo_gpu, elbo_ref, rtol=RTOL)
        # ALL cells are compared: a restart stopped after 20 iterations is not
        # converged -- where two donor columns still describe the same donor, its cells sit between
        # them and the two largest posteriors are equal to rounding (observed: 8 % of the cells of
        # restart 31) -- so the rule is tie-aware: the GPU's call is the oracle's, or the oracle
        # itself holds the GPU's donor within 10 x rtol of its own largest posterior
        ga, oa = g.ID_prob.argmax(1), st.ID_prob.argmax(1)
        top = st.ID_prob.max(1)
        at_gpu_call = st.ID_prob[np.arange(M), ga]
        tie_ok = top - at_gpu_call <= 10 * RTOL * top
        print("restart %d: %.2f %% of the cells called differently, all of them ties: %s"
              % (i, 100 * np.mean(ga != oa), bool(np.all(tie_ok[ga != oa]))))
        assert np.all((ga == oa) | tie_ok)
        assert np.mean(ga == oa) > 0.5
        del st, g
    # restart 0 is the timing protocol of bench.py (the same seed, the first constructor)
    np.random.seed(1)
    m0 = va.Vireo(n_var=N, n_cell=M, n_donor=K)
    tr = m0._fit_VB(counts, None, min_iter=5, max_iter=20, delay_fit_theta=3, verbose=False)
    assert tr[-1] + counts.binom_const() == LB[0]
    # the refinement of the winner (vireo_wrap.py:93: fit(AD, DP, min_iter=5), all defaults else)
    win = fitted[best]
    win.fit(counts, None, min_iter=5, verbose=False)
    assert np.array_equal(rv["ID_prob"], win.ID_prob)
    assert np.array_equal(rv["GT_prob"], win.GT_prob)
    assert np.array_equal(rv["theta_mean"], win.beta_mu) and np.array_equal(rv["theta_sum"], win.beta_sum)
    assert rv["LB_doublet"] == win.ELBO_[-1]
    assert rv["doublet_prob"].shape == (M, K * (K - 1) // 2) and not rv["doublet_prob"].any()
    # The DEFAULT post-step of vireo_wrap (check_doublet=True, vireo_wrap.py:151-156 ->
    # vireo_doublet.py:11-82) at K = 16: K + K(K-1)/2 = 136 columns (9 sweeps of the cell pass),
    # 6 genotype classes per donor pair.  Given GT_prob and theta the step is independent per
    # cell, so the oracle on a 2 000-cell column subset of AD / DP is exact for those cells
    # (the whole matrix costs the reference > 5 minutes and a 653-MB table).
    sub = np.sort(np.random.default_rng(3).choice(M, 2000, replace=False))
    st = O.vireo_new(sub.size, N, K, ID_prob_init=win.ID_prob[sub], GT_prob_init=win.GT_prob,
                     beta_mu_init=win.beta_mu.copy(), beta_sum_init=win.beta_sum.copy())
    st.ID_prob, st.GT_prob = win.ID_prob[sub].copy(), win.GT_prob.copy()
    dp_o, ip_o, llr_o = O.vireo_doublet(st, AD[:, sub], DP[:, sub], doublet_rate_prior=min(0.5, M / 100000))
    import time
    t0 = time.perf_counter()
    dp_g, ip_g, llr_g = va.predict_doublet(win, counts, None)
    print("predict_doublet at c3 / K = 16: %.3f s" % (time.perf_counter() - t0))
    assert dp_g.shape == (M, K * (K - 1) // 2) and ip_g.shape == (M, K) and llr_g.shape == (M,)
    np.testing.assert_allclose(dp_g[sub], dp_o, rtol=RTOL, atol=1e-290)
    np.testing.assert_allclose(ip_g[sub], ip_o, rtol=RTOL, atol=1e-290)
    np.testing.assert_allclose(llr_g[sub], llr_o, rtol=RTOL, atol=1e-8)
    both_g, both_o = np.append(ip_g[sub], dp_g[sub], axis=1), np.append(ip_o, dp_o, axis=1)
    assert np.array_equal(both_g.argmax(1), both_o.argmax(1))
    assert np.array_equal(dp_g[sub].sum(1) > 0.9, dp_o.sum(1) > 0.9)        # io_utils.py:104-106's doublet call
    assert np.allclose(both_g.sum(1), 1.0, rtol=0, atol=1e-12)
    assert win.ID_prob is ip_g or np.array_equal(win.ID_prob, ip_g)          # the side effect, vireo_doublet.py:71


def test_restart_shard_over_rccl_world2(va, tmp_path):
    """vireo_wrap(n_init=4) on the demo data with the restarts sharded over TWO GPUs (one
    process each, RCCL all-gather of the ELBOs, winner broadcast) against the reference's
    golden output.  Needs two visible devices; the driver's multi-GPU tier has them."""
    import os
    import pickle
    import socket
    import subprocess
    import sys
    from vireo_amd import _lib
    from tests import gold
    if _lib.device_count() < 2:
        pytest.skip("needs >= 2 GPUs (this box shows %d)" % _lib.device_count())
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    outs = [str(tmp_path / ("rank%d.pkl" % r)) for r in range(2)]
    procs = []
    for r in range(2):
        env = dict(os.environ, PYTHONPATH=root, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(root, "tests", "_rccl_worker.py"),
                                       outs[r]], env=env))
    for p in procs:
        assert p.wait(timeout=600) == 0
    g = gold.load("c1_wrap_seed2_init4")
    for path in outs:
        rv = pickle.load(open(path, "rb"))
        np.testing.assert_allclose(rv["LB_list"], g["LB_list"], rtol=RTOL)
        np.testing.assert_allclose(rv["ID_prob"], g["ID_prob"], rtol=RTOL, atol=1e-290)
        np.testing.assert_allclose(rv["GT_prob"], g["GT_prob"], rtol=RTOL, atol=1e-290)
        assert rv["LB_doublet"] == pytest.approx(g["LB_doublet"], rel=RTOL)
        assert np.array_equal(rv["ID_prob"].argmax(1), g["ID_prob"].argmax(1))
