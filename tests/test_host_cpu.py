"""CPU-only tests: the C ABI exports what include/vireo_hip.h declares, the host logic
(count canonicalisation, restart sharding with gloo at world size 2, K x K helpers), and
the loud failure of the product path when no GPU is present."""
import os
import pickle
import re
import socket
import subprocess
import sys
import time
import ctypes

import numpy as np
import pytest
from scipy.sparse import csc_matrix, csr_matrix, coo_matrix

import __graft_entry__ as entry
from tests import gold

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="session", autouse=True)
def built():
    entry.build()


def test_abi_exports_every_declared_symbol():
    from vireo_amd import _lib
    text = open(os.path.join(ROOT, "include", "vireo_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    declared = set(re.findall(r"\b(vrx_[a-z0-9_]+)\s*\(", text))
    assert len(declared) >= 25
    h = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(h, name), "library does not export " + name
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)


def test_error_reporting_without_compute():
    from vireo_amd import _lib
    n = ctypes.c_int(-1)
    assert _lib.lib().vrx_device_count(ctypes.byref(n)) == 0 and n.value >= 0
    rc = _lib.lib().vrx_problem_binom_const(None, None)
    assert rc < 0 and b"null" in _lib.lib().vrx_last_error()


def test_product_path_fails_loudly_without_gpu():
    import vireo_amd
    from vireo_amd import _lib
    if _lib.device_count() > 0:
        pytest.skip("a GPU is visible here")
    AD, DP = gold.c1()
    m = vireo_amd.Vireo(n_var=AD.shape[0], n_cell=AD.shape[1], n_donor=3)
    with pytest.raises(_lib.VrxError):
        m.fit(AD, DP, max_iter=2, verbose=False)
    with pytest.raises(_lib.VrxError):
        vireo_amd.BinomMixtureVB(n_var=AD.shape[0], n_cell=AD.shape[1], n_donor=3).fit(AD, DP)
    with pytest.raises(_lib.VrxError):
        vireo_amd.vireo_wrap(AD, DP, n_donor=3, n_init=1)


def test_merge_counts_union_pattern_and_formats():
    from vireo_amd.counts import merge_counts
    rng = np.random.default_rng(0)
    dp = (rng.random((40, 30)) < 0.2) * rng.integers(1, 9, (40, 30))
    ad = rng.binomial(dp, 0.5)
    ad[3, 4], dp[3, 4] = 5, 0               # AD outside DP's pattern
    base = merge_counts(csc_matrix(ad), csc_matrix(dp))
    shape, ptr, idx, a, d = base
    assert shape == (40, 30) and ptr[-1] == idx.size == a.size == d.size
    dense_a = csc_matrix((a, idx, ptr), shape=shape).toarray()
    dense_d = csc_matrix((d, idx, ptr), shape=shape).toarray()
    assert np.array_equal(dense_a, ad) and np.array_equal(dense_d, dp)
    assert idx.size == np.count_nonzero((ad != 0) | (dp != 0))
    for conv in (csr_matrix, lambda X: csc_matrix(X).astype(np.float64), np.asarray,
                 lambda X: coo_matrix(X)):
        other = merge_counts(conv(ad), conv(dp))
        for x, y in zip(base[1:], other[1:]):
            assert np.array_equal(x, y)
    # duplicates are summed, like scipy does implicitly
    r = np.array([0, 0, 2]); c = np.array([1, 1, 0])
    dup = coo_matrix((np.array([1, 2, 4]), (r, c)), shape=(3, 2))
    _, ptr, idx, a, d = merge_counts(dup, dup)
    assert list(d) == [4, 3] and list(idx) == [2, 0]
    with pytest.raises(ValueError):
        merge_counts(csc_matrix(ad * 0.5), csc_matrix(dp))
    with pytest.raises(ValueError):
        merge_counts(csc_matrix(-ad), csc_matrix(dp))
    with pytest.raises(ValueError):
        merge_counts(csc_matrix(ad[:5]), csc_matrix(dp))


def test_restart_sharding_arithmetic():
    from vireo_amd.dist import my_restarts, gather_restart_elbos, LocalComm

    class FakeComm:
        def __init__(self, rank, world, board):
            self.rank, self.world, self.board = rank, world, board

        def allgather(self, local):
            self.board[self.rank] = np.asarray(local).copy()
            return np.concatenate([self.board[r] for r in range(self.world)])

    for n_init, world in [(1, 1), (5, 2), (32, 8), (3, 8), (50, 4)]:
        elbo = np.random.default_rng(n_init).normal(size=n_init)
        owned = [my_restarts(n_init, r, world) for r in range(world)]
        assert sorted(sum(owned, [])) == list(range(n_init))
        board = {}
        for r in range(world):     # what every other rank contributes to the all-gather
            board[r] = np.pad(elbo[owned[r]], (0, -(-n_init // world) - len(owned[r])),
                              constant_values=-np.inf)
        for r in range(world):
            got = gather_restart_elbos(FakeComm(r, world, board), n_init,
                                       {i: elbo[i] for i in owned[r]})
            assert np.array_equal(got, elbo)
    assert np.array_equal(gather_restart_elbos(LocalComm(), 3, {0: 1., 1: 5., 2: 5.}), [1, 5, 5])


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_sharded_wrap_gloo_world2(tmp_path):
    """vireo_wrap's restart shard with 2 ranks over gloo (fits replaced by the CPU oracle):
    both ranks return the single-process result, which is the reference's golden output."""
    port = _free_port()
    outs = [str(tmp_path / ("rank%d.pkl" % r)) for r in range(2)]
    env = dict(os.environ, PYTHONPATH=ROOT)
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_dist_worker.py"),
                               str(r), "2", str(port), outs[r]], env=env) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    g = gold.load("c1_wrap_seed2_init4")
    fits = []
    for path in outs:
        rv = pickle.load(open(path, "rb"))
        fits.append(rv["n_fits_on_rank"])
        for k in ("ID_prob", "GT_prob", "doublet_prob", "doublet_LLR", "theta_shapes",
                  "theta_mean", "theta_sum", "LB_list"):
            assert np.array_equal(rv[k], g[k]), k
        assert rv["LB_doublet"] == g["LB_doublet"]
    # 4 restarts over 2 ranks: 2 each, plus the winner's final fit on exactly one rank
    assert sorted(fits) == [2, 3]


def test_sharded_wrap_gloo_wider_than_n_init(tmp_path):
    """A shard wider than n_init (a donor VCF fixes the genotypes -> n_init = 1, on 2 GPUs): the
    rank that owns no restart still sizes its runner, joins the all-gather and receives the
    winner (restart_batch raised on n_owned = 0 and rank 0 hung in the gather)."""
    from vireo_amd.restarts import restart_batch
    for wide in (True, False):
        assert restart_batch(8, 0, 1 << 22, wide) == 1
        assert restart_batch(4, 0, 1000, wide) == 1
    port = _free_port()
    outs = [str(tmp_path / ("rank%d.pkl" % r)) for r in range(2)]
    env = dict(os.environ, PYTHONPATH=ROOT)
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_dist_worker.py"),
                               str(r), "2", str(port), outs[r], "1"], env=env) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    g = gold.load("c1_wrap_seed2_init1")
    fits = []
    for path in outs:
        rv = pickle.load(open(path, "rb"))
        fits.append(rv["n_fits_on_rank"])
        for k in ("ID_prob", "GT_prob", "doublet_prob", "doublet_LLR", "theta_mean", "theta_sum",
                  "LB_list"):
            assert np.array_equal(rv[k], g[k]), k
    assert sorted(fits) == [0, 2]     # the restart and its refinement on rank 0, nothing on rank 1


def test_sharded_clone_mode_gloo_world2(tmp_path):
    """``BinomMixtureVB.fit(comm=)`` with 2 ranks over gloo (fits replaced by the CPU oracle):
    initialisation i on rank i % 2, every rank walks the whole random stream, the ELBOs are
    all-gathered, the owner of the first maximum re-fits and broadcasts -- both ranks return the
    single-process result, which is the reference's golden output (the notebook's known answer),
    and leave the global stream where the reference's loop would (bmm_model.py:242-254).
    n_init = 7: ranks own 4 and 3 initialisations (a padded gather)."""
    from oracle import vireo_oracle as O
    g = gold.load("mito_bmm_k3_seed1")
    for n_init, exact_gold in ((50, True), (7, False)):
        port = _free_port()
        outs = [str(tmp_path / ("bmm%d_rank%d.pkl" % (n_init, r))) for r in range(2)]
        env = dict(os.environ, PYTHONPATH=ROOT)
        procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_dist_worker_bmm.py"),
                                   str(r), "2", str(port), outs[r], str(n_init)], env=env) for r in range(2)]
        for p in procs:
            assert p.wait(timeout=600) == 0
        rvs = [pickle.load(open(o, "rb")) for o in outs]
        for name in ("ID_prob", "beta_mu", "beta_sum", "ELBO_iters", "ELBO_inits"):
            assert np.array_equal(rvs[0][name], rvs[1][name]), name
            if exact_gold:
                assert np.array_equal(rvs[0][name], g[name]), name
        assert np.array_equal(rvs[0]["rng_after"][0], rvs[1]["rng_after"][0]) and rvs[0]["rng_after"][1] == rvs[1]["rng_after"][1]
        # the shard: n_init / 2 short fits per rank (rounded up on rank 0), + the final fit on the owner
        fits = sorted(rv["n_fits_on_rank"] for rv in rvs)
        assert sum(fits) == n_init + 1 and fits[1] - fits[0] <= 2
        if exact_gold:
            assert rvs[0]["ELBO_iters"][-1] == -190779.74335041404      # examples/vireoSNP_clones.ipynb
        else:                                                           # against the oracle's own loop
            AD, DP = gold.mito()
            st = O.bmm_new(AD.shape[1], AD.shape[0], 3)
            O.bmm_fit(st, AD, DP, n_init=n_init, random_seed=1, min_iter=30)
            assert np.array_equal(rvs[0]["ELBO_inits"], st.ELBO_inits)
            assert np.array_equal(rvs[0]["ID_prob"], st.ID_prob) and np.array_equal(rvs[0]["ELBO_iters"], st.ELBO_iters)


def test_generator_jump_equals_stepping():
    """LegacyStream.skip over far distances jumps (polynomial of the MT19937 transition matrix,
    vrx_host.cpp) instead of stepping: bitwise the state stepping leaves, for the skips a restart
    shard makes at c1 and c3 sizes (one and seven foreign restarts of 8 ranks, all 28 of them),
    and the draws that follow equal np.random.rand's."""
    import ctypes as C
    from vireo_amd import _lib
    from vireo_amd.restarts import LegacyStream
    lib = _lib.lib()

    def advance(key, pos, n, out=None):
        key, p = key.copy(), C.c_int32(pos)
        _lib.check(lib.vrx_mt19937_random_sample(key.ctypes.data_as(C.POINTER(C.c_uint32)),
                                                 C.byref(p), _lib.dptr(out), int(n)))
        return key, p.value

    c1 = 952 * 4 + 3784 * 4 * 3                 # doubles one constructor draws at c1
    c3 = 50000 * 16 + 100000 * 16 * 3           # ... at c3
    np.random.seed(5)
    np.random.rand(123)                         # (a state in the middle of a block)
    _, key, pos, _, _ = np.random.get_state()
    key = np.ascontiguousarray(key, dtype=np.uint32)
    for n in (c1, 7 * c1, c3, 7 * c3, 7 * c3 + 1, 28 * c3, 12480 * 312 + 5):
        key2, pos2 = advance(key, pos, n)       # (the library's own threshold decides)
        # both paths, forced through the explicit entry point
        jk, jp = key.copy(), C.c_int32(pos)
        _lib.check(lib.vrx_mt19937_skip(jk.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(jp), int(n), 1))
        sk, sp = key.copy(), C.c_int32(pos)
        _lib.check(lib.vrx_mt19937_skip(sk.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(sp), int(n), 0))
        assert jp.value == sp.value == pos2
        assert np.array_equal(jk, sk) and np.array_equal(sk, key2)
        a, b = np.empty(1500), np.empty(1500)
        advance(jk, jp.value, 1500, a)
        advance(sk, sp.value, 1500, b)
        assert np.array_equal(a, b)
    # the stream object against NumPy itself: skip(n) == discarding rand(n)
    n = 13_000_000
    np.random.seed(9)
    np.random.rand(n)
    want = np.random.rand(4, 5)
    np.random.seed(9)
    LegacyStream().skip(n)
    assert np.array_equal(LegacyStream().rand(4, 5), want)


def test_match_and_optimal_match():
    from vireo_amd import match, optimal_match
    assert list(match([5, 9, 1], [1, 2, 5, 7, 9])) == [2, 4, 0]
    assert list(match([1, 2, 5, 7, 9], [5, 9, 1])) == [2, None, 0, None, 1]
    rng = np.random.default_rng(0)
    X = rng.random((50, 4, 3))
    perm = np.array([2, 0, 3, 1])
    i0, i1 = optimal_match(X, X[:, perm, :] + 1e-3 * rng.random((50, 4, 3)))
    assert list(i0) == [0, 1, 2, 3] and list(perm[i1]) == [0, 1, 2, 3]


def test_fast_generator_equals_oracle_generator():
    """vireo_amd.synth (one packed sort) draws the same matrices as the oracle's scipy
    construction of SURVEY.md 8(d)."""
    from oracle import vireo_oracle as O
    from vireo_amd import synth
    from vireo_amd.counts import merge_counts
    for (n, m, k, d) in [(300, 200, 3, 0.05), (2000, 1000, 4, 0.02)]:
        w = synth.donor_workload(n, m, k, d, seed=0)
        AD, DP = O.synth_donor(n, m, k, d, seed=0)
        sAD, sDP = synth.as_scipy(w)
        assert (sAD != AD).nnz == 0 and (sDP != DP).nnz == 0
        shape, ptr, idx, a, dp = merge_counts(AD, DP)
        assert np.array_equal(ptr, w["colptr"]) and np.array_equal(idx, w["rowidx"])
        assert np.array_equal(a, w["ad"]) and np.array_equal(dp, w["dp"])
    # clone mode (BASELINE.json configs[4]): the product-side generator bench.py's GPU leg uses
    for (n, m, k, seed) in [(40, 900, 3, 0), (25, 1200, 8, 5)]:
        AD, DP = synth.clone_workload(n, m, k, seed=seed)
        oAD, oDP = O.synth_clone(n, m, k, seed=seed)
        assert (AD != oAD).nnz == 0 and (DP != oDP).nnz == 0 and AD.nnz == oAD.nnz


def test_socket_exchange_world3():
    """the torch-free rendezvous that hands rank 0's RCCL unique id to the other ranks"""
    import multiprocessing as mp
    from vireo_amd import _lib
    port = _free_port()
    payload = bytes(range(_lib.UNIQUE_ID_BYTES))

    def worker(rank, q):
        from vireo_amd.dist import socket_exchange
        ex = socket_exchange(rank, 3, addr="127.0.0.1", port=port, timeout=60)
        q.put((rank, ex(payload if rank == 0 else None)))

    ctx = mp.get_context("fork")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, q)) for r in (1, 2, 0)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(3))
    for p in procs:
        p.join(timeout=30)
    assert got == {0: payload, 1: payload, 2: payload}


def test_tcp_comm_world3_and_make_comm(monkeypatch):
    """vireo_amd.dist.TcpComm (VIREO_COMM=tcp: the communicator of ranks that share one device):
    all-gather, broadcast from every root and barrier with three ranks; make_comm's choices from
    the launcher's environment."""
    import multiprocessing as mp
    from vireo_amd import _lib, dist
    port = _free_port()

    def worker(rank, q):
        from vireo_amd.dist import TcpComm
        c = TcpComm(rank, 3, port, timeout=60)
        got = c.allgather(np.array([rank, 10.0 * rank]))
        big = np.arange(5000, dtype=np.float64).reshape(50, 100) * (rank + 1)
        b = [c.bcast(big, root) for root in (0, 1, 2)]
        c.barrier()
        c.close()
        q.put((rank, got, [x[3, 7] for x in b], [x.shape for x in b]))

    ctx = mp.get_context("fork")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, q)) for r in (2, 0, 1)]
    for p in procs:
        p.start()
    res = {r: (g, v, sh) for r, g, v, sh in (q.get(timeout=120) for _ in range(3))}
    for p in procs:
        p.join(timeout=30)
    for r in range(3):
        g, v, sh = res[r]
        assert list(g) == [0.0, 0.0, 1.0, 10.0, 2.0, 20.0]
        assert v == [307.0, 614.0, 921.0] and sh == [(50, 100)] * 3      # root's array on every rank
    # make_comm: world 1 -> LocalComm; an unknown backend is an error; tcp at world 1 is local too
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "VIREO_COMM", "VIREO_FORCE_RCCL"):
        monkeypatch.delenv(k, raising=False)
    assert isinstance(dist.make_comm(), dist.LocalComm)
    monkeypatch.setenv("VIREO_COMM", "tcp")
    assert isinstance(dist.make_comm(), dist.LocalComm)
    monkeypatch.setenv("VIREO_COMM", "mpi")
    with pytest.raises(_lib.VrxError):
        dist.make_comm()


def test_legacy_stream_continues_numpy_bit_for_bit():
    """vrx_mt19937_random_sample (host C): the doubles np.random.rand would give, at every
    alignment of the 624-word state, mixed with skips and with real np.random calls."""
    from vireo_amd.restarts import LegacyStream
    st = LegacyStream()
    for seed in (0, 2, 987654321):
        shapes = [(5,), (1,), (311, 1), (2, 156), (313,), (4, 5, 3), (7919,), (0,), (3,)]
        np.random.seed(seed)
        want = [np.random.rand(*sh) for sh in shapes]
        tail = np.random.rand(4)
        np.random.seed(seed)
        got = [st.rand(*sh) for sh in shapes]
        assert all(np.array_equal(a, b) for a, b in zip(want, got))
        assert np.array_equal(np.random.rand(4), tail)        # numpy continues where C stopped
        np.random.seed(seed)
        for i, sh in enumerate(shapes):                       # skipping == drawing and dropping
            if i % 2:
                st.skip(int(np.prod(sh)))
            else:
                assert np.array_equal(st.rand(*sh), want[i])
        assert np.array_equal(st.rand(4), tail)


def test_models_pickle_without_device_handles():
    """reference models are plain NumPy objects that travel through multiprocessing.Pool
    (vireo_wrap.py:74-83); the device-problem handle a fitted model remembers must not break that"""
    import vireo_amd
    np.random.seed(0)
    m = vireo_amd.Vireo(n_cell=20, n_var=30, n_donor=3)
    m._last_counts = ctypes.c_void_p(1234)        # what a fit leaves behind (unpicklable)
    back = pickle.loads(pickle.dumps(m))
    assert not hasattr(back, "_last_counts")
    assert np.array_equal(back.ID_prob, m.ID_prob) and np.array_equal(back.GT_prob, m.GT_prob)


def test_package_surface_matches_the_reference():
    """names a user of `import vireoSNP` finds at the top level (vireoSNP/__init__.py:3-15;
    VireoBulk / plot are out of scope) and the console script"""
    import vireo_amd
    for name in ("__version__", "vcf", "base", "model", "load_VCF", "match_SNPs", "read_cellSNP",
                 "read_vartrix", "normalize", "loglik_amplify", "get_binom_coeff", "match",
                 "optimal_match", "vireo_wrap", "Vireo", "BinomMixtureVB"):
        assert hasattr(vireo_amd, name), name
    text = open(os.path.join(ROOT, "pyproject.toml")).read()
    assert 'vireo = "vireo_amd.vireo:main"' in text
    AD, DP = gold.c1()
    c = vireo_amd.get_binom_coeff(AD, DP)
    assert c.dtype == np.float32 and c.shape == (1, int((DP > 0).sum()))
    g = gold.load("binom_const")
    assert np.float32(np.sum(c)) == np.float32(g["c1"])     # summed like vireo_model.py:313
    t1 = np.array([[0.3, 29.7], [3, 3], [29.7, 0.3]])
    t2 = np.array([[364, 24197], [5886, 7475], [6075, 397.]])
    assert vireo_amd.beta_entropy(t2, t1) > 0 and vireo_amd.beta_entropy(t1, t1) == 0


def test_socket_rendezvous_ignores_strangers():
    """the unique-id exchange serves each rank once and survives a stray connection"""
    import threading
    import time
    from vireo_amd.dist import socket_exchange
    port, res = _free_port(), {}

    def run(r):
        res[r] = socket_exchange(r, 3, addr="127.0.0.1", port=port, timeout=30)(
            bytes(range(128)) if r == 0 else None)
    threads = [threading.Thread(target=run, args=(0,))]
    threads[0].start()
    time.sleep(0.3)
    with socket.create_connection(("127.0.0.1", port)) as s:
        s.sendall(b"port probe")
    for r in (1, 2):
        threads.append(threading.Thread(target=run, args=(r,)))
        threads[-1].start()
    for t in threads:
        t.join()
    assert res == {r: bytes(range(128)) for r in range(3)}


def test_float32_sum_is_numpys():
    """vrx_np_sum_f32 (host C; adds the binomial-coefficient terms, vireo_model.py:313) against
    np.sum itself at sizes around every block boundary of NumPy's chunked pairwise summation,
    and on the reference's own terms of the demo data"""
    from vireo_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(3)
    terms = gold.load("binom_const")["c1_terms"]
    cases = [terms] + [(rng.random(n) * 9).astype(np.float32)
                       for n in (0, 1, 7, 8, 9, 127, 128, 129, 255, 1000, 8191, 8192, 8193, 16385,
                                 72858, 100003, 300001)]
    for a in cases:
        a = np.ascontiguousarray(a, dtype=np.float32)
        out = ctypes.c_float(0)
        _lib.check(L.vrx_np_sum_f32(a.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), a.size,
                                    ctypes.byref(out)))
        assert np.float32(out.value) == np.sum(a), a.size
    assert np.float32(out.value) != 0


@pytest.mark.parametrize("chunk_bytes", [None, "2000"])
def test_native_text_writers_format_like_python(tmp_path, monkeypatch, chunk_bytes):
    """prob_*.tsv.gz (io_utils.py:147-170) and the donor genotype VCF (vcf_utils.py:234-296) come
    out of the library's threaded writers byte for byte as the reference's Python formatting
    produces them -- also when the rows are cut into many chunks (one gzip member each)"""
    import gzip
    from vireo_amd import io_utils, vcf_utils
    if chunk_bytes:
        monkeypatch.setenv("VIREO_WRITER_CHUNK_BYTES", chunk_bytes)
    rng = np.random.default_rng(0)
    M, K = 257, 5
    names = ["CELL%d-1" % i for i in range(M)]
    T = rng.random((M, K)) ** 8
    T[3, 2], T[4, 1], T[5, 0], T[6, 0], T[7, 3] = 0.0, 1.0, 1e-300, np.nan, 9.995e-5
    header = ["cell"] + ["donor%d" % i for i in range(K)]
    io_utils._write_table_gz(str(tmp_path / "t.tsv.gz"), header, names, T)
    want = "\t".join(header) + "\n" + "".join(
        "\t".join([names[i]] + ["%.2e" % x for x in T[i]]) + "\n" for i in range(M))
    assert gzip.open(tmp_path / "t.tsv.gz", "rt").read() == want
    io_utils._write_table_gz(str(tmp_path / "e.tsv.gz"), ["cell"], names, np.zeros((M, 0)))
    assert gzip.open(tmp_path / "e.tsv.gz", "rt").read() == "cell\n" + "".join(n + "\n" for n in names)

    N = 123
    GT = rng.dirichlet([0.3, 0.3, 0.3], (N, K))
    GT[2, 1] = [1.0, 0.0, 0.0]                       # floored at 1e-10 -> PL 100
    AD = rng.random((N, K)) * 50
    geno = vcf_utils.GenoINFO_maker(GT.copy(), AD, AD + rng.random((N, K)) * 30)
    fixed = {c: [str(x) for x in (["1"] * N if c == "CHROM" else range(N))]
             for c in ["CHROM", "POS", "ID", "REF", "ALT", "QUAL", "FILTER", "INFO"]}
    dat = dict(comments=["##fileformat=VCFv4.2", "##FORMAT=<ID=GT,old>", "##contig=<ID=1>"],
               samples=["s%d" % i for i in range(K)], variants=["v%d" % i for i in range(N)],
               FixedINFO=fixed, GenoINFO=geno)
    vcf_utils.write_VCF(str(tmp_path / "a.vcf.gz"), dat)
    plain = dict(dat, GenoINFO={k: geno[k] for k in ("GT", "AD", "DP", "PL")})   # the reference's lists
    assert plain["GenoINFO"]["PL"][2][1] == "0,100,100"
    vcf_utils.write_VCF(str(tmp_path / "b.vcf.gz"), plain)
    fast, slow = (gzip.open(tmp_path / f, "rt").read() for f in ("a.vcf.gz", "b.vcf.gz"))
    assert fast == slow and fast.count("\n") == N + 7


@pytest.mark.parametrize("tag", ["GT", "GP", "PL"])
def test_donor_genotype_codes_parse_the_same_in_bulk(tag, monkeypatch):
    """parse_donor_GPb (vcf_utils.py:299-336) converts all codes of the donor VCF in a few array
    operations; bit for bit what the one-code-at-a-time conversion gives, and irregular input
    still takes that path"""
    from vireo_amd import vcf_utils
    rng = np.random.default_rng(3)
    N, K = 400, 6
    if tag == "GT":
        pool = ["0/0", "0/1", "1/0", "1|1", "./.", ".", "1/1"]
        dat = [[pool[x] for x in rng.integers(0, len(pool), K)] for _ in range(N)]
    elif tag == "GP":
        dat = [["." if rng.random() < 0.05 else ",".join("%.4f" % v for v in rng.dirichlet([1, 1, 1]))
                for _ in range(K)] for _ in range(N)]
    else:
        dat = [["./." if rng.random() < 0.05 else ",".join(str(v) for v in rng.integers(0, 255, 3))
                for _ in range(K)] for _ in range(N)]
    fast = vcf_utils.parse_donor_GPb(dat, tag, min_prob=1e-4)
    assert vcf_utils._parse_codes_vectorised(dat, tag) is not None
    monkeypatch.setattr(vcf_utils, "_parse_codes_vectorised", lambda *a: None)
    slow = vcf_utils.parse_donor_GPb(dat, tag, min_prob=1e-4)
    assert np.array_equal(fast, slow)
    monkeypatch.undo()
    if tag != "GT":
        odd = [row[:] for row in dat]
        odd[5][2] = "1,2"                           # not three fields: the bulk path declines
        assert vcf_utils._parse_codes_vectorised(odd, tag) is None


def test_one_ahead_iterator_hands_over_items_errors_and_early_exits():
    """the helper-thread iterator behind the restart draws (vireo_wrap._one_ahead): order kept,
    a producer error re-raised in the consumer, a consumer that stops early does not hang"""
    import sys
    import threading
    import time
    W = sys.modules["vireo_amd.vireo_wrap"]
    assert list(W._one_ahead(iter(range(7)), depth=3)) == list(range(7))
    assert list(W._one_ahead(iter(()))) == []

    def broken():
        yield 1
        raise RuntimeError("draw failed")
    got = []
    with pytest.raises(RuntimeError, match="draw failed"):
        for x in W._one_ahead(broken()):
            got.append(x)
    assert got == [1]

    n0 = threading.active_count()
    it = W._one_ahead(iter(range(100)))
    assert next(it) == 0
    it.close()                                    # consumer leaves: the helper must finish
    t0 = time.time()
    while threading.active_count() > n0 and time.time() - t0 < 5:
        time.sleep(0.05)
    assert threading.active_count() <= n0


def test_tcp_comm_three_ranks():
    """tests/tcp_comm.py (the host-side communicator of tests/test_gpu_shard.py): all-gather in
    rank order, broadcast from any root, barrier -- three ranks as threads of this process."""
    import socket
    import threading
    from tests.tcp_comm import TcpComm
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    world, out, errs = 3, {}, []

    def rank_main(r):
        try:
            c = TcpComm(r, world, port, timeout=30.0)
            got = c.allgather(np.array([r + 0.5, -r]))
            big = np.arange(6.0).reshape(2, 3) * (r + 1)
            b1 = c.bcast(big, 1)
            b0 = c.bcast(np.full(4, float(r)), 0)
            c.barrier()
            out[r] = (got, b1, b0)
            c.close()
        except Exception as e:      # noqa: BLE001
            errs.append((r, e))

    ths = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    [t.start() for t in ths]
    [t.join(60) for t in ths]
    assert not errs, errs
    for r in range(world):
        got, b1, b0 = out[r]
        assert np.array_equal(got, [0.5, 0, 1.5, -1, 2.5, -2])
        assert np.array_equal(b1, np.arange(6.0).reshape(2, 3) * 2) and b1.shape == (2, 3)
        assert np.array_equal(b0, np.zeros(4))


# ---------------------------------------------------------------- the launcher (--nGPU / --gpus N)
_STUB_RANK = r'''
import os, sys, time
r, w = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert os.environ["LOCAL_RANK"] == str(r) and os.environ["MASTER_ADDR"] == "127.0.0.1"
assert int(os.environ["MASTER_PORT"]) > 0 and os.environ["VIREO_LAUNCHED"] == "1"
mode = sys.argv[1]
if mode == "fail" and r == 2:
    sys.exit(7)
if mode == "fail":
    time.sleep(60)          # a rank waiting in a collective for the one that died
print("rank %d of %d says %s" % (r, w, sys.argv[2]))
'''


def test_launcher_spawns_ranks_and_forwards_rank0(tmp_path):
    """vireo_amd/launch.py: N copies of a command with the rendezvous environment of
    torch.distributed.run, rank 0's stdout = the launcher's, the others' on stderr"""
    import subprocess
    stub = tmp_path / "stub.py"
    stub.write_text(_STUB_RANK)
    code = ("import sys; sys.path.insert(0, %r); from vireo_amd import launch; "
            "sys.exit(launch.spawn_ranks([sys.executable, %r, 'ok', 'hello'], 4))" % (ROOT, str(stub)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, env=env)
    assert p.returncode == 0, p.stderr
    assert p.stdout == "rank 0 of 4 says hello\n"
    assert sorted(ln for ln in p.stderr.splitlines() if ln.startswith("rank")) == [
        "rank %d of 4 says hello" % r for r in (1, 2, 3)]


def test_launcher_stops_the_others_when_a_rank_fails(tmp_path):
    """a failing rank's code is the launcher's, and the ranks that would wait for it forever in
    a collective are terminated"""
    import subprocess
    import time
    stub = tmp_path / "stub.py"
    stub.write_text(_STUB_RANK)
    code = ("import sys; sys.path.insert(0, %r); from vireo_amd import launch; "
            "sys.exit(launch.spawn_ranks([sys.executable, %r, 'fail', 'x'], 4))" % (ROOT, str(stub)))
    t0 = time.time()
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert p.returncode == 7
    assert time.time() - t0 < 30
    assert "rank 2 exited with code 7" in p.stderr and p.stdout == ""


def test_launched_externally_and_device_map():
    from vireo_amd import launch
    env = launch.rank_env(3, 8, 29400, base={"PATH": "/bin"}, devices=[0] * 8)
    assert env["RANK"] == "3" and env["LOCAL_RANK"] == "0" and env["WORLD_SIZE"] == "8"
    assert env["MASTER_PORT"] == "29400" and env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    p = launch.free_port()
    assert 0 < p < 65535


def test_bench_side_legs_cannot_cost_the_headline(capsys):
    """bench.py's side legs (c2, c5, doublet, c3_skew, e2e) are wrapped: an exception inside one
    becomes an {"error": ...} entry of the JSON line, never a lost headline"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench.side_leg("ok", lambda x: {"v": x + 1}, 1) == {"v": 2}
    out = bench.side_leg("broken", lambda: 1 / 0)
    assert set(out) == {"error"} and "ZeroDivisionError" in out["error"] and "broken" in out["error"]
    assert "broken leg failed" in capsys.readouterr().err
    B = bench.algorithmic_bytes(100000, 50000, 16, 3, 99006675)      # 2.58 GB at c3
    assert B["total"] == B["variant"] + B["cell"] + B["dense"] and abs(B["total"] - 2.5816e9) < 1e6
    assert B["cell"] == 12 * 99006675 + 4 * 50001 + 16 * 100000 * 16 + 8 * 50000 * 16


# ---------------------------------------------------------------- round 6: first contact with a multi-GPU node
def test_free_port_reserves_the_three_rendezvous_ports():
    """MASTER_PORT, the unique-id port (+ 1) and the TcpComm harness's port (+ 2) are all free"""
    import socket
    from vireo_amd import launch
    p = launch.free_port()
    for q in (p, p + 1, p + 2):
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
            s.bind(("127.0.0.1", q))


def test_tcp_comm_rank0_survives_strangers():
    """a connection that closes early or never speaks must not abort (or stall) rank 0's
    accept loop, and a peer cannot announce an absurd message length"""
    import socket
    import struct
    import threading
    from vireo_amd import dist
    port = _free_port()
    out, errs = {}, []

    def rank_main(r):
        try:
            c = dist.TcpComm(r, 2, port, timeout=30.0, rdzv_timeout=60.0)
            out[r] = c.allgather(np.array([float(r)]))
            c.close()
        except Exception as e:      # noqa: BLE001
            errs.append((r, e))

    t0 = threading.Thread(target=rank_main, args=(0,))
    t0.start()
    time.sleep(0.3)
    quick = socket.create_connection(("127.0.0.1", port), timeout=5)
    quick.close()                                   # closes before saying hello
    silent = socket.create_connection(("127.0.0.1", port), timeout=5)   # never says anything
    t1 = threading.Thread(target=rank_main, args=(1,))
    t1.start()
    t0.join(60)
    t1.join(60)
    silent.close()
    assert not errs, errs
    assert list(out[0]) == [0.0, 1.0] and list(out[1]) == [0.0, 1.0]
    a, b = socket.socketpair()
    a.sendall(struct.pack("<q", 1 << 40))
    with pytest.raises(ConnectionError):
        dist._recv(b)
    a.close()
    b.close()


def test_winner_rules_follow_numpy_with_nans():
    """vireo_wrap keeps np.argmax(ELBOs) (vireo_wrap.py:89-90: a NaN wins, the first one);
    BinomMixtureVB.fit keeps the last i with elbo[i] > np.max(elbo[:i]) (bmm_model.py:248: a NaN
    freezes the choice).  Every rank applies the rule to ITS restarts in order; the owner of the
    global winner must end up holding it."""
    from vireo_amd.bmm_model import _is_record
    from vireo_amd.dist import first_record, my_restarts
    from vireo_amd.restarts import _argmax_takes
    nan = float("nan")
    rng = np.random.default_rng(3)
    cases = [[1.0], [1, 3, 2, 3, 5, 4], [2, 2, 1], [nan, 1, 2], [1, nan, 9], [1, 5, nan, 9, nan], [3, 1, 9, nan]]
    cases += [list(rng.normal(size=n)) for n in (7, 32, 33)]
    for e in cases:
        e = np.array(e, dtype=float)
        # the reference's loop, literally
        best = None
        inits = []
        for i, v in enumerate(e):
            inits.append(v)
            if i == 0 or v > np.max(inits[:-1]):
                best = i
        assert first_record(e) == best
        seen, kept = [], None
        for i, v in enumerate(e):
            if _is_record(seen, v):
                kept = i
            seen.append(v)
        assert kept == best
        for world in (1, 2, 3, 8):
            win = int(np.argmax(e))
            owner = win % world
            held = None
            for i in my_restarts(len(e), owner, world):
                if _argmax_takes(held, e[i]):
                    held = (e[i], i)
            assert held[1] == win, (e, world)


def test_bench_refuses_an_inherited_tcp_communicator():
    """VIREO_COMM=tcp left in an environment must not turn `bench.py --gpus 8` into a
    host-socket run that looks like an RCCL one -- refused (exit code 2) before anything starts,
    unless --comm tcp is on the command line itself"""
    import subprocess
    env = dict(os.environ, VIREO_COMM="tcp")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "20",
                        "--warmup", "5"], capture_output=True, text=True, timeout=120, env=env)
    assert p.returncode == 2 and p.stdout == "" and time.time() - t0 < 60
    assert "VIREO_COMM=tcp" in p.stderr and "--comm tcp" in p.stderr


def test_failing_rccl_initialisation_ends_every_rank_with_the_reason(tmp_path):
    """First contact: when the RCCL communicator cannot be made (here: no GPU at all -- ncclGetUniqueId
    fails on rank 0 while rank 1 waits for the id) every rank is gone within 60 s, the launcher
    returns non-zero, and stderr carries RCCL's error string and the tail of its NCCL_DEBUG=WARN log."""
    import subprocess
    from vireo_amd import _lib
    if _lib.device_count() > 0:
        pytest.skip("a GPU is visible here: the communicator would come up")
    stub = tmp_path / "rank.py"
    stub.write_text("import sys\nsys.path.insert(0, %r)\nfrom vireo_amd import dist\n"
                    "c = dist.make_comm()\nprint('up', c.backend)\n" % ROOT)
    code = ("import sys; sys.path.insert(0, %r); from vireo_amd import launch; "
            "sys.exit(launch.spawn_ranks([sys.executable, %r], 4))" % (ROOT, str(stub)))
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "VIREO_COMM", "NCCL_DEBUG", "NCCL_DEBUG_FILE")}
    t0 = time.time()
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=150, env=env)
    assert p.returncode != 0 and time.time() - t0 < 60
    assert "up rccl" not in p.stdout
    assert "failed:" in p.stderr and "NCCL_DEBUG=WARN tail" in p.stderr
    assert "stopping the other ranks" in p.stderr


def test_rccl_init_watchdog_ends_a_rank_that_hangs(tmp_path):
    """a rank whose ncclCommInitRank neither returns nor fails (a peer died behind the rendezvous) is
    ended by the watchdog with exit code 3 and a message (the library call is replaced by a sleep)"""
    import subprocess
    stub = tmp_path / "hang.py"
    stub.write_text(
        "import sys, time\nsys.path.insert(0, %r)\nfrom vireo_amd import dist, _lib\n"
        "class Fake:\n"
        "    def vrx_comm_unique_id(self, uid): return 0\n"
        "    def vrx_comm_create(self, *a): time.sleep(60); return 0\n"
        "    def vrx_last_error(self): return b''\n"
        "_lib.lib = lambda: Fake()\n"
        "dist.RcclComm(0, 1, 0, lambda raw: raw, init_timeout=1.0)\nprint('not reached')\n" % ROOT)
    t0 = time.time()
    p = subprocess.run([sys.executable, str(stub)], capture_output=True, text=True, timeout=120)
    assert p.returncode == 3 and time.time() - t0 < 30
    assert "ncclCommInitRank had not returned" in p.stderr and "not reached" not in p.stdout


def test_balance_policy_order_of_precedence(monkeypatch):
    """Balanced slabs are a build option: the caller's word first, then VIREO_BALANCE, then the
    announced number of iterations against VIREO_BALANCE_MIN_ITERS (default 1200)."""
    from vireo_amd.counts import balance_policy
    monkeypatch.delenv("VIREO_BALANCE", raising=False)
    monkeypatch.delenv("VIREO_BALANCE_MIN_ITERS", raising=False)
    assert balance_policy() is False
    assert balance_policy(expected_iterations=840) is False        # bench.py's c4 job: 32 x 20 + 200
    assert balance_policy(expected_iterations=1200) is True        # vireo's defaults: 50 x 20 + 200
    assert balance_policy(expected_iterations=1200, nnz=10 ** 8) is True
    assert balance_policy(expected_iterations=1200, nnz=16 * 10 ** 8) is False      # 16x c3: from 6 000 on
    assert balance_policy(expected_iterations=6000, nnz=16 * 10 ** 8) is True
    assert balance_policy(balance=False, expected_iterations=10 ** 6) is False
    assert balance_policy(balance=True) is True
    monkeypatch.setenv("VIREO_BALANCE_MIN_ITERS", "500")
    assert balance_policy(expected_iterations=840) is True
    monkeypatch.setenv("VIREO_BALANCE", "0")
    assert balance_policy(expected_iterations=10 ** 6) is False
    assert balance_policy(balance=True) is True                     # the caller outranks the environment
    monkeypatch.setenv("VIREO_BALANCE", "1")
    assert balance_policy() is True
