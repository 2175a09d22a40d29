"""Synthetic AD/DP workloads of BASELINE.json (generator fixed by SURVEY.md section 8(d)).

Same random draws, in the same order, as ``oracle.vireo_oracle.synth_donor`` (tested equal
in tests/test_host_cpu.py), but the COO -> CSC conversion is ONE sort of packed
(column, row) keys carrying (ad << 32 | dp) instead of two scipy conversions, and the
result is the merged (ad, dp) CSC that ``DeviceCounts.from_merged`` uploads directly.
"""
import numpy as np
from scipy.sparse import csc_matrix

CONFIGS = {
    # name: (N variants, M cells, K donors, density)        BASELINE.json configs[1..3]
    "c2": (10000, 5000, 4, 0.01),
    "c3": (100000, 50000, 16, 0.02),
    "mid": (50000, 20000, 16, 0.02),
    "small": (2000, 1000, 4, 0.02),
    # crossover probes for the LDS-resident passes (nnz ~ 1 M, 2 M, 4.5 M, 8 M, 16 M)
    "x1m": (14000, 7000, 8, 0.01),
    "x2m": (20000, 10000, 8, 0.01),
    "x4m": (30000, 15000, 8, 0.01),
    "x8m": (40000, 20000, 8, 0.01),
    "x16m": (56000, 28000, 8, 0.01),
    # locality probes (same nnz and nnz/row as c3, dense operand of ONE pass fits an XCD's L2)
    "l2c": (6250, 50000, 16, 0.32),     # W  = 1.6 MB
    "l2v": (100000, 6250, 16, 0.16),    # ID = 0.8 MB
}


# Heavy-tailed companion of c3: the same shape, donors and target density,
# but per-variant coverage and per-cell depth drawn with log-normal weights (sigma 1.0 / 0.7) --
# what a real cellSNP matrix looks like (vireoSNP/utils/io_utils.py:42-59 loads those): row
# lengths from a handful to tens of thousands, counts in the hundreds where popular variants
# meet deep cells.  ``donor_workload(*CONFIGS["c3"], seed=0, skew=C3_SKEW)``.
C3_SKEW = (1.0, 0.7)


def _skewed_draw(rng, n, size, sigma):
    """indices in [0, n) with log-normal(sigma) weights: heavy-tailed coverage / depth"""
    w = rng.lognormal(0.0, sigma, n)
    cdf = np.cumsum(w)
    cdf /= cdf[-1]
    return np.minimum(np.searchsorted(cdf, rng.random(size)), n - 1)


def donor_workload(N, M, K, density, seed=0, skew=None):
    """-> dict(shape, colptr int64, rowidx int32, ad int32, dp int32) on DP's pattern
    (duplicate (row, col) draws summed), plus the planted structure: GT (N, K) genotype classes
    and z (M,) donor of every cell.  skew=(sigma_variant, sigma_cell) replaces the
    uniform (variant, cell) draws of the 8(d) generator by log-normal weighted ones (real
    scRNA-seq data: per-variant coverage and per-cell depth are heavy-tailed); it is a
    robustness workload, not a BASELINE.json configuration."""
    rng = np.random.default_rng(seed)
    nnz_t = int(N * M * density)
    if skew is None:
        r = rng.integers(0, N, nnz_t)
        c = rng.integers(0, M, nnz_t)
    else:
        r = _skewed_draw(rng, N, nnz_t, skew[0])
        c = _skewed_draw(rng, M, nnz_t, skew[1])
    dp = 1 + rng.poisson(1.0, nnz_t)
    GT = rng.integers(0, 3, (N, K))
    z = rng.integers(0, K, M)
    theta = np.array([0.01, 0.5, 0.99])[GT[r, z[c]]]
    ad = rng.binomial(dp, theta)
    del theta
    key = c * np.int64(N) + r
    del r, c
    val = dp.astype(np.int64) | (ad.astype(np.int64) << 32)
    del dp, ad
    order = np.argsort(key, kind="stable")
    key = key[order]
    val = val[order]
    del order
    first = np.empty(nnz_t, dtype=bool)
    first[0] = True
    np.not_equal(key[1:], key[:-1], out=first[1:])
    starts = np.flatnonzero(first)
    del first
    val = np.add.reduceat(val, starts)       # duplicates: both 32-bit fields add, no carry
    key = key[starts]
    col = key // N
    rowidx = (key - col * N).astype(np.int32)
    colptr = np.zeros(M + 1, dtype=np.int64)
    np.cumsum(np.bincount(col, minlength=M), out=colptr[1:])
    return dict(shape=(N, M), colptr=colptr, rowidx=rowidx,
                ad=(val >> 32).astype(np.int32), dp=(val & 0xFFFFFFFF).astype(np.int32), GT=GT, z=z)


def planted_gt_prior(GT, sharp):
    """(N, K, 3) genotype prior from genotype classes: ``sharp`` on the given class, the rest
    shared -- what the command makes of a donor VCF's GT / PL field (vireo.py:165-170,
    vcf_utils.parse_donor_GPb) before ``set_prior`` clips it (vireo_model.py:129-137)."""
    N, K = GT.shape
    P = np.full((N, K, 3), (1.0 - sharp) / 2.0)
    np.put_along_axis(P, GT[:, :, None], sharp, axis=2)
    return P


def big_workload(N, M, K, density, seed=0, threads=32, block_cols=2048):
    """The 8(d) donor model at sizes where its generator's global sort is the bottleneck (16x c3 and
    beyond: 1.6e9 ... 2.2e9 entries): the same planted structure (genotype classes GT, donors z,
    theta = 0.01 / 0.5 / 0.99, dp = 1 + Poisson(1), ad ~ Binomial(dp, theta)) on a Bernoulli(density)
    pattern drawn column block by column block as geometric GAPS -- positions come out sorted and
    distinct, so there is nothing to sort or merge -- on several threads (NumPy's generators release
    the GIL).  NOT the 8(d) generator (its pattern has duplicate draws summed): a size probe, used by
    tests/perf/big_probe.py and the big-problem test.  -> the dict of ``donor_workload`` (int32
    rowidx / ad / dp, int64 colptr, GT, z)."""
    from concurrent.futures import ThreadPoolExecutor
    root = np.random.default_rng(seed)
    GT = root.integers(0, 3, (N, K)).astype(np.int8)
    z = root.integers(0, K, M).astype(np.int32)
    tv = np.array([0.01, 0.5, 0.99])
    blocks = [(c0, min(M, c0 + block_cols)) for c0 in range(0, M, block_cols)]
    seeds = np.random.SeedSequence(seed + 1).spawn(len(blocks))

    def one(i):
        c0, c1 = blocks[i]
        rng = np.random.default_rng(seeds[i])
        L = (c1 - c0) * N                                  # slots of this block, column-major
        pos = np.zeros(0, dtype=np.int64)
        at = -1
        while at < L:                                      # (one round almost always)
            g = rng.geometric(density, size=int((L - max(at, 0)) * density * 1.02) + 4096)
            q = at + np.cumsum(g)
            at = int(q[-1])
            pos = np.concatenate([pos, q])
        pos = pos[pos < L]
        col = pos // N
        row = (pos - col * N).astype(np.int32)
        dp = (1 + rng.poisson(1.0, pos.size)).astype(np.int32)
        th = tv[GT[row, z[c0 + col]]]
        ad = rng.binomial(dp, th).astype(np.int32)
        cnt = np.bincount(col, minlength=c1 - c0)
        return row, ad, dp, cnt

    with ThreadPoolExecutor(max(1, threads)) as ex:
        parts = list(ex.map(one, range(len(blocks))))
    colptr = np.zeros(M + 1, dtype=np.int64)
    np.cumsum(np.concatenate([p[3] for p in parts]), out=colptr[1:])
    nnz = int(colptr[-1])
    rowidx, ad, dp = (np.empty(nnz, dtype=np.int32) for _ in range(3))
    at = 0
    for row, a, d, _ in parts:
        rowidx[at:at + row.size], ad[at:at + row.size], dp[at:at + row.size] = row, a, d
        at += row.size
    return dict(shape=(N, M), colptr=colptr, rowidx=rowidx, ad=ad, dp=dp, GT=GT.astype(np.int64), z=z.astype(np.int64))


def as_scipy(w):
    """(AD, DP) int64 CSC like scipy's coo->csc of the generator (AD's zeros dropped)."""
    DP = csc_matrix((w["dp"].astype(np.int64), w["rowidx"], w["colptr"]), shape=w["shape"])
    AD = csc_matrix((w["ad"].astype(np.int64), w["rowidx"].copy(), w["colptr"].copy()),
                    shape=w["shape"])
    AD.eliminate_zeros()
    return AD, DP


def clone_workload(N=200, M=200000, K=8, seed=0):
    """BASELINE.json configs[4] (BinomMixtureVB clone mode; generator fixed by SURVEY.md
    section 8(d)): 90 % dense, DP ~ Poisson(50), allele frequencies ~ Beta(0.3, 3) per
    (variant, clone).  Same draws in the same order as ``oracle.vireo_oracle.synth_clone``
    (tested equal in tests/test_host_cpu.py) -> (AD, DP) as scipy CSC."""
    rng = np.random.default_rng(seed)
    mask = rng.random((N, M)) < 0.9
    dp = rng.poisson(50, (N, M)) * mask
    del mask
    z = rng.integers(0, K, M)
    af = rng.beta(0.3, 3, (N, K))
    ad = rng.binomial(dp, af[:, z])
    return csc_matrix(ad), csc_matrix(dp)


def write_cellsnp_folder(w, path, n_threads=0):
    """A cellSNP output folder (what ``read_cellSNP`` loads, io_utils.py:42-59) holding workload
    ``w`` of ``donor_workload``: cellSNP.tag.AD.mtx / cellSNP.tag.DP.mtx (MatrixMarket coordinate
    integer, entries variant-major like cellSNP writes them, zeros of AD left out),
    cellSNP.base.vcf.gz (one record per variant) and cellSNP.samples.tsv.  The matrices are
    written by the library (vrx_mtx_write: all cores; NumPy's savetxt needs minutes at 1e8
    entries).  -> bytes written"""
    import ctypes as C
    import gzip
    import os
    from . import _lib
    N, M = w["shape"]
    os.makedirs(path, exist_ok=True)
    cols = np.repeat(np.arange(M, dtype=np.int32), np.diff(w["colptr"]))
    order = np.argsort(w["rowidx"], kind="stable")           # variant-major, cells increasing
    rows, cols = w["rowidx"][order], cols[order]
    i32 = C.POINTER(C.c_int32)
    total = 0
    for name, val in (("AD", w["ad"][order]), ("DP", w["dp"][order])):
        keep = val > 0
        r, c, v = (np.ascontiguousarray(x[keep], dtype=np.int32) for x in (rows, cols, val))
        f = os.path.join(path, "cellSNP.tag.%s.mtx" % name)
        _lib.check(_lib.lib().vrx_mtx_write(f.encode(), N, M, r.size, r.ctypes.data_as(i32),
                                            c.ctypes.data_as(i32), v.ctypes.data_as(i32)))
        total += os.path.getsize(f)
    f = os.path.join(path, "cellSNP.base.vcf.gz")
    with gzip.open(f, "wt", compresslevel=1) as fh:
        fh.write("##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n")
        fh.write("".join("1\t%d\t.\tA\tG\t.\tPASS\tAD=1;DP=2;OTH=0\n" % (i + 1) for i in range(N)))
    total += os.path.getsize(f)
    f = os.path.join(path, "cellSNP.samples.tsv")
    with open(f, "w") as fh:
        fh.write("\n".join("CELL%07d-1" % i for i in range(M)) + "\n")
    return total + os.path.getsize(f)
