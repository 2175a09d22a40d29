"""AD/DP count matrices resident in HBM.

The reference hands scipy.sparse matrices (CSC from read_cellSNP, io_utils.py:57; CSR
from the VCF path, vcf_utils.py:204; int64 or float64; or dense ndarrays) straight to
SciPy's SpMM.  Here the pair is canonicalised ONCE on the host into a single CSC
pattern (the union of both patterns) with an int32 (ad, dp) pair per entry, uploaded,
and kept on the device behind a ``DeviceCounts`` handle; "BD = DP - AD" is never
formed.  Format independence of the reference's results makes
this canonicalisation safe.
"""
import ctypes as C
import weakref

import numpy as np
from scipy.sparse import csc_matrix, issparse

from . import _lib

_I32_MAX = 2 ** 31 - 1


def _as_canonical_csc(X, name):
    if not issparse(X):
        X = np.asarray(X)
        if X.ndim != 2:
            raise ValueError("%s must be a 2-D matrix" % name)
    X = csc_matrix(X)           # no copy when X is already CSC
    if not X.has_canonical_format:
        X = X.copy()
        X.sum_duplicates()      # also sorts the indices
    if X.data.dtype.kind not in "fiub":
        raise ValueError("%s has unsupported dtype %s" % (name, X.data.dtype))
    return X


_KIND = {np.dtype(np.int32): 0, np.dtype(np.int64): 1, np.dtype(np.float64): 2}


def _view(X):
    """(indptr, indices, data, is64(indptr), is64(indices), count kind) for vrx_merge_counts"""
    data = X.data if X.data.dtype in _KIND else X.data.astype(
        np.float64 if X.data.dtype.kind == "f" else np.int64)
    ptr = np.ascontiguousarray(X.indptr)
    idx = np.ascontiguousarray(X.indices)
    if ptr.dtype not in (np.int32, np.int64):
        ptr = ptr.astype(np.int64)
    if idx.dtype not in (np.int32, np.int64):
        idx = idx.astype(np.int64)
    return ptr, idx, np.ascontiguousarray(data), int(ptr.dtype == np.int64), \
        int(idx.dtype == np.int64), _KIND[data.dtype]


def merge_counts(AD, DP):
    """-> (shape, colptr int64[M+1], rowidx int32[nnz], ad int32[nnz], dp int32[nnz])
    on the union pattern of AD and DP (entries where both are zero are dropped): a per-column
    two-pointer merge in the library (vrx_merge_counts, all cores), which also checks that the
    counts are non-negative integers below 2^31."""
    AD = _as_canonical_csc(AD, "AD")
    DP = _as_canonical_csc(DP, "DP")
    if AD.shape != DP.shape:
        raise ValueError("AD %s and DP %s differ in shape" % (AD.shape, DP.shape))
    n_var, n_cell = AD.shape
    a, d = _view(AD), _view(DP)
    vp = C.c_void_p

    def call(colptr, rowidx, ad, dp):
        i32 = C.POINTER(C.c_int32)
        rc = _lib.lib().vrx_merge_counts(
            n_var, n_cell, a[0].ctypes.data_as(vp), a[1].ctypes.data_as(vp), a[2].ctypes.data_as(vp),
            a[3], a[4], a[5], d[0].ctypes.data_as(vp), d[1].ctypes.data_as(vp),
            d[2].ctypes.data_as(vp), d[3], d[4], d[5],
            colptr.ctypes.data_as(C.POINTER(C.c_int64)),
            None if rowidx is None else rowidx.ctypes.data_as(i32),
            None if ad is None else ad.ctypes.data_as(i32),
            None if dp is None else dp.ctypes.data_as(i32), 0)
        if rc != 0:
            raise ValueError(_lib.lib().vrx_last_error().decode())

    colptr = np.zeros(n_cell + 1, dtype=np.int64)
    call(colptr, None, None, None)
    np.cumsum(colptr, out=colptr)
    nnz = int(colptr[-1])
    rowidx = np.empty(nnz, dtype=np.int32)
    ad = np.empty(nnz, dtype=np.int32)
    dp = np.empty(nnz, dtype=np.int32)
    call(colptr, rowidx, ad, dp)
    return ((n_var, n_cell), colptr, rowidx, ad, dp)


def balance_policy(balance=None, expected_iterations=None, nnz=None):
    """Should the problem be built with *balanced slabs* (``vrx_problem_create2``, VRX_PROBLEM_BALANCED)?

    They shorten the sparse passes by 15-17 % at c3 and cost a one-off ~0.065 s at 1e8 entries (break-even at
    c3: ~800 iterations on the same problem).  ``balance`` True / False decides; None: the environment
    (VIREO_BALANCE=1 / 0), else on when the caller expects at least VIREO_BALANCE_MIN_ITERS iterations (default
    1200; ``vireo_wrap`` announces n_init x max_iter_init + 200) -- five times as many from 5e8 entries on,
    where the build's transient buffers (tens of GB allocated and freed) cost more than in proportion: 16x c3
    builds in 12.9 instead of 4.8 s and breaks even at ~5 000 iterations.  The library applies the flag only
    where it can (LDS-resident passes on AD/BD words); ``DeviceCounts.build_info`` says what was built."""
    import os
    if balance is not None:
        return bool(balance)
    env = os.environ.get("VIREO_BALANCE")
    if env in ("0", "1"):
        return env == "1"
    need = int(os.environ.get("VIREO_BALANCE_MIN_ITERS", "1200"))
    if nnz is not None and nnz >= 500_000_000:
        need *= 5
    return expected_iterations is not None and expected_iterations >= need


class DeviceCounts:
    """(AD, DP) on one GPU, in both orientations (C handle ``vrx_problem``)."""

    def __init__(self, AD, DP, device=None, _merged=None, balance=None, expected_iterations=None):
        _lib.require_gpu()
        if device is None:
            device = default_device()
        if _merged is None:
            _merged = merge_counts(AD, DP)
        (self.n_var, self.n_cell), colptr, rowidx, ad, dp = _merged
        colptr = np.ascontiguousarray(colptr, dtype=np.int64)
        rowidx = np.ascontiguousarray(rowidx, dtype=np.int32)
        ad = np.ascontiguousarray(ad, dtype=np.int32)
        dp = np.ascontiguousarray(dp, dtype=np.int32)
        self.shape = (self.n_var, self.n_cell)
        self.nnz = int(rowidx.size)
        self.device = device
        self._h = C.c_void_p()
        flags = _lib.PROBLEM_BALANCED if balance_policy(balance, expected_iterations, self.nnz) else 0
        _lib.check(_lib.lib().vrx_problem_create2(
            device, self.n_var, self.n_cell, self.nnz,
            colptr.ctypes.data_as(C.POINTER(C.c_int64)),
            rowidx.ctypes.data_as(C.POINTER(C.c_int32)),
            ad.ctypes.data_as(C.POINTER(C.c_int32)),
            dp.ctypes.data_as(C.POINTER(C.c_int32)), flags, C.byref(self._h)))
        self._binom = None
        self._fin = weakref.finalize(self, _lib.lib().vrx_problem_destroy, self._h)

    @classmethod
    def from_merged(cls, shape, colptr, rowidx, ad, dp, device=None, balance=None, expected_iterations=None):
        """from an already merged CSC pattern carrying (ad, dp) per entry (row indices
        strictly increasing inside each column; validated by the library)."""
        return cls(None, None, device=device, balance=balance, expected_iterations=expected_iterations,
                   _merged=((int(shape[0]), int(shape[1])), colptr, rowidx, ad, dp))

    def build_info(self):
        """what the library built: balanced slabs per orientation, the seconds they added, device build"""
        a = np.zeros(4)
        _lib.check(_lib.lib().vrx_problem_build_info(self._h, _lib.dptr(a)))
        return dict(balanced_variant=bool(a[0]), balanced_cell=bool(a[1]), balance_seconds=float(a[2]),
                    device_built=bool(a[3]))

    @property
    def handle(self):
        return self._h

    def binom_const(self):
        """np.sum(get_binom_coeff(AD, DP)): the float32 scalar the reference adds to the ELBO
        trace (vireo_model.py:313, bmm_model.py:239).  The float32 terms come from the device,
        in the reference's row-major entry order, and are added in NumPy's float32 pairwise
        order, so the value equals the reference's bit for bit."""
        if self._binom is None:
            s = C.c_double(0.0)
            _lib.check(_lib.lib().vrx_problem_binom_const(self._h, C.byref(s)))
            self._binom = np.float32(s.value)
        return self._binom

    def digest(self):
        """checksums of the device arrays (vrx_problem_digest): equal for identical builds"""
        out = (C.c_uint64 * 12)()
        _lib.check(_lib.lib().vrx_problem_digest(self._h, out))
        return [int(x) for x in out]

    def n_vars(self):
        """number of variants with DP > 0 per cell (vireo.py:191)."""
        out = np.zeros(self.n_cell, dtype=np.int32)
        _lib.check(_lib.lib().vrx_problem_n_vars(self._h, out.ctypes.data_as(C.POINTER(C.c_int32))))
        return out

    def donor_reads(self, ID_prob):
        """(AD @ ID_prob, DP @ ID_prob), each (n_var, n_donor): the expected reads per donor
        the command line writes to GT_donors.vireo.vcf.gz (vireo.py:240-241)."""
        ID = _lib.f64(ID_prob)
        if ID.ndim != 2 or ID.shape[0] != self.n_cell:
            raise ValueError("ID_prob has shape %s" % (ID.shape,))
        A = np.empty((self.n_var, ID.shape[1]))
        D = np.empty((self.n_var, ID.shape[1]))
        _lib.check(_lib.lib().vrx_problem_donor_reads(self._h, ID.shape[1], _lib.dptr(ID),
                                                      _lib.dptr(A), _lib.dptr(D)))
        return A, D

    def close(self):
        self._fin()


# (AD, DP) -> DeviceCounts.  vireo_wrap calls fit() n_init+1 times on the same matrices
# (vireo_wrap.py:84-94); upload once.  A cached entry is reused only for the SAME objects
# (held by weak reference: an entry dies with its matrices, so a recycled id() or buffer
# address can never hit it) whose buffers still hash to what was uploaded (in-place edits of
# data, indices or indptr are noticed).
_cache = []          # [(ref(AD), ref(DP) or None, device, digest, DeviceCounts)]
_CACHE_MAX = 2


def _hasher():
    try:
        import xxhash
        return xxhash.xxh3_64()
    except ImportError:         # pragma: no cover
        import hashlib
        return hashlib.blake2b(digest_size=8)


def _digest(*mats):
    """position-sensitive content hash of the matrices' buffers (and shapes / formats)"""
    h = _hasher()
    for X in mats:
        if X is None:
            h.update(b"none")
        elif issparse(X):
            h.update(("%s%s" % (X.format, X.shape)).encode())
            for part in (X.data, X.indices, X.indptr) if X.format in ("csc", "csr") else \
                    (X.tocsc().data,):
                h.update(np.ascontiguousarray(part).view(np.uint8))
        else:
            A = np.ascontiguousarray(X)
            h.update(("dense%s%s" % (A.shape, A.dtype)).encode())
            h.update(A.view(np.uint8).reshape(-1))
    return h.hexdigest()


def default_device():
    """VIREO_DEVICE, else LOCAL_RANK (one process per GPU under torchrun), else 0."""
    import os
    return int(os.environ.get("VIREO_DEVICE", os.environ.get("LOCAL_RANK", "0")))


def device_counts(AD, DP=None, device=None, expected_iterations=None):
    """Accepts a DeviceCounts (returned as is) or an (AD, DP) pair in any of the
    reference's input formats.  ``expected_iterations``: how many iterations the caller will run on the
    problem (``balance_policy``; only a problem built here takes it into account, a cached one is reused)."""
    if isinstance(AD, DeviceCounts):
        return AD
    if device is None:
        device = default_device()
    _cache[:] = [e for e in _cache if e[0]() is not None and (e[1] is None or e[1]() is not None)]
    digest = None
    for ra, rd, dev, dig, dc in _cache:
        if ra() is AD and (rd() if rd is not None else None) is DP and dev == device:
            digest = digest or _digest(AD, DP)
            if dig == digest:
                return dc
    dc = DeviceCounts(AD, DP, device=device, expected_iterations=expected_iterations)
    try:
        entry = (weakref.ref(AD), weakref.ref(DP) if DP is not None else None, device,
                 digest or _digest(AD, DP), dc)
    except TypeError:            # not weak-referenceable (a list, ...): do not cache
        return dc
    del _cache[:max(0, len(_cache) - _CACHE_MAX + 1)]   # evicted handles die with their last reference
    _cache.append(entry)
    return dc


def clear_cache():
    del _cache[:]
