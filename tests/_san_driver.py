"""Driver of tests/test_host_sanitizers_cpu.py (runs in its own process, under
LD_PRELOAD=libasan): binds vireo_amd._lib to an AddressSanitizer + UBSan build of
csrc/vrx_host.cpp (the host-only translation unit: MatrixMarket parser, count merge, text / VCF
writers, MT19937 continuation + jump, NumPy float32 sum) and drives it with

  1. the CPU tests that reach those entry points through the product's Python layer
     (tests/test_cli_io_cpu.py, the host-function tests of tests/test_host_cpu.py), and
  2. malformed MatrixMarket files (truncated, garbage tokens, huge numbers, out-of-range
     indices, wrong counts, CRLF, no final newline): every one must come back as a status code
     or as the same matrix scipy.io.mmread parses -- never as a sanitizer report.

usage: _san_driver.py <libvrx_host_san.so> <scratch dir>
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def bind(path):
    import vireo_amd._lib as L
    h = C.CDLL(path)
    bound = []
    for name, (res, args) in L.SIGNATURES.items():
        try:
            fn = getattr(h, name)
        except AttributeError:
            continue                    # device entry points: not in the host-only library
        fn.restype, fn.argtypes = res, args
        bound.append(name)
    L._lib = h                          # vireo_amd._lib.lib() now returns the sanitizer build
    return L, bound


# accepted here, rejected by scipy.io.mmread (a superset of the format is harmless): a comment
# line between entry lines (the format allows comments in the header only)
LENIENT = {"comment_between_entries"}


def mtx_fuzz(L, tmp):
    import numpy as np
    from scipy.io import mmread
    from vireo_amd import io_utils
    rng = np.random.default_rng(5)
    good = ["%%MatrixMarket matrix coordinate integer general", "% a comment", "7 5 6",
            "1 1 3", "7 5 1", "2 3 40", "2 3 2", "4 4 7", "6 2 100000"]
    cases = {
        "good": "\n".join(good) + "\n",
        "no_final_newline": "\n".join(good),
        "crlf": "\r\n".join(good) + "\r\n",
        "blank_lines": "\n".join(good[:4] + ["", "   "] + good[4:]) + "\n",
        "comment_between_entries": "\n".join(good[:4] + ["% mid comment"] + good[4:]) + "\n",
        "real_field": "\n".join([good[0].replace("integer", "real")] + good[1:3] +
                                [l + ".0" for l in good[3:]]) + "\n",
        "pattern_field": "\n".join([good[0].replace("integer", "pattern")] + good[1:3] +
                                   [" ".join(l.split()[:2]) for l in good[3:]]) + "\n",
        "truncated_mid_line": ("\n".join(good) + "\n")[:-7],
        "fewer_lines_than_declared": "\n".join(good[:-2]) + "\n",
        "more_lines_than_declared": "\n".join(good + ["3 3 3", "5 5 5"]) + "\n",
        "garbage_token": "\n".join(good[:5] + ["2 x 40"] + good[6:]) + "\n",
        "negative_index": "\n".join(good[:5] + ["-2 3 40"] + good[6:]) + "\n",
        "index_out_of_range": "\n".join(good[:5] + ["8 3 40"] + good[6:]) + "\n",
        "zero_index": "\n".join(good[:5] + ["0 3 40"] + good[6:]) + "\n",
        "huge_value": "\n".join(good[:5] + ["2 3 99999999999999999999999999999999"] + good[6:]) + "\n",
        "huge_index": "\n".join(good[:5] + ["99999999999999999999999 3 4"] + good[6:]) + "\n",
        "value_beyond_int32": "\n".join(good[:5] + ["2 3 4294967297"] + good[6:]) + "\n",
        "negative_value": "\n".join(good[:5] + ["2 3 -4"] + good[6:]) + "\n",
        "missing_value": "\n".join(good[:5] + ["2 3"] + good[6:]) + "\n",
        "huge_size_line": good[0] + "\n99999999999999999999 5 6\n1 1 1\n",
        "negative_size": good[0] + "\n-7 5 6\n",
        "size_line_only": good[0] + "\n7 5 0\n",
        "no_size_line": good[0] + "\n% only comments\n",
        "banner_only": good[0],
        "empty": "",
        "not_mtx": "hello\n1 2 3\n",
        "array_storage": "%%MatrixMarket matrix array integer general\n2 2\n1\n2\n3\n4\n",
        "symmetric": good[0].replace("general", "symmetric") + "\n3 3 1\n2 1 5\n",
        "binary_noise": good[0] + "\n7 5 6\n" + bytes(rng.integers(0, 256, 400, dtype=np.uint8)).decode("latin-1"),
        "long_line": good[0] + "\n7 5 1\n" + "1 " * 5000 + "\n",
    }
    n_ok = n_err = 0
    for name, text in cases.items():
        path = os.path.join(tmp, "fuzz_%s.mtx" % name)
        with open(path, "wb") as f:
            f.write(text.encode("latin-1"))
        rows, cols, nnz = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        rc = L.lib().vrx_mtx_header(path.encode(), C.byref(rows), C.byref(cols), C.byref(nnz))
        if rc != 0:
            n_err += 1
            assert L.lib().vrx_last_error(), name
            continue
        if nnz.value > 10 ** 6:           # (a size line may promise anything; the reader checks it)
            nnz_buf = 16
        else:
            nnz_buf = nnz.value
        r = np.full(max(nnz_buf, 1), -7, dtype=np.int32)
        c = np.full(max(nnz_buf, 1), -7, dtype=np.int32)
        v = np.full(max(nnz_buf, 1), -7, dtype=np.int32)
        i32 = C.POINTER(C.c_int32)
        for threads in (1, 3):
            rc = L.lib().vrx_mtx_read(path.encode(), nnz_buf, r.ctypes.data_as(i32), c.ctypes.data_as(i32),
                                      v.ctypes.data_as(i32), threads)
            if rc != 0:
                assert L.lib().vrx_last_error(), name
        if rc != 0:
            n_err += 1
            continue
        n_ok += 1
        try:
            want = mmread(path).tocoo()
        except Exception as e:          # noqa: BLE001
            if name in LENIENT:
                continue
            raise AssertionError("%s: the reader accepted what scipy.io.mmread rejects (%s)" % (name, e))
        assert (rows.value, cols.value) == want.shape, name
        got = io_utils.read_mtx(path)
        assert (got.tocsc() != want.tocsc()).nnz == 0, name
    # the parsed cases must include the well-formed ones, the rejected ones the malformed
    assert n_ok >= 5 and n_err >= 12, (n_ok, n_err)
    print("mtx fuzz: %d parsed like scipy, %d rejected with a status" % (n_ok, n_err))


def balance_tiles(lib_path):
    """vrx_balance_tile (the per-tile greedy of the balanced-slab build; a C++ symbol of the host unit, called
    by the device builder only): random tiles through both record widths (<= 2048 rows and counts of < 32
    words, else 32-bit records), the 16- and 32-bit score paths, blocks of at most 64 slabs and the whole range, empty
    columns, rows outside the tile.  posmap / perm must be inverse of each other, no slab over its capacity,
    and the row-slab loads flatter than contiguous slabs."""
    import subprocess
    import numpy as np
    names = [l.split()[-1] for l in subprocess.run(["nm", "-D", lib_path], capture_output=True, text=True,
                                                   check=True).stdout.splitlines() if "vrx_balance_tile" in l]
    assert len(names) == 1, names
    fn = getattr(C.CDLL(lib_path), names[0])
    fn.restype = None
    vp = C.c_void_p
    fn.argtypes = [vp, C.c_int64, vp, vp, vp, C.c_int64, C.c_int, C.c_int, C.c_int, vp, vp]
    rng = np.random.default_rng(9)
    n_cases = 0
    for (n_all, n_tile, NC, S, dens, wmax) in ((300, 200, 700, 64, 0.05, 3), (2500, 2300, 900, 64, 0.02, 3),
                                               (150, 150, 600, 32, 0.2, 60), (64, 48, 9000, 32, 0.05, 2),
                                               (40, 40, 100, 512, 0.3, 2), (500, 400, 1000, 16, 0.9, 200)):
        mask = rng.random((n_all, NC)) < dens
        mask[:, rng.integers(0, NC, NC // 10)] = False                    # columns without an entry
        ptr = np.zeros(n_all + 1, np.int64)
        np.cumsum(mask.sum(1), out=ptr[1:])
        idx = np.nonzero(mask)[1].astype(np.int32)
        words = rng.integers(0, wmax + 1, idx.size).astype(np.uint8)      # (0: an entry without words)
        rows = np.sort(rng.choice(n_all, n_tile, replace=False)).astype(np.int32)
        n_slab = -(-NC // S)
        posmap = np.full(NC, -1, np.int32)
        perm = np.full(n_slab * S, -1, np.int32)
        fn(rows.ctypes.data, n_tile, ptr.ctypes.data, idx.ctypes.data, words.ctypes.data, NC, n_slab, S,
           64 if n_cases % 2 == 0 else 0, posmap.ctypes.data, perm.ctypes.data)
        assert np.array_equal(np.sort(posmap), np.unique(posmap)) and posmap.min() >= 0 and posmap.max() < n_slab * S
        assert np.array_equal(perm[posmap], np.arange(NC))
        assert np.bincount(posmap // S, minlength=n_slab).max() <= S

        def spread(slab_of):
            load = np.zeros((n_tile, n_slab))
            for i, r in enumerate(rows):
                e = slice(ptr[r], ptr[r + 1])
                np.add.at(load[i], slab_of[idx[e]], words[e])
            return load.std()
        if wmax <= 3:       # (deep counts saturate the greedy's loads at 127: no promise there)
            assert spread(posmap // S) <= spread(np.arange(NC) // S) + 1e-9
        n_cases += 1
    print("balance tiles: %d cases" % n_cases)


def main(lib_path, tmp):
    L, bound = bind(lib_path)
    assert "vrx_mtx_read" in bound and "vrx_model_fit" not in bound
    mtx_fuzz(L, tmp)
    balance_tiles(lib_path)
    import pytest
    keep = ("merge_counts or generator_jump or legacy_stream_continues or fast_generator or "
            "float32_sum or native_text_writers or donor_genotype_codes")
    rc = pytest.main(["-x", "-q", "-p", "no:cacheprovider",
                      os.path.join(ROOT, "tests", "test_cli_io_cpu.py")])
    if rc != 0:
        return int(rc)
    rc = pytest.main(["-x", "-q", "-p", "no:cacheprovider", "-k", keep,
                      os.path.join(ROOT, "tests", "test_host_cpu.py")])
    return int(rc)


if __name__ == "__main__":
    sys.exit(main(sys.argv[1], sys.argv[2]))
