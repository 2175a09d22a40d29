"""CPU tests of the command-line I/O layer (no GPU compute): the loaders reproduce the
reference loaders' outputs (fixtures captured in the build container), the option surface
is the reference's, and the writers' text layout matches a reference output file."""
import contextlib
import io
import os

import numpy as np

from tests import gold

HERE = os.path.dirname(os.path.abspath(__file__))
DATA = os.path.join(HERE, "golden", "data")


def _quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def test_cell_vcf_and_folder_load_the_same_matrices():
    from vireo_amd import vireo as V
    AD, DP = gold.c1()
    for src in (DATA + "/cells.cellSNP.vcf.gz", DATA + "/cellSNP_mat"):
        opts, _ = V.build_parser().parse_args(["-c", src])
        d = _quiet(V.load_cells, opts)
        assert (d["AD"].tocsc() != AD).nnz == 0 and (d["DP"].tocsc() != DP).nnz == 0
        assert len(d["samples"]) == 952 and len(d["variants"]) == 3784
        assert d["variants"][0] == "1_1065797_G_C"


def test_donor_vcf_genotype_probabilities():
    from vireo_amd import vcf_utils, io_utils, vireo as V
    g = gold.load("cli_inputs")
    opts, _ = V.build_parser().parse_args(["-c", DATA + "/cells.cellSNP.vcf.gz"])
    cells = _quiet(V.load_cells, opts)
    donors = vcf_utils.load_VCF(DATA + "/donors.cellSNP.vcf.gz", biallelic_only=True,
                                sparse=False, format_list=["PL"])
    assert donors["samples"] == ["MantonCB1", "MantonCB2", "MantonCB3", "MantonCB4"]
    cells, donors = _quiet(io_utils.match_donor_VCF, cells, donors)
    assert len(cells["variants"]) == int(g["n_matched"]) == cells["AD"].shape[0]
    GPb = vcf_utils.parse_donor_GPb(donors["GenoINFO"]["PL"], "PL")
    assert np.array_equal(GPb, g["donor_GPb"])            # bit-identical to the reference's
    gt = vcf_utils.parse_donor_GPb([["0/1", "1|1", "./.", "0/0"]], "GT")
    assert np.array_equal(gt[0], [[0, 1, 0], [0, 0, 1], [1 / 3] * 3, [1, 0, 0]])


def test_option_surface_matches_reference():
    from vireo_amd import vireo as V
    p = V.build_parser()
    flags = {o for opt in p._get_all_options() for o in opt._long_opts + opt._short_opts}
    for f in ["--cellData", "-c", "--nDonor", "-N", "--outDir", "-o", "--vartrixData",
              "--donorFile", "-d", "--genoTag", "-t", "--noDoublet", "--nInit", "-M",
              "--extraDonor", "--extraDonorMode", "--forceLearnGT", "--ASEmode", "--noPlot",
              "--randSeed", "--cellRange", "--callAmbientRNAs", "--nproc", "-p"]:
        assert f in flags, f
    o, _ = p.parse_args([])
    assert (o.n_init, o.geno_tag, o.n_extra_donor, o.extra_donor_mode, o.nproc) == \
        (50, "PL", 0, "distance", 1)


def test_writers_layout(tmp_path):
    """write_donor_id on a golden vireo_wrap result: header, thresholds and number formats
    as in the reference's donor_ids.tsv / summary.tsv."""
    from vireo_amd import io_utils
    g = gold.load("c1_wrap_seed2_init4")
    AD, DP = gold.c1()
    n_vars = np.asarray((DP > 0).sum(axis=0)).ravel()
    names = ["donor%d" % i for i in range(4)]
    cells = ["cell%d" % i for i in range(AD.shape[1])]
    res = dict(ID_prob=g["ID_prob"], doublet_prob=g["doublet_prob"], doublet_LLR=g["doublet_LLR"],
               LB_doublet=float(g["LB_doublet"]), theta_shapes=g["theta_shapes"])
    _quiet(io_utils.write_donor_id, str(tmp_path), names, cells, n_vars, res)
    rows = [l.rstrip("\n").split("\t") for l in open(tmp_path / "donor_ids.tsv")]
    assert rows[0] == ["cell", "donor_id", "prob_max", "prob_doublet", "n_vars", "best_singlet",
                       "best_doublet", "doublet_logLikRatio"]
    assert len(rows) == 953 and rows[1][2] == "%.2e" % g["ID_prob"][0].max()
    labels = {r[1] for r in rows[1:]}
    assert labels <= set(names) | {"doublet", "unassigned"}
    summ = open(tmp_path / "summary.tsv").read().split("\n")
    assert summ[0] == "Var1\tFreq" and sum(int(l.split("\t")[1]) for l in summ[1:] if l) == 952
    assert open(tmp_path / "_log.txt").read().startswith("logLik: %.3e\n" % g["LB_doublet"])
    assert os.path.exists(tmp_path / "prob_singlet.tsv.gz")


def test_mtx_writer_round_trip(tmp_path, monkeypatch):
    """vrx_mtx_write (bench.py's end-to-end leg writes 1e8-entry cellSNP matrices with it) against
    the library's own parser and against scipy.io.mmread: same shape, same entries in file order;
    and the synthetic cellSNP folder loads back to the workload it was written from."""
    import ctypes as C
    from scipy.io import mmread
    from vireo_amd import _lib, io_utils, synth
    monkeypatch.setenv("VIREO_WRITER_CHUNK_BYTES", "4096")      # many chunks, several threads
    rng = np.random.default_rng(0)
    n = 20000
    row = rng.integers(0, 700, n).astype(np.int32)
    col = rng.integers(0, 90, n).astype(np.int32)
    val = rng.integers(1, 5000, n).astype(np.int32)
    path = str(tmp_path / "m.mtx")
    i32 = C.POINTER(C.c_int32)
    _lib.check(_lib.lib().vrx_mtx_write(path.encode(), 700, 90, n, row.ctypes.data_as(i32),
                                        col.ctypes.data_as(i32), val.ctypes.data_as(i32)))
    got = io_utils.read_mtx(path)
    assert got.shape == (700, 90) and np.array_equal(got.row, row) and np.array_equal(got.col, col)
    assert np.array_equal(got.data, val)
    ref = mmread(path)
    assert (ref.tocsc() != got.tocsc()).nnz == 0
    w = synth.donor_workload(300, 200, 3, 0.05, seed=1)
    synth.write_cellsnp_folder(w, str(tmp_path / "cells"))
    dat = io_utils.read_cellSNP(str(tmp_path / "cells"))
    AD, DP = synth.as_scipy(w)
    assert (dat["AD"] != AD).nnz == 0 and (dat["DP"] != DP).nnz == 0
    assert len(dat["samples"]) == 200 and len(dat["variants"]) == 300


def test_coo_to_csc_equals_scipys_conversion(tmp_path):
    """read_mtx_csc (vrx_mtx_read + vrx_coo_to_csc: a threaded counting sort by column) against
    ``mmread(...).tocsc()`` (io_utils.py:57): a variant-major file (the canonical path), a shuffled
    file and one with duplicate entries (both fall back to SciPy's sort / sum), an empty matrix."""
    from scipy.io import mmread, mmwrite
    from scipy.sparse import coo_matrix
    from vireo_amd import io_utils
    rng = np.random.default_rng(3)
    dense = (rng.random((400, 230)) < 0.1) * rng.integers(1, 90, (400, 230))
    dense[:, 17] = 0                                          # an empty column
    M0 = coo_matrix(dense)                                    # row-major entry order
    cases = {"rowmajor": M0}
    perm = rng.permutation(M0.nnz)
    cases["shuffled"] = coo_matrix((M0.data[perm], (M0.row[perm], M0.col[perm])), shape=M0.shape)
    cases["duplicates"] = coo_matrix((np.r_[M0.data, M0.data[:50]], (np.r_[M0.row, M0.row[:50]],
                                                                    np.r_[M0.col, M0.col[:50]])), shape=M0.shape)
    cases["empty"] = coo_matrix((5, 7), dtype=np.int64)
    for name, X in cases.items():
        path = str(tmp_path / (name + ".mtx"))
        with open(path, "w") as f:                            # (mmwrite would sum the duplicates)
            f.write("%%MatrixMarket matrix coordinate integer general\n%d %d %d\n" % (X.shape + (X.nnz,)))
            for r, c, v in zip(X.row, X.col, X.data):
                f.write("%d %d %d\n" % (r + 1, c + 1, v))
        got = io_utils.read_mtx_csc(path)
        want = mmread(path).tocsc()
        assert got.shape == want.shape and got.format == "csc" and got.data.dtype == want.data.dtype
        assert (got != want).nnz == 0 and got.nnz == want.nnz, name
        assert got.has_canonical_format
        assert np.array_equal(got.indptr, want.indptr) and np.array_equal(got.indices, want.indices)
        assert np.array_equal(got.data, want.data)
