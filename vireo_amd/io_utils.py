"""Cell-data loaders and result writers of the ``vireo`` command (host side).

Own implementation of vireoSNP/utils/io_utils.py: read_cellSNP (:42-59), read_vartrix
(:62-88), match_donor_VCF (:10-39) and write_donor_id (:91-170: donor_ids.tsv,
summary.tsv, prob_singlet/doublet.tsv.gz, _log.txt).  Output text is formatted exactly like
the reference's so that downstream tools (and users' grep) keep working.
"""
import gzip
import shutil
from itertools import combinations

import ctypes as C

import numpy as np
from scipy.sparse import coo_matrix, csc_matrix

from . import _lib
from .vcf_utils import _labels_blob, load_VCF, match_SNPs


def _read_mtx_arrays(path, n_threads=0):
    """-> ((n_rows, n_cols), row, col, val) int32 COO arrays in file order from the library's
    multi-threaded parser (vrx_mtx_read), or None for what it does not read (gzipped files,
    symmetric / complex / array storage)"""
    path = str(path)
    L = _lib.lib()
    n_rows, n_cols, nnz = C.c_int64(0), C.c_int64(0), C.c_int64(0)
    if path.endswith(".gz") or L.vrx_mtx_header(path.encode(), C.byref(n_rows), C.byref(n_cols),
                                                C.byref(nnz)) != 0:
        return None
    row = np.empty(nnz.value, dtype=np.int32)
    col = np.empty(nnz.value, dtype=np.int32)
    val = np.empty(nnz.value, dtype=np.int32)
    i32 = C.POINTER(C.c_int32)
    _lib.check(L.vrx_mtx_read(path.encode(), nnz.value, row.ctypes.data_as(i32),
                              col.ctypes.data_as(i32), val.ctypes.data_as(i32), int(n_threads)))
    return (n_rows.value, n_cols.value), row, col, val


def read_mtx(path, n_threads=0):
    """MatrixMarket coordinate file -> scipy COO matrix (int64 counts), parsed by the library's
    multi-threaded reader (vrx_mtx_read) instead of scipy.io.mmread (io_utils.py:57): the same
    entries in file order, duplicates kept (``.tocsc()`` sums them, like after mmread).
    Gzipped files and symmetric / complex / array storage fall back to scipy."""
    got = _read_mtx_arrays(path, n_threads)
    if got is None:
        from scipy.io import mmread
        return mmread(str(path))
    shape, row, col, val = got
    return coo_matrix((val.astype(np.int64), (row, col)), shape=shape)


def read_mtx_csc(path, n_threads=0):
    """``mmread(path).tocsc()`` (io_utils.py:57) without SciPy's single-threaded COO -> CSC
    conversion: the library parses the file (vrx_mtx_read) and sorts the entries into columns on
    all cores (vrx_coo_to_csc).  A file whose columns do not come out strictly increasing
    (duplicate entries, or not variant-major) goes through SciPy's conversion, which sorts and sums
    duplicates like the reference's."""
    got = _read_mtx_arrays(path, n_threads)
    if got is None:
        from scipy.io import mmread
        return mmread(str(path)).tocsc()
    shape, row, col, val = got
    nnz = row.size
    indptr = np.empty(shape[1] + 1, dtype=np.int64)
    indices = np.empty(nnz, dtype=np.int32)
    data = np.empty(nnz, dtype=np.int64)
    canonical = C.c_int32(0)
    i32, i64 = C.POINTER(C.c_int32), C.POINTER(C.c_int64)
    _lib.check(_lib.lib().vrx_coo_to_csc(shape[0], shape[1], nnz, row.ctypes.data_as(i32),
                                         col.ctypes.data_as(i32), val.ctypes.data_as(i32),
                                         indptr.ctypes.data_as(i64), indices.ctypes.data_as(i32),
                                         data.ctypes.data_as(i64), C.byref(canonical), int(n_threads)))
    if not canonical.value:
        return coo_matrix((val.astype(np.int64), (row, col)), shape=shape).tocsc()
    out = csc_matrix((data, indices, indptr), shape=shape)
    out.has_canonical_format = True                           # (checked column by column in the library)
    return out


def read_cellSNP(dir_name, layers=['AD', 'DP']):
    """cellSNP output folder -> dict with AD, DP (CSC), samples, variants, ...
    (io_utils.py:42-59)."""
    dat = load_VCF(dir_name + "/cellSNP.base.vcf.gz", load_sample=False, biallelic_only=False)
    for layer in layers:
        dat[layer] = read_mtx_csc(dir_name + "/cellSNP.tag.%s.mtx" % layer)
    dat['samples'] = np.genfromtxt(dir_name + "/cellSNP.samples.tsv", dtype=str)
    return dat


def read_vartrix(alt_mtx, ref_mtx, cell_file, vcf_file=None):
    """VarTrix alt/ref matrices -> AD, DP = ref + alt (io_utils.py:62-88)."""
    if vcf_file is not None:
        dat = load_VCF(vcf_file, load_sample=False, biallelic_only=False)
        dat['variants'] = np.array(dat['variants'])
    else:
        dat = {}
    dat['AD'] = read_mtx_csc(alt_mtx)
    dat['DP'] = read_mtx_csc(ref_mtx) + dat['AD']
    dat['samples'] = np.genfromtxt(cell_file, dtype=str)
    return dat


def match_donor_VCF(cell_dat, donor_vcf):
    """keep the variants present in both the cell data and the donor VCF, in cell order
    (io_utils.py:10-39)."""
    mm = match_SNPs(cell_dat['variants'], donor_vcf['variants'])
    keep = np.where(mm != None)[0]                                   # noqa: E711
    if len(keep) == 0:
        print("[vireo] warning: no variants matched to donor VCF, please check chr format!")
    else:
        print("[vireo] %d out %d variants matched to donor VCF"
              % (len(keep), len(cell_dat['variants'])))
    other = mm[keep].astype(int)
    cell_dat['AD'] = cell_dat['AD'][keep, :]
    cell_dat['DP'] = cell_dat['DP'][keep, :]
    cell_dat["variants"] = [cell_dat["variants"][x] for x in keep]
    for k in cell_dat["FixedINFO"]:
        cell_dat["FixedINFO"][k] = [cell_dat["FixedINFO"][k][x] for x in keep]
    donor_vcf["variants"] = [donor_vcf["variants"][x] for x in other]
    for k in donor_vcf["FixedINFO"]:
        donor_vcf["FixedINFO"][k] = [donor_vcf["FixedINFO"][k][x] for x in other]
    for k in donor_vcf["GenoINFO"]:
        donor_vcf["GenoINFO"][k] = [donor_vcf["GenoINFO"][k][x] for x in other]
    return cell_dat, donor_vcf


def _gzip_file(path):
    with open(path, "rb") as src, gzip.open(path + ".gz", "wb") as dst:
        shutil.copyfileobj(src, dst)
    import os
    os.remove(path)


def _write_table_gz(path, header, row_names, table):
    """<header>, then <row name> TAB "%.2e" ... per row, gzipped (the reference writes the text
    with a Python loop and runs ``gzip -f``, io_utils.py:147-170); formatted and deflated on
    several threads by libvireo_hip.so"""
    table = np.ascontiguousarray(table, dtype=np.float64)
    blob, off = _labels_blob(row_names)
    _lib.check(_lib.lib().vrx_write_table(
        path.encode(), ("\t".join(header) + "\n").encode(), blob,
        off.ctypes.data_as(C.c_void_p), _lib.dptr(table), table.shape[0], table.shape[1],
        b"%.2e", 1))


def write_donor_id(out_dir, donor_names, cell_names, n_vars, res_vireo):
    """donor_ids.tsv, summary.tsv, prob_singlet.tsv.gz, prob_doublet.tsv.gz, _log.txt with the
    reference's thresholds (prob_max < 0.9 -> unassigned, doublet >= 0.9, n_vars < 10 ->
    unassigned) and number formats (io_utils.py:91-170)."""
    ID_prob, doublet_prob = res_vireo['ID_prob'], res_vireo['doublet_prob']
    prob_max = np.max(ID_prob, axis=1)
    prob_dbl = np.max(doublet_prob, axis=1)
    best_singlet = np.array(donor_names, "U100")[np.argmax(ID_prob, axis=1)]
    pair_names = [",".join(x) for x in combinations(donor_names, 2)]
    best_doublet = np.array(pair_names, "U100")[np.argmax(doublet_prob, axis=1)]
    donor_ids = best_singlet.copy()
    donor_ids[prob_max < 0.9] = "unassigned"
    donor_ids[prob_dbl >= 0.9] = "doublet"
    donor_ids[n_vars < 10] = "unassigned"

    with open(out_dir + "/_log.txt", "w") as f:
        f.write("logLik: %.3e\n" % (res_vireo['LB_doublet']))
        f.write("thetas: \n%s\n" % (res_vireo['theta_shapes']))

    uniq, count = np.unique(donor_ids, return_counts=True)
    with open(out_dir + "/summary.tsv", "w") as f:
        f.write("Var1\tFreq\n")
        for u, c in zip(uniq, count):
            f.write("%s\t%d\n" % (u, c))
    print("[vireo] final donor size:")
    print("\t".join(str(x) for x in uniq))
    print("\t".join(str(x) for x in count))

    with open(out_dir + "/donor_ids.tsv", "w") as f:
        f.write("\t".join(["cell", "donor_id", "prob_max", "prob_doublet", "n_vars",
                           "best_singlet", "best_doublet", "doublet_logLikRatio"]) + "\n")
        for i in range(len(cell_names)):
            f.write("\t".join([cell_names[i], donor_ids[i], "%.2e" % prob_max[i],
                               "%.2e" % prob_dbl[i], "%d" % n_vars[i], best_singlet[i],
                               best_doublet[i], "%.3f" % res_vireo['doublet_LLR'][i]]) + "\n")

    for name, header, table in (("prob_singlet.tsv", donor_names, ID_prob),
                                ("prob_doublet.tsv", pair_names, doublet_prob)):
        _write_table_gz(out_dir + "/" + name + ".gz", ["cell"] + list(header), cell_names, table)
