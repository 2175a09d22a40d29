"""VCF text I/O used by the ``vireo`` command (host side, no GPU work).

Own implementation of what the command needs from vireoSNP/utils/vcf_utils.py: loading a
cellSNP / donor VCF (:80-159), its sparse per-cell AD/DP (:192-205), donor genotype
probabilities from GT / GP / PL tags (:299-336), SNP matching with or without the ``chr``
prefix (:339-350) and writing the estimated donor genotypes (:208-296).  Return
structures keep the reference's dict keys so downstream code reads the same.
"""
import gzip
import shutil
import subprocess

import ctypes as C

import numpy as np
from scipy.sparse import csr_matrix

from . import _lib
from .vireo_base import match

_MISSING = (".", "./.", ".|.")


def _open_text(path):
    if path.endswith(".gz") or path.endswith(".bgz"):
        return gzip.open(path, "rt")
    return open(path, "r")


def _sparse_geno(rows, formats, format_list):
    """cells x variants sparse layout of the FORMAT fields (vcf_utils.py:30-57): for every
    variant the cells with a call, one string per requested key."""
    out = {k: [] for k in format_list}
    indices, indptr = [], [0]
    tagged = np.zeros(len(format_list), np.int64)
    want = set(format_list)
    for fmt, calls in zip(formats, rows):
        if set(fmt) != want:
            print("Error: require the same format for all variants.")
            raise SystemExit
        pos = [fmt.index(k) for k in format_list]
        blank = ":".join(["."] * len(format_list))
        for cell, call in enumerate(calls):
            if call == blank or call == ".":
                continue
            parts = call.split(":")
            for k, p in zip(format_list, pos):
                out[k].append(parts[p])
            indices.append(cell)
            tagged += 1
        indptr.append(len(indices))
    out["indices"], out["indptr"] = indices, indptr
    out["shape"] = (len(rows[0]) if rows else 0, len(rows))
    return out, tagged


def _dense_geno(rows, formats, format_list):
    """variants x samples lists per key (vcf_utils.py:58-69); absent keys become '.'"""
    out = {k: [] for k in format_list}
    tagged = np.zeros(len(format_list), np.int64)
    for fmt, calls in zip(formats, rows):
        split = [c.split(":") for c in calls]
        for j, k in enumerate(format_list):
            if k in fmt:
                p = fmt.index(k)
                out[k].append([s[p] for s in split])
                tagged[j] += 1
            else:
                out[k].append(["."] * len(split))
    return out, tagged


def load_VCF(vcf_file, biallelic_only=False, load_sample=True, sparse=True,
             format_list=None):
    """-> dict(variants, FixedINFO, contigs, comments[, samples, GenoINFO, n_SNP_tagged])
    like the reference loader (vcf_utils.py:80-159)."""
    fixed, contigs, comments = {}, [], []
    keys, samples, variants, rows, formats = [], [], [], [], []
    with _open_text(vcf_file) as fh:
        for line in fh:
            line = line.rstrip()
            if line.startswith("#"):
                if line.startswith("##contig="):
                    contigs.append(line)
                if line.startswith("#CHROM"):
                    cols = line[1:].split("\t")
                    keys = cols[:8]
                    fixed = {k: [] for k in keys}
                    if load_sample:
                        samples = cols[9:]
                else:
                    comments.append(line)
                continue
            f = line.split("\t")
            if biallelic_only and (len(f[3]) > 1 or len(f[4]) > 1):
                continue
            for k, v in zip(keys, f):
                fixed[k].append(v)
            variants.append("_".join((f[0], f[1], f[3], f[4])))
            if load_sample:
                formats.append(f[8].split(":"))
                rows.append(f[9:])
    rv = dict(variants=variants, FixedINFO=fixed, contigs=contigs, comments=comments)
    if load_sample:
        rv["samples"] = samples
        if not rows:
            rv["GenoINFO"], rv["n_SNP_tagged"] = None, None
        else:
            use = format_list if format_list is not None else formats[0]
            geno, tagged = (_sparse_geno if sparse else _dense_geno)(rows, formats, use)
            low = np.where(tagged < 0.1 * len(rows))[0]
            if len(low) > 0:
                print('[vireo] Warning: too few variants with tags!',
                      '\t'.join(use[k] + ": " + str(tagged[k]) for k in range(len(use))))
            rv["GenoINFO"], rv["n_SNP_tagged"] = geno, tagged
    return rv


def read_sparse_GeneINFO(GenoINFO, keys=['AD', 'DP'], axes=[-1, -1]):
    """float64 CSR (variants x cells) per key from the sparse FORMAT layout
    (vcf_utils.py:192-205)."""
    n_cell, n_var = (int(x) for x in GenoINFO['shape'])
    indptr = np.asarray(GenoINFO['indptr'], dtype=int)
    indices = np.asarray(GenoINFO['indices'], dtype=int)
    out = {}
    for key, ax in zip(keys, axes):
        vals = [x.split(",")[ax] for x in GenoINFO[key]]
        data = np.array([v if v != '.' else '0' for v in vals]).astype(float)
        out[key] = csr_matrix((data, indices, indptr), shape=(n_var, n_cell))
    return out


def _parse_codes_vectorised(GT_dat, tag):
    """All genotype codes of a donor VCF at once (the reference converts them one string at a time,
    vcf_utils.py:311-331: seconds per 10^5 variants): the same arithmetic on whole arrays.
    None when the input is not a regular table of well-formed codes."""
    n_var = len(GT_dat)
    n_don = len(GT_dat[0]) if n_var else 0
    if any(len(row) != n_don for row in GT_dat):
        return None
    flat = [code for row in GT_dat for code in row]
    missing = np.fromiter((code in _MISSING for code in flat), dtype=bool, count=len(flat))
    good = [code for code, m in zip(flat, missing) if not m]
    P = np.zeros((len(flat), 3))
    P[missing] = 1 / 3
    try:
        if tag == 'GT':
            if any(len(code) < 1 for code in good):
                return None
            first = np.array([code[0] for code in good], dtype=float)
            last = np.array([code[-1] for code in good], dtype=float)
            idx = (first + last).astype(int)
            if good and (idx.min() < 0 or idx.max() > 2):
                return None
            P[np.flatnonzero(~missing), idx] = 1
        else:
            if any(code.count(",") != 2 for code in good):
                return None
            vals = np.array(",".join(good).split(","), dtype=float).reshape(-1, 3) if good \
                else np.zeros((0, 3))
            if tag == 'PL':
                vals = 10 ** (-0.1 * (vals - vals.min(axis=1, keepdims=True)) - 0.025)
            P[~missing] = vals
    except ValueError:
        return None
    return P.reshape(n_var, n_don, 3)


def parse_donor_GPb(GT_dat, tag='GT', min_prob=0.0):
    """(n_var, n_donor, 3) genotype probabilities from GT / GP / PL strings
    (vcf_utils.py:299-336); missing calls are uniform."""
    if tag not in ('GT', 'GP', 'PL'):
        print("[parse_donor_GPb] Error: no support tag: %s" % tag)
        return None
    P = _parse_codes_vectorised(GT_dat, tag)
    if P is None:           # ragged rows or fields that are not three numbers: one code at a time
        P = np.zeros((len(GT_dat), len(GT_dat[0]), 3))
        for i, row in enumerate(GT_dat):
            for j, code in enumerate(row):
                if code in _MISSING:
                    P[i, j] = 1 / 3
                elif tag == 'GT':
                    P[i, j, int(float(code[0]) + float(code[-1]))] = 1
                elif tag == 'GP':
                    P[i, j] = np.array(code.split(','), float)
                else:
                    phred = np.array(code.split(','), float)
                    P[i, j] = 10 ** (-0.1 * (phred - min(phred)) - 0.025)
    P += min_prob
    P /= P.sum(axis=2, keepdims=True)
    return P


def match_SNPs(SNP_ids1, SNPs_ids2):
    """match() with a retry adding the 'chr' prefix on either side (vcf_utils.py:339-350)."""
    idx = match(SNP_ids1, SNPs_ids2)
    if np.mean(idx == None) == 1:                                  # noqa: E711
        idx = match(["chr" + x for x in SNP_ids1], SNPs_ids2)
    if np.mean(idx == None) == 1:                                  # noqa: E711
        idx = match(SNP_ids1, ["chr" + x for x in SNPs_ids2])
    return idx


class _GenoArrays(dict):
    """What ``GenoINFO_maker`` returns: the reference's dict of per-variant string lists
    (vcf_utils.py:208-231), formed only when somebody reads a tag; ``write_VCF`` takes the integer
    arrays behind it and lets the library format the records."""
    _NAMES = ('0/0', '1/0', '1/1')

    def __init__(self, call, AD, DP, PL):
        super().__init__()
        self.call, self.AD, self.DP, self.PL = call, AD, DP, PL

    def _strings(self, tag):
        n = self.call.shape[0]
        if tag == 'GT':
            return [[self._NAMES[x] for x in self.call[i]] for i in range(n)]
        if tag == 'PL':
            txt = self.PL.astype(str)
            return [[",".join(x) for x in txt[i]] for i in range(n)]
        txt = getattr(self, tag).astype(str)
        return [list(txt[i]) for i in range(n)]

    def __missing__(self, tag):
        if tag not in ('GT', 'AD', 'DP', 'PL'):
            raise KeyError(tag)
        self[tag] = self._strings(tag)
        return dict.__getitem__(self, tag)

    def __contains__(self, tag):
        return tag in ('GT', 'AD', 'DP', 'PL')

    def keys(self):
        return ['GT', 'AD', 'DP', 'PL']

    def __iter__(self):
        return iter(self.keys())

    def __len__(self):
        return 4

    def items(self):
        return [(k, self[k]) for k in self.keys()]


def _labels_blob(labels):
    """(concatenated utf-8 labels, int64 offsets) for the native writers"""
    enc = [str(x).encode() for x in labels]
    off = np.zeros(len(enc) + 1, dtype=np.int64)
    np.cumsum([len(b) for b in enc], out=off[1:])
    return b"".join(enc), off


def GenoINFO_maker(GT_prob, AD_reads, DP_reads):
    """GT / AD / DP / PL of the estimated donor genotypes (vcf_utils.py:208-231): the called
    genotype, the rounded read counts and PL = round(-10 log10 GT_prob).  Floors GT_prob at
    1e-10 in place, like the reference."""
    call = np.argmax(GT_prob, axis=2)
    GT_prob[GT_prob < 10 ** (-10)] = 10 ** (-10)
    PL = np.round(-10 * np.log10(GT_prob)).astype(int)
    return _GenoArrays(call, np.round(AD_reads).astype(int), np.round(DP_reads).astype(int), PL)


_FORMAT_HEADER = {
    "GT": '##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">',
    "AD": '##FORMAT=<ID=AD,Number=1,Type=Integer,Description="Read depth for each allele">',
    "DP": '##FORMAT=<ID=DP,Number=1,Type=Integer,Description="Read Depth">',
    "PL": '##FORMAT=<ID=PL,Number=G,Type=Integer,Description="Phred-scaled genotype likelihoods">',
}


def write_VCF(out_file, VCF_dat, GenoTags=['GT', 'AD', 'DP', 'PL']):
    """VCF of the donor genotypes, (b)gzipped when the name ends in .gz
    (vcf_utils.py:234-296)."""
    plain = out_file.split(".gz")[0] if out_file.endswith(".gz") else out_file
    if "samples" not in VCF_dat:
        VCF_dat["samples"] = []
        if GenoTags != []:
            print("No sample available: GenoTags will be ignored.")
    cols = ["CHROM", "POS", "ID", "REF", "ALT", "QUAL", "FILTER", "INFO", "FORMAT"]
    head = [line for line in VCF_dat['comments']
            if not any(line.startswith("##FORMAT=<ID=" + t) for t in GenoTags)]
    head += [_FORMAT_HEADER[t] for t in GenoTags if t in _FORMAT_HEADER]
    head.append("#" + "\t".join(cols + list(VCF_dat['samples'])))
    geno = VCF_dat.get('GenoINFO')
    if (isinstance(geno, _GenoArrays) and list(GenoTags) == ['GT', 'AD', 'DP', 'PL']
            and geno.PL.shape[2:] == (3,) and geno.call.shape[1] == len(VCF_dat['samples'])
            and len({len(VCF_dat['variants']), geno.call.shape[0], geno.AD.shape[0],
                     geno.DP.shape[0], geno.PL.shape[0]}) == 1
            and geno.AD.shape[1] == geno.DP.shape[1] == geno.PL.shape[1] == geno.call.shape[1]):
        # the records are formatted (and gzipped) by the library from the integer arrays
        fixed = VCF_dat['FixedINFO']
        n = len(VCF_dat['variants'])
        prefix = ["\t".join([fixed[c][i] for c in cols[:8]] + ["GT:AD:DP:PL"]) for i in range(n)]
        blob, off = _labels_blob(prefix)
        i64 = lambda a: np.ascontiguousarray(a, dtype=np.int64)     # noqa: E731
        call = np.ascontiguousarray(geno.call, dtype=np.int8)
        ad, dp, pl = i64(geno.AD), i64(geno.DP), i64(geno.PL)
        void = lambda a: a.ctypes.data_as(C.c_void_p)               # noqa: E731
        # (like the generic path below: the result is always <plain name>.gz)
        _lib.check(_lib.lib().vrx_write_vcf_records(
            (plain + ".gz").encode(), ("\n".join(head) + "\n").encode(), blob, void(off),
            void(call), void(ad), void(dp), void(pl), n, call.shape[1], 1))
        return
    with open(plain, "w") as out:
        for line in head:
            out.write(line + "\n")
        fmt = ":".join(GenoTags)
        for i in range(len(VCF_dat['variants'])):
            rec = [VCF_dat['FixedINFO'][c][i] for c in cols[:8]] + [fmt]
            for s in range(len(VCF_dat['samples'])):
                rec.append(":".join(VCF_dat['GenoINFO'][t][i][s] for t in GenoTags))
            out.write("\t".join(rec) + "\n")
    tool = "bgzip" if shutil.which("bgzip") is not None else "gzip"
    subprocess.run([tool, "-f", plain], stdout=subprocess.PIPE)
