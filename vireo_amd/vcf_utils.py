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

import numpy as np
from scipy.sparse import csr_matrix

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


def parse_donor_GPb(GT_dat, tag='GT', min_prob=0.0):
    """(n_var, n_donor, 3) genotype probabilities from GT / GP / PL strings
    (vcf_utils.py:299-336); missing calls are uniform."""
    if tag not in ('GT', 'GP', 'PL'):
        print("[parse_donor_GPb] Error: no support tag: %s" % tag)
        return None
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


def GenoINFO_maker(GT_prob, AD_reads, DP_reads):
    """GT / AD / DP / PL strings of the estimated donor genotypes (vcf_utils.py:208-231).
    Floors GT_prob at 1e-10 in place, like the reference."""
    call = np.argmax(GT_prob, axis=2)
    GT_prob[GT_prob < 10 ** (-10)] = 10 ** (-10)
    PL = np.round(-10 * np.log10(GT_prob)).astype(int).astype(str)
    AD = np.round(AD_reads).astype(int).astype(str)
    DP = np.round(DP_reads).astype(int).astype(str)
    names = ['0/0', '1/0', '1/1']
    return {
        'GT': [[names[x] for x in call[i]] for i in range(GT_prob.shape[0])],
        'AD': [list(AD[i]) for i in range(GT_prob.shape[0])],
        'DP': [list(DP[i]) for i in range(GT_prob.shape[0])],
        'PL': [[",".join(x) for x in PL[i]] for i in range(GT_prob.shape[0])],
    }


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
    with open(plain, "w") as out:
        for line in VCF_dat['comments']:
            if not any(line.startswith("##FORMAT=<ID=" + t) for t in GenoTags):
                out.write(line + "\n")
        for t in GenoTags:
            if t in _FORMAT_HEADER:
                out.write(_FORMAT_HEADER[t] + "\n")
        out.write("#" + "\t".join(cols + list(VCF_dat['samples'])) + "\n")
        fmt = ":".join(GenoTags)
        for i in range(len(VCF_dat['variants'])):
            rec = [VCF_dat['FixedINFO'][c][i] for c in cols[:8]] + [fmt]
            for s in range(len(VCF_dat['samples'])):
                rec.append(":".join(VCF_dat['GenoINFO'][t][i][s] for t in GenoTags))
            out.write("\t".join(rec) + "\n")
    tool = "bgzip" if shutil.which("bgzip") is not None else "gzip"
    subprocess.run([tool, "-f", plain], stdout=subprocess.PIPE)
