"""Run the REFERENCE ``vireo`` command (build container only, /root/reference) on its bundled
demo data -- the five modes of examples/demo.sh with --randSeed 2 -- and keep the text
outputs as fixtures under tests/golden/cli/.  The input files under tests/golden/data/ are
copies of the reference's own demo data files (data, not source)."""
import gzip
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
DATA = os.path.join(HERE, "data")
OUT = os.path.join(HERE, "cli")
MODES = {
    "mode1_noGT": ["-c", DATA + "/cellSNP_mat", "-N", "4"],
    "mode2_PL": ["-c", DATA + "/cells.cellSNP.vcf.gz", "-d", DATA + "/donors.cellSNP.vcf.gz", "-N", "4"],
    "mode3_part": ["-c", DATA + "/cells.cellSNP.vcf.gz", "-d", DATA + "/donors.two.cellSNP.vcf.gz", "-N", "4"],
    "mode4_learn": ["-c", DATA + "/cells.cellSNP.vcf.gz", "-d", DATA + "/donors.cellSNP.vcf.gz", "-N", "4",
                    "--forceLearnGT"],
    "mode5_PL3": ["-c", DATA + "/cells.cellSNP.vcf.gz", "-d", DATA + "/donors.cellSNP.vcf.gz", "-N", "3"],
    "mode1_M1_noDoublet": ["-c", DATA + "/cellSNP_mat", "-N", "4", "-M", "1", "--noDoublet"],
    # (r3) the remaining inputs / flags of vireo.py:36-84
    "vartrix": ["--vartrixData", ",".join([DATA + "/vartrix/alt.mtx", DATA + "/vartrix/ref.mtx",
                                           DATA + "/vartrix/barcodes.tsv", DATA + "/cellSNP_mat/cellSNP.base.vcf.gz"]),
                "-N", "4", "-M", "2"],
    "cellRange": ["-c", DATA + "/cellSNP_mat", "-N", "4", "-M", "2", "--cellRange", "100-600"],
    "extraDonor": ["-c", DATA + "/cellSNP_mat", "-N", "3", "-M", "4", "--extraDonor", "1"],
    "extraDonor_size": ["-c", DATA + "/cellSNP_mat", "-N", "3", "-M", "4", "--extraDonor", "1",
                        "--extraDonorMode", "size"],
    "ASEmode": ["-c", DATA + "/cellSNP_mat", "-N", "4", "-M", "2", "--ASEmode"],
}
KEEP = ["donor_ids.tsv", "summary.tsv", "_log.txt"]
KEEP_GZ = ["prob_singlet.tsv.gz", "prob_doublet.tsv.gz"]     # kept as gzip -n of the decompressed text

# VarTrix-style inputs (io_utils.py:62-88: alt.mtx, ref.mtx, barcodes.tsv) derived from the
# reference's own demo matrices: alt = AD, ref = DP - AD
if not os.path.exists(DATA + "/vartrix/ref.mtx"):
    from scipy.io import mmread, mmwrite
    os.makedirs(DATA + "/vartrix", exist_ok=True)
    ad = mmread(DATA + "/cellSNP_mat/cellSNP.tag.AD.mtx").tocsc()
    dp = mmread(DATA + "/cellSNP_mat/cellSNP.tag.DP.mtx").tocsc()
    ref = (dp - ad).tocsc()
    ref.eliminate_zeros()
    mmwrite(DATA + "/vartrix/alt.mtx", ad.astype(int), field="integer")
    mmwrite(DATA + "/vartrix/ref.mtx", ref.astype(int), field="integer")
    shutil.copy(DATA + "/cellSNP_mat/cellSNP.samples.tsv", DATA + "/vartrix/barcodes.tsv")

for name, args in MODES.items():
    tmp = "/tmp/vireo_cli_gold_" + name
    shutil.rmtree(tmp, ignore_errors=True)
    env = dict(os.environ, PYTHONPATH="/root/reference", MPLBACKEND="Agg")
    subprocess.run([sys.executable, "-m", "vireoSNP.vireo"] + args +
                   ["-o", tmp, "--randSeed", "2", "--noPlot"], env=env, check=True,
                   stdout=subprocess.DEVNULL)
    dst = os.path.join(OUT, name)
    os.makedirs(dst, exist_ok=True)
    for f in KEEP:
        shutil.copy(os.path.join(tmp, f), os.path.join(dst, f))
    for f in KEEP_GZ:                  # the reference's own bytes of the probability tables
        with gzip.open(os.path.join(tmp, f), "rt") as src, open(os.path.join(dst, f[:-3]), "w") as d:
            d.write(src.read())
        subprocess.run(["gzip", "-nf", os.path.join(dst, f[:-3])], check=True)
    vcf = os.path.join(tmp, "GT_donors.vireo.vcf.gz")
    if os.path.exists(vcf):            # keep decompressed text re-gzipped deterministically
        with gzip.open(vcf, "rt") as src, open(os.path.join(dst, "GT_donors.vireo.vcf"), "w") as d:
            d.write(src.read())
        subprocess.run(["gzip", "-nf", os.path.join(dst, "GT_donors.vireo.vcf")], check=True)
    print(name, sorted(os.listdir(dst)))

# ---- cli_inputs.npz: what the REFERENCE's loaders make of the donor VCF (mode 2 of demo.sh) ----
# the matched variant count and the genotype-probability tensor that parse_donor_GPb derives from
# the PL tags; tests/test_cli_io_cpu.py holds vireo_amd's own loaders to them bit for bit
if True:
    import io
    import contextlib
    import numpy as np
    sys.path.insert(0, "/root/reference")
    from vireoSNP.utils.io_utils import match_donor_VCF
    from vireoSNP.utils.vcf_utils import load_VCF, parse_donor_GPb, read_sparse_GeneINFO
    with contextlib.redirect_stdout(io.StringIO()):
        cell_vcf = load_VCF(DATA + "/cells.cellSNP.vcf.gz", biallelic_only=True)
        cell_dat = read_sparse_GeneINFO(cell_vcf['GenoINFO'], keys=['AD', 'DP'])
        for _key in ['samples', 'variants', 'FixedINFO', 'contigs', 'comments']:
            cell_dat[_key] = cell_vcf[_key]
        donor_vcf = load_VCF(DATA + "/donors.cellSNP.vcf.gz", biallelic_only=True, sparse=False,
                             format_list=["PL"])
        cell_dat, donor_vcf = match_donor_VCF(cell_dat, donor_vcf)
        GPb = parse_donor_GPb(donor_vcf['GenoINFO']["PL"], "PL")
    np.savez_compressed(os.path.join(HERE, "cli_inputs.npz"), donor_GPb=GPb,
                        n_matched=np.int64(len(cell_dat['variants'])))
