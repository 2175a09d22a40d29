"""``vireo`` command: donor deconvolution for multiplexed scRNA-seq, MI355X edition.

Same options and output files as the reference command (vireoSNP/vireo.py:30-247,
registered as ``vireo`` in setup.py:53-55):

    python -m vireo_amd.vireo -c CELL_DATA -N n_donor -o OUT_DIR [-d DONOR_VCF] ...

The model fits run on the GPU (vireo_amd.vireo_wrap); loading and writing text files is
host work.  Not carried over: the genotype-distance figure (``--noPlot`` is accepted and
is the only behaviour) and ``--callAmbientRNAs`` (experimental upstream, vireo.py:79-81).
"""
import os
import sys
import time
from optparse import OptionParser, OptionGroup

import numpy as np

from . import __version__
from .counts import device_counts
from .io_utils import match_donor_VCF, read_cellSNP, read_vartrix, write_donor_id
from .vcf_utils import (GenoINFO_maker, load_VCF, parse_donor_GPb, read_sparse_GeneINFO,
                        write_VCF)
from .vireo_base import optimal_match
from .vireo_wrap import vireo_wrap


def build_parser():
    """the reference's option surface (vireo.py:35-88), flag for flag"""
    p = OptionParser()
    p.add_option("--cellData", "-c", dest="cell_data", default=None,
                 help="The cell genotype file in VCF format or cellSNP folder with sparse "
                      "matrices.")
    p.add_option("--nDonor", "-N", type="int", dest="n_donor", default=None,
                 help="Number of donors to demultiplex; can be larger than provided in "
                      "donor_file")
    p.add_option("--outDir", "-o", dest="out_dir", default=None,
                 help="Dirtectory for output files [default: $cellFilePath/vireo]")
    g0 = OptionGroup(p, "Optional input files")
    g0.add_option("--vartrixData", dest="vartrix_data", default=None,
                  help="The cell genotype files in vartrix outputs (three/four files, comma "
                       "separated): alt.mtx,ref.mtx,barcodes.tsv,SNPs.vcf.gz. This will "
                       "suppress cellData argument.")
    g0.add_option("--donorFile", "-d", dest="donor_file", default=None,
                  help="The donor genotype file in VCF format. Please filter the sample and "
                       "region with bcftools -s and -R first!")
    g0.add_option("--genoTag", "-t", dest="geno_tag", default='PL',
                  help="The tag for donor genotype: GT, GP, PL [default: %default]")
    g1 = OptionGroup(p, "Optional arguments")
    g1.add_option("--noDoublet", dest="no_doublet", action="store_true", default=False,
                  help="If use, not checking doublets.")
    g1.add_option("--nInit", "-M", type="int", dest="n_init", default=50,
                  help="Number of random initializations, when GT needs to learn "
                       "[default: %default]")
    g1.add_option("--extraDonor", type=int, dest="n_extra_donor", default=0,
                  help="Number of extra donor in pre-cluster, when GT needs to learn "
                       "[default: %default]")
    g1.add_option("--extraDonorMode", dest="extra_donor_mode", default="distance",
                  help="Method for searching from extra donors. size: n_cell per donor; "
                       "distance: GT distance between donors [default: %default]")
    g1.add_option("--forceLearnGT", dest="force_learnGT", default=False, action="store_true",
                  help="If use, treat donor GT as prior only.")
    g1.add_option("--ASEmode", dest="ASE_mode", default=False, action="store_true",
                  help="If use, turn on SNP specific allelic ratio.")
    g1.add_option("--noPlot", dest="no_plot", default=False, action="store_true",
                  help="If use, turn off plotting GT distance (always off in vireo_amd).")
    g1.add_option("--randSeed", type="int", dest="rand_seed", default=None,
                  help="Seed for random initialization [default: %default]")
    g1.add_option("--cellRange", type="str", dest="cell_range", default=None,
                  help="Range of cells to process, eg. 0-10000 [default: all]")
    g1.add_option("--callAmbientRNAs", dest="check_ambient", default=False,
                  action="store_true", help="Not supported by vireo_amd (experimental upstream)")
    g1.add_option("--nproc", "-p", type="int", dest="nproc", default=1,
                  help="Accepted for compatibility; restarts run on the GPU [default: %default]")
    p.add_option_group(g0)
    p.add_option_group(g1)
    return p


def load_cells(options):
    """cellSNP folder, cell VCF or vartrix triplet (vireo.py:108-133)"""
    if options.cell_data is None and options.vartrix_data is None:
        print("Error: need cell data in vcf file, or cellSNP output folder, or "
              "vartrix's alt.mtx,ref.mtx,barcodes.tsv.")
        sys.exit(1)
    if options.vartrix_data is not None:
        print("[vireo] Loading vartrix files ...")
        files = options.vartrix_data.split(",")
        if len(files) < 3 or len(files) > 4:
            print("Error: vartrixData requires 3 or 4 comma separated files")
            sys.exit(1)
        files += [None] * (4 - len(files))
        return read_vartrix(*files)
    if os.path.isdir(os.path.abspath(options.cell_data)):
        print("[vireo] Loading cell folder ...")
        return read_cellSNP(options.cell_data)
    print("[vireo] Loading cell VCF file ...")
    vcf = load_VCF(options.cell_data, biallelic_only=True)
    dat = read_sparse_GeneINFO(vcf['GenoINFO'], keys=['AD', 'DP'])
    for k in ('samples', 'variants', 'FixedINFO', 'contigs', 'comments'):
        dat[k] = vcf[k]
    return dat


def _fail(*lines):
    for line in lines:
        print(line)
    sys.exit(1)


def resolve_donors(options, cell_dat):
    """What the donor arguments ask for (the reference decides this inline, vireo.py:149-189):
    how many donors, whether their genotypes are learned, the genotype prior from the donor
    VCF (matched to the cell variants) and the donor names.  Returns the (possibly
    variant-filtered) cell data and a dict(n_donor, learn_GT, GPb, names, vcf)."""
    n_donor = options.n_donor
    if options.donor_file is None:
        return cell_dat, dict(n_donor=n_donor, learn_GT=True, GPb=None, vcf=None,
                              names=['donor%d' % x for x in range(n_donor)])
    if "variants" not in cell_dat:
        _fail("Error: No variants information is loaded, please provide base.vcf.gz")
    print("[vireo] Loading donor VCF file ...")
    tag = options.geno_tag
    vcf = load_VCF(options.donor_file, biallelic_only=True, sparse=False, format_list=[tag])
    if vcf['n_SNP_tagged'][0] < 0.1 * len(vcf['GenoINFO'][tag]):
        _fail("Error: No " + tag + " tag in donor genotype; please try another tag for genotype, e.g., GT",
              "        %s" % options.donor_file)
    cell_dat, vcf = match_donor_VCF(cell_dat, vcf)
    if len(vcf['GenoINFO'][tag]) == 0:
        _fail("Error: No matching variants found between cell data and donor VCF.")
    GPb = parse_donor_GPb(vcf['GenoINFO'][tag], tag)
    known = GPb.shape[1]
    if n_donor is None or n_donor == known:       # every donor genotyped: nothing to learn
        n_donor, learn_GT, names = known, False, vcf['samples']
    elif n_donor < known:                          # a subset of the genotyped donors is pooled
        learn_GT, names = False, ['donor%d' % x for x in range(n_donor)]
    else:                                          # extra, ungenotyped donors
        learn_GT = True
        names = vcf['samples'] + ['donor%d' % x for x in range(known, n_donor)]
    return cell_dat, dict(n_donor=n_donor, learn_GT=learn_GT, GPb=GPb, names=names, vcf=vcf)


def main(argv=None):
    start = time.time()
    parser = build_parser()
    argv = sys.argv[1:] if argv is None else argv
    (options, _args) = parser.parse_args(argv)
    if len(argv) == 0:
        print("Welcome to vireoSNP v%s (vireo_amd)!\n" % __version__)
        print("use -h or --help for help on argument.")
        sys.exit(1)

    if options.out_dir is None:                                    # vireo.py:96-106
        print("Warning: no outDir provided, we use $cellFilePath/vireo.")
        out_dir = os.path.dirname(os.path.abspath(options.cell_data)) + "/vireo"
    elif os.path.dirname(options.out_dir) == "":
        out_dir = "./" + options.out_dir
    else:
        out_dir = options.out_dir
    if not os.path.exists(out_dir):
        os.mkdir(out_dir)

    cell_dat = load_cells(options)
    if options.cell_range is not None:                              # vireo.py:136-142
        lo, hi = (int(x) for x in options.cell_range.split("-"))
        cell_dat['AD'] = cell_dat['AD'][:, lo:hi]
        cell_dat['DP'] = cell_dat['DP'][:, lo:hi]
        cell_dat['samples'] = cell_dat['samples'][lo:hi]
    if cell_dat['AD'].shape[0] == 0:
        print("Error: cell data in vcf file, or cellSNP output folder, or "
              "vartrix's alt.mtx,ref.mtx,barcodes.tsv does not contain any variants.")
        sys.exit(1)

    cell_dat, donors = resolve_donors(options, cell_dat)
    n_donor, learn_GT, donor_GPb = donors["n_donor"], donors["learn_GT"], donors["GPb"]
    donor_names, donor_vcf = donors["names"], donors["vcf"]

    counts = device_counts(cell_dat['AD'], cell_dat['DP'])         # one upload for everything
    n_vars = counts.n_vars()                                        # vireo.py:191
    if options.force_learnGT:
        learn_GT = True
    n_extra_donor = 0
    if learn_GT:
        n_extra_donor = options.n_extra_donor
    n_init = options.n_init if learn_GT else 1
    check_doublet = options.no_doublet == False                     # noqa: E712

    print("[vireo] Demultiplex %d cells to %d donors with %d variants."
          % (counts.n_cell, n_donor, counts.n_var))
    res = vireo_wrap(counts, None, n_donor=n_donor, GT_prior=donor_GPb, learn_GT=learn_GT,
                     n_init=n_init, n_extra_donor=n_extra_donor,
                     extra_donor_mode=options.extra_donor_mode, check_doublet=check_doublet,
                     random_seed=options.rand_seed, ASE_mode=options.ASE_mode,
                     check_ambient=options.check_ambient, nproc=options.nproc)

    if n_donor is not None and donor_GPb is not None and n_donor < donor_GPb.shape[1]:
        idx = optimal_match(res['GT_prob'], donor_GPb)[1]           # vireo.py:219-222
        donor_names = [donor_vcf['samples'][x] for x in idx]

    write_donor_id(out_dir, donor_names, cell_dat['samples'], n_vars, res)

    if learn_GT and 'variants' in cell_dat:                         # vireo.py:236-242
        AD_reads, DP_reads = counts.donor_reads(res['ID_prob'])
        out = cell_dat
        out['samples'] = donor_names
        out['GenoINFO'] = GenoINFO_maker(res['GT_prob'], AD_reads, DP_reads)
        write_VCF(out_dir + "/GT_donors.vireo.vcf.gz", out)

    run_time = time.time() - start
    print("[vireo] All done: %d min %.1f sec" % (int(run_time / 60), run_time % 60))
    print()


if __name__ == "__main__":
    main()
