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
from . import launch
from .dist import LocalComm, env_rank_world, make_comm
from .io_utils import match_donor_VCF, read_cellSNP, read_vartrix, write_donor_id
from .vcf_utils import (GenoINFO_maker, load_VCF, parse_donor_GPb, read_sparse_GeneINFO,
                        write_VCF)
from .vireo_base import optimal_match
from .vireo_wrap import vireo_wrap


# The reference's option surface (vireo.py:35-88): same flags, destinations, types and
# defaults; the help texts are this package's own.
#   (flags, dest, kind, default, help)      kind: None = string, int, or "flag" = store_true
_MAIN_OPTIONS = [
    (("--cellData", "-c"), "cell_data", None, None,
     "cell genotypes: a cellSNP output folder (sparse matrices) or a VCF file"),
    (("--nDonor", "-N"), "n_donor", int, None,
     "donors in the pool; may exceed the number of donors in the donor VCF"),
    (("--outDir", "-o"), "out_dir", None, None,
     "output folder [default: <cell data>/vireo]"),
]
_INPUT_OPTIONS = [
    (("--vartrixData",), "vartrix_data", None, None,
     "vartrix output instead of --cellData: alt.mtx,ref.mtx,barcodes.tsv[,SNPs.vcf.gz]"),
    (("--donorFile", "-d"), "donor_file", None, None,
     "donor genotypes (VCF; subset samples and regions with bcftools beforehand)"),
    (("--genoTag", "-t"), "geno_tag", None, "PL",
     "FORMAT tag holding the donor genotypes: GT, GP or PL [default: %default]"),
]
_MODEL_OPTIONS = [
    (("--noDoublet",), "no_doublet", "flag", False, "skip the doublet step"),
    (("--nInit", "-M"), "n_init", int, 50,
     "random restarts when genotypes are learned [default: %default]"),
    (("--extraDonor",), "n_extra_donor", int, 0,
     "additional donors searched for first and dropped afterwards [default: %default]"),
    (("--extraDonorMode",), "extra_donor_mode", None, "distance",
     "which of the extra donors to drop: 'size' (fewest cells) or 'distance' (closest "
     "genotypes) [default: %default]"),
    (("--forceLearnGT",), "force_learnGT", "flag", False,
     "use the donor genotypes as a prior only and learn them"),
    (("--ASEmode",), "ASE_mode", "flag", False, "one allelic ratio per variant"),
    (("--noPlot",), "no_plot", "flag", False,
     "accepted; vireo_amd never draws the genotype-distance figure"),
    (("--randSeed",), "rand_seed", int, None, "seed of the restarts [default: %default]"),
    (("--cellRange",), "cell_range", "str", None, "cells to process, e.g. 0-10000 [default: all]"),
    (("--callAmbientRNAs",), "check_ambient", "flag", False,
     "not available in vireo_amd (experimental upstream)"),
    (("--nproc", "-p"), "nproc", int, 1,
     "accepted; the restarts run on the GPU [default: %default]"),
    (("--nGPU",), "n_gpu", int, 1,
     "GPUs of this node to share the restarts over: the command starts one process per GPU "
     "(restart i on GPU i % nGPU, one RCCL all-gather picks the best) [default: %default]"),
]


def build_parser():
    parser = OptionParser()

    def declare(target, table):
        for flags, dest, kind, default, text in table:
            extra = dict(action="store_true") if kind == "flag" else (
                {} if kind is None else dict(type=kind))
            target.add_option(*flags, dest=dest, default=default, help=text, **extra)

    declare(parser, _MAIN_OPTIONS)
    for title, table in (("Optional input files", _INPUT_OPTIONS),
                         ("Optional arguments", _MODEL_OPTIONS)):
        group = OptionGroup(parser, title)
        declare(group, table)
        parser.add_option_group(group)
    return parser


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

    # --nGPU N: the GPU counterpart of the reference's -p / nproc (a multiprocessing.Pool over the
    # restarts, vireo_wrap.py:74-91).  This process becomes the launcher of N copies of itself
    # (vireo_amd/launch.py); under an external launcher (torch.distributed.run) the flag is moot.
    if options.n_gpu is not None and options.n_gpu > 1 and not launch.launched_externally():
        here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        env = dict(os.environ)
        env["PYTHONPATH"] = here + (os.pathsep + env["PYTHONPATH"] if env.get("PYTHONPATH") else "")
        sys.stdout.flush()
        sys.exit(launch.spawn_ranks([sys.executable, "-c",
                                     "from vireo_amd.vireo import main; main()"] + list(argv),
                                    options.n_gpu, env=env))

    rank, world, local = env_rank_world()
    if world > 1 and rank != 0:
        sys.stdout = open(os.devnull, "w")                          # rank 0 speaks for all

    if options.out_dir is None:                                    # vireo.py:96-106
        print("Warning: no outDir provided, we use $cellFilePath/vireo.")
        out_dir = os.path.dirname(os.path.abspath(options.cell_data)) + "/vireo"
    elif os.path.dirname(options.out_dir) == "":
        out_dir = "./" + options.out_dir
    else:
        out_dir = options.out_dir
    os.makedirs(out_dir, exist_ok=True)     # (one process per GPU may get here at once)

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

    # Launched once per GPU (--nGPU N, or python -m torch.distributed.run --nproc-per-node N -m
    # vireo_amd.vireo ...; RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment) the
    # restarts are shared out over the ranks; every rank computes the same result and rank 0
    # writes the files.
    comm = make_comm(rank, world, local,                           # (the flag: a 1-GPU test of the RCCL path)
                     force_rccl=os.environ.get("VIREO_CLI_FORCE_RCCL") == "1")
    if isinstance(comm, LocalComm):
        comm = None

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
                     check_ambient=options.check_ambient, nproc=options.nproc,
                     **({} if comm is None else {"comm": comm}))
    if comm is not None:
        comm.barrier()
        comm.close()
        if rank != 0:
            return

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
