"""The ``vireo`` command end to end on the GPU against the text outputs of the reference
command (tests/golden/cli/, produced by tests/golden/make_cli_golden.py): the five modes of
the reference's examples/demo.sh, a single-init / no-doublet run, and one run for each remaining
input / flag (--vartrixData, --cellRange, --extraDonor in both modes, --ASEmode), --randSeed 2."""
import gzip
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
DATA = os.path.join(HERE, "golden", "data")
GOLD = os.path.join(HERE, "golden", "cli")

MODES = {
    "mode1_noGT": ["-c", DATA + "/cellSNP_mat", "-N", "4"],
    "mode2_PL": ["-c", DATA + "/cells.cellSNP.vcf.gz", "-d", DATA + "/donors.cellSNP.vcf.gz",
                 "-N", "4"],
    "mode3_part": ["-c", DATA + "/cells.cellSNP.vcf.gz", "-d",
                   DATA + "/donors.two.cellSNP.vcf.gz", "-N", "4"],
    "mode4_learn": ["-c", DATA + "/cells.cellSNP.vcf.gz", "-d", DATA + "/donors.cellSNP.vcf.gz",
                    "-N", "4", "--forceLearnGT"],
    "mode5_PL3": ["-c", DATA + "/cells.cellSNP.vcf.gz", "-d", DATA + "/donors.cellSNP.vcf.gz",
                  "-N", "3"],
    "mode1_M1_noDoublet": ["-c", DATA + "/cellSNP_mat", "-N", "4", "-M", "1", "--noDoublet"],
    # the remaining inputs / flags of the reference command (vireo.py:36-84, :136-142;
    # io_utils.py:62-88 read_vartrix)
    "vartrix": ["--vartrixData", ",".join([DATA + "/vartrix/alt.mtx", DATA + "/vartrix/ref.mtx",
                                           DATA + "/vartrix/barcodes.tsv",
                                           DATA + "/cellSNP_mat/cellSNP.base.vcf.gz"]),
                "-N", "4", "-M", "2"],
    "cellRange": ["-c", DATA + "/cellSNP_mat", "-N", "4", "-M", "2", "--cellRange", "100-600"],
    "extraDonor": ["-c", DATA + "/cellSNP_mat", "-N", "3", "-M", "4", "--extraDonor", "1"],
    "extraDonor_size": ["-c", DATA + "/cellSNP_mat", "-N", "3", "-M", "4", "--extraDonor", "1",
                        "--extraDonorMode", "size"],
    "ASEmode": ["-c", DATA + "/cellSNP_mat", "-N", "4", "-M", "2", "--ASEmode"],
}


# (rows of donor_ids.tsv, lines of GT_donors.vireo.vcf.gz) whose text differs from the reference's
OBSERVED_TEXT_DIFFS = {}


def _table(path):
    with open(path) as f:
        return [line.rstrip("\n").split("\t") for line in f]


def _num_close(a, b, rtol):
    try:
        x, y = float(a), float(b)
    except ValueError:
        return a == b
    return abs(x - y) <= rtol * max(abs(x), abs(y)) + 1e-12


@pytest.mark.parametrize("mode", sorted(MODES))
def test_cli_matches_reference_outputs(mode, tmp_path, capsys):
    from vireo_amd import _lib
    from vireo_amd.vireo import main
    _lib.require_gpu()
    out = str(tmp_path / mode)
    main(MODES[mode] + ["-o", out, "--randSeed", "2", "--noPlot"])
    capsys.readouterr()
    ref = os.path.join(GOLD, mode)

    # summary.tsv: identical text (the donor / doublet / unassigned counts)
    assert open(out + "/summary.tsv").read() == open(ref + "/summary.tsv").read()

    # donor_ids.tsv: identical labels; printed probabilities (3 significant digits) may
    # differ in the last printed digit when a value sits on a rounding boundary
    got, want = _table(out + "/donor_ids.tsv"), _table(ref + "/donor_ids.tsv")
    assert len(got) == len(want)
    n_text_diff = 0
    for g, w in zip(got, want):
        assert [g[i] for i in (0, 1, 4, 5, 6)] == [w[i] for i in (0, 1, 4, 5, 6)]
        assert _num_close(g[2], w[2], 2e-2) and _num_close(g[3], w[3], 2e-2)
        assert abs(float(g[7]) - float(w[7])) <= 2e-3 if g[7] != "doublet_logLikRatio" else True
        n_text_diff += g != w

    # _log.txt: logLik line identical, theta shapes to 1e-6 relative
    glog, wlog = open(out + "/_log.txt").read().split("\n"), open(ref + "/_log.txt").read().split("\n")
    assert glog[0] == wlog[0]
    # (--ASEmode: one row per variant, which NumPy abbreviates with "...")
    nums = lambda lines: np.array([x for x in " ".join(lines).replace("[", " ").replace("]", " ").split()   # noqa: E731
                                   if x != "..."], float)
    assert len(glog) == len(wlog)
    np.testing.assert_allclose(nums(glog[2:]), nums(wlog[2:]), rtol=1e-6)

    # estimated donor genotypes: identical VCF text where the reference writes one
    ref_vcf = ref + "/GT_donors.vireo.vcf.gz"
    assert os.path.exists(out + "/GT_donors.vireo.vcf.gz") == os.path.exists(ref_vcf)
    if os.path.exists(ref_vcf):
        g = gzip.open(out + "/GT_donors.vireo.vcf.gz", "rt").read().split("\n")
        w = gzip.open(ref_vcf, "rt").read().split("\n")
        assert len(g) == len(w)
        diff = [i for i, (a, b) in enumerate(zip(g, w)) if a != b]
        n_vcf_diff = len(diff)
    else:
        n_vcf_diff = 0
    # Observed on MI355X: the output TEXT equals the reference's byte for byte in every mode
    # (3-significant-digit probabilities, integer PL / AD / DP fields: a value would have to sit
    # within ~1e-10 of a rounding boundary to print differently).  Anything else is a regression.
    print("%s: donor_ids.tsv rows that differ as text: %d; VCF lines that differ: %d"
          % (mode, n_text_diff, n_vcf_diff))
    assert (n_text_diff, n_vcf_diff) == OBSERVED_TEXT_DIFFS.get(mode, (0, 0))
    # prob_singlet / prob_doublet (io_utils.py:147-170; here the library's threaded writer):
    # the decompressed text against the reference's own bytes
    for name in ("prob_singlet.tsv.gz", "prob_doublet.tsv.gz"):
        assert os.path.exists(out + "/" + name)
        g = gzip.open(out + "/" + name, "rt").read().split("\n")
        w = gzip.open(ref + "/" + name, "rt").read().split("\n")
        assert len(g) == len(w) and g[0] == w[0]
        bad = [i for i, (a, b) in enumerate(zip(g, w)) if a != b]
        print("%s: %s lines that differ as text: %d" % (mode, name, len(bad)))
        assert not bad, (name, bad[:3], g[bad[0]], w[bad[0]])
    # the probability tables (written by the library's threaded writer): one row per cell in
    # donor_ids.tsv order, and the largest singlet probability of a row prints as prob_max does
    rows = [line.split("\t") for line in gzip.open(out + "/prob_singlet.tsv.gz", "rt").read().splitlines()]
    assert rows[0][0] == "cell" and [r[0] for r in rows[1:]] == [g[0] for g in got[1:]]
    for r, g in zip(rows[1:], got[1:]):
        assert "%.2e" % max(float(x) for x in r[1:]) == g[2]


def test_cli_with_a_communicator_writes_the_same_files(tmp_path, capsys, monkeypatch):
    """the command launched once per GPU shares its restarts over the ranks (RCCL) and rank 0
    writes; here the same code path at world size 1 against the plain run"""
    import socket
    from vireo_amd import _lib
    from vireo_amd.vireo import main
    _lib.require_gpu()
    mode = sorted(MODES)[0]
    plain, shared = str(tmp_path / "plain"), str(tmp_path / "shared")
    main(MODES[mode] + ["-o", plain, "--randSeed", "2", "--noPlot"])
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    for k, v in dict(VIREO_CLI_FORCE_RCCL="1", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                     MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port)).items():
        monkeypatch.setenv(k, v)
    main(MODES[mode] + ["-o", shared, "--randSeed", "2", "--noPlot"])
    capsys.readouterr()
    for name in ("summary.tsv", "donor_ids.tsv", "_log.txt"):
        assert open(plain + "/" + name).read() == open(shared + "/" + name).read()
    for name in ("prob_singlet.tsv.gz", "prob_doublet.tsv.gz"):
        assert gzip.open(plain + "/" + name).read() == gzip.open(shared + "/" + name).read()


def test_cli_nGPU_launches_its_own_ranks(tmp_path):
    """``vireo --nGPU 2`` with no launcher around it: the command starts one process per rank
    (vireo_amd/launch.py; the GPU counterpart of the reference's -p / nproc, vireo_wrap.py:74-91),
    the ranks share the restarts, rank 0 writes.  Two ranks on the ONE device of the test box
    (VIREO_DEVICE=0, host-socket communicator: RCCL refuses two ranks on a device): the files
    equal the reference command's byte for byte, like the single-process run's."""
    import subprocess
    import sys
    from vireo_amd import _lib
    _lib.require_gpu()
    mode = "mode1_noGT"
    out = str(tmp_path / "ngpu2")
    env = dict(os.environ, PYTHONPATH=os.path.dirname(HERE), VIREO_COMM="tcp", VIREO_DEVICE="0",
               VIREO_RESTART_BATCH="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, "-m", "vireo_amd.vireo"] + MODES[mode] +
                       ["-o", out, "--randSeed", "2", "--noPlot", "--nGPU", "2"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=os.path.dirname(HERE))
    assert p.returncode == 0, p.stderr[-2000:]
    assert p.stdout.count("[vireo] All done") == 1          # rank 0 speaks for all
    ref = os.path.join(GOLD, mode)
    for name in ("summary.tsv", "donor_ids.tsv"):
        assert open(out + "/" + name).read() == open(ref + "/" + name).read(), name
    for name in ("prob_singlet.tsv.gz", "prob_doublet.tsv.gz", "GT_donors.vireo.vcf.gz"):
        assert gzip.open(out + "/" + name, "rt").read() == gzip.open(ref + "/" + name, "rt").read(), name
    # a rank that fails takes the command down with a non-zero code instead of a hang
    bad = subprocess.run([sys.executable, "-m", "vireo_amd.vireo", "-c", str(tmp_path / "nowhere"),
                          "-N", "4", "-o", out + "_bad", "--nGPU", "2"],
                         env=env, capture_output=True, text=True, timeout=300, cwd=os.path.dirname(HERE))
    assert bad.returncode != 0
