"""The `vireo` command end to end on a synthetic cellSNP folder (scratch measurement):
load (VCF + two MatrixMarket files), device problem, restarts, doublets, output files.
usage: e2e_cli.py [config=mid] [n_init=8]"""
import cProfile, gzip, io, os, pstats, shutil, sys, time, contextlib
import numpy as np
sys.path.insert(0, os.getcwd())
from vireo_amd import synth
from vireo_amd import vireo as cli

cfg = sys.argv[1] if len(sys.argv) > 1 else "mid"
n_init = sys.argv[2] if len(sys.argv) > 2 else "8"
N, M, K, d = synth.CONFIGS[cfg]
w = synth.donor_workload(N, M, K, d, seed=0)
root = "/tmp/e2e_cli_%s" % cfg
shutil.rmtree(root, ignore_errors=True)
os.makedirs(root + "/cells")
t = time.time()
cols = np.repeat(np.arange(M, dtype=np.int64), np.diff(w["colptr"])) + 1
rows = w["rowidx"].astype(np.int64) + 1
for name, val, keep in (("AD", w["ad"], w["ad"] > 0), ("DP", w["dp"], w["dp"] > 0)):
    with open(root + "/cells/cellSNP.tag.%s.mtx" % name, "w") as f:
        f.write("%%MatrixMarket matrix coordinate integer general\n%\n")
        f.write("%d %d %d\n" % (N, M, int(keep.sum())))
        np.savetxt(f, np.stack([rows[keep], cols[keep], val[keep]], 1), fmt="%d")
with gzip.open(root + "/cells/cellSNP.base.vcf.gz", "wt") as f:
    f.write("##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n")
    for i in range(N):
        f.write("1\t%d\t.\tA\tG\t.\tPASS\tAD=1;DP=2;OTH=0\n" % (i + 1))
with open(root + "/cells/cellSNP.samples.tsv", "w") as f:
    f.write("\n".join("CELL%07d-1" % i for i in range(M)) + "\n")
print("wrote the cellSNP folder in %.1f s (nnz %d)" % (time.time() - t, rows.size), flush=True)
pr = cProfile.Profile()
t = time.time()
pr.enable()
with contextlib.redirect_stdout(io.StringIO()) as out:
    cli.main(["-c", root + "/cells", "-N", str(K), "-o", root + "/out", "--randSeed", "1", "-M", n_init])
pr.disable()
print("vireo command: %.2f s" % (time.time() - t))
print(out.getvalue()[-600:])
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
print(sorted(os.listdir(root + "/out")))
