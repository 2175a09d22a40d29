cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python scratch/ab100.py scratch/lib_base.so default > gpurun_out/ab100.txt 2>&1
python - >> gpurun_out/ab100.txt 2>&1 <<'P'
import numpy as np, sys
sys.path.insert(0,'.')
from vireo_amd import _lib, synth
from vireo_amd.counts import DeviceCounts
from vireo_amd.engine import DeviceBatch
N, M, K, dens = synth.CONFIGS["c2"]
w = synth.donor_workload(N, M, K, dens, seed=0)
counts = DeviceCounts.from_merged(w["shape"], w["colptr"], w["rowidx"], w["ad"], w["dp"], device=0)
rng = np.random.default_rng(0)
mu, sm = np.linspace(0.01, 0.99, 3)[None, :], np.full((1, 3), 50.0)
db = DeviceBatch(counts, _lib.KIND_VIREO, K, 1)
db.set_restart(0, rng.random((M, K)), rng.random((N, K, 3)), mu, sm, raw=True)
tr, ms = db.run_iters(60)
print(np.ravel(tr))
P
cat gpurun_out/ab100.txt
