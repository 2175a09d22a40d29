"""c3 data, K = 16: one restart per model against R restarts per model (column blocks of 16):
per-pass and dense ms per iteration (scratch measurement)"""
import sys, json
import numpy as np
sys.path.insert(0, ".")
from vireo_amd import _lib, synth
from vireo_amd.counts import DeviceCounts
from vireo_amd.engine import DeviceBatch
N, M, K, d = synth.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "c3"]
K = int(sys.argv[2]) if len(sys.argv) > 2 else K
RS = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 2, 4]
w = synth.donor_workload(N, M, K, d, seed=0)
c = DeviceCounts.from_merged(w["shape"], w["colptr"], w["rowidx"], w["ad"], w["dp"], device=0)
rng = np.random.default_rng(0)
mu, sm = np.linspace(0.01, 0.99, 3)[None, :], np.full((1, 3), 50.0)
for R in RS:
    db = DeviceBatch(c, _lib.KIND_VIREO, K, R)
    for r in range(R):
        db.set_restart(r, rng.random((M, K)), rng.random((N, K, 3)), mu, sm, raw=True)
    db.set_prior(np.full((1, K), 1.0 / K), np.full((1, K, 3), 1.0 / 3), np.array([[0.3, 3.0, 29.7]]), np.array([[29.7, 3.0, 0.3]]))
    db.run_iters(100)        # (clocks up: 100 iterations before the 100 that are timed)
    _, ms = db.run_iters(100)
    ms = ms / 5
    db.profile(True); db.run_iters(10); pm, n = db.profile_read(); db.profile(False)
    print(json.dumps(dict(K=K, R=R, ms_iter=round(ms / 20, 4), per_restart=round(ms / 20 / R, 4),
                          variant=round(pm[0] / 10, 4), cell=round(pm[1] / 10, 4), dense=round(pm[2] / 10, 4))), flush=True)
    db.close()
