"""VIREO_ELBO_RIDE forced on / off for restart BATCHES at c2 size (nnz x columns around the 2^24
default threshold): us per restart-iteration, and for x1m / x2m single models"""
import os, sys
sys.path.insert(0, ".")
import numpy as np
from vireo_amd import _lib, synth
from vireo_amd.counts import DeviceCounts
from vireo_amd.engine import DeviceBatch
for cfg, Rs in (("c2", (1, 4, 8, 16)), ("x1m", (1, 2)), ("x2m", (1,))):
    N, M, K, dens = synth.CONFIGS[cfg]
    w = synth.donor_workload(N, M, K, dens, seed=0)
    nnz = int(w["rowidx"].size)
    counts = DeviceCounts.from_merged(w["shape"], w["colptr"], w["rowidx"], w["ad"], w["dp"])
    rng = np.random.default_rng(0)
    mu, sm = np.linspace(0.01, 0.99, 3)[None, :], np.full((1, 3), 50.0)
    for R in Rs:
        out = {}
        for rep in range(3):
            for ride in ("1", "0"):
                os.environ["VIREO_ELBO_RIDE"] = ride
                db = DeviceBatch(counts, _lib.KIND_VIREO, K, R)
                db.set_prior(np.full((1, K), 1.0 / K), np.full((1, K, 3), 1.0 / 3),
                             np.array([[0.3, 3.0, 29.7]]), np.array([[29.7, 3.0, 0.3]]))
                for r in range(R):
                    db.set_restart(r, rng.random((M, K)), rng.random((N, K, 3)), mu, sm, raw=True)
                db.run_iters(10)
                tr, ms = db.run_iters(200)
                out.setdefault(ride, []).append(ms / 200 / R * 1e3)
                info = db.info()
                db.close()
        print("%s K=%d R=%d: nnz x columns = %.1f M (2^24 = 16.8 M), lds %d/%d: us per restart-iteration ride=1 %s ride=0 %s"
              % (cfg, K, R, nnz * K * R / 1e6, info["lds_variant"], info["lds_cell"],
                 ["%.2f" % x for x in out["1"]], ["%.2f" % x for x in out["0"]]), flush=True)
    counts.close()
