"""cost of the per-iteration convergence read-back in vrx_model_fit (c2-size problem)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vireo_amd
from vireo_amd import synth
from vireo_amd.counts import DeviceCounts
for cfg in ("c2", "x4m"):
    N, M, K, d = synth.CONFIGS[cfg]
    w = synth.donor_workload(N, M, K, d, seed=0)
    counts = DeviceCounts.from_merged(w["shape"], w["colptr"], w["rowidx"], w["ad"], w["dp"], device=0)
    for min_iter in (200, 0):
        np.random.seed(1)
        m = vireo_amd.Vireo(n_cell=M, n_var=N, n_donor=K)
        m.fit(counts, None, max_iter=3, min_iter=3, verbose=False)
        t = time.perf_counter()
        m.fit(counts, None, max_iter=200, min_iter=min_iter, epsilon_conv=-1e300, verbose=False)
        dt = time.perf_counter() - t
        print(cfg, "min_iter", min_iter, "iterations", len(m.ELBO_) - 3 + 1, "us/iter %.1f" % (dt / 200 * 1e6), flush=True)
