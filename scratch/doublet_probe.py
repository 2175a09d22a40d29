"""time predict_doublet (vireo_doublet.py:11-82) at c3 / K = 16 on a fitted model"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import vireo_amd as va
from vireo_amd import synth
from vireo_amd.counts import DeviceCounts
from vireo_amd.vireo_doublet import predict_doublet
cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
N, M, K, dens = synth.CONFIGS[cfg]
w = synth.donor_workload(N, M, K, dens, seed=0)
counts = DeviceCounts.from_merged(w["shape"], w["colptr"], w["rowidx"], w["ad"], w["dp"])
np.random.seed(1)
m = va.Vireo(n_var=N, n_cell=M, n_donor=K)
m.fit(counts, None, min_iter=5, max_iter=20, delay_fit_theta=3, verbose=False)
for rep in range(3):
    t0 = time.perf_counter()
    dp, ip, llr = predict_doublet(m, counts, None)
    t1 = time.perf_counter()
    print("predict_doublet %s K=%d: %.3f s  (doublet_prob %s, max %.3g)" % (cfg, K, t1 - t0, dp.shape, dp.max()), flush=True)
