"""why is the first timed run after the warm-up ~1.3 % slower than its repeats?  wall and device ms"""
import sys, time, json
import numpy as np
sys.path.insert(0, ".")
from vireo_amd import _lib, synth
from vireo_amd.counts import DeviceCounts
from vireo_amd.engine import DeviceModel
from vireo_amd.vireo_model import Vireo
N, M, K, d = synth.CONFIGS["c3"]
w = synth.donor_workload(N, M, K, d, seed=0)
counts = DeviceCounts.from_merged(w["shape"], w["colptr"], w["rowidx"], w["ad"], w["dp"], device=0)
np.random.seed(1)
host = Vireo(n_var=N, n_cell=M, n_donor=K)
dm = DeviceModel(counts, _lib.KIND_VIREO, K, n_gt=3)
dm.set_state(host.ID_prob, host.GT_prob, host.beta_mu, host.beta_sum)
dm.set_prior(host.ID_prior, host.GT_prior, host.theta_s1_prior, host.theta_s2_prior)
mode = sys.argv[1] if len(sys.argv) > 1 else "a"
if mode == "a":
    dm.run_iters(600, theta_from_iter=3)
elif mode == "b":      # the warm-up as three calls of 200
    for _ in range(3):
        dm.run_iters(200, theta_from_iter=0)
out = []
for _ in range(5):
    t0 = time.perf_counter()
    tr, ms = dm.run_iters(200, theta_from_iter=0)
    out.append((round((time.perf_counter() - t0) * 5, 4), round(ms / 200, 4)))
print(mode, out)
