"""the heavy-tailed c3 workload alone, for rocprofv3 --kernel-trace --stats"""
import sys, time, json
import numpy as np
sys.path.insert(0, ".")
from vireo_amd import _lib, synth
from vireo_amd.counts import DeviceCounts
from vireo_amd.engine import DeviceModel
from vireo_amd.vireo_model import Vireo
N, M, K, d = synth.CONFIGS["c3"]
w = synth.donor_workload(N, M, K, d, seed=0, skew=synth.C3_SKEW)
counts = DeviceCounts.from_merged(w["shape"], w["colptr"], w["rowidx"], w["ad"], w["dp"], device=0)
np.random.seed(1)
host = Vireo(n_var=N, n_cell=M, n_donor=K)
dm = DeviceModel(counts, _lib.KIND_VIREO, K, n_gt=3)
dm.set_state(host.ID_prob, host.GT_prob, host.beta_mu, host.beta_sum)
dm.set_prior(host.ID_prior, host.GT_prior, host.theta_s1_prior, host.theta_s2_prior)
dm.run_iters(30, theta_from_iter=3)
t0 = time.perf_counter()
dm.run_iters(100, theta_from_iter=0)
print("ms per iteration", (time.perf_counter() - t0) * 10, dm.info())
