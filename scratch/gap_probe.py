"""What makes the first timed run after a short warm-up slower than its repeats: clocks (idle gap
before it) or the trajectory (which iterations of the protocol are timed)?"""
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
A = DeviceModel(counts, _lib.KIND_VIREO, K, n_gt=3)
B = DeviceModel(counts, _lib.KIND_VIREO, K, n_gt=3)
for dm in (A, B):
    dm.set_state(host.ID_prob, host.GT_prob, host.beta_mu, host.beta_sum)
    dm.set_prior(host.ID_prior, host.GT_prior, host.theta_s1_prior, host.theta_s2_prior)

def timed(dm, n=20):
    t0 = time.perf_counter()
    tr, ms = dm.run_iters(n, theta_from_iter=0)
    return round((time.perf_counter() - t0) / n * 1e3, 4), round(ms / n, 4)

res = {}
for gap in (0.0, 0.005, 0.02, 0.1, 1.0):
    B.run_iters(500, theta_from_iter=0)                 # ~0.35 s of work: clocks are up
    time.sleep(gap)
    A.set_state(host.ID_prob, host.GT_prob, host.beta_mu, host.beta_sum)     # (~45 MB upload)
    t_up = time.perf_counter()
    A.run_iters(5, theta_from_iter=3)
    runs = [timed(A) for _ in range(5)]
    res["gap %.3f s + upload" % gap] = runs
# no upload between the heavy work and the warm-up: state restored on the device
A.set_state(host.ID_prob, host.GT_prob, host.beta_mu, host.beta_sum)
A.snapshot()
for gap in (0.0, 0.02):
    B.run_iters(500, theta_from_iter=0)
    time.sleep(gap)
    A.restore()
    A.run_iters(5, theta_from_iter=3)
    res["gap %.3f s, device restore" % gap] = [timed(A) for _ in range(5)]
# trajectory: the same 20 iterations (5..25 of the protocol) timed in the middle of continuous work
B.run_iters(500, theta_from_iter=0)
A.restore()
A.run_iters(5, theta_from_iter=3)
first = timed(A)
A.restore()
A.run_iters(5, theta_from_iter=3)
second = timed(A)
res["same iterations twice (restore, warm-up 5, time 20)"] = [first, second]
# per-iteration cost along the trajectory: chunks of 5 iterations
A.restore()
A.run_iters(5, theta_from_iter=3)
res["chunks of 5 from iteration 5"] = [timed(A, 5) for _ in range(10)]
print(json.dumps(res, indent=1))
