"""VIREO_ELBO_RIDE A/B on launch-bound problems: us per iteration of a lone restart (run_iters) and
wall ms of whole fits (stop rule active), c2 size and the reference's demo-data size"""
import os, sys, time
sys.path.insert(0, ".")
import numpy as np
import vireo_amd as va
from vireo_amd import _lib, synth
from vireo_amd.counts import DeviceCounts
from vireo_amd.engine import DeviceModel
for name, (N, M, K, dens) in (("c2", synth.CONFIGS["c2"]), ("demo-size", (3784, 952, 4, 0.02)), ("small", synth.CONFIGS["small"])):
    w = synth.donor_workload(N, M, K, dens, seed=0)
    counts = DeviceCounts.from_merged(w["shape"], w["colptr"], w["rowidx"], w["ad"], w["dp"])
    np.random.seed(1)
    host = va.Vireo(n_var=N, n_cell=M, n_donor=K)
    res = {"1": [], "0": []}
    fits = {"1": [], "0": []}
    for rep in range(4):
        for ride in ("1", "0"):
            os.environ["VIREO_ELBO_RIDE"] = ride
            dm = DeviceModel(counts, _lib.KIND_VIREO, K, n_gt=3)
            dm.set_state(host.ID_prob, host.GT_prob, host.beta_mu, host.beta_sum)
            dm.set_prior(host.ID_prior, host.GT_prior, host.theta_s1_prior, host.theta_s2_prior)
            dm.run_iters(20, theta_from_iter=3)
            tr, ms = dm.run_iters(400, theta_from_iter=0)
            res[ride].append(ms / 400 * 1e3)
            ts = []
            for _ in range(5):
                dm.set_state(host.ID_prob, host.GT_prob, host.beta_mu, host.beta_sum)
                t0 = time.perf_counter()
                trace, it, fl = dm.fit(200, 5, 1e-2, 3)
                ts.append((time.perf_counter() - t0) * 1e3)
            fits[ride].append((min(ts), it + 1))
            dm.close()
    print("%s N=%d M=%d K=%d nnz=%d: us per iteration (run_iters)  ride=1 %s  ride=0 %s;  whole fit ms (iterations)  ride=1 %s  ride=0 %s"
          % (name, N, M, K, w["rowidx"].size, ["%.2f" % x for x in res["1"]], ["%.2f" % x for x in res["0"]],
             ["%.3f (%d)" % x for x in fits["1"]], ["%.3f (%d)" % x for x in fits["0"]]), flush=True)
    counts.close()
