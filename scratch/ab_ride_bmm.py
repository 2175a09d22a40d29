"""VIREO_ELBO_RIDE A/B in clone mode: c5 (200 x 200k, K = 8) and a mito-sized problem: ms per iteration
(run_iters) and the wall time of BinomMixtureVB.fit(n_init=10)"""
import os, sys, time
sys.path.insert(0, ".")
import numpy as np
from vireo_amd import _lib, synth
from vireo_amd.counts import DeviceCounts
from vireo_amd.engine import DeviceModel
from vireo_amd.bmm_model import BinomMixtureVB
for name, (N, M, K) in (("c5", (200, 200000, 8)), ("small clone", (60, 3000, 5))):
    AD, DP = synth.clone_workload(N, M, K, seed=0)
    counts = DeviceCounts(AD, DP)
    np.random.seed(1)
    host = BinomMixtureVB(n_var=N, n_cell=M, n_donor=K)
    its, fits = {"1": [], "0": []}, {"1": [], "0": []}
    for rep in range(3):
        for ride in ("1", "0"):
            os.environ["VIREO_ELBO_RIDE"] = ride
            dm = DeviceModel(counts, _lib.KIND_BMM, K)
            host._push(dm)
            dm.run_iters(10)
            tr, ms = dm.run_iters(100)
            its[ride].append(ms / 100 * 1e3)
            dm.close()
            b = BinomMixtureVB(n_var=N, n_cell=M, n_donor=K)
            t0 = time.perf_counter()
            b.fit(counts, None, n_init=10, random_seed=1, verbose=False)
            fits[ride].append((time.perf_counter() - t0) * 1e3)
    print("%s N=%d M=%d K=%d nnz=%d: us per iteration ride=1 %s ride=0 %s; fit(n_init=10) ms ride=1 %s ride=0 %s"
          % (name, N, M, K, counts.nnz, ["%.1f" % x for x in its["1"]], ["%.1f" % x for x in its["0"]],
             ["%.1f" % x for x in fits["1"]], ["%.1f" % x for x in fits["0"]]), flush=True)
