"""VIREO_SOFTMAX_BLOCKS_PER_CU A/B at c5 (clone mode, M = 200k cells): us per iteration and the dense part"""
import os, sys
sys.path.insert(0, ".")
import numpy as np
from vireo_amd import _lib, synth
from vireo_amd.counts import DeviceCounts
from vireo_amd.engine import DeviceModel
from vireo_amd.bmm_model import BinomMixtureVB
AD, DP = synth.clone_workload(200, 200000, 8, seed=0)
counts = DeviceCounts(AD, DP)
np.random.seed(1)
host = BinomMixtureVB(n_var=200, n_cell=200000, n_donor=8)
for rep in range(2):
    for cap in ("8", "0", "4", "16"):
        os.environ["VIREO_SOFTMAX_BLOCKS_PER_CU"] = cap
        dm = DeviceModel(counts, _lib.KIND_BMM, 8)
        host._push(dm)
        dm.run_iters(10)
        tr, ms = dm.run_iters(100)
        dm.profile(True); dm.run_iters(50); pm, pn = dm.profile_read()
        print("c5 softmax blocks per CU %2s: %.1f us per iteration; dense kernels %.1f us; ELBO %.10g" % (cap, ms * 10, pm[2] / 50 * 1e3, tr[-1]), flush=True)
        dm.close()
