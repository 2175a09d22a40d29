"""mid-size (N=50k x M=20k) parity of the LDS-resident passes for odd K against the oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
os.environ["VIREO_LDS"] = "1"
import vireo_amd
from vireo_amd import synth
from vireo_amd.counts import DeviceCounts
from oracle import vireo_oracle as O
N, M, _, d = synth.CONFIGS["mid"]
w = synth.donor_workload(N, M, 16, d, seed=0)
AD, DP = synth.as_scipy(w)
counts = DeviceCounts.from_merged(w["shape"], w["colptr"], w["rowidx"], w["ad"], w["dp"], device=0)
for K in (7, 10, 13):
    np.random.seed(5)
    ref = O.vireo_new(M, N, K)
    np.random.seed(5)
    dev = vireo_amd.Vireo(n_cell=M, n_var=N, n_donor=K)
    O.vireo_fit(ref, AD, DP, min_iter=1, max_iter=3)
    dev.fit(counts, None, min_iter=1, max_iter=3, verbose=False)
    e = np.max(np.abs(dev.ELBO_ - ref.ELBO_) / np.abs(ref.ELBO_))
    i = np.max(np.abs(dev.ID_prob - ref.ID_prob) / np.maximum(ref.ID_prob, 1e-300))
    g = np.max(np.abs(dev.GT_prob - ref.GT_prob))
    print("K=%d iterations %d/%d  ELBO rel %.2e  ID_prob max rel %.2e  GT_prob max abs %.2e"
          % (K, len(dev.ELBO_), len(ref.ELBO_), e, i, g), flush=True)
