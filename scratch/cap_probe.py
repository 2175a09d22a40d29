"""TIMING PROBE: what would a capped stream buy the passes?  scratch/lib_cap.so
(-DVRX_CAP_PROBE) DROPS the words of a (row, slab) beyond the cap -- wrong results, right timing of
the walk on a stream with less lock-step padding; the overflow words (their share is printed) would
still have to be added by some other route, whose cost scratch/gather_bench.hip prices."""
import os, sys, time
sys.path.insert(0, ".")
import numpy as np
from vireo_amd import _lib, synth
from vireo_amd.counts import DeviceCounts
from vireo_amd.engine import DeviceModel
from vireo_amd.vireo_model import Vireo
N, M, K, dens = synth.CONFIGS["c3"]
w = synth.donor_workload(N, M, K, dens, seed=0)
nnz = int(w["rowidx"].size)
np.random.seed(1)
host = Vireo(n_var=N, n_cell=M, n_donor=K)
caps = [(0, 0), (16, 0), (0, 16), (16, 16), (12, 12), (20, 20), (0, 0)]
for cc, cv in caps:
    os.environ["VIREO_CAP_PROBE_CELL"], os.environ["VIREO_CAP_PROBE_VAR"] = str(cc), str(cv)
    counts = DeviceCounts.from_merged(w["shape"], w["colptr"], w["rowidx"], w["ad"], w["dp"])
    dm = DeviceModel(counts, _lib.KIND_VIREO, K, n_gt=3)
    dm.set_state(host.ID_prob, host.GT_prob, host.beta_mu, host.beta_sum)
    dm.set_prior(host.ID_prior, host.GT_prior, host.theta_s1_prior, host.theta_s2_prior)
    dm.run_iters(60, theta_from_iter=3)
    t0 = time.perf_counter()
    dm.run_iters(100, theta_from_iter=0)
    ms_it = (time.perf_counter() - t0) * 10
    dm.profile(True)
    dm.run_iters(100, theta_from_iter=0)
    pm, pn = dm.profile_read()
    info = dm.info()
    print("cap cell %2d var %2d: iteration %.4f ms; variant pass %.4f, cell pass %.4f, dense %.4f; stream slots per non-zero: cell %.3f variant %.3f"
          % (cc, cv, ms_it, pm[0] / max(pn[0], 1), pm[1] / max(pn[1], 1), pm[2] / 100, info["pad_cell"], info["pad_variant"]), flush=True)
    dm.close()
    counts.close()
