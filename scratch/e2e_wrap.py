import sys, os, time, io, contextlib, numpy as np
sys.path.insert(0, os.getcwd())
import vireo_amd
from vireo_amd import synth
from vireo_amd.counts import DeviceCounts
cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
n_init = int(sys.argv[2]) if len(sys.argv) > 2 else 4
N, M, K, d = synth.CONFIGS[cfg]
t = time.time(); w = synth.donor_workload(N, M, K, d, seed=0); print("generate %.1fs" % (time.time() - t))
t = time.time(); AD, DP = synth.as_scipy(w); print("as_scipy %.1fs" % (time.time() - t))
t = time.time(); counts = vireo_amd.device_counts(AD, DP); print("device_counts (merge+upload+transpose+tiling) %.1fs" % (time.time() - t))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
t = time.time()
with contextlib.redirect_stdout(io.StringIO()):
    rv = vireo_amd.vireo_wrap(AD, DP, n_donor=K, n_init=n_init, random_seed=1, check_doublet=True)
dt = time.time() - t
pr.disable()
print("vireo_wrap n_init=%d: %.2fs; LB_list %s LB_doublet %.3f" % (n_init, dt, rv["LB_list"], rv["LB_doublet"]))
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
