import sys, os, time, numpy as np, ctypes as C
sys.path.insert(0, os.getcwd())
from vireo_amd import synth, _lib
from vireo_amd.counts import merge_counts, DeviceCounts
N, M, K, d = synth.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "c3"]
w = synth.donor_workload(N, M, K, d, seed=0)
AD, DP = synth.as_scipy(w)
t = time.time(); mc = merge_counts(AD, DP); print("merge_counts %.2fs" % (time.time() - t))
t = time.time(); dc = DeviceCounts(None, None, _merged=mc); print("vrx_problem_create (validate, transpose, pack, segments, tiled streams, upload) %.2fs" % (time.time() - t))
os.environ["VIREO_LDS"] = "0"
t = time.time(); dc2 = DeviceCounts(None, None, _merged=mc); print("  same without the tiled LDS streams %.2fs" % (time.time() - t))
