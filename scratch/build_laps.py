"""wall-clock laps of the c3 problem build, default and balanced (VIREO_BUILD_TIMING=1)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["VIREO_BUILD_TIMING"] = "1"
import numpy as np
from vireo_amd import synth
from vireo_amd.counts import DeviceCounts
N, M, K, d = synth.CONFIGS["c3"]
w = synth.donor_workload(N, M, K, d, seed=0)
for balance in (False, True, False, True, True):
    t0 = time.perf_counter()
    c = DeviceCounts.from_merged(w["shape"], w["colptr"], w["rowidx"], w["ad"], w["dp"], device=0, balance=balance)
    print("balance=%s: %.3f s" % (balance, time.perf_counter() - t0), file=sys.stderr, flush=True)
    c.close()
