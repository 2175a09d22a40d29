"""VRAM in use after repeated balanced builds of the c3 problem (a leak of the build's transients would show)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vireo_amd import synth
from vireo_amd.counts import DeviceCounts
from tests.perf.big_probe import vram_used
N, M, K, d = synth.CONFIGS["c3"]
w = synth.donor_workload(N, M, K, d, seed=0)
print("before", vram_used() / 1e9, flush=True)
for balance in (False, True, True, True, False):
    c = DeviceCounts.from_merged(w["shape"], w["colptr"], w["rowidx"], w["ad"], w["dp"], device=0, balance=balance)
    a = vram_used()
    time.sleep(0.5)
    b = vram_used()
    c.close()
    print("balance=%s: built %.3f GB (0.5 s later %.3f), closed %.3f GB" % (balance, a / 1e9, b / 1e9, vram_used() / 1e9), flush=True)
