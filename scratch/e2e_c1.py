"""whole-job timing on the reference's demo data (c1: 3784 variants x 952 cells, K=4, n_init=50)"""
import sys, os, time, io, contextlib, numpy as np
sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import gold
import vireo_amd
AD, DP = gold.c1()
import cProfile, pstats
for rep in range(2):
    pr = cProfile.Profile(); pr.enable()
    t = time.time()
    with contextlib.redirect_stdout(io.StringIO()):
        rv = vireo_amd.vireo_wrap(AD, DP, n_donor=4, n_init=50, random_seed=1, check_doublet=True)
    dt = time.time() - t
    pr.disable()
    print("vireo_wrap n_init=50: %.3fs; best LB %.3f LB_doublet %.3f" % (dt, max(rv["LB_list"]), rv["LB_doublet"]))
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
