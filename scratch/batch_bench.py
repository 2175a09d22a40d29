"""restart-iterations/s of R restarts in one device model (scratch measurement)"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from oracle import vireo_oracle as O
from vireo_amd import _lib
from vireo_amd.counts import DeviceCounts
from vireo_amd.engine import DeviceBatch, DeviceModel

def go(N, M, K, dens, Rs, iters=40):
    AD, DP = O.synth_donor(N, M, K, dens, seed=1)
    c = DeviceCounts(AD, DP)
    print("N=%d M=%d K=%d nnz=%d" % (N, M, K, c.nnz), flush=True)
    rng = np.random.default_rng(0)
    mu, sm = np.array([[0.01, 0.5, 0.99]]), np.full((1, 3), 50.0)
    for R in Rs:
        db = DeviceBatch(c, _lib.KIND_VIREO, K, R)
        for r in range(R):
            db.set_restart(r, rng.random((M, K)), rng.random((N, K, 3)), mu, sm, raw=True)
        db.run_iters(5)
        tr, ms = db.run_iters(iters)
        print("  R=%2d  %.1f us/iteration  %.1f us/restart-iteration  lds=%s" % (
            R, ms / iters * 1e3, ms / iters / R * 1e3, db.info()["lds_cell"]), flush=True)
        db.close()

go(10000, 5000, 4, 0.02, [1, 2, 4, 8, 16])
go(10000, 5000, 8, 0.02, [1, 2, 4])
go(1000, 400, 4, 0.05, [1, 4, 16])
go(30000, 20000, 4, 0.02, [1, 2, 4, 8])
go(30000, 20000, 16, 0.02, [1, 2])
