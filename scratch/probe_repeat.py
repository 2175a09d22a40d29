"""Repeat the probed iteration a few times: which workgroups are the slowest, and where they ran
(scratch/lib_probe.so, -DVRX_PROBE_BUILD)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["VIREO_LIB"] = os.path.join(ROOT, "scratch", "lib_probe.so")
import numpy as np
from vireo_amd import _lib, synth
from vireo_amd.counts import DeviceCounts
from vireo_amd.engine import DeviceModel
from vireo_amd.vireo_model import Vireo

N, M, K, d = synth.CONFIGS["c3"]
cache = "/tmp/ab_c3.npz"
if os.path.exists(cache):
    w = dict(np.load(cache))
    w["shape"] = tuple(int(x) for x in w["shape"])
else:
    w = synth.donor_workload(N, M, K, d, seed=0)
counts = DeviceCounts.from_merged(w["shape"], w["colptr"], w["rowidx"], w["ad"], w["dp"], device=0)
np.random.seed(1)
host = Vireo(n_var=N, n_cell=M, n_donor=K)
dm = DeviceModel(counts, _lib.KIND_VIREO, K, n_gt=3)
dm.set_state(host.ID_prob, host.GT_prob, host.beta_mu, host.beta_sum)
dm.set_prior(host.ID_prior, host.GT_prior, host.theta_s1_prior, host.theta_s2_prior)
dm.run_iters(3, theta_from_iter=10 ** 9)
MAXW = 16384
buf = (ctypes.c_ulonglong * (2 * MAXW * 8))()
_lib.lib().vrx_debug_probe(buf)
for rep in range(2):
    dm.run_iters(1, theta_from_iter=0)
    _lib.lib().vrx_debug_probe(buf)
    rec = np.frombuffer(buf, dtype=np.uint64).reshape(2, MAXW, 8).astype(np.int64)
    for mode, name in ((0, "variant"), (1, "cell")):
        r = rec[mode]
        r = r[r[:, 1] > 0]
        wg = r[: len(r) // 16 * 16].reshape(-1, 16, 8)
        dur = (wg[:, :, 1].max(1) - wg[:, :, 0].min(1)).astype(float)
        o = np.argsort(dur)[::-1][:4]
        where = ["x%d:%03x" % ((wg[i, 0, 7] >> 32) & 15, (wg[i, 0, 7] >> 8) & 0xfff) for i in o]
        print(rep, name, "median %d" % np.median(dur), "slowest", list(o), (dur[o] / np.median(dur)).round(3), where,
              "stage/visit-ish", [int(wg[i, :, 4].mean()) for i in o], flush=True)

vis = (ctypes.c_ulonglong * (2 * 16 * 128 * 3))()
_lib.lib().vrx_debug_probe_visits(vis)
np.save(os.path.join(ROOT, "gpurun_out", "probe_visits.npy"),
        np.frombuffer(vis, dtype=np.uint64).reshape(2, 16, 128, 3).astype(np.int64))
