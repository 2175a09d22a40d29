"""beyond c3: N = 200k x M = 100k, K = 16, density 0.02 (4e8 non-zeros, 4x c3) -- build time, memory, ms per iteration"""
import sys, time, json
import numpy as np
sys.path.insert(0, ".")
from vireo_amd import _lib, synth
from vireo_amd.counts import DeviceCounts
from vireo_amd.engine import DeviceModel
from vireo_amd.vireo_model import Vireo
N, M, K, d = 200000, 100000, 16, 0.02
t0 = time.time()
w = synth.donor_workload(N, M, K, d, seed=0)
t_gen = time.time() - t0
nnz = int(w["rowidx"].size)
t0 = time.time()
counts = DeviceCounts.from_merged(w["shape"], w["colptr"], w["rowidx"], w["ad"], w["dp"], device=0)
t_build = time.time() - t0
del w
np.random.seed(1)
host = Vireo(n_var=N, n_cell=M, n_donor=K)
dm = DeviceModel(counts, _lib.KIND_VIREO, K, n_gt=3)
dm.set_state(host.ID_prob, host.GT_prob, host.beta_mu, host.beta_sum)
dm.set_prior(host.ID_prior, host.GT_prior, host.theta_s1_prior, host.theta_s2_prior)
dm.run_iters(30, theta_from_iter=3)
t0 = time.perf_counter()
tr, _ = dm.run_iters(50, theta_from_iter=0)
ms = (time.perf_counter() - t0) / 50 * 1e3
dm.profile(True); dm.run_iters(20); pm, pn = dm.profile_read()
import subprocess
mem = subprocess.run(["rocm-smi", "--showmeminfo", "vram"], capture_output=True, text=True).stdout
B = 2 * 12 * nnz + 4 * (N + M + 2) + 8 * (2 * M * K + 2 * N * K * 3 + 8 * N * K)
print(json.dumps(dict(N=N, M=M, K=K, nnz=nnz, generate_s=round(t_gen, 1), build_s=round(t_build, 2), ms_per_iteration=ms,
                      passes_ms=[pm[0] / max(pn[0], 1), pm[1] / max(pn[1], 1), pm[2] / 20], elbo_finite=bool(np.all(np.isfinite(tr))),
                      roofline_frac=B / (ms * 1e-3) / 8e12, info=dm.info())))
print(mem[-400:])
