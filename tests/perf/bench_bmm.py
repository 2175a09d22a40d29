"""Secondary number (BASELINE.json configs[4]): BinomMixtureVB clone mode, N=200 x M=200k, K=8.
EM iterations/s of vrx_model_run_iters next to one oracle iteration on the same inputs."""
import sys, os, time, json, numpy as np
sys.path.insert(0, os.getcwd())
from vireo_amd import _lib
from vireo_amd.counts import DeviceCounts
from vireo_amd.engine import DeviceModel
from vireo_amd.bmm_model import BinomMixtureVB
from oracle import vireo_oracle as O
N, M, K = 200, 200000, 8
t = time.time(); AD, DP = O.synth_clone(N, M, K, seed=0); tg = time.time() - t
t = time.time(); counts = DeviceCounts(AD, DP); tu = time.time() - t
np.random.seed(1)
host = BinomMixtureVB(n_var=N, n_cell=M, n_donor=K)
dm = DeviceModel(counts, _lib.KIND_BMM, K)
host._push(dm)
dm.run_iters(3)
steps = 20
t0 = time.perf_counter(); tr, ms = dm.run_iters(steps); wall = time.perf_counter() - t0
dm.profile(True); dm.run_iters(steps); pm, pn = dm.profile_read()
np.random.seed(1); ref = O.bmm_new(M, N, K)
t = time.time(); O.bmm_theta_step(ref, AD, DP); L = O.bmm_cell_loglik(ref, AD, DP); O.bmm_id_step(ref, L); e = O.bmm_elbo(ref, L); tc = time.time() - t
host2 = BinomMixtureVB(n_var=N, n_cell=M, n_donor=K, ID_prob_init=host.ID_prob)
dm2 = DeviceModel(counts, _lib.KIND_BMM, K); host2._push(dm2); tr1, _ = dm2.run_iters(1)
print(json.dumps(dict(workload="c5 BinomMixtureVB N=200 x M=200000 K=8 nnz=%d" % counts.nnz, it_per_s=steps / wall,
      ms_per_iteration=wall / steps * 1e3, passes_ms=dict(variant=pm[0] / max(pn[0], 1), cell=pm[1] / max(pn[1], 1), dense=pm[2] / steps),
      info=dm.info(), cpu_oracle_s_per_iteration=tc, speedup=steps / wall * tc,
      elbo_rel_err_first_iteration=abs(tr1[0] - e) / abs(e), host_s=dict(generate=tg, upload=tu))))
if "--passes-only" in sys.argv:
    sys.exit(0)
# the whole BinomMixtureVB.fit (10 initialisations of up to 100 iterations + the final fit), one
# initialisation at a time against the packed batches
for batch in ("1", "0"):
    os.environ["VIREO_RESTART_BATCH"] = batch
    b = BinomMixtureVB(n_var=N, n_cell=M, n_donor=K)
    t0 = time.perf_counter(); b.fit(counts, None, n_init=10, random_seed=1, verbose=False); dt = time.perf_counter() - t0
    print(json.dumps(dict(fit="BinomMixtureVB.fit(n_init=10)", restart_batch="auto" if batch == "0" else 1,
                          wall_s=round(dt, 3), ELBO_final=float(b.ELBO_iters[-1]), iterations_final_fit=len(b.ELBO_iters))))
