"""LDS-resident vs global-gather passes at c3 size for every K: ms per iteration."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 2 and sys.argv[1] == "child":
    import numpy as np
    from vireo_amd import _lib, synth
    from vireo_amd.counts import DeviceCounts
    from vireo_amd.engine import DeviceModel
    from vireo_amd.vireo_model import Vireo
    N, M, _, d = synth.CONFIGS["c3"]
    w = synth.donor_workload(N, M, 16, d, seed=0)
    counts = DeviceCounts.from_merged(w["shape"], w["colptr"], w["rowidx"], w["ad"], w["dp"], device=0)
    for K in [int(k) for k in sys.argv[2].split(",")]:
        np.random.seed(1)
        host = Vireo(n_var=N, n_cell=M, n_donor=K)
        dm = DeviceModel(counts, _lib.KIND_VIREO, K, n_gt=3)
        dm.set_state(host.ID_prob, host.GT_prob, host.beta_mu, host.beta_sum)
        dm.set_prior(host.ID_prior, host.GT_prior, host.theta_s1_prior, host.theta_s2_prior)
        dm.run_iters(3, theta_from_iter=10 ** 9)
        tr, ms = dm.run_iters(10, theta_from_iter=0)
        print(json.dumps(dict(K=K, lds=dm.info()["lds_cell"], ms_iter=round(ms / 10, 4), elbo=float(tr[-1]))), flush=True)
        del dm
else:
    ks = sys.argv[1] if len(sys.argv) > 1 else "2,3,4,5,6,7,8,10,12,14,16"
    for lds in ("1", "0"):
        e = dict(os.environ, VIREO_LDS=lds)
        out = subprocess.run([sys.executable, __file__, "child", ks], env=e, capture_output=True, text=True)
        print("VIREO_LDS=" + lds); print(out.stdout.strip() or out.stderr[-1500:], flush=True)
