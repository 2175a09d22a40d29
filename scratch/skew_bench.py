"""LDS pass robustness on heavy-tailed data: ms/iteration with VIREO_LDS_SORT on/off and
with the global-gather pass, same inputs.  usage: skew_bench.py [config] [sigma_var] [sigma_cell]"""
import os, sys, time, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

def one():
    from vireo_amd import _lib, synth
    from vireo_amd.counts import DeviceCounts
    from vireo_amd.engine import DeviceModel
    from vireo_amd.vireo_model import Vireo
    cfg = sys.argv[1]; sv = float(sys.argv[2]); sc = float(sys.argv[3])
    N, M, K, d = synth.CONFIGS[cfg]
    w = synth.donor_workload(N, M, K, d, seed=0, skew=None if sv == 0 and sc == 0 else (sv, sc))
    rows = np.bincount(w["rowidx"], minlength=N); cols = np.diff(w["colptr"])
    counts = DeviceCounts.from_merged(w["shape"], w["colptr"], w["rowidx"], w["ad"], w["dp"], device=0)
    np.random.seed(1)
    host = Vireo(n_var=N, n_cell=M, n_donor=K)
    dm = DeviceModel(counts, _lib.KIND_VIREO, K, n_gt=3)
    dm.set_state(host.ID_prob, host.GT_prob, host.beta_mu, host.beta_sum)
    dm.set_prior(host.ID_prior, host.GT_prior, host.theta_s1_prior, host.theta_s2_prior)
    dm.run_iters(3, theta_from_iter=10**9)
    dm.profile(True)
    tr, ms = dm.run_iters(10, theta_from_iter=0)
    pm, n = dm.profile_read()
    print(json.dumps(dict(nnz=int(w["rowidx"].size), maxdp=int(w["dp"].max()),
        var_nnz=[int(rows.min()), float(np.median(rows)), int(rows.max())],
        cell_nnz=[int(cols.min()), float(np.median(cols)), int(cols.max())],
        ms_iter=ms / 10, variant=pm[0] / max(n[0], 1), cell=pm[1] / max(n[1], 1),
        info=dm.info(), elbo=float(tr[-1]))))

if len(sys.argv) > 4 and sys.argv[4] == "child":
    one()
else:
    for name, env in [("lds+sort", {}), ("lds nosort", {"VIREO_LDS_SORT": "0", "VIREO_LDS_MAX_PAD": "100"}),
                      ("global", {"VIREO_LDS": "0"})]:
        e = dict(os.environ); e.update(env)
        out = subprocess.run([sys.executable, __file__] + sys.argv[1:4] + ["child"], env=e,
                             capture_output=True, text=True)
        print(name, out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-2000:], flush=True)
