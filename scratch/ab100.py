"""Steady-state A/B of library variants / env knobs at c3: ms per iteration over 100 iterations after
50 warm-up ones, three times (the 20-iteration ab_bench.py numbers include the clock ramp of an
idle GPU).  usage: ab100.py default default@VIREO_X=1 scratch/lib_y.so ..."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "child":
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
        np.savez(cache, **{k: w[k] for k in ("shape", "colptr", "rowidx", "ad", "dp")})
    counts = DeviceCounts.from_merged(w["shape"], w["colptr"], w["rowidx"], w["ad"], w["dp"], device=0)
    np.random.seed(1)
    host = Vireo(n_var=N, n_cell=M, n_donor=K)
    dm = DeviceModel(counts, _lib.KIND_VIREO, K, n_gt=3)
    dm.set_state(host.ID_prob, host.GT_prob, host.beta_mu, host.beta_sum)
    dm.set_prior(host.ID_prior, host.GT_prior, host.theta_s1_prior, host.theta_s2_prior)
    dm.run_iters(50, theta_from_iter=10 ** 9)
    runs = []
    for _ in range(3):
        t0 = time.perf_counter()
        tr, ms = dm.run_iters(100, theta_from_iter=0)
        runs.append(round((time.perf_counter() - t0) * 10, 4))
    dm.profile(True)
    dm.run_iters(50, theta_from_iter=0)
    pm, n = dm.profile_read()
    print(json.dumps(dict(ms_iter=runs, variant=round(pm[0] / max(n[0], 1), 4), cell=round(pm[1] / max(n[1], 1), 4),
                          dense=round(pm[2] / 50, 4), elbo=float(tr[-1]))))
else:
    for arg in sys.argv[1:]:
        lib, _, knobs = arg.partition("@")
        e = dict(os.environ)
        for kv in filter(None, knobs.split(",")):
            k, _, v = kv.partition("=")
            e[k] = v
        if lib != "default":
            e["VIREO_LIB"] = os.path.join(ROOT, lib)
        out = subprocess.run([sys.executable, __file__, "child"], env=e, capture_output=True, text=True)
        print(arg, out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-1500:], flush=True)
