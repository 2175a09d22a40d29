"""A/B of library variants / env knobs on the launch-bound sizes: us per iteration of ONE restart
(n_batch = 1) at c2 (N=10k x M=5k, K=4) and at the demo-data size (N=3784 x M=952, K=4), 500
iterations after 50 warm-up ones, three times.  usage: ab_small.py default default@VIREO_X=1 lib.so"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np
    from vireo_amd import _lib, synth
    from vireo_amd.counts import DeviceCounts
    from vireo_amd.engine import DeviceBatch
    out = {}
    for name, (N, M, K, dens) in (("c2", synth.CONFIGS["c2"]), ("c1size", (3784, 952, 4, 0.02))):
        w = synth.donor_workload(N, M, K, dens, seed=0)
        counts = DeviceCounts.from_merged(w["shape"], w["colptr"], w["rowidx"], w["ad"], w["dp"], device=0)
        rng = np.random.default_rng(0)
        mu, sm = np.linspace(0.01, 0.99, 3)[None, :], np.full((1, 3), 50.0)
        db = DeviceBatch(counts, _lib.KIND_VIREO, K, 1)
        db.set_prior(np.full((1, K), 1.0 / K), np.full((1, K, 3), 1.0 / 3),
                     np.array([[0.3, 3.0, 29.7]]), np.array([[29.7, 3.0, 0.3]]))
        db.set_restart(0, rng.random((M, K)), rng.random((N, K, 3)), mu, sm, raw=True)
        db.run_iters(50)
        runs = []
        for _ in range(3):
            tr, ms = db.run_iters(500)
            runs.append(round(ms / 500 * 1e3, 2))
        out[name] = dict(us=runs, elbo=float(np.ravel(tr)[-1]))
        db.close()
    print(json.dumps(out))
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
