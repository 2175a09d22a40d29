"""A/B timing of library variants (scratch/lib_*.so) and / or environment knobs at c3: per-pass ms.
usage: ab_bench.py default scratch/libA.so default@VIREO_CELL_FORM=0,VIREO_LDS_PARITY=0 ..."""
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
    from vireo_amd.engine import DeviceModel
    from vireo_amd.vireo_model import Vireo
    N, M, K, d = synth.CONFIGS[os.environ.get("AB_CONFIG", "c3")]
    K = int(os.environ.get("AB_K", K))
    cache = "/tmp/ab_%s.npz" % os.environ.get("AB_CONFIG", "c3")
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
    dm.run_iters(3, theta_from_iter=10 ** 9)
    tr, ms = dm.run_iters(20, theta_from_iter=0)
    dm.profile(True)
    dm.run_iters(10, theta_from_iter=0)
    pm, n = dm.profile_read()
    info = dm.info()
    timing = None
    if hasattr(_lib.lib(), "vrx_debug_timing"):     # scratch builds with -DVRX_TIMING
        import ctypes
        buf = (ctypes.c_ulonglong * 16)()
        _lib.lib().vrx_debug_timing(buf)
        dm.run_iters(10, theta_from_iter=0)
        _lib.lib().vrx_debug_timing(buf)
        timing = {}
        for mode, name in ((0, "variant"), (1, "cell")):
            tot, b1, b2, st, nw = [buf[mode * 8 + i] for i in range(5)]
            if nw:
                timing[name] = dict(cycles_per_wave=tot // nw, barrier1=round(b1 / tot, 3),
                                    barrier2=round(b2 / tot, 3), stage=round(st / tot, 3))
    print(json.dumps(dict(timing=timing, ms_iter=round(ms / 20, 4), variant=round(pm[0] / max(n[0], 1), 4),
                          cell=round(pm[1] / max(n[1], 1), 4), dense=round(pm[2] / 10, 4),
                          elbo=float(tr[-1]), pad_v=info["pad_variant"], pad_c=info["pad_cell"],
                          form=info["cell_form"])))
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
        print(arg, out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-1500:],
              flush=True)
