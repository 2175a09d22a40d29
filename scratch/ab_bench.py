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
    if hasattr(_lib.lib(), "vrx_debug_probe"):     # scratch builds with -DVRX_PROBE_BUILD
        import ctypes
        MAXW, WPG = 16384, int(os.environ.get("AB_WAVES", "16"))
        buf = (ctypes.c_ulonglong * (2 * MAXW * 8))()
        _lib.lib().vrx_debug_probe(buf)
        dm.run_iters(1, theta_from_iter=0)
        _lib.lib().vrx_debug_probe(buf)
        rec = np.frombuffer(buf, dtype=np.uint64).reshape(2, MAXW, 8).astype(np.int64)
        if os.environ.get("AB_DUMP"):
            np.save(os.environ["AB_DUMP"], rec)
        timing = {}
        for mode, name in ((0, "variant"), (1, "cell")):
            r = rec[mode]
            r = r[r[:, 1] > 0]
            if not len(r):
                continue
            life = (r[:, 1] - r[:, 0]).astype(float)
            frac = lambda c: round(float(r[:, c].sum() / life.sum()), 3)      # noqa: E731
            # workgroups = WPG consecutive waves; CU = (xcc, se/sh/cu bits of HW_ID)
            wg = r[: len(r) // WPG * WPG].reshape(-1, WPG, 8)
            w0, w1 = wg[:, :, 0].min(1), wg[:, :, 1].max(1)
            cu = (wg[:, 0, 7] >> 32) * 65536 + ((wg[:, 0, 7] >> 8) & 0xff)
            busy, gaps, tails = [], [], []
            xspan = {}
            for x in np.unique(wg[:, 0, 7] >> 32):
                m = (wg[:, 0, 7] >> 32) == x
                xspan[int(x)] = (w0[m].min(), w1[m].max())
            for c in np.unique(cu):
                m = cu == c
                o = np.argsort(w0[m])
                a, b = w0[m][o], w1[m][o]
                lo, hi = xspan[int(c // 65536)]
                busy.append(float((b - a).sum()) / float(hi - lo))
                gaps += list(a[1:] - b[:-1])
                tails.append(float(hi - b[-1]) / float(hi - lo))
            skew = (wg[:, :, 1].max(1) - wg[:, :, 1].min(1)).astype(float)   # first-to-last wave end
            timing[name] = dict(
                ticks_per_wave=int(life.mean()), barrier1=frac(2), barrier2=frac(3), stage=frac(4),
                stream_wait=frac(5), prologue=round(float((r[:, 6] - r[:, 0]).sum() / life.sum()), 3),
                n_wg=int(len(wg)), n_cu=int(len(np.unique(cu))),
                xcd_span_ticks=int(np.mean([hi - lo for lo, hi in xspan.values()])),
                wg_ticks_mean=int((w1 - w0).mean()), wg_ticks_max=int((w1 - w0).max()),
                wave_end_skew_mean=int(skew.mean()),
                cu_busy_mean=round(float(np.mean(busy)), 3), cu_busy_min=round(float(np.min(busy)), 3),
                gap_ticks_mean=int(np.mean(gaps)) if gaps else None,
                gap_ticks_max=int(np.max(gaps)) if gaps else None,
                cu_tail_idle_mean=round(float(np.mean(tails)), 3), cu_tail_idle_max=round(float(np.max(tails)), 3),
                wgs_per_cu=[int(x) for x in np.bincount(np.unique(cu, return_counts=True)[1])])
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
