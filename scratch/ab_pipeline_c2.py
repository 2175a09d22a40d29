"""pipelined polls on small, launch-bound problems -- vireo_wrap turnaround at c2 size
with VIREO_FIT_PIPELINE=0 / 1, one restart per model and the default packing (16 per model)"""
import contextlib, io, os, sys, time
sys.path.insert(0, ".")
import numpy as np
import vireo_amd as va
from vireo_amd import synth
from vireo_amd.counts import DeviceCounts
for cfg in ("c2", "small"):
    N, M, K, dens = synth.CONFIGS[cfg]
    w = synth.donor_workload(N, M, K, dens, seed=0)
    counts = DeviceCounts.from_merged(w["shape"], w["colptr"], w["rowidx"], w["ad"], w["dp"])
    for batch in ("1", "0"):
        os.environ["VIREO_RESTART_BATCH"] = batch
        res = {}
        for rep in range(5):
            for pipe in ("1", "0"):
                os.environ["VIREO_FIT_PIPELINE"] = pipe
                t0 = time.perf_counter()
                with contextlib.redirect_stdout(io.StringIO()):
                    rv = va.vireo_wrap(counts, None, n_donor=K, n_init=32, random_seed=1, check_doublet=False)
                res.setdefault(pipe, []).append(time.perf_counter() - t0)
        print("%s N=%d M=%d K=%d, n_init=32, VIREO_RESTART_BATCH=%s: vireo_wrap wall ms  pipeline=1 %s (min %.1f)  pipeline=0 %s (min %.1f)"
              % (cfg, N, M, K, batch, ["%.1f" % (x * 1e3) for x in res["1"]], min(res["1"]) * 1e3,
                 ["%.1f" % (x * 1e3) for x in res["0"]], min(res["0"]) * 1e3), flush=True)
