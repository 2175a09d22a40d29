"""tests/test_gpu_fuzz.py's random cases for seeds beyond the suite's 48 (one-off robustness sweep)"""
import os, sys
sys.path.insert(0, ".")
import numpy as np
from tests.test_gpu_fuzz import draw_case, close
from oracle import vireo_oracle as O
import vireo_amd as va
from vireo_amd.counts import DeviceCounts
lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad = []
for seed in range(lo, hi):
    AD, DP, K, rng = draw_case(seed)
    N, M = AD.shape
    os.environ["VIREO_LDS"] = "1" if seed % 2 else "0"
    os.environ["VIREO_LDS_BLOCKS"] = str(int(rng.choice([1, 16, 1024])))
    try:
        counts = DeviceCounts(AD, DP)
        if seed % 4 == 3:
            K = max(K, 2)
            np.random.seed(seed)
            init = np.random.rand(M, K)
            ref = O.bmm_new(M, N, K, ID_prob_init=init.copy())
            dev = va.BinomMixtureVB(n_cell=M, n_var=N, n_donor=K, ID_prob_init=init.copy())
            O.bmm_fit_vb(ref, AD, DP, min_iter=2, max_iter=4)
            dev._fit_BV(AD, DP, min_iter=2, max_iter=4, verbose=False)
            assert len(dev.ELBO_iters) == len(ref.ELBO_iters)
            close(dev.ELBO_iters, ref.ELBO_iters); close(dev.ID_prob, ref.ID_prob); close(dev.beta_mu, ref.beta_mu)
        else:
            flags = dict(ASE_mode=bool(rng.random() < 0.2), fix_beta_sum=bool(rng.random() < 0.2),
                         learn_theta=bool(rng.random() < 0.85))
            np.random.seed(seed); ref = O.vireo_new(M, N, K, **flags)
            np.random.seed(seed); dev = va.Vireo(n_cell=M, n_var=N, n_donor=K, **flags)
            O.vireo_fit(ref, AD, DP, min_iter=2, max_iter=5, delay_fit_theta=1)
            dev.fit(counts, None, min_iter=2, max_iter=5, delay_fit_theta=1, verbose=False)
            assert len(dev.ELBO_) == len(ref.ELBO_)
            close(dev.ELBO_, ref.ELBO_); close(dev.ID_prob, ref.ID_prob); close(dev.GT_prob, ref.GT_prob)
            close(dev.beta_mu, ref.beta_mu); close(dev.beta_sum, ref.beta_sum)
    except Exception as e:      # noqa: BLE001
        bad.append((seed, repr(e)[:300]))
        print("seed", seed, "FAILED", repr(e)[:300], flush=True)
print("seeds %d..%d: %d failures" % (lo, hi - 1, len(bad)), bad)
