"""who is closer to the mathematics on the clone-mode fuzz cases that miss rtol 1e-5 on tiny
posteriors -- the GPU or the oracle (= the reference's float64 arithmetic)?  The same fit in
80-bit arithmetic (dense long-double products; digamma through mpmath)."""
import os, sys
sys.path.insert(0, ".")
import numpy as np
import mpmath
from tests.test_gpu_fuzz import draw_case
from oracle import vireo_oracle as O
import vireo_amd as va
LD = np.longdouble
mpmath.mp.dps = 40
def psi_ld(x):
    flat = x.ravel()
    return np.array([LD(mpmath.nstr(mpmath.digamma(mpmath.mpf(float(v))), 30)) for v in flat], dtype=LD).reshape(x.shape)
for seed in [int(a) for a in sys.argv[1:]]:
    AD, DP, K, rng = draw_case(seed)
    N, M = AD.shape
    os.environ["VIREO_LDS"] = "1" if seed % 2 else "0"
    os.environ["VIREO_LDS_BLOCKS"] = str(int(rng.choice([1, 16, 1024])))
    K = max(K, 2)
    np.random.seed(seed)
    init = np.random.rand(M, K)
    ref = O.bmm_new(M, N, K, ID_prob_init=init.copy())
    dev = va.BinomMixtureVB(n_cell=M, n_var=N, n_donor=K, ID_prob_init=init.copy())
    O.bmm_fit_vb(ref, AD, DP, min_iter=2, max_iter=4)
    dev._fit_BV(AD, DP, min_iter=2, max_iter=4, verbose=False)
    # exact: float64 state of the digamma ARGUMENTS is part of both implementations (theta is stored
    # as float64), so the arbiter follows the iteration in long double throughout
    A, D = AD.toarray().astype(LD), DP.toarray().astype(LD)
    B = D - A
    ID = (init / init.sum(1, keepdims=True)).astype(LD)
    ID = (init.astype(LD) / init.astype(LD).sum(1, keepdims=True))
    n_it = len(ref.ELBO_iters) + 1
    for it in range(n_it):
        t1 = A @ ID + LD(1); t2 = B @ ID + LD(1)
        mu, sm = t1 / (t1 + t2), t1 + t2
        s1, s2 = mu * sm, (1 - mu) * sm
        d1, d2, ds = psi_ld(s1.astype(np.float64)), psi_ld(s2.astype(np.float64)), psi_ld((s1 + s2).astype(np.float64))
        # (digamma arguments rounded to float64 before mpmath: they are O(1..1e6), rel 1e-16)
        L = A.T @ (d1 - ds) + B.T @ (d2 - ds)
        Z = L - np.log(LD(K))
        Z = Z - Z.max(1, keepdims=True)
        ID = np.exp(Z); ID = ID / ID.sum(1, keepdims=True)
    ex = ID.astype(np.float64)
    def worst(x):
        m = ex > 1e-290
        return float(np.max(np.abs(x[m] - ex[m]) / ex[m]))
    print("seed %d (N=%d M=%d K=%d, max count %d): iterations oracle %d gpu %d; worst relative error of ID_prob vs exact: GPU %.2e, oracle %.2e; GPU vs oracle %.2e"
          % (seed, N, M, K, DP.max(), len(ref.ELBO_iters), len(dev.ELBO_iters), worst(dev.ID_prob), worst(ref.ID_prob),
             float(np.max(np.abs(dev.ID_prob - ref.ID_prob)[ref.ID_prob > 1e-290] / ref.ID_prob[ref.ID_prob > 1e-290]))), flush=True)
