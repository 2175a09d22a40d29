"""Worker of tests/test_host_cpu.py::test_sharded_clone_mode_gloo_world2 (one process per rank).

``BinomMixtureVB.fit(comm=)`` ("BMM: same restart shard"; bmm_model.py:242-254) with
the device model replaced by the CPU oracle (test infrastructure), so that the sharded control flow
-- ownership of the initialisations, the C continuation of the NumPy stream skipping the other
rank's draws, the ELBO all-gather, the owner's final fit, the broadcast -- runs on CPU with gloo."""
import os
import pickle
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(rank, world, port, out_path, n_init):
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank,
                            world_size=world)
    from oracle import vireo_oracle as O
    from tests import gold
    from tests.gloo_comm import GlooComm
    import vireo_amd
    B = sys.modules["vireo_amd.bmm_model"]

    AD, DP = gold.mito()
    N, M = AD.shape
    fits = []

    class OracleModel:
        """vireo_amd.engine.DeviceModel's surface as BinomMixtureVB uses it, on the oracle"""

        def __init__(self, counts, kind, n_donor, fix_beta_sum=False, **kw):
            self.st = O.bmm_new(M, N, n_donor, fix_beta_sum=fix_beta_sum,
                                ID_prob_init=np.ones((M, n_donor)))
            self.snap = None

        def set_state(self, ID, GT, mu, sm):
            self.st.ID_prob, self.st.beta_mu, self.st.beta_sum = ID.copy(), np.array(mu), np.array(sm)

        def set_prior(self, ID_prior, GT_prior, s1, s2):
            shape = self.st.beta_mu.shape
            self.st.ID_prior = np.broadcast_to(ID_prior, self.st.ID_prob.shape).copy()
            self.st.theta_s1_prior = np.broadcast_to(s1, shape).copy()
            self.st.theta_s2_prior = np.broadcast_to(s2, shape).copy()

        def fit(self, max_iter, min_iter, eps, delay=0):
            self.st.ELBO_iters = np.array([])
            it = O.bmm_fit_vb(self.st, AD, DP, max_iter=max_iter, min_iter=min_iter, epsilon_conv=eps)
            fits.append(it)
            return np.append(self.st.ELBO_iters, 0.0), it, 0      # (trace[:it] is what the caller keeps)

        def get_state(self):
            return self.st.ID_prob, None, self.st.beta_mu, self.st.beta_sum

        def snapshot(self):
            self.snap = (self.st.ID_prob.copy(), self.st.beta_mu.copy(), self.st.beta_sum.copy())

        def restore(self):
            self.st.ID_prob, self.st.beta_mu, self.st.beta_sum = (x.copy() for x in self.snap)

        def close(self):
            pass

    fake_counts = types.SimpleNamespace(shape=AD.shape, nnz=DP.nnz, binom_const=lambda: O.binom_const(AD, DP))
    B.DeviceModel = OracleModel
    B.device_counts = lambda a, b=None, **kw: fake_counts
    B.restart_batch = lambda *a, **k: 1                     # one initialisation per (oracle) model
    b = vireo_amd.BinomMixtureVB(n_var=N, n_cell=M, n_donor=3)
    b.fit(AD, DP, min_iter=30, n_init=n_init, random_seed=1, verbose=False, comm=GlooComm())
    rv = dict(ID_prob=b.ID_prob, beta_mu=b.beta_mu, beta_sum=b.beta_sum, ELBO_iters=b.ELBO_iters,
              ELBO_inits=b.ELBO_inits, n_fits_on_rank=len(fits),
              rng_after=(np.random.get_state()[1][:8].copy(), int(np.random.get_state()[2])))
    with open(out_path, "wb") as f:
        pickle.dump(rv, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5]))
