"""Worker of tests/test_host_cpu.py::test_sharded_wrap_gloo_world2 (one process per rank).

The GPU fit is replaced by the CPU oracle (test infrastructure) so that the complete
sharded control flow of vireo_amd.vireo_wrap -- RNG consumption on every rank, restart
ownership, ELBO all-gather, winner broadcast -- runs on CPU with gloo."""
import os
import pickle
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(rank, world, port, out_path):
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank,
                            world_size=world)
    from oracle import vireo_oracle as O
    from tests import gold
    import vireo_amd
    W = sys.modules["vireo_amd.vireo_wrap"]   # the package attribute is the function
    from vireo_amd import Vireo
    from vireo_amd.dist import GlooComm

    AD, DP = gold.c1()
    fake_counts = types.SimpleNamespace(shape=AD.shape, n_var=AD.shape[0], n_cell=AD.shape[1])
    fitted = []

    def to_oracle(self):
        st = O.vireo_new(self.n_cell, self.n_var, self.n_donor, n_GT=self.n_GT,
                         learn_GT=self.learn_GT, learn_theta=self.learn_theta,
                         ASE_mode=self.ASE_mode, fix_beta_sum=self.fix_beta_sum,
                         ID_prob_init=np.ones((self.n_cell, self.n_donor)),
                         GT_prob_init=np.ones((self.n_var, self.n_donor, self.n_GT)))
        for k in ("ID_prob", "GT_prob", "beta_mu", "beta_sum", "ID_prior", "GT_prior",
                  "theta_s1_prior", "theta_s2_prior", "ELBO_"):
            setattr(st, k, getattr(self, k))
        return st

    def from_oracle(self, st):
        for k in ("ID_prob", "GT_prob", "beta_mu", "beta_sum", "ELBO_"):
            setattr(self, k, getattr(st, k))

    def fake_fit(self, counts, _dp, max_iter=200, min_iter=5, epsilon_conv=1e-2,
                 delay_fit_theta=0, verbose=True, **_):
        st = to_oracle(self)
        O.vireo_fit(st, AD, DP, max_iter, min_iter, epsilon_conv, delay_fit_theta)
        from_oracle(self, st)
        fitted.append(len(self.ELBO_))

    def fake_doublet(vobj, counts, _dp):
        st = to_oracle(vobj)
        r = O.vireo_doublet(st, AD, DP)
        from_oracle(vobj, st)
        return r

    Vireo.fit = fake_fit
    W.predict_doublet = fake_doublet
    W.device_counts = lambda a, b=None: fake_counts
    import io
    import contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        rv = W.vireo_wrap(AD, DP, n_donor=4, n_init=4, random_seed=2, comm=GlooComm())
    rv["n_fits_on_rank"] = len(fitted)
    with open(out_path, "wb") as f:
        pickle.dump(rv, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4])
