"""Worker of tests/test_host_cpu.py::test_sharded_wrap_gloo_world2 / _wider_than_n_init (one
process per rank).

The device restarts are replaced by the CPU oracle (test infrastructure) so that the complete
sharded control flow of vireo_amd.vireo_wrap -- the C continuation of the NumPy random
stream (drawn for owned restarts, skipped for the others), restart ownership, ELBO
all-gather, winner broadcast -- runs on CPU with gloo."""
import os
import pickle
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(rank, world, port, out_path, n_init=4):
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank,
                            world_size=world)
    from oracle import vireo_oracle as O
    from tests import gold
    from tests.gloo_comm import GlooComm
    import vireo_amd
    W = sys.modules["vireo_amd.vireo_wrap"]   # the package attribute is the function

    AD, DP = gold.c1()
    fake_counts = types.SimpleNamespace(shape=AD.shape, n_var=AD.shape[0], n_cell=AD.shape[1])
    fitted = []
    STATE = ("ID_prob", "GT_prob", "beta_mu", "beta_sum", "ELBO_")

    def to_oracle(m):
        st = O.vireo_new(m.n_cell, m.n_var, m.n_donor, n_GT=m.n_GT,
                         learn_GT=m.learn_GT, learn_theta=m.learn_theta,
                         ASE_mode=m.ASE_mode, fix_beta_sum=m.fix_beta_sum,
                         ID_prob_init=np.ones((m.n_cell, m.n_donor)),
                         GT_prob_init=np.ones((m.n_var, m.n_donor, m.n_GT)))
        for k in STATE + ("ID_prior", "GT_prior", "theta_s1_prior", "theta_s2_prior"):
            setattr(st, k, getattr(m, k))
        return st

    class OracleRestarts:
        """vireo_amd.restarts.DeviceRestarts with the device model replaced by the CPU oracle"""

        def __init__(self, counts, template):
            self.t, self.best = template, None

        def run(self, im, ID_raw, GT_raw, ID_fixed, GT_fixed, max_iter, delay_fit_theta):
            st = to_oracle(self.t)
            st.ID_prob = O.unit_sum(ID_raw) if ID_raw is not None else ID_fixed
            st.GT_prob = O.unit_sum(GT_raw) if GT_raw is not None else GT_fixed
            st.ELBO_ = np.zeros(0)
            O.vireo_fit(st, AD, DP, max_iter, 5, 1e-2, delay_fit_theta)
            fitted.append(im)
            if self.best is None or st.ELBO_[-1] > self.best[0]:
                self.best = (st.ELBO_[-1], im, st)
            return st.ELBO_[-1]

        def winner(self, im, refine):
            assert self.best[1] == im
            st = self.best[2]
            if refine:
                O.vireo_fit(st, AD, DP, 200, 5, 1e-2, 0)
                fitted.append(-1)
            for k in STATE:
                setattr(self.t, k, getattr(st, k))
            return self.t

        def close(self):
            pass

    def fake_doublet(vobj, counts, _dp):
        st = to_oracle(vobj)
        r = O.vireo_doublet(st, AD, DP)
        for k in STATE:
            setattr(vobj, k, getattr(st, k))
        return r

    W.DeviceRestarts = OracleRestarts
    W.predict_doublet = fake_doublet
    W.device_counts = lambda a, b=None, **kw: fake_counts
    import io
    import contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        rv = W.vireo_wrap(AD, DP, n_donor=4, n_init=n_init, random_seed=2, comm=GlooComm())
    rv["n_fits_on_rank"] = len(fitted)
    with open(out_path, "wb") as f:
        pickle.dump(rv, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4],
         int(sys.argv[5]) if len(sys.argv) > 5 else 4)
