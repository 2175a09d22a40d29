"""One rank of tests/test_gpu_shard.py: the sharded ``vireo_wrap`` with REAL device fits, every
rank on GPU 0 (VIREO_DEVICE=0), the ranks talking through tests/tcp_comm.py (RCCL refuses two
ranks on one device; the shard only needs an all-gather of n_init doubles and a broadcast).

    python tests/_tcp_worker.py <rank> <world> <port> <out.pkl> <case> <n_init>

case "c1": the demo data (tests/golden/c1_data.npz), n_donor=4, random_seed=2
case "c2": the SURVEY.md 8(d) generator at N=10k x M=5k, K=4 (BASELINE.json configs[1]),
           random_seed=5, no doublets
case "bmm": BinomMixtureVB(n_donor=3).fit(min_iter=30, n_init=<n_init>, random_seed=1, comm=comm)
           on the mitoDNA demo data (tests/golden/mito_data.npz; bmm_model.py:204-263)
"""
import contextlib
import io
import os
import pickle
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(rank, world, port, out_path, case, n_init):
    import numpy as np
    import vireo_amd
    from vireo_amd import restarts, synth
    from tests.tcp_comm import TcpComm
    W = sys.modules["vireo_amd.vireo_wrap"]
    comm = TcpComm(rank, world, port)
    if case == "bmm":
        from tests import gold
        AD, DP = gold.mito()
        b = vireo_amd.BinomMixtureVB(n_var=AD.shape[0], n_cell=AD.shape[1], n_donor=3)
        b.fit(AD, DP, min_iter=30, n_init=n_init, random_seed=1, verbose=False, comm=comm)
        rv = dict(ID_prob=b.ID_prob, beta_mu=b.beta_mu, beta_sum=b.beta_sum, ELBO_iters=b.ELBO_iters,
                  ELBO_inits=b.ELBO_inits,
                  rng_after=(np.random.get_state()[1][:8].copy(), int(np.random.get_state()[2])))
        with open(out_path, "wb") as f:
            pickle.dump(rv, f)
        comm.barrier()
        comm.close()
        return
    if case == "c1":
        from tests import gold
        AD, DP = gold.c1()
        counts = vireo_amd.DeviceCounts(AD, DP, device=0)
        kw = dict(n_donor=4, n_init=n_init, random_seed=2)
    else:
        N, M, K, dens = synth.CONFIGS["c2"]
        w = synth.donor_workload(N, M, K, dens, seed=0)
        counts = vireo_amd.DeviceCounts.from_merged(w["shape"], w["colptr"], w["rowidx"], w["ad"],
                                                    w["dp"], device=0)
        kw = dict(n_donor=K, n_init=n_init, random_seed=5, check_doublet=False)
    restarts.PHASES = {}
    with contextlib.redirect_stdout(io.StringIO()):
        rv = vireo_amd.vireo_wrap(counts, None, comm=comm, **kw)
    rv["search"] = dict(W.LAST_SEARCH)
    rv["phases"] = dict(restarts.PHASES)
    rv["rng_after"] = np.random.get_state()[1][:8].copy(), int(np.random.get_state()[2])
    with open(out_path, "wb") as f:
        pickle.dump(rv, f)
    comm.barrier()
    comm.close()


if __name__ == "__main__":
    main(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5], int(sys.argv[6]))
