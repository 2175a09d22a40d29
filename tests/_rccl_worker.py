"""One rank of tests/test_gpu_fullsize.py::test_restart_shard_over_rccl_world2: the sharded
vireo_wrap on GPU LOCAL_RANK with the library's RCCL communicator (no PyTorch)."""
import contextlib
import io
import os
import pickle
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(out_path):
    import vireo_amd
    from vireo_amd import dist as vdist
    from tests import gold
    rank, world, local = vdist.env_rank_world()
    comm = vdist.RcclComm(rank, world, local, vdist.socket_exchange(rank, world))
    AD, DP = gold.c1()
    counts = vireo_amd.DeviceCounts(AD, DP, device=local)
    with contextlib.redirect_stdout(io.StringIO()):
        rv = vireo_amd.vireo_wrap(counts, None, n_donor=4, n_init=4, random_seed=2, comm=comm)
    with open(out_path, "wb") as f:
        pickle.dump(rv, f)
    comm.barrier()
    comm.close()


if __name__ == "__main__":
    main(sys.argv[1])
