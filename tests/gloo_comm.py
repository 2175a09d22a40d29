"""A communicator with vireo_amd.dist's interface (rank, world, allgather, bcast, barrier)
over an initialised torch.distributed process group -- test infrastructure for the CPU-only,
world-size-2 tests of the restart shard.  The product package has no PyTorch in it."""
import numpy as np


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class GlooComm:
    """Same interface over an initialised torch.distributed process group (CPU tests)."""

    def __init__(self):
        import torch.distributed as dist
        self._dist = dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()

    def allgather(self, local):
        import torch
        t = torch.from_numpy(_f64(local).ravel().copy())
        outs = [torch.empty_like(t) for _ in range(self.world)]
        self._dist.all_gather(outs, t)
        return torch.cat(outs).numpy()

    def bcast(self, arr, root):
        import torch
        t = torch.from_numpy(_f64(arr).copy())
        self._dist.broadcast(t, src=int(root))
        return t.numpy()

    def barrier(self):
        self._dist.barrier()



def torch_store_exchange():
    """unique-id exchange through an initialised torch.distributed group (any backend).
    Only for processes that use torch on the CPU alone (see socket_exchange)."""
    import torch.distributed as dist

    def exchange(raw):
        box = [raw]
        dist.broadcast_object_list(box, src=0)
        return box[0]
    return exchange


