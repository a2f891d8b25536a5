"""The gloo communicator of the world-size-2 CPU tests: host-array collectives over an initialised torch.distributed
group, with the surface of trtools_amd.dist.RcclComm / SocketGroup (test infrastructure: the package itself has no
torch in it)."""
import numpy as np


class TorchComm:
    """Host-array collectives over torch.distributed (any backend that handles CPU tensors)."""

    def __init__(self):
        import torch.distributed as dist
        self.dist = dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()

    def allreduce_sum_i64(self, arr):
        import torch
        t = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.int64).copy())
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return t.numpy()

    def allgather_bytes(self, arr):
        """Gather variable-length uint8 payloads; returns the list in rank order."""
        import torch
        n = torch.tensor([arr.size], dtype=torch.int64)
        sizes = [torch.zeros(1, dtype=torch.int64) for _ in range(self.world)]
        self.dist.all_gather(sizes, n)
        m = int(max(int(s[0]) for s in sizes))
        buf = torch.zeros(m, dtype=torch.uint8)
        buf[:arr.size] = torch.from_numpy(np.array(arr, dtype=np.uint8).reshape(-1))
        outs = [torch.zeros(m, dtype=torch.uint8) for _ in range(self.world)]
        self.dist.all_gather(outs, buf)
        return [o.numpy()[:int(s[0])] for o, s in zip(outs, sizes)]
