"""Multi-GPU plumbing: contiguous site sharding + the one all-reduce per EP
iteration.  torch.distributed is used as transport only
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).
"""

import numpy as np


def site_range(K, rank, world):
    """Contiguous block of sites owned by `rank`."""
    return (K * rank) // world, (K * (rank + 1)) // world


class LocalComm(object):
    """Single process: every collective is the identity."""
    rank = 0
    world = 1

    def allreduce_sum(self, x):
        return x

    def allreduce_min_int(self, v):
        return int(v)

    def allreduce_max(self, arr):
        return np.asarray(arr, dtype=np.float64)

    def allgather_sites(self, local, K):
        return local

    def barrier(self):
        pass


class TorchComm(object):
    """torch.distributed process group wrapper (already initialised by the caller)."""

    def __init__(self, group=None, device=None):
        import torch
        import torch.distributed as dist
        self.torch = torch
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.device = device      # torch.device of this rank's GPU (None on CPU/gloo)

    def _small(self, arr):
        t = self.torch.as_tensor(np.ascontiguousarray(arr))
        if self.device is not None:
            t = t.to(self.device)
        return t

    def allreduce_sum(self, x):
        """x: CUDA torch tensor (reduced in place over RCCL) or NumPy array."""
        if isinstance(x, np.ndarray):
            t = self._small(x)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
            x[...] = t.cpu().numpy()
            return x
        self.dist.all_reduce(x, op=self.dist.ReduceOp.SUM, group=self.group)
        return x

    def allreduce_min_int(self, v):
        t = self._small(np.array([int(v)], dtype=np.int64))
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN, group=self.group)
        return int(t.cpu()[0])

    def allreduce_max(self, arr):
        t = self._small(np.asarray(arr, dtype=np.float64))
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
        return t.cpu().numpy()

    def allgather_sites(self, local, K):
        """local: F-ordered (..., K_local) slice; returns the full (..., K) array."""
        lead = local.shape[:-1]
        per = int(np.prod(lead)) if lead else 1
        counts = [site_range(K, r, self.world) for r in range(self.world)]
        kmax = max(hi - lo for lo, hi in counts)
        buf = np.zeros(per * kmax)
        flat = np.asarray(local).reshape(-1, order='F')
        buf[:flat.shape[0]] = flat
        t = self._small(buf)
        outs = [self.torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(outs, t, group=self.group)
        full = np.empty(lead + (K,), order='F')
        for r, (lo, hi) in enumerate(counts):
            part = outs[r].cpu().numpy()[:per * (hi - lo)]
            full[..., lo:hi] = part.reshape(lead + (hi - lo,), order='F')
        return full

    def barrier(self):
        self.dist.barrier(group=self.group)
