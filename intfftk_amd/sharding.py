"""Batch sharding across the GPUs of one node: one process per GPU, frames are independent units
(int_fftNk keeps per-frame state only), so the transform itself needs NO collective.
The only data movement is an optional scatter of a root-resident batch before and a gather after,
done with point-to-point sends over torch.distributed (backend "nccl" = RCCL over xGMI on the GPUs;
"gloo" in the CPU tests): the root drives all of its 7 xGMI links concurrently, there is no ring
and no reduction anywhere.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple


def shard_bounds(batch: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous [start, stop) frame ranges per rank; the remainder goes to the LAST ranks."""
    if world <= 0 or batch < 0:
        raise ValueError("bad batch/world")
    base, rem = divmod(batch, world)
    out, start = [], 0
    for r in range(world):
        n = base + (1 if r >= world - rem else 0)
        out.append((start, start + n))
        start += n
    return out


class ShardedTransform:
    """scatter -> per-rank transform -> gather.  `transform` maps a local [frames, N, 2] tensor to
    the local result (an intfftk_amd.IntFFTCore on a GPU rank)."""

    def __init__(self, transform: Callable, n: int, in_dtype, out_dtype, device, group=None):
        import torch.distributed as dist

        self.dist = dist
        self.transform = transform
        self.n = n
        self.in_dtype, self.out_dtype = in_dtype, out_dtype
        self.device = device
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)

    # -- data movement ------------------------------------------------------------------------
    def scatter(self, root_batch, batch: int, root: int = 0):
        """root_batch: [batch, N, 2] on the root (ignored elsewhere) -> this rank's shard."""
        import torch

        dist = self.dist
        bounds = shard_bounds(batch, self.world)
        lo, hi = bounds[self.rank]
        if self.rank == root:
            reqs = []
            for r, (a, b) in enumerate(bounds):
                if r != root and b > a:
                    reqs.append(dist.isend(root_batch[a:b].contiguous(), dst=r, group=self.group))
            local = root_batch[lo:hi].clone()  # the root keeps its shard by local copy
            for q in reqs:
                q.wait()
            return local
        local = torch.empty((hi - lo, self.n, 2), dtype=self.in_dtype, device=self.device)
        if hi > lo:
            dist.recv(local, src=root, group=self.group)
        return local

    def gather(self, local, batch: int, root: int = 0):
        """This rank's result shard -> [batch, N, 2] on the root (None elsewhere)."""
        import torch

        dist = self.dist
        bounds = shard_bounds(batch, self.world)
        lo, hi = bounds[self.rank]
        if self.rank != root:
            if hi > lo:
                dist.send(local.contiguous(), dst=root, group=self.group)
            return None
        out = torch.empty((batch, self.n, 2), dtype=self.out_dtype, device=self.device)
        reqs = []
        for r, (a, b) in enumerate(bounds):
            if r != root and b > a:
                reqs.append(dist.irecv(out[a:b], src=r, group=self.group))
        out[lo:hi] = local
        for q in reqs:
            q.wait()
        return out

    # -- the three ways to run --------------------------------------------------------------
    def run_resident(self, local):
        """Data-resident sharding (the bench's mode): every rank transforms its own shard."""
        return self.transform(local)

    def run_from_root(self, root_batch, batch: int, root: int = 0):
        """End-to-end: scatter from the root, transform, gather back to the root."""
        local = self.scatter(root_batch, batch, root)
        res = self.transform(local) if local.shape[0] else local.new_empty((0, self.n, 2), dtype=self.out_dtype)
        return self.gather(res, batch, root)


def max_over_ranks(seconds: float, device=None, group=None) -> float:
    """The bench's timing rule: a step is as slow as the slowest rank."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
