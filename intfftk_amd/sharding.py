"""Batch sharding across the GPUs of one node: one process per GPU, frames are independent units
(int_fftNk keeps per-frame state only), so the transform itself needs NO collective.
The only data movement is an optional scatter of a root-resident batch before and a gather after.  Each is
ONE group of point-to-point operations (`torch.distributed.batch_isend_irecv`; on the "nccl" backend = RCCL that
is ncclGroupStart, 7 x ncclSend / ncclRecv, ncclGroupEnd), so the root drives all of its 7 xGMI links
concurrently (xGMI is point to point: 7 links x ~153 GB/s per GPU); there is no ring and no reduction anywhere.
"gloo" carries the same code in the CPU tests.
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
    the local result (an intfftk_amd.IntFFTCore on a GPU rank).

    stage_via_cpu: move the point-to-point payloads through host memory (gloo cannot send device tensors;
    only the single-GPU diagnostics mode of bench.py needs this -- RCCL sends device memory directly)."""

    def __init__(self, transform: Callable, n: int, in_dtype, out_dtype, device, group=None, stage_via_cpu: bool = False,
                 out_sample_shape: Optional[Tuple[int, ...]] = None):
        import torch.distributed as dist

        self.dist = dist
        self.transform = transform
        self.n = n
        self.in_dtype, self.out_dtype = in_dtype, out_dtype
        # trailing shape of one result frame: [N, 2] containers, or [N, 2, 2] int64 words for results beyond 64 bits
        # (IntFFTCore.out_shape); taken from the transform when it has one
        if out_sample_shape is None and hasattr(transform, "out_shape"):
            out_sample_shape = tuple(transform.out_shape(1))[1:]
        self.out_sample_shape = tuple(out_sample_shape) if out_sample_shape is not None else (n, 2)
        self.device = device
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.stage_via_cpu = stage_via_cpu
        self.last_group_sizes: List[int] = []  # p2p operations per issued group (tests / bench introspection)

    # -- one group of point-to-point operations ---------------------------------------------------------
    def _run_group(self, ops):
        """ops: list of (kind, tensor, peer).  Issues them as ONE batch_isend_irecv group and waits."""
        dist = self.dist
        self.last_group_sizes.append(len(ops))
        if not ops:
            return
        for req in self._issue_group(ops, count=False):
            req.wait()

    def _issue_group(self, ops, count=True):
        """Issues ONE batch_isend_irecv group and returns its requests without waiting (what runs between issue and wait overlaps the
        transfers: the root's own transform in the pipelined schedule)."""
        dist = self.dist
        if count:
            self.last_group_sizes.append(len(ops))
        if not ops:
            return []
        p2p = [dist.P2POp(dist.isend if kind == "send" else dist.irecv, t, peer, self.group) for kind, t, peer in ops]
        return dist.batch_isend_irecv(p2p)

    # -- data movement ------------------------------------------------------------------------
    def scatter(self, root_batch, batch: int, root: int = 0):
        """root_batch: [batch, N, 2] on the root (ignored elsewhere) -> this rank's shard.
        ALIASING: on the root the returned shard is a VIEW of root_batch[lo:hi] (no copy); on every other rank it is a fresh buffer.  A
        caller that goes on to mutate root_batch, or runs an in-place transform on the shard, must `.clone()` the root's shard first."""
        import torch

        bounds = shard_bounds(batch, self.world)
        lo, hi = bounds[self.rank]
        if self.rank == root:
            ops = []
            for r, (a, b) in enumerate(bounds):
                if r != root and b > a:
                    piece = root_batch[a:b]  # contiguous rows of a contiguous batch: no copy
                    if self.stage_via_cpu:
                        piece = piece.cpu()
                    ops.append(("send", piece.contiguous(), r))
            local = root_batch[lo:hi]  # the root keeps its shard where it is: a view of its rows, no copy
            self._run_group(ops)
            return local
        local = torch.empty((hi - lo, self.n, 2), dtype=self.in_dtype, device="cpu" if self.stage_via_cpu else self.device)
        self._run_group([("recv", local, root)] if hi > lo else [])
        return local.to(self.device) if self.stage_via_cpu else local

    def gather(self, local, batch: int, root: int = 0):
        """This rank's result shard -> [batch, *out_sample_shape] on the root (None elsewhere)."""
        import torch

        bounds = shard_bounds(batch, self.world)
        lo, hi = bounds[self.rank]
        if self.rank != root:
            if hi > lo:
                piece = local.cpu() if self.stage_via_cpu else local
                self._run_group([("send", piece.contiguous(), root)])
            else:
                self._run_group([])
            return None
        out = torch.empty((batch,) + self.out_sample_shape, dtype=self.out_dtype, device=self.device)
        ops, staged = [], []
        for r, (a, b) in enumerate(bounds):
            if r != root and b > a:
                if self.stage_via_cpu:
                    buf = torch.empty((b - a,) + self.out_sample_shape, dtype=self.out_dtype, device="cpu")
                    staged.append((a, b, buf))
                    ops.append(("recv", buf, r))
                else:
                    ops.append(("recv", out[a:b], r))  # received straight into its rows of the result
        out[lo:hi] = local
        self._run_group(ops)
        for a, b, buf in staged:
            out[a:b] = buf.to(self.device)
        return out

    # -- the three ways to run --------------------------------------------------------------
    def run_resident(self, local):
        """Data-resident sharding (the bench's mode): every rank transforms its own shard."""
        return self.transform(local)

    def run_from_root(self, root_batch, batch: int, root: int = 0):
        """End-to-end: scatter from the root, transform, gather back to the root."""
        local = self.scatter(root_batch, batch, root)
        res = self.transform(local) if local.shape[0] else local.new_empty((0,) + self.out_sample_shape, dtype=self.out_dtype)
        return self.gather(res, batch, root)


    def run_from_root_pipelined(self, root_batch, batch: int, root: int = 0, pieces: int = 4):
        """End-to-end as a pipeline (this path is link-bound, and xGMI links are full duplex): every shard is cut into
        `pieces` pieces and step t = 0 .. pieces is ONE point-to-point group holding the scatter of piece t AND the gather of piece t - 1
        (on RCCL one ncclGroupStart .. ncclGroupEnd: the root sends to and receives from every peer at once), after which the peers
        transform piece t.  The root transforms its own shard while the first group is in flight.  pieces + 1 groups of <= 2 (world - 1)
        operations on the root instead of 2 groups; the same rows as run_from_root."""
        import torch

        if pieces < 1:
            raise ValueError("pieces must be >= 1")
        bounds = shard_bounds(batch, self.world)
        lo, hi = bounds[self.rank]
        dev = "cpu" if self.stage_via_cpu else self.device

        def piece(r, k):  # [start, stop) of piece k of rank r's shard, in rows of the whole batch
            a, b = bounds[r]
            pa, pb = shard_bounds(b - a, pieces)[k]
            return a + pa, a + pb

        if self.rank == root:
            out = torch.empty((batch,) + self.out_sample_shape, dtype=self.out_dtype, device=self.device)
            staged = []
            for t in range(pieces + 1):
                ops, keep = [], []
                for r in range(self.world):
                    if r == root:
                        continue
                    if t < pieces:
                        a, b = piece(r, t)
                        if b > a:
                            src = root_batch[a:b].cpu() if self.stage_via_cpu else root_batch[a:b]
                            keep.append(src)
                            ops.append(("send", src.contiguous(), r))
                    if t >= 1:
                        a, b = piece(r, t - 1)
                        if b > a:
                            if self.stage_via_cpu:
                                buf = torch.empty((b - a,) + self.out_sample_shape, dtype=self.out_dtype, device="cpu")
                                staged.append((a, b, buf))
                                ops.append(("recv", buf, r))
                            else:
                                ops.append(("recv", out[a:b], r))
                reqs = self._issue_group(ops)
                if t == 0 and hi > lo:  # the root's own shard, beside the first group
                    out[lo:hi] = self.transform(root_batch[lo:hi])
                for req in reqs:
                    req.wait()
            for a, b, buf in staged:
                out[a:b] = buf.to(self.device)
            return out
        local = torch.empty((hi - lo, self.n, 2), dtype=self.in_dtype, device=dev)
        res = torch.empty((hi - lo,) + self.out_sample_shape, dtype=self.out_dtype, device=dev)
        for t in range(pieces + 1):
            ops = []
            if t < pieces:
                a, b = piece(self.rank, t)
                if b > a:
                    ops.append(("recv", local[a - lo:b - lo], root))
            if t >= 1:
                a, b = piece(self.rank, t - 1)
                if b > a:
                    ops.append(("send", res[a - lo:b - lo], root))
            for req in self._issue_group(ops):
                req.wait()
            if t < pieces:
                a, b = piece(self.rank, t)
                if b > a:
                    x = local[a - lo:b - lo]
                    y = self.transform(x.to(self.device) if self.stage_via_cpu else x)
                    res[a - lo:b - lo] = y.cpu() if self.stage_via_cpu else y
        return None


def max_over_ranks(seconds: float, device=None, group=None) -> float:
    """The bench's timing rule: a step is as slow as the slowest rank."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


def gather_floats(value: float, device=None, group=None) -> List[float]:
    """Every rank's `value`, in rank order (per-GPU figures of the bench line)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return [value]
    t = torch.tensor([value], dtype=torch.float64, device=device)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size(group))]
    dist.all_gather(out, t, group=group)
    return [float(o.item()) for o in out]
