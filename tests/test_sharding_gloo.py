"""N > 1 path on CPU: world_size-2 (and 3) `gloo` process groups exercising the batch sharding,
scatter/gather plumbing and the bench's max-over-ranks timing rule.  The per-rank transform is a
stand-in (the oracle) because the product has no CPU execution path."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from intfftk_amd.sharding import ShardedTransform, max_over_ranks, shard_bounds


def test_shard_bounds_partition():
    for batch in (0, 1, 7, 8, 65536, 131072 + 5):
        for world in (1, 2, 3, 8):
            b = shard_bounds(batch, world)
            assert b[0][0] == 0 and b[-1][1] == batch
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1
            assert sizes == sorted(sizes)  # remainder goes to the last ranks


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, batch, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle_c as C

        log2n, n = 6, 64
        p = C.make_params(log2n, 16, 16, 0, 0, True)

        def transform(x):  # stand-in for IntFFTCore on a GPU rank
            return torch.from_numpy(C.execute_i16(x.numpy(), p, C.FWD))

        sh = ShardedTransform(transform, n, torch.int16, torch.int16, torch.device("cpu"))
        rng = np.random.default_rng(5)
        full = torch.from_numpy(rng.integers(-2 ** 14, 2 ** 14, size=(batch, n, 2)).astype(np.int16))
        out = sh.run_from_root(full if rank == 0 else None, batch, root=0)
        ok = True
        # scatter and gather are ONE point-to-point group each (on RCCL: ncclGroupStart .. ncclGroupEnd), holding one
        # operation per non-empty peer shard on the root and one operation on every other rank with a shard
        peers = sum(1 for r, (a, b) in enumerate(shard_bounds(batch, world)) if r != 0 and b > a)
        lo0, hi0 = shard_bounds(batch, world)[rank]
        want_ops = peers if rank == 0 else (1 if hi0 > lo0 else 0)
        ok = ok and sh.last_group_sizes == [want_ops, want_ops]
        if rank == 0:
            want = torch.from_numpy(C.execute_i16(full.numpy(), p, C.FWD))
            ok = torch.equal(out, want)
        else:
            ok = out is None
        # the pipelined schedule: pieces + 1 groups; step t holds the scatter of piece t and the gather of piece t - 1
        for pieces in (1, 3, 4):
            sh.last_group_sizes.clear()
            outp = sh.run_from_root_pipelined(full if rank == 0 else None, batch, root=0, pieces=pieces)
            ok = ok and (torch.equal(outp, want) if rank == 0 else outp is None)
            b_all = shard_bounds(batch, world)

            def nonempty(r, k):
                a, b = b_all[r]
                pa, pb = shard_bounds(b - a, pieces)[k]
                return pb > pa

            sizes = []
            for t in range(pieces + 1):
                who = [r for r in range(world) if r != 0] if rank == 0 else [rank]
                sizes.append(sum(1 for r in who if t < pieces and nonempty(r, t)) + sum(1 for r in who if t >= 1 and nonempty(r, t - 1)))
            ok = ok and sh.last_group_sizes == sizes and len(sizes) == pieces + 1
        # data-resident mode: each rank transforms its own shard; rows equal the unsharded result
        lo, hi = shard_bounds(batch, world)[rank]
        loc = sh.run_resident(full[lo:hi])
        ok = ok and torch.equal(loc, torch.from_numpy(C.execute_i16(full.numpy(), p, C.FWD))[lo:hi])
        # timing rule: max over ranks
        t = max_over_ranks(0.5 + rank, torch.device("cpu"))
        ok = ok and abs(t - (0.5 + world - 1)) < 1e-12
        q.put((rank, bool(ok)))
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("world,batch", [(2, 10), (2, 7), (3, 8), (2, 1)])
def test_scatter_transform_gather_gloo(world, batch):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, batch, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(r, True) for r in range(world)]
