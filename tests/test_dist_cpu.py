"""world_size-2 gloo tests (CPU) of the multi-GPU exchange logic: shard ranges, the all-gather of per-rank
(best gain, camera index) records and the all-gather of occupancy rows."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from macarons_amd import dist as mdist
    try:
        # camera shards: every rank scores its block of a known gain vector
        gains_all = torch.tensor([[0.2, 0.9, 0.1, 0.9, 0.3, 0.05, 0.7], [0.5, 0.1, 0.1, 0.2, 0.8, 0.8, 0.0]])   # [B=2, C=7]
        c0, c1 = mdist.shard_range(7, rank, world)
        local = gains_all[:, c0:c1]
        best = torch.max(local, dim=1)
        v, i = mdist.allgather_argmax(best.values, best.indices + c0)
        ref = torch.max(gains_all, dim=1)                    # first occurrence wins ties, like the reference
        ok1 = torch.equal(v, ref.values) and torch.equal(i, ref.indices)
        v2, i2 = mdist.allgather_best(local.contiguous(), c0)      # the entry nbv_step uses (HIP kernels on a GPU, torch on CPU)
        ok1 = ok1 and torch.equal(v2, ref.values) and torch.equal(i2, ref.indices)
        # occupancy rows: uneven shards
        full = torch.arange(11 * 3, dtype=torch.float32).view(11, 3)
        q0, q1 = mdist.shard_range(11, rank, world)
        got = mdist.allgather_rows(full[q0:q1].clone(), 11)
        ok2 = torch.equal(got, full)
        # fewer items than ranks: rank 1 owns an empty shard and still joins both collectives (no hang, right answer)
        one = torch.tensor([[0.25]])
        e0, e1 = mdist.shard_range(1, rank, world)
        v3, i3 = mdist.allgather_best(one[:, e0:e1].contiguous(), e0)
        ok3 = float(v3[0]) == 0.25 and int(i3[0]) == 0
        rows = mdist.allgather_rows(torch.full((e1 - e0, 2), 7.0), 1)
        ok3 = ok3 and rows.shape == (1, 2) and bool((rows == 7.0).all())
        # a row of -inf / NaN keeps torch.max's answer (first index / first NaN)
        bad = torch.tensor([[float("-inf")] * 4, [1.0, float("nan"), 3.0, float("nan")]])
        b0, b1 = mdist.shard_range(4, rank, world)
        v4, i4 = mdist.allgather_best(bad[:, b0:b1].contiguous(), b0)
        rb = torch.max(bad, dim=1)
        ok3 = ok3 and int(i4[0]) == int(rb.indices[0]) and int(i4[1]) == int(rb.indices[1]) and bool(torch.isnan(v4[1]))
        # hidden draws: every rank draws its own randperms / uniforms, rank 0's reach everybody in one broadcast (odd and even counts)
        g = torch.Generator().manual_seed(100 + rank)
        perms = [torch.randperm(50, generator=g)[:20], torch.randperm(7, generator=g), torch.randperm(9, generator=g)[:3]]
        g0 = torch.Generator().manual_seed(100)
        want = [torch.randperm(50, generator=g0)[:20], torch.randperm(7, generator=g0), torch.randperm(9, generator=g0)[:3]]
        for n_u in (8, 5):
            u = torch.rand(n_u, 1, generator=torch.Generator().manual_seed(7 + rank))
            u0 = torch.rand(n_u, 1, generator=torch.Generator().manual_seed(7))
            got_p, got_u = mdist.broadcast_draws(perms, u)
            ok3 = ok3 and all(torch.equal(a, b) for a, b in zip(got_p, want)) and torch.equal(got_u, u0) and got_u.shape == (n_u, 1)
        only_p, none_u = mdist.broadcast_draws(perms, None)
        ok3 = ok3 and none_u is None and all(torch.equal(a, b) for a, b in zip(only_p, want))
        q.put((rank, bool(ok1), bool(ok2 and ok3), (c0, c1), (q0, q1)))
    finally:
        dist.destroy_process_group()


def test_shard_range_partition():
    from macarons_amd.dist import shard_range
    for n in (1, 7, 8, 200, 512, 100_000):
        for w in (1, 2, 3, 8):
            r = [shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n and all(r[k][1] == r[k + 1][0] for k in range(w - 1))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.timeout(120)
def test_allgather_argmax_and_rows_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=90) for _ in range(2)]
    for p in procs:
        p.join(30)
    assert all(r[1] and r[2] for r in res), res
