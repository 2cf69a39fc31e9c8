"""Multi-GPU sharding of the (scene-batch x candidate-camera) outer product.

One process per GPU (torch.distributed; backend "nccl" == RCCL over xGMI on ROCm, "gloo" on CPU tests).
The reference scores every camera on one GPU and takes torch.max (testers/shapenet.py:172); here each
rank scores its shard and the only exchange is an all-gather of one (best_gain fp32, global_cam_idx)
pair per cloud per rank — 8 B per cloud, latency-bound, so it uses a persistent buffer and a single
all_gather_into_tensor (no ring all-reduce).
"""
import torch
import torch.distributed as dist


def shard_range(n_items, rank, world):
    """Contiguous block partition of n_items over world ranks (first ranks get the remainder)."""
    q, r = divmod(n_items, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


_bufs = {}


def allgather_argmax(best_vals, best_global_idx, group=None):
    """best_vals [B] fp32, best_global_idx [B] int64 (this rank's best camera per cloud, GLOBAL index).
    Returns (vals [B], idx [B]) of the global arg-max, identical on every rank.
    Ties -> lowest camera index (torch.max's first-occurrence rule, testers/shapenet.py:172)."""
    world = dist.get_world_size(group)
    B = best_vals.shape[0]
    key = (best_vals.device, B, world)
    if key not in _bufs:
        _bufs[key] = (torch.empty(B, 2, dtype=torch.float32, device=best_vals.device),
                      torch.empty(world, B, 2, dtype=torch.float32, device=best_vals.device))
    send, recv = _bufs[key]
    send[:, 0] = best_vals
    # camera indices < 2^24 are exact in fp32: one 8-byte record per cloud, one collective
    send[:, 1] = best_global_idx.to(torch.float32)
    dist.all_gather_into_tensor(recv.view(-1), send.view(-1), group=group)
    vals = recv[:, :, 0]                      # [world, B]
    idx = recv[:, :, 1]
    vmax = vals.max(dim=0).values             # [B]
    cand = torch.where(vals == vmax[None], idx, torch.full_like(idx, float("inf")))
    best_idx = cand.min(dim=0).values
    return vmax, best_idx.to(torch.int64)


def allgather_rows(local, n_total, group=None):
    """Concatenate the ranks' row blocks (block-partitioned with shard_range) into the full [n_total, ...] tensor.
    Shards may differ by one row, so every rank pads to the largest shard for one all_gather_into_tensor."""
    world = dist.get_world_size(group)
    sizes = [shard_range(n_total, r, world) for r in range(world)]
    m = max(b - a for a, b in sizes)
    tail = local.shape[1:]
    send = local.new_zeros((m,) + tuple(tail))
    send[:local.shape[0]] = local
    recv = local.new_empty((world * m,) + tuple(tail))
    dist.all_gather_into_tensor(recv, send, group=group)
    return torch.cat([recv[r * m: r * m + (b - a)] for r, (a, b) in enumerate(sizes)], dim=0)
