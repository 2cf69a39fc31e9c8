"""Multi-GPU sharding of the (scene-batch x candidate-camera) outer product.

One process per GPU (torch.distributed; backend "nccl" == RCCL over xGMI on ROCm, "gloo" on CPU tests).
The reference scores every camera on one GPU and takes torch.max (testers/shapenet.py:172); here each
rank scores its shard and the only exchange is an all-gather of one (best_gain fp32, global_cam_idx)
pair per cloud per rank — 8 B per cloud, latency-bound, so it uses a persistent buffer and a single
all_gather_into_tensor (no ring all-reduce).
"""
import torch
import torch.distributed as dist


def group_world_rank(group):
    """(world, rank) of a sharded / replicated call.  `group=None` means LOCAL -- (1, 0) -- even when a default process group
    exists: a data-parallel job whose ranks each own a different scene (how upstream's DDP trainers call these routines) must not
    find an implicit collective inside a method that used to be local.  Pass `torch.distributed.group.WORLD` (or a sub-group)
    to shard over it."""
    if group is None or not (dist.is_available() and dist.is_initialized()):
        return 1, 0
    return dist.get_world_size(group), dist.get_rank(group)


def exchange_on(group):
    """True when the exchange path of a sharded routine (broadcast of rank 0's draws, all-gathers, record merge) is to run: a group of
    more than one rank -- or, with env MCR_FORCE_DIST_PATH set, any explicitly passed group, so that a one-GPU box runs every
    collective of every sharded leg through RCCL on a one-rank group (tests/test_nbv_gpu.py)."""
    import os
    if group is None or not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or bool(os.environ.get("MCR_FORCE_DIST_PATH"))


def shard_range(n_items, rank, world):
    """Contiguous block partition of n_items over world ranks (first ranks get the remainder)."""
    q, r = divmod(n_items, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def _host_staged(t, group):
    """gloo has no device transport worth relying on: when the group's backend is gloo and the tensor lives on a HIP device
    (world-size-2 tests of the sharded step with two processes on ONE GPU -- RCCL refuses duplicate devices), the collective
    runs on a host copy.  RCCL groups (the product path) never take this branch."""
    return t.is_cuda and dist.get_backend(group) == "gloo"


def all_gather_into(recv, send, group=None):
    """dist.all_gather_into_tensor(recv, send) on flat tensors, host-staged for (gloo, device tensor)."""
    if _host_staged(send, group):
        r = torch.empty(recv.shape, dtype=recv.dtype)
        dist.all_gather_into_tensor(r, send.cpu(), group=group)
        recv.copy_(r)
    else:
        dist.all_gather_into_tensor(recv, send, group=group)


def broadcast(buf, src, group=None):
    """dist.broadcast(buf, src) (`src` = rank inside `group`), host-staged for (gloo, device tensor)."""
    gsrc = dist.get_global_rank(group, src) if group is not None else src
    if _host_staged(buf, group):
        h = buf.cpu()
        dist.broadcast(h, gsrc, group=group)
        buf.copy_(h)
    else:
        dist.broadcast(buf, gsrc, group=group)


def all_reduce_max(t, group=None):
    """MAX all-reduce of a small tensor (the range-guard flag of the sharded step); returns the reduced tensor."""
    if _host_staged(t, group):
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.MAX, group=group)
        return h
    t = t.clone()
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return t


_bufs = {}


def allgather_argmax(best_vals, best_global_idx, group=None):
    """best_vals [B] fp32, best_global_idx [B] int64 (this rank's best camera per cloud, GLOBAL index).
    Returns (vals [B], idx [B]) of the global arg-max, identical on every rank.
    Ties -> lowest camera index (torch.max's first-occurrence rule, testers/shapenet.py:172)."""
    world = dist.get_world_size(group)
    B = best_vals.shape[0]
    key = (best_vals.device, B, world, id(group))
    if key not in _bufs:
        _bufs[key] = (torch.empty(B, 2, dtype=torch.float32, device=best_vals.device),
                      torch.empty(world, B, 2, dtype=torch.float32, device=best_vals.device))
    send, recv = _bufs[key]
    send[:, 0] = best_vals
    # camera indices < 2^24 are exact in fp32: one 8-byte record per cloud, one collective
    send[:, 1] = best_global_idx.to(torch.float32)
    all_gather_into(recv.view(-1), send.view(-1), group)
    vals = recv[:, :, 0]                      # [world, B]
    idx = recv[:, :, 1]
    vmax = vals.max(dim=0).values             # [B]  (NaN if any rank holds one, like torch.max over the full row)
    same = (vals == vmax[None]) | (torch.isnan(vals) & torch.isnan(vmax)[None])
    cand = torch.where(same, idx, torch.full_like(idx, float("inf")))
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
    all_gather_into(recv, send, group)
    return torch.cat([recv[r * m: r * m + (b - a)] for r, (a, b) in enumerate(sizes)], dim=0)


def allgather_best(gains, idx_offset, group=None):
    """gains [B, C_local] of this rank's camera shard (global index of column 0 = idx_offset) -> (max_gain [B], nbv_idx [B])
    over all ranks' shards, identical on every rank.  On a HIP device: one record kernel, one all-gather of 8 B per cloud,
    one merge kernel (ops.best_record / ops.best_merge); on CPU tensors (gloo tests of the host logic) plain torch."""
    B = gains.shape[0]
    if gains.shape[1] == 0:                       # empty camera shard (C < world): join the exchange with a record that cannot win
        send_v = torch.full((B,), float("-inf"), dtype=torch.float32, device=gains.device)
        if not gains.is_cuda:
            return allgather_argmax(send_v, torch.full((B,), 2 ** 24, dtype=torch.int64), group)
    elif not gains.is_cuda:
        best = torch.max(gains, dim=1)
        return allgather_argmax(best.values, best.indices + idx_offset, group)
    from . import ops
    world = dist.get_world_size(group)
    if gains.shape[1] == 0:
        send = torch.stack((send_v, torch.full_like(send_v, 3.0e38)), dim=1).contiguous()
    else:
        send = ops.best_record(gains, idx_offset)
    recv = torch.empty((world, B, 2), dtype=torch.float32, device=gains.device)
    all_gather_into(recv.view(-1), send.view(-1), group)
    return ops.best_merge(recv)


def broadcast_draws(int_tensors, uniforms=None, src=0, group=None):
    """Rank `src`'s hidden random draws reach every rank in ONE broadcast.  `int_tensors`: list of int64 tensors (SconeOcc's
    randperm index tensors), `uniforms`: optional float32 tensor (the sampling uniforms); every rank passes tensors of the same
    shapes (its own draws), the returned ones hold rank src's values.  The float32 values travel bit-cast inside the int64
    buffer (an odd count is padded by one word)."""
    parts = [t.reshape(-1) for t in int_tensors]
    if any(t.dtype != torch.int64 for t in parts):
        raise TypeError("broadcast_draws: index tensors must be int64")
    n_u = 0
    if uniforms is not None:
        u = uniforms.reshape(-1).to(torch.float32)
        n_u = u.numel()
        if n_u & 1:
            u = torch.cat((u, u.new_zeros(1)))
        parts.append(u.contiguous().view(torch.int64))
    if not parts:
        return [], None
    buf = torch.cat(parts)
    broadcast(buf, src, group)
    out, off = [], 0
    for t in int_tensors:
        out.append(buf[off:off + t.numel()].reshape(t.shape))
        off += t.numel()
    u_out = buf[off:].view(torch.float32)[:n_u].reshape(uniforms.shape) if uniforms is not None else None
    return out, u_out


class PipelinedBest:
    """Arg-max exchange for a stream of decisions, sized for xGMI: the record kernel of every decision runs on the scoring
    stream, and once `batch` decisions have accumulated ONE all-gather (8 B x batch x clouds per rank) and ONE merge run on a
    side stream under the scoring of the following decisions -- a per-decision collective costs more host and link latency
    than the 90 us scoring pass it follows.  `depth` batches may be in flight; every slot owns its buffers, so nothing is
    allocated (or freed across streams) inside the loop."""

    def __init__(self, B, device, group=None, batch=8, depth=3, producers=()):
        """producers (optional): the streams decisions are submitted from when there is more than one (independent decisions scored
        round-robin on several streams): a batch's exchange then waits for all of them, not only for the stream of the submit that
        filled it."""
        from . import ops
        self._ops, self.group, self.depth, self.batch, self.B = ops, group, depth, batch, B
        self.producers = tuple(producers)
        self.world = dist.get_world_size(group)
        self.comm = torch.cuda.Stream(device=device)
        mk = lambda *shape, dtype=torch.float32: torch.empty(shape, dtype=dtype, device=device)
        self.slots = [dict(send=mk(batch, B, 2), recv=mk(self.world, batch * B, 2), vals=mk(batch * B),
                           idx=mk(batch * B, dtype=torch.int64), ready=torch.cuda.Event(), done=torch.cuda.Event(),
                           used=False, fill=0, gen=0) for _ in range(depth)]
        self._cur = 0

    def submit(self, gains, idx_offset):
        """Queue the exchange for `gains` [B, C_local] (produced on the current stream); returns a handle for result()."""
        s = self.slots[self._cur]
        if s["used"] and (s["fill"] == 0 or self.producers):
            torch.cuda.current_stream(gains.device).wait_event(s["done"])   # its previous exchange has consumed the send buffer
            # (several producer streams: every one of them writes into the slot, so every submit waits)
        j = s["fill"]
        self._ops.best_record(gains, idx_offset, out=s["send"][j])
        s["fill"] = j + 1
        handle = (s, j, s["gen"])
        if s["fill"] == self.batch:
            self._exchange(s)
        return handle

    def _exchange(self, s):
        n = s["fill"]
        if n == 0:
            return
        dev = s["send"].device
        if n < self.batch:
            s["send"][n:].zero_()                            # a short last batch: defined bytes on the wire
        cur = torch.cuda.current_stream(dev)
        for p in self.producers:                             # records written on the other scoring streams
            if p != cur:
                cur.wait_stream(p)
        s["ready"].record(cur)
        with torch.cuda.stream(self.comm):
            self.comm.wait_event(s["ready"])
            all_gather_into(s["recv"].view(-1), s["send"].view(-1), self.group)
            self._ops.best_merge(s["recv"], out_vals=s["vals"], out_idx=s["idx"])
            s["done"].record(self.comm)
        s["used"], s["fill"] = True, 0
        s["gen"] += 1                                # handles of this batch stay valid until the slot is refilled `depth` batches later
        self._cur = (self._cur + 1) % self.depth

    def flush(self):
        """Exchange a partially filled batch (call after the last submit)."""
        self._exchange(self.slots[self._cur])

    def result(self, handle):
        """(max_gain [B], nbv_idx [B]) of one submitted decision; the current stream waits for its batch's exchange."""
        s, j, gen = handle
        if s["gen"] == gen:
            raise RuntimeError("PipelinedBest.result: the decision's batch has not been exchanged yet (call flush())")
        if s["gen"] != gen + 1:
            raise RuntimeError("PipelinedBest.result: stale handle -- its slot has been reused by a later batch "
                               "(read results within depth x batch submits)")
        torch.cuda.current_stream(s["vals"].device).wait_event(s["done"])
        return s["vals"][j * self.B:(j + 1) * self.B], s["idx"][j * self.B:(j + 1) * self.B]
