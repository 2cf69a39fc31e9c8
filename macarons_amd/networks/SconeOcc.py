"""Host-side mirror of macarons/networks/SconeOcc.py: the occupancy-probability network, running on the
MI355X through libmacarons_hip.so.  Same constructors, parameter names and tensor layouts as the reference
(XEmbedding :7, PCTransformer :45, SconeOcc :133, forward :250).

Hidden RNG: SconeOcc.forward draws torch.randperm on the CPU default generator three times
(SconeOcc.py:269 and :311 twice).  This class draws them on the host in the same order, or takes them
explicitly through `perms=` so tests can pin them.
"""
import numpy as np
import os
import torch
import torch.nn.functional as F          # noqa: F401
from torch import nn

from .. import autograd as A
from .. import ops, _lib
from .Attention import (Embedding, Encoder, FeedForward, MultiHeadSelfAttention, attention, knn_gather,   # noqa: F401
                        _f32c, _inference_only)     # the public ones are what upstream's `from .Attention import *` hands on
from ..utility.utils import get_knn_points   # noqa: F401  (SconeOcc.py:4)
from .packing import encoder_weight_planes, padded_weight_planes, weight_planes, RangeGuard, BlobCache, HeadPlaneCache, TableCache, param_key, invalidate as _invalidate_key, freeze as _freeze_key


class XEmbedding(nn.Module):
    """SconeOcc.py:7-42: 3 -> E/4 -> E/2 -> E with GELU after every layer."""

    def __init__(self, x_dim, x_embedding_dim, dropout=None, gelu=True):
        super().__init__()
        if not gelu or dropout is not None:
            raise NotImplementedError("HIP path implements the GELU / no-dropout configuration")
        self.linear1 = nn.Linear(x_dim, x_embedding_dim // 4)
        self.linear2 = nn.Linear(x_embedding_dim // 4, x_embedding_dim // 2)
        self.linear3 = nn.Linear(x_embedding_dim // 2, x_embedding_dim)
        self.non_linear1, self.non_linear2, self.non_linear3 = nn.GELU(), nn.GELU(), nn.GELU()
        self.dropout = None

    def forward(self, x):
        _inference_only(self, x)
        res = ops.linear(x, _f32c(self.linear1.weight), _f32c(self.linear1.bias), gelu=True)
        res = ops.linear(res, _f32c(self.linear2.weight), _f32c(self.linear2.bias), gelu=True)
        return ops.linear(res, _f32c(self.linear3.weight), _f32c(self.linear3.bias), gelu=True)


_NATIVE_RAGGED = []


def _native_ragged_tables():
    """Is torch.ops.macarons.ragged_tables there (the C++ extension built)?  MCR_NATIVE_RAGGED_TABLES=0: the numpy restatement (A/B, tests)."""
    if not _NATIVE_RAGGED:
        import os
        ok = os.environ.get("MCR_NATIVE_RAGGED_TABLES", "1") != "0"
        if ok:
            try:
                from .. import torch_ops  # noqa: F401
                ok = hasattr(torch.ops.macarons, "ragged_tables")
            except Exception:
                ok = False
        _NATIVE_RAGGED.append(ok)
    return _NATIVE_RAGGED[0]


class PCTransformer(nn.Module):
    """SconeOcc.py:45-130."""

    def __init__(self, seq_len, pts_dim=3, pts_embedding_dim=256, feature_dim=512, concatenate_input=True, n_code=2,
                 n_heads=4, FF=True, gelu=True, dropout=None):
        super().__init__()
        self.seq_len, self.pts_dim, self.pts_embedding_dim = seq_len, pts_dim, pts_embedding_dim
        self.n_code, self.n_heads, self.FF, self.gelu = n_code, n_heads, FF, gelu
        self.feature_dim = feature_dim
        self.dropout = dropout
        self.embedding = Embedding(input_dim=pts_dim, output_dim=pts_embedding_dim, dropout=None, gelu=gelu,
                                   global_feature=False, additional_feature_dim=0, concatenate_input=concatenate_input,
                                   k_for_knn=0)
        self.encoders = nn.ModuleList([Encoder(seq_len=seq_len, embedding_dim=pts_embedding_dim,
                                               qk_dim=pts_embedding_dim // 4, n_heads=n_heads, dropout=dropout, gelu=gelu,
                                               FF=FF) for _ in range(n_code)])
        self.norm = nn.LayerNorm(pts_embedding_dim)
        self.linear0 = nn.Linear(pts_embedding_dim, feature_dim // 2)

    def _is_default_arch(self):
        return (self.pts_dim == 3 and self.pts_embedding_dim == 128 and self.n_code == 2 and self.n_heads == 4
                and self.FF and self.embedding.concatenate_input and self.feature_dim in (256, 512))

    def weight_table(self):
        t = [_f32c(self.embedding.linear1.weight), _f32c(self.embedding.linear1.bias),
             _f32c(self.embedding.linear2.weight), _f32c(self.embedding.linear2.bias)]
        for e in self.encoders:
            t += e.weight_table()
        t += [_f32c(self.norm.weight), _f32c(self.norm.bias), _f32c(self.linear0.weight), _f32c(self.linear0.bias)]
        return t

    def forward(self, pc, mask=None):
        """pc [n_clouds, seq_len, 3] -> [n_clouds, feature_dim]."""
        _inference_only(self, pc)
        if not self._is_default_arch():
            raise NotImplementedError("the fused MI355X PCTransformer implements the reference's default architecture")
        if mask is not None:                            # SconeOcc.py:106-118: the mask goes to every encoder; block by block (see SconeVis.forward)
            x = self.embedding(pc)
            for enc in self.encoders:
                x = enc(x, mask=mask)
            x = ops.linear(ops.layernorm(x, _f32c(self.norm.weight), _f32c(self.norm.bias)), _f32c(self.linear0.weight), _f32c(self.linear0.bias))
            return ops.pool_max_avg(x)
        if pc.shape[1] >= 512:                          # long sequences: planes of every GEMM's weights, built once per parameter version
            if not hasattr(self, "_table_cache"):
                self._table_cache = TableCache()
            return ops.pc_transformer_forward(pc, self._table_cache.get(self, self.weight_table_with_planes), self.feature_dim)
        return ops.pc_transformer_forward(pc, self.weight_table(), self.feature_dim)

    def weight_table_with_planes(self):
        """weight_table() + the encoders' weights as fp16 hi/lo planes (8 blobs) + the end layers' (3): include/macarons_hip.h, PLANES / END PLANES."""
        t = self.weight_table()
        for e in self.encoders:
            t += encoder_weight_planes(e)
        t += list(padded_weight_planes(self.embedding.linear2.weight, self.embedding.linear2.bias, 128, 128))
        t += [weight_planes(self.linear0.weight)]
        return t


class SconeOcc(RangeGuard, nn.Module):
    def __init__(self, seq_len=2048, pts_dim=3, pts_embedding_dim=128, concatenate_input=True, n_code=2, n_heads=4,
                 FF=True, gelu=True, global_feature_dim=512, n_scale=3, local_feature_dim=256, k_for_knn=16, x_dim=3,
                 x_embedding_dim=512, n_harmonics=64, output_dim=1, dropout=None, offset=True):
        super().__init__()
        self.seq_len, self.pts_dim, self.pts_embedding_dim = seq_len, pts_dim, pts_embedding_dim
        self.n_code, self.n_heads, self.FF, self.gelu = n_code, n_heads, FF, gelu
        self.n_scale = n_scale
        self.x_dim, self.x_embedding_dim = x_dim, x_embedding_dim
        self.output_dim = output_dim
        self.dropout = dropout
        self.encoding_dim = pts_embedding_dim
        self.k_for_knn = k_for_knn
        self.offset = offset
        if self.offset:
            print("Offset set to True.")                                     # SconeOcc.py:199-200
        self.global_feature_dim, self.local_feature_dim = global_feature_dim, local_feature_dim
        self.n_harmonics = n_harmonics
        self.all_feature_size = x_embedding_dim + n_scale * local_feature_dim + global_feature_dim + n_harmonics
        self.global_transformer = PCTransformer(seq_len=seq_len, pts_dim=pts_dim, pts_embedding_dim=pts_embedding_dim,
                                                feature_dim=global_feature_dim, concatenate_input=concatenate_input,
                                                n_code=n_code, n_heads=n_heads, FF=FF, gelu=gelu, dropout=dropout)
        self.local_transformers = nn.ModuleList([
            PCTransformer(seq_len=k_for_knn, pts_dim=pts_dim, pts_embedding_dim=pts_embedding_dim,
                          feature_dim=local_feature_dim, concatenate_input=concatenate_input, n_code=n_code,
                          n_heads=n_heads, FF=FF, gelu=gelu, dropout=dropout) for _ in range(n_scale)])
        self.x_embedding = XEmbedding(x_dim=x_dim, x_embedding_dim=x_embedding_dim, dropout=dropout, gelu=gelu)
        self.linear1 = nn.Linear(self.all_feature_size, 512)
        self.linear2 = nn.Linear(512, 256)
        self.linear3 = nn.Linear(256, output_dim)
        self.non_linear1, self.non_linear2, self.non_linear3 = nn.GELU(), nn.GELU(), nn.GELU()
        # fused LDS-resident local transformers (local_pct.hip); set False to run them layer by layer
        self.fused_local = True
        self._blob_caches = [BlobCache() for _ in range(n_scale)]
        self._table_cache = TableCache()
        self._head_cache = HeadPlaneCache()
        self._key_cache = {}
        # Range guard of the default numerics (variant 6: matrix products on fp16 hi/lo planes, valid for |activation| < 65504;
        # the reference is plain fp32, Attention.py:98-128): the kernels flag a non-finite occupancy -- what an out-of-range
        # activation turns into -- and the forward is repeated on variant 5 (bf16 hi/mid/lo, the whole fp32 range).
        #   "sync"  (default: correct for a drop-in caller -- the reference never returns NaN here) read the flag after every forward
        #           (one 4-byte read-back = a pipeline drain) and repeat that forward at once on variant 5: the FIRST overflowed
        #           stand-alone forward already returns finite occupancies within the 1e-4 contract;
        #   "defer" leave it in range_flag() for the caller: nbv_step / macarons_nbv_decision switch to this for their duration and
        #           check ONE flag once, at the end of the decision (the performance paths never pay the per-forward read-back);
        #   "async" (opt-in) never stalls: after every forward the flag is copied to pinned host memory behind the kernels; the NEXT
        #           forward (or check_range()) looks at copies that have landed.  A set flag means an earlier forward returned
        #           non-finite occupancies (visible as such to the caller): it is reported once (RuntimeWarning) and every later
        #           forward of this module runs on variant 5 -- for a caller that keeps upstream's chunk loop and cannot afford a
        #           device->host round trip per chunk, and checks check_range(wait=True) at its own boundaries;
        #   "off"   no check.
        self.range_guard = "sync"
        self._range_flag = None
        self._range_pending = []            # (pinned host int32 [1], event) of forwards whose flag has not been looked at yet
        self._full_range = False            # True once an overflow was seen: variant 5 from then on

    def _effective_guard(self):
        """range_guard, except under stream capture: a read-back ("sync") or a host copy ("async") cannot be part of a graph, so a
        captured forward leaves the flag in range_flag() ("defer") for whoever replays the graph to look at."""
        g = self.range_guard
        if g in ("sync", "async") and torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
            return "defer"
        return g

    def freeze_weight_caches(self, on=True):
        """Inference mode: fingerprint the parameters once, now, and trust them unchanged until freeze_weight_caches(False) or
        invalidate_weight_caches() (an optimizer step or load_state_dict in between would go unnoticed: opt-in)."""
        _freeze_key(self, self._key_cache, on)

    def invalidate_weight_caches(self):
        """Drop every derived weight image (pointer table, packed local-transformer blobs, stacked QKV, split head planes): call
        after editing parameters in a way no fingerprint can see (in place through `p.data`, see packing._param_key)."""
        _invalidate_key(self._key_cache)                             # the shared per-forward fingerprint: every cache below rebuilds
        self._table_cache.invalidate()
        self._head_cache.invalidate()
        for c in self._blob_caches:
            c.invalidate()
        for m in self.modules():
            if hasattr(m, "_packed"):
                m._packed = None

    def _is_default_arch(self):
        return (self.n_scale == 3 and self.k_for_knn == 16 and self.offset and self.x_dim == 3 and self.x_embedding_dim == 512
                and self.global_feature_dim == 512 and self.local_feature_dim == 256 and self.n_harmonics == 64
                and self.output_dim == 1 and self.global_transformer._is_default_arch()
                and all(t._is_default_arch() for t in self.local_transformers))

    def weight_table(self):
        t = self.global_transformer.weight_table()
        for lt in self.local_transformers:
            t += lt.weight_table()
        for lin in (self.x_embedding.linear1, self.x_embedding.linear2, self.x_embedding.linear3, self.linear1,
                    self.linear2, self.linear3):
            t += [_f32c(lin.weight), _f32c(lin.bias)]
        return t

    def weight_table_with_planes(self):
        """weight_table() + the global transformer's encoder weights as fp16 hi/lo planes (8 blobs): see SconeVis.weight_table_with_planes."""
        t = self.weight_table()
        g = self.global_transformer
        for e in g.encoders:
            t += encoder_weight_planes(e)
        # the layers either side of the encoders (3 blobs): the embedding's second layer zero-padded to 128 x 128 (+ its padded bias), linear0
        t += list(padded_weight_planes(g.embedding.linear2.weight, g.embedding.linear2.bias, 128, 128))
        t += [weight_planes(g.linear0.weight)]
        return t

    def ds_factor(self, full_seq_len):
        """SconeOcc.py:281-288."""
        if self.n_scale > 1:
            ds = int(np.power(full_seq_len / (self.k_for_knn * 8), 1. / (self.n_scale - 1)))
            return 2 if ds == 0 else ds
        return 1

    def draw_perms(self, full_seq_len):
        """The three torch.randperm draws of one forward, in the reference's order on the CPU default generator:
        global down-sample (SconeOcc.py:269), then scale 0->1 and 1->2 (:311).  Returns index tensors (already cut)."""
        perms = [torch.randperm(full_seq_len)[:self.seq_len]]
        ds, m = self.ds_factor(full_seq_len), full_seq_len
        for _ in range(self.n_scale - 1):
            perms.append(torch.randperm(m)[:m // ds])
            m = m // ds
        return perms

    @staticmethod
    def _take(pc, p):
        """pc[:, p] for a shared 1-D index (SconeOcc.py:269, :311) or, extension, a per-cloud [n_clouds, n] index (a scene batch in
        which every object keeps the draws its own single-cloud call would have made)."""
        p = p.to(pc.device)
        if p.dim() == 1:
            return pc[:, p].contiguous()
        return torch.gather(pc, 1, p[..., None].expand(-1, -1, pc.shape[-1])).contiguous()

    def forward_ragged(self, pc, cloud_sizes, x, view_harmonics, query_sizes, perms=None, index_arrays=None, out=None):
        """J forward() calls of different sizes as ONE launch sequence (extension; the reference calls forward once per grid cell and
        chunk from a Python loop, macarons_utils.py:1395-1540).  Job j: surface cloud = the next cloud_sizes[j] rows of pc [sum M, 3],
        queries = the next query_sizes[j] rows of x [T,3] / view_harmonics [T,64] (host lists).  perms: per job the three index
        tensors draw_perms(cloud_sizes[j]) returns; None = drawn here in job order on the CPU generator, exactly the draws J
        sequential forward() calls would make; index_arrays: the uploaded index arrays of an earlier pass (`last_ragged_perms`: a
        caller that repeats a pass, or hands rank 0's draws to every rank).
        -> [T,1] (`out`: written there).  Inference only (no autograd graph).  = forward_ragged_begin + forward_ragged_finish."""
        h = self.forward_ragged_begin(pc, cloud_sizes, x, view_harmonics, query_sizes)
        return self.forward_ragged_finish(h, perms=perms, index_arrays=index_arrays, out=out)

    def forward_ragged_begin(self, pc, cloud_sizes, x, view_harmonics, query_sizes, row_job=None, arena="scone_occ_ragged"):
        """First half of forward_ragged: what needs no hidden draw is uploaded and LAUNCHED (phase 1: the x embedding, the scale-0 search
        and local transformer on the whole clouds), so that the GPU works while the host makes the ~3 J torch.randperm draws
        (forward_ragged_finish).  row_job (int32 device [T], optional): the job of every query row when the caller already holds it.
        arena: tag of the scratch arena that keeps phase 1's results until the finish (several begun passes need one each)."""
        if not self._is_default_arch() or not self.fused_local:
            raise NotImplementedError("forward_ragged implements the default architecture on the fused local-transformer path")
        dev = x.device
        J = len(cloud_sizes)
        L = _lib.lib()
        Lg = self.seq_len
        pc = pc.contiguous()
        variant = ops.current_variant()
        rows = int(L.mcr_knn_rows_per_block())
        if _native_ragged_tables():                                      # one C++ loop (this runs in front of the pass's first launch)
            host = torch.ops.macarons.ragged_tables([int(m) for m in cloud_sizes], [int(q) for q in query_sizes], rows, row_job is None)
            n_blk4 = host.numel() - (J + 1) - (sum(int(q) for q in query_sizes) if row_job is None else 0)
            off0 = host[:J + 1].numpy()
            early = ops.h2d(host, torch.int64, dev)
        else:
            off0 = np.concatenate(([0], np.cumsum(cloud_sizes))).astype(np.int64)
            qs = np.asarray(query_sizes, np.int64)
            q0 = np.concatenate(([0], np.cumsum(qs)))
            nb = -(-qs // rows)                                              # query blocks per job
            bj = np.repeat(np.arange(J, dtype=np.int64), nb)                 # job of every block
            b_in = np.arange(int(nb.sum()), dtype=np.int64) - np.repeat(np.cumsum(nb) - nb, nb)     # block index inside its job
            blocks = np.stack((bj, q0[bj] + b_in * rows, np.minimum(rows, qs[bj] - b_in * rows), np.zeros_like(bj)), 1)
            parts = [off0, blocks.reshape(-1)]
            if row_job is None:
                parts.append(np.repeat(np.arange(J, dtype=np.int64), qs))
            early = ops.h2d(np.concatenate(parts), torch.int64, dev)
            n_blk4 = blocks.size
        d_off0 = early[:J + 1]
        d_blocks = early[J + 1:J + 1 + n_blk4].to(torch.int32).view(-1, 4)
        d_row_job = early[J + 1 + n_blk4:].to(torch.int32) if row_job is None else row_job
        state = {}

        def caches(v):
            key = param_key(self, self._key_cache)                  # one fingerprint of the parameters for every derived image
            state[v] = ([c.get(t, v, key) for c, t in zip(self._blob_caches, self.local_transformers)],
                        self._head_cache.get(self, key) if v in (6, 7) else None, self._table_cache.get(self, self.weight_table_with_planes, key))
            return state[v]

        def phase1(v):
            blobs, head, table = state[v] if v in state else caches(v)
            ops.scone_occ_forward_ragged(None, None, [pc], [d_off0], x, view_harmonics, d_row_job, d_blocks, table, blobs, head, None,
                                         phase=1, Lg=Lg, arena=arena)

        with torch.no_grad():
            phase1(variant)
        return {"pc": pc, "x": x, "vh": view_harmonics, "J": J, "Lg": Lg, "variant": variant, "off0": off0, "d_off0": d_off0,
                "d_row_job": d_row_job, "d_blocks": d_blocks, "state": state, "caches": caches, "phase1": phase1,
                "cloud_sizes": cloud_sizes, "arena": arena, "epoch1": ops.scone_occ_epoch(dev, arena)}

    def forward_ragged_finish(self, h, perms=None, index_arrays=None, out=None):
        """Second half of forward_ragged: the hidden draws (unless given), the down-sampled clouds, phase 2."""
        L = _lib.lib()
        pc, x, view_harmonics, J, Lg, variant, off0 = h["pc"], h["x"], h["vh"], h["J"], h["Lg"], h["variant"], h["off0"]
        cloud_sizes, d_off0, d_row_job, d_blocks, state, caches, phase1 = (h["cloud_sizes"], h["d_off0"], h["d_row_job"], h["d_blocks"],
                                                                          h["state"], h["caches"], h["phase1"])
        dev = x.device
        if index_arrays is not None:
            ia = index_arrays
            self.last_ragged_perms = ia                                    # (a caller that repeats the pass on another variant re-uses the draws)
            pc_global = pc[ia["g_idx"]].view(J, Lg, 3)
            pc1 = pc[ia["idx1"]]
            pc2 = pc1[ia["idx2"]]
            g_len_d, d_off1, d_off2 = ia["g_len"], ia["off1"], ia["off2"]
        else:
            from ..utility.host import batched_draws_ok
            if perms is None and not batched_draws_ok():                 # torch.randperm was replaced from Python: honour it
                perms = [self.draw_perms(int(m)) for m in cloud_sizes]
            if perms is None:
                # the 3 J draws in job order on the CPU generator -- exactly the draws J sequential forward() calls make -- by ONE call
                # into the C++ extension (the same at::randperm calls; no dispatcher round trip and no numpy per draw), which returns
                # the index arrays below ready to upload
                from .. import torch_ops  # noqa: F401
                sz = [self.scale_sizes(int(m)) for m in cloud_sizes]
                flat = torch.ops.macarons.scone_occ_draws([s_[0] for s_ in sz], [s_[1] for s_ in sz], [s_[2] for s_ in sz], Lg)
                n1, n2 = sum(s_[1] for s_ in sz), sum(s_[2] for s_ in sz)
            else:
                # ---- index arrays of the down-sampled clouds, built on the host from the given draws
                p0s = [np.asarray(p[0], dtype=np.int64)[:Lg] for p in perms]
                p1s = [np.asarray(p[1], dtype=np.int64) for p in perms]
                p2s = [np.asarray(p[2], dtype=np.int64) for p in perms]
                n0 = np.asarray([len(p) for p in p0s], np.int64)
                n1a = np.asarray([len(p) for p in p1s], np.int64)
                n2a = np.asarray([len(p) for p in p2s], np.int64)
                off1 = np.concatenate(([0], np.cumsum(n1a)))
                off2 = np.concatenate(([0], np.cumsum(n2a)))
                g_idx = np.repeat(off0[:J], Lg).reshape(J, Lg)                # padding rows: any valid point (masked by global_len)
                col = np.arange(int(n0.sum()), dtype=np.int64) - np.repeat(np.cumsum(n0) - n0, n0)
                g_idx[np.repeat(np.arange(J), n0), col] = np.concatenate(p0s) + np.repeat(off0[:J], n0)
                idx1 = np.concatenate(p1s) + np.repeat(off0[:J], n1a)
                idx2 = np.concatenate(p2s) + np.repeat(off1[:J], n2a)          # scale 2 indexes scale 1's rows (SconeOcc.py:311)
                flat = np.concatenate([g_idx.reshape(-1), idx1, idx2, off1, off2, n0])
                n1, n2 = int(off1[-1]), int(off2[-1])
            d = ops.h2d(flat, torch.int64, dev)                            # ONE upload
            cut, o = [], 0
            for n in (J * Lg, n1, n2, J + 1, J + 1, J):
                cut.append(d[o:o + n]); o += n
            ia = {"g_idx": cut[0], "idx1": cut[1], "idx2": cut[2], "off1": cut[3], "off2": cut[4], "g_len": cut[5].to(torch.int32)}
            self.last_ragged_perms = ia                                    # (a caller that repeats the pass re-uses the draws: index_arrays=)
            pc_global = pc[ia["g_idx"]].view(J, Lg, 3)
            pc1 = pc[ia["idx1"]]
            pc2 = pc1[ia["idx2"]]
            g_len_d, d_off1, d_off2 = ia["g_len"], ia["off1"], ia["off2"]

        def run(v, flag, redo_phase1, out_):
            if redo_phase1:
                phase1(v)
            blobs, head, table = state[v] if v in state else caches(v)
            return ops.scone_occ_forward_ragged(pc_global, g_len_d, [pc, pc1, pc2], [d_off0, d_off1, d_off2], x, view_harmonics,
                                                d_row_job, d_blocks, table, blobs, head, flag, phase=2, out=out_, arena=h["arena"])
        flag = None
        guard = self._effective_guard()
        if variant in (6, 7) and guard != "off":
            if self._range_flag is None or self._range_flag.device != dev:
                self._range_flag = torch.zeros(1, dtype=torch.int32, device=dev)
            elif guard in ("sync", "async"):
                self._range_flag.zero_()
            flag = self._range_flag
        with torch.no_grad():
            # (phase 1's results live in the stream's arena: if anything else wrote it since -- another thread on this stream -- redo it)
            res = run(variant, flag, ops.scone_occ_epoch(dev, h["arena"]) != h["epoch1"], out)
            if flag is not None and guard == "sync" and int(flag):
                with ops.variant(5):
                    res = run(5, None, True, out)
            elif flag is not None and guard == "async":
                self._post_range_check(flag)
        return res

    def scale_sizes(self, full_seq_len):
        """Points of the three neighbourhood clouds of a forward on a cloud of full_seq_len points (SconeOcc.py:282-288, :311)."""
        ds, sizes = self.ds_factor(full_seq_len), [full_seq_len]
        for _ in range(self.n_scale - 1):
            sizes.append(sizes[-1] // ds)
        return sizes

    def _images(self, variant):
        key = param_key(self, self._key_cache)                      # one fingerprint of the parameters for every derived image
        blobs = [c.get(t, variant, key) for c, t in zip(self._blob_caches, self.local_transformers)] if self.fused_local else None
        head = self._head_cache.get(self, key) if variant in (6, 7) else None
        return self._table_cache.get(self, self.weight_table_with_planes, key), blobs, head

    def forward_begin(self, pc, x):
        """Extension: queue the part of forward(pc, x, ...) that needs neither the view harmonics nor the hidden draws (the query
        order of the neighbour search, scale 0: search + local transformer over the whole cloud) and return a handle for
        forward(..., begun=handle), which then runs the rest.  A caller that still has host work to do before it can call forward --
        view state, harmonics, down-sampled clouds (nbv_step) -- hides that work behind this first long kernel.  None (and nothing
        queued) when the split does not apply (gradients wanted, non-default architecture, layer-by-layer path)."""
        if not self._is_default_arch() or not self.fused_local or A.needs_grad(self, pc, x) or os.environ.get("MCR_OCC_BEGIN") == "0":
            return None
        variant = ops.current_variant()
        pc0, x_ = pc.contiguous(), x.contiguous()
        sizes = self.scale_sizes(pc.shape[1])
        if min(sizes) < self.k_for_knn:
            return None
        with torch.no_grad():
            table, blobs, head = self._images(variant)
            ops.scone_occ_forward(None, [pc0, None, None], x_, None, table, blobs, head, None, phase=1, M_scale=sizes, Lg=self.seq_len)
        return {"pc": pc0, "x": x_, "variant": variant, "sizes": sizes, "src": (pc, x), "epoch": ops.scone_occ_epoch(pc0.device, "scone_occ")}

    def forward(self, pc, x, view_harmonics, mask=None, verbose=False, perms=None, begun=None):
        """pc [n_clouds, M, 3], x [n_clouds, Q, 3], view_harmonics [n_clouds, Q, 64] -> [n_clouds, Q, 1].
        `perms` (optional): the three index tensors draw_perms() would return, to pin the hidden RNG; each either 1-D (shared by
        the clouds, as the reference draws them) or [n_clouds, n] (one draw per cloud).  `begun`: the handle of forward_begin(pc, x)
        (same tensors): only the remaining part runs."""
        if mask is not None:
            # upstream hands ONE mask to the 2048-token global transformer and to the 16-token local ones (SconeOcc.py:271, :301): no
            # shape fits both, so no caller can pass one; PCTransformer.forward(mask=...) itself is supported
            raise NotImplementedError("SconeOcc.forward(mask=...): upstream applies the same mask to sequences of 2048 and of 16 tokens")
        if not self._is_default_arch():
            raise NotImplementedError("the fused MI355X SconeOcc forward implements the reference's default architecture")
        n_clouds, full_seq_len = pc.shape[0], pc.shape[1]
        n_sample = x.shape[1]
        if perms is None:
            perms = self.draw_perms(full_seq_len)
        dev = pc.device
        L = _lib.lib()
        if self.range_guard == "async" and (self._range_pending or self._full_range):
            self.check_range()
        if self._full_range and ops.current_variant() in (6, 7):        # an earlier forward overflowed the fp16 split: full range from now on
            with ops.variant(5):
                return self.forward(pc, x, view_harmonics, mask, verbose, perms, None)
        variant = ops.current_variant()
        # A handle of forward_begin is valid only for the very tensors, numerics and cloud sizes it was queued with -- and only while
        # nothing else has written the arena that holds its phase-1 results (ops.scone_occ_epoch).  Checked BEFORE anything of the
        # handle is used: a stale or foreign handle is ignored and the whole forward runs on `pc` / `x`.
        sizes = self.scale_sizes(full_seq_len)
        if begun is not None and not (begun["src"][0] is pc and begun["src"][1] is x and begun["variant"] == variant
                                      and begun["sizes"] == sizes and [p_.shape[-1] for p_ in perms] == [self.seq_len] + sizes[1:]
                                      and begun["epoch"] == ops.scone_occ_epoch(dev, "scone_occ")):
            begun = None
        pc_global = self._take(pc, perms[0])                                          # SconeOcc.py:269
        scales = [begun["pc"] if begun is not None else pc.contiguous()]
        for p in perms[1:]:
            scales.append(self._take(scales[-1], p))                                  # :311
        phase = 0
        if begun is not None:
            phase, x = 2, begun["x"]

        def run(variant, pc_global, scales, x_, vh_, flag, phase=0):
            table, blobs, head = self._images(variant)
            return ops.scone_occ_forward(pc_global, scales, x_, vh_, table, blobs, head, flag, phase=phase)

        flag = None
        guard = self._effective_guard()
        if variant in (6, 7) and guard != "off":
            if self._range_flag is None or self._range_flag.device != dev:
                self._range_flag = torch.zeros(1, dtype=torch.int32, device=dev)
            elif guard in ("sync", "async"):
                self._range_flag.zero_()
            flag = self._range_flag
        if A.needs_grad(self, pc, x, view_harmonics):   # trainers: HIP forward, composite-torch backward (autograd.py)
            pidx = [p.to(dev) for p in perms]

            def clouds(pc_):
                sc = [pc_.contiguous()]
                for p in pidx[1:]:
                    sc.append(self._take(sc[-1], p))
                return self._take(pc_, pidx[0]), sc

            def hip(pc_, x_, vh_):
                g, sc = clouds(pc_)
                return run(variant, g, sc, x_, vh_, None)

            def composite(pc_, x_, vh_):
                g, sc = clouds(pc_)
                with torch.no_grad():
                    idx = [ops.knn_points(x_.detach().contiguous(), s_.detach(), self.k_for_knn)[2] for s_ in sc]
                return A.scone_occ(self, g, sc, x_, vh_, idx)
            res = A.with_torch_backward(hip, composite, (pc, x, view_harmonics), self)
        else:
            res = run(variant, pc_global, scales, x, view_harmonics, flag, phase)
            if flag is not None and guard == "sync" and int(flag):      # out of the fp16 range: the full-range path
                with ops.variant(5):
                    res = run(5, pc_global, scales, x, view_harmonics, None)
            elif flag is not None and guard == "async":
                self._post_range_check(flag)
        return res.view(n_clouds, n_sample, self.output_dim)
