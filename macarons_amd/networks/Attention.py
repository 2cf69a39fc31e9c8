"""Host-side mirror of macarons/networks/Attention.py — same class names, constructor arguments,
sub-module / parameter names (so reference checkpoints load unchanged) and tensor layouts; every forward()
runs on the MI355X through libmacarons_hip.so (include/macarons_hip.h).  No CPU fallback.

Reference: macarons/networks/Attention.py  (attention :8, Embedding :39, MultiHeadSelfAttention :131,
FeedForward :207, Encoder :239).
"""
import warnings

import numpy as np
import torch
from torch import nn
import torch.nn.functional as F          # noqa: F401  (handed on by upstream's `from .Attention import *`)

from .. import ops
from ..utility.utils import knn_gather   # noqa: F401  (Attention.py:5 imports it from pytorch3d.ops; same gather)

_warned = [False]


def _inference_only(module, *tensors):
    """The BLOCK-level modules (Embedding, Encoder, ... called on their own) record no autograd graph; gradients are provided at the
    entry points the trainers use (SconeVis.forward, SconeOcc.forward, the scorer methods: macarons_amd/autograd.py)."""
    if torch.is_grad_enabled() and not _warned[0]:
        if any(p.requires_grad for p in module.parameters()) or any(getattr(t, "requires_grad", False) for t in tensors):
            warnings.warn("macarons_amd: this block-level module records no gradients on the MI355X HIP path (SconeVis / SconeOcc "
                          "forward and the scorer methods do); wrap calls in torch.no_grad() to silence this.", RuntimeWarning,
                          stacklevel=3)
            _warned[0] = True


def _f32c(p):
    """Parameter -> contiguous fp32 tensor sharing storage when it already is (no copy in the common case)."""
    t = p.detach()
    if t.dtype != torch.float32:
        raise TypeError("macarons_amd HIP path needs fp32 parameters")
    return t if t.is_contiguous() else t.contiguous()


def attention(q, k, v, mask=None, dropout=None):
    """Attention.py:8-36.  q,k [B,H,N,d], v [B,H,N,dv] -> [B,H,N,dv].  mask: anything that broadcasts against the [B,H,N,N] scores
    as upstream's masked_fill does ([B,1,N,N], [N,N], [B,H,N,N]); masked pairs score -1e3 BEFORE the 1/sqrt(d) scale (:24-27).
    dropout: inference path, None only."""
    if dropout is not None:
        raise NotImplementedError("macarons_amd.attention: dropout is not used on the hot path (SURVEY §8 a9)")
    B, H, N, d = q.shape
    dv = v.shape[-1]
    packed = torch.cat((q.transpose(1, 2).reshape(B, N, H * d), k.transpose(1, 2).reshape(B, N, H * d),
                        v.transpose(1, 2).reshape(B, N, H * dv)), dim=-1).contiguous()
    out = ops.attention_packed(packed, H, H * d, H * dv, mask=mask)
    return out.reshape(B, N, H, dv).transpose(1, 2)


class Embedding(nn.Module):
    """Attention.py:39-128 (k_for_knn = 0 only: the kNN max-pool variant is never used)."""

    def __init__(self, input_dim, output_dim, dropout=None, gelu=True, global_feature=False, additional_feature_dim=0,
                 concatenate_input=True, k_for_knn=0):
        super().__init__()
        if k_for_knn > 0:
            raise NotImplementedError("Embedding(k_for_knn > 0) is not on the hot path")
        if not gelu or dropout is not None:
            raise NotImplementedError("HIP path implements the GELU / no-dropout configuration used by every call site")
        self.use_knn = False
        self.k = k_for_knn
        self.input_dim = input_dim
        self.global_feature = global_feature
        self.additional_feature_dim = additional_feature_dim
        self.concatenate_input = concatenate_input
        # dimension bookkeeping exactly as Attention.py:71-87
        self.inner_dim = output_dim // 2
        self.feature_dim = output_dim
        if additional_feature_dim > 0:
            self.feature_dim -= additional_feature_dim
            self.inner_dim = self.feature_dim
        if concatenate_input:
            self.feature_dim -= input_dim
            self.inner_dim = self.feature_dim
        if global_feature:
            self.feature_dim = self.feature_dim // 2
            self.inner_dim = self.feature_dim
        self.linear1 = nn.Linear(self.input_dim, self.inner_dim)
        self.linear2 = nn.Linear(self.inner_dim, self.feature_dim)
        self.dropout = None
        self.nonlinear = nn.GELU()

    def forward(self, x, additional_feature=None):
        _inference_only(self, x)
        res = ops.linear(x, _f32c(self.linear1.weight), _f32c(self.linear1.bias), gelu=True)
        res = ops.linear(res, _f32c(self.linear2.weight), _f32c(self.linear2.bias))
        parts = [res]
        if self.global_feature:
            parts.append(ops.colmax_broadcast(res))
        if self.additional_feature_dim > 0:
            parts.append(additional_feature)
        if self.concatenate_input:
            parts.append(x)
        return torch.cat(parts, dim=-1) if len(parts) > 1 else res


class MultiHeadSelfAttention(nn.Module):
    """Attention.py:131-204."""

    def __init__(self, n_heads, in_dim, qk_dim, dropout=None):
        super().__init__()
        if dropout is not None:
            raise NotImplementedError("dropout is None in every call site")
        self.n_heads, self.in_dim, self.qk_dim, self.v_dim = n_heads, in_dim, qk_dim, in_dim
        self.qk_dim_per_head = qk_dim // n_heads
        self.v_dim_per_head = in_dim // n_heads
        self.w_q = nn.Linear(in_dim, qk_dim)
        self.w_k = nn.Linear(in_dim, qk_dim)
        self.w_v = nn.Linear(in_dim, in_dim)
        self.dropout = None
        if n_heads > 1:
            self.out = nn.Linear(in_dim, in_dim)
        self._packed = None

    def packed_qkv(self):
        """(weight [2*qk+E, E], bias) = rows of w_q, w_k, w_v stacked; rebuilt when a parameter changes."""
        ps = (self.w_q.weight, self.w_q.bias, self.w_k.weight, self.w_k.bias, self.w_v.weight, self.w_v.bias)
        key = tuple((p.data_ptr(), p._version, p.device) for p in ps)
        if self._packed is None or self._packed[0] != key:
            with torch.no_grad():
                w = torch.cat((ps[0], ps[2], ps[4]), dim=0).float().contiguous()
                b = torch.cat((ps[1], ps[3], ps[5]), dim=0).float().contiguous()
            self._packed = (key, w, b)
        return self._packed[1], self._packed[2]

    def split_heads(self, q, k, v):
        return (q.reshape(q.shape[0], -1, self.n_heads, self.qk_dim_per_head),
                k.reshape(k.shape[0], -1, self.n_heads, self.qk_dim_per_head),
                v.reshape(v.shape[0], -1, self.n_heads, self.v_dim_per_head))

    def forward(self, x, mask=None):
        """mask (optional): [B,1,N,N] / [N,N] / [B,H,N,N] as upstream broadcasts it, or [B,N,N] (the shape upstream's docstring
        names), or a [B,N] key mask; see ops.attention_packed."""
        _inference_only(self, x)
        w, b = self.packed_qkv()
        qkv = ops.linear(x, w, b)
        scores = ops.attention_packed(qkv, self.n_heads, self.qk_dim, self.v_dim, mask=mask)
        if self.n_heads > 1:
            scores = ops.linear(scores, _f32c(self.out.weight), _f32c(self.out.bias))
        return scores


class FeedForward(nn.Module):
    """Attention.py:207-236."""

    def __init__(self, input_dim, inner_dim, gelu=True, dropout=None):
        super().__init__()
        if not gelu or dropout is not None:
            raise NotImplementedError("HIP path implements the GELU / no-dropout configuration")
        self.linear1 = nn.Linear(input_dim, inner_dim)
        self.linear2 = nn.Linear(inner_dim, input_dim)
        self.dropout = None
        self.nonlinear = nn.GELU()

    def forward(self, x):
        _inference_only(self, x)
        res = ops.linear(x, _f32c(self.linear1.weight), _f32c(self.linear1.bias), gelu=True)
        return ops.linear(res, _f32c(self.linear2.weight), _f32c(self.linear2.bias))


class Encoder(nn.Module):
    """Attention.py:239-300 (pre-LN; FF=True)."""

    def __init__(self, seq_len, qk_dim, embedding_dim=128, n_heads=1, dropout=None, gelu=True, FF=True):
        super().__init__()
        if dropout is not None or not gelu:
            raise NotImplementedError("HIP path implements the GELU / no-dropout configuration")
        self.seq_len, self.embedding_dim, self.n_heads, self.qk_dim = seq_len, embedding_dim, n_heads, qk_dim
        self.dropout, self.FF = None, FF
        self.norm1 = nn.LayerNorm(embedding_dim)
        self.mhsa = MultiHeadSelfAttention(n_heads=n_heads, in_dim=embedding_dim, qk_dim=qk_dim, dropout=None)
        self.dropout1 = None
        if FF:
            self.norm2 = nn.LayerNorm(embedding_dim)
            self.ff = FeedForward(input_dim=embedding_dim, inner_dim=2 * embedding_dim, gelu=gelu, dropout=None)
            self.dropout2 = None

    def weight_table(self):
        """The 12 tensors of one ENCODER block in the order include/macarons_hip.h documents."""
        w, b = self.mhsa.packed_qkv()
        return [_f32c(self.norm1.weight), _f32c(self.norm1.bias), w, b, _f32c(self.mhsa.out.weight),
                _f32c(self.mhsa.out.bias), _f32c(self.norm2.weight), _f32c(self.norm2.bias), _f32c(self.ff.linear1.weight),
                _f32c(self.ff.linear1.bias), _f32c(self.ff.linear2.weight), _f32c(self.ff.linear2.bias)]

    def forward(self, x, mask=None):
        _inference_only(self, x)
        res = ops.layernorm(x, _f32c(self.norm1.weight), _f32c(self.norm1.bias))
        w, b = self.mhsa.packed_qkv()
        qkv = ops.linear(res, w, b)
        att = ops.attention_packed(qkv, self.n_heads, self.qk_dim, self.embedding_dim, mask=mask)    # mask: see MultiHeadSelfAttention
        res = ops.linear(att, _f32c(self.mhsa.out.weight), _f32c(self.mhsa.out.bias), residual=x) \
            if self.n_heads > 1 else x + att
        if self.FF:
            res2 = ops.layernorm(res, _f32c(self.norm2.weight), _f32c(self.norm2.bias))
            res2 = ops.linear(res2, _f32c(self.ff.linear1.weight), _f32c(self.ff.linear1.bias), gelu=True)
            res = ops.linear(res2, _f32c(self.ff.linear2.weight), _f32c(self.ff.linear2.bias), residual=res)
        return res
