"""SconeVis / SconeOcc forwards in numpy from a state_dict (dict name -> ndarray).  TEST INFRASTRUCTURE ONLY.

Restates (upstream tree):
  macarons/networks/Attention.py:8-36    attention        :98-128  Embedding.forward
  macarons/networks/Attention.py:174-204 MHSA.forward     :232-235 FeedForward    :278-300 Encoder.forward
  macarons/networks/SconeVis.py:121-162  SconeVis.forward
  macarons/networks/SconeOcc.py:35-42    XEmbedding       :104-130 PCTransformer.forward   :250-347 SconeOcc.forward
dtype selects fp32 (like the reference) or fp64 (accuracy yard-stick).
"""
import numpy as np
from scipy.special import erf

from . import knn as _knn


def _lin(sd, name, x):
    return x @ sd[name + ".weight"].astype(x.dtype).T + sd[name + ".bias"].astype(x.dtype)


def gelu(x):
    return (0.5 * x * (1.0 + erf(x / np.sqrt(2.0)))).astype(x.dtype)           # nn.GELU() = erf form


def layernorm(sd, name, x, eps=1e-5):
    mu = x.mean(-1, keepdims=True)
    var = ((x - mu) ** 2).mean(-1, keepdims=True)
    return ((x - mu) / np.sqrt(var + x.dtype.type(eps)) * sd[name + ".weight"].astype(x.dtype)
            + sd[name + ".bias"].astype(x.dtype)).astype(x.dtype)


def attention(q, k, v, mask=None):
    """Attention.py:8-36: q,k [...,N,d], v [...,N,dv]; mask (optional, broadcastable to the scores): where it is 0 the score is
    REPLACED by -1e3 before the division by sqrt(d) (:24-27)."""
    s = q @ np.swapaxes(k, -1, -2)
    if mask is not None:
        s = np.where(np.asarray(mask) == 0, q.dtype.type(-1e3), s)
    s = s / np.sqrt(q.dtype.type(q.shape[-1]))
    s = s - s.max(-1, keepdims=True)
    e = np.exp(s)
    return ((e / e.sum(-1, keepdims=True)) @ v).astype(q.dtype)


def embedding(sd, pre, x, global_feature):
    """Attention.py:98-128 (k_for_knn=0, no additional feature, concatenate_input=True)."""
    res = _lin(sd, pre + ".linear2", gelu(_lin(sd, pre + ".linear1", x)))
    parts = [res]
    if global_feature:
        parts.append(np.broadcast_to(res.max(axis=1, keepdims=True), res.shape))
    parts.append(x)
    return np.concatenate(parts, axis=-1).astype(x.dtype)


def mhsa(sd, pre, x, n_heads, mask=None):
    B, N, E = x.shape
    q, k, v = _lin(sd, pre + ".w_q", x), _lin(sd, pre + ".w_k", x), _lin(sd, pre + ".w_v", x)
    sp = lambda t: t.reshape(B, N, n_heads, -1).transpose(0, 2, 1, 3)           # :174-176,195
    o = attention(sp(q), sp(k), sp(v), mask).transpose(0, 2, 1, 3).reshape(B, N, E)
    return _lin(sd, pre + ".out", o)


def encoder(sd, pre, x, n_heads=4, mask=None):
    res = x + mhsa(sd, pre + ".mhsa", layernorm(sd, pre + ".norm1", x), n_heads, mask)                # :287-290
    ffin = layernorm(sd, pre + ".norm2", res)
    return (res + _lin(sd, pre + ".ff.linear2", gelu(_lin(sd, pre + ".ff.linear1", ffin)))).astype(x.dtype)   # :293-298


def scone_vis_forward(sd, pts, view_harmonics, dtype=np.float32, n_code=3, mask=None):
    pts = np.asarray(pts, dtype)
    vh = np.asarray(view_harmonics, dtype)
    x = embedding(sd, "embedding", pts, True)
    for i in range(n_code):
        x = encoder(sd, f"encoders.{i}", x, mask=mask)
    res = layernorm(sd, "norm", x)
    res = gelu(_lin(sd, "fc1", res))
    res = np.concatenate((res, vh), axis=-1)
    res = gelu(_lin(sd, "fc2", res))
    return _lin(sd, "fc3", res).astype(dtype)


def pc_transformer(sd, pre, pc, dtype=np.float32, n_code=2, mask=None):
    pc = np.asarray(pc, dtype)
    x = embedding(sd, pre + "embedding", pc, False)
    for i in range(n_code):
        x = encoder(sd, f"{pre}encoders.{i}", x, mask=mask)
    f = _lin(sd, pre + "linear0", layernorm(sd, pre + "norm", x))
    return np.concatenate((f.max(axis=1), f.mean(axis=1)), axis=-1).astype(dtype)       # max || avg pooling


def x_embedding(sd, pre, x):
    r = gelu(_lin(sd, pre + ".linear1", x))
    r = gelu(_lin(sd, pre + ".linear2", r))
    return gelu(_lin(sd, pre + ".linear3", r))


def scone_occ_forward(sd, pc, x, view_harmonics, perms, dtype=np.float32, seq_len=2048, k=16, chunk=4096):
    """perms: the three index arrays the reference would draw (global, scale0->1, scale1->2)."""
    pc32, x32 = np.asarray(pc, np.float32), np.asarray(x, np.float32)
    pc, x, vh = pc32.astype(dtype), x32.astype(dtype), np.asarray(view_harmonics, dtype)
    B, Q, _ = x.shape
    gfeat = pc_transformer(sd, "global_transformer.", pc[:, np.asarray(perms[0])], dtype)          # [B,512]
    scales = [pc32]
    for p in perms[1:]:
        scales.append(scales[-1][:, np.asarray(p)])
    local = []
    for s in range(3):
        offs, _, _ = _knn.knn_offsets(x32, scales[s], k)                                             # kNN always in fp32
        offs = offs.astype(dtype).reshape(B * Q, k, 3)
        f = np.concatenate([pc_transformer(sd, f"local_transformers.{s}.", offs[i:i + chunk], dtype)
                            for i in range(0, B * Q, chunk)], axis=0)
        local.append(f.reshape(B, Q, -1))
    xf = x_embedding(sd, "x_embedding", x)
    g = np.broadcast_to(gfeat[:, None, :], (B, Q, gfeat.shape[-1]))
    res = np.concatenate([g] + local + [xf, vh], axis=-1).astype(dtype)                              # SconeOcc.py:333
    res = gelu(_lin(sd, "linear1", res))
    res = gelu(_lin(sd, "linear2", res))
    return gelu(_lin(sd, "linear3", res)).astype(dtype)                                              # GELU after the last too
