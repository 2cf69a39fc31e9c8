"""Gradients for the trainers (`loss.backward()` through SconeVis.forward / compute_coverage_gain at
macarons/trainers/pretrain_scone_vis.py:224, through SconeOcc.forward at pretrain_scone_occ.py, train_macarons.py:1159-1162).

The forward passes stay on the hand-written HIP kernels.  Each entry point is wrapped in a torch.autograd.Function whose
backward RECOMPUTES the same mathematics with plain torch ops on the same device (composite functions below, written against
the modules' own parameters) under autograd and back-propagates through that: no HIP backward kernels, no activations kept
between forward and backward.  The composites are ordinary differentiable torch code, so they are also what the parity tests
differentiate numerically (tests/test_autograd.py: fp64 finite differences on CPU; on the GPU the composite forward must
reproduce the HIP forward to 1e-4, which makes its gradient the gradient of the kernels' function).

The k-nearest-neighbour indices of SconeOcc are taken from the HIP forward (the selection is piecewise constant: no gradient
flows through it, exactly as with torch.topk indices in the reference, utils.py:1505-1509).
"""
import math

import torch
import torch.nn.functional as F


# ---- real spherical harmonics of a direction, Cartesian form (polar axis +Y, azimuth from +Z toward +X, Condon-Shortley;
# channel k = l*l + l + m: spherical_harmonics.py:111-157 with CustomGeometry.py:27-45 folded in) --------------------------------
def sh_basis(n, max_rank=8):
    """n [..., 3] unit vectors -> [..., max_rank^2].  Y_l^m = N_lm (-1)^m Q_l^m(n_y) {Re, Im}[(n_z + i n_x)^m] where
    P_l^m(x) = (-1)^m (1 - x^2)^(m/2) Q_l^m(x): polynomials only, differentiable everywhere (also on the +-Y axis)."""
    nx, ny, nz = n[..., 0], n[..., 1], n[..., 2]
    re, im = [torch.ones_like(ny)], [torch.zeros_like(ny)]             # (n_z + i n_x)^m = sin^m(polar) e^{i m azimuth}
    for m in range(1, max_rank):
        r_prev, i_prev = re[-1], im[-1]
        re.append(r_prev * nz - i_prev * nx)
        im.append(i_prev * nz + r_prev * nx)
    out = [None] * (max_rank * max_rank)
    for m in range(max_rank):
        dfact = 1.0
        for v in range(2 * m - 1, 1, -2):
            dfact *= v
        q_prev2, q_prev = None, torch.full_like(ny, dfact)              # Q_m^m = (2m-1)!!
        for l in range(m, max_rank):
            if l == m:
                q = q_prev
            elif l == m + 1:
                q = (2 * m + 1) * ny * q_prev
            else:
                q = ((2 * l - 1) * ny * q_prev - (l + m - 1) * q_prev2) / (l - m)
            if l > m:
                q_prev2, q_prev = q_prev, q
            norm = math.sqrt((2 * l + 1) / (4 * math.pi))
            if m == 0:
                out[l * l + l] = norm * q
            else:
                norm *= math.sqrt(2.0 * math.factorial(l - m) / math.factorial(l + m)) * (-1) ** m
                out[l * l + l + m] = norm * q * re[m]
                out[l * l + l - m] = norm * q * im[m]
    return torch.stack(out, dim=-1)


def visibilities(pts, harmonics, X_cam, use_sigmoid=True):
    """[B,C,N]: SconeVis.compute_visibilities (SconeVis.py:164-208) in differentiable torch ops."""
    rays = X_cam[:, :, None, :] - pts[:, None, :, :3]
    n = rays / torch.linalg.norm(rays, dim=-1, keepdim=True)
    z = (sh_basis(n) * harmonics[:, None, :, :]).sum(-1)
    return torch.sigmoid(z) if use_sigmoid else torch.relu(z)


def coverage_gain(pts, harmonics, X_cam, use_sigmoid=True):
    """[B,C]: SconeVis.compute_coverage_gain (SconeVis.py:210-252)."""
    return visibilities(pts, harmonics, X_cam, use_sigmoid).mean(dim=-1)


# ---- the networks --------------------------------------------------------------------------------------------------------------
def _lin(x, layer):
    return F.linear(x, layer.weight, layer.bias)


def embedding(emb, x, lengths=None):
    """Attention.py:98-128 (k_for_knn = 0): linear-GELU-linear, optional cloud-wide max, optional raw input."""
    res = _lin(F.gelu(_lin(x, emb.linear1)), emb.linear2)
    parts = [res]
    if emb.global_feature:
        r = res
        if lengths is not None:                     # padded batch: the maximum runs over each cloud's own rows
            valid = torch.arange(x.shape[1], device=x.device)[None, :, None] < lengths.view(-1, 1, 1)
            r = torch.where(valid, res, torch.full_like(res, float("-inf")))
        parts.append(r.max(dim=1, keepdim=True)[0].expand_as(res))
    if emb.concatenate_input:
        parts.append(x)
    return torch.cat(parts, dim=-1)


def encoder(enc, x, lengths=None):
    """Attention.py:278-300: pre-LN multi-head self-attention + residual, pre-LN feed-forward + residual."""
    E, H = enc.embedding_dim, enc.n_heads
    h = F.layer_norm(x, (E,), enc.norm1.weight, enc.norm1.bias)
    B, L = h.shape[0], h.shape[1]
    q = _lin(h, enc.mhsa.w_q).view(B, L, H, -1).transpose(1, 2)
    k = _lin(h, enc.mhsa.w_k).view(B, L, H, -1).transpose(1, 2)
    v = _lin(h, enc.mhsa.w_v).view(B, L, H, -1).transpose(1, 2)
    s = q @ k.transpose(-1, -2) / math.sqrt(q.shape[-1])
    if lengths is not None:
        s = s.masked_fill(torch.arange(L, device=x.device)[None, None, None, :] >= lengths.view(-1, 1, 1, 1), float("-inf"))
    att = (torch.softmax(s, dim=-1) @ v).transpose(1, 2).reshape(B, L, E)
    x = x + (_lin(att, enc.mhsa.out) if H > 1 else att)
    if enc.FF:
        h = F.layer_norm(x, (E,), enc.norm2.weight, enc.norm2.bias)
        x = x + _lin(F.gelu(_lin(h, enc.ff.linear1)), enc.ff.linear2)
    return x


def scone_vis(model, pts, view_harmonics, lengths=None):
    """SconeVis.forward (SconeVis.py:121-162), default architecture."""
    if lengths is not None:                         # the kernels treat an empty cloud as its first row (max(1, length)); so does this
        lengths = lengths.clamp(min=1)
    x = embedding(model.embedding, pts, lengths)
    for enc in model.encoders:
        x = encoder(enc, x, lengths)
    x = F.layer_norm(x, (x.shape[-1],), model.norm.weight, model.norm.bias)
    x = F.gelu(_lin(x, model.fc1))
    x = F.gelu(_lin(torch.cat((x, view_harmonics), dim=-1), model.fc2))
    return _lin(x, model.fc3)


def pc_transformer(pct, x):
    """PCTransformer.forward (SconeOcc.py:104-130): [S,L,3] -> [S, feature_dim] = max || mean over the sequence."""
    x = embedding(pct.embedding, x)
    for enc in pct.encoders:
        x = encoder(enc, x)
    x = _lin(F.layer_norm(x, (x.shape[-1],), pct.norm.weight, pct.norm.bias), pct.linear0)
    return torch.cat((x.max(dim=1)[0], x.mean(dim=1)), dim=-1)


def scone_occ(model, pc_global, scales, x, view_harmonics, knn_idx):
    """SconeOcc.forward (SconeOcc.py:250-347) given the down-sampled clouds and, per scale, the neighbour indices [B,Q,16]."""
    B, Q = x.shape[0], x.shape[1]
    feats = [pc_transformer(model.global_transformer, pc_global)[:, None, :].expand(-1, Q, -1)]
    for pc_s, idx, lt in zip(scales, knn_idx, model.local_transformers):
        nb = torch.gather(pc_s[:, None].expand(-1, Q, -1, -1), 2, idx[..., None].expand(-1, -1, -1, 3))      # [B,Q,16,3]
        off = nb - x[:, :, None, :]
        feats.append(pc_transformer(lt, off.reshape(B * Q, idx.shape[-1], 3)).view(B, Q, -1))
    xe = model.x_embedding
    feats.append(F.gelu(_lin(F.gelu(_lin(F.gelu(_lin(x, xe.linear1)), xe.linear2)), xe.linear3)))
    feats.append(view_harmonics)
    h = torch.cat(feats, dim=-1)
    return F.gelu(_lin(F.gelu(_lin(F.gelu(_lin(h, model.linear1)), model.linear2)), model.linear3))


# ---- HIP forward + composite backward --------------------------------------------------------------------------------------
class _HipForwardTorchBackward(torch.autograd.Function):
    """apply(hip_fn, torch_fn, n_tensor_inputs, *tensor_inputs_then_params): forward = hip_fn(*inputs) without a graph;
    backward = autograd through torch_fn(*inputs) (recomputed) for every input / parameter that requires a gradient."""

    @staticmethod
    def forward(ctx, hip_fn, torch_fn, n_in, *tensors):
        ctx.torch_fn, ctx.n_in = torch_fn, n_in
        ctx.inputs = tensors[:n_in]                 # plain references: nothing but the op's own inputs is kept for the backward
        ctx.params = tensors[n_in:]
        with torch.no_grad():
            return hip_fn(*tensors[:n_in])

    @staticmethod
    def backward(ctx, grad_out):
        n_in = ctx.n_in
        with torch.enable_grad():
            ins = [t.detach().requires_grad_(True) if (t.is_floating_point() and ctx.needs_input_grad[3 + i]) else t
                   for i, t in enumerate(ctx.inputs)]
            out = ctx.torch_fn(*ins)                # the composite reads the module's own parameters
            wrt = [(i, t) for i, t in enumerate(ins) if t.requires_grad and ctx.needs_input_grad[3 + i]]
            wrt += [(n_in + j, p) for j, p in enumerate(ctx.params) if ctx.needs_input_grad[3 + n_in + j]]
            grads = torch.autograd.grad(out, [t for _, t in wrt], grad_out, allow_unused=True) if wrt else []
        res = [None] * (n_in + len(ctx.params))
        for (i, _), g in zip(wrt, grads):
            res[i] = g
        return (None, None, None, *res)


def needs_grad(module, *tensors):
    return torch.is_grad_enabled() and (any(getattr(t, "requires_grad", False) for t in tensors if t is not None)
                                        or (module is not None and any(p.requires_grad for p in module.parameters())))


def with_torch_backward(hip_fn, torch_fn, inputs, module=None):
    """Run hip_fn(*inputs); if a gradient is needed, make the result differentiable through torch_fn(*inputs)."""
    params = tuple(module.parameters()) if module is not None else ()
    return _HipForwardTorchBackward.apply(hip_fn, torch_fn, len(inputs), *inputs, *params)
