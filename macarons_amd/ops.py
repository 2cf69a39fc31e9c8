"""Thin torch-tensor wrappers over the C ABI (include/macarons_hip.h).

torch is plumbing here: device memory, streams.  Every function validates layout, allocates outputs /
scratch on the input's device and launches on the current HIP stream.  No CPU fallback.
"""
import os
import ctypes
import torch
import numpy as np

from ._lib import lib, check, MacaronsHipError, c_i64, c_int, c_size, c_vp, c_f32
from .utility.host import limit_host_threads

# torch's intra-op pool follows the machine's core count, not the container's CPU quota: an oversubscribed parallel region gets
# the whole process throttled, the launching thread included (utility/host.py has the measurement)
limit_host_threads()


def _stream():
    return c_vp(torch.cuda.current_stream().cuda_stream)


def _req(t, name, dtype=torch.float32):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if not t.is_cuda:
        raise MacaronsHipError(f"{name} must live on a HIP device (got {t.device}); "
                               "the MI355X hot path has no CPU fallback")
    if t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype} (got {t.dtype})")
    return t.contiguous()


def _p(t):
    return c_vp(t.data_ptr())


# ---- numerics variant of the network entry points: a property of the CALL (mcr_call_variant), scoped per thread on the host ------------
import contextlib
import threading

_tls = threading.local()


@contextlib.contextmanager
def variant(v):
    """`with ops.variant(5): ...` -- every network entry point called inside, ON THIS THREAD, runs on variant v (the one-shot per-call
    argument of the C ABI, mcr_call_variant, set before each of them).  The process default (mcr_set_local_pct_variant) is never
    touched: another thread, or another model, does not see the choice, and an exception cannot leave a flipped switch behind."""
    prev = getattr(_tls, "variant", 0)
    _tls.variant = int(v)
    try:
        yield
    finally:
        _tls.variant = prev


def current_variant():
    """The variant a network call made here and now would run on."""
    return getattr(_tls, "variant", 0) or int(lib().mcr_get_local_pct_variant())


def _net(L_):
    """Hand the scoped variant (if any) to the next network entry point of this thread; returns the library."""
    v = getattr(_tls, "variant", 0)
    if v:
        L_.mcr_call_variant(c_int(v))
    return L_


# ---- K9 scorer -------------------------------------------------------------------------------------
def _scorer_args(pts, harmonics, cams):
    pts, harmonics, cams = _req(pts, "pts"), _req(harmonics, "harmonics"), _req(cams, "X_cam")
    B, N, P = pts.shape
    C = cams.shape[1]
    if harmonics.shape != (B, N, 64):
        raise ValueError(f"harmonics must be [B,N,64] = {(B, N, 64)}, got {tuple(harmonics.shape)}")
    if cams.shape != (B, C, 3):
        raise ValueError(f"X_cam must be [B,C,3], got {tuple(cams.shape)}")
    return pts, harmonics, cams, B, N, P, C


def sh_coverage_gain(pts, harmonics, cams, use_sigmoid=True, waves_per_simd=0):
    """gains [B,C]; replaces SconeVis.compute_coverage_gain (SconeVis.py:210-252)."""
    pts, harmonics, cams, B, N, P, C = _scorer_args(pts, harmonics, cams)
    gains = torch.empty((B, C), dtype=torch.float32, device=pts.device)
    L = lib()
    ws = _workspace(pts.device, L.mcr_sh_coverage_gain_workspace_bytes(c_i64(B), c_i64(N), c_i64(C)))
    with torch.cuda.device(pts.device):
        check(L.mcr_sh_coverage_gain(_p(pts), c_int(P), _p(harmonics), _p(cams), _p(gains), c_i64(B), c_i64(N), c_i64(C),
                                     c_int(int(bool(use_sigmoid))), c_int(waves_per_simd), _p(ws), c_size(ws.numel()), _stream()),
              "mcr_sh_coverage_gain")
    return gains


def sh_coverage_gain_best(pts, harmonics, cams, use_sigmoid=True, waves_per_simd=0):
    """(gains [B,C], record [B,2] = (max gain, first arg-max camera as fp32)): SconeVis.compute_coverage_gain (SconeVis.py:210-252) and
    the decision of testers/shapenet.py:172 (torch.max over the cameras; a NaN gain wins) in one call (mcr_sh_coverage_gain_best)."""
    pts, harmonics, cams, B, N, P, C = _scorer_args(pts, harmonics, cams)
    gains = torch.empty((B, C), dtype=torch.float32, device=pts.device)
    record = torch.empty((B, 2), dtype=torch.float32, device=pts.device)
    L = lib()
    ws = _workspace(pts.device, L.mcr_sh_coverage_gain_workspace_bytes(c_i64(B), c_i64(N), c_i64(C)))
    with torch.cuda.device(pts.device):
        check(L.mcr_sh_coverage_gain_best(_p(pts), c_int(P), _p(harmonics), _p(cams), _p(gains), _p(record), c_i64(B), c_i64(N), c_i64(C),
                                          c_int(int(bool(use_sigmoid))), c_int(waves_per_simd), _p(ws), c_size(ws.numel()), _stream()),
              "mcr_sh_coverage_gain_best")
    return gains, record


def sh_coverage_gain_partials(pts, harmonics, cams, use_sigmoid=True, waves_per_simd=0):
    """First stage of the scorer only (sh_gain_kernel; per-(wave tile, camera) partial sums stay in the scratch arena): lets
    bench.py time the dominant kernel alone with HIP events.  Returns nothing."""
    pts, harmonics, cams, B, N, P, C = _scorer_args(pts, harmonics, cams)
    L = lib()
    ws = _workspace(pts.device, max(L.mcr_sh_coverage_gain_workspace_bytes(c_i64(B), c_i64(N), c_i64(C)), 4))
    with torch.cuda.device(pts.device):
        check(L.mcr_sh_coverage_gain_partials(_p(pts), c_int(P), _p(harmonics), _p(cams), c_i64(B), c_i64(N), c_i64(C),
                                              c_int(int(bool(use_sigmoid))), c_int(waves_per_simd), _p(ws), c_size(ws.numel()),
                                              _stream()), "mcr_sh_coverage_gain_partials")


def sh_visibilities(pts, harmonics, cams, use_sigmoid=True):
    """vis [B,C,N]; replaces SconeVis.compute_visibilities (SconeVis.py:164-208)."""
    pts, harmonics, cams = _req(pts, "pts"), _req(harmonics, "harmonics"), _req(cams, "X_cam")
    B, N, P = pts.shape
    C = cams.shape[1]
    if harmonics.shape != (B, N, 64):
        raise ValueError(f"harmonics must be [B,N,64] = {(B, N, 64)}, got {tuple(harmonics.shape)}")
    if cams.shape != (B, C, 3):
        raise ValueError(f"X_cam must be [B,C,3], got {tuple(cams.shape)}")
    vis = torch.empty((B, C, N), dtype=torch.float32, device=pts.device)
    with torch.cuda.device(pts.device):
        check(lib().mcr_sh_visibilities(_p(pts), c_int(P), _p(harmonics), _p(cams), _p(vis), c_i64(B), c_i64(N),
                                        c_i64(C), c_int(int(bool(use_sigmoid))), _stream()), "mcr_sh_visibilities")
    return vis


# ---- K1 kNN ----------------------------------------------------------------------------------------
def knn_points(X, pc, k, subtract_query=False):
    """(pts [B,Q,k,3], dists [B,Q,k], idx [B,Q,k] int64); replaces utils.get_knn_points (utils.py:1497-1509);
    subtract_query=True also applies SconeOcc.py:297-298 (neighbours minus the query)."""
    X, pc = _req(X, "X"), _req(pc, "pc")
    B, Q, d = X.shape
    M = pc.shape[1]
    if d != 3 or pc.shape[0] != B or pc.shape[2] != 3:
        raise ValueError(f"X must be [B,Q,3] and pc [B,M,3]; got {tuple(X.shape)}, {tuple(pc.shape)}")
    idx = torch.empty((B, Q, k), dtype=torch.int64, device=X.device)
    dists = torch.empty((B, Q, k), dtype=torch.float32, device=X.device)
    pts = torch.empty((B, Q, k, 3), dtype=torch.float32, device=X.device)
    with torch.cuda.device(X.device):
        nb = int(lib().mcr_knn_grid_workspace_bytes(c_i64(B), c_i64(Q), c_i64(M)))
        ws = _workspace(X.device, nb)
        check(lib().mcr_knn_points_grid(_p(X), _p(pc), _p(idx), _p(dists), _p(pts), c_i64(B), c_i64(Q), c_i64(M), c_int(k),
                                        c_int(int(bool(subtract_query))), _p(ws), ctypes.c_size_t(ws.numel()), _stream()),
              "mcr_knn_points_grid")
    return pts, dists, idx


def knn_offsets_segmented(X, pc, cloud_sizes, query_sizes, split=True):
    """offsets [T,16,3] (neighbour minus query) of J independent k = 16 searches in one launch: job j = query rows
    sum(query_sizes[:j]) .. against cloud rows sum(cloud_sizes[:j]) ..; neighbours in knn_points' order (the ragged occupancy pass).
    split=False withholds the scratch that lets a small launch split the candidates over several workgroups (same result)."""
    X, pc = _req(X, "X"), _req(pc, "pc")
    T, J = X.shape[0], len(cloud_sizes)
    if sum(query_sizes) != T or sum(cloud_sizes) != pc.shape[0] or len(query_sizes) != J or min(cloud_sizes) < 16:
        raise ValueError("knn_offsets_segmented: sizes do not match (every cloud needs >= 16 points)")
    L = lib()
    rows = int(L.mcr_knn_rows_per_block())
    blocks, r0 = [], 0
    for j, q in enumerate(query_sizes):
        blocks += [(j, r0 + b0, min(rows, q - b0), 0) for b0 in range(0, q, rows)]
        r0 += q
    d_off = h2d(np.concatenate(([0], np.cumsum(cloud_sizes))).astype(np.int64), torch.int64, X.device)
    d_blocks = h2d(np.asarray(blocks, np.int32).reshape(-1, 4), torch.int32, X.device)
    out = torch.empty((T, 16, 3), dtype=torch.float32, device=X.device)
    with torch.cuda.device(X.device):
        ws = _workspace(X.device, max(int(L.mcr_knn_offsets_segmented_workspace_bytes(c_i64(T))), 4)) if split else None
        check(L.mcr_knn_offsets_segmented(_p(X), _p(pc), _p(d_off), _p(d_blocks), c_i64(len(blocks)), c_i64(T), _p(out),
                                          _p(ws) if ws is not None else c_vp(0), c_size(ws.numel() * ws.element_size() if ws is not None else 0),
                                          _stream()), "mcr_knn_offsets_segmented")
    return out


# ---- K4/K5 building blocks ---------------------------------------------------------------------------
def _rows(t):
    """View a [..., E] tensor as rows: returns (2-D contiguous tensor, leading shape)."""
    lead = t.shape[:-1]
    return t.reshape(-1, t.shape[-1]), lead


def linear(x, weight, bias=None, gelu=False, residual=None):
    """nn.Linear (+ exact GELU) (+ residual) on the last dim; replaces Attention.py:98-103,186-188,232-235."""
    x = _req(x, "x")
    weight = _req(weight, "weight")
    x2, lead = _rows(x)
    M, K = x2.shape
    N = weight.shape[0]
    if weight.shape[1] != K:
        raise ValueError(f"weight {tuple(weight.shape)} does not match input width {K}")
    y = torch.empty((M, N), dtype=torch.float32, device=x.device)
    b = _req(bias, "bias") if bias is not None else None
    r = _req(residual, "residual").reshape(M, N) if residual is not None else None
    with torch.cuda.device(x.device):
        check(_net(lib()).mcr_linear(_p(x2), c_i64(K), _p(weight), _p(b) if b is not None else c_vp(0),
                               _p(r) if r is not None else c_vp(0), c_i64(N), _p(y), c_i64(N), c_i64(M), c_int(N), c_int(K),
                               c_int(int(gelu)), _stream()), "mcr_linear")
    return y.reshape(*lead, N)


def layernorm(x, weight, bias):
    x = _req(x, "x")
    x2, lead = _rows(x)
    M, E = x2.shape
    y = torch.empty_like(x2)
    with torch.cuda.device(x.device):
        check(lib().mcr_layernorm(_p(x2), c_i64(E), _p(_req(weight, "weight")), _p(_req(bias, "bias")), _p(y), c_i64(E),
                                  c_i64(M), c_int(E), _stream()), "mcr_layernorm")
    return y.reshape(*lead, E)


def _mask_bytes(mask, S, H, L, device):
    """A reference-style attention mask -> (uint8 device tensor [s, h, q, L], seq / head / query strides in bytes) for
    mcr_attention_masked.  Accepted: whatever broadcasts against the [S, H, L, L] scores the way upstream's
    `scores.masked_fill(mask == 0, -1e3)` does (Attention.py:24-27) -- [L, L], [S, 1, L, L], [1, 1, L, L], [S, H, L, L] -- plus
    [S, L, L] (the shape upstream's docstrings name: read as [S, 1, L, L]; REFUSED when S == H != 1, where upstream's broadcast
    means one mask per head instead) and key masks [S, L] (a 2-D mask of shape [L, L] is
    upstream's; use key_mask() to force the key reading when S == L) / [S, 1, 1, L].  Non-zero = attend."""
    m = mask if torch.is_tensor(mask) else torch.as_tensor(mask)
    m = (m != 0).to(device=device, dtype=torch.uint8)
    if m.dim() == 3 and tuple(m.shape) == (S, L, L) and S == H and S != 1:
        # upstream's masked_fill broadcasts [B,N,N] against [B,H,N,N] as [1,B,N,N]: with B == H that is one mask PER HEAD shared by
        # the sequences, while upstream's docstrings (and the [S,1,L,L] reading below) mean one mask per sequence -- refuse to guess
        raise ValueError(f"a 3-D attention mask of shape {tuple(m.shape)} with as many sequences as heads ({H}) is ambiguous: upstream "
                         "broadcasts it per HEAD, its docstrings mean per SEQUENCE; pass mask[:, None] ([S,1,L,L]) or mask[None] "
                         "([1,H,L,L]) explicitly")
    if m.dim() == 2 and tuple(m.shape) == (L, L):
        if S == L and S != 1:
            import warnings
            warnings.warn(f"2-D attention mask [{L}, {L}] with {S} sequences of {L} tokens: read as upstream's [L, L] pair mask; for a "
                          "[S, L] key mask use ops.key_mask()", stacklevel=3)
        m4 = m.view(1, 1, L, L)
    elif m.dim() == 2 and tuple(m.shape) == (S, L):
        m4 = m.view(S, 1, 1, L)
    elif m.dim() == 3 and tuple(m.shape) == (S, L, L):
        m4 = m.view(S, 1, L, L)
    elif m.dim() == 4 and m.shape[3] == L and m.shape[2] in (1, L) and m.shape[0] in (1, S) and m.shape[1] in (1, H):
        m4 = m
    else:
        raise ValueError(f"attention mask of shape {tuple(m.shape)} does not fit {S} sequences x {H} heads x {L} tokens")
    m4 = m4.contiguous()
    s_seq = m4.stride(0) if m4.shape[0] > 1 else 0
    s_head = m4.stride(1) if m4.shape[1] > 1 else 0
    s_q = m4.stride(2) if m4.shape[2] > 1 else 0
    return m4, s_seq, s_head, s_q


def key_mask(mask, S, L, device):
    """[S, L] key mask as the [S, 1, 1, L] form _mask_bytes takes (also when S == L)."""
    m = mask if torch.is_tensor(mask) else torch.as_tensor(mask)
    if tuple(m.shape) != (S, L):
        raise ValueError(f"key mask must be [S, L] = {(S, L)}, got {tuple(m.shape)}")
    return (m != 0).to(device=device, dtype=torch.uint8).view(S, 1, 1, L)


def attention_packed(qkv, n_heads, qk_dim, v_dim, split=True, mask=None):
    """qkv [S, L, 2*qk_dim + v_dim] -> [S, L, v_dim]; attention() + head split/merge of Attention.py:8-36,174-198.
    split=True hands the kernel a scratch buffer so that one or two long sequences can split their keys over two blocks.
    mask (optional): see _mask_bytes; masked pairs score -1e3 before the 1/sqrt(d) scale, as upstream (:24-27)."""
    qkv = _req(qkv, "qkv")
    S, L, W = qkv.shape
    if W != 2 * qk_dim + v_dim:
        raise ValueError("packed qkv width mismatch")
    out = torch.empty((S, L, v_dim), dtype=torch.float32, device=qkv.device)
    with torch.cuda.device(qkv.device):
        use_ws = split and L >= 512 and -(-L // 64) * n_heads * S <= 256      # the kernel's own key-split predicate (launch_attention)
        if mask is not None:
            m4, s_seq, s_head, s_q = _mask_bytes(mask, S, n_heads, L, qkv.device)
            ws = _workspace(qkv.device, int(lib().mcr_attention_workspace_bytes(c_i64(S), c_i64(L), c_int(n_heads), c_int(v_dim)))) if use_ws else None
            check(_net(lib()).mcr_attention_masked(_p(qkv), c_i64(W), _p(out), c_i64(v_dim), c_i64(S), c_i64(L), c_int(n_heads), c_int(qk_dim),
                                             c_int(v_dim), _p(m4), c_i64(s_seq), c_i64(s_head), c_i64(s_q),
                                             _p(ws) if ws is not None else c_vp(0), ctypes.c_size_t(ws.numel() if ws is not None else 0),
                                             _stream()), "mcr_attention_masked")
        elif use_ws:
            ws = _workspace(qkv.device, int(lib().mcr_attention_workspace_bytes(c_i64(S), c_i64(L), c_int(n_heads), c_int(v_dim))))
            check(_net(lib()).mcr_attention_ws(_p(qkv), c_i64(W), _p(out), c_i64(v_dim), c_i64(S), c_i64(L), c_int(n_heads),
                                         c_int(qk_dim), c_int(v_dim), _p(ws), ctypes.c_size_t(ws.numel()), _stream()), "mcr_attention_ws")
        else:
            check(_net(lib()).mcr_attention(_p(qkv), c_i64(W), _p(out), c_i64(v_dim), c_i64(S), c_i64(L), c_int(n_heads),
                                      c_int(qk_dim), c_int(v_dim), _stream()), "mcr_attention")
    return out


def attention_packed_planes(qkv, n_heads, qk_dim, v_dim, lens=None, split_mode=-1):
    """attention_packed on the planes pipeline of the long-sequence encoders (variant 6): the packed rows are split once into fp16 hi/lo
    planes, the kernel stages K / V tiles by LDS DMA and multiplies on fp16 pairs (attention_planes.hip).  |q|, |k|, |v| < 65504.
    lens (optional int32 [S], device): keys of sequence s = its first lens[s] rows.  split_mode: see mcr_attention_planes."""
    qkv = _req(qkv, "qkv")
    S, L, W = qkv.shape
    if W != 2 * qk_dim + v_dim:
        raise ValueError("packed qkv width mismatch")
    out = torch.empty((S, L, v_dim), dtype=torch.float32, device=qkv.device)
    if lens is not None:
        lens = _req(lens, "lens", torch.int32)
        if lens.numel() != S:
            raise ValueError("lens must hold one length per sequence")
    with torch.cuda.device(qkv.device):
        nbytes = int(lib().mcr_attention_planes_workspace_bytes(c_i64(S), c_i64(L), c_int(n_heads), c_int(qk_dim), c_int(v_dim)))
        ws = _workspace(qkv.device, nbytes)
        check(lib().mcr_attention_planes(_p(qkv), c_i64(W), _p(out), c_i64(v_dim), c_i64(S), c_i64(L), c_int(n_heads), c_int(qk_dim),
                                         c_int(v_dim), _p(lens) if lens is not None else c_vp(0), c_int(split_mode), _p(ws),
                                         ctypes.c_size_t(ws.numel()), _stream()), "mcr_attention_planes")
    return out


def colmax_broadcast(x):
    """x [S, L, E] -> [S, L, E] where every row holds the column-wise max over the L rows (Attention.py:117-121)."""
    x = _req(x, "x")
    S, L, E = x.shape
    y = torch.empty_like(x)
    with torch.cuda.device(x.device):
        check(lib().mcr_colmax_broadcast(_p(x), c_i64(E), _p(y), c_i64(E), c_i64(S), c_i64(L), c_int(E), _stream()),
              "mcr_colmax_broadcast")
    return y


def pool_max_avg(x):
    """x [S, L, E] -> [S, 2E] = (max over L || mean over L)   (SconeOcc.py:123-126)."""
    x = _req(x, "x")
    S, L, E = x.shape
    y = torch.empty((S, 2 * E), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        check(lib().mcr_pool_max_avg(_p(x), c_i64(E), _p(y), c_i64(2 * E), c_i64(S), c_i64(L), c_int(E), _stream()),
              "mcr_pool_max_avg")
    return y


# ---- network forwards ----------------------------------------------------------------------------------
_ws_cache = {}


def _workspace(device, nbytes, tag=None):
    """One grow-only scratch arena per (device, stream): the C ABI never allocates, and two streams running workspace-using ops
    concurrently must not share scratch.  A grown arena replaces the old one; the old tensor is freed by torch's caching
    allocator only after the work queued on its stream (record_stream).  `tag`: an arena of its own, for an op whose scratch has
    to survive other ops on the stream (the two-phase SconeOcc forwards keep their features in it between the calls)."""
    stream = torch.cuda.current_stream(device)
    key = (device.type, device.index, stream.cuda_stream, tag)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(int(nbytes * 1.1) + 256, dtype=torch.uint8, device=device)
        buf.record_stream(stream)
        _ws_cache[key] = buf
    return buf


_ws_epoch = {}


def _bump_epoch(device, tag):
    key = (device.index if device.index is not None else torch.cuda.current_device(), torch.cuda.current_stream(device).cuda_stream, tag)
    _ws_epoch[key] = _ws_epoch.get(key, 0) + 1


def scone_occ_epoch(device, tag="scone_occ"):
    """How many SconeOcc launch sequences have written the tagged arena of the current stream of `device`.  The two-phase forwards
    keep phase 1's results (feature planes, query order, park counters) in that arena: a handle remembers the count after its phase 1
    and is honoured only while the count still stands (any other forward on the stream in between would have clobbered them)."""
    device = torch.device(device)
    key = (device.index if device.index is not None else torch.cuda.current_device(), torch.cuda.current_stream(device).cuda_stream, tag)
    return _ws_epoch.get(key, 0)


def _ptr_table(tensors):
    """tensors: a list of tensors, or a (tensors, ready-made ctypes pointer array) pair from networks.packing.TableCache."""
    if isinstance(tensors, tuple):
        return tensors[1]
    arr = (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
    return arr


def _n_weights(tensors):
    return len(tensors[0]) if isinstance(tensors, tuple) else len(tensors)


def pc_transformer_forward(pc, weights, feature_dim):
    pc = _req(pc, "pc")
    S, L, d = pc.shape
    if d != 3:
        raise ValueError("PCTransformer HIP path needs pts_dim = 3")
    L_ = lib()
    out = torch.empty((S, feature_dim), dtype=torch.float32, device=pc.device)
    nb = L_.mcr_pc_transformer_workspace_bytes(c_i64(S), c_i64(L))
    ws = _workspace(pc.device, nb)
    tab = _ptr_table(weights)
    with torch.cuda.device(pc.device):
        check(_net(L_).mcr_pc_transformer_forward(_p(pc), _p(out), c_i64(S), c_i64(L), c_int(feature_dim), tab,
                                            c_int(_n_weights(weights)), _p(ws), c_size(ws.numel()), _stream()),
              "mcr_pc_transformer_forward")
    return out


def scone_vis_forward(pts, view_harmonics, weights, lengths=None):
    """lengths (optional, int32 device tensor [B]): cloud b = its first lengths[b] rows (padded variable-length batch)."""
    pts, view_harmonics = _req(pts, "pts"), _req(view_harmonics, "view_harmonics")
    B, N, d = pts.shape
    if d != 4 or view_harmonics.shape != (B, N, 64):
        raise ValueError(f"SconeVis HIP path needs pts [B,N,4] and view_harmonics [B,N,64]; got {tuple(pts.shape)}, "
                         f"{tuple(view_harmonics.shape)}")
    L_ = lib()
    out = torch.empty((B, N, 64), dtype=torch.float32, device=pts.device)
    nb = L_.mcr_scone_vis_workspace_bytes(c_i64(B), c_i64(N))
    ws = _workspace(pts.device, nb)
    tab = _ptr_table(weights)
    if lengths is not None:
        lengths = _req(lengths, "lengths", torch.int32).reshape(-1)
        if lengths.numel() != B:
            raise ValueError(f"lengths must hold one int32 per cloud ({B}), got {lengths.numel()}")
    with torch.cuda.device(pts.device):
        check(_net(L_).mcr_scone_vis_forward(_p(pts), _p(view_harmonics), _p(out), c_i64(B), c_i64(N), tab, c_int(_n_weights(weights)),
                                       _p(lengths) if lengths is not None else c_vp(0), _p(ws), c_size(ws.numel()), _stream()),
              "mcr_scone_vis_forward")
    return out


def nonfinite_flag_(x, flag):
    """flag (int32 device [1]) |= 1 if x holds an inf / NaN (mcr_nonfinite_flag); no read-back."""
    x = _req(x, "x")
    if x.numel():
        with torch.cuda.device(x.device):
            check(lib().mcr_nonfinite_flag(_p(x), c_i64(x.numel()), _p(_req(flag, "flag", torch.int32)), _stream()), "mcr_nonfinite_flag")
    return flag


def local_pct_forward(offsets, blob):
    """offsets [S,16,3] -> [S,256]: fused local PCTransformer (local_pct.hip)."""
    offsets, blob = _req(offsets, "offsets"), _req(blob, "blob")
    S = offsets.shape[0]
    if tuple(offsets.shape[1:]) != (16, 3):
        raise ValueError("offsets must be [S,16,3]")
    out = torch.empty((S, 256), dtype=torch.float32, device=offsets.device)
    with torch.cuda.device(offsets.device):
        check(_net(lib()).mcr_local_pct_forward(_p(offsets), _p(out), c_i64(256), c_i64(S), _p(blob), _stream()),
              "mcr_local_pct_forward")
    return out


def scone_occ_forward(pc_global, pc_scales, x, view_harmonics, weights, local_blobs=None, head_planes=None, range_flag=None, phase=0,
                      M_scale=None, Lg=None, out=None):
    """head_planes: packing.HeadPlaneCache.get() triple (pre-split fp16 planes of the head matrices, variant 6) or None;
    range_flag: int32 device tensor [1] the call ORs 1 into when an occupancy comes out non-finite (or None).
    phase 1 / 2 (mcr_scone_occ_forward_phase): scale 0 and the query order / the rest, as two calls on the same stream.  Phase 1
    takes pc_scales = [whole cloud, None, None] with the three sizes in M_scale and Lg, no pc_global / view_harmonics, and returns
    None; phase 2 takes everything (and `out`, if the caller already has the buffer)."""
    late = phase != 1
    x = _req(x, "x")
    B, Q = x.shape[0], x.shape[1]
    if late:
        pc_global, view_harmonics = _req(pc_global, "pc_global"), _req(view_harmonics, "view_harmonics")
        pc_scales = [_req(p, "pc_scale") for p in pc_scales]
        Lg = pc_global.shape[1]
        M_scale = [p.shape[1] for p in pc_scales]
        if pc_global.shape[0] != B or len(pc_scales) != 3 or x.shape != (B, Q, 3) or view_harmonics.shape != (B, Q, 64):
            raise ValueError("SconeOcc HIP path needs 3 scales, x [B,Q,3] and view_harmonics [B,Q,64]")
    else:
        pc_scales = [_req(pc_scales[0], "pc_scale"), None, None]
        if x.shape != (B, Q, 3) or M_scale is None or len(M_scale) != 3 or Lg is None or pc_scales[0].shape[1] != M_scale[0]:
            raise ValueError("SconeOcc phase 1 needs x [B,Q,3], the whole cloud, the three scale sizes and Lg")
    L_ = lib()
    if late and out is None:
        out = torch.empty((B, Q, 1), dtype=torch.float32, device=x.device)
    nb = L_.mcr_scone_occ_workspace_bytes(c_i64(B), c_i64(Q), c_i64(Lg))
    ws = _workspace(x.device, nb, "scone_occ")
    _bump_epoch(x.device, "scone_occ")
    tab = _ptr_table(weights)
    sc_ptrs = (ctypes.c_void_p * 3)(*[p.data_ptr() if p is not None else None for p in pc_scales])
    sc_m = (ctypes.c_int64 * 3)(*[int(m) for m in M_scale])
    blobs = (ctypes.c_void_p * 3)(*[_req(b, "local_blob").data_ptr() for b in local_blobs]) if local_blobs else None
    with torch.cuda.device(x.device):
        check(_net(L_).mcr_scone_occ_forward_phase(_p(pc_global) if late else c_vp(0), c_i64(Lg), sc_ptrs, sc_m, _p(x),
                                             _p(view_harmonics) if late else c_vp(0), _p(out) if late else c_vp(0), c_i64(B),
                                             c_i64(Q), tab, c_int(_n_weights(weights)), blobs,
                                             head_planes[1] if head_planes is not None else None,
                                             head_planes[2] if head_planes is not None else None,
                                             _p(_req(range_flag, "range_flag", torch.int32)) if range_flag is not None else c_vp(0),
                                             _p(ws), c_size(ws.numel()), c_int(int(phase)), _stream()),
              "mcr_scone_occ_forward")
    return out


def scone_occ_forward_ragged(pc_global, global_len, pc_scales, scale_offsets, x, view_harmonics, row_job, knn_blocks, weights,
                             local_blobs, head_planes=None, range_flag=None, phase=0, Lg=None, out=None, arena="scone_occ_ragged"):
    """J SconeOcc jobs of different sizes in one launch sequence (mcr_scone_occ_forward_ragged).  pc_global [J,Lg,3],
    global_len int32 [J], pc_scales: 3 ragged clouds [sum M_s,3], scale_offsets: 3 int64 [J+1], x [T,3], view_harmonics [T,64],
    row_job int32 [T], knn_blocks int32 [n_blocks,4] -> out [T,1].
    phase 1 / 2 (mcr_scone_occ_forward_ragged_phase): the part that needs no hidden draw / the rest, as two calls on the same stream
    (phase 1: pc_global / global_len may be None with Lg given, pc_scales[1:], scale_offsets[1:] are ignored; returns None)."""
    x, view_harmonics = _req(x, "x"), _req(view_harmonics, "view_harmonics")
    row_job = _req(row_job, "row_job", torch.int32)
    knn_blocks = _req(knn_blocks, "knn_blocks", torch.int32)
    late = phase != 1
    pc_scales = [_req(p, "pc_scale") for p in (pc_scales if late else [pc_scales[0]] * 3)]
    scale_offsets = [_req(o, "scale_offsets", torch.int64) for o in (scale_offsets if late else [scale_offsets[0]] * 3)]
    J = scale_offsets[0].numel() - 1
    if late:
        pc_global, global_len = _req(pc_global, "pc_global"), _req(global_len, "global_len", torch.int32)
        Lg = pc_global.shape[1]
        if pc_global.shape[0] != J or global_len.numel() != J:
            raise ValueError("scone_occ_forward_ragged: pc_global [J,Lg,3], global_len [J]")
    T = x.shape[0]
    if x.shape != (T, 3) or view_harmonics.shape != (T, 64) or row_job.numel() != T:
        raise ValueError("scone_occ_forward_ragged: x [T,3], view_harmonics [T,64], row_job [T]")
    if any(o.numel() != J + 1 for o in scale_offsets) or knn_blocks.dim() != 2 or knn_blocks.shape[1] != 4:
        raise ValueError("scone_occ_forward_ragged: scale_offsets must be [J+1], knn_blocks [n_blocks,4]")
    L_ = lib()
    if late and out is None:
        out = torch.empty((T, 1), dtype=torch.float32, device=x.device)
    ws = _workspace(x.device, L_.mcr_scone_occ_ragged_workspace_bytes(c_i64(J), c_i64(T), c_i64(Lg)), arena)
    _bump_epoch(x.device, arena)
    sc_ptrs = (ctypes.c_void_p * 3)(*[p.data_ptr() for p in pc_scales])
    off_ptrs = (ctypes.c_void_p * 3)(*[o.data_ptr() for o in scale_offsets])
    blobs = (ctypes.c_void_p * 3)(*[_req(b, "local_blob").data_ptr() for b in local_blobs])
    with torch.cuda.device(x.device):
        check(_net(L_).mcr_scone_occ_forward_ragged_phase(_p(pc_global) if late else c_vp(0), _p(global_len) if late else c_vp(0), c_i64(Lg), sc_ptrs,
                                                    off_ptrs, _p(x), _p(view_harmonics), _p(row_job), _p(knn_blocks), c_i64(knn_blocks.shape[0]),
                                                    _p(out) if late else c_vp(0), c_i64(J), c_i64(T),
                                                    _ptr_table(weights), c_int(_n_weights(weights)), blobs,
                                                    head_planes[1] if head_planes is not None else None,
                                                    head_planes[2] if head_planes is not None else None,
                                                    _p(_req(range_flag, "range_flag", torch.int32)) if range_flag is not None else c_vp(0),
                                                    _p(ws), c_size(ws.numel()), c_int(int(phase)), _stream()), "mcr_scone_occ_forward_ragged")
    return out


# ---- glue -----------------------------------------------------------------------------------
def view_state(pts, X_view, n_elev, n_azim):
    """[n_clouds, seq_len, n_elev*n_azim] fp32 0/1; replaces compute_view_state (scone_utils.py:799-860).
    X_view [n_view,3] (shared by all clouds, as the reference) or [n_clouds,n_view,3] (every cloud of a scene batch against its
    own past camera positions)."""
    pts, X_view = _req(pts, "pts"), _req(X_view, "X_view")
    B, Q, d = pts.shape
    out = torch.empty((B, Q, n_elev * n_azim), dtype=torch.float32, device=pts.device)
    with torch.cuda.device(pts.device):
        if X_view.dim() == 3:
            if X_view.shape[0] != B:
                raise ValueError(f"per-cloud X_view must be [n_clouds={B}, n_view, 3], got {tuple(X_view.shape)}")
            check(lib().mcr_view_state_batched(_p(pts), c_int(d), _p(X_view), _p(out), c_i64(B), c_i64(Q), c_int(X_view.shape[1]),
                                               c_int(n_elev), c_int(n_azim), c_vp(0), c_int(0), _stream()), "mcr_view_state_batched")
        else:
            check(lib().mcr_view_state(_p(pts), c_int(d), _p(X_view), _p(out), c_i64(B * Q), c_int(X_view.shape[0]), c_int(n_elev),
                                       c_int(n_azim), _stream()), "mcr_view_state")
    return out


def view_state_update_(table, rows, pts, X_view, n_elev, n_azim):
    """In place: table [P_total, n_elev*n_azim] (0/1 fp32) gets, in row rows[i], the bins of the directions from pts[i] to every
    X_view position OR'ed in: `view_states[mask] += compute_view_state(...)` followed by torch.heaviside(., 0)
    (macarons_utils.py:2867-2877) as one launch.  rows: int64 device tensor [n] (None: row i <- point i)."""
    table, pts, X_view = _req(table, "view_states"), _req(pts, "pts"), _req(X_view, "X_view")
    n, d = pts.shape
    if table.shape[-1] != n_elev * n_azim:
        raise ValueError("view-state table width does not match the lattice")
    if n == 0:
        return table
    r = _req(rows, "rows", torch.int64) if rows is not None else None
    if r is not None and r.numel() != n:
        raise ValueError("rows must hold one table row per point")
    with torch.cuda.device(pts.device):
        check(lib().mcr_view_state_batched(_p(pts), c_int(d), _p(X_view), _p(table), c_i64(1), c_i64(n), c_int(X_view.shape[0]),
                                           c_int(n_elev), c_int(n_azim), _p(r) if r is not None else c_vp(0), c_int(1), _stream()),
              "mcr_view_state_batched")
    return table


def transform_points_batched_(pts, M_view, center, inv_diag, cloud_of=None):
    """In place, all clouds in one launch: pts [K,S,d] (or, with cloud_of int32 [n], ragged rows [n,d]); M_view [K,4,4],
    center [K,3], inv_diag [K]:  pts[..., :3] <- (([x y z 1] M_view[c])[:3] - center[c]) * inv_diag[c]."""
    pts = _req(pts, "pts")
    M_view, center, inv_diag = _req(M_view, "M_view"), _req(center, "center"), _req(inv_diag, "inv_diag")
    K = M_view.shape[0]
    d = pts.shape[-1]
    if cloud_of is not None:
        cloud_of = _req(cloud_of, "cloud_of", torch.int32)
        n, per = pts.shape[0], 0
    else:
        n, per = 0, pts.shape[1]
        if pts.shape[0] != K:
            raise ValueError("transform_points_batched_: one matrix per cloud")
    if pts.numel() == 0:
        return pts
    with torch.cuda.device(pts.device):
        check(lib().mcr_transform_points_batched(_p(pts), c_int(d), c_i64(K), c_i64(per), _p(M_view), _p(center), _p(inv_diag),
                                                 _p(cloud_of) if cloud_of is not None else c_vp(0), c_i64(n), _stream()),
              "mcr_transform_points_batched")
    return pts


def sample_proxy_batched(X, preds, view_harmonics, u, min_occ):
    """B clouds at once, no host sync: X [B,P,3], preds [B,P], view_harmonics [B,P,64] or None, u [B,n] ->
    (res [B,n,4], res_harmonics [B,n,64] | None, inverse [B,n] int64, uniq [B,n] int64, n_unique int32 [B], volume fp64 [B]);
    rows beyond n_unique[b] are zero.  Cloud b is sampled exactly as sample_proxy(X[b], preds[b], ...) would."""
    X, preds, u = _req(X, "X"), _req(preds, "preds"), _req(u, "samples")
    vh = _req(view_harmonics, "view_harmonics") if view_harmonics is not None else None
    shared = X.dim() == 2                          # X [P,3] / view_harmonics [P,64] common to the B distributions (preds [B,P])
    B, P = (preds.shape[0], X.shape[0]) if shared else (X.shape[0], X.shape[1])
    n = u.shape[-1]
    if preds.numel() != B * P or u.numel() != B * n:
        raise ValueError("sample_proxy_batched: preds must be [B,P] and samples [B,n]")
    dev = X.device
    res = torch.empty((B, n, 4), dtype=torch.float32, device=dev)
    resh = torch.empty((B, n, 64), dtype=torch.float32, device=dev) if vh is not None else None
    uniq = torch.empty((B, n), dtype=torch.int64, device=dev)
    inv = torch.empty((B, n), dtype=torch.int64, device=dev)
    nu = torch.empty(B, dtype=torch.int32, device=dev)
    vol = torch.empty(B, dtype=torch.float64, device=dev)
    L_ = lib()
    ws = _workspace(dev, L_.mcr_sample_proxy_batched_workspace_bytes(c_i64(B), c_i64(P), c_int(n)))
    with torch.cuda.device(dev):
        fn = L_.mcr_sample_proxy_shared if shared else L_.mcr_sample_proxy_batched
        check(fn(_p(X), _p(preds), c_i64(1), _p(vh) if vh is not None else c_vp(0), c_i64(B), c_i64(P),
                                          c_f32(float(min_occ)), _p(u), c_int(n), _p(res), _p(resh) if resh is not None else c_vp(0),
                                          _p(uniq), _p(inv), _p(nu), _p(vol), _p(ws), c_size(ws.numel()), _stream()),
              "mcr_sample_proxy_batched")
    return res, resh, inv, uniq, nu, vol


def sample_proxy(X, preds, view_harmonics, u, min_occ, return_volume=False, padded=False):
    """(res [n_u,4], res_harmonics [n_u,64], inverse [n_sample] int64, unique original indices [n_u] int64[, volume]);
    replaces sample_proxy_points (scone_utils.py:1030-1061).  One host sync to read n_u (torch.unique syncs too) -- unless
    padded=True: then res / res_harmonics / uniq keep their n_sample rows (zeros beyond n_u) and the count comes back as an
    int32 device tensor [1] in place of the slicing: (res, res_harmonics, inverse, uniq, n_unique[, volume]), no host sync."""
    X, preds, u = _req(X, "X"), _req(preds, "preds"), _req(u, "samples")
    vh = _req(view_harmonics, "view_harmonics") if view_harmonics is not None else None    # None: res_harmonics comes back as None
    P = X.shape[0]
    n = u.numel()
    dev = X.device
    res = torch.empty((n, 4), dtype=torch.float32, device=dev)
    resh = torch.empty((n, 64), dtype=torch.float32, device=dev) if vh is not None else None
    uniq = torch.empty(n, dtype=torch.int64, device=dev)
    inv = torch.empty(n, dtype=torch.int64, device=dev)
    nu = torch.zeros(1, dtype=torch.int32, device=dev)
    vol = torch.zeros(1, dtype=torch.float64, device=dev)
    L_ = lib()
    ws = _workspace(dev, L_.mcr_sample_proxy_workspace_bytes(c_i64(P), c_int(n)))
    with torch.cuda.device(dev):
        check(L_.mcr_sample_proxy(_p(X), _p(preds), c_i64(1), _p(vh) if vh is not None else c_vp(0), c_i64(P),
                                  c_f32(float(min_occ)), _p(u), c_int(n), _p(res), _p(resh) if resh is not None else c_vp(0),
                                  _p(uniq), _p(inv), _p(nu), _p(vol), _p(ws), c_size(ws.numel()), _stream()),
              "mcr_sample_proxy")
    if padded:
        return (res, resh, inv, uniq, nu, vol) if return_volume else (res, resh, inv, uniq, nu)
    k = int(nu.item())
    if return_volume:
        return res[:k], resh[:k], inv, uniq[:k], vol
    return res[:k], resh[:k], inv, uniq[:k]


def points_in_fov(pts, cameras):
    """pts [P,3], cameras [n_cam,40] (layout in include/macarons_hip.h) -> bool mask [n_cam,P];
    replaces Camera.get_points_in_fov (macarons_utils.py:2400-2435)."""
    pts, cameras = _req(pts, "pts"), _req(cameras, "cameras")
    P, C = pts.shape[0], cameras.shape[0]
    if cameras.shape[1] != 40 or pts.shape[1] != 3:
        raise ValueError("cameras must be [n_cam,40] and pts [P,3]")
    mask = torch.empty((C, P), dtype=torch.uint8, device=pts.device)
    with torch.cuda.device(pts.device):
        check(lib().mcr_points_in_fov(_p(pts), c_i64(P), _p(cameras), c_int(C), _p(mask), _stream()), "mcr_points_in_fov")
    return mask.bool()


def coverage_gain_multiple(pts, harmonics, cams, n_cam, use_sigmoid=True):
    """(gains [B, C^n], idx [C^n, n]); replaces SconeVis.compute_coverage_gain_multiple (SconeVis.py:254-303)."""
    vis = sh_visibilities(pts, harmonics, cams, use_sigmoid)
    B, C, N = vis.shape
    out = torch.empty((B, C ** n_cam), dtype=torch.float32, device=vis.device)
    with torch.cuda.device(vis.device):
        check(lib().mcr_coverage_gain_multiple(_p(vis), _p(out), c_i64(B), c_i64(C), c_i64(N), c_int(n_cam), _stream()),
              "mcr_coverage_gain_multiple")
    single = torch.arange(0, C)
    n_idx = torch.cartesian_prod(*([single] * n_cam))                  # SconeVis.py:292-296 (index table, host)
    return out, n_idx


_pinned_pool = {}        # (device index) -> list of [pinned uint8 buffer, event of its last copy]


_NATIVE_H2D = []


def _native_h2d():
    """Is torch.ops.macarons.h2d there (the C++ extension built)?  MCR_NATIVE_H2D=0: the Python restatement below (A/B)."""
    if not _NATIVE_H2D:
        ok = os.environ.get("MCR_NATIVE_H2D", "1") != "0"
        if ok:
            try:
                from . import torch_ops  # noqa: F401
                ok = hasattr(torch.ops.macarons, "h2d")
            except Exception:
                ok = False
        _NATIVE_H2D.append(ok)
    return _NATIVE_H2D[0]


def h2d(data, dtype, device):
    """Host data (list / numpy array / CPU tensor) -> device tensor WITHOUT stalling the host: a `.to(device)` from pageable memory
    is a stream-ordered blocking copy, i.e. the host waits for every kernel queued before it (the glue of a MACARONS decision did
    that ~15 times per decision).  The data is staged in a small pool of re-used pinned buffers (power-of-two sizes; a buffer is
    taken again once the event behind its last copy has fired) and copied asynchronously.  (torch's own pin_memory() allocates a
    new pinned block for every new size: 80 ms for the 1 MB index array of a ragged occupancy pass.)"""
    t = data if torch.is_tensor(data) else torch.as_tensor(data)
    t = t.to(dtype) if t.dtype != dtype else t
    device = torch.device(device)
    if device.type != "cuda" or t.device.type != "cpu":
        return t.to(device)
    if _native_h2d():                                   # the same steps from C++ (libmacarons_torch.so): one dispatcher call
        return torch.ops.macarons.h2d(t, device.index if device.index is not None else torch.cuda.current_device())
    t = t.contiguous()
    nbytes = t.numel() * t.element_size()
    if nbytes == 0:
        return torch.empty(t.shape, dtype=dtype, device=device)
    pool = _pinned_pool.setdefault(device.index if device.index is not None else torch.cuda.current_device(), [])
    slot = None
    for ent in pool:
        if ent[0].numel() >= nbytes and (slot is None or ent[0].numel() < slot[0].numel()) and ent[1].query():
            slot = ent
    if slot is None:
        cap = 1 << max(12, (nbytes - 1).bit_length())
        slot = [torch.empty(cap, dtype=torch.uint8).pin_memory(), torch.cuda.Event()]
        pool.append(slot)
    stage = slot[0][:nbytes].view(dtype).view(t.shape)
    stage.copy_(t)
    with torch.cuda.device(device):
        out = stage.to(device, non_blocking=True)
        slot[1].record(torch.cuda.current_stream(device))
    return out


def gather_columns(x, idx):
    """x [..., V] fp32, idx [V] integer (values in [0, V)) -> x[..., idx] (torch.gather along the last dim with one index row).
    A host idx is range-checked on the host and uploaded without a stall; a device idx is trusted (checking it would read it
    back)."""
    x = _req(x, "x")
    V = x.shape[-1]
    if idx.device.type == "cpu":
        if idx.numel() != V or int(idx.min()) < 0 or int(idx.max()) >= V:
            raise ValueError("gather_columns: idx must hold V indices in [0, V)")
        idx = h2d(idx, torch.int32, x.device)
    else:
        if idx.numel() != V:
            raise ValueError("gather_columns: idx must hold V indices in [0, V)")
        idx = idx.to(device=x.device, dtype=torch.int32).contiguous()
    out = torch.empty_like(x)
    with torch.cuda.device(x.device):
        check(lib().mcr_gather_columns(_p(x), _p(idx), _p(out), c_i64(x.numel() // V), c_int(V), _stream()), "mcr_gather_columns")
    return out


def filter_proxy_mask(X, pc, proj, filter_tol):
    """X [P,3], pc [M,3], proj [n_view,4,4] (row-vector full-projection matrices) -> (mask bool [P], bounds [n_view,4])."""
    X, pc, proj = _req(X, "X"), _req(pc, "pc"), _req(proj, "proj")
    if X.dim() != 2 or pc.dim() != 2 or X.shape[1] != 3 or pc.shape[1] != 3:
        raise NameError("Wrong shapes! X must have shape (n_proxy_points, 3) and pc must have shape (N, 3).")
    n_view = proj.shape[0]
    if tuple(proj.shape[1:]) != (4, 4):
        raise ValueError("proj must be [n_view,4,4]")
    mask = torch.empty(X.shape[0], dtype=torch.uint8, device=X.device)
    bounds = torch.empty((n_view, 4), dtype=torch.float32, device=X.device)
    with torch.cuda.device(X.device):
        check(lib().mcr_filter_proxy_points(_p(X), c_i64(X.shape[0]), _p(pc), c_i64(pc.shape[0]), _p(proj), c_int(n_view),
                                            c_f32(float(filter_tol)), _p(bounds), _p(mask), _stream()), "mcr_filter_proxy_points")
    return mask.bool(), bounds


def best_record(gains, idx_offset=0, out=None):
    """gains [B,C] -> records [B,2] fp32 = (max, idx_offset + first arg-max): torch.max(gains, dim=1) as one 8-byte record."""
    gains = _req(gains, "gains")
    B, C = gains.shape
    if out is None:
        out = torch.empty((B, 2), dtype=torch.float32, device=gains.device)
    with torch.cuda.device(gains.device):
        check(lib().mcr_best_record(_p(gains), c_i64(B), c_i64(C), c_i64(int(idx_offset)), _p(out), _stream()), "mcr_best_record")
    return out


def nbv_decide(gains, n_unique=None, range_flag=None):
    """gains [B,C] (modified in place: NaN rows where n_unique < 1) -> (max_gain [B], nbv_idx [B] int64, record float64 [1 + 2B] =
    (range flag, indices, maxima)); mcr_nbv_decide."""
    gains = _req(gains, "gains")
    B, C = gains.shape
    dev = gains.device
    max_gain = torch.empty(B, dtype=torch.float32, device=dev)
    nbv_idx = torch.empty(B, dtype=torch.int64, device=dev)
    record = torch.empty(1 + 2 * B, dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        check(lib().mcr_nbv_decide(_p(gains), c_i64(B), c_i64(C),
                                   _p(_req(n_unique, "n_unique", torch.int32)) if n_unique is not None else c_vp(0),
                                   _p(_req(range_flag, "range_flag", torch.int32)) if range_flag is not None else c_vp(0),
                                   _p(max_gain), _p(nbv_idx), _p(record), _stream()), "mcr_nbv_decide")
    return max_gain, nbv_idx, record


def best_merge(records, out_vals=None, out_idx=None):
    """records [world,B,2] -> (vals [B] fp32, idx [B] int64) of the global arg-max, ties to the lowest index."""
    records = _req(records, "records")
    world, B = records.shape[0], records.shape[1]
    vals = out_vals if out_vals is not None else torch.empty(B, dtype=torch.float32, device=records.device)
    idx = out_idx if out_idx is not None else torch.empty(B, dtype=torch.int64, device=records.device)
    with torch.cuda.device(records.device):
        check(lib().mcr_best_merge(_p(records), c_int(world), c_i64(B), _p(vals), _p(idx), _stream()), "mcr_best_merge")
    return vals, idx


def fov_mask_occ(mask, occ):
    """mask [n_cam,P] bool/uint8, occ [P] -> [n_cam,P] occupancy zeroed outside each frustum."""
    occ = _req(occ, "occ")
    m = mask.to(torch.uint8).contiguous()
    C, P = m.shape
    out = torch.empty((C, P), dtype=torch.float32, device=occ.device)
    with torch.cuda.device(occ.device):
        check(lib().mcr_fov_mask_occ(_p(m), _p(occ), c_i64(1), _p(out), c_i64(P), c_int(C), _stream()), "mcr_fov_mask_occ")
    return out


def transform_points_(pts, M_view, center, inv_diag):
    """In place: pts[:, :3] <- (([x y z 1] M_view)[:3] - center) * inv_diag."""
    pts = _req(pts, "pts")
    n, d = pts.shape
    with torch.cuda.device(pts.device):
        check(lib().mcr_transform_points(_p(pts), c_int(d), c_i64(n), _p(_req(M_view, "M_view")), _p(_req(center, "center")),
                                         c_f32(float(inv_diag)), _stream()), "mcr_transform_points")
    return pts


def macarons_gain_(vis, pts_world, cam_world, volume, distance_th, smooth=False):
    """vis [B,N] (scaled in place by the distance factor), pts_world [B,N,>=3], cam_world [B,3], volume [B] -> gains [B].
    smooth=False: min(1, (th/d)^2); smooth=True: 1/(1+(d/th)^2)."""
    vis, pts_world, cam_world, volume = _req(vis, "vis"), _req(pts_world, "pts_world"), _req(cam_world, "cam_world"), _req(volume, "volume")
    B, N = vis.shape
    gains = torch.empty(B, dtype=torch.float32, device=vis.device)
    with torch.cuda.device(vis.device):
        check(lib().mcr_macarons_gain(_p(vis), _p(pts_world), c_int(pts_world.shape[-1]), _p(cam_world), _p(volume),
                                      c_f32(float(distance_th)), c_int(int(bool(smooth))), c_i64(B), c_i64(N), _p(gains), _stream()),
              "mcr_macarons_gain")
    return gains


# ---- scene-side bookkeeping ---------------------------------------------------------------
def min_dist_segmented(A, a_offsets, B, b_offsets, max_a=None):
    """fp64 distance of every A point to the nearest B point of its segment (grid cell); +inf for empty B segments.
    Replaces torch.min(torch.cdist(a.double(), b.double())) (macarons_utils.py:2566, 3022, 3049)."""
    A, B = _req(A, "A"), _req(B, "B")
    a_off = _req(a_offsets, "a_offsets", torch.int64)
    b_off = _req(b_offsets, "b_offsets", torch.int64)
    nseg = a_off.numel() - 1
    out = torch.empty(A.shape[0], dtype=torch.float64, device=A.device)
    if max_a is None:                                 # (an upper bound of the largest A segment avoids this read-back: it only sizes the grid)
        max_a = int((a_off[1:] - a_off[:-1]).max().item()) if nseg > 0 else 0
    if max_a == 0 or A.shape[0] == 0:
        return out
    if B.shape[0] == 0:                                  # every segment's B is empty
        return out.fill_(float("inf"))
    with torch.cuda.device(A.device):
        check(lib().mcr_min_dist_segmented(_p(A), _p(a_off), _p(B), _p(b_off), c_i64(nseg), c_i64(max_a), _p(out), _stream()),
              "mcr_min_dist_segmented")
    return out


def unproject_depth(depth, cameras):
    """depth [n_cam,H,W(,1)], cameras [n_cam,18] (Minv[16], k22, k32) -> world points [n_cam, H*W, 3]."""
    depth, cameras = _req(depth, "depth"), _req(cameras, "cameras")
    n, H, W = depth.shape[0], depth.shape[1], depth.shape[2]
    out = torch.empty((n, H * W, 3), dtype=torch.float32, device=depth.device)
    with torch.cuda.device(depth.device):
        check(lib().mcr_unproject_depth(_p(depth), c_int(H), c_int(W), _p(cameras), c_i64(n), _p(out), _stream()),
              "mcr_unproject_depth")
    return out


def signed_distance_to_depth(pts, camera, depth, mask=None, fill=0.0):
    """pts [n,3], camera: >= 32 floats (M_view[16] | M_full_projection[16], the head of a points_in_fov record), depth [H,W],
    mask [H,W] bool/uint8 or None -> signed distances [n]; replaces Camera.get_signed_distance_to_depth_maps
    (macarons_utils.py:2451-2500) for one camera (pixels outside the mask count as `fill` = 1.1 zfar)."""
    pts, camera, depth = _req(pts, "pts"), _req(camera, "camera"), _req(depth, "depth")
    H, W = depth.shape[-2], depth.shape[-1]
    n = pts.shape[0]
    out = torch.empty(n, dtype=torch.float32, device=pts.device)
    if n == 0:
        return out
    m = mask.to(torch.uint8).contiguous() if mask is not None else None
    with torch.cuda.device(pts.device):
        check(lib().mcr_signed_distance_to_depth(_p(pts), c_i64(n), _p(camera), _p(depth), _p(m) if m is not None else c_vp(0),
                                                 c_int(H), c_int(W), c_f32(float(fill)), _p(out), _stream()),
              "mcr_signed_distance_to_depth")
    return out


def proxy_scene_update_(proxy_points, fov_mask, camera, depth, depth_mask, fill, X_cam, distance_to_surface, tol, score_threshold,
                        n_elev, n_azim, view_states, n_inside, n_behind, supervision_occ, out_of_field, return_sgn=False):
    """The proxy-point bookkeeping of one MACARONS step (testers/scene.py:402-418) fused in one pass, in place on the state
    tensors (see mcr_proxy_scene_update).  fov_mask [P] bool/uint8; returns the signed distances [P] if asked (0 outside the mask)."""
    pp = _req(proxy_points, "proxy_points")
    P = pp.shape[0]
    fm = fov_mask.to(torch.uint8).contiguous()
    dm = depth_mask.to(torch.uint8).contiguous() if depth_mask is not None else None
    depth = _req(depth, "depth")
    H, W = depth.shape[-2], depth.shape[-1]
    for name, t in (("view_states", view_states), ("n_inside", n_inside), ("n_behind", n_behind), ("supervision_occ", supervision_occ),
                    ("out_of_field", out_of_field)):
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise ValueError(f"{name} must be a contiguous fp32 device tensor (updated in place)")
    sgn = torch.zeros(P, dtype=torch.float32, device=pp.device) if return_sgn else None
    with torch.cuda.device(pp.device):
        check(lib().mcr_proxy_scene_update(_p(pp), c_i64(P), _p(fm), _p(_req(camera, "camera")), _p(depth),
                                           _p(dm) if dm is not None else c_vp(0), c_int(H), c_int(W), c_f32(float(fill)),
                                           _p(_req(X_cam, "X_cam")), c_f32(float(distance_to_surface)), c_f32(float(tol)),
                                           c_f32(float(score_threshold)), c_int(n_elev), c_int(n_azim), _p(view_states), _p(n_inside),
                                           _p(n_behind), _p(supervision_occ), _p(out_of_field), _p(sgn) if sgn is not None else c_vp(0),
                                           _stream()), "mcr_proxy_scene_update")
    return sgn


# ---- scene-grid bookkeeping, fused (Scene.fill_cells / the cell lookup of the occupancy field) -----------------------------------------
def cell_keys(pts, grid_consts, grid, lo_tab=None, hi_tab=None, valid=None):
    """pts [N,3] -> int32 [N]: linear id of the cell each point falls in (upstream's floor rule); with the cells' bounds lo_tab / hi_tab
    [n_cells,3] also Cell.fill's tests: n_cells for a point outside the scene box, not strictly inside its cell, or with valid[i] == 0.
    grid_consts: fp32 device tensor [9] = x_min, x_max, step; grid = (grid_l, grid_w, grid_h)."""
    pts, gc = _req(pts, "pts"), _req(grid_consts, "grid_consts")
    N = pts.shape[0]
    key = torch.empty(N, dtype=torch.int32, device=pts.device)
    if N == 0:
        return key
    box = lo_tab is not None
    v = valid.to(torch.uint8).contiguous() if valid is not None else None
    with torch.cuda.device(pts.device):
        check(lib().mcr_cell_keys(_p(pts), c_i64(N), _p(v) if v is not None else c_vp(0), _p(gc), c_int(grid[0]), c_int(grid[1]), c_int(grid[2]),
                                  _p(_req(lo_tab, "lo_tab")) if box else c_vp(0), _p(_req(hi_tab, "hi_tab")) if box else c_vp(0),
                                  c_int(int(box)), _p(key), _stream()), "mcr_cell_keys")
    return key


def key_histogram(key, nk):
    """key int32 [N] -> (counts int64 [nk+1], exclusive offsets int64 [nk+2])."""
    key = _req(key, "key", torch.int32)
    if nk > 1023:                                   # grids beyond the one-block kernel's LDS table (> 1023 cells): device ops, same result
        counts = torch.zeros(nk + 1, dtype=torch.int64, device=key.device).scatter_add_(0, key.long().clamp_(0, nk), torch.ones_like(key, dtype=torch.int64))
        offsets = torch.zeros(nk + 2, dtype=torch.int64, device=key.device)
        offsets[1:] = torch.cumsum(counts, 0)
        return counts, offsets
    counts = torch.empty(nk + 1, dtype=torch.int64, device=key.device)
    offsets = torch.empty(nk + 2, dtype=torch.int64, device=key.device)
    with torch.cuda.device(key.device):
        check(lib().mcr_key_histogram(_p(key), c_i64(key.numel()), c_int(nk), _p(counts), _p(offsets), _stream()), "mcr_key_histogram")
    return counts, offsets


def admit_keys(d, key_s, cand, resolution, n_point_min, nk):
    """Cell.fill's admission on the sorted candidates (mcr_admit_keys) -> key2 int32 [N]."""
    d, key_s, cand = _req(d, "d", torch.float64), _req(key_s, "key_s", torch.int32), _req(cand, "cand", torch.int64)
    key2 = torch.empty_like(key_s)
    if key_s.numel() == 0:
        return key2
    with torch.cuda.device(d.device):
        check(lib().mcr_admit_keys(_p(d), _p(key_s), _p(cand), c_i64(key_s.numel()), ctypes.c_double(float(resolution)), c_i64(int(n_point_min)),
                                   c_int(nk), _p(key2), _stream()), "mcr_admit_keys")
    return key2


# ---- one MACARONS decision without the host glue (scene.hip): each phase = a handful of launches behind one C call ----------------
def group_by_key(key, nk):
    """key int32 [N] in 0 .. nk (anything else counts as nk) -> (order int32 [N] = torch.sort(key, stable=True).indices,
    counts int64 [nk+1], exclusive offsets int64 [nk+2]); a three-launch counting sort (nk <= 1023)."""
    key = _req(key, "key", torch.int32)
    N, dev = key.numel(), key.device
    order = torch.empty(N, dtype=torch.int32, device=dev)
    co = torch.empty(2 * nk + 3, dtype=torch.int64, device=dev)
    L_ = lib()
    ws = _workspace(dev, max(int(L_.mcr_group_by_key_workspace_bytes(c_i64(N), c_int(nk))), 4))
    with torch.cuda.device(dev):
        check(L_.mcr_group_by_key(_p(key), c_i64(N), c_int(nk), _p(order), _p(co), c_vp(co.data_ptr() + 8 * (nk + 1)), _p(ws),
                                  c_size(ws.numel()), _stream()), "mcr_group_by_key")
    return order, co[:nk + 1], co[nk + 1:]


class PendingFill:
    """What mcr_scene_fill_begin left on the device: the candidates grouped by cell and the admitted ones among them; `counts`
    (int64 [4 nk + 6] = cand | a_off | adm | adm_off) is what the host reads back before it draws the cells' permutations."""
    __slots__ = ("pts", "features", "N", "nk", "key", "order", "dmin", "key2", "order2", "counts")


def scene_fill_begin(pts, valid, grid_consts, grid, lo_tab, hi_tab, store_pts, store_off, resolution, n_point_min, features=None):
    """The device part of Scene.fill_cells (mcr_scene_fill_begin): nothing returns to the host.  pts [N,3]; valid bool/uint8 [N] or
    None; store_pts [n_store,3] = every cell's stored points (cells in linear order), store_off int64 device [n_cells+1]."""
    pts, gc = _req(pts, "pts"), _req(grid_consts, "grid_consts")
    N, dev = pts.shape[0], pts.device
    nk = grid[0] * grid[1] * grid[2]
    h = PendingFill()
    h.pts, h.N, h.nk = pts, N, nk
    h.features = None if features is None else features.to(torch.float32).contiguous()
    ib = torch.empty(4 * N, dtype=torch.int32, device=dev)
    h.key, h.order, h.key2, h.order2 = ib[:N], ib[N:2 * N], ib[2 * N:3 * N], ib[3 * N:]
    h.dmin = torch.empty(N, dtype=torch.float64, device=dev)
    h.counts = torch.empty(4 * nk + 6, dtype=torch.int64, device=dev)
    v = valid.to(torch.uint8).contiguous() if valid is not None else None
    L_ = lib()
    ws = _workspace(dev, max(int(L_.mcr_scene_fill_workspace_bytes(c_i64(N), c_int(nk))), 4))
    with torch.cuda.device(dev):
        check(L_.mcr_scene_fill_begin(_p(pts), c_i64(N), _p(v) if v is not None else c_vp(0), _p(gc), c_int(grid[0]), c_int(grid[1]),
                                      c_int(grid[2]), _p(_req(lo_tab, "lo_tab")), _p(_req(hi_tab, "hi_tab")),
                                      _p(store_pts) if store_pts is not None and store_pts.numel() else c_vp(0),
                                      _p(_req(store_off, "store_off", torch.int64)), ctypes.c_double(float(resolution)), c_i64(int(n_point_min)),
                                      _p(h.key), _p(h.order), _p(h.dmin), _p(h.key2), _p(h.order2), _p(h.counts), _p(ws), c_size(ws.numel()),
                                      _stream()), "mcr_scene_fill_begin")
    return h


def scene_fill_gather(g, h, store_pts, store_fts, n_store, F):
    """New store rows = rows g of [old store | admitted candidates of the pending fill h in cell order] -> (pts [n,3], features [n,F] | None)."""
    g = _req(g, "g", torch.int64)
    n, dev = g.numel(), g.device
    new_pts = torch.empty((n, 3), dtype=torch.float32, device=dev)
    new_fts = torch.empty((n, F), dtype=torch.float32, device=dev) if F > 0 else None
    with torch.cuda.device(dev):
        check(lib().mcr_scene_fill_gather(_p(g), c_i64(n), _p(store_pts) if n_store else c_vp(0),
                                          _p(store_fts) if (n_store and store_fts is not None) else c_vp(0), c_i64(n_store), c_int(F), _p(h.pts),
                                          _p(h.features) if h.features is not None else c_vp(0), _p(h.order), _p(h.order2), _p(new_pts),
                                          _p(new_fts) if new_fts is not None else c_vp(0), _stream()), "mcr_scene_fill_gather")
    return new_pts, new_fts


def scene_fill_gather_perm(pm_tab, n_pm, n_cells, n_new, h, store_pts, store_fts, n_store, F):
    """As scene_fill_gather with the row map evaluated on the device: pm_tab = ONE uploaded int64 buffer holding the five (n_cells + 1)
    tables of mcr_scene_fill_gather_perm followed (as raw bytes) by the n_pm int32 permutation entries."""
    dev = pm_tab.device
    new_pts = torch.empty((n_new, 3), dtype=torch.float32, device=dev)
    new_fts = torch.empty((n_new, F), dtype=torch.float32, device=dev) if F > 0 else None
    base = pm_tab.data_ptr()
    with torch.cuda.device(dev):
        check(lib().mcr_scene_fill_gather_perm(c_vp(base + 40 * (n_cells + 1)), c_vp(base), c_int(n_cells), c_i64(n_new),
                                               _p(store_pts) if n_store else c_vp(0),
                                               _p(store_fts) if (n_store and store_fts is not None) else c_vp(0), c_i64(n_store), c_int(F),
                                               _p(h.pts), _p(h.features) if h.features is not None else c_vp(0), _p(h.order), _p(h.order2),
                                               _p(new_pts), _p(new_fts) if new_fts is not None else c_vp(0), _stream()),
              "mcr_scene_fill_gather_perm")
    return new_pts, new_fts


class FieldSelection:
    """What mcr_field_select left on the device; `counts` (int64 [3 nk + 9] = visit | sel_counts | sel_off | oof_counts | oof_off)."""
    __slots__ = ("P", "nk", "stored_cell", "rows_order", "oof_order", "counts")


def field_select(proxy_points, supervision_occ, out_of_field, proxy_proba, store_fts, n_store, store_off, grid_consts, grid, use_mask,
                 pending=None):
    """Selection + grouping of the occupancy-field pass (mcr_field_select); proxy_proba is updated in place (:1431)."""
    pp = _req(proxy_points, "proxy_points")
    P, dev = pp.shape[0], pp.device
    nk = grid[0] * grid[1] * grid[2]
    for name, t in (("proxy_supervision_occ", supervision_occ), ("out_of_field", out_of_field), ("proxy_proba", proxy_proba)):
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.numel() == P):
            raise ValueError(f"{name} must be a contiguous fp32 device tensor with one entry per proxy point")
    F = store_fts.shape[1] if (store_fts is not None and store_fts.dim() == 2) else 1
    if pending is not None and (pending.features is None or pending.features.shape[1] != F):
        raise ValueError("field_select: the pending fill carries no features (the proxy indices)")
    s = FieldSelection()
    s.P, s.nk = P, nk
    ib = torch.empty(5 * P, dtype=torch.int32, device=dev)
    s.stored_cell, key_sel, key_oof, s.rows_order, s.oof_order = (ib[k * P:(k + 1) * P] for k in range(5))
    s.counts = torch.empty(3 * nk + 9, dtype=torch.int64, device=dev)
    L_ = lib()
    ws = _workspace(dev, max(int(L_.mcr_field_select_workspace_bytes(c_i64(P), c_int(nk))), 4))
    pe = pending
    with torch.cuda.device(dev):
        check(L_.mcr_field_select(_p(pp), c_i64(P), _p(supervision_occ), _p(out_of_field), _p(proxy_proba),
                                  _p(store_fts) if n_store else c_vp(0), c_int(F), c_i64(n_store), _p(_req(store_off, "store_off", torch.int64)),
                                  _p(pe.features) if pe is not None else c_vp(0), _p(pe.order) if pe is not None else c_vp(0),
                                  _p(pe.order2) if pe is not None else c_vp(0), _p(pe.key2) if pe is not None else c_vp(0),
                                  c_vp(pe.counts.data_ptr() + 8 * (3 * nk + 4)) if pe is not None else c_vp(0), c_i64(pe.N if pe is not None else 0),
                                  _p(_req(grid_consts, "grid_consts")), c_int(grid[0]), c_int(grid[1]), c_int(grid[2]), c_int(int(bool(use_mask))),
                                  _p(s.stored_cell), _p(key_sel), _p(key_oof), _p(s.rows_order), _p(s.oof_order), _p(s.counts), _p(ws),
                                  c_size(ws.numel()), _stream()), "mcr_field_select")
    return s


def field_build(tables, J, n_seg, sel, proxy_points, S_all, view_states, bin_perm, vh_matrix_t, T, tot, X_world, vh):
    """The jobs of the occupancy-field pass (mcr_field_build).  tables: ONE uploaded fp64-free buffer = int64 [J*4 + n_seg*4] followed
    (as raw bytes) by fp32 [J*20]; X_world [>=T,3] and vh [>=T,64] are written in their first T rows.
    -> (rows int32 [T], row_job int32 [T], X_q [T,3], pc_all [tot,3])."""
    dev = proxy_points.device
    ib = torch.empty(2 * T, dtype=torch.int32, device=dev)
    rows, row_job = ib[:T], ib[T:]
    fb = torch.empty(3 * (T + tot), dtype=torch.float32, device=dev)
    X_q, pc_all = fb[:3 * T].view(T, 3), fb[3 * T:].view(tot, 3)
    base = tables.data_ptr()
    n_bins = view_states.shape[1]
    with torch.cuda.device(dev):
        check(lib().mcr_field_build(c_vp(base), c_int(J), c_vp(base + 32 * J), c_int(n_seg), c_vp(base + 32 * (J + n_seg)), _p(sel.rows_order),
                                    _p(proxy_points), _p(S_all), _p(view_states), c_int(n_bins), _p(bin_perm), _p(vh_matrix_t), c_i64(T),
                                    c_i64(tot), _p(rows), _p(row_job), _p(X_world), _p(X_q), _p(vh), _p(pc_all), _stream()), "mcr_field_build")
    return rows, row_job, X_q, pc_all


def view_harmonics_rows(view_states, rows, bin_perm, vh_matrix_t, out=None):
    """vh [T,64] of rows `rows` (int32 device, None = all) of a view-state table: bins permuted by bin_perm (int32 device, None = identity),
    then the [n_bins, 64] product (mcr_view_harmonics_rows)."""
    vs = _req(view_states, "view_states")
    T = rows.numel() if rows is not None else vs.shape[0]
    if out is None:
        out = torch.empty((T, 64), dtype=torch.float32, device=vs.device)
    if T == 0:
        return out
    with torch.cuda.device(vs.device):
        check(lib().mcr_view_harmonics_rows(_p(vs), c_int(vs.shape[1]), _p(rows) if rows is not None else c_vp(0),
                                            _p(bin_perm) if bin_perm is not None else c_vp(0), _p(_req(vh_matrix_t, "vh_matrix_t")), c_i64(T),
                                            _p(out), _stream()), "mcr_view_harmonics_rows")
    return out


def field_finish(rows, occ, T, proxy_proba, sel, n_oof, proxy_points, X_tail, occ_tail):
    with torch.cuda.device(proxy_points.device):
        check(lib().mcr_field_finish(_p(rows) if T else c_vp(0), _p(occ) if T else c_vp(0), c_i64(T), _p(proxy_proba), _p(sel.oof_order),
                                     c_i64(n_oof), _p(proxy_points), _p(X_tail) if n_oof else c_vp(0), _p(occ_tail) if n_oof else c_vp(0),
                                     _stream()), "mcr_field_finish")


def camera_boxes(sampled, n_unique, M_view, cam_world, inv_diag):
    """sampled [K,S,4], n_unique int32 [K], M_view [K,4,4], cam_world [K,3] -> (centre [K,3] of each camera's prediction box in view space,
    camera centres [K,3] in the normalised prediction space)   (mcr_camera_boxes)."""
    sampled = _req(sampled, "sampled")
    K, S = sampled.shape[0], sampled.shape[1]
    out = torch.empty((2, K, 3), dtype=torch.float32, device=sampled.device)
    with torch.cuda.device(sampled.device):
        check(lib().mcr_camera_boxes(_p(sampled), _p(_req(n_unique, "n_unique", torch.int32)), c_i64(K), c_int(S), _p(_req(M_view, "M_view")),
                                     _p(_req(cam_world, "cam_world")), c_f32(float(inv_diag)), _p(out[0]), _p(out[1]), _stream()),
              "mcr_camera_boxes")
    return out[0], out[1]


def macarons_gain_indexed(vis_unique, world_unique, inverse, n_unique, cam_world, volume, distance_th, smooth=False):
    """gains [K] from the per-UNIQUE-point visibility gains vis_unique [K,S] and the inverse map of the Monte-Carlo samples
    (mcr_macarons_gain_indexed)."""
    vis_unique, world_unique = _req(vis_unique, "vis_unique"), _req(world_unique, "world_unique")
    K, S = vis_unique.shape
    gains = torch.empty(K, dtype=torch.float32, device=vis_unique.device)
    with torch.cuda.device(vis_unique.device):
        check(lib().mcr_macarons_gain_indexed(_p(vis_unique), _p(world_unique), _p(_req(inverse, "inverse", torch.int64)),
                                              _p(_req(n_unique, "n_unique", torch.int32)), _p(_req(cam_world, "cam_world")),
                                              _p(_req(volume, "volume")), c_f32(float(distance_th)), c_int(int(bool(smooth))), c_i64(K), c_int(S),
                                              _p(gains), _stream()), "mcr_macarons_gain_indexed")
    return gains


_PHILOX_MAPPING = 1          # rocRAND's (0, 1] map (what torch.rand uses on ROCm); tests/test_glue_gpu.py pins it against torch.rand


def uniform_rows(K, S, device, generator=None):
    """[K, S] uniforms = what K consecutive torch.rand(S, 1, device=device) calls return (upstream's per-camera sampling draws,
    scone_utils.py:1052), from ONE launch; the device generator advances exactly as those K calls would advance it."""
    device = torch.device(device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    gen = generator if generator is not None else torch.cuda.default_generators[idx]
    if S > 65536:                                    # beyond one element per thread torch's kernel changes its indexing: the literal draws
        return torch.cat([torch.rand(S, 1, device=device, generator=generator) for _ in range(K)], 1).t().contiguous()
    seed, off = gen.initial_seed(), gen.get_offset()
    out = torch.empty((K, S), dtype=torch.float32, device=device)
    with torch.cuda.device(device):
        check(lib().mcr_philox_uniform_rows(ctypes.c_uint64(seed & (2 ** 64 - 1)), ctypes.c_uint64(off), c_i64(K), c_int(S),
                                            c_int(_PHILOX_MAPPING), _p(out), _stream()), "mcr_philox_uniform_rows")
    gen.set_offset(off + 4 * K)
    return out
