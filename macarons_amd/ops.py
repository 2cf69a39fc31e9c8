"""Thin torch-tensor wrappers over the C ABI (include/macarons_hip.h).

torch is plumbing here: device memory, streams.  Every function validates layout, allocates outputs /
scratch on the input's device and launches on the current HIP stream.  No CPU fallback.
"""
import ctypes
import torch

from ._lib import lib, check, MacaronsHipError, c_i64, c_int, c_size, c_vp, c_f32


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


# ---- K9 scorer -------------------------------------------------------------------------------------
def sh_coverage_gain(pts, harmonics, cams, use_sigmoid=True, waves_per_simd=0):
    """gains [B,C]; replaces SconeVis.compute_coverage_gain (SconeVis.py:210-252)."""
    pts, harmonics, cams = _req(pts, "pts"), _req(harmonics, "harmonics"), _req(cams, "X_cam")
    B, N, P = pts.shape
    C = cams.shape[1]
    if harmonics.shape != (B, N, 64):
        raise ValueError(f"harmonics must be [B,N,64] = {(B, N, 64)}, got {tuple(harmonics.shape)}")
    if cams.shape != (B, C, 3):
        raise ValueError(f"X_cam must be [B,C,3], got {tuple(cams.shape)}")
    L = lib()
    gains = torch.empty((B, C), dtype=torch.float32, device=pts.device)
    ws_bytes = L.mcr_sh_coverage_gain_workspace_bytes(c_i64(B), c_i64(N), c_i64(C))
    ws = torch.empty((max(ws_bytes, 4) + 3) // 4, dtype=torch.float32, device=pts.device)
    with torch.cuda.device(pts.device):
        check(L.mcr_sh_coverage_gain(_p(pts), c_int(P), _p(harmonics), _p(cams), _p(gains), c_i64(B), c_i64(N),
                                     c_i64(C), c_int(int(bool(use_sigmoid))), c_int(waves_per_simd), _p(ws),
                                     c_size(ws.numel() * 4), _stream()), "mcr_sh_coverage_gain")
    return gains


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
        check(lib().mcr_knn_points(_p(X), _p(pc), _p(idx), _p(dists), _p(pts), c_i64(B), c_i64(Q), c_i64(M), c_int(k),
                                   c_int(int(bool(subtract_query))), _stream()), "mcr_knn_points")
    return pts, dists, idx
