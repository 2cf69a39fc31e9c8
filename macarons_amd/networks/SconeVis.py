"""Host-side mirror of macarons/networks/SconeVis.py: the visibility-gain network and the per-candidate-camera
coverage-gain scorer, running on the MI355X through libmacarons_hip.so.

Same constructor, attributes (n_harmonics, max_harmonic_rank, use_sigmoid ...), sub-module / parameter names
(reference checkpoints load unchanged) and method surface as the reference:
  SconeVis.forward                         macarons/networks/SconeVis.py:121-162
  SconeVis.compute_visibilities            :164-208
  SconeVis.compute_coverage_gain           :210-252
  SconeVis.compute_coverage_gain_multiple  :254-303
The three loss modules the trainers import from this module (`scone_utils.py:14`: KLDivCE :306, L1_loss :322,
Uncentered_L1_loss :353) are defined at the bottom: plain torch (training-only, no kernel), restated from their definitions so
that `from ..networks.SconeVis import SconeVis, KLDivCE, L1_loss, Uncentered_L1_loss` resolves after `patch_reference()`.
"""
import numpy as np                       # noqa: F401  (upstream's `from .SconeVis import *` hands these names on: Macarons.py:5-6)
import torch
import torch.nn.functional as F
from torch import nn

from .. import autograd as A
from .. import ops
from .Attention import Embedding, Encoder, FeedForward, MultiHeadSelfAttention, attention, knn_gather, _f32c   # noqa: F401
from ..utility.CustomGeometry import get_spherical_coords                                                # noqa: F401
from ..utility.spherical_harmonics import clear_spherical_harmonics_cache, get_spherical_harmonics       # noqa: F401
from .packing import RangeGuard, TableCache, encoder_weight_planes, padded_weight_planes, weight_planes, freeze as _freeze_key


class SconeVis(RangeGuard, nn.Module):
    def __init__(self, pts_dim=4, seq_len=2048, pts_embedding_dim=256, n_heads=4, n_code=3, n_harmonics=64,
                 max_harmonic_rank=8, FF=True, gelu=True, dropout=None, use_view_state=True, use_global_feature=True,
                 view_state_mode="end", concatenate_input=True, k_for_knn=0, alt=False, use_sigmoid=True):
        super().__init__()
        self.n_harmonics = n_harmonics
        self.pts_dim, self.seq_len, self.pts_embedding_dim = pts_dim, seq_len, pts_embedding_dim
        self.n_heads, self.n_code, self.max_harmonic_rank = n_heads, n_code, max_harmonic_rank
        self.use_view_state, self.use_global_feature, self.view_state_mode = use_view_state, use_global_feature, view_state_mode
        self.alt = alt
        self.use_sigmoid = use_sigmoid
        print("Use sigmoid in model." if use_sigmoid else "Use ReLU for output in model.")      # SconeVis.py:72-75

        additional_feature_dim = n_harmonics if (use_view_state and view_state_mode == "start") else 0
        self.embedding = Embedding(pts_dim, pts_embedding_dim, gelu=gelu, global_feature=use_global_feature,
                                   additional_feature_dim=additional_feature_dim, concatenate_input=concatenate_input,
                                   k_for_knn=k_for_knn, dropout=None)
        self.encoders = nn.ModuleList([Encoder(seq_len=seq_len, embedding_dim=pts_embedding_dim,
                                               qk_dim=pts_embedding_dim // 4, n_heads=n_heads, dropout=dropout, gelu=gelu,
                                               FF=FF) for _ in range(n_code)])
        self.norm = nn.LayerNorm(pts_embedding_dim)
        if not alt:
            fc1_input_dim = pts_embedding_dim
            inner_feature_factor = 3 if (use_view_state and view_state_mode == "end") else 4
        else:
            fc1_input_dim = pts_embedding_dim + n_harmonics
            inner_feature_factor = 4
        self.fc1 = nn.Linear(fc1_input_dim, inner_feature_factor * n_harmonics)
        self.nonlinear1 = nn.GELU()
        self.fc2 = nn.Linear(4 * n_harmonics, 2 * n_harmonics)
        self.nonlinear2 = nn.GELU()
        self.fc3 = nn.Linear(2 * n_harmonics, n_harmonics)
        self._table_cache = TableCache()

    def freeze_weight_caches(self, on=True):
        """Inference mode: see SconeOcc.freeze_weight_caches."""
        _freeze_key(self, self._table_cache._c, on)

    def invalidate_weight_caches(self):
        """Drop the derived weight images (pointer table, stacked QKV); see packing._param_key for when this is needed."""
        self._table_cache.invalidate()
        for m in self.modules():
            if hasattr(m, "_packed"):
                m._packed = None

    # ---- the architecture the fused HIP forward implements (the one every call site builds) ----
    def _is_default_arch(self):
        return (self.pts_dim == 4 and self.pts_embedding_dim == 256 and self.n_heads == 4 and self.n_code == 3
                and self.n_harmonics == 64 and self.use_view_state and self.use_global_feature
                and self.view_state_mode == "end" and self.embedding.concatenate_input and not self.alt
                and all(e.FF for e in self.encoders))

    def weight_table(self):
        t = [_f32c(self.embedding.linear1.weight), _f32c(self.embedding.linear1.bias),
             _f32c(self.embedding.linear2.weight), _f32c(self.embedding.linear2.bias)]
        for e in self.encoders:
            t += e.weight_table()
        t += [_f32c(self.norm.weight), _f32c(self.norm.bias)]
        for fc in (self.fc1, self.fc2, self.fc3):
            t += [_f32c(fc.weight), _f32c(fc.bias)]
        return t

    def weight_table_with_planes(self):
        """weight_table() + per encoder the four weight matrices as fp16 hi/lo planes (the planes GEMMs of variant 6 then skip their
        per-call split launches: 12 of a forward's launches)."""
        t = self.weight_table()
        for e in self.encoders:
            t += encoder_weight_planes(e)
        # the layers either side of the encoders (5 blobs): the embedding's second layer zero-padded to 128 x 128 (+ its padded bias), fc1 / fc2 / fc3
        t += list(padded_weight_planes(self.embedding.linear2.weight, self.embedding.linear2.bias, 128, 128))
        t += [weight_planes(self.fc1.weight), weight_planes(self.fc2.weight), weight_planes(self.fc3.weight)]
        return t

    def forward(self, pts, mask=None, view_harmonics=None, lengths=None):
        """pts [n_clouds, seq_len, 4], view_harmonics [n_clouds, seq_len, 64] -> [n_clouds, seq_len, 64].
        lengths (extension, optional int32 device tensor [n_clouds]): cloud b is its first lengths[b] rows; the rest of the
        batch is padding (the reference slices on the host instead, which costs a device->host sync per decision)."""
        if not self._is_default_arch():
            raise NotImplementedError("the fused MI355X SconeVis forward implements the reference's default architecture")
        if view_harmonics is None:
            raise ValueError("view_harmonics is required (view_state_mode='end')")
        n_clouds, seq_len = pts.shape[0], pts.shape[1]
        if mask is not None:
            # An attention mask (SconeVis.py:121-141 hands it to every encoder): no call site of the hot path builds one, so the
            # fused launch sequence does not carry it; the same kernels run block by block (inference only).
            if lengths is not None or A.needs_grad(self, pts, view_harmonics):
                raise NotImplementedError("SconeVis.forward(mask=...) is the inference path without `lengths`")
            x = self.embedding(pts)
            for enc in self.encoders:
                x = enc(x, mask=mask)
            res = ops.layernorm(x, _f32c(self.norm.weight), _f32c(self.norm.bias))
            res = ops.linear(res, _f32c(self.fc1.weight), _f32c(self.fc1.bias), gelu=True)
            res = ops.linear(torch.cat((res, view_harmonics), dim=-1), _f32c(self.fc2.weight), _f32c(self.fc2.bias), gelu=True)
            return ops.linear(res, _f32c(self.fc3.weight), _f32c(self.fc3.bias)).view(n_clouds, seq_len, self.n_harmonics)
        # Range guard (packing.RangeGuard): on variant 6 the encoders of a cloud of >= 512 points run their GEMMs on fp16 hi/lo planes
        # (|activation| < 65504); harmonics that come out non-finite raise the flag.  "sync" (default): read now, repeat on variant 5 --
        # a stand-alone forward never returns non-finite harmonics; "defer": left in range_flag() (nbv_step and macarons_nbv_decision
        # share SconeOcc's flag and read it once per decision); "async" (opt-in): looked at without a stall by a later forward /
        # check_range(); "off": nothing.
        if self.range_guard == "async" and (self._range_pending or self._full_range):
            self.check_range()
        if self._full_range and ops.current_variant() in (6, 7):
            with ops.variant(5):
                return self.forward(pts, mask, view_harmonics, lengths)
        guarded = ops.current_variant() in (6, 7) and seq_len >= 512 and self.range_guard != "off" and not torch.cuda.is_current_stream_capturing()

        def hip(p, vh):
            res_ = ops.scone_vis_forward(p, vh, self._table_cache.get(self, self.weight_table_with_planes), lengths)
            if guarded:
                if self._range_flag is None or self._range_flag.device != p.device:
                    self._range_flag = torch.zeros(1, dtype=torch.int32, device=p.device)
                elif self.range_guard in ("sync", "async"):
                    self._range_flag.zero_()
                ops.nonfinite_flag_(res_, self._range_flag)
                if self.range_guard == "sync" and int(self._range_flag):
                    with ops.variant(5):
                        res_ = ops.scone_vis_forward(p, vh, self._table_cache.get(self, self.weight_table_with_planes), lengths)
                elif self.range_guard == "async":
                    self._post_range_check(self._range_flag)
            return res_
        if A.needs_grad(self, pts, view_harmonics):     # trainers (pretrain_scone_vis.py:224): HIP forward, composite-torch backward
            res = A.with_torch_backward(hip, lambda p, vh: A.scone_vis(self, p, vh, lengths), (pts, view_harmonics), self)
        else:
            res = hip(pts, view_harmonics)
        return res.view(n_clouds, seq_len, self.n_harmonics)

    @staticmethod
    def _scorer_ops():
        """The scorer's operators: `torch.ops.macarons.*` of the C++ TORCH_LIBRARY extension (libmacarons_torch.so over the C ABI; loaded
        on first use, raises if it was not built -- there is no Python or CPU fallback)."""
        from .. import torch_ops  # noqa: F401  (loads the library, registers the operators)
        return torch.ops.macarons

    @staticmethod
    def _degenerate(pts, X_cam, reduced, n_tuple=1):
        """What upstream's tensor algebra returns when a dimension is 0 (the kernels refuse empty problems): no clouds / no cameras
        -> an empty tensor; no points -> empty visibilities, and gains = sum over nothing / 0 = NaN (SconeVis.py:250).  None otherwise."""
        B, N, C = pts.shape[0], pts.shape[1], X_cam.shape[1]
        if B and N and C:
            return None
        if not reduced:
            return torch.empty((B, C, N), dtype=torch.float32, device=pts.device)
        return torch.full((B, C ** n_tuple), float("nan"), dtype=torch.float32, device=pts.device)    # (empty when B or C is 0)

    def compute_visibilities(self, pts, harmonics, X_cam):
        """-> [n_clouds, n_camera_candidates, seq_len]   (SconeVis.py:164-208)."""
        self._check_scorer(harmonics)
        d = self._degenerate(pts, X_cam, reduced=False)
        if d is not None:
            return d
        sig = self.use_sigmoid
        hip = lambda p, h, c: self._scorer_ops().sh_visibilities(p, h, c, sig)
        if A.needs_grad(None, pts, harmonics, X_cam):
            return A.with_torch_backward(hip, lambda p, h, c: A.visibilities(p, h, c, sig), (pts, harmonics, X_cam))
        return hip(pts, harmonics, X_cam)

    def compute_coverage_gain(self, pts, harmonics, X_cam):
        """-> [n_clouds, n_camera_candidates]   (SconeVis.py:210-252)."""
        self._check_scorer(harmonics)
        d = self._degenerate(pts, X_cam, reduced=True)
        if d is not None:
            return d
        sig = self.use_sigmoid
        hip = lambda p, h, c: self._scorer_ops().sh_coverage_gain(p, h, c, sig)
        if A.needs_grad(None, pts, harmonics, X_cam):
            return A.with_torch_backward(hip, lambda p, h, c: A.coverage_gain(p, h, c, sig), (pts, harmonics, X_cam))
        return hip(pts, harmonics, X_cam)

    def compute_coverage_gain_multiple(self, pts, harmonics, X_cam, n_cam):
        """Every ordered n_cam-tuple of cameras: mean over points of the max over the tuple (SconeVis.py:254-303)."""
        self._check_scorer(harmonics)
        if n_cam not in (2, 3):
            raise NameError("n_cam is too large.")                       # SconeVis.py:298
        d = self._degenerate(pts, X_cam, reduced=True, n_tuple=n_cam)
        if d is not None:
            single = torch.arange(0, X_cam.shape[1])
            return d, torch.cartesian_prod(*([single] * n_cam)).reshape(-1, n_cam)
        return ops.coverage_gain_multiple(pts, harmonics, X_cam, n_cam, self.use_sigmoid)

    def _check_scorer(self, harmonics):
        if self.n_harmonics != 64 or self.max_harmonic_rank != 8 or harmonics.shape[-1] != 64:
            # the reference hard-codes 64 at SconeVis.py:241
            raise NotImplementedError("the SH scorer is specialised for 64 harmonics (rank 8), as the reference hard-codes")


# ---- training losses (SconeVis.py:306-378).  Training-only; torch composite ops, autograd does the backward. ----
class KLDivCE(nn.Module):
    """KL(softmax(y) || softmax(x)) over dim 1, summed and divided by the batch size ('batchmean').  SconeVis.py:306-319."""

    def forward(self, x, y):
        return F.kl_div(F.log_softmax(x, dim=1), F.softmax(y, dim=1), reduction="batchmean")


def _camera_axis_l1(nx, ny):
    # mean over the cameras (dim 1) of |nx - ny|, then over everything left
    return (nx - ny).abs().mean(dim=1).mean()


class L1_loss(nn.Module):
    """L1 distance between the two coverage distributions after each is standardised over the cameras (dim 1): minus its mean, over
    its (unbiased) standard deviation + epsilon.  x, y [batch, n_camera, 1].  SconeVis.py:322-350."""

    def __init__(self):
        super().__init__()
        self.epsilon = 1e-7

    def _standardise(self, t):
        return (t - t.mean(dim=1, keepdim=True)) / (t.std(dim=1, keepdim=True) + self.epsilon)

    def forward(self, x, y):
        return _camera_axis_l1(self._standardise(x), self._standardise(y))


class Uncentered_L1_loss(nn.Module):
    """L1 distance between the two coverage distributions after each is divided by its mean over the cameras (+ epsilon).
    SconeVis.py:353-378."""

    def __init__(self):
        super().__init__()
        self.epsilon = 1e-7

    def forward(self, x, y):
        scale = lambda t: t / (t.mean(dim=1, keepdim=True) + self.epsilon)
        return _camera_axis_l1(scale(x), scale(y))
