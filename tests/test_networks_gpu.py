"""Parity of the HIP network path (through the C ABI) with the reference goldens and the numpy oracle.
Bar: 1e-4 relative (max-norm) in fp32, the north_star tolerance."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import golden, rel_err

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
import weights  # noqa: E402
from oracle import nets  # noqa: E402

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _mod(cls, seed, dev, **kw):
    m = cls(**kw)
    sd = weights.make_state_dict(weights.shapes_of(m), seed)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    return m.to(dev).eval(), sd


def T(x, dev):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev)


def test_blocks(dev):
    from macarons_amd.networks import Attention as A
    g = golden("blocks")
    with torch.no_grad():
        for tag, (E, qk) in {"vis": (256, 64), "occ": (128, 32)}.items():
            enc, _ = _mod(lambda: A.Encoder(seq_len=16, qk_dim=qk, embedding_dim=E, n_heads=4), 100 + E, dev)
            y = enc(T(g[f"enc_{tag}_x"], dev)).cpu().numpy()
            assert rel_err(y, g[f"enc_{tag}_y"]) < TOL
            y = A.attention(T(g[f"att_{tag}_q"], dev), T(g[f"att_{tag}_k"], dev), T(g[f"att_{tag}_v"], dev)).cpu().numpy()
            assert rel_err(y, g[f"att_{tag}_y"]) < TOL
        ev, _ = _mod(lambda: A.Embedding(4, 256, global_feature=True, concatenate_input=True), 7, dev)
        eo, _ = _mod(lambda: A.Embedding(3, 128, global_feature=False, concatenate_input=True), 8, dev)
        assert rel_err(ev(T(g["emb_vis_x"], dev)).cpu().numpy(), g["emb_vis_y"]) < TOL
        assert rel_err(eo(T(g["emb_occ_x"], dev)).cpu().numpy(), g["emb_occ_y"]) < TOL


@pytest.mark.parametrize("M,N,K,gelu,res", [(1, 1, 1, False, False), (130, 126, 4, True, False), (1000, 125, 125, False, False),
                                             (257, 512, 256, True, False), (4097, 128, 256, False, True),
                                             (300, 1, 256, True, False), (64, 192, 128, False, False), (33, 64, 1344, True, True),
                                             # large enough for the split-precision GEMM (linear3.hip), ragged in M, N and the K chunk
                                             (40000, 200, 72, True, True), (33000, 128, 256, False, False)])
def test_linear_vs_numpy(dev, M, N, K, gelu, res):
    from macarons_amd import ops
    rng = np.random.default_rng(M + N + K)
    x = rng.standard_normal((M, K)).astype(np.float32)
    w = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    r = rng.standard_normal((M, N)).astype(np.float32)
    y = ops.linear(T(x, dev), T(w, dev), T(b, dev), gelu=gelu, residual=T(r, dev) if res else None).cpu().numpy()
    ref = x.astype(np.float64) @ w.astype(np.float64).T + b
    if gelu:
        ref = nets.gelu(ref)
    if res:
        ref = ref + r
    assert np.abs(y - ref).max() < 2e-5 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("S,L,H,qk,v", [(1, 2048, 4, 64, 256), (1, 1777, 4, 64, 256), (3, 333, 4, 32, 128), (2, 65, 4, 32, 128)])
def test_long_sequence_attention_vs_numpy(dev, S, L, H, qk, v):
    """The MFMA flash-attention kernel (attention() of Attention.py:8-36 on packed q|k|v, mask=None) against an fp64
    softmax(QK^T/sqrt(d))V, incl. sequence lengths that are not a multiple of the 64-key tile / 16-query wave."""
    from macarons_amd import ops
    rng = np.random.default_rng(S * 1000 + L)
    qkv = rng.standard_normal((S, L, 2 * qk + v)).astype(np.float32)
    y = ops.attention_packed(T(qkv, dev), H, qk, v).cpu().numpy()
    x = qkv.astype(np.float64)
    hs = lambda t, d: t.reshape(S, L, H, d).transpose(0, 2, 1, 3)
    q, k, vv = hs(x[..., :qk], qk // H), hs(x[..., qk:2 * qk], qk // H), hs(x[..., 2 * qk:], v // H)
    sc = q @ k.transpose(0, 1, 3, 2) / np.sqrt(qk // H)
    sc = np.exp(sc - sc.max(-1, keepdims=True))
    ref = ((sc / sc.sum(-1, keepdims=True)) @ vv).transpose(0, 2, 1, 3).reshape(S, L, v)
    assert np.abs(y - ref).max() < 2e-5 * max(1.0, np.abs(ref).max())


def test_long_sequence_attention_outside_the_half_range(dev):
    """On the fp16-split variant the P V product of the long-sequence attention runs on fp16 hi/lo pairs; a 64-key tile holding a
    value the fp16 range cannot take (|v| >= 32768, inf) is detected while it is staged and takes the fp32 path: huge, mixed and
    infinite values against the fp64 reference (an infinite v gives inf / NaN in its column in both)."""
    from macarons_amd import ops
    S, L, H, qk, v = 2, 400, 4, 64, 256
    rng = np.random.default_rng(99)
    qkv = rng.standard_normal((S, L, 2 * qk + v)).astype(np.float32)
    qkv[0, 70:90, 2 * qk:] *= 1e6                                       # one tile of sequence 0 far outside, the others inside
    qkv[1, :, 2 * qk:2 * qk + 64] *= 40000.                             # head 0 of sequence 1: every tile outside
    qkv[1, 5, 2 * qk + 200] = 3e38                                      # and one value near the fp32 maximum
    y = ops.attention_packed(T(qkv, dev), H, qk, v).cpu().numpy()
    x = qkv.astype(np.float64)
    hs = lambda t, d: t.reshape(S, L, H, d).transpose(0, 2, 1, 3)
    q, k, vv = hs(x[..., :qk], qk // H), hs(x[..., qk:2 * qk], qk // H), hs(x[..., 2 * qk:], v // H)
    sc = q @ k.transpose(0, 1, 3, 2) / np.sqrt(qk // H)
    sc = np.exp(sc - sc.max(-1, keepdims=True))
    ref = ((sc / sc.sum(-1, keepdims=True)) @ vv).transpose(0, 2, 1, 3).reshape(S, L, v)
    assert np.isfinite(y).all()
    col = np.abs(ref).max(axis=(0, 1), keepdims=True)                    # per output column: the columns differ by 44 orders of magnitude
    assert (np.abs(y - ref) / np.maximum(col, 1.0)).max() < 2e-5


@pytest.mark.parametrize("S,L,H,qk,v", [(1, 2048, 4, 64, 256), (2, 1777, 4, 64, 256), (3, 700, 4, 32, 128), (12, 1100, 4, 32, 128), (9, 1500, 4, 64, 256),
                                        (3, 65, 4, 64, 256), (70, 130, 4, 32, 128), (1, 16, 4, 32, 128)])
def test_attention_on_planes_vs_numpy(dev, S, L, H, qk, v):
    """The attention of the long-sequence encoders on the fp16-split path takes q | k | v as fp16 hi/lo planes (K / V tiles by LDS DMA, both
    products on fp16 pairs; attention_planes.hip): against fp64 softmax(QK^T/sqrt(d))V with and without the key split, with per-sequence
    key counts (padded batches; incl. a sequence shorter than one tile), and batch == single sequence bit for bit (128- vs 64-query blocks)."""
    from macarons_amd import ops
    rng = np.random.default_rng(S * 77 + L)
    qkv = rng.standard_normal((S, L, 2 * qk + v)).astype(np.float32)
    lens = rng.integers(max(1, L // 3), L + 1, size=S).astype(np.int32)
    lens[0] = L
    if S > 2:
        lens[1] = min(37, L)
    x = qkv.astype(np.float64)

    def ref(n_keys):
        out = np.zeros((S, L, v))
        for s_ in range(S):
            q = x[s_, :, :qk].reshape(L, H, qk // H).transpose(1, 0, 2)
            k = x[s_, :n_keys[s_], qk:2 * qk].reshape(-1, H, qk // H).transpose(1, 0, 2)
            vv = x[s_, :n_keys[s_], 2 * qk:].reshape(-1, H, v // H).transpose(1, 0, 2)
            sc = q @ k.transpose(0, 2, 1) / np.sqrt(qk // H)
            sc = np.exp(sc - sc.max(-1, keepdims=True))
            out[s_] = ((sc / sc.sum(-1, keepdims=True)) @ vv).transpose(1, 0, 2).reshape(L, v)
        return out

    xt = T(qkv, dev)
    full = ref([L] * S)
    outs = {}
    for mode in (1, 0, -1):
        y = ops.attention_packed_planes(xt, H, qk, v, split_mode=mode)
        outs[mode] = y
        assert np.abs(y.cpu().numpy() - full).max() < 2e-5 * max(1.0, np.abs(full).max()), mode
    ragged = ref(lens)
    for mode in (1, 0):
        y = ops.attention_packed_planes(xt, H, qk, v, lens=T(lens, dev), split_mode=mode).cpu().numpy()
        assert np.abs(y - ragged).max() < 2e-5 * max(1.0, np.abs(ragged).max()), mode
    for b in (0, S - 1):                                               # one sequence alone: 64-query blocks
        for mode in (1, 0):
            one = ops.attention_packed_planes(xt[b:b + 1].contiguous(), H, qk, v, split_mode=mode)
            assert torch.equal(one[0], outs[mode][b]), (mode, b)


@pytest.mark.parametrize("S,L,H,qk,v", [(12, 1777, 4, 64, 256), (18, 1500, 4, 32, 128)])
def test_batched_attention_returns_the_single_sequence_bits(dev, S, L, H, qk, v):
    """A batch that fills the chip runs the long-sequence attention in 128-query blocks (two 16-query groups per wave share every
    staged K / V tile); one sequence alone runs in 64-query blocks with its keys split over two blocks when the sequence is long.
    Same arithmetic per query in the 128- and 64-query forms: equal bit for bit when neither splits its keys, within rounding of the
    key-split combine otherwise -- incl. a ragged length (not a multiple of 128 / 64 / 16) and a V tile outside the fp16 range
    (fp32 fallback inside the 128-query form); and the batch against fp64."""
    import ctypes
    from macarons_amd import ops, _lib
    rng = np.random.default_rng(S + L)
    qkv = rng.standard_normal((S, L, 2 * qk + v)).astype(np.float32)
    qkv[3, 130:150, 2 * qk:] *= 1e6                                    # one tile of sequence 3 leaves the fp16 range
    x = T(qkv, dev)
    prev = _lib.lib().mcr_get_local_pct_variant()
    try:
        for variant in (5, prev):                                      # 5: P V on the fp32 matrix pipe in both forms; default: fp16 pairs
            _lib.lib().mcr_set_local_pct_variant(ctypes.c_int(variant))
            big = ops.attention_packed(x, H, qk, v)
            for b in (0, 3, S - 1):
                one = ops.attention_packed(x[b:b + 1].contiguous(), H, qk, v, split=False)   # 64-query blocks, all keys in one block
                assert torch.equal(one[0], big[b]), (variant, b)
                two = ops.attention_packed(x[b:b + 1].contiguous(), H, qk, v, split=True)    # keys over two blocks + combine
                assert float((two[0] - big[b]).abs().max()) <= 2e-6 * float(big[b].abs().max()), (variant, b)
    finally:
        _lib.lib().mcr_set_local_pct_variant(ctypes.c_int(prev))
    xs = qkv[:2].astype(np.float64)
    hs = lambda t, d: t.reshape(2, L, H, d).transpose(0, 2, 1, 3)
    q, k, vv = hs(xs[..., :qk], qk // H), hs(xs[..., qk:2 * qk], qk // H), hs(xs[..., 2 * qk:], v // H)
    sc = q @ k.transpose(0, 1, 3, 2) / np.sqrt(qk // H)
    sc = np.exp(sc - sc.max(-1, keepdims=True))
    ref = ((sc / sc.sum(-1, keepdims=True)) @ vv).transpose(0, 2, 1, 3).reshape(2, L, v)
    assert np.abs(big[:2].cpu().numpy() - ref).max() < 2e-5 * max(1.0, np.abs(ref).max())


def test_scone_vis_padded_lengths_in_a_chip_filling_batch(dev):
    """lengths= (device-side key counts) inside the 128-query-block attention: a batch of 12 padded clouds of up to 2048 points ==
    the same clouds in batches of 2 (64-query blocks), bit for bit."""
    from macarons_amd.networks import SconeVis
    m, _ = _mod(SconeVis, 1, dev)
    rng = np.random.default_rng(13)
    N, lens = 2048, [2048, 1999, 1025, 1024, 700, 129, 128, 127, 65, 17, 16, 1]
    pts = np.concatenate([rng.uniform(-0.5, 0.5, (12, N, 3)), rng.uniform(0.1, 1.0, (12, N, 1))], -1).astype(np.float32)
    vh = (rng.standard_normal((12, N, 64)) * 0.3).astype(np.float32)
    for b, n in enumerate(lens):
        pts[b, n:] = 1e3
        vh[b, n:] = -7.0
    ln = torch.tensor(lens, dtype=torch.int32, device=dev)
    with torch.no_grad():
        y = m(T(pts, dev), view_harmonics=T(vh, dev), lengths=ln)
        for b in range(0, 12, 2):
            yb = m(T(pts[b:b + 2], dev), view_harmonics=T(vh[b:b + 2], dev), lengths=ln[b:b + 2].contiguous())
            for i in range(2):
                assert torch.equal(yb[i, :lens[b + i]], y[b + i, :lens[b + i]]), b + i


def test_layernorm_and_pools(dev):
    from macarons_amd import ops
    rng = np.random.default_rng(0)
    for E in (128, 256):
        x = (rng.standard_normal((37, 5, E)) * 3 + 1).astype(np.float32)
        g_, b_ = rng.standard_normal(E).astype(np.float32), rng.standard_normal(E).astype(np.float32)
        y = ops.layernorm(T(x, dev), T(g_, dev), T(b_, dev)).cpu().numpy()
        ref = nets.layernorm({"n.weight": g_, "n.bias": b_}, "n", x.astype(np.float64))
        assert np.abs(y - ref).max() < 2e-5
    x = rng.standard_normal((3, 333, 126)).astype(np.float32)
    assert np.array_equal(ops.colmax_broadcast(T(x, dev)).cpu().numpy(), np.broadcast_to(x.max(1, keepdims=True), x.shape))
    for S, L, E in ((1, 2048, 126), (3, 513, 126), (2, 1000, 64), (5, 640, 250), (20, 600, 126)):   # the long-sequence kernels (L >= 512)
        x = rng.standard_normal((S, L, E)).astype(np.float32)
        assert np.array_equal(ops.colmax_broadcast(T(x, dev)).cpu().numpy(), np.broadcast_to(x.max(1, keepdims=True), x.shape))
        y = ops.pool_max_avg(T(x, dev)).cpu().numpy()
        assert np.array_equal(y[:, :E], x.max(1)) and np.abs(y[:, E:] - x.mean(1)).max() < 1e-6
        if S > 1:                                  # a cloud's mean has the same bits alone and in a batch
            y1 = ops.pool_max_avg(T(x[1:2].copy(), dev)).cpu().numpy()
            assert np.array_equal(y1[0], y[1])
    x = rng.standard_normal((70000, 16, 128)).astype(np.float32)        # > 65535 sequences
    y = ops.pool_max_avg(T(x, dev)).cpu().numpy()
    assert np.array_equal(y[:, :128], x.max(1)) and np.abs(y[:, 128:] - x.mean(1)).max() < 1e-6


def test_scone_vis_forward(dev):
    from macarons_amd.networks import SconeVis
    m, sd = _mod(SconeVis, 1, dev)
    g = golden("scone_vis")
    with torch.no_grad():
        for N in (16, 333, 2048):
            y = m(T(g[f"pts_{N}"], dev), view_harmonics=T(g[f"vh_{N}"], dev)).cpu().numpy()
            assert y.shape == (1, N, 64)
            assert rel_err(y, g[f"y_{N}"]) < TOL
        assert rel_err(m(T(g["pts_333"], dev), view_harmonics=T(g["vh_333"], dev)).cpu().numpy(), g["y64_333"]) < TOL
        y = m(T(g["pts_b3"], dev), view_harmonics=T(g["vh_b3"], dev)).cpu().numpy()
        assert rel_err(y, g["y_b3"]) < TOL
        # ragged size vs the oracle, and the module-by-module path agrees with the fused one
        rng = np.random.default_rng(3)
        pts = rng.uniform(-.5, .5, (2, 777, 4)).astype(np.float32)
        vh = (rng.standard_normal((2, 777, 64)) * .3).astype(np.float32)
        y = m(T(pts, dev), view_harmonics=T(vh, dev)).cpu().numpy()
        assert rel_err(y, nets.scone_vis_forward(sd, pts, vh, np.float64)) < TOL
        x = m.embedding(T(pts, dev))
        for e in m.encoders:
            x = e(x)
        assert rel_err(x.cpu().numpy(), nets_encoders(sd, pts)) < TOL


def test_end_layers_on_the_planes_route_with_large_activations(dev):
    """On the default variant the layers either side of the long-sequence encoders (the embedding's second layer, final LayerNorm,
    fc1 / fc2 / fc3; the global transformer's linear0) run as planes GEMMs (fp16 hi/lo, three MFMAs per product) instead of exact
    fp32: against the fp64 oracle at 2048 tokens, with the end layers' weights scaled so that their activations reach the hundreds
    (pin these layers' tolerance, large magnitudes included), and against the exact-fp32 variant."""
    import ctypes
    from macarons_amd import _lib
    from macarons_amd.networks import SconeVis, SconeOcc
    vis, sd = _mod(SconeVis, 1, dev)
    rng = np.random.default_rng(11)
    for scale in (1.0, 60.0):
        sd2 = dict(sd)
        for k in ("fc1.weight", "fc2.weight", "embedding.linear2.weight"):
            sd2[k] = (sd[k] * scale).astype(np.float32)
        vis.load_state_dict({k: torch.from_numpy(v) for k, v in sd2.items()}, strict=True)
        pts = np.concatenate([rng.uniform(-.5, .5, (2, 2048, 3)), rng.uniform(.1, 1, (2, 2048, 1))], -1).astype(np.float32)
        vh = (rng.standard_normal((2, 2048, 64)) * .3).astype(np.float32)
        ref = nets.scone_vis_forward(sd2, pts, vh, np.float64)
        with torch.no_grad():
            y = vis(T(pts, dev), view_harmonics=T(vh, dev)).cpu().numpy()
            assert rel_err(y, ref) < TOL, scale
            prev = _lib.lib().mcr_get_local_pct_variant()
            try:
                _lib.lib().mcr_set_local_pct_variant(ctypes.c_int(1))      # exact-fp32 MFMA everywhere
                y1 = vis(T(pts, dev), view_harmonics=T(vh, dev)).cpu().numpy()
            finally:
                _lib.lib().mcr_set_local_pct_variant(ctypes.c_int(prev))
            assert rel_err(y, y1) < TOL, scale
    occ, sdo = _mod(SconeOcc, 2, dev)
    sd3 = dict(sdo)
    for k in ("global_transformer.linear0.weight", "global_transformer.embedding.linear2.weight"):
        sd3[k] = (sdo[k] * 40.0).astype(np.float32)
    occ.load_state_dict({k: torch.from_numpy(v) for k, v in sd3.items()}, strict=True)
    pc = rng.uniform(-.5, .5, (3, 2048, 3)).astype(np.float32)
    with torch.no_grad():
        gf = occ.global_transformer(T(pc, dev)).cpu().numpy()
    assert rel_err(gf, nets.pc_transformer(sd3, "global_transformer.", pc, np.float64)) < TOL


def nets_encoders(sd, pts):
    x = nets.embedding(sd, "embedding", pts.astype(np.float64), True)
    for i in range(3):
        x = nets.encoder(sd, f"encoders.{i}", x)
    return x


def test_scone_occ_forward(dev):
    from macarons_amd.networks import SconeOcc
    m, sd = _mod(SconeOcc, 2, dev)
    g = golden("scone_occ")
    with torch.no_grad():
        for tag in ("m100_q17", "m1024_q300", "m4096_q512"):
            perms = [torch.from_numpy(g[f"{tag}_perm{i}"].astype(np.int64)) for i in range(3)]
            pc, x, vh = T(g[f"{tag}_pc"], dev), T(g[f"{tag}_x"], dev), T(g[f"{tag}_vh"], dev)
            gf = m.global_transformer(pc[:, perms[0].to(dev)].contiguous()).cpu().numpy()
            assert rel_err(gf, g[f"{tag}_gfeat"]) < TOL
            y = m(pc, x, vh, perms=perms).cpu().numpy()
            assert y.shape == g[f"{tag}_y"].shape
            assert rel_err(y, g[f"{tag}_y"]) < TOL
            # hidden-RNG path: seeding torch like the reference run reproduces the same draws and output
            torch.manual_seed(int(g[f"{tag}_seed"]))
            y2 = m(pc, x, vh).cpu().numpy()
            assert np.array_equal(y, y2)


def test_weight_caches_follow_parameter_changes(dev):
    """The derived weight images (packed local-transformer blobs, pointer table, head planes) share one fingerprint per forward:
    every kind of parameter change an optimizer or a checkpoint load makes must reach all of them, and an edit no fingerprint can
    see (in place through .data) must reach them through invalidate_weight_caches()."""
    from macarons_amd.networks import SconeOcc
    m, sd = _mod(SconeOcc, 2, dev)
    g = golden("scone_occ")
    tag = "m1024_q300"
    perms = [torch.from_numpy(g[f"{tag}_perm{i}"].astype(np.int64)) for i in range(3)]
    pc, x, vh = T(g[f"{tag}_pc"], dev), T(g[f"{tag}_x"], dev), T(g[f"{tag}_vh"], dev)

    def expected(edit):                            # a freshly built module with the same edit: no cache history
        r, _ = _mod(SconeOcc, 2, dev)
        with torch.no_grad():
            edit(r)
            return r(pc, x, vh, perms=perms)

    with torch.no_grad():
        y0 = m(pc, x, vh, perms=perms)
        # (a) in-place update as an optimizer does it (version counter): blob of local transformer 1, head planes, table
        def step(r):
            r.local_transformers[1].linear0.weight.mul_(1.25)
            r.linear2.weight.add_(0.01)
            r.linear3.bias.add_(0.1)
        step(m)
        y1 = m(pc, x, vh, perms=perms)
        assert not torch.equal(y1, y0) and torch.equal(y1, expected(step))
        # (b) a new Parameter object (identity) and new storage (load_state_dict(assign=True)-like)
        def swap(r):
            step(r)
            r.x_embedding.linear2.weight = torch.nn.Parameter(r.x_embedding.linear2.weight.detach().clone() * 0.5)
        m.x_embedding.linear2.weight = torch.nn.Parameter(m.x_embedding.linear2.weight.detach().clone() * 0.5)
        y2 = m(pc, x, vh, perms=perms)
        assert not torch.equal(y2, y1) and torch.equal(y2, expected(swap))
        # (c) in place through .data: invisible to the fingerprint (documented) until the caches are invalidated
        def hidden(r):
            swap(r)
            r.local_transformers[0].linear0.weight.data.mul_(1.5)
        m.local_transformers[0].linear0.weight.data.mul_(1.5)
        m.invalidate_weight_caches()
        y3 = m(pc, x, vh, perms=perms)
        assert not torch.equal(y3, y2) and torch.equal(y3, expected(hidden))


def test_scone_occ_chunking_and_batch(dev):
    """Q larger than one chunk and B > 1 against the oracle on a sample of queries."""
    from macarons_amd.networks import SconeOcc
    m, sd = _mod(SconeOcc, 2, dev)
    rng = np.random.default_rng(9)
    B, M, Q = 2, 700, 20000
    pc = rng.uniform(-.4, .4, (B, M, 3)).astype(np.float32)
    x = rng.uniform(-.5, .5, (B, Q, 3)).astype(np.float32)
    vh = (rng.standard_normal((B, Q, 64)) * .3).astype(np.float32)
    torch.manual_seed(5)
    perms = m.draw_perms(M)
    with torch.no_grad():
        y = m(T(pc, dev), T(x, dev), T(vh, dev), perms=perms).cpu().numpy()
    sel = np.concatenate([np.arange(5), rng.choice(Q, 40, replace=False), [Q - 1, 16383, 16384]])
    ref = nets.scone_occ_forward(sd, pc, x[:, sel], vh[:, sel], [p.numpy() for p in perms], np.float64)
    assert rel_err(y[:, sel], ref) < TOL


@pytest.mark.parametrize("variant", [1, 5, 6])
def test_fused_local_transformer(dev, variant):
    """Fused local transformer kernels (1: exact-fp32 MFMA, 5: split-precision bf16 hi/mid/lo x 6 MFMAs, 6: two-term fp16
    split x 3 MFMAs, the default) vs the layer-by-layer HIP path and the fp64 oracle.  All must be fp32-class: 2e-5, far
    inside the 1e-4 bar."""
    import ctypes
    from macarons_amd import ops, _lib
    from macarons_amd.networks import SconeOcc
    from macarons_amd.networks.packing import pack_local_pct
    m, sd = _mod(SconeOcc, 2, dev)
    rng = np.random.default_rng(4)
    prev = _lib.lib().mcr_get_local_pct_variant()
    _lib.lib().mcr_set_local_pct_variant(ctypes.c_int(variant))
    try:
        for S in (1, 3, 4, 1001):
            offs = (rng.standard_normal((S, 16, 3)) * (0.05 if S != 3 else 0.5)).astype(np.float32)
            for sc in range(3):
                lt = m.local_transformers[sc]
                with torch.no_grad():
                    fused = ops.local_pct_forward(T(offs, dev), pack_local_pct(lt, variant)).cpu().numpy()
                    plain = lt(T(offs, dev)).cpu().numpy()
                ref = nets.pc_transformer(sd, f"local_transformers.{sc}.", offs, np.float64)
                assert rel_err(fused, ref) < 2e-5 and rel_err(plain, ref) < 2e-5
    finally:
        _lib.lib().mcr_set_local_pct_variant(ctypes.c_int(prev))


def test_split_precision_is_exact_split(dev):
    """The host-side hi/mid/lo bf16 planes of the v5 blob reconstruct every weight bit for bit (asserted inside
    _pack_bf16x3); the split-precision kernels (5: bf16 x 6, 6: fp16 x 3) agree with the exact-fp32 MFMA kernel."""
    import ctypes
    from macarons_amd import ops, _lib
    from macarons_amd.networks import SconeOcc
    from macarons_amd.networks.packing import pack_local_pct
    m, sd = _mod(SconeOcc, 2, dev)
    rng = np.random.default_rng(6)
    offs = (rng.standard_normal((257, 16, 3)) * 0.05).astype(np.float32)
    prev = _lib.lib().mcr_get_local_pct_variant()
    try:
        out = {}
        for v in (1, 5, 6):
            _lib.lib().mcr_set_local_pct_variant(ctypes.c_int(v))
            out[v] = ops.local_pct_forward(T(offs, dev), pack_local_pct(m.local_transformers[1], v)).cpu().numpy()
        assert rel_err(out[5], out[1]) < 5e-6 and rel_err(out[6], out[1]) < 5e-6
    finally:
        _lib.lib().mcr_set_local_pct_variant(ctypes.c_int(prev))


@pytest.mark.parametrize("where", ["local_ff", "head"])
def test_range_guard_falls_back_to_full_range_variant(dev, where):
    """The default matrix path (variant 6: fp16 hi/lo planes) needs |activation| < 65504; the reference is plain fp32
    (Attention.py:98-128).  With weights scaled by 2^16 activations leave that range: variant 6 ALONE returns garbage (non-finite
    occupancies) and raises the device flag; under the DEFAULT guard ("sync") the FIRST overflowed stand-alone forward of a fresh module
    already returns finite occupancies within 1e-4 of the fp64 oracle (repeated on variant 5); with "async" (opt-in: no read-back per
    forward) the overflowed forward returns the non-finite occupancies, the flag is noticed without a stall and every later forward of
    the module runs on variant 5; nbv_step checks the same flag once at the end of the decision."""
    from macarons_amd.networks import SconeOcc, SconeVis
    from macarons_amd.nbv import nbv_step, ViewStateGrid
    from macarons_amd import _lib
    if _lib.lib().mcr_get_local_pct_variant() != 6:
        pytest.skip("the range guard belongs to variant 6 (suite running on another variant)")
    m, sd = _mod(SconeOcc, 2, dev)
    sd = {k: v.copy() for k, v in sd.items()}
    keys = (["local_transformers.1.encoders.0.ff.linear1.weight", "local_transformers.1.encoders.0.ff.linear1.bias"] if where == "local_ff"
            else ["linear1.weight", "linear1.bias", "x_embedding.linear2.weight", "x_embedding.linear2.bias"])
    for k in keys:
        sd[k] = sd[k] * np.float32(65536.0)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    g = golden("scone_occ")
    tag = "m1024_q300"
    perms = [torch.from_numpy(g[f"{tag}_perm{i}"].astype(np.int64)) for i in range(3)]
    pc, x, vh = T(g[f"{tag}_pc"], dev), T(g[f"{tag}_x"], dev), T(g[f"{tag}_vh"], dev)
    ref = nets.scone_occ_forward(sd, g[f"{tag}_pc"], g[f"{tag}_x"], g[f"{tag}_vh"], [p.numpy() for p in perms], np.float64)
    with torch.no_grad():
        m.range_guard = "off"
        y6 = m(pc, x, vh, perms=perms).cpu().numpy()
        assert not np.isfinite(y6).all()                                 # variant 6 alone: out of range
        m.range_guard = "defer"
        m.clear_range_flag()
        m(pc, x, vh, perms=perms)
        assert int(m.range_flag()) == 1                                  # ... and says so
    # the default guard on a FRESH module: the first overflowed stand-alone forward returns finite values inside the contract
    m0, _ = _mod(SconeOcc, 2, dev)
    m0.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    assert m0.range_guard == "sync"
    with torch.no_grad():
        y = m0(pc, x, vh, perms=perms).cpu().numpy()
    assert np.isfinite(y).all() and rel_err(y, ref) < 1e-4               # guarded: repeated on variant 5
    # the opt-in guard: nothing is read back inside forward; the overflow is noticed afterwards and the module moves to variant 5
    m3, _ = _mod(SconeOcc, 2, dev)
    m3.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m3.range_guard = "async"
    with torch.no_grad():
        ya = m3(pc, x, vh, perms=perms)
        assert not torch.isfinite(ya).all()                              # the overflowed forward itself returns what variant 6 computed
        with pytest.warns(RuntimeWarning, match="full-range variant 5"):
            assert m3.check_range(wait=True) is True
        yb = m3(pc, x, vh, perms=perms).cpu().numpy()
        assert _lib.lib().mcr_get_local_pct_variant() == 6               # the process default was never touched
    assert np.isfinite(yb).all() and rel_err(yb, ref) < 1e-4
    # an in-range model leaves the flag clear
    m2, _ = _mod(SconeOcc, 2, dev)
    with torch.no_grad():
        m2.range_guard = "defer"
        m2(pc, x, vh, perms=perms)
    assert int(m2.range_flag()) == 0
    # the whole decision: one deferred check at the end, repeat on variant 5
    vis, _ = _mod(SconeVis, 1, dev)
    with torch.no_grad():
        m.linear3.bias += 0.5
    gg = golden("e2e_grid_config1")
    a = (m, vis, T(gg["pc"], dev), T(gg["X"], dev), T(gg["X_view"], dev), T(gg["X_cam"], dev), ViewStateGrid(dev))
    p2 = [torch.from_numpy(gg[f"perm{i}"].astype(np.int64)) for i in range(3)]
    r = nbv_step(*a, occ_perms=p2, samples=T(gg["samples"], dev))
    assert r.get("fallback_variant") == 5 and torch.isfinite(r["occ"]).all()
    assert torch.isfinite(r["gains"]).all() or int(r["n_unique"]) == 0     # (a head scaled by 2^16 may leave no point above min_occ)
    r0 = nbv_step(*a, occ_perms=p2, samples=T(gg["samples"], dev), range_guard=False)
    assert int(r0["range_flag"]) == 1 and "fallback_variant" not in r0


def test_scone_vis_encoders_on_planes_and_their_range_guard(dev):
    """Variant 6 runs the encoder GEMMs of clouds of >= 512 points on fp16 hi/lo planes (LayerNorm / GELU epilogues write the planes,
    linear3p.hip): against the fp64 oracle at 1e-4 like every other path (scone_vis.npz holds it to the reference), a batch gives each
    cloud the bits of its single call, and activations beyond the fp16 range are caught: under the default guard ("sync") the first
    overflowed forward is repeated on variant 5 and returns finite harmonics within 1e-4; the opt-in "async" guard notices without a
    read-back inside forward and moves the module to variant 5."""
    from macarons_amd.networks import SconeVis
    from macarons_amd import _lib, ops
    if _lib.lib().mcr_get_local_pct_variant() != 6:
        pytest.skip("the planes encoders belong to variant 6 (suite running on another variant)")
    m, sd = _mod(SconeVis, 1, dev)
    rng = np.random.default_rng(77)
    pts = np.concatenate([rng.uniform(-.5, .5, (3, 700, 3)), rng.uniform(.1, 1., (3, 700, 1))], -1).astype(np.float32)
    vh = (rng.standard_normal((3, 700, 64)) * 0.3).astype(np.float32)
    with torch.no_grad():
        y = m(T(pts, dev), view_harmonics=T(vh, dev))
        for b in range(3):
            assert torch.equal(y[b:b + 1], m(T(pts[b:b + 1], dev), view_harmonics=T(vh[b:b + 1], dev))), b
        with ops.variant(5):
            y5 = m(T(pts, dev), view_harmonics=T(vh, dev))
    ref = nets.scone_vis_forward(sd, pts, vh, np.float64)
    assert rel_err(y.cpu().numpy(), ref) < 1e-5 and rel_err(y5.cpu().numpy(), ref) < 1e-5
    assert not torch.equal(y, y5)                                      # (two different matrix paths did run)
    assert m.check_range(wait=True) is False
    # out of range: an FF layer scaled by 2^17
    sd2 = {k: v.copy() for k, v in sd.items()}
    for k in ("encoders.1.ff.linear1.weight", "encoders.1.ff.linear1.bias"):
        sd2[k] = sd2[k] * np.float32(131072.0)
    m2, _ = _mod(SconeVis, 1, dev)
    m2.load_state_dict({k: torch.from_numpy(v) for k, v in sd2.items()})
    ref2 = nets.scone_vis_forward(sd2, pts[:1], vh[:1], np.float64)
    m2.range_guard = "async"
    with torch.no_grad():
        ya = m2(T(pts[:1], dev), view_harmonics=T(vh[:1], dev))
        assert not torch.isfinite(ya).all()
        with pytest.warns(RuntimeWarning, match="full-range variant 5"):
            assert m2.check_range(wait=True) is True
        yb = m2(T(pts[:1], dev), view_harmonics=T(vh[:1], dev))
        m3, _ = _mod(SconeVis, 1, dev)
        m3.load_state_dict({k: torch.from_numpy(v) for k, v in sd2.items()})
        assert m3.range_guard == "sync"                                  # the default: the FIRST overflowed forward is already repeated
        yc = m3(T(pts[:1], dev), view_harmonics=T(vh[:1], dev))
    assert torch.isfinite(yb).all() and rel_err(yb.cpu().numpy(), ref2) < 1e-4 and torch.equal(yb, yc)


def test_scone_occ_ragged_equals_job_by_job(dev):
    """SconeOcc.forward_ragged (J clouds / query chunks of different sizes in one launch sequence: segmented kNN, padded global
    down-samples with lengths, per-row job bias in the head) == J forward() calls with the same draws: local features, x-embedding
    and head are the same kernels on the same rows (bit-equal unless the global feature differs); the global feature goes through
    the padded attention, so the bar is 2e-6."""
    from macarons_amd.networks import SconeOcc
    m, sd = _mod(SconeOcc, 2, dev)
    rng = np.random.default_rng(12)
    sizes_m, sizes_q = [100, 3000, 65, 2048, 900], [17, 300, 1, 129, 4097]
    clouds = [rng.uniform(-.4, .4, (n, 3)).astype(np.float32) for n in sizes_m]
    xs = [rng.uniform(-.5, .5, (q, 3)).astype(np.float32) for q in sizes_q]
    vhs = [(rng.standard_normal((q, 64)) * .3).astype(np.float32) for q in sizes_q]
    torch.manual_seed(21)
    perms = [m.draw_perms(n) for n in sizes_m]
    with torch.no_grad():
        y = m.forward_ragged(T(np.concatenate(clouds), dev), sizes_m, T(np.concatenate(xs), dev), T(np.concatenate(vhs), dev), sizes_q,
                             perms=perms).cpu().numpy()
        ref = np.concatenate([m(T(c[None], dev), T(x[None], dev), T(v[None], dev), perms=p).cpu().numpy().reshape(-1, 1)
                              for c, x, v, p in zip(clouds, xs, vhs, perms)])
        # the hidden draws: seeding reproduces the job-by-job sequence
        torch.manual_seed(21)
        y2 = m.forward_ragged(T(np.concatenate(clouds), dev), sizes_m, T(np.concatenate(xs), dev), T(np.concatenate(vhs), dev), sizes_q).cpu().numpy()
    assert y.shape == ref.shape and rel_err(y, ref) < 2e-6
    assert np.array_equal(y, y2)
    o64 = np.concatenate([nets.scone_occ_forward(sd, c[None], x[None], v[None], [q.numpy() for q in p], np.float64).reshape(-1, 1)
                          for c, x, v, p in zip(clouds[:3], xs[:3], vhs[:3], perms[:3])])
    assert rel_err(y[:len(o64)], o64) < TOL


def test_scone_occ_fused_equals_unfused(dev):
    from macarons_amd.networks import SconeOcc
    m, sd = _mod(SconeOcc, 2, dev)
    g = golden("scone_occ")
    tag = "m1024_q300"
    perms = [torch.from_numpy(g[f"{tag}_perm{i}"].astype(np.int64)) for i in range(3)]
    pc, x, vh = T(g[f"{tag}_pc"], dev), T(g[f"{tag}_x"], dev), T(g[f"{tag}_vh"], dev)
    with torch.no_grad():
        m.fused_local = True
        y1 = m(pc, x, vh, perms=perms).cpu().numpy()
        m.fused_local = False
        y0 = m(pc, x, vh, perms=perms).cpu().numpy()
    assert rel_err(y1, g[f"{tag}_y"]) < TOL and rel_err(y0, g[f"{tag}_y"]) < TOL


def test_scone_vis_padded_lengths_equal_sliced(dev):
    """SconeVis.forward(lengths=...) on a zero/garbage-padded batch == the forward on the sliced clouds (to rounding): the cloud-wide max of
    the embedding and the attention keys stop at lengths[b] (this is what lets the NBV step keep the number of unique sampled
    points on the device)."""
    from macarons_amd.networks import SconeVis
    m, _ = _mod(SconeVis, 1, dev)
    rng = np.random.default_rng(12)
    N, lens = 700, [700, 333, 64, 1]
    pts = np.concatenate([rng.uniform(-0.5, 0.5, (4, N, 3)), rng.uniform(0.1, 1.0, (4, N, 1))], -1).astype(np.float32)
    vh = (rng.standard_normal((4, N, 64)) * 0.3).astype(np.float32)
    for b, n in enumerate(lens):                         # padding: zeros for one cloud, large garbage for the others
        pts[b, n:] = 0.0 if b == 1 else 1e3
        vh[b, n:] = 0.0 if b == 1 else -7.0
    with torch.no_grad():
        y = m(T(pts, dev), view_harmonics=T(vh, dev), lengths=torch.tensor(lens, dtype=torch.int32, device=dev)).cpu().numpy()
        for b, n in enumerate(lens):
            yb = m(T(pts[b:b + 1, :n], dev), view_harmonics=T(vh[b:b + 1, :n], dev)).cpu().numpy()
            # same arithmetic, but a long padded batch may split its attention keys over two blocks (different summation order)
            assert np.abs(y[b, :n] - yb[0]).max() <= 2e-5 * np.abs(yb[0]).max(), (b, n, np.abs(y[b, :n] - yb[0]).max())


def test_trainer_gradients_hip_forward_torch_backward(dev):
    """With gradients enabled the entry points the trainers differentiate (pretrain_scone_vis.py:224, pretrain_scone_occ.py,
    train_macarons.py:1159) run the HIP forward and a composite-torch backward: the forward value is the HIP one (bit for bit
    what the no-grad call returns), the composite reproduces it to 1e-4 and the gradients equal autograd through the composite."""
    from macarons_amd import autograd as A
    from macarons_amd.networks import SconeVis, SconeOcc
    vis, _ = _mod(SconeVis, 1, dev)
    occ, _ = _mod(SconeOcc, 2, dev)
    rng = np.random.default_rng(21)
    pts = T(np.concatenate([rng.uniform(-.5, .5, (2, 60, 3)), rng.uniform(.1, 1, (2, 60, 1))], -1).astype(np.float32), dev)
    vh = T((rng.standard_normal((2, 60, 64)) * 0.3).astype(np.float32), dev)
    cams = T(rng.standard_normal((2, 7, 3)).astype(np.float32), dev)
    w = T(rng.standard_normal((2, 7)).astype(np.float32), dev)
    with torch.no_grad():
        h_ref = vis(pts, view_harmonics=vh)
        g_ref = vis.compute_coverage_gain(pts, h_ref, cams)
    for p in vis.parameters():
        p.grad = None
    h = vis(pts, view_harmonics=vh)                            # grad enabled, parameters require grad
    g = vis.compute_coverage_gain(pts, h, cams)
    assert torch.equal(h, h_ref) and torch.equal(g, g_ref) and g.requires_grad
    (g * w).sum().backward()
    got = {n: p.grad.clone() for n, p in vis.named_parameters()}
    for p in vis.parameters():
        p.grad = None
    hc = A.scone_vis(vis, pts, vh)
    gc = A.coverage_gain(pts, hc, cams)
    assert rel_err(hc.detach().cpu().numpy(), h_ref.cpu().numpy()) < 1e-4 and rel_err(gc.detach().cpu().numpy(), g_ref.cpu().numpy()) < 1e-4
    (gc * w).sum().backward()
    scale = max(float(p.grad.abs().max()) for p in vis.parameters())
    for n, p in vis.named_parameters():
        # relative to the layer's own gradient, floored at 1e-4 of the largest one (w_k.bias has a mathematically ZERO gradient:
        # softmax ignores a constant added to every key's score, what is left there is rounding noise)
        ref = p.grad.cpu().numpy()
        assert got[n] is not None and np.abs(got[n].cpu().numpy() - ref).max() < 1e-3 * max(np.abs(ref).max(), 1e-4 * scale), n
    # SconeOcc: gradient w.r.t. parameters and the query points
    pc = T(rng.uniform(-.3, .3, (1, 300, 3)).astype(np.float32), dev)
    x = T(rng.uniform(-.4, .4, (1, 50, 3)).astype(np.float32), dev).requires_grad_(True)
    vq = T((rng.standard_normal((1, 50, 64)) * 0.3).astype(np.float32), dev)
    torch.manual_seed(4)
    perms = occ.draw_perms(300)
    with torch.no_grad():
        y_ref = occ(pc, x.detach(), vq, perms=perms)
    y = occ(pc, x, vq, perms=perms)
    assert torch.equal(y, y_ref)
    y.sum().backward()
    gx = x.grad.clone()
    gw = occ.linear2.weight.grad.clone()
    x.grad = None
    for p in occ.parameters():
        p.grad = None
    from macarons_amd import ops
    dev_perms = [p.to(dev) for p in perms]
    scales = [pc, pc[:, dev_perms[1]]]
    scales.append(scales[1][:, dev_perms[2]])
    idx = [ops.knn_points(x.detach().contiguous(), s_.contiguous(), 16)[2] for s_ in scales]
    yc = A.scone_occ(occ, pc[:, dev_perms[0]], scales, x, vq, idx)
    assert rel_err(yc.detach().cpu().numpy(), y_ref.cpu().numpy()) < 1e-4
    yc.sum().backward()
    assert rel_err(gx.cpu().numpy(), x.grad.cpu().numpy()) < 1e-3 and rel_err(gw.cpu().numpy(), occ.linear2.weight.grad.cpu().numpy()) < 1e-3


def test_torch_ops_namespace_matches_module_path(dev):
    """torch.ops.macarons.* forwards to the same C ABI as the module surface: identical tensors."""
    import macarons_amd.torch_ops  # noqa: F401  (registers the operators)
    from macarons_amd import ops
    from macarons_amd.networks import SconeVis
    rng = np.random.default_rng(21)
    pts = T(np.concatenate([rng.uniform(-.5, .5, (1, 300, 3)), rng.uniform(.1, 1, (1, 300, 1))], -1).astype(np.float32), dev)
    harm = T((rng.standard_normal((1, 300, 64)) * 0.5).astype(np.float32), dev)
    cams = T(rng.standard_normal((1, 9, 3)).astype(np.float32), dev)
    assert torch.equal(torch.ops.macarons.sh_coverage_gain(pts, harm, cams, True), ops.sh_coverage_gain(pts, harm, cams, True))
    assert torch.equal(torch.ops.macarons.sh_visibilities(pts, harm, cams, False), ops.sh_visibilities(pts, harm, cams, False))
    X, pc = T(rng.uniform(-.5, .5, (1, 70, 3)).astype(np.float32), dev), T(rng.uniform(-.5, .5, (1, 99, 3)).astype(np.float32), dev)
    for a, b in zip(torch.ops.macarons.knn_gather_offset(X, pc, 16), ops.knn_points(X, pc, 16, True)):
        assert torch.equal(a, b)
    assert torch.equal(torch.ops.macarons.view_state(X, cams[0, :3].contiguous(), 7, 14), ops.view_state(X, cams[0, :3].contiguous(), 7, 14))
    m, _ = _mod(SconeVis, 1, dev)
    with torch.no_grad():
        y = torch.ops.macarons.scone_vis_forward(pts, harm, m.weight_table())
        assert torch.equal(y, m(pts, view_harmonics=harm))
    # the rest of the operator list (C++ TORCH_LIBRARY shims over the same C ABI): frustum, harmonics product, sampler, SconeOcc
    from macarons_amd.networks import SconeOcc
    from macarons_amd.networks.packing import pack_local_pct
    from macarons_amd import _lib
    from macarons_amd.utility.macarons_utils import camera_record
    g = golden("fov_camera")
    recs = torch.stack([camera_record(g["Mview"][c], g["Mfull"][c], g["ndc"], g["center"][c], 40.0) for c in range(2)]).to(dev)
    P3 = T(g["pts"][:5000], dev)
    assert torch.equal(torch.ops.macarons.points_in_fov(P3, recs), ops.points_in_fov(P3, recs))
    vs = ops.view_state(X, cams[0, :3].contiguous(), 7, 14)
    mat = T(rng.standard_normal((64, 98)).astype(np.float32), dev)
    assert torch.equal(torch.ops.macarons.view_harmonics(vs, mat), ops.linear(vs, mat))
    Xp, pr = T(rng.uniform(-.5, .5, (900, 3)).astype(np.float32), dev), T(rng.uniform(0, 1, 900).astype(np.float32), dev)
    vh = T((rng.standard_normal((900, 64)) * .3).astype(np.float32), dev)
    u = T(rng.uniform(0, 1, 256).astype(np.float32), dev)
    for a, b in zip(torch.ops.macarons.sample_proxy(Xp, pr, vh, u, 0.1), ops.sample_proxy(Xp, pr, vh, u, 0.1)):
        assert torch.equal(a, b)
    occ, _ = _mod(SconeOcc, 2, dev)
    torch.manual_seed(2)
    perms = occ.draw_perms(99)
    xq, vq = T(rng.uniform(-.5, .5, (1, 70, 3)).astype(np.float32), dev), T((rng.standard_normal((1, 70, 64)) * .3).astype(np.float32), dev)
    pcg = pc[:, perms[0].to(dev)].contiguous()
    scales = [pc, pc[:, perms[1].to(dev)].contiguous()]
    scales.append(scales[1][:, perms[2].to(dev)].contiguous())
    blobs = [pack_local_pct(t_, _lib.lib().mcr_get_local_pct_variant()) for t_ in occ.local_transformers]
    with torch.no_grad():
        yo = torch.ops.macarons.scone_occ_forward(pcg, scales, xq, vq, occ.weight_table(), blobs)
        occ.range_guard = "off"
        ref = occ(pc, xq, vq, perms=perms)
    assert rel_err(yo.cpu().numpy(), ref.cpu().numpy()) < 2e-6        # (the module passes host-split head planes, the operator lets the kernel split)
    with pytest.raises(RuntimeError):
        torch.ops.macarons.sh_coverage_gain(pts, harm[:, :10], cams, True)           # TORCH_CHECK on a shape mismatch


@pytest.mark.parametrize("B,M,Q", [(1, 10240, 30000), (2, 1500, 9000), (1, 300, 5000)])
def test_scone_occ_two_call_forward_equals_the_single_call(dev, B, M, Q):
    """forward_begin(pc, x) + forward(..., begun=handle) (mcr_scone_occ_forward_phase 1 / 2: scale 0 queued before the caller has the
    view harmonics) returns the very bits of the single call -- with other work on the stream between the two calls, on the
    grid-pruned and the brute-force search, and a handle is refused when it was made for other tensors."""
    from macarons_amd.networks import SconeOcc
    from macarons_amd import ops
    m, _ = _mod(SconeOcc, 2, dev)
    rng = np.random.default_rng(M + Q)
    pc = T(rng.uniform(-.4, .4, (B, M, 3)).astype(np.float32), dev)
    x = T(rng.uniform(-.5, .5, (B, Q, 3)).astype(np.float32), dev)
    vh = T((rng.standard_normal((B, Q, 64)) * .3).astype(np.float32), dev)
    torch.manual_seed(5)
    perms = m.draw_perms(M)
    with torch.no_grad():
        one = m(pc, x, vh, perms=perms).clone()
        h = m.forward_begin(pc, x)
        assert h is not None
        ops.knn_points(x[:, :4096].contiguous(), pc, 16)          # a workspace-using op in between must not disturb the first part
        scratch = torch.randn(1 << 20, device=dev).sum()          # nor an allocation + kernel of torch's
        two = m(pc, x, vh, perms=perms, begun=h)
        assert torch.equal(one, two)
        other = x.clone()
        h2 = m.forward_begin(pc, other)
        three = m(pc, x, vh, perms=perms, begun=h2)               # made for another tensor: ignored, the whole forward runs
        assert torch.equal(one, three)
        # a handle made for ANOTHER CLOUD: nothing of it may be used -- not its scale-0 cloud either
        pc_other = T(rng.uniform(-.4, .4, (B, M, 3)).astype(np.float32), dev)
        h3 = m.forward_begin(pc_other, x)
        four = m(pc, x, vh, perms=perms, begun=h3)
        assert torch.equal(one, four)
        # a valid handle whose arena was overwritten by another forward on the stream in between: stale, ignored
        h4 = m.forward_begin(pc, x)
        m(pc_other, other[:, :Q // 2].contiguous(), vh[:, :Q // 2].contiguous(), perms=perms)
        five = m(pc, x, vh, perms=perms, begun=h4)
        assert torch.equal(one, five)
        h5 = m.forward_begin(pc, x)
        m.forward_begin(pc_other, other)                          # ... or by another phase 1
        six = m(pc, x, vh, perms=perms, begun=h5)
        assert torch.equal(one, six)
    assert float(scratch) == float(scratch)


def test_masked_attention_matches_reference(dev):
    """attention / Encoder / SconeVis.forward / PCTransformer.forward with a MASK vs the reference's own outputs (make_golden.py:
    gen_masked; Attention.py:24-27: the score of a masked pair is replaced by -1e3 BEFORE the division by sqrt(d) -- not -inf: a
    query with every key masked attends uniformly).  [B,1,N,N] masks on the 16-token kernel, the MFMA kernel (130 tokens: a ragged
    last tile) and its key-split form (520 tokens), a [N,N] mask shared by batch and heads, the [B,N,N] reading and a [B,N] key mask."""
    from macarons_amd.networks import SconeVis, Attention as A
    from macarons_amd.networks.SconeOcc import PCTransformer
    from macarons_amd import ops
    g = golden("blocks_masked")
    f = lambda a: T(np.asarray(a, np.float32), dev)
    um = lambda key, shape: np.unpackbits(g[key])[:int(np.prod(shape))].reshape(shape).astype(bool)
    with torch.no_grad():
        for tag, (E, qk, N, Bb) in {"vis": (256, 64, 130, 1), "occ": (128, 32, 16, 9), "long": (128, 32, 520, 1)}.items():
            mask = um(f"{tag}_mask", (Bb, 1, N, N))
            y = A.attention(f(g[f"{tag}_q"]), f(g[f"{tag}_k"]), f(g[f"{tag}_v"]), mask=torch.from_numpy(mask)).cpu().numpy()
            assert rel_err(y if tag != "long" else y[:, :, ::4], g[f"{tag}_att"]) < TOL, tag
            vmean = np.asarray(g[f"{tag}_v"], np.float32)[0].mean(axis=1)
            assert rel_err(y[0, :, 3], vmean) < TOL, tag                          # the fully masked query: uniform weights (-1e3, not -inf)
            y3 = A.attention(f(g[f"{tag}_q"]), f(g[f"{tag}_k"]), f(g[f"{tag}_v"]), mask=torch.from_numpy(mask[:, 0]).to(dev)).cpu().numpy()
            assert np.array_equal(y3, y), tag                                       # [B,N,N] on the device == [B,1,N,N] from the host
            if tag != "long":
                enc, _ = _mod(lambda: A.Encoder(seq_len=N, qk_dim=qk, embedding_dim=E, n_heads=4), 100 + E, dev)
                assert rel_err(enc(f(g[f"{tag}_x"]), mask=torch.from_numpy(mask)).cpu().numpy(), g[f"{tag}_enc"]) < TOL, tag
        shared = um("occ_mask", (9, 1, 16, 16))[0, 0]
        y = A.attention(f(g["occ_q"]), f(g["occ_k"]), f(g["occ_v"]), mask=torch.from_numpy(shared)).cpu().numpy()
        assert rel_err(y, g["occ_att_shared"]) < TOL
        # a [B,N] key mask == the [B,1,N,N] mask with every query row equal to it
        km = um("vis_mask", (1, 1, 130, 130))[:, 0, 9]                   # one row of the random mask as the key mask
        q, k, v = f(g["vis_q"]), f(g["vis_k"]), f(g["vis_v"])
        full = np.broadcast_to(km[:, None, None, :], (1, 1, 130, 130)).copy()
        a_ = A.attention(q, k, v, mask=torch.from_numpy(full))
        b_ = A.attention(q, k, v, mask=ops.key_mask(torch.from_numpy(km), 1, 130, dev))
        assert torch.equal(a_, b_)
        vis, _ = _mod(SconeVis, 1, dev)
        m = um("sv_mask", (2, 1, 150, 150))
        y = vis(f(g["sv_pts"]), mask=torch.from_numpy(m), view_harmonics=f(g["sv_vh"])).cpu().numpy()
        assert rel_err(y, g["sv_y"]) < TOL
        all_on = vis(f(g["sv_pts"]), mask=torch.ones(2, 1, 150, 150, dtype=torch.bool), view_harmonics=f(g["sv_vh"]))
        assert rel_err(all_on.cpu().numpy(), vis(f(g["sv_pts"]), view_harmonics=f(g["sv_vh"])).cpu().numpy()) < 2e-6    # mask of ones == no mask
        pct, _ = _mod(lambda: PCTransformer(seq_len=150, pts_embedding_dim=128, feature_dim=512), 12, dev)
        assert rel_err(pct(f(g["pct_pc"]), mask=torch.from_numpy(m)).cpu().numpy(), g["pct_y"]) < TOL
        with pytest.raises(ValueError):
            A.attention(q, k, v, mask=torch.ones(3, 130))


def test_one_cloud_returns_the_bits_it_has_inside_a_batch_of_30(dev):
    """Launch-shape independence of the 2048-token encoders at the sizes of a MACARONS decision: SconeVis on ONE cloud (narrow column
    tiles, key-split attention) must return, bit for bit, the rows it returns for that cloud inside a batch of 30 (the neighbour
    cameras: wide tiles) -- at 2048 tokens and at a ragged 1500."""
    from macarons_amd.networks import SconeVis
    vis, _ = _mod(SconeVis, 1, dev)
    rng = np.random.default_rng(77)
    for N in (2048, 1500):
        pts = T(np.concatenate([rng.uniform(-.5, .5, (30, N, 3)), rng.uniform(.1, 1, (30, N, 1))], -1).astype(np.float32), dev)
        vh = T((rng.standard_normal((30, N, 64)) * .3).astype(np.float32), dev)
        with torch.no_grad():
            big = vis(pts, view_harmonics=vh)
            for b in (0, 17, 29):
                one = vis(pts[b:b + 1].contiguous(), view_harmonics=vh[b:b + 1].contiguous())
                assert torch.equal(one[0], big[b]), (N, b)


def test_stand_alone_forwards_can_be_captured_under_the_default_guard(dev):
    """The default range guard reads the flag back after a stand-alone forward -- which a stream capture forbids: under capture both
    networks leave the flag in range_flag() instead ("defer"), so a user who records `scone_occ(...)` / `scone_vis(...)` in a graph of
    their own is not broken by the default; the replay reproduces the eager bits."""
    from macarons_amd.networks import SconeOcc, SconeVis
    occ, _ = _mod(SconeOcc, 2, dev)
    vis, _ = _mod(SconeVis, 1, dev)
    assert occ.range_guard == "sync" and vis.range_guard == "sync"
    g = golden("scone_occ")
    tag = "m1024_q300"
    perms = [torch.from_numpy(g[f"{tag}_perm{i}"].astype(np.int64)).to(dev) for i in range(3)]
    pc, x, vh = T(g[f"{tag}_pc"], dev), T(g[f"{tag}_x"], dev), T(g[f"{tag}_vh"], dev)
    rng = np.random.default_rng(3)
    pts = T(np.concatenate([rng.uniform(-.5, .5, (1, 700, 3)), rng.uniform(.1, 1., (1, 700, 1))], -1).astype(np.float32), dev)
    vhs = T((rng.standard_normal((1, 700, 64)) * 0.3).astype(np.float32), dev)
    with torch.no_grad():
        y_e, h_e = occ(pc, x, vh, perms=perms), vis(pts, view_harmonics=vhs)          # eager (also builds every cache and arena)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            occ(pc, x, vh, perms=perms); vis(pts, view_harmonics=vhs)                  # warm-up on the capture stream
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            y_g, h_g = occ(pc, x, vh, perms=perms), vis(pts, view_harmonics=vhs)
        graph.replay()
        torch.cuda.synchronize()
    assert torch.equal(y_g, y_e) and torch.equal(h_g, h_e)
    assert int(occ.range_flag()) == 0
