"""Host-side weight packing of the fused local transformers (no GPU): blob sizes match what the kernels expect and the fp16
hi/lo planes carry the scaled weights to 2^-22."""
import io
import contextlib

import numpy as np
import torch


def _pct():
    from macarons_amd.networks import SconeOcc
    torch.manual_seed(3)
    with contextlib.redirect_stdout(io.StringIO()):
        occ = SconeOcc()
    return occ.local_transformers[0]


def test_blob_sizes_match_the_kernels():
    from macarons_amd import _lib
    from macarons_amd.networks.packing import pack_local_pct
    pct = _pct()
    L = _lib.lib()
    assert pack_local_pct(pct, 1).numel() == L.mcr_local_pct_blob_floats()
    assert pack_local_pct(pct, 5).numel() == L.mcr_local_pct3_blob_floats()
    assert pack_local_pct(pct, 6).numel() == L.mcr_local_pct6_blob_floats()
    assert pack_local_pct(pct, 8).numel() == L.mcr_local_pct8_blob_floats() == 33 * 33 * 256


def test_stream_fragments_reconstruct_the_scaled_weights():
    """_frags16: lane (i, kg), element e of block (tile t, k-step s) holds W[16 t + i][32 s + (4 kg + e | 16 + 4 kg + e - 4)];
    hi + lo reproduces W * scale to 2^-22 relative (two-term fp16 split)."""
    from macarons_amd.networks.packing import _frags16, _pow2_scale
    rng = np.random.default_rng(0)
    W = torch.from_numpy((rng.standard_normal((48, 64)) * 0.1).astype(np.float32))
    sc = _pow2_scale(W)
    F = _frags16(W, sc)                                              # [3 tiles, 2 steps, 2 planes, 256 floats]
    assert F.shape == (3, 2, 2, 256)
    h = F.view(torch.int16).reshape(3, 2, 2, 64, 8).view(torch.float16).double()          # [t, s, pl, lane, e]
    rec = h[:, :, 0] + h[:, :, 1]
    for t in range(3):
        for s in range(2):
            for lane in (0, 5, 17, 38, 63):
                i, kg = lane & 15, lane >> 4
                for e in range(8):
                    k = 32 * s + (4 * kg + e if e < 4 else 16 + 4 * kg + e - 4)
                    want = float(W[16 * t + i, k]) * sc
                    assert abs(float(rec[t, s, lane, e]) - want) <= 2.0 ** -21 * abs(want) + 2.0 ** -24


def test_stream_group_headers():
    """Every group of the v8 stream is 33 blocks of 256 floats; the headers carry the pre-scaled biases and the power-of-two
    scales (and their exact inverses)."""
    from macarons_amd.networks.packing import pack_local_pct, L8_GROUPS, L8_GROUP_BLOCKS
    pct = _pct()
    g = pack_local_pct(pct, 8).reshape(L8_GROUPS, L8_GROUP_BLOCKS, 256)
    # group 0 = emb1: 1/scale at [128], scale at [129], bias * scale at [0..125)
    inv, sc = float(g[0, 0, 128]), float(g[0, 0, 129])
    assert inv * sc == 1.0 and np.log2(sc) == round(np.log2(sc))
    b = pct.embedding.linear1.bias.detach().float()
    assert torch.equal(g[0, 0, :b.numel()], b * sc)
    # FF groups of the first encoder: 1 + 2 + 3 + 2 = 8 .. 16; FF2 bias (x its scale) in every one of them at [64..192)
    b2 = pct.encoders[0].ff.linear2.bias.detach().float()
    for p in range(9):
        sb = float(g[8 + p, 0, 194])
        assert float(g[8 + p, 0, 193]) * sb == 1.0
        assert torch.equal(g[8 + p, 0, 64:192], b2 * sb)
