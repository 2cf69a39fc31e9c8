"""Weight stream of the shelved register-resident local transformer (tools/experiments/local_pct8.hip): fp16 hi/lo planes of the
power-of-two scaled weights as 1 KB fragment blocks in consumption order.  Kept with the kernel; not part of the product."""
import torch

from macarons_amd.networks.packing import _pad, _padv, _pow2_scale


# ---- variant 8: weight stream of local_pct8.hip ----------------------------------------------------------------------------
L8_GROUP_BLOCKS = 33          # 1 header block + 32 weight blocks of 1 KB
L8_GROUPS = 1 + 2 + 2 * (3 + 2 + 9) + 2


def _frags16(W, scale):
    """[N, K] fp32 (N % 16 == 0, K % 32 == 0) -> fp16 hi/lo fragments of W * scale for v_mfma_f32_16x16x32_f16, one 1 KB block
    (256 floats) per (tile t, k-step s, plane): [N/16, K/32, 2, 256].  Lane (i = lane & 15, kg = lane >> 4), element e holds
    W[16 t + i][32 s + (4 kg + e if e < 4 else 16 + 4 kg + e - 4)]: the k order in which a lane of the previous product's result
    (features 4 g + r of the 16-feature tiles 2 s and 2 s + 1) holds its eight values."""
    N, K = W.shape
    assert N % 16 == 0 and K % 32 == 0
    Ws = W * scale
    hi = Ws.to(torch.float16)
    lo = (Ws - hi.float()).to(torch.float16)
    planes = torch.stack([hi.view(torch.int16), lo.view(torch.int16)], 0)              # [2, N, K]
    kg = torch.arange(4).view(4, 1)
    e = torch.arange(8).view(1, 8)
    off = torch.where(e < 4, 4 * kg + e, 16 + 4 * kg + e - 4).reshape(-1).to(W.device)    # [32] = (kg, e) -> offset in the step
    t = planes.reshape(2, N // 16, 16, K // 32, 32)[..., off]                          # [pl, t, i, s, (kg, e)]
    t = t.reshape(2, N // 16, 16, K // 32, 4, 8).permute(1, 3, 0, 4, 2, 5).contiguous()  # [t, s, pl, kg, i, e]
    return t.reshape(N // 16, K // 32, 2, 512).view(torch.float32)                      # 512 halves = 256 floats


def _pack_local_pct8(pct):
    """Stream of L8_GROUPS groups x 33 blocks x 256 floats in the order local_pct8.hip consumes them (layout of every group in
    that file's header)."""
    with torch.no_grad():
        f = lambda p: p.detach().float()
        dev = pct.linear0.weight.device
        groups = []

        def group():
            g = torch.zeros(L8_GROUP_BLOCKS, 256, dtype=torch.float32, device=dev)
            groups.append(g)
            return g

        def tiles_group(g, F, t0, nt, bias, scale):
            """tile-outer group: tiles t0 .. t0 + nt of F [T, S, 2, 256]; slot = (tl * S + s) * 2 + pl; header: bias of those tiles
            at [16 tl ..], 1 / scale at [128], scale at [129]."""
            S = F.shape[1]
            for tl in range(nt):
                for s_ in range(S):
                    for pl in range(2):
                        g[1 + (tl * S + s_) * 2 + pl] = F[t0 + tl, s_, pl]
            g[0, :16 * nt] = bias[16 * t0:16 * (t0 + nt)] * scale
            g[0, 128], g[0, 129] = 1.0 / scale, scale

        def ksteps_group(g, F, s0, ns, bias, scale, slot0=0, hdr0=0):
            """k-outer group: k-steps s0 .. s0 + ns for all 8 output tiles; slot = slot0 + (sl * 8 + t) * 2 + pl; header: the 128
            biases at [hdr0 ..]."""
            for sl in range(ns):
                for t in range(8):
                    for pl in range(2):
                        g[1 + slot0 + (sl * 8 + t) * 2 + pl] = F[t, s0 + sl, pl]
            g[0, hdr0:hdr0 + 128] = bias * scale

        emb = pct.embedding
        w = _pad(f(emb.linear1.weight), 128, 32)
        s1 = _pow2_scale(w)
        tiles_group(group(), _frags16(w, s1), 0, 8, _padv(f(emb.linear1.bias), 128), s1)
        w = _pad(f(emb.linear2.weight), 128, 128)
        s2 = _pow2_scale(w)
        F = _frags16(w, s2)
        for h in range(2):
            g = group()
            ksteps_group(g, F, 2 * h, 2, _padv(f(emb.linear2.bias), 128), s2)
            g[0, 128], g[0, 129] = 1.0 / s2, s2
        for enc in pct.encoders:
            g1, b1 = f(enc.norm1.weight), f(enc.norm1.bias)
            g2, b2 = f(enc.norm2.weight), f(enc.norm2.bias)
            wqkv = torch.cat((f(enc.mhsa.w_q.weight), f(enc.mhsa.w_k.weight), f(enc.mhsa.w_v.weight)), 0)
            bqkv = torch.cat((f(enc.mhsa.w_q.bias), f(enc.mhsa.w_k.bias), f(enc.mhsa.w_v.bias)), 0) + wqkv @ b1
            wq = wqkv * g1[None, :]
            sq = _pow2_scale(wq)
            F = _frags16(wq, sq)
            for h in range(3):
                tiles_group(group(), F, 4 * h, 4, bqkv, sq)
            wo = f(enc.mhsa.out.weight)
            so = _pow2_scale(wo)
            F = _frags16(wo, so)
            for h in range(2):
                g = group()
                ksteps_group(g, F, 2 * h, 2, f(enc.mhsa.out.bias), so)
                g[0, 128], g[0, 129] = 1.0 / so, so
            w1, w2 = f(enc.ff.linear1.weight), f(enc.ff.linear2.weight)
            c1 = f(enc.ff.linear1.bias) + w1 @ b2
            w1 = w1 * g2[None, :]
            sa, sb = _pow2_scale(w1), _pow2_scale(w2)
            F1, F2 = _frags16(w1, sa), _frags16(w2, sb)             # F1 [16, 4, 2, 256], F2 [8, 8, 2, 256]
            for p_ in range(9):
                g = group()
                if p_ < 8:                                          # FF1 hidden tiles 2p, 2p + 1 -> slots 0 .. 15, bias at [0 .. 32)
                    for tl in range(2):
                        for s_ in range(4):
                            for pl in range(2):
                                g[1 + (tl * 4 + s_) * 2 + pl] = F1[2 * p_ + tl, s_, pl]
                    g[0, :32] = c1[32 * p_:32 * p_ + 32] * sa
                if p_ > 0:                                          # FF2 k-step p - 1 -> slots 16 .. 31
                    for t in range(8):
                        for pl in range(2):
                            g[1 + 16 + t * 2 + pl] = F2[t, p_ - 1, pl]
                g[0, 64:192] = f(enc.ff.linear2.bias) * sb
                g[0, 192], g[0, 193], g[0, 194] = 1.0 / sa, 1.0 / sb, sb
        gn, bn = f(pct.norm.weight), f(pct.norm.bias)
        w0 = f(pct.linear0.weight)
        b0 = f(pct.linear0.bias) + w0 @ bn
        w0 = w0 * gn[None, :]
        s0 = _pow2_scale(w0)
        F = _frags16(w0, s0)
        for h in range(2):
            tiles_group(group(), F, 4 * h, 4, b0, s0)
        assert len(groups) == L8_GROUPS
        blob = torch.stack(groups).reshape(-1).contiguous()
    assert blob.numel() == L8_GROUPS * L8_GROUP_BLOCKS * 256
    return blob
