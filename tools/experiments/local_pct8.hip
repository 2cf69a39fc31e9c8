// K5-local v8 -- fused per-query local PCTransformer of SconeOcc, REGISTER-RESIDENT: one wave = one query (16 neighbours x 128
// channels), activations never leave the wave's registers between the xyz offsets and the pooled feature; LDS carries nothing
// but the weight stream.  Same numerics as local_pct6.hip (two-term fp16 split, three MFMAs per fp32 product, power-of-two
// weight scales, exact-erf GELU, fp32 attention on v_mfma_f32_16x16x4_f32); reference mapping in local_pct.hip
// (SconeOcc.py:104-130: Embedding -> 2 x Encoder -> LayerNorm -> linear0 -> max || avg pool).
//
// Why it exists.  In v6 a workgroup of four waves shares 64 tokens: every product is followed by an epilogue that writes fp16
// planes to LDS, a barrier, and fragment reads by all waves.  Here a wave owns its 16 tokens outright and evaluates every product
// TRANSPOSED on v_mfma_f32_16x16x32_f16, D^T[feature][token] = W[feature][:] . act[token][:]: in the result lane (token j,
// g = lane >> 4) holds features 4g .. 4g+3 of each 16-feature tile.  Two adjacent tiles of that result ARE the B operand (8
// k-values per lane) of the next product's k-step once the weights' k order is permuted to match on the host
// (networks/packing.py:_frags16) -- no transpose, no LDS round trip, no barrier between products; LayerNorm is 32 registers + two
// lane swaps; the attention takes q | k (transposed tiles) and v (untransposed tiles: activations as the A operand) straight
// from the QKV accumulators.  Products alternate between two loop orders so they chain through registers: "tile-outer" (all of
// K for a pair of output tiles: emb1, QKV, FF1, linear0) produces its output pair by pair, "k-outer" (one k-step for all 8 output
// tiles: emb2, out-projection, FF2) consumes its input pair by pair and accumulates into the residual registers in place.
// FF1 -> GELU -> FF2 is fused per hidden pair: the 256-wide hidden layer never exists; GELU of pair p-1 is issued beside the
// MFMAs of FF1 pair p.
//
// Weights.  The NW = 8 waves (= queries) of a workgroup need the same weights at about the same time: they stream through a
// double-buffered LDS ring in GROUPS of 33 KB (1 KB header: biases and scales; 32 blocks of 1 KB = one (tile, k-step, plane)
// fragment each, lane-linear), filled by global_load_lds_dwordx4 (no registers, 4-5 instructions per wave and group) one group
// ahead; one __syncthreads() per group is the only synchronisation.  Group g+2 is requested right after the barrier that retires
// group g.  Every weight byte fetched from L2 serves 8 queries (4 in v6).  Stream order = consumption order (33 groups, 1.09 MB
// per transformer):
//   emb1 [8 tiles x 1 step] | emb2 [2 x (2 k-steps x 8 tiles)] | per encoder: QKV [3 x 4 tiles] , out [2 x 2 k-steps] ,
//   FF [9 x (FF1 hidden pair p in slots 0..15 , FF2 k-step p-1 in slots 16..31)] | linear0 [2 x 4 tiles]
// LDS = 66 KB, 512 threads, one workgroup per CU (2 waves per SIMD, <= 256 VGPRs).
//
// Status: same accuracy as v6 (7e-7 against the fp64 oracle) and the SAME time within 2 % (0.52-0.53 ms per
// 16 384 queries; SconeOcc.forward 11.5 ms either way) although it moves no activation through LDS, halves the L2 weight
// traffic and uses a third fewer barriers -- the two kernels execute the same MFMAs and the same GELU / LayerNorm / softmax
// arithmetic, and on this part both are held at ~0.98 M shader cycles per launch with the shader clock managed down to
// 1.8-1.9 GHz by the operand data (the identical instruction stream on all-zero weights runs at 2.17 GHz, 19 % faster).  v6 stays
// the default; this variant (mcr_set_local_pct_variant(8)) is kept selectable as the better starting point for hand scheduling.
#include "lp_split.h"

namespace mcr {
namespace v8 {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int GB = 33, NG = 33, L8_BLOB_FLOATS = NG * GB * 256;

__device__ __forceinline__ int opaque(int v) {
    asm volatile("" : "+v"(v));
    return v;
}
__device__ __forceinline__ f32x4 mfma16(uint4 a, uint4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
template <class Op>
__device__ __forceinline__ float rows_allreduce(float v, Op op) {           // over lanes l, l^16, l^32, l^48
    const unsigned b = __builtin_bit_cast(unsigned, v);
    const auto r = __builtin_amdgcn_permlane16_swap(b, b, false, false);
    v = op(__builtin_bit_cast(float, (unsigned)r[0]), __builtin_bit_cast(float, (unsigned)r[1]));
    const unsigned c = __builtin_bit_cast(unsigned, v);
    const auto q = __builtin_amdgcn_permlane32_swap(c, c, false, false);
    return op(__builtin_bit_cast(float, (unsigned)q[0]), __builtin_bit_cast(float, (unsigned)q[1]));
}

// two adjacent 16-feature tiles of a transposed result = the 8 k-values per lane of one k-step of the next product
struct Pk { uint4 hi, lo; };
__device__ __forceinline__ Pk pack_pair(const f32x4 a, const f32x4 b) {
    Pk p;
    split2h(a[0], a[1], p.hi.x, p.lo.x);
    split2h(a[2], a[3], p.hi.y, p.lo.y);
    split2h(b[0], b[1], p.hi.z, p.lo.z);
    split2h(b[2], b[3], p.hi.w, p.lo.w);
    return p;
}

// the current group in LDS: W(slot) = this lane's 16 bytes of weight block `slot`; header floats
struct Grp {
    const uint4* w;            // ring buffer + lane (block 0 = header)
    const float* h;            // header
    __device__ __forceinline__ uint4 W(int slot) const { return w[(1 + slot) * 64]; }
    __device__ __forceinline__ f32x4 bias4(int off, int g) const {      // floats off + 4 g .. + 4
        const float4 b = *reinterpret_cast<const float4*>(h + off + 4 * g);
        return f32x4{b.x, b.y, b.z, b.w};
    }
};

// ---- matrix work comes in UNITS of 4 or 8 weight fragments; every group runs as a fenced software pipeline in program order:
//   [ds_reads of unit u+1] | [MFMAs of unit u, with whatever vector work belongs beside them] | ...
// The fences (sched_barrier) keep the fragment reads one unit ahead of their use -- left alone, LLVM sinks every ds_read next to
// its first use and each MFMA waits out the LDS latency -- and the empty asm pins the accumulator chains in program order (the
// MFMA builtins are pure: instruction selection otherwise moves whole chains across the fences).
#define L8_FENCE() __builtin_amdgcn_sched_barrier(0)
__device__ __forceinline__ void pin(f32x4& a, f32x4& b) { asm volatile("" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void pin(f32x4& a, f32x4& b, f32x4& c, f32x4& d) { asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }

// tile-outer unit: k-step s of a pair of output tiles; slots (tl * S + s) * 2 + plane from slot0
struct FT { uint4 w0h, w0l, w1h, w1l; };
template <int S>
__device__ __forceinline__ FT load_T(const Grp& G, int slot0, int s) {
    return FT{G.W(slot0 + s * 2), G.W(slot0 + s * 2 + 1), G.W(slot0 + (S + s) * 2), G.W(slot0 + (S + s) * 2 + 1)};
}
// TR: transposed (weights = A operand: lane (token, g) <- features 4g..4g+3), else untransposed (activations = A operand: lane
// (feature, g) <- tokens 4g..4g+3)
template <bool TR>
__device__ __forceinline__ void mma_T(f32x4& a0, f32x4& a1, const FT& f, const Pk& x) {
    if (TR) {
        a0 = mfma16(f.w0l, x.hi, a0);                      // smallest terms first
        a1 = mfma16(f.w1l, x.hi, a1);
        a0 = mfma16(f.w0h, x.lo, a0);
        a1 = mfma16(f.w1h, x.lo, a1);
        a0 = mfma16(f.w0h, x.hi, a0);
        a1 = mfma16(f.w1h, x.hi, a1);
    } else {
        a0 = mfma16(x.hi, f.w0l, a0);
        a1 = mfma16(x.hi, f.w1l, a1);
        a0 = mfma16(x.lo, f.w0h, a0);
        a1 = mfma16(x.lo, f.w1h, a1);
        a0 = mfma16(x.hi, f.w0h, a0);
        a1 = mfma16(x.hi, f.w1h, a1);
    }
    pin(a0, a1);
}
// k-outer unit: one k-step (input pair x) for 4 of the 8 output tiles (half hf); slots slot0 + t * 2 + plane
struct FK { uint4 h[4], l[4]; };
__device__ __forceinline__ FK load_K(const Grp& G, int slot0, int hf) {
    FK f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f.h[i] = G.W(slot0 + (4 * hf + i) * 2);
        f.l[i] = G.W(slot0 + (4 * hf + i) * 2 + 1);
    }
    return f;
}
__device__ __forceinline__ void mma_K(f32x4 (&acc)[8], int hf, const FK& f, const Pk& x) {
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[4 * hf + i] = mfma16(f.l[i], x.hi, acc[4 * hf + i]);
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[4 * hf + i] = mfma16(f.h[i], x.lo, acc[4 * hf + i]);
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[4 * hf + i] = mfma16(f.h[i], x.hi, acc[4 * hf + i]);
    pin(acc[4 * hf], acc[4 * hf + 1], acc[4 * hf + 2], acc[4 * hf + 3]);
}
// a whole group of 4 tiles (two pairs, 4 k-steps each = 8 units); f = unit 0, already requested
template <bool TR>
__device__ __forceinline__ void tiles4(f32x4& a0, f32x4& a1, f32x4& a2, f32x4& a3, const Grp& G, const Pk (&x)[4], FT f) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        FT n = f;
        if (u < 7) n = load_T<4>(G, ((u + 1) >> 2) * 16, (u + 1) & 3);
        L8_FENCE();
        if (u < 4) mma_T<TR>(a0, a1, f, x[u & 3]);
        else mma_T<TR>(a2, a3, f, x[u & 3]);
        L8_FENCE();
        f = n;
    }
}
// a whole group of 2 k-steps for all 8 output tiles (4 units); f = unit 0, already requested
__device__ __forceinline__ void ksteps2(f32x4 (&acc)[8], const Grp& G, const Pk& x0, const Pk& x1, FK f) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        FK n = f;
        if (u < 3) n = load_K(G, ((u + 1) >> 1) * 16, (u + 1) & 1);
        L8_FENCE();
        mma_K(acc, u & 1, f, u < 2 ? x0 : x1);
        L8_FENCE();
        f = n;
    }
}

// LayerNorm over the 128 channels of each token (eps 1e-5; gamma / beta are folded into the next weights): 32 registers per lane
// x the 4 lanes of a token; two-pass in registers; result split and packed as the next product's B operands
__device__ __forceinline__ void layer_norm(const f32x4 (&x)[8], Pk (&o)[4]) {
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t) s += (x[t][0] + x[t][1]) + (x[t][2] + x[t][3]);
    s = rows_allreduce(s, [](float a, float b) { return a + b; });
    const float mu = s * (1.0f / 128.f);
    float q = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float d = x[t][r] - mu;
            q = fmaf(d, d, q);
        }
    q = rows_allreduce(q, [](float a, float b) { return a + b; });
    const float rstd = 1.0f / sqrtf(q * (1.0f / 128.f) + 1e-5f);
#pragma unroll
    for (int p = 0; p < 4; ++p) o[p] = pack_pair((x[2 * p] - mu) * rstd, (x[2 * p + 1] - mu) * rstd);
}

__device__ __forceinline__ f32x4 gelu4(const f32x4 v, float sc) {
    return f32x4{l3_gelu(v[0] * sc), l3_gelu(v[1] * sc), l3_gelu(v[2] * sc), l3_gelu(v[3] * sc)};
}

#ifdef L8_TRACE             // dev only: per-group timestamps of one wave (tools/build_variant.py ... -DL8_TRACE=<block>, tools/trace_local_pct8.py)
__device__ long long l8_trace_buf[160];
#define L8_T() do { if (blockIdx.x == (L8_TRACE) && tid == 0) { l8_trace_buf[tp] = clock64(); } ++tp; } while (0)
#else
#define L8_T() do { } while (0)
#endif

// grid = ceil(S / NW); S sequences of 16 offsets [S,16,3]; features[s*ld_feat + 0:256] = max(128) || avg(128)
#ifndef L8_NW
#define L8_NW 8              // waves = queries per workgroup: every weight byte fetched from L2 serves NW queries
#endif
constexpr int NW = L8_NW;
__global__ __launch_bounds__(64 * NW, NW <= 4 ? 2 : 1) void local_pct8_kernel(const float* __restrict__ offs, float* __restrict__ feat,
                                                          long long ld_feat, long long S,
                                                          const float* __restrict__ blob) {
    __shared__ __attribute__((aligned(16))) uint4 ring[2 * GB * 64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long long q = (long long)blockIdx.x * NW + wave;     // this wave's query
    const int li = lane & 15, g4 = lane >> 4;

    // ---- weight stream ----
    const uint4* src = reinterpret_cast<const uint4*>(blob) + lane;
    int g = 0;                                                  // group being consumed
    auto load_group = [&](int gi) {
        const uint4* s_ = src + (size_t)gi * (GB * 64);
        uint4* d = ring + (gi & 1) * (GB * 64);
#pragma unroll
        for (int b = 0; b < (GB + NW - 1) / NW; ++b) {     // block b * NW + wave (block 0 = the header)
            const int blk = b * NW + wave;
            if (blk < GB)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(s_ + blk * 64),
                                                 (__attribute__((address_space(3))) void*)(d + blk * 64), 16, 0, 0);
        }
    };
    auto cur = [&]() {
        const uint4* b = ring + (g & 1) * (GB * 64);
        return Grp{b + opaque(lane), reinterpret_cast<const float*>(b)};
    };
    // end of group g: my DMA of group g+1 has landed (vmcnt), everybody's has and everybody is done reading group g (barrier) ...
    int tp = 0; (void)tp;
    auto sync_next = [&]() {
        L8_T();                                            // work of group g done
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        L8_T();                                            // barrier passed
        ++g;
    };
    // ... and group g+2 goes into the buffer just retired (issued AFTER the first fragment reads of the new group)
    auto request = [&]() {
        L8_FENCE();
        if (g + 1 < NG) load_group(g + 1);
        L8_FENCE();
    };
    load_group(0);
    load_group(1);

    // ---- Embedding (Attention.py:98-128): linear1 3->125 (K padded to 32), GELU ; linear2 125->125 || xyz ----
    float xyz[3] = {0.f, 0.f, 0.f};
    if (q < S) {
        const float* p = offs + (q * 16 + li) * 3;
        xyz[0] = p[0]; xyz[1] = p[1]; xyz[2] = p[2];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    f32x4 xres[8];
    {
        Pk x0;                                             // k = 0..2 = xyz in lane group 0, zeros elsewhere
        x0.hi = make_uint4(0, 0, 0, 0); x0.lo = make_uint4(0, 0, 0, 0);
        const bool k0 = g4 == 0;
        split2h(k0 ? xyz[0] : 0.f, k0 ? xyz[1] : 0.f, x0.hi.x, x0.lo.x);
        split2h(k0 ? xyz[2] : 0.f, 0.f, x0.hi.y, x0.lo.y);
        Pk x1[4];
        {
            const Grp G = cur();
            const float isc = G.h[128];
            FT f = load_T<1>(G, 0, 0);
            f32x4 a0 = G.bias4(0, g4), a1 = G.bias4(16, g4), p0 = a0, p1 = a1;
#pragma unroll
            for (int p = 0; p < 5; ++p) {                  // pair p on the matrix pipe, GELU of pair p-1 beside it
                FT n = f;
                f32x4 b0 = a0, b1 = a1;
                if (p < 3) {
                    n = load_T<1>(G, 4 * (p + 1), 0);
                    b0 = G.bias4(32 * (p + 1), g4); b1 = G.bias4(32 * (p + 1) + 16, g4);
                }
                L8_FENCE();
                if (p < 4) mma_T<true>(a0, a1, f, x0);
                if (p > 0) x1[p - 1] = pack_pair(gelu4(p0, isc), gelu4(p1, isc));
                L8_FENCE();
                p0 = a0; p1 = a1; a0 = b0; a1 = b1; f = n;
            }
        }
        sync_next();
        {
            const Grp G = cur();
            const FK k = load_K(G, 0, 0);
#pragma unroll
            for (int t = 0; t < 8; ++t) xres[t] = G.bias4(16 * t, g4);
            request();
            ksteps2(xres, G, x1[0], x1[1], k);
        }
        sync_next();
        {
            const Grp G = cur();
            const FK k = load_K(G, 0, 0);
            const float isc = G.h[128];
            request();
            ksteps2(xres, G, x1[2], x1[3], k);
#pragma unroll
            for (int t = 0; t < 8; ++t) xres[t] *= isc;
        }
        if (g4 == 3) {                                     // features 125..127 = the raw xyz (Attention.py:123-126): tile 7, g = 3, r = 1..3
            xres[7][1] = xyz[0]; xres[7][2] = xyz[1]; xres[7][3] = xyz[2];
        }
        sync_next();
        request();
    }

    // at the top of every encoder pass (and on leaving the loop) the current group is synchronised and the next one requested
#pragma unroll 1
    for (int e = 0; e < 2; ++e) {
        Pk x[4];
        f32x4 t[12];
        float sc;
        // ---- norm1 (folded) ; QKV (Attention.py:186-188, 287): q, k transposed tiles, v untransposed ----
        {
            const Grp G = cur();
            const FT f = load_T<4>(G, 0, 0);               // in flight during the LayerNorm
            sc = G.h[128];
#pragma unroll
            for (int k = 0; k < 4; ++k) t[k] = G.bias4(16 * k, g4);
            layer_norm(xres, x);
            tiles4<true>(t[0], t[1], t[2], t[3], G, x, f);
        }
#pragma unroll
        for (int hgrp = 0; hgrp < 2; ++hgrp) {
            sync_next();
            const Grp G = cur();
            const FT f = load_T<4>(G, 0, 0);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float b = G.h[16 * k + li];
                t[4 + 4 * hgrp + k] = f32x4{b, b, b, b};
            }
            request();
            tiles4<false>(t[4 + 4 * hgrp], t[5 + 4 * hgrp], t[6 + 4 * hgrp], t[7 + 4 * hgrp], G, x, f);
        }
        sync_next();
        // ---- out projection + residual (Attention.py:201-202, 290): x += att W_o^T + b, accumulated in place; its first weight
        // fragments and the bias are requested before the attention ----
        {
            const Grp G = cur();
            const FK k = load_K(G, 0, 0);
            const float so = G.h[129];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const f32x4 b = G.bias4(16 * j, g4);
#pragma unroll
                for (int r = 0; r < 4; ++r) xres[j][r] = fmaf(xres[j][r], so, b[r]);
            }
            request();
            // ---- attention (Attention.py:8-36) on v_mfma_f32_16x16x4_f32 (exact fp32 fma chains), operands = the accumulators above.
            // Per head hh (features 8 hh .. 8 hh + 7 = lane groups g >> 1 == hh & 1 of tile hh >> 1):
            //   S^T[key][query] = sum over the 4 steps s of  A = k[tile][s] (other head zeroed) , B = q[tile][s]   (d = 4 g + s)
            //   softmax over the keys = 4 registers x the 4 lane groups
            //   O^T[feature][query] = sum_s  A = v[tile][s] , B = p[s]                                          (key = 4 g + s)
            // The accumulators carry the factor 2^e of the scaled weights: scores take 2^-2e, the output 2^-e.
            Pk xo[4];
            {
                const float sscale = 0.35355339059327376220f * sc * sc;
                f32x4 pr[4];
#pragma unroll
                for (int hh = 0; hh < 4; ++hh) {
                    const bool mine = (g4 >> 1) == (hh & 1);
                    f32x4 st = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int s = 0; s < 4; ++s)
                        st = __builtin_amdgcn_mfma_f32_16x16x4f32(mine ? t[2 + (hh >> 1)][s] : 0.f, t[hh >> 1][s], st, 0, 0, 0);
                    float mx = fmaxf(fmaxf(st[0], st[1]), fmaxf(st[2], st[3]));
                    mx = rows_allreduce(mx, [](float a, float b) { return fmaxf(a, b); });
                    mx *= sscale;
                    float den = 0.f;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        st[r] = __expf(fmaf(st[r], sscale, -mx));
                        den += st[r];
                    }
                    den = rows_allreduce(den, [](float a, float b) { return a + b; });
                    const float inv = __builtin_amdgcn_rcpf(den) * sc;
#pragma unroll
                    for (int r = 0; r < 4; ++r) pr[hh][r] = st[r] * inv;
                }
                f32x4 o[8];
#pragma unroll
                for (int vt = 0; vt < 8; ++vt) {
                    o[vt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int s = 0; s < 4; ++s)
                        o[vt] = __builtin_amdgcn_mfma_f32_16x16x4f32(t[4 + vt][s], pr[vt >> 1][s], o[vt], 0, 0, 0);
                }
#pragma unroll
                for (int p = 0; p < 4; ++p) xo[p] = pack_pair(o[2 * p], o[2 * p + 1]);
            }
            ksteps2(xres, G, xo[0], xo[1], k);
            sync_next();
            {
                const Grp G2 = cur();
                const FK k2 = load_K(G2, 0, 0);
                const float iso = G2.h[128];
                request();
                ksteps2(xres, G2, xo[2], xo[3], k2);
#pragma unroll
                for (int j = 0; j < 8; ++j) xres[j] *= iso;
            }
        }
        sync_next();
        // ---- norm2 (folded) ; FF 128 -> 256 (GELU) -> 128 + residual (Attention.py:293-298), fused per hidden pair: group p holds
        // FF1's hidden tiles 2p, 2p+1 and FF2's k-step p-1; GELU of pair p-1 runs beside the MFMAs of FF1 pair p ----
        f32x4 h0, h1;
        float isa, isb;
        {
            const Grp G = cur();
            FT f = load_T<4>(G, 0, 0);
            isa = G.h[192]; isb = G.h[193];
            const float sb = G.h[194];
            h0 = G.bias4(0, g4); h1 = G.bias4(16, g4);
            request();
            layer_norm(xres, x);
#pragma unroll
            for (int j = 0; j < 8; ++j) {                  // FF2 accumulates onto the residual in place
                const f32x4 b = G.bias4(64 + 16 * j, g4);
#pragma unroll
                for (int r = 0; r < 4; ++r) xres[j][r] = fmaf(xres[j][r], sb, b[r]);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                FT n = f;
                if (s < 3) n = load_T<4>(G, 0, s + 1);
                L8_FENCE();
                mma_T<true>(h0, h1, f, x[s]);
                L8_FENCE();
                f = n;
            }
        }
        sync_next();
#pragma unroll 1
        for (int p = 1; p < 8; ++p) {
            const Grp G = cur();
            FT f = load_T<4>(G, 0, 0);
            f32x4 a0 = G.bias4(0, g4), a1 = G.bias4(16, g4);
            request();
            f32x4 g0, g1;
            FK k0;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                FT n = f;
                if (s < 3) n = load_T<4>(G, 0, s + 1);
                else k0 = load_K(G, 16, 0);
                L8_FENCE();
                mma_T<true>(a0, a1, f, x[s]);              // FF1 pair p ...
                g0[s] = l3_gelu(h0[s] * isa);               // ... a quarter of GELU(pair p-1) beside each k-step
                g1[s] = l3_gelu(h1[s] * isa);
                L8_FENCE();
                f = n;
            }
            const Pk hk = pack_pair(g0, g1);
            const FK k1 = load_K(G, 16, 1);
            L8_FENCE();
            mma_K(xres, 0, k0, hk);                         // FF2 k-step p-1
            L8_FENCE();
            mma_K(xres, 1, k1, hk);
            L8_FENCE();
            h0 = a0; h1 = a1;
            sync_next();
        }
        {
            const Grp G = cur();
            const FK k0 = load_K(G, 16, 0);
            request();
            const Pk hk = pack_pair(gelu4(h0, isa), gelu4(h1, isa));
            const FK k1 = load_K(G, 16, 1);
            L8_FENCE();
            mma_K(xres, 0, k0, hk);
            L8_FENCE();
            mma_K(xres, 1, k1, hk);
            L8_FENCE();
#pragma unroll
            for (int j = 0; j < 8; ++j) xres[j] *= isb;
        }
        sync_next();
        request();
    }
    // ---- final norm (folded) + linear0 128 -> 128 (SconeOcc.py:119-122), untransposed: lane (feature, g) <- tokens 4g .. 4g+3 ;
    // max || avg pool over the 16 tokens (SconeOcc.py:124-126) = 4 registers x the 4 lane groups ----
    {
        Pk x[4];
#pragma unroll
        for (int hgrp = 0; hgrp < 2; ++hgrp) {
            const Grp G = cur();
            const FT f = load_T<4>(G, 0, 0);
            const float isc = G.h[128];
            f32x4 a[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float b = G.h[16 * k + li];
                a[k] = f32x4{b, b, b, b};
            }
            if (hgrp == 0) layer_norm(xres, x);
            tiles4<false>(a[0], a[1], a[2], a[3], G, x, f);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float mx = fmaxf(fmaxf(a[k][0], a[k][1]), fmaxf(a[k][2], a[k][3]));
                float sm = (a[k][0] + a[k][1]) + (a[k][2] + a[k][3]);
                mx = rows_allreduce(mx, [](float u, float v) { return fmaxf(u, v); });
                sm = rows_allreduce(sm, [](float u, float v) { return u + v; });
                if (lane < 16 && q < S) {
                    feat[q * ld_feat + 64 * hgrp + 16 * k + lane] = mx * isc;        // isc > 0: max commutes with the scaling
                    feat[q * ld_feat + 128 + 64 * hgrp + 16 * k + lane] = sm * isc * (1.0f / 16.f);
                }
            }
            if (hgrp == 0) sync_next();
        }
    }
}

}  // namespace v8

#ifdef L8_TRACE
extern "C" int mcr_dev_read_trace8(long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(v8::l8_trace_buf), sizeof(long long) * 160); }
#endif

void launch_local_pct8(hipStream_t s, const float* offs, float* feat, int64_t ld_feat, int64_t S, const float* blob) {
    if (S <= 0) return;
    hipLaunchKernelGGL(v8::local_pct8_kernel, dim3((unsigned)cdiv(S, v8::NW)), dim3(64 * v8::NW), 0, s, offs, feat, (long long)ld_feat,
                       (long long)S, blob);
}

int local_pct8_blob_floats() { return v8::L8_BLOB_FLOATS; }

}  // namespace mcr
