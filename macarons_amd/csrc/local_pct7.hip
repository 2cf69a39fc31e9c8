// K5-local v7 (OPT-IN, never the default) -- the fused per-query local PCTransformer of SconeOcc on ONE fp16 plane per matrix
// operand: one MFMA per product, fp32 accumulation.  This is BASELINE.json's config 3 as it is named ("bf16": a 16-bit matrix
// path); fp16 rather than bf16 because its 11 significant bits cost 5.6e-4 on the occupancies where bf16's 8 cost ~4e-3
// (NOTES, round 5: pricing table).  Tolerance of the variant: stated and measured in tests/test_variant7_gpu.py (NOT the 1e-4
// of variants 1 / 5 / 6).
// Reference mapping (SconeOcc.py:104-130: Embedding -> 2 x Encoder -> LayerNorm -> linear0 -> max || avg pool) in local_pct.hip.
//
// What stays fp32: the accumulators, the residual stream, LayerNorm statistics, the attention SCORES and the soft-max over them, GELU,
// the pooling.  What is fp16: the operands of every matrix product -- the twelve 128-wide products per encoder pair, the embeddings,
// linear0, and the 16-token attention's q | k | v and soft-max weights (v_mfma_f32_16x16x16_f16; variant 6 keeps them fp32 on
// v_mfma_f32_16x16x4_f32: 80 long fp32 MFMAs and 112 LDS dword reads per workgroup-wave against 24 short ones and 32 eight-byte reads
// here).  Weights: fp16(W); activations rounded once where they are produced (v_cvt_pk_f16_f32).
// Range: |activation| < 65504 as on variant 6 (a non-finite occupancy raises the same range flag).
//
// Structure = local_pct6.hip (4 waves = 64 tokens x 128 channels on chip, products transposed, residual stream in registers,
// Chan-combined LayerNorm statistics, GELU epilogues beside the next product's MFMAs) minus everything the low planes needed:
// a third of the MFMAs, half the fragment reads, half the LDS plane stores, half the weight bytes from L2, no split arithmetic.
// No per-matrix power-of-two weight scale either (variant 6 needs it for its LOW plane): weights and biases as they are, rounded to
// fp16 -- what a 16-bit matrix path means everywhere else; |w| >= 65520 becomes inf and ends in the range guard.
// The kernel is bound by VECTOR issue once two thirds of the matrix work are gone (first version: 6.5 k vector instructions per
// wave, vector unit 92 % busy at three waves per SIMD, matrix pipe 33 %), so this version also sheds vector work that variant 6
// hides under its MFMAs: plane rows are PADDED (272 bytes: 17 chunks) instead of XOR-swizzled -- every fragment address of a
// product is one base register plus an immediate (the swizzle cost five integer instructions per k-step) --, LayerNorm
// normalises with one v_rsq and one fma per value, and no epilogue multiplies by a scale.
// LDS = 51 KB (three workgroups per CU):
//               P  17 KB  fp16 plane [64 rows][17 chunks of 8, the last one padding], or q | k as fp16 in halves 0..63 of the rows
//               H  32 KB  fp16 plane (FF hidden half / GELU(emb1) / v) in its first 17 KB, or the final fp32 tile [64][128]
//               St  2 KB  LayerNorm partials [4 waves][64 tokens] (mean, M2)
// Row stride 272 bytes = 68 banks: the 16 rows a ds_read_b128 / ds_read_b32 group touches land 4 banks apart (conflict-free);
// the fp32 [64][128] view keeps variant 6's XOR swizzle by (row & 15).
// MCR_HIPCC_FLAGS: -fno-slp-vectorize
#include "lp_split.h"

namespace mcr {
namespace v7 {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// The swizzled addresses are loop-invariant functions of the lane; left alone, LLVM hoists all of them out of the encoder
// loop and spills them.  Re-deriving them from an opaque copy of the lane id per phase is cheaper.
__device__ __forceinline__ int opaque(int v) {
    asm volatile("" : "+v"(v));
    return v;
}

#ifndef L7_PF
#define L7_PF 3
#endif
constexpr int PF = L7_PF;                          // k16-steps of weights in flight per wave and tile
// erf-GELU (Attention.py:232, nn.GELU()) for values that are rounded to fp16 right after: erf by Abramowitz-Stegun 7.1.25 (three
// terms, |error| <= 2.5e-5 -- a tenth of the fp16 rounding unit of the result; measured on the GELU itself: 2.6e-5 absolute, 1.1e-5
// of |x|) with the constants arranged so that nothing is multiplied twice: w = |x| sqrt(log2(e) / 2), exp(-x^2 / 2) = exp2(-w w),
// 1 + p |x| / sqrt(2) = fma(w, p / sqrt(log2 e), 1).  Nine vector instructions + rcp + exp2 (l3_gelu: thirteen + two).
__device__ __forceinline__ float l7_gelu(float x) {
    const float w = fabsf(x) * 0.84932180028801904f;
    const float t = __builtin_amdgcn_rcpf(fmaf(w, 0.39169198f, 1.0f));
    float q = fmaf(0.7478556f, t, -0.0958798f);
    q = fmaf(q, t, 0.3480242f);
    q *= t;
    const float e = __builtin_amdgcn_exp2f(-(w * w));
    const float erfa = fmaf(-q, e, 1.0f);
    const float hx = 0.5f * x;
    return fmaf(fabsf(hx), erfa, hx);
}

constexpr int PR = 17;                             // 16-byte chunks per plane row (16 + 1 padding)
struct WRing { uint4 b[PF]; };                     // [slot]

__device__ __forceinline__ const uint4* wptr(const float* Wp, int nt, int S, int lane) {
    return reinterpret_cast<const uint4*>(Wp) + (size_t)nt * S * 64 + lane;
}
// Request the first PF k-steps of a weight tile.  The sched_barrier pins the requests HERE (in front of the epilogue /
// LayerNorm / barrier that precedes the product): left alone, LLVM sinks them down to their first use and every product
// starts with an exposed L2 round trip.
template <int S>
__device__ __forceinline__ void wload(WRing& r, const uint4* bp) {
#pragma unroll
    for (int p = 0; p < (S < PF ? S : PF); ++p) {
        r.b[p] = bp[p * 64];
    }
    __builtin_amdgcn_sched_barrier(0);
}

// Pin accumulator chains to this point of the program: the MFMA builtins are pure, so instruction selection is free to delay a
// whole chain to its next use (it deferred one m-tile's 24 MFMAs of a product past the following product and two barriers,
// spilling the fragments it had loaded for them).  An empty asm that "rewrites" the accumulators orders them like a fence.
__device__ __forceinline__ void pin(f32x16& a, f32x16& b) { asm volatile("" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void pin(f32x16& a, f32x16& b, f32x16& c) { asm volatile("" : "+v"(a), "+v"(b), "+v"(c)); }

__device__ __forceinline__ void zero(f32x16& a) {
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = 0.f;
}

// acc[mt] (+)= W(n-tile) . act^T for both 32-token m-tiles; act = fp16 plane A [64][16] in LDS; the ring holds the
// first PF k-steps of the weight tile (wload) and is refilled here.
template <int S>
__device__ __forceinline__ void gemm(f32x16 (&acc)[2], const uint4* __restrict__ A, WRing& r, const uint4* __restrict__ bp,
                                     int lane_) {
    const int lane = opaque(lane_);
    const int i = lane & 31, h = lane >> 5;
    uint4 an[2];
    auto fetch = [&](int s) {
        const int c = 2 * s + h;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) an[mt] = A[(mt * 32 + i) * PR + c];
    };
    fetch(0);
    // software pipeline in program order, fenced: [next step's 2 fragment reads][the weight request PF steps ahead] |
    // [this step's 2 MFMAs].
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const uint4 a0 = an[0], a1 = an[1];
        if (s + 1 < S) fetch(s + 1);
        const uint4 w = r.b[s % PF];
        if (s + PF < S) r.b[s % PF] = bp[(s + PF) * 64];
        __builtin_amdgcn_sched_barrier(0);
        acc[0] = mfma_h(w, a0, acc[0]);
        acc[1] = mfma_h(w, a1, acc[1]);
        pin(acc[0], acc[1]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// The same product with a slice of ANOTHER accumulator's epilogue beside every k-step's two MFMAs (fn(s), s = 0 .. S-1): vector
// work of a finished product issued in the shadow of the next product's matrix work by the same wave.
template <int S, class Fn>
__device__ __forceinline__ void gemm_with(f32x16 (&acc)[2], const uint4* __restrict__ A, WRing& r, const uint4* __restrict__ bp,
                                          int lane_, Fn fn) {
    const int lane = opaque(lane_);
    const int i = lane & 31, h = lane >> 5;
    uint4 an[2];
    auto fetch = [&](int s) {
        const int c = 2 * s + h;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) an[mt] = A[(mt * 32 + i) * PR + c];
    };
    fetch(0);
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const uint4 a0 = an[0], a1 = an[1];
        if (s + 1 < S) fetch(s + 1);
        const uint4 w = r.b[s % PF];
        if (s + PF < S) r.b[s % PF] = bp[(s + PF) * 64];
        __builtin_amdgcn_sched_barrier(0);
        acc[0] = mfma_h(w, a0, acc[0]);
        acc[1] = mfma_h(w, a1, acc[1]);
        fn(s);
        pin(acc[0], acc[1]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// QKV product (N = 192 = 6 n-tiles over 4 waves), balanced: every wave takes its own n-tile w for both m-tiles plus HALF of
// n-tile 4 + (w >> 1): the m-tile w & 1.
__device__ __forceinline__ void gemm_qkv(f32x16 (&acc)[2], f32x16& acch, const uint4* __restrict__ A, WRing& r0, WRing& r1,
                                         const uint4* __restrict__ bp0, const uint4* __restrict__ bp1, int wave, int lane_) {
    constexpr int S = 8;
    const int lane = opaque(lane_);
    const int i = lane & 31, h = lane >> 5, mh = wave & 1;
    uint4 an[2];
    auto fetch = [&](int s) {
        const int c = 2 * s + h;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) an[mt] = A[(mt * 32 + i) * PR + c];
    };
    fetch(0);
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const uint4 a0 = an[0], a1 = an[1];
        const uint4 ah = mh ? a1 : a0;
        if (s + 1 < S) fetch(s + 1);
        const uint4 w = r0.b[s % PF], v = r1.b[s % PF];
        if (s + PF < S) {
            r0.b[s % PF] = bp0[(s + PF) * 64];
            r1.b[s % PF] = bp1[(s + PF) * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
        acc[0] = mfma_h(w, a0, acc[0]);
        acc[1] = mfma_h(w, a1, acc[1]);
        acch = mfma_h(v, ah, acch);
        pin(acc[0], acc[1], acch);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ---- epilogue helpers: a C fragment t (n-tile nt, m-tile mt) holds, in lane (j, h), the features 32 nt + 8 g + 4 h + e
// (register r = 4 g + e) of token 32 mt + j ----------------------------------------------------------------------------------
// The accumulator of a product starts from its bias: the four 16-byte loads land directly in the accumulator registers.  Products
// whose result is added to the residual stream (out projection, FF2) accumulate IN PLACE on x + b.
__device__ __forceinline__ void add_bias(f32x16& x, const float* __restrict__ vec, int nt, int lane_) {
    const float4* p = reinterpret_cast<const float4*>(vec + 32 * nt + 4 * (opaque(lane_) >> 5));
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 b = p[2 * g];
        x[4 * g] += b.x; x[4 * g + 1] += b.y; x[4 * g + 2] += b.z; x[4 * g + 3] += b.w;
    }
}
__device__ __forceinline__ void init_bias(f32x16& a, const float* __restrict__ vec, int nt, int lane_) {
    const float4* p = reinterpret_cast<const float4*>(vec + 32 * nt + 4 * (opaque(lane_) >> 5));
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 b = p[2 * g];
        a[4 * g] = b.x; a[4 * g + 1] = b.y; a[4 * g + 2] = b.z; a[4 * g + 3] = b.w;
    }
}

// store 4 consecutive features of one token as fp16: two v_cvt_pk_f16_f32, one ds_write_b64
__device__ __forceinline__ void put4(uint2* __restrict__ plane, int idx, float v0, float v1, float v2, float v3) {
    plane[idx] = make_uint2(pack2h(v0, v1), pack2h(v2, v3));
}
// the plane of one 32-feature column block (n-tile nt) for m-tile mt from t through f(value, g, e); uint2 index = row * 34 + 2 chunk + h
template <class Fn>
__device__ __forceinline__ void put_planes(uint4* __restrict__ buf, int nt, int mt, const f32x16& t, int lane_, Fn f) {
    const int lane = opaque(lane_);
    const int j = lane & 31, h = lane >> 5;
    uint2* b2 = reinterpret_cast<uint2*>(buf) + (mt * 32 + j) * (2 * PR) + h + 8 * nt;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        put4(b2, 2 * g, f(t[4 * g], g, 0), f(t[4 * g + 1], g, 1), f(t[4 * g + 2], g, 2), f(t[4 * g + 3], g, 3));
        __builtin_amdgcn_sched_barrier(0);         // one quad at a time: interleaving all 32 GELUs of an epilogue spills
    }
}
// one quad (g) of put_planes: `row` = (lane & 31) * 34 + (lane >> 5) is computed once per product by the caller
template <class Fn>
__device__ __forceinline__ void put_quad(uint4* __restrict__ buf, int nt, int mt, int g, const f32x16& t, int row, Fn f) {
    uint2* b2 = reinterpret_cast<uint2*>(buf);
    put4(b2, mt * (32 * 2 * PR) + row + 2 * (4 * nt + g), f(t[4 * g]), f(t[4 * g + 1]), f(t[4 * g + 2]), f(t[4 * g + 3]));
}
// fp32 tile with row stride LD floats: features (4 floats = one chunk) chunk0 + 2 g + h of token 32 mt + j, ds_write_b128;
// kmask = 15: chunks XOR-swizzled by the row (the [64][128] view), 0: plain rows (the padded q | k view)
template <class Fn>
__device__ __forceinline__ void put_f32(float* __restrict__ buf, int LD, int kmask, int chunk0, int mt, const f32x16& t, int lane_, Fn f) {
    const int lane = opaque(lane_);
    const int j = lane & 31, h = lane >> 5, key = j & kmask;
    float* row = buf + (mt * 32 + j) * LD;
#pragma unroll
    for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(row + (((chunk0 + 2 * g + h) ^ key) << 2)) =
            make_float4(f(t[4 * g], g, 0), f(t[4 * g + 1], g, 1), f(t[4 * g + 2], g, 2), f(t[4 * g + 3], g, 3));
}

// lanes l and l ^ 32 (the two halves of a token's 32 features in this wave): sum on both
__device__ __forceinline__ float halves_sum(float v) {
    const unsigned c = __builtin_bit_cast(unsigned, v);
    const auto q = __builtin_amdgcn_permlane32_swap(c, c, false, false);
    return __builtin_bit_cast(float, (unsigned)q[0]) + __builtin_bit_cast(float, (unsigned)q[1]);
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

// LayerNorm, part 1: this wave's (mean, M2) over its 32 features of every token -> St[wave][token].  One pass (sum and sum of
// squares in fp32, M2 = sum x^2 - 32 mean^2: exact enough for an fp16 result unless |mean| > ~30 standard deviations inside a
// 32-feature slice); variant 6 subtracts the mean first (32 more instructions per token and wave).
__device__ __forceinline__ void ln_partial(const f32x16 (&x)[2], float2* __restrict__ St, int wave, int lane_) {
    const int lane = opaque(lane_);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#ifdef L7_LN_TWO_PASS
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 16; r += 4) s += (x[mt][r] + x[mt][r + 1]) + (x[mt][r + 2] + x[mt][r + 3]);
        const float m = halves_sum(s) * (1.0f / 32.f);
        float q = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float d = x[mt][r] - m;
            q = fmaf(d, d, q);
        }
        q = halves_sum(q);
        St[wave * 64 + mt * 32 + (lane & 31)] = make_float2(m, q);
#else
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int r = 0; r < 16; r += 4) s += (x[mt][r] + x[mt][r + 1]) + (x[mt][r + 2] + x[mt][r + 3]);
#pragma unroll
        for (int r = 0; r < 16; ++r) q = fmaf(x[mt][r], x[mt][r], q);
        s = halves_sum(s);
        q = halves_sum(q);
        const float m = s * (1.0f / 32.f);
        St[wave * 64 + mt * 32 + (lane & 31)] = make_float2(m, fmaxf(fmaf(-s, m, q), 0.f));   // both lane halves hold (and store) the same pair: no branch
#endif
    }
}
// part 2 (after a barrier): combine the 4 partials (Chan), normalise (eps 1e-5; gamma / beta are folded into the next
// weights), split, store this wave's 32 features of every token as planes
__device__ __forceinline__ void ln_finish(const f32x16 (&x)[2], const float2* __restrict__ St, uint4* __restrict__ P, int wave,
                                          int lane_) {
    const int j = opaque(lane_) & 31;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        float2 p[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) p[w] = St[w * 64 + mt * 32 + j];
        const float mu = ((p[0].x + p[1].x) + (p[2].x + p[3].x)) * 0.25f;
        float m2 = (p[0].y + p[1].y) + (p[2].y + p[3].y), dd = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float d = p[w].x - mu;
            dd = fmaf(d, d, dd);
        }
        m2 = fmaf(32.f, dd, m2);
        const float rstd = __builtin_amdgcn_rsqf(fmaf(m2, 1.0f / 128.f, 1e-5f));
        const float shift = -mu * rstd;
        put_planes(P, wave, mt, x[mt], lane_, [&](float v, int, int) { return fmaf(v, rstd, shift); });
    }
}

#ifdef L7_TRACE             // dev only: per-phase timestamps of one workgroup (tools/build_variant.py ... -DL7_TRACE=<block>)
__device__ long long l7_trace_buf[64];
#define L7_T() do { if (blockIdx.x == (L7_TRACE) && tid == 0) { l7_trace_buf[tp] = clock64(); l7_trace_buf[tp ? 63 : 62] = wall_clock64(); } ++tp; } while (0)
#else
#define L7_T() do { } while (0)
#endif

// grid = ceil(S / 4); S sequences of 16 offsets [S,16,3]; features[s*ld_feat + 0:256] = max(128) || avg(128)
// feat_h != NULL: the pooled features go out as ONE fp16 plane (row stride ld_feat halves) instead of fp32: the head GEMM
// (linear3p.hip, single-plane form) reads it as it is
#ifndef L7_WGS
#define L7_WGS 3
#endif
__global__ __launch_bounds__(256, L7_WGS) void local_pct7_kernel(const float* __restrict__ offs, float* __restrict__ feat,
                                                                long long ld_feat, long long S,
                                                                const float* __restrict__ blob, _Float16* __restrict__ feat_h) {
    __shared__ __attribute__((aligned(16))) uint4 P[64 * PR];
    __shared__ __attribute__((aligned(16))) uint4 H[2 * 64 * 16];
    __shared__ __attribute__((aligned(16))) float2 St[4 * 64];
    __shared__ __attribute__((aligned(16))) uint4 Z[4];            // 64 bytes of zeros: the k = 8..15 half of the Q fragments (attention)
    float* F = reinterpret_cast<float*>(H);                // fp32 [64][128] view: v / the final tile
    const int tid = threadIdx.x, lane0 = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);        // provably uniform: weight / bias addresses = SGPR base + lane offset
    if (tid < 4) Z[tid] = make_uint4(0, 0, 0, 0);                     // (visible to every wave behind the first barrier)
    const float* mats = blob;
    const float* vecs = blob + L7_MATS_TOTAL;
    // One workgroup per 4-query group, NOT persistent: persistent workgroups (2 per CU, looping over groups with the next group's
    // offsets prefetched) measured 5 % slower -- the two co-resident workgroups then run in lock-step and collide in the same
    // phase (both on the matrix pipe, then both on the vector ALU); fresh workgroups start staggered and overlap better.
    const long long s0 = (long long)blockIdx.x * L3_QPB;
    const int lane = lane0;
    int tp = 0; (void)tp;
    L7_T();
    WRing ring;
    f32x16 acc[2], xres[2];
    // ---- Embedding (Attention.py:98-128): linear1 3->125 (K padded to 16), GELU -> H planes ; linear2 125->125 || xyz ----
    wload<1>(ring, wptr(mats + l7_mat_off(0), wave, 1, lane));
    init_bias(acc[0], vecs + L3_VEC_EMB1, wave, lane);
    init_bias(acc[1], vecs + L3_VEC_EMB1, wave, lane);
    float xyz[2][3];
    {
        const int j = lane & 31;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int t = mt * 32 + j;
            const long long seq = s0 + (t >> 4);
            xyz[mt][0] = xyz[mt][1] = xyz[mt][2] = 0.f;
            if (seq < S) {
                const float* p = offs + (seq * 16 + (t & 15)) * 3;
                xyz[mt][0] = p[0]; xyz[mt][1] = p[1]; xyz[mt][2] = p[2];
            }
        }
        // the activation fragment of the K = 16 step straight from registers: k = 0..2 = xyz in the lower lane half, zeros elsewhere
        const bool lowh = lane < 32;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            uint4 a = make_uint4(0, 0, 0, 0);
            a.x = pack2h(lowh ? xyz[mt][0] : 0.f, lowh ? xyz[mt][1] : 0.f);
            a.y = pack2h(lowh ? xyz[mt][2] : 0.f, 0.f);
            acc[mt] = mfma_h(ring.b[0], a, acc[mt]);
        }
    }
    L7_T();                                        // emb1 product done
    wload<8>(ring, wptr(mats + l7_mat_off(1), wave, 8, lane));
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
        put_planes(H, wave, mt, acc[mt], lane, [&](float v, int, int) { return l7_gelu(v); });
    init_bias(xres[0], vecs + L3_VEC_EMB2, wave, lane);
    init_bias(xres[1], vecs + L3_VEC_EMB2, wave, lane);
    __syncthreads();
    L7_T();                                        // emb1 gelu + barrier
    gemm<8>(xres, H, ring, wptr(mats + l7_mat_off(1), wave, 8, lane), lane);
    L7_T();                                        // emb2 gemm
    {
        const bool cat = wave == 3 && lane >= 32;  // features 125..127 = the raw xyz (Attention.py:123-126): n-tile 3, g = 3, h = 1, e = 1..3
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            xres[mt][13] = cat ? xyz[mt][0] : xres[mt][13];
            xres[mt][14] = cat ? xyz[mt][1] : xres[mt][14];
            xres[mt][15] = cat ? xyz[mt][2] : xres[mt][15];
        }
    }

#pragma unroll 1
    for (int e = 0; e < 2; ++e) {
        const float* em = mats + l7_mat_off(2) + e * (L7_MAT_QKV + 5 * L7_MAT_128);
        const float* ev = vecs + L3_VEC_ENC0 + e * L3_VEC_ENC_STRIDE;
        const float* w_out = em + L7_MAT_QKV;
        const float* w_ff1a = w_out + L7_MAT_128;
        const float* w_ff1b = w_out + L7_MAT_128 * 2;
        const float* w_ff2a = w_out + L7_MAT_128 * 3;
        const float* w_ff2b = w_out + L7_MAT_128 * 4;
        // ---- norm1 (folded) -> planes P ; QKV (Attention.py:186-188, 287): q | k -> P rows (fp16), v -> the H plane ----
        WRing ring1;
        const uint4* bq0 = wptr(em, wave, 8, lane);
        const uint4* bq1 = wptr(em, 4 + (wave >> 1), 8, lane);
        wload<8>(ring, bq0);
        wload<8>(ring1, bq1);
        L7_T();                                    // (previous epilogue)
        ln_partial(xres, St, wave, lane);
        __syncthreads();                           // St visible; every wave is past its last read of P and H
        L7_T();                                    // norm1 partial + barrier
        ln_finish(xres, St, P, wave, lane);
        __syncthreads();                           // x^ planes visible
        L7_T();                                    // norm1 finish + barrier
        {
            f32x16 aq[2], ah;
            const int nth = 4 + (wave >> 1);
            init_bias(aq[0], ev, wave, lane);
            init_bias(aq[1], ev, wave, lane);
            init_bias(ah, ev, nth, lane);
            gemm_qkv(aq, ah, P, ring, ring1, bq0, bq1, wave, lane);
            L7_T();                                // qkv gemm
            wload<8>(ring, wptr(w_out, wave, 8, lane));
            __syncthreads();                       // x^ planes consumed: P may take q|k
            L7_T();                                // barrier
            // n-tiles 0, 1 = q, k -> halves 0..31 / 32..63 of the P rows; n-tiles 2..5 = v -> the H plane (halves 32 (nt - 2) ..): one
            // rounding to fp16 where the values are produced, like every other matrix operand of this variant
            auto fb = [&](float v, int, int) { return v; };
            uint4* dst = wave < 2 ? P : H;          // branch-free: the whole encoder stays one basic block
            const int ntd = wave < 2 ? wave : wave - 2;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) put_planes(dst, ntd, mt, aq[mt], lane, fb);
            put_planes(H, nth - 2, wave & 1, ah, lane, fb);
        }
        __syncthreads();
        L7_T();                                    // qkv put + barrier
        // ---- attention (Attention.py:8-36) on v_mfma_f32_16x16x16_f16 (fp16 q | k | v and soft-max weights, fp32 scores / soft-max /
        // accumulation): wave = query (16 tokens); per head  S^T = K Q^T (A = K: lane (key li, g) holds dims 4 (g & 1) .., B = Q^T: lane
        // (query li, g) holds dims 4 g .. for g < 2 and ZEROS for g >= 2 -- the head has 8 dims, the MFMA's k is 16), so lane (qi, g) owns
        // S[qi][4g..4g+3]; soft-max over the keys = 4 registers x the 4 lane groups; O^T = V^T P^T: A = V^T through the transposing LDS
        // read (lane (li, g) addresses token 4 g + (li >> 2), columns 4 (li & 3) ..; lane c receives tokens 4 g .. 4 g + 3 of column c),
        // B = P^T = the soft-max registers rounded to fp16: lane (qi, g) ends up with the features 4g..4g+3 of every 16-feature block of
        // its token -> the fp16 plane, written over the wave's own q | k rows.
        {
            typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
            typedef short s16x4 __attribute__((ext_vector_type(4)));
            typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
            const int ln = opaque(lane);
            const int r0 = wave * 16, li = ln & 15, g = ln >> 4;
            const char* Pb = reinterpret_cast<const char*>(P);
            const char* ka = Pb + (r0 + li) * (16 * PR) + 64 + (g & 1) * 8;                     // k[token r0 + li][8 hh + 4 (g & 1) ..]: + 16 hh
            const char* qa = g < 2 ? Pb + (r0 + li) * (16 * PR) + g * 8 : reinterpret_cast<const char*>(Z);   // q[...][8 hh + 4 g ..] or zeros
            const char* va = reinterpret_cast<const char*>(H) + (r0 + 4 * g + (li >> 2)) * (16 * PR) + (li & 3) * 8;   // v[token][16 blk + 4 (li & 3) ..]: + 32 blk
            h16x4 pf[4];
#pragma unroll
            for (int hh = 0; hh < 4; ++hh) {
                const h16x4 kf = *reinterpret_cast<const h16x4*>(ka + 16 * hh);
                const h16x4 qf = *reinterpret_cast<const h16x4*>(qa + 16 * hh);
                f32x4 st = __builtin_amdgcn_mfma_f32_16x16x16f16(kf, qf, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                float mx = fmaxf(fmaxf(st[0], st[1]), fmaxf(st[2], st[3]));
                mx = rows_allreduce(mx, [](float a, float b) { return fmaxf(a, b); });
                mx *= 0.35355339059327376220f;                                           // scores / sqrt(8)
                float den = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    st[r] = __expf(fmaf(st[r], 0.35355339059327376220f, -mx));
                    den += st[r];
                }
                den = rows_allreduce(den, [](float a, float b) { return a + b; });
                const float inv = __builtin_amdgcn_rcpf(den);
                const uint2 pk = make_uint2(pack2h(st[0] * inv, st[1] * inv), pack2h(st[2] * inv, st[3] * inv));
                pf[hh] = __builtin_bit_cast(h16x4, pk);
            }
            f32x4 o[4][2];
#pragma unroll
            for (int hh = 0; hh < 4; ++hh)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const s16x4 vt = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(va + 64 * hh + 32 * nt));
                    o[hh][nt] = __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(h16x4, vt), pf[hh], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                }
            // every q | k read of this wave's 16 rows precedes the plane writes over them (other waves own other rows)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            uint2* p2 = reinterpret_cast<uint2*>(P);
            const int base = (r0 + li) * (2 * PR) + g;        // chunk 4 hh + 2 nt + (g >> 1), half g & 1  ->  + 2 (4 hh + 2 nt)
#pragma unroll
            for (int hh = 0; hh < 4; ++hh)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    put4(p2, base + 2 * (4 * hh + 2 * nt), o[hh][nt][0], o[hh][nt][1], o[hh][nt][2], o[hh][nt][3]);
        }
        // ---- out projection + residual (Attention.py:201-202, 290): x += att W_o^T + b, accumulated in place ----
        L7_T();                                    // attention
        add_bias(xres[0], ev + 192, wave, lane);
        add_bias(xres[1], ev + 192, wave, lane);
        __syncthreads();
        L7_T();                                    // bias + barrier
        gemm<8>(xres, P, ring, wptr(w_out, wave, 8, lane), lane);
        L7_T();                                    // out gemm
        wload<8>(ring, wptr(w_ff1a, wave, 8, lane));
        // ---- norm2 (folded) -> planes P ; FF 128 -> 256 (GELU) -> 128 + residual (Attention.py:293-298), two halves ----
        ln_partial(xres, St, wave, lane);
        init_bias(acc[0], ev + 192 + 128, wave, lane);
        init_bias(acc[1], ev + 192 + 128, wave, lane);
        __syncthreads();                           // St visible; the attention output in P is consumed by every wave
        L7_T();                                    // res + norm2 partial + barrier
        ln_finish(xres, St, P, wave, lane);
        __syncthreads();
        L7_T();                                    // norm2 finish + barrier
        // FF1a, then FF1b with GELU(a) beside its MFMAs (hidden half a -> H), barrier, FF2a with GELU(b) beside its MFMAs
        // (hidden half b -> P, free once every wave is past FF1b), barrier, FF2b: two barriers instead of three and both GELU
        // epilogues in the shadow of matrix work
        gemm<8>(acc, P, ring, wptr(w_ff1a, wave, 8, lane), lane);
        L7_T();                                    // ff1a gemm
        wload<8>(ring, wptr(w_ff1b, wave, 8, lane));
        f32x16 accb[2];
        init_bias(accb[0], ev + 192 + 128 + 128, wave, lane);
        init_bias(accb[1], ev + 192 + 128 + 128, wave, lane);
        const int pq_lane = opaque(lane), pq_row = (pq_lane & 31) * (2 * PR) + (pq_lane >> 5);
        gemm_with<8>(accb, P, ring, wptr(w_ff1b, wave, 8, lane), lane, [&](int s_) {
            put_quad(H, wave, s_ >> 2, s_ & 3, acc[s_ >> 2], pq_row, [&](float v) { return l7_gelu(v); });
        });
        L7_T();                                    // ff1b gemm + gelu a
        wload<8>(ring, wptr(w_ff2a, wave, 8, lane));
        add_bias(xres[0], ev + 192 + 128 + 256, wave, lane);   // FF2 accumulates onto the residual in place
        add_bias(xres[1], ev + 192 + 128 + 256, wave, lane);
        __syncthreads();                           // hidden half a visible; x^ planes in P consumed by every wave
        L7_T();                                    // barrier
        gemm_with<8>(xres, H, ring, wptr(w_ff2a, wave, 8, lane), lane, [&](int s_) {
            put_quad(P, wave, s_ >> 2, s_ & 3, accb[s_ >> 2], pq_row, [&](float v) { return l7_gelu(v); });
        });
        L7_T();                                    // ff2a gemm + gelu b
        wload<8>(ring, wptr(w_ff2b, wave, 8, lane));
        __syncthreads();                           // hidden half b visible
        L7_T();                                    // barrier
        L7_T();                                    // (spare stamp: keeps the trace table of tools/trace_local_pct.py aligned)
        gemm<8>(xres, P, ring, wptr(w_ff2b, wave, 8, lane), lane);
        L7_T();                                    // ff2b gemm
    }
    L7_T();
    // ---- final norm (folded) + linear0 128 -> 128 (SconeOcc.py:119-122) ----
    wload<8>(ring, wptr(mats + l7_mat_off(14), wave, 8, lane));
    ln_partial(xres, St, wave, lane);
    init_bias(acc[0], vecs + L3_VEC_LIN0, wave, lane);
    init_bias(acc[1], vecs + L3_VEC_LIN0, wave, lane);
    __syncthreads();
    ln_finish(xres, St, P, wave, lane);
    __syncthreads();
    gemm<8>(acc, P, ring, wptr(mats + l7_mat_off(14), wave, 8, lane), lane);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
        put_f32(F, 128, 15, 8 * wave, mt, acc[mt], lane, [&](float v, int, int) { return v; });
    __syncthreads();
    L7_T();                                        // final norm + lin0
    // ---- max || avg pool over the 16 tokens of each query (SconeOcc.py:124-126) ----
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int o = tid + r * 256, q = o >> 7, c = o & 127;
        if (s0 + q < S) {
            float mx = -__builtin_inff(), sm = 0.f;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int row = q * 16 + j;
                const float v = F[row * 128 + ((((c >> 2) ^ (row & 15)) << 2) | (c & 3))];
                mx = fmaxf(mx, v);
                sm += v;
            }
            const float av = sm * (1.0f / 16.f);
            if (feat_h) {
                const unsigned hv = pack2h(mx, av);
                const long long o2 = (s0 + q) * ld_feat + c;
                feat_h[o2] = __builtin_bit_cast(_Float16, (unsigned short)(hv & 0xffffu));
                feat_h[o2 + 128] = __builtin_bit_cast(_Float16, (unsigned short)(hv >> 16));
            } else {
                feat[(s0 + q) * ld_feat + c] = mx;
                feat[(s0 + q) * ld_feat + 128 + c] = av;
            }
        }
    }
    L7_T();
}

}  // namespace v7

#ifdef L7_TRACE
extern "C" int mcr_dev_read_trace(long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(v7::l7_trace_buf), sizeof(long long) * 64); }
#endif

void launch_local_pct7(hipStream_t s, const float* offs, float* feat, int64_t ld_feat, int64_t S, const float* blob, void* feat_h) {
    if (S <= 0) return;
    hipLaunchKernelGGL(v7::local_pct7_kernel, dim3((unsigned)cdiv(S, L3_QPB)), dim3(256), 0, s, offs, feat, (long long)ld_feat,
                       (long long)S, blob, (_Float16*)feat_h);
}

int local_pct7_blob_floats() { return L7_BLOB_FLOATS; }

}  // namespace mcr
