// Shared pieces of the split-precision ("bf16 x 6") fused local transformers (local_pct3.hip, local_pct4.hip):
// the parameter-blob layout, the exact three-way fp32 -> bf16 split, the block GEMM with its weight ring, the
// C-fragment iterator and the exact-erf GELU.  See local_pct3.hip for the numerics and local_pct.hip for the
// reference mapping (SconeOcc.py:104-130).
#pragma once
#include "nn_kernels.h"

namespace mcr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int L3_T = 64, L3_QPB = 4, L3_XLD = 132, L3_SLD = 196;
// blob: matrices as [n-tile][k16 step][plane hi,mid,lo][lane][8 bf16]  (1.5 floats per weight), then the v1 vectors
constexpr int L3_MAT_K16 = 128 * 16 * 3 / 2, L3_MAT_128 = 128 * 128 * 3 / 2, L3_MAT_QKV = 192 * 128 * 3 / 2;
__host__ __device__ constexpr int l3_mat_off(int idx) {
    int off = 0;
    for (int i = 0; i < idx; ++i) {
        const bool is_qkv = (i >= 2 && i < 14 && ((i - 2) % 6) == 0);
        off += i == 0 ? L3_MAT_K16 : (is_qkv ? L3_MAT_QKV : L3_MAT_128);
    }
    return off;
}
constexpr int L3_MATS_TOTAL = l3_mat_off(15);
constexpr int L3_VEC_EMB1 = 0, L3_VEC_EMB2 = 128, L3_VEC_ENC0 = 256, L3_VEC_ENC_STRIDE = 192 + 128 + 256 + 128,
              L3_VEC_LIN0 = L3_VEC_ENC0 + 2 * L3_VEC_ENC_STRIDE, L3_VECS_TOTAL = L3_VEC_LIN0 + 128;
constexpr int L3_BLOB_FLOATS = L3_MATS_TOTAL + L3_VECS_TOTAL;

__device__ __forceinline__ float l3_gelu(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * z * z);
    const float erfa = fmaf(-p * t, e, 1.0f);
    return 0.5f * x + 0.5f * fabsf(x) * erfa;
}

// exact three-way split of 8 consecutive fp32 values into packed bf16x8 planes
struct Split3 { uint4 hi, mid, lo; };
__device__ __forceinline__ unsigned pack_top(float a, float b) {           // {top16(b), top16(a)}: element 0 in the low half
    return __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, b), __builtin_bit_cast(unsigned, a), 0x07060302u);
}
#ifdef L3_SPLIT_PK          // experiment: residuals with v_pk_add_f32 (two elements per instruction)
typedef float l3_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned l3_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ Split3 split8(const float4 p, const float4 q) {
    const l3_f32x2 x[4] = {{p.x, p.y}, {p.z, p.w}, {q.x, q.y}, {q.z, q.w}};
    l3_f32x2 r[4], r2[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const l3_f32x2 hi = __builtin_bit_cast(l3_f32x2, __builtin_bit_cast(l3_u32x2, x[e]) & 0xffff0000u);
        r[e] = x[e] - hi;
        const l3_f32x2 mid = __builtin_bit_cast(l3_f32x2, __builtin_bit_cast(l3_u32x2, r[e]) & 0xffff0000u);
        r2[e] = r[e] - mid;
    }
    Split3 s;
    s.hi = make_uint4(pack_top(x[0].x, x[0].y), pack_top(x[1].x, x[1].y), pack_top(x[2].x, x[2].y), pack_top(x[3].x, x[3].y));
    s.mid = make_uint4(pack_top(r[0].x, r[0].y), pack_top(r[1].x, r[1].y), pack_top(r[2].x, r[2].y), pack_top(r[3].x, r[3].y));
    s.lo = make_uint4(pack_top(r2[0].x, r2[0].y), pack_top(r2[1].x, r2[1].y), pack_top(r2[2].x, r2[2].y), pack_top(r2[3].x, r2[3].y));
    return s;
}
#else
__device__ __forceinline__ Split3 split8(const float4 p, const float4 q) {
    const float x[8] = {p.x, p.y, p.z, p.w, q.x, q.y, q.z, q.w};
    float r[8], r2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float hi = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, x[e]) & 0xffff0000u);
        r[e] = x[e] - hi;                                                    // exact
        const float mid = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, r[e]) & 0xffff0000u);
        r2[e] = r[e] - mid;                                                  // exact, <= 8 significant bits
    }
    Split3 s;
    s.hi = make_uint4(pack_top(x[0], x[1]), pack_top(x[2], x[3]), pack_top(x[4], x[5]), pack_top(x[6], x[7]));
    s.mid = make_uint4(pack_top(r[0], r[1]), pack_top(r[2], r[3]), pack_top(r[4], r[5]), pack_top(r[6], r[7]));
    s.lo = make_uint4(pack_top(r2[0], r2[1]), pack_top(r2[2], r2[3]), pack_top(r2[4], r2[5]), pack_top(r2[6], r2[7]));
    return s;
}
#endif
__device__ __forceinline__ f32x16 mfma_bf(uint4 a, uint4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// acc[t] (+)= A[64 x 16*S] * W^T for this wave's TPW tiles: m-tile = wave & 1, n-tiles (wave >> 1) * TPW + t, so the
// A rows are split once per k16-step per wave and reused by its TPW column tiles.
// A: fp32 in LDS (row stride lda); lane (i = l&31, h = l>>5) owns A[row][16 s + 8 h .. +7] and B[k = 16 s + 8 h .. +7][n].
// The weight ring (L3_PF k16-steps in flight) is shared by consecutive products like in local_pct.hip: the first
// steps of the NEXT product's planes are requested during this product's tail.
#ifdef L3_PROBE_SAMEW
#define L3_IDX(i) ((i) % 192)
#else
#define L3_IDX(i) (i)
#endif
#ifndef L3_PF_N
#define L3_PF_N 2
#endif
constexpr int L3_PF = L3_PF_N;
typedef uint4 l3_ring_t[L3_PF][3][3];            // [step slot][tile][plane]

template <int TPW>
__device__ __forceinline__ const uint4* l3_bptr(const float* Wp, int S, int wave, int lane, int t) {
#ifdef L3_PROBE_SAMEW      // timing probe only (wrong results): every wave re-reads one 9 KB window -> L1 hits
    return reinterpret_cast<const uint4*>(Wp) + lane - (size_t)0;
#else
    return reinterpret_cast<const uint4*>(Wp) + (size_t)((wave >> 1) * TPW + t) * S * 3 * 64 + lane;
#endif
}

template <int S, int TPW, bool INIT, bool PRE, int NEXT_TPW>
__device__ __forceinline__ void l3_gemm(f32x16 (&acc)[TPW], const float* __restrict__ A, int lda,
                                        const float* __restrict__ Wp, l3_ring_t& b, const float* __restrict__ next_Wp,
                                        int wave, int lane) {
    static_assert(S == 1 || S % L3_PF == 0, "ring slots of consecutive products must line up");
    const int i = lane & 31, h = lane >> 5;
    const uint4* bp[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        bp[t] = l3_bptr<TPW>(Wp, S, wave, lane, t);
        if (INIT) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        }
    }
    const uint4* np[NEXT_TPW > 0 ? NEXT_TPW : 1];
#pragma unroll
    for (int t = 0; t < NEXT_TPW; ++t) np[t] = l3_bptr<(NEXT_TPW > 0 ? NEXT_TPW : 1)>(next_Wp, 8, wave, lane, t);
    const float* a0 = A + ((wave & 1) * 32 + i) * lda + 8 * h;
    constexpr int PF = S < L3_PF ? S : L3_PF;
    if (!PRE) {
#pragma unroll
        for (int p = 0; p < PF; ++p)
#pragma unroll
            for (int t = 0; t < TPW; ++t)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) b[p][t][pl] = bp[t][L3_IDX((p * 3 + pl) * 64)];
    }
    float4 ra[2];                                           // raw A of the next step
    ra[0] = *reinterpret_cast<const float4*>(a0); ra[1] = *reinterpret_cast<const float4*>(a0 + 4);
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const Split3 sa = split8(ra[0], ra[1]);
        if (s + 1 < S) {
            ra[0] = *reinterpret_cast<const float4*>(a0 + 16 * (s + 1)); ra[1] = *reinterpret_cast<const float4*>(a0 + 16 * (s + 1) + 4);
        }
        uint4 bc[TPW][3];
#pragma unroll
        for (int t = 0; t < TPW; ++t)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) bc[t][pl] = b[s % PF][t][pl];
        if (s + PF < S) {
#pragma unroll
            for (int t = 0; t < TPW; ++t)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) b[s % PF][t][pl] = bp[t][L3_IDX(((s + PF) * 3 + pl) * 64)];
        } else if (NEXT_TPW > 0 && S > 1) {
#pragma unroll
            for (int t = 0; t < NEXT_TPW; ++t)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) b[s % PF][t][pl] = np[t][((s + PF - S) * 3 + pl) * 64];
        }
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            acc[t] = mfma_bf(sa.lo, bc[t][0], acc[t]);      // smallest terms first
            acc[t] = mfma_bf(sa.hi, bc[t][2], acc[t]);
            acc[t] = mfma_bf(sa.mid, bc[t][1], acc[t]);
            acc[t] = mfma_bf(sa.mid, bc[t][0], acc[t]);
            acc[t] = mfma_bf(sa.hi, bc[t][1], acc[t]);
            acc[t] = mfma_bf(sa.hi, bc[t][0], acc[t]);
        }
    }
}

template <int TPW, class F>
__device__ __forceinline__ void l3_foreach(f32x16 (&acc)[TPW], int wave, int lane, F f) {
    const int j = lane & 31, h = lane >> 5;
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int nt = (wave >> 1) * TPW + t, mt = wave & 1;
#pragma unroll
        for (int r = 0; r < 16; ++r) f(mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, nt * 32 + j, (float)acc[t][r]);
    }
}

}  // namespace mcr
