// Shared pieces of the split-precision fused local transformers (local_pct5.hip: exact bf16 hi/mid/lo, six MFMAs per
// fp32 product; local_pct6.hip: fp16 hi/lo, three MFMAs): parameter-blob layouts, the operand splits, the MFMA
// wrappers and the exact-erf GELU.  Reference mapping in local_pct.hip (SconeOcc.py:104-130).
#pragma once
#include "nn_kernels.h"

namespace mcr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

constexpr int L3_T = 64, L3_QPB = 4;
// ---- blob of variant 5: matrices as [n-tile][k16 step][plane hi,mid,lo][lane][8 bf16] (1.5 floats per weight), then
// the vectors of the v1 blob
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

// ---- blob of variant 6: matrices as [n-tile][k16 step][plane hi,lo][lane][8 fp16] (1 float per weight), every matrix
// multiplied by a power of two 2^e_i on the host so that its fp16 low plane stays in the normal range; then the same
// vectors (every bias times its matrix's 2^e_i); then 16 floats: 2^-e_i for the 15 matrices (applied in the epilogues) + one
// pad; then 16 floats: 2^e_i (products accumulated onto the residual stream scale it up first)
constexpr int L6_MAT_K16 = 128 * 16, L6_MAT_128 = 128 * 128, L6_MAT_QKV = 192 * 128;
__host__ __device__ constexpr int l6_mat_off(int idx) {
    int off = 0;
    for (int i = 0; i < idx; ++i) {
        const bool is_qkv = (i >= 2 && i < 14 && ((i - 2) % 6) == 0);
        off += i == 0 ? L6_MAT_K16 : (is_qkv ? L6_MAT_QKV : L6_MAT_128);
    }
    return off;
}
constexpr int L6_MATS_TOTAL = l6_mat_off(15);
constexpr int L6_SCALES = L3_VECS_TOTAL;                      // offset of the 16 inverse scales inside the vector section
constexpr int L6_BLOB_FLOATS = L6_MATS_TOTAL + L3_VECS_TOTAL + 32;

// ---- blob of variant 7 (opt-in 16-bit matrix path): matrices as [n-tile][k16 step][lane][8 fp16] -- the HIGH plane of variant 6's
// blob WITHOUT the per-matrix power of two (fp16(W), half a float per weight); then the same vectors (biases as they are)
constexpr int L7_MAT_K16 = 128 * 16 / 2, L7_MAT_128 = 128 * 128 / 2, L7_MAT_QKV = 192 * 128 / 2;
__host__ __device__ constexpr int l7_mat_off(int idx) {
    int off = 0;
    for (int i = 0; i < idx; ++i) {
        const bool is_qkv = (i >= 2 && i < 14 && ((i - 2) % 6) == 0);
        off += i == 0 ? L7_MAT_K16 : (is_qkv ? L7_MAT_QKV : L7_MAT_128);
    }
    return off;
}
constexpr int L7_MATS_TOTAL = l7_mat_off(15);
constexpr int L7_BLOB_FLOATS = L7_MATS_TOTAL + L3_VECS_TOTAL;

__device__ __forceinline__ unsigned pack2h(const float a, const float b) {      // {fp16(b), fp16(a)}, round to nearest even: v_cvt_pk_f16_f32
    const f32x2 x = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(x, f16x2));
}

__device__ __forceinline__ float l3_gelu(float x) {           // exact-erf GELU, erf by Abramowitz-Stegun 7.1.26
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

// ---- exact three-way split of 8 consecutive fp32 values into packed bf16x8 planes (variant 5) ----
struct Split3 { uint4 hi, mid, lo; };
__device__ __forceinline__ unsigned pack_top(float a, float b) {           // {top16(b), top16(a)}: element 0 in the low half
    return __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, b), __builtin_bit_cast(unsigned, a), 0x07060302u);
}
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
__device__ __forceinline__ f32x16 mfma_bf(uint4 a, uint4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// ---- two-term fp16 split (variant 6): x = hi + lo + e with hi = fp16(x) (round to nearest even), lo = fp16(x - hi);
// x - hi is exact in fp32, |lo| <= 2^-11 |x|, so |e| <= max(2^-22 |x|, 2^-25) (the second bound when lo is an fp16
// subnormal).  Needs |x| < 65504 (fp16 range).  v_cvt_pk_f16_f32 converts two values per instruction.
struct Split2 { uint4 hi, lo; };
// Three instructions: v_cvt_pk_f16_f32, then x - hi as a mixed-precision fma (fp16 source, fp32 addend: exact) whose result is rounded
// to fp16 straight into the low / high half of the destination (v_fma_mixlo_f16 / v_fma_mixhi_f16) -- the same values as converting hi
// back, subtracting and converting again (five instructions; MCR_SPLIT2H_PLAIN selects that form for the A/B).
__device__ __forceinline__ void split2h(const float a, const float b, unsigned& hi, unsigned& lo) {
    const f32x2 x = {a, b};
    const f16x2 h = __builtin_convertvector(x, f16x2);
    hi = __builtin_bit_cast(unsigned, h);
#ifdef MCR_SPLIT2H_PLAIN
    const f32x2 r = x - __builtin_convertvector(h, f32x2);
    const f16x2 l = __builtin_convertvector(r, f16x2);
    lo = __builtin_bit_cast(unsigned, l);
#else
    unsigned l;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l) : "v"(hi), "v"(a));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(hi), "v"(b));
    lo = l;
#endif
}
__device__ __forceinline__ Split2 split8h(const float4 p, const float4 q) {
    Split2 s;
    split2h(p.x, p.y, s.hi.x, s.lo.x);
    split2h(p.z, p.w, s.hi.y, s.lo.y);
    split2h(q.x, q.y, s.hi.z, s.lo.z);
    split2h(q.z, q.w, s.hi.w, s.lo.w);
    return s;
}
__device__ __forceinline__ f32x16 mfma_h(uint4 a, uint4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

}  // namespace mcr
