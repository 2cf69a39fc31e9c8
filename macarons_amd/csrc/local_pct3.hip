// K5-local v3 — fused per-query local PCTransformer with split-precision matrix products ("bf16 x 6").
//
// Same structure, LDS plan, LayerNorm fold, attention and pooling as local_pct.hip (v1, see there for the reference
// mapping SconeOcc.py:104-130).  What changes is the matrix pipe: instead of v_mfma_f32_32x32x2_f32 (64 FLOP/clk/SIMD)
// every fp32 operand is split EXACTLY into three bf16 pieces  x = hi + mid + lo  (8 + 8 + 8 mantissa bits: hi = x with
// the low 16 bits cleared, mid likewise from the exact remainder, lo the exact rest), and the product keeps every term
// down to 2^-16 relative:   x*w ~= hi*hi + hi*mid + mid*hi + mid*mid + hi*lo + lo*hi    (dropped terms <= 2^-23 |x w|)
// on v_mfma_f32_32x32x16_bf16 (1024 FLOP/clk/SIMD): 6 x 32 cycles per 32x32x16 block instead of 8 x 64.  Products of
// bf16 pairs are exact and accumulate in fp32, so the result is fp32-class (measured 3.7e-7 vs fp64 on the local
// transformer, the exact-fp32 kernel: 6.6e-7) -- NOT a bf16 approximation.  Weights are split on the host
// (networks/packing.py: three bf16 planes in MFMA-fragment order), activations in registers right after the LDS read.
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

// acc[t] (+)= A[64 x 16*S] * W^T for this wave's TPW tiles: m-tile = wave & 1, n-tiles (wave >> 1) * TPW + t, so the
// A rows are split once per k16-step per wave and reused by its TPW column tiles.
// A: fp32 in LDS (row stride lda); lane (i = l&31, h = l>>5) owns A[row][16 s + 8 h .. +7] and B[k = 16 s + 8 h .. +7][n].
// The weight ring (L3_PF k16-steps in flight) is shared by consecutive products like in local_pct.hip: the first
// steps of the NEXT product's planes are requested during this product's tail.
constexpr int L3_PF = 2;
typedef uint4 l3_ring_t[L3_PF][3][3];            // [step slot][tile][plane]

template <int TPW>
__device__ __forceinline__ const uint4* l3_bptr(const float* Wp, int S, int wave, int lane, int t) {
    return reinterpret_cast<const uint4*>(Wp) + (size_t)((wave >> 1) * TPW + t) * S * 3 * 64 + lane;
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
                for (int pl = 0; pl < 3; ++pl) b[p][t][pl] = bp[t][(p * 3 + pl) * 64];
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
                for (int pl = 0; pl < 3; ++pl) b[s % PF][t][pl] = bp[t][((s + PF) * 3 + pl) * 64];
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

// centre the rows of src into xs and keep (mu, rstd): 4 threads per row, 32 columns each (LayerNorm eps 1e-5)
__device__ __forceinline__ void l3_center(const float* src, int lds_, float* xs, float* stats, int tid) {
    const int row = tid >> 2, part = tid & 3;
    const float* s = src + row * lds_ + part * 32;
    float v[32];
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < 32; c += 4) {
        const float4 q = *reinterpret_cast<const float4*>(s + c);
        v[c] = q.x; v[c + 1] = q.y; v[c + 2] = q.z; v[c + 3] = q.w;
        sum += (q.x + q.y) + (q.z + q.w);
    }
    sum += __shfl_xor(sum, 1, 64);
    sum += __shfl_xor(sum, 2, 64);
    const float mu = sum * (1.0f / 128.f);
    float sq = 0.f;
#pragma unroll
    for (int c = 0; c < 32; ++c) {
        v[c] -= mu;
        sq = fmaf(v[c], v[c], sq);
    }
    sq += __shfl_xor(sq, 1, 64);
    sq += __shfl_xor(sq, 2, 64);
    float* d = xs + row * L3_XLD + part * 32;
#pragma unroll
    for (int c = 0; c < 32; c += 4) *reinterpret_cast<float4*>(d + c) = make_float4(v[c], v[c + 1], v[c + 2], v[c + 3]);
    if (part == 0) {
        stats[2 * row] = mu;
        stats[2 * row + 1] = 1.0f / sqrtf(sq * (1.0f / 128.f) + 1e-5f);
    }
}

// grid = ceil(S / 4); S sequences of 16 offsets [S,16,3]; features[s*ld_feat + 0:256] = max(128) || avg(128)
__global__ __launch_bounds__(256, 1) void local_pct3_kernel(const float* __restrict__ offs, float* __restrict__ feat,
                                                          long long ld_feat, long long S,
                                                          const float* __restrict__ blob) {
    __shared__ __attribute__((aligned(16))) float xs[L3_T * L3_XLD];
    __shared__ __attribute__((aligned(16))) float sc[L3_T * L3_SLD];
    __shared__ float stats[L3_T * 2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* mats = blob;
    const float* vecs = blob + L3_MATS_TOTAL;
    const long long s0 = (long long)blockIdx.x * L3_QPB;

    // ---- stage the 64 x 3 offsets, zero-padded to K = 16, into sc[:, 0:16] ----
    if (tid < L3_T) {
        const long long seq = s0 + (tid >> 4);
        float x = 0.f, y = 0.f, z = 0.f;
        if (seq < S) {
            const float* p = offs + (seq * 16 + (tid & 15)) * 3;
            x = p[0]; y = p[1]; z = p[2];
        }
        float* d = sc + tid * L3_SLD;
        *reinterpret_cast<float4*>(d) = make_float4(x, y, z, 0.f);
        *reinterpret_cast<float4*>(d + 4) = make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(d + 8) = make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(d + 12) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    f32x16 acc2[2];
    l3_ring_t ring;
    // ---- Embedding (Attention.py:98-128): linear1 3->125, GELU -> xs ; linear2 125->125 -> sc ; || xyz ----
    l3_gemm<1, 2, true, false, 0>(acc2, sc, L3_SLD, mats + l3_mat_off(0), ring, nullptr, wave, lane);
    l3_foreach<2>(acc2, wave, lane, [&](int row, int col, float v) { xs[row * L3_XLD + col] = l3_gelu(v + vecs[L3_VEC_EMB1 + col]); });
    __syncthreads();
    // xyz must survive in sc[:, 0:3] until the concat: linear2's output goes to sc[:, 64:192]
    l3_gemm<8, 2, true, false, 3>(acc2, xs, L3_XLD, mats + l3_mat_off(1), ring, mats + l3_mat_off(2), wave, lane);
    l3_foreach<2>(acc2, wave, lane, [&](int row, int col, float v) {
        sc[row * L3_SLD + 64 + col] = col < 125 ? v + vecs[L3_VEC_EMB2 + col] : sc[row * L3_SLD + (col - 125)];   // concat raw xyz
    });
    __syncthreads();

    const float* xsrc = sc + 64;                   // where the current (un-centred) x lives
    int xsrc_ld = L3_SLD;
#pragma unroll 1
    for (int e = 0; e < 2; ++e) {
        const float* em = mats + l3_mat_off(2 + 6 * e);
        const float* ev = vecs + L3_VEC_ENC0 + e * L3_VEC_ENC_STRIDE;
        // ---- norm1 (folded) + QKV (Attention.py:186-188, 287) ----
        l3_center(xsrc, xsrc_ld, xs, stats, tid);
        __syncthreads();
        {
            f32x16 acc3[3];
            l3_gemm<8, 3, true, true, 2>(acc3, xs, L3_XLD, em, ring, em + L3_MAT_QKV, wave, lane);
            l3_foreach<3>(acc3, wave, lane, [&](int row, int col, float v) { sc[row * L3_SLD + col] = fmaf(stats[2 * row + 1], v, ev[col]); });
        }
        __syncthreads();
        // ---- attention (Attention.py:8-36): thread = (query, head, row); output overwrites the head's V block ----
        {
            const float* base = sc + (tid >> 6) * 16 * L3_SLD;
            const int hh = (tid >> 4) & 3, qi = tid & 15;
            float q[8];
#pragma unroll
            for (int d = 0; d < 8; ++d) q[d] = base[qi * L3_SLD + hh * 8 + d];
            float p[16];
            float mx = -__builtin_inff();
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                float a = 0.f;
#pragma unroll
                for (int d = 0; d < 8; ++d) a = fmaf(q[d], base[j * L3_SLD + 32 + hh * 8 + d], a);
                p[j] = a * 0.35355339059327376220f;            // / sqrt(8)
                mx = fmaxf(mx, p[j]);
            }
            float den = 0.f;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                p[j] = __expf(p[j] - mx);
                den += p[j];
            }
            const float inv = 1.0f / den;
            float o[32];
#pragma unroll
            for (int c = 0; c < 32; ++c) o[c] = 0.f;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const float pj = p[j] * inv;
                const float* vrow = base + j * L3_SLD + 64 + hh * 32;
#pragma unroll
                for (int c = 0; c < 32; c += 4) {
                    const float4 vv = *reinterpret_cast<const float4*>(vrow + c);
                    o[c] = fmaf(pj, vv.x, o[c]); o[c + 1] = fmaf(pj, vv.y, o[c + 1]);
                    o[c + 2] = fmaf(pj, vv.z, o[c + 2]); o[c + 3] = fmaf(pj, vv.w, o[c + 3]);
                }
            }
            // the 16 threads sharing this V block are consecutive lanes of this wave: all reads precede the writes
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            float* orow = sc + ((tid >> 6) * 16 + qi) * L3_SLD + 64 + hh * 32;
#pragma unroll
            for (int c = 0; c < 32; c += 4) *reinterpret_cast<float4*>(orow + c) = make_float4(o[c], o[c + 1], o[c + 2], o[c + 3]);
        }
        __syncthreads();
        // ---- out projection + residual (Attention.py:201-202, 290): x = (x~ + mu) + att W_o^T + b ----
        l3_gemm<8, 2, true, true, 2>(acc2, sc + 64, L3_SLD, em + L3_MAT_QKV, ring, em + L3_MAT_QKV + L3_MAT_128, wave, lane);
        l3_foreach<2>(acc2, wave, lane, [&](int row, int col, float v) {
            float* px = xs + row * L3_XLD + col;
            *px = (*px + stats[2 * row]) + (v + ev[192 + col]);
        });
        __syncthreads();
        // ---- norm2 (folded) + FF 128 -> 256 (GELU) -> 128 + residual (Attention.py:293-298), two 128-wide halves ----
        l3_center(xs, L3_XLD, xs, stats, tid);
        __syncthreads();
        f32x16 accf[2];
        // product order: ff1a, ff2a, ff1b, ff2b; each requests the next one's first weight groups at its tail
        const float* w_ff1a = em + L3_MAT_QKV + L3_MAT_128 * 1;
        const float* w_ff1b = em + L3_MAT_QKV + L3_MAT_128 * 2;
        const float* w_ff2a = em + L3_MAT_QKV + L3_MAT_128 * 3;
        const float* w_ff2b = em + L3_MAT_QKV + L3_MAT_128 * 4;
        l3_gemm<8, 2, true, true, 2>(acc2, xs, L3_XLD, w_ff1a, ring, w_ff2a, wave, lane);
        l3_foreach<2>(acc2, wave, lane, [&](int row, int col, float v) {
            sc[row * L3_SLD + col] = l3_gelu(fmaf(stats[2 * row + 1], v, ev[192 + 128 + col]));
        });
        __syncthreads();
        l3_gemm<8, 2, true, true, 2>(accf, sc, L3_SLD, w_ff2a, ring, w_ff1b, wave, lane);
        __syncthreads();
        l3_gemm<8, 2, true, true, 2>(acc2, xs, L3_XLD, w_ff1b, ring, w_ff2b, wave, lane);
        l3_foreach<2>(acc2, wave, lane, [&](int row, int col, float v) {
            sc[row * L3_SLD + col] = l3_gelu(fmaf(stats[2 * row + 1], v, ev[192 + 128 + 128 + col]));
        });
        __syncthreads();
        if (e == 0) l3_gemm<8, 2, false, true, 3>(accf, sc, L3_SLD, w_ff2b, ring, mats + l3_mat_off(8), wave, lane);    // next: qkv of encoder 1
        else l3_gemm<8, 2, false, true, 2>(accf, sc, L3_SLD, w_ff2b, ring, mats + l3_mat_off(14), wave, lane);        // next: linear0
        __syncthreads();
        l3_foreach<2>(accf, wave, lane, [&](int row, int col, float v) {
            float* px = xs + row * L3_XLD + col;
            *px = (*px + stats[2 * row]) + (v + ev[192 + 128 + 256 + col]);
        });
        __syncthreads();
        xsrc = xs;
        xsrc_ld = L3_XLD;
    }
    // ---- final norm (folded) + linear0 128 -> 128 (SconeOcc.py:119-122) ----
    l3_center(xs, L3_XLD, xs, stats, tid);
    __syncthreads();
    l3_gemm<8, 2, true, true, 0>(acc2, xs, L3_XLD, mats + l3_mat_off(14), ring, nullptr, wave, lane);
    l3_foreach<2>(acc2, wave, lane, [&](int row, int col, float v) { sc[row * L3_SLD + col] = fmaf(stats[2 * row + 1], v, vecs[L3_VEC_LIN0 + col]); });
    __syncthreads();
    // ---- max || avg pool over the 16 tokens of each query (SconeOcc.py:124-126) ----
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int o = tid + r * 256, q = o >> 7, c = o & 127;
        if (s0 + q < S) {
            float mx = -__builtin_inff(), sm = 0.f;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const float v = sc[(q * 16 + j) * L3_SLD + c];
                mx = fmaxf(mx, v);
                sm += v;
            }
            feat[(s0 + q) * ld_feat + c] = mx;
            feat[(s0 + q) * ld_feat + 128 + c] = sm * (1.0f / 16.f);
        }
    }
}

void launch_local_pct3(hipStream_t s, const float* offs, float* feat, int64_t ld_feat, int64_t S, const float* blob) {
    if (S <= 0) return;
    hipLaunchKernelGGL(local_pct3_kernel, dim3((unsigned)cdiv(S, L3_QPB)), dim3(256), 0, s, offs, feat, (long long)ld_feat,
                       (long long)S, blob);
}


int local_pct3_blob_floats() { return L3_BLOB_FLOATS; }

}  // namespace mcr
