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
#include "lp_split.h"

namespace mcr {

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
