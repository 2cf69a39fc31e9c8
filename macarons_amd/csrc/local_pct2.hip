// K5-local v2 — fused per-query local PCTransformer, two workgroups per CU.
//
// Same math and blob as local_pct.hip (see there for the reference mapping, SconeOcc.py:104-130), re-laid for
// occupancy: the PMC profile of v1 showed the MFMA pipe 49.5 % busy with one wave per SIMD (84.5 KB LDS, 350
// registers): epilogues, attention, LayerNorm and every s_waitcnt were exposed.  v2 fits TWO workgroups per CU:
//   * LDS exactly 80 KB: xs [64][128] + sc [64][192] without padding; bank conflicts are avoided by XOR-swizzling
//     the 16-byte chunk index with (row & 15) on every access (A-fragment ds_read_b128 stay conflict free).
//   * LayerNorm statistics live in registers: lane l of every wave computes (mu, rstd) of row l (two read-only
//     sweeps), epilogues fetch them with ds_bpermute.  x is NOT centred: y = rstd * (acc - mu * s_n) + c_n with
//     s_n = sum_k W'[n][k] folded on the host; residual adds read the raw x.
//   * <= 256 VGPR+AGPR (launch_bounds(256, 2)); GELU uses a 1.5e-7-accurate erf (Abramowitz-Stegun 7.1.26).
#include "nn_kernels.h"

namespace mcr {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int L2_T = 64, L2_QPB = 4, L2_XW = 128, L2_SW = 192;
constexpr int L2_MAT_K8 = 128 * 8, L2_MAT_128 = 128 * 128, L2_MAT_QKV = 192 * 128;
__host__ __device__ constexpr int l2_mat_off(int idx) {
    int off = 0;
    for (int i = 0; i < idx; ++i) {
        const bool is_qkv = (i >= 2 && i < 14 && ((i - 2) % 6) == 0);
        off += i == 0 ? L2_MAT_K8 : (is_qkv ? L2_MAT_QKV : L2_MAT_128);
    }
    return off;
}
constexpr int L2_MATS_TOTAL = l2_mat_off(15);
constexpr int L2_VEC_EMB1 = 0, L2_VEC_EMB2 = 128, L2_VEC_ENC0 = 256, L2_VEC_ENC_STRIDE = 192 + 128 + 256 + 128,
              L2_VEC_LIN0 = L2_VEC_ENC0 + 2 * L2_VEC_ENC_STRIDE, L2_VEC_V1_TOTAL = L2_VEC_LIN0 + 128;
// appended for v2: column sums of the gamma-folded weights
constexpr int L2_VEC_S_ENC0 = L2_VEC_V1_TOTAL, L2_VEC_S_ENC_STRIDE = 192 + 256, L2_VEC_S_LIN0 = L2_VEC_S_ENC0 + 2 * L2_VEC_S_ENC_STRIDE,
              L2_VECS_TOTAL = L2_VEC_S_LIN0 + 128;
constexpr int L2_BLOB_FLOATS = L2_MATS_TOTAL + L2_VECS_TOTAL;

// swizzled element / chunk addressing: 16-byte chunk index XOR (row & 15)
__device__ __forceinline__ int swz(int row, int col, int width) { return row * width + ((((col >> 2) ^ (row & 15)) << 2) | (col & 3)); }
__device__ __forceinline__ int swz4(int row, int chunk, int width) { return row * width + ((chunk ^ (row & 15)) << 2); }

// erf with |error| <= 1.5e-7 (A&S 7.1.26) -> exact-erf GELU within 3e-7 * |x|
__device__ __forceinline__ float l2_gelu(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * z * z);
    const float erfa = fmaf(-p * t, e, 1.0f);                 // erf(|x|/sqrt2)
    return 0.5f * x + 0.5f * fabsf(x) * erfa;                  // 0.5 x (1 + sign(x) erf)
}

template <int K, int TPW, bool INIT>
__device__ __forceinline__ void l2_gemm(f32x16 (&acc)[TPW], const float* __restrict__ A, int width, int chunk0,
                                        const float* __restrict__ Wp, int wave, int lane) {
    constexpr int G = K / 8;
    const int i = lane & 31, h = lane >> 5;
    const float4* bp[TPW];
    const float* arow[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int id = wave * TPW + t, nt = id >> 1, mt = id & 1;
        bp[t] = reinterpret_cast<const float4*>(Wp) + (size_t)nt * G * 64 + lane;
        arow[t] = A + (mt * 32 + i) * width;
        if (INIT) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        }
    }
    const int sw = i & 15;
    constexpr int PF = G < 3 ? G : 3;
    float4 b[PF][TPW];
#pragma unroll
    for (int p = 0; p < PF; ++p)
#pragma unroll
        for (int t = 0; t < TPW; ++t) b[p][t] = bp[t][p * 64];
    float4 a_cur[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) a_cur[t] = *reinterpret_cast<const float4*>(arow[t] + (((chunk0 + h * G) ^ sw) << 2));
    __builtin_amdgcn_sched_group_barrier(0x020, PF * TPW, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, TPW, 0);
#pragma unroll
    for (int g = 0; g < G; ++g) {
        float4 a_nxt[TPW], bc[TPW];
#pragma unroll
        for (int t = 0; t < TPW; ++t) bc[t] = b[g % PF][t];
        if (g + 1 < G) {
#pragma unroll
            for (int t = 0; t < TPW; ++t) a_nxt[t] = *reinterpret_cast<const float4*>(arow[t] + (((chunk0 + h * G + g + 1) ^ sw) << 2));
        }
        if (g + PF < G) {
#pragma unroll
            for (int t = 0; t < TPW; ++t) b[g % PF][t] = bp[t][(g + PF) * 64];
        }
#pragma unroll
        for (int t = 0; t < TPW; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[t].x, bc[t].x, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < TPW; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[t].y, bc[t].y, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < TPW; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[t].z, bc[t].z, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < TPW; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[t].w, bc[t].w, acc[t], 0, 0, 0);
        if (g + 1 < G) __builtin_amdgcn_sched_group_barrier(0x100, TPW, 0);
        if (g + PF < G) __builtin_amdgcn_sched_group_barrier(0x020, TPW, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * TPW, 0);
        if (g + 1 < G) {
#pragma unroll
            for (int t = 0; t < TPW; ++t) a_cur[t] = a_nxt[t];
        }
    }
}

template <int TPW, class F>
__device__ __forceinline__ void l2_foreach(f32x16 (&acc)[TPW], int wave, int lane, F f) {
    const int j = lane & 31, h = lane >> 5;
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int id = wave * TPW + t, nt = id >> 1, mt = id & 1;
#pragma unroll
        for (int r = 0; r < 16; ++r) f(mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, nt * 32 + j, (float)acc[t][r]);
    }
}

// LayerNorm statistics of row `lane` (64 rows <-> 64 lanes), computed redundantly by every wave: read-only.
__device__ __forceinline__ void l2_stats(const float* src, int width, int chunk0, int lane, float& mu, float& rstd) {
    const float* row = src + lane * width;
    const int sw = lane & 15;
    float s = 0.f;
#pragma unroll 8
    for (int c = 0; c < 32; ++c) {
        const float4 q = *reinterpret_cast<const float4*>(row + (((chunk0 + c) ^ sw) << 2));
        s += (q.x + q.y) + (q.z + q.w);
    }
    mu = s * (1.0f / 128.f);
    float v = 0.f;
#pragma unroll 8
    for (int c = 0; c < 32; ++c) {
        const float4 q = *reinterpret_cast<const float4*>(row + (((chunk0 + c) ^ sw) << 2));
        const float a = q.x - mu, b = q.y - mu, cc = q.z - mu, d = q.w - mu;
        v += (a * a + b * b) + (cc * cc + d * d);
    }
    rstd = 1.0f / sqrtf(v * (1.0f / 128.f) + 1e-5f);
}

__global__ __launch_bounds__(256, 2) void local_pct2_kernel(const float* __restrict__ offs, float* __restrict__ feat,
                                                           long long ld_feat, long long S,
                                                           const float* __restrict__ blob) {
    __shared__ __attribute__((aligned(16))) float xs[L2_T * L2_XW];
    __shared__ __attribute__((aligned(16))) float sc[L2_T * L2_SW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* mats = blob;
    const float* vecs = blob + L2_MATS_TOTAL;
    const long long s0 = (long long)blockIdx.x * L2_QPB;

    // ---- offsets (64 x 3, zero-padded to K = 8) -> sc chunks 0,1 ----
    if (tid < L2_T) {
        const long long seq = s0 + (tid >> 4);
        float x = 0.f, y = 0.f, z = 0.f;
        if (seq < S) {
            const float* p = offs + (seq * 16 + (tid & 15)) * 3;
            x = p[0]; y = p[1]; z = p[2];
        }
        *reinterpret_cast<float4*>(sc + swz4(tid, 0, L2_SW)) = make_float4(x, y, z, 0.f);
        *reinterpret_cast<float4*>(sc + swz4(tid, 1, L2_SW)) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    f32x16 acc2[2];
    // ---- Embedding: linear1 (3->125) GELU -> xs ; linear2 (125->125) || xyz -> sc[:, 64:192] ----
    l2_gemm<8, 2, true>(acc2, sc, L2_SW, 0, mats + l2_mat_off(0), wave, lane);
    l2_foreach<2>(acc2, wave, lane, [&](int row, int col, float v) { xs[swz(row, col, L2_XW)] = l2_gelu(v + vecs[L2_VEC_EMB1 + col]); });
    __syncthreads();
    l2_gemm<128, 2, true>(acc2, xs, L2_XW, 0, mats + l2_mat_off(1), wave, lane);
    l2_foreach<2>(acc2, wave, lane, [&](int row, int col, float v) {
        sc[swz(row, 64 + col, L2_SW)] = col < 125 ? v + vecs[L2_VEC_EMB2 + col] : sc[swz(row, col - 125, L2_SW)];
    });
    __syncthreads();
    // move x into xs (raw, un-centred): wave w copies chunk range [8w, 8w+8) of every row
    {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int ch = wave * 8 + c;
            *reinterpret_cast<float4*>(xs + swz4(lane, ch, L2_XW)) = *reinterpret_cast<const float4*>(sc + swz4(lane, 16 + ch, L2_SW));
        }
    }
    __syncthreads();

    float mu, rstd;
#pragma unroll 1
    for (int e = 0; e < 2; ++e) {
        const float* em = mats + l2_mat_off(2 + 6 * e);
        const float* ev = vecs + L2_VEC_ENC0 + e * L2_VEC_ENC_STRIDE;
        const float* es = vecs + L2_VEC_S_ENC0 + e * L2_VEC_S_ENC_STRIDE;
        // ---- norm1 (folded) + QKV ----
        l2_stats(xs, L2_XW, 0, lane, mu, rstd);
        {
            f32x16 acc3[3];
            l2_gemm<128, 3, true>(acc3, xs, L2_XW, 0, em, wave, lane);
            l2_foreach<3>(acc3, wave, lane, [&](int row, int col, float v) {
                const float m = __shfl(mu, row, 64), r = __shfl(rstd, row, 64);
                sc[swz(row, col, L2_SW)] = fmaf(r, fmaf(-m, es[col], v), ev[col]);
            });
        }
        __syncthreads();
        // ---- attention: thread = (query = wave, head, row); output overwrites the head's V chunks ----
        {
            const int hh = (lane >> 4) & 3, qi = lane & 15, rb = wave * 16;
            float q[8];
            {
                const float4 q0 = *reinterpret_cast<const float4*>(sc + swz4(rb + qi, hh * 2, L2_SW));
                const float4 q1 = *reinterpret_cast<const float4*>(sc + swz4(rb + qi, hh * 2 + 1, L2_SW));
                q[0] = q0.x; q[1] = q0.y; q[2] = q0.z; q[3] = q0.w; q[4] = q1.x; q[5] = q1.y; q[6] = q1.z; q[7] = q1.w;
            }
            float p[16];
            float mx = -__builtin_inff();
#pragma unroll 4
            for (int j = 0; j < 16; ++j) {
                const float4 k0 = *reinterpret_cast<const float4*>(sc + swz4(rb + j, 8 + hh * 2, L2_SW));
                const float4 k1 = *reinterpret_cast<const float4*>(sc + swz4(rb + j, 8 + hh * 2 + 1, L2_SW));
                float a = q[0] * k0.x;
                a = fmaf(q[1], k0.y, a); a = fmaf(q[2], k0.z, a); a = fmaf(q[3], k0.w, a);
                a = fmaf(q[4], k1.x, a); a = fmaf(q[5], k1.y, a); a = fmaf(q[6], k1.z, a); a = fmaf(q[7], k1.w, a);
                p[j] = a * 0.35355339059327376220f;
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
#pragma unroll 2
            for (int j = 0; j < 16; ++j) {
                const float pj = p[j] * inv;
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const float4 vv = *reinterpret_cast<const float4*>(sc + swz4(rb + j, 16 + hh * 8 + c, L2_SW));
                    o[4 * c] = fmaf(pj, vv.x, o[4 * c]); o[4 * c + 1] = fmaf(pj, vv.y, o[4 * c + 1]);
                    o[4 * c + 2] = fmaf(pj, vv.z, o[4 * c + 2]); o[4 * c + 3] = fmaf(pj, vv.w, o[4 * c + 3]);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // every V read of this wave precedes the writes
#pragma unroll
            for (int c = 0; c < 8; ++c)
                *reinterpret_cast<float4*>(sc + swz4(rb + qi, 16 + hh * 8 + c, L2_SW)) = make_float4(o[4 * c], o[4 * c + 1], o[4 * c + 2], o[4 * c + 3]);
        }
        __syncthreads();
        // ---- out projection + residual: x += att W_o^T + b ----
        l2_gemm<128, 2, true>(acc2, sc, L2_SW, 16, em + L2_MAT_QKV, wave, lane);
        l2_foreach<2>(acc2, wave, lane, [&](int row, int col, float v) {
            float* px = xs + swz(row, col, L2_XW);
            *px = *px + (v + ev[192 + col]);
        });
        __syncthreads();
        // ---- norm2 (folded) + FF in two 128-wide halves + residual ----
        l2_stats(xs, L2_XW, 0, lane, mu, rstd);
        f32x16 accf[2];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            l2_gemm<128, 2, true>(acc2, xs, L2_XW, 0, em + L2_MAT_QKV + L2_MAT_128 * (1 + half), wave, lane);
            l2_foreach<2>(acc2, wave, lane, [&](int row, int col, float v) {
                const float m = __shfl(mu, row, 64), r = __shfl(rstd, row, 64);
                sc[swz(row, col, L2_SW)] = l2_gelu(fmaf(r, fmaf(-m, es[192 + half * 128 + col], v), ev[192 + 128 + half * 128 + col]));
            });
            __syncthreads();
            if (half == 0) l2_gemm<128, 2, true>(accf, sc, L2_SW, 0, em + L2_MAT_QKV + L2_MAT_128 * 3, wave, lane);
            else l2_gemm<128, 2, false>(accf, sc, L2_SW, 0, em + L2_MAT_QKV + L2_MAT_128 * 4, wave, lane);
            __syncthreads();
        }
        l2_foreach<2>(accf, wave, lane, [&](int row, int col, float v) {
            float* px = xs + swz(row, col, L2_XW);
            *px = *px + (v + ev[192 + 128 + 256 + col]);
        });
        __syncthreads();
    }
    // ---- final norm (folded) + linear0 ----
    l2_stats(xs, L2_XW, 0, lane, mu, rstd);
    l2_gemm<128, 2, true>(acc2, xs, L2_XW, 0, mats + l2_mat_off(14), wave, lane);
    l2_foreach<2>(acc2, wave, lane, [&](int row, int col, float v) {
        const float m = __shfl(mu, row, 64), r = __shfl(rstd, row, 64);
        sc[swz(row, col, L2_SW)] = fmaf(r, fmaf(-m, vecs[L2_VEC_S_LIN0 + col], v), vecs[L2_VEC_LIN0 + col]);
    });
    __syncthreads();
    // ---- max || avg pool over the 16 tokens of each query ----
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int o = tid + r * 256, q = o >> 7, c = o & 127;
        if (s0 + q < S) {
            float mx = -__builtin_inff(), sm = 0.f;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const float v = sc[swz(q * 16 + j, c, L2_SW)];
                mx = fmaxf(mx, v);
                sm += v;
            }
            feat[(s0 + q) * ld_feat + c] = mx;
            feat[(s0 + q) * ld_feat + 128 + c] = sm * (1.0f / 16.f);
        }
    }
}

void launch_local_pct2(hipStream_t s, const float* offs, float* feat, int64_t ld_feat, int64_t S, const float* blob) {
    if (S <= 0) return;
    hipLaunchKernelGGL(local_pct2_kernel, dim3((unsigned)cdiv(S, L2_QPB)), dim3(256), 0, s, offs, feat, (long long)ld_feat,
                       (long long)S, blob);
}

int local_pct_blob_floats() { return L2_BLOB_FLOATS; }

}  // namespace mcr
