// K5-local — fused per-query local PCTransformer of SconeOcc for gfx950.
//
// Replaces, for the k = 16 neighbourhood sequences, the whole of macarons/networks/SconeOcc.py:104-130
// (PCTransformer.forward: Embedding -> 2 x Encoder -> LayerNorm -> linear0 -> max || avg pool) that the
// reference runs as ~40 eager ops over [Q*16, 128..256] tensors (1.6 GB per FF activation at Q = 100k).  Here a
// workgroup keeps 4 queries x 16 tokens x 128 channels in LDS from the xyz offsets to the pooled 256-float
// feature; only the kNN offsets are read from and the pooled feature written to HBM.  This is ~90 % of the
// FLOPs of an NBV step (3 scales x 7.9 MFLOP per query).
//
// Structure (4 waves, 64 tokens, 84 KB LDS -> 1 workgroup / CU, one wave per SIMD):
//   xs [64][132]  activations x (kept mean-centred between a LayerNorm and its residual add)
//   sc [64][196]  scratch: embedding input / q|k|v (attention output overwrites v in place) / FF hidden half
//   * every matrix product is v_mfma_f32_32x32x2_f32 (exact fp32 fma chain).  A fragments come from LDS
//     (ds_read_b128, row stride = 4 mod 64 banks -> conflict free); B fragments stream from L2 out of a
//     host-packed image [n-tile][k-group][lane][4] so each wave-load is one contiguous 1 KB line set, three
//     groups in flight ahead of the MFMAs.
//   * LayerNorm is folded: x is centred in place (mu, rstd kept per row), gamma is folded into the next
//     weight (W' = W * gamma) and beta into its bias (c = b + W beta) on the host, so the GEMM epilogue is
//     y = rstd * acc + c and the residual add is x~ + mu + out.  No normalised copy of x is materialised.
//   * FF (128 -> 256 -> 128) runs as two 128-wide halves; the second GEMM accumulates across halves in
//     registers.
//   * attention: one thread per (query, head, row): 16 scores (d = 8), softmax, 32-wide output written over
//     the head's V block (the 16 threads of a (query, head) are consecutive lanes of one wave).
#include "nn_kernels.h"

namespace mcr {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int LP_T = 64;          // tokens per workgroup (4 queries x 16)
constexpr int LP_QPB = 4;         // queries per workgroup
constexpr int LP_XLD = 132;       // xs row stride (floats)
constexpr int LP_SLD = 196;       // sc row stride (floats)
constexpr int LP_E = 128;

// ---- packed parameter blob (built on the host: macarons_amd/networks/packing.py) -------------------------------
// matrices (each [NT][G][64][4] floats), in order:
//   0 emb1 (N128,K8)  1 emb2 (N128,K128)
//   per encoder e (base 2 + 6e): qkv (N192,K128)  out (N128,K128)  ff1a ff1b (N128,K128)  ff2a ff2b (N128,K128)
//   14 lin0 (N128,K128)
// vectors, in order: emb1_b[128] emb2_b[128] | per encoder: qkv_c[192] out_b[128] ff1_c[256] ff2_b[128] | lin0_c[128]
constexpr int LP_MAT_K8 = 128 * 8, LP_MAT_128 = 128 * 128, LP_MAT_QKV = 192 * 128;
__host__ __device__ constexpr int lp_mat_off(int idx) {
    // idx: 0 emb1, 1 emb2, 2+6e+{0 qkv,1 out,2 ff1a,3 ff1b,4 ff2a,5 ff2b}, 14 lin0
    int off = 0;
    for (int i = 0; i < idx; ++i) {
        const bool is_qkv = (i >= 2 && i < 14 && ((i - 2) % 6) == 0);
        off += i == 0 ? LP_MAT_K8 : (is_qkv ? LP_MAT_QKV : LP_MAT_128);
    }
    return off;
}
constexpr int LP_MATS_TOTAL = lp_mat_off(15);
constexpr int LP_VEC_EMB1 = 0, LP_VEC_EMB2 = 128, LP_VEC_ENC0 = 256, LP_VEC_ENC_STRIDE = 192 + 128 + 256 + 128,
              LP_VEC_LIN0 = LP_VEC_ENC0 + 2 * LP_VEC_ENC_STRIDE, LP_VECS_TOTAL = LP_VEC_LIN0 + 128;
constexpr int LP_BLOB_FLOATS = LP_MATS_TOTAL + LP_VECS_TOTAL;

// exact-erf GELU with erf from Abramowitz-Stegun 7.1.26 (|erf error| <= 1.5e-7): ~14 VALU ops instead of the ~35 of
// the branchy libm erff; the epilogues are not overlapped with MFMA work in this kernel, so they matter.
__device__ __forceinline__ float lp_gelu(float x) {
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

// ---- block GEMM: acc[t] (+)= A[64 x K] * Wp^T for this wave's TPW output tiles -----------------------------------
// tile id = wave*TPW + t  ->  n-tile = id >> 1, m-tile = id & 1.   A: LDS, row stride lda, K-halves per lane half.
//
// Software pipeline (the scheduler otherwise sinks every ds_read next to its first use and waits lgkmcnt(0) in
// front of each MFMA group: measured 28 % of the wave's life in s_waitcnt):
//   * A fragments (LDS) one k-group ahead, B fragments (L2) LP_PF k-groups ahead of the MFMAs consuming them;
//     sched_group_barrier pins  [next A reads][next B loads][4*TPW MFMAs]  per group;
//   * the B ring is SHARED BY CONSECUTIVE GEMMs: while the last LP_PF groups of this product are in the matrix pipe
//     the first LP_PF groups of the NEXT product's weights are already requested, so the epilogue / LayerNorm /
//     attention / barrier between two products hides the L2 latency instead of exposing a pipeline refill
//     (13 refills per workgroup otherwise).  Needs G % LP_PF == 0 so ring slots line up.
constexpr int LP_PF = 4;
typedef float4 lp_ring_t[LP_PF][3];

template <int TPW>
__device__ __forceinline__ const float4* lp_bptr(const float* Wp, int G, int wave, int lane, int t) {
    const int id = wave * TPW + t, nt = id >> 1;
    return reinterpret_cast<const float4*>(Wp) + (size_t)nt * G * 64 + lane;
}

// K = 128 products of the chain.  PRE: the ring already holds this product's first LP_PF groups.
// NEXT_TPW: tiles per wave of the next product whose first LP_PF groups are requested at the tail (0 = none).
template <int TPW, bool INIT, bool PRE, int NEXT_TPW>
__device__ __forceinline__ void lp_gemm128(f32x16 (&acc)[TPW], const float* __restrict__ A, int lda,
                                           const float* __restrict__ Wp, lp_ring_t& b, const float* __restrict__ next_Wp,
                                           int wave, int lane) {
    constexpr int K = 128, G = K / 8;
    static_assert(G % LP_PF == 0, "ring slots of consecutive products must line up");
    const int i = lane & 31, h = lane >> 5;
    const float4* bp[TPW];
    const float* ap[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        bp[t] = lp_bptr<TPW>(Wp, G, wave, lane, t);
        ap[t] = A + (((wave * TPW + t) & 1) * 32 + i) * lda + h * (K / 2);
        if (INIT) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        }
    }
    const float4* np[NEXT_TPW > 0 ? NEXT_TPW : 1];
#pragma unroll
    for (int t = 0; t < NEXT_TPW; ++t) np[t] = lp_bptr<(NEXT_TPW > 0 ? NEXT_TPW : 1)>(next_Wp, G, wave, lane, t);
    if (!PRE) {
#pragma unroll
        for (int p = 0; p < LP_PF; ++p)
#pragma unroll
            for (int t = 0; t < TPW; ++t) b[p][t] = bp[t][p * 64];
    }
    float4 a_cur[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) a_cur[t] = *reinterpret_cast<const float4*>(ap[t]);
    if (!PRE) __builtin_amdgcn_sched_group_barrier(0x020, LP_PF * TPW, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, TPW, 0);
#pragma unroll
    for (int g = 0; g < G; ++g) {
        float4 a_nxt[TPW], bc[TPW];
#pragma unroll
        for (int t = 0; t < TPW; ++t) bc[t] = b[g % LP_PF][t];
        if (g + 1 < G) {
#pragma unroll
            for (int t = 0; t < TPW; ++t) a_nxt[t] = *reinterpret_cast<const float4*>(ap[t] + 4 * (g + 1));
        }
        if (g + LP_PF < G) {
#pragma unroll
            for (int t = 0; t < TPW; ++t) b[g % LP_PF][t] = bp[t][(g + LP_PF) * 64];
        } else if (NEXT_TPW > 0) {
#pragma unroll
            for (int t = 0; t < NEXT_TPW; ++t) b[g % LP_PF][t] = np[t][(g + LP_PF - G) * 64];
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
        if (g + LP_PF < G) __builtin_amdgcn_sched_group_barrier(0x020, TPW, 0);
        else if (NEXT_TPW > 0) __builtin_amdgcn_sched_group_barrier(0x020, NEXT_TPW, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * TPW, 0);
        if (g + 1 < G) {
#pragma unroll
            for (int t = 0; t < TPW; ++t) a_cur[t] = a_nxt[t];
        }
    }
}

// the K = 8 embedding product (one k-group, no ring)
template <int TPW>
__device__ __forceinline__ void lp_gemm8(f32x16 (&acc)[TPW], const float* __restrict__ A, int lda,
                                         const float* __restrict__ Wp, int wave, int lane) {
    const int i = lane & 31, h = lane >> 5;
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const float4 bq = lp_bptr<TPW>(Wp, 1, wave, lane, t)[0];
        const float4 aq = *reinterpret_cast<const float4*>(A + (((wave * TPW + t) & 1) * 32 + i) * lda + h * 4);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq.x, bq.x, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq.y, bq.y, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq.z, bq.z, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq.w, bq.w, acc[t], 0, 0, 0);
    }
}

// visit every element this lane owns in its TPW tiles:  f(row 0..63, col 0..N-1, value&)
template <int TPW, class F>
__device__ __forceinline__ void lp_foreach(f32x16 (&acc)[TPW], int wave, int lane, F f) {
    const int j = lane & 31, h = lane >> 5;
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int id = wave * TPW + t, nt = id >> 1, mt = id & 1;
#pragma unroll
        for (int r = 0; r < 16; ++r) f(mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, nt * 32 + j, (float)acc[t][r]);
    }
}

// centre the rows of src into xs and keep (mu, rstd): 4 threads per row, 32 columns each (LayerNorm eps 1e-5)
__device__ __forceinline__ void lp_center(const float* src, int lds_, float* xs, float* stats, int tid) {
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
    float* d = xs + row * LP_XLD + part * 32;
#pragma unroll
    for (int c = 0; c < 32; c += 4) *reinterpret_cast<float4*>(d + c) = make_float4(v[c], v[c + 1], v[c + 2], v[c + 3]);
    if (part == 0) {
        stats[2 * row] = mu;
        stats[2 * row + 1] = 1.0f / sqrtf(sq * (1.0f / 128.f) + 1e-5f);
    }
}

// grid = ceil(S / 4); S sequences of 16 offsets [S,16,3]; features[s*ld_feat + 0:256] = max(128) || avg(128)
__global__ __launch_bounds__(256, 1) void local_pct_kernel(const float* __restrict__ offs, float* __restrict__ feat,
                                                          long long ld_feat, long long S,
                                                          const float* __restrict__ blob) {
    __shared__ __attribute__((aligned(16))) float xs[LP_T * LP_XLD];
    __shared__ __attribute__((aligned(16))) float sc[LP_T * LP_SLD];
    __shared__ float stats[LP_T * 2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* mats = blob;
    const float* vecs = blob + LP_MATS_TOTAL;
    const long long s0 = (long long)blockIdx.x * LP_QPB;

    // ---- stage the 64 x 3 offsets, zero-padded to K = 8, into sc[:, 0:8] ----
    if (tid < LP_T) {
        const long long seq = s0 + (tid >> 4);
        float x = 0.f, y = 0.f, z = 0.f;
        if (seq < S) {
            const float* p = offs + (seq * 16 + (tid & 15)) * 3;
            x = p[0]; y = p[1]; z = p[2];
        }
        float* d = sc + tid * LP_SLD;
        *reinterpret_cast<float4*>(d) = make_float4(x, y, z, 0.f);
        *reinterpret_cast<float4*>(d + 4) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    f32x16 acc2[2];
    lp_ring_t ring;
    // ---- Embedding (Attention.py:98-128): linear1 3->125, GELU -> xs ; linear2 125->125 -> sc ; || xyz ----
    lp_gemm8<2>(acc2, sc, LP_SLD, mats + lp_mat_off(0), wave, lane);
    lp_foreach<2>(acc2, wave, lane, [&](int row, int col, float v) { xs[row * LP_XLD + col] = lp_gelu(v + vecs[LP_VEC_EMB1 + col]); });
    __syncthreads();
    // xyz must survive in sc[:, 0:3] until the concat: linear2's output goes to sc[:, 64:192]
    lp_gemm128<2, true, false, 3>(acc2, xs, LP_XLD, mats + lp_mat_off(1), ring, mats + lp_mat_off(2), wave, lane);
    lp_foreach<2>(acc2, wave, lane, [&](int row, int col, float v) {
        sc[row * LP_SLD + 64 + col] = col < 125 ? v + vecs[LP_VEC_EMB2 + col] : sc[row * LP_SLD + (col - 125)];   // concat raw xyz
    });
    __syncthreads();

    const float* xsrc = sc + 64;                   // where the current (un-centred) x lives
    int xsrc_ld = LP_SLD;
#pragma unroll 1
    for (int e = 0; e < 2; ++e) {
        const float* em = mats + lp_mat_off(2 + 6 * e);
        const float* ev = vecs + LP_VEC_ENC0 + e * LP_VEC_ENC_STRIDE;
        // ---- norm1 (folded) + QKV (Attention.py:186-188, 287) ----
        lp_center(xsrc, xsrc_ld, xs, stats, tid);
        __syncthreads();
        {
            f32x16 acc3[3];
            lp_gemm128<3, true, true, 2>(acc3, xs, LP_XLD, em, ring, em + LP_MAT_QKV, wave, lane);
            lp_foreach<3>(acc3, wave, lane, [&](int row, int col, float v) { sc[row * LP_SLD + col] = fmaf(stats[2 * row + 1], v, ev[col]); });
        }
        __syncthreads();
        // ---- attention (Attention.py:8-36): thread = (query, head, row); output overwrites the head's V block ----
        {
            const float* base = sc + (tid >> 6) * 16 * LP_SLD;
            const int hh = (tid >> 4) & 3, qi = tid & 15;
            float q[8];
#pragma unroll
            for (int d = 0; d < 8; ++d) q[d] = base[qi * LP_SLD + hh * 8 + d];
            float p[16];
            float mx = -__builtin_inff();
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                float a = 0.f;
#pragma unroll
                for (int d = 0; d < 8; ++d) a = fmaf(q[d], base[j * LP_SLD + 32 + hh * 8 + d], a);
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
                const float* vrow = base + j * LP_SLD + 64 + hh * 32;
#pragma unroll
                for (int c = 0; c < 32; c += 4) {
                    const float4 vv = *reinterpret_cast<const float4*>(vrow + c);
                    o[c] = fmaf(pj, vv.x, o[c]); o[c + 1] = fmaf(pj, vv.y, o[c + 1]);
                    o[c + 2] = fmaf(pj, vv.z, o[c + 2]); o[c + 3] = fmaf(pj, vv.w, o[c + 3]);
                }
            }
            // the 16 threads sharing this V block are consecutive lanes of this wave: all reads precede the writes
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            float* orow = sc + ((tid >> 6) * 16 + qi) * LP_SLD + 64 + hh * 32;
#pragma unroll
            for (int c = 0; c < 32; c += 4) *reinterpret_cast<float4*>(orow + c) = make_float4(o[c], o[c + 1], o[c + 2], o[c + 3]);
        }
        __syncthreads();
        // ---- out projection + residual (Attention.py:201-202, 290): x = (x~ + mu) + att W_o^T + b ----
        lp_gemm128<2, true, true, 2>(acc2, sc + 64, LP_SLD, em + LP_MAT_QKV, ring, em + LP_MAT_QKV + LP_MAT_128, wave, lane);
        lp_foreach<2>(acc2, wave, lane, [&](int row, int col, float v) {
            float* px = xs + row * LP_XLD + col;
            *px = (*px + stats[2 * row]) + (v + ev[192 + col]);
        });
        __syncthreads();
        // ---- norm2 (folded) + FF 128 -> 256 (GELU) -> 128 + residual (Attention.py:293-298), two 128-wide halves ----
        lp_center(xs, LP_XLD, xs, stats, tid);
        __syncthreads();
        f32x16 accf[2];
        // product order: ff1a, ff2a, ff1b, ff2b; each requests the next one's first weight groups at its tail
        const float* w_ff1a = em + LP_MAT_QKV + LP_MAT_128 * 1;
        const float* w_ff1b = em + LP_MAT_QKV + LP_MAT_128 * 2;
        const float* w_ff2a = em + LP_MAT_QKV + LP_MAT_128 * 3;
        const float* w_ff2b = em + LP_MAT_QKV + LP_MAT_128 * 4;
        lp_gemm128<2, true, true, 2>(acc2, xs, LP_XLD, w_ff1a, ring, w_ff2a, wave, lane);
        lp_foreach<2>(acc2, wave, lane, [&](int row, int col, float v) {
            sc[row * LP_SLD + col] = lp_gelu(fmaf(stats[2 * row + 1], v, ev[192 + 128 + col]));
        });
        __syncthreads();
        lp_gemm128<2, true, true, 2>(accf, sc, LP_SLD, w_ff2a, ring, w_ff1b, wave, lane);
        __syncthreads();
        lp_gemm128<2, true, true, 2>(acc2, xs, LP_XLD, w_ff1b, ring, w_ff2b, wave, lane);
        lp_foreach<2>(acc2, wave, lane, [&](int row, int col, float v) {
            sc[row * LP_SLD + col] = lp_gelu(fmaf(stats[2 * row + 1], v, ev[192 + 128 + 128 + col]));
        });
        __syncthreads();
        if (e == 0) lp_gemm128<2, false, true, 3>(accf, sc, LP_SLD, w_ff2b, ring, mats + lp_mat_off(8), wave, lane);   // next: qkv of encoder 1
        else lp_gemm128<2, false, true, 2>(accf, sc, LP_SLD, w_ff2b, ring, mats + lp_mat_off(14), wave, lane);       // next: linear0
        __syncthreads();
        lp_foreach<2>(accf, wave, lane, [&](int row, int col, float v) {
            float* px = xs + row * LP_XLD + col;
            *px = (*px + stats[2 * row]) + (v + ev[192 + 128 + 256 + col]);
        });
        __syncthreads();
        xsrc = xs;
        xsrc_ld = LP_XLD;
    }
    // ---- final norm (folded) + linear0 128 -> 128 (SconeOcc.py:119-122) ----
    lp_center(xs, LP_XLD, xs, stats, tid);
    __syncthreads();
    lp_gemm128<2, true, true, 0>(acc2, xs, LP_XLD, mats + lp_mat_off(14), ring, nullptr, wave, lane);
    lp_foreach<2>(acc2, wave, lane, [&](int row, int col, float v) { sc[row * LP_SLD + col] = fmaf(stats[2 * row + 1], v, vecs[LP_VEC_LIN0 + col]); });
    __syncthreads();
    // ---- max || avg pool over the 16 tokens of each query (SconeOcc.py:124-126) ----
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int o = tid + r * 256, q = o >> 7, c = o & 127;
        if (s0 + q < S) {
            float mx = -__builtin_inff(), sm = 0.f;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const float v = sc[(q * 16 + j) * LP_SLD + c];
                mx = fmaxf(mx, v);
                sm += v;
            }
            feat[(s0 + q) * ld_feat + c] = mx;
            feat[(s0 + q) * ld_feat + 128 + c] = sm * (1.0f / 16.f);
        }
    }
}

void launch_local_pct(hipStream_t s, const float* offs, float* feat, int64_t ld_feat, int64_t S, const float* blob) {
    if (S <= 0) return;
    hipLaunchKernelGGL(local_pct_kernel, dim3((unsigned)cdiv(S, LP_QPB)), dim3(256), 0, s, offs, feat, (long long)ld_feat,
                       (long long)S, blob);
}


int local_pct_blob_floats() { return LP_BLOB_FLOATS; }

}  // namespace mcr
