// K5-local v4 — split-precision fused local PCTransformer, two workgroups per CU.
//
// Same mathematics and parameter blob as local_pct3.hip (exact bf16 hi/mid/lo split, six MFMAs per fp32 product;
// reference mapping in local_pct.hip: SconeOcc.py:104-130).  v3 runs one 4-wave workgroup per CU = one wave per SIMD,
// and its counters show the three phases of a wave strictly serialised: matrix pipe busy 30 %, vector ALU (operand
// splitting, epilogues, attention, LayerNorm) 33 %, waits 33 %.  Nothing can overlap them inside one in-order wave,
// so v4 makes TWO workgroups resident per CU (two waves per SIMD: one in its MFMA phase while the other is in a
// VALU/LDS phase).  That needs <= 80 KB of LDS and <= 256 registers per lane:
//   * the residual stream x lives in REGISTERS in MFMA C-fragment layout (2 x 16 per lane: exactly the 64 x 128
//     tile over 256 lanes), so LDS holds only the normalised operand and the scratch:
//       xs [64][132]  x^ = (x - mu) * rstd  (A operand of qkv / ff1 / linear0), or q|k during attention
//       sc [64][132]  embedding input / v and the attention output (in place) / one 128-wide half of the FF hidden
//     = 67.6 KB.  LayerNorm becomes "store raw x, normalise the row in place"; gamma/beta stay folded into the next
//     weights (W' = W gamma, c = b + W beta) so every epilogue is y = acc + c and no per-row statistics are kept.
//   * q and k go to xs (free once the qkv product has consumed x^), v to sc.
#include "lp_split.h"

namespace mcr {

constexpr int L4_LD = 132;        // row stride of both LDS tiles (floats): 4 mod 64 banks -> conflict-free b128 rows

// ---- block GEMM, "one n-tile column per wave" mapping ---------------------------------------------------------------
// Wave w owns the n-tiles {w, w + 4, ...} (NTW of them) for BOTH 32-row m-tiles.  No two waves of a workgroup request
// the same weight line: with the v3 mapping (pairs of waves sharing their n-tiles) the second request of every line
// hit the vector L1 while the first was still in flight, and those "pending" hits stalled the L1 for 45 % of the
// kernel (TCP_PENDING_STALL_CYCLES, profiles/).  Price: every wave splits the A rows of both m-tiles.
#ifndef L4_PF_N
#define L4_PF_N 3
#endif
constexpr int L4_PF = L4_PF_N;                    // k16-steps of weights in flight per wave

template <int S, int NTW, bool INIT>
__device__ __forceinline__ void l4_gemm(f32x16 (&acc)[NTW][2], const float* __restrict__ A, const float* __restrict__ Wp,
                                        int nt0, int lane) {
    const int i = lane & 31, h = lane >> 5;
    const uint4* bp[NTW];
#pragma unroll
    for (int u = 0; u < NTW; ++u) {
        bp[u] = reinterpret_cast<const uint4*>(Wp) + (size_t)(nt0 + 4 * u) * S * 3 * 64 + lane;
        if (INIT) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[u][mt][r] = 0.f;
        }
    }
    constexpr int PF = S < L4_PF ? S : L4_PF;
    uint4 b[PF][NTW][3];
#pragma unroll
    for (int p = 0; p < PF; ++p)
#pragma unroll
        for (int u = 0; u < NTW; ++u)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) b[p][u][pl] = bp[u][(p * 3 + pl) * 64];
    const float* a0 = A + i * L4_LD + 8 * h;
    float4 ra[2][2];                                        // raw A rows of the next step, both m-tiles
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        ra[mt][0] = *reinterpret_cast<const float4*>(a0 + mt * 32 * L4_LD);
        ra[mt][1] = *reinterpret_cast<const float4*>(a0 + mt * 32 * L4_LD + 4);
    }
#pragma unroll
    for (int s = 0; s < S; ++s) {
        Split3 sa[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) sa[mt] = split8(ra[mt][0], ra[mt][1]);
        if (s + 1 < S) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                ra[mt][0] = *reinterpret_cast<const float4*>(a0 + mt * 32 * L4_LD + 16 * (s + 1));
                ra[mt][1] = *reinterpret_cast<const float4*>(a0 + mt * 32 * L4_LD + 16 * (s + 1) + 4);
            }
        }
        uint4 bc[NTW][3];
#pragma unroll
        for (int u = 0; u < NTW; ++u)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) bc[u][pl] = b[s % PF][u][pl];
        if (s + PF < S) {
#pragma unroll
            for (int u = 0; u < NTW; ++u)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) b[s % PF][u][pl] = bp[u][((s + PF) * 3 + pl) * 64];
        }
#pragma unroll
        for (int u = 0; u < NTW; ++u)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                acc[u][mt] = mfma_bf(sa[mt].lo, bc[u][0], acc[u][mt]);      // smallest terms first
                acc[u][mt] = mfma_bf(sa[mt].hi, bc[u][2], acc[u][mt]);
                acc[u][mt] = mfma_bf(sa[mt].mid, bc[u][1], acc[u][mt]);
                acc[u][mt] = mfma_bf(sa[mt].mid, bc[u][0], acc[u][mt]);
                acc[u][mt] = mfma_bf(sa[mt].hi, bc[u][1], acc[u][mt]);
                acc[u][mt] = mfma_bf(sa[mt].hi, bc[u][0], acc[u][mt]);
            }
    }
}

// f(row, col, value) over this wave's C fragments
template <int NTW, class F>
__device__ __forceinline__ void l4_foreach(f32x16 (&acc)[NTW][2], int nt0, int lane, F f) {
    const int j = lane & 31, h = lane >> 5;
#pragma unroll
    for (int u = 0; u < NTW; ++u)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) f(mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, (nt0 + 4 * u) * 32 + j, (float)acc[u][mt][r]);
}

// xres (this wave's n-tile column, both m-tiles) -> xs, raw
__device__ __forceinline__ void l4_store_x(f32x16 (&xres)[1][2], float* xs, int wave, int lane) {
    l4_foreach<1>(xres, wave, lane, [&](int row, int col, float v) { xs[row * L4_LD + col] = v; });
}

// LayerNorm (eps 1e-5, affine folded away) of the 64 rows of xs in place: 4 threads per row, 32 columns each.
// Two-pass (mean, then centred variance) like torch's.  A row is touched only by its own 4 threads.
__device__ __forceinline__ void l4_norm(float* xs, int tid) {
    const int row = tid >> 2, part = tid & 3;
    float* s = xs + row * L4_LD + part * 32;
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
    const float rstd = 1.0f / sqrtf(sq * (1.0f / 128.f) + 1e-5f);
#pragma unroll
    for (int c = 0; c < 32; c += 4)
        *reinterpret_cast<float4*>(s + c) = make_float4(v[c] * rstd, v[c + 1] * rstd, v[c + 2] * rstd, v[c + 3] * rstd);
}

// grid = ceil(S / 4); S sequences of 16 offsets [S,16,3]; features[s*ld_feat + 0:256] = max(128) || avg(128)
__global__ __launch_bounds__(256, 2) void local_pct4_kernel(const float* __restrict__ offs, float* __restrict__ feat,
                                                          long long ld_feat, long long S,
                                                          const float* __restrict__ blob) {
    __shared__ __attribute__((aligned(16))) float xs[L3_T * L4_LD];
    __shared__ __attribute__((aligned(16))) float sc[L3_T * L4_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef L4_PRIO             // experiment: break the lock-step of the two co-resident waves of a SIMD by issue priority
    if (__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | ((4 - 1) << 11)) & 1) __builtin_amdgcn_s_setprio(3);
#endif
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
        float* d = sc + tid * L4_LD;
        *reinterpret_cast<float4*>(d) = make_float4(x, y, z, 0.f);
        *reinterpret_cast<float4*>(d + 4) = make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(d + 8) = make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(d + 12) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    f32x16 acc[1][2], xres[1][2];
    // ---- Embedding (Attention.py:98-128): linear1 3->125, GELU -> xs ; linear2 125->125 || xyz -> xres ----
    l4_gemm<1, 1, true>(acc, sc, mats + l3_mat_off(0), wave, lane);
    l4_foreach<1>(acc, wave, lane, [&](int row, int col, float v) { xs[row * L4_LD + col] = l3_gelu(v + vecs[L3_VEC_EMB1 + col]); });
    __syncthreads();
    l4_gemm<8, 1, true>(xres, xs, mats + l3_mat_off(1), wave, lane);
    {
        const int j = lane & 31, h = lane >> 5, col = wave * 32 + j;
        const float b = col < 125 ? vecs[L3_VEC_EMB2 + col] : 0.f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                xres[0][mt][r] = col < 125 ? xres[0][mt][r] + b : sc[row * L4_LD + (col - 125)];        // concat raw xyz
            }
    }
    __syncthreads();                               // every wave is done reading xs as the A operand
    l4_store_x(xres, xs, wave, lane);

#pragma unroll 1
    for (int e = 0; e < 2; ++e) {
        const float* em = mats + l3_mat_off(2 + 6 * e);
        const float* ev = vecs + L3_VEC_ENC0 + e * L3_VEC_ENC_STRIDE;
        // ---- norm1 (folded) + QKV (Attention.py:186-188, 287): q|k -> xs[:, 0:64], v -> sc ----
        // 6 n-tiles over 4 waves: waves 0,1 take two (w, w + 4), waves 2,3 one.
        __syncthreads();
        l4_norm(xs, tid);
        __syncthreads();
        {
            f32x16 aq[2][2];
            auto put = [&](int row, int col, float v) {
                const float y = v + ev[col];
                if (col < 64) xs[row * L4_LD + col] = y;
                else sc[row * L4_LD + (col - 64)] = y;
            };
            if (wave < 2) l4_gemm<8, 2, true>(aq, xs, em, wave, lane);
            else l4_gemm<8, 1, true>(reinterpret_cast<f32x16 (&)[1][2]>(aq), xs, em, wave, lane);
            __syncthreads();                       // x^ consumed: xs may take q|k
            if (wave < 2) l4_foreach<2>(aq, wave, lane, put);
            else l4_foreach<1>(reinterpret_cast<f32x16 (&)[1][2]>(aq), wave, lane, put);
        }
        __syncthreads();
        // ---- attention (Attention.py:8-36): thread = (query, head, row); output overwrites the head's V block ----
        {
            const float* qk = xs + (tid >> 6) * 16 * L4_LD;
            const float* vb = sc + (tid >> 6) * 16 * L4_LD;
            const int hh = (tid >> 4) & 3, qi = tid & 15;
            float q[8];
#pragma unroll
            for (int d = 0; d < 8; ++d) q[d] = qk[qi * L4_LD + hh * 8 + d];
            float p[16];
            float mx = -__builtin_inff();
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                float a = 0.f;
#pragma unroll
                for (int d = 0; d < 8; ++d) a = fmaf(q[d], qk[j * L4_LD + 32 + hh * 8 + d], a);
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
                const float* vrow = vb + j * L4_LD + hh * 32;
#pragma unroll
                for (int c = 0; c < 32; c += 4) {
                    const float4 vv = *reinterpret_cast<const float4*>(vrow + c);
                    o[c] = fmaf(pj, vv.x, o[c]); o[c + 1] = fmaf(pj, vv.y, o[c + 1]);
                    o[c + 2] = fmaf(pj, vv.z, o[c + 2]); o[c + 3] = fmaf(pj, vv.w, o[c + 3]);
                }
            }
            // the 16 threads sharing this V block are consecutive lanes of this wave: all reads precede the writes
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            float* orow = sc + ((tid >> 6) * 16 + qi) * L4_LD + hh * 32;
#pragma unroll
            for (int c = 0; c < 32; c += 4) *reinterpret_cast<float4*>(orow + c) = make_float4(o[c], o[c + 1], o[c + 2], o[c + 3]);
        }
        __syncthreads();
        // ---- out projection + residual (Attention.py:201-202, 290): x += att W_o^T + b ----
        l4_gemm<8, 1, true>(acc, sc, em + L3_MAT_QKV, wave, lane);
        {
            const float b = ev[192 + wave * 32 + (lane & 31)];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) xres[0][mt][r] += acc[0][mt][r] + b;
        }
        l4_store_x(xres, xs, wave, lane);          // q|k were consumed before the barrier above
        __syncthreads();
        // ---- norm2 (folded) + FF 128 -> 256 (GELU) -> 128 + residual (Attention.py:293-298), two 128-wide halves ----
        l4_norm(xs, tid);
        __syncthreads();
        f32x16 accf[1][2];
        // product order: ff1a, ff2a, ff1b, ff2b
        const float* w_ff1a = em + L3_MAT_QKV + L3_MAT_128 * 1;
        const float* w_ff1b = em + L3_MAT_QKV + L3_MAT_128 * 2;
        const float* w_ff2a = em + L3_MAT_QKV + L3_MAT_128 * 3;
        const float* w_ff2b = em + L3_MAT_QKV + L3_MAT_128 * 4;
        l4_gemm<8, 1, true>(acc, xs, w_ff1a, wave, lane);
        l4_foreach<1>(acc, wave, lane, [&](int row, int col, float v) { sc[row * L4_LD + col] = l3_gelu(v + ev[192 + 128 + col]); });
        __syncthreads();
        l4_gemm<8, 1, true>(accf, sc, w_ff2a, wave, lane);
        __syncthreads();
        l4_gemm<8, 1, true>(acc, xs, w_ff1b, wave, lane);
        l4_foreach<1>(acc, wave, lane, [&](int row, int col, float v) { sc[row * L4_LD + col] = l3_gelu(v + ev[192 + 128 + 128 + col]); });
        __syncthreads();
        l4_gemm<8, 1, false>(accf, sc, w_ff2b, wave, lane);
        {
            const float b = ev[192 + 128 + 256 + wave * 32 + (lane & 31)];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) xres[0][mt][r] += accf[0][mt][r] + b;
        }
        l4_store_x(xres, xs, wave, lane);          // x^ was last read by ff1b, two barriers ago
    }
    // ---- final norm (folded) + linear0 128 -> 128 (SconeOcc.py:119-122) ----
    __syncthreads();
    l4_norm(xs, tid);
    __syncthreads();
    l4_gemm<8, 1, true>(acc, xs, mats + l3_mat_off(14), wave, lane);
    l4_foreach<1>(acc, wave, lane, [&](int row, int col, float v) { sc[row * L4_LD + col] = v + vecs[L3_VEC_LIN0 + col]; });
    __syncthreads();
    // ---- max || avg pool over the 16 tokens of each query (SconeOcc.py:124-126) ----
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int o = tid + r * 256, q = o >> 7, c = o & 127;
        if (s0 + q < S) {
            float mx = -__builtin_inff(), sm = 0.f;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const float v = sc[(q * 16 + j) * L4_LD + c];
                mx = fmaxf(mx, v);
                sm += v;
            }
            feat[(s0 + q) * ld_feat + c] = mx;
            feat[(s0 + q) * ld_feat + 128 + c] = sm * (1.0f / 16.f);
        }
    }
}

void launch_local_pct4(hipStream_t s, const float* offs, float* feat, int64_t ld_feat, int64_t S, const float* blob) {
    if (S <= 0) return;
    hipLaunchKernelGGL(local_pct4_kernel, dim3((unsigned)cdiv(S, L3_QPB)), dim3(256), 0, s, offs, feat, (long long)ld_feat,
                       (long long)S, blob);
}

}  // namespace mcr
