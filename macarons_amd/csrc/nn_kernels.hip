// K4-K8 building blocks of the SCONE networks for gfx950: fp32-MFMA linear with fused epilogues, LayerNorm,
// attention (short-sequence and flash-style long-sequence), pooling / broadcast helpers.
//
// Reference semantics (macarons/networks/Attention.py): nn.Linear, nn.GELU() (erf form), nn.LayerNorm (eps
// 1e-5), attention() :8-36 (softmax(QK^T / sqrt(d)) V; mask=None in every call site of the hot path).
// All arithmetic is fp32; matrix products use v_mfma_f32_32x32x2_f32, which is bit-for-bit an fp32 fma chain
// (no TF32/bf16 anywhere), so results differ from PyTorch only by summation order.
#include "nn_kernels.h"
#include <cstdlib>

namespace mcr {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)); }

// =====================================================================================================
// linear:  block = 4 waves = 128 rows x (NT*32) columns; K streamed through LDS in chunks of 32.
//   MFMA 32x32x2: lane l feeds A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31].  Inside a K-chunk the two
//   lane halves take k = h*16 + s (s = 0..15), so every lane reads 16 consecutive floats of "its" row with
//   four ds_read_b128 (row stride 36 floats keeps the 16-lane read groups on distinct banks).
// =====================================================================================================
constexpr int LIN_BM = 128, LIN_BK = 32, LIN_LD = LIN_BK + 4;

// NT = 8 holds 128 accumulator registers: capping it at 256 total keeps two waves per SIMD resident (measured
// 1.84 -> 1.57 ms on the 1344 -> 512 head GEMM); pipelining the LDS fragment reads on top of that spills.
template <int NT>
__global__ __launch_bounds__(256, NT == 8 ? 2 : 1) void linear_kernel(const float* __restrict__ X, long long ldx,
                                                     const float* __restrict__ W, long long ldw,
                                                     const float* __restrict__ bias, const float* __restrict__ row_bias,
                                                     long long rows_per_group, const float* __restrict__ R, long long ldr,
                                                     float* __restrict__ Y, long long ldy, long long M, int N, int K,
                                                     int act, int vec_x, int vec_w) {
    __shared__ __attribute__((aligned(16))) float As[LIN_BM * LIN_LD];
    __shared__ __attribute__((aligned(16))) float Bs[NT * 32 * LIN_LD];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const long long m0 = (long long)blockIdx.x * LIN_BM;
    const int n0 = blockIdx.y * NT * 32;
    const int i = lane & 31, h = lane >> 5;

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // Staging registers: chunk k+1 is fetched from global memory while chunk k is in the MFMA phase
    // (issue-early / write-late split), so HBM/L2 latency hides behind the matrix pipe.
    float4 ra[4], rb[NT];
    auto fetch = [&](int k0) {
        if (vec_x) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int idx = tid + r * 256, row = idx >> 3, c4 = (idx & 7) * 4;
                ra[r] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (m0 + row < M && k0 + c4 < K) ra[r] = *reinterpret_cast<const float4*>(X + (m0 + row) * ldx + k0 + c4);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int idx = tid + r * 256, row = idx >> 3, c4 = (idx & 7) * 4;
                const float* px = X + (m0 + row) * ldx + k0 + c4;
                const bool rv = m0 + row < M;
                ra[r].x = rv && k0 + c4 + 0 < K ? px[0] : 0.f;
                ra[r].y = rv && k0 + c4 + 1 < K ? px[1] : 0.f;
                ra[r].z = rv && k0 + c4 + 2 < K ? px[2] : 0.f;
                ra[r].w = rv && k0 + c4 + 3 < K ? px[3] : 0.f;
            }
        }
        if (vec_w) {
#pragma unroll
            for (int r = 0; r < NT; ++r) {
                const int idx = tid + r * 256, row = idx >> 3, c4 = (idx & 7) * 4;
                rb[r] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (n0 + row < N && k0 + c4 < K) rb[r] = *reinterpret_cast<const float4*>(W + (long long)(n0 + row) * ldw + k0 + c4);
            }
        } else {
#pragma unroll
            for (int r = 0; r < NT; ++r) {
                const int idx = tid + r * 256, row = idx >> 3, c4 = (idx & 7) * 4;
                const float* pw = W + (long long)(n0 + row) * ldw + k0 + c4;
                const bool rv = n0 + row < N;
                rb[r].x = rv && k0 + c4 + 0 < K ? pw[0] : 0.f;
                rb[r].y = rv && k0 + c4 + 1 < K ? pw[1] : 0.f;
                rb[r].z = rv && k0 + c4 + 2 < K ? pw[2] : 0.f;
                rb[r].w = rv && k0 + c4 + 3 < K ? pw[3] : 0.f;
            }
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < K; k0 += LIN_BK) {
        // ---- write the staged chunk to LDS (A: 128 x 32, B: NT*32 x 32, zero-padded) ----
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int idx = tid + r * 256, row = idx >> 3, c4 = (idx & 7) * 4;
            *reinterpret_cast<float4*>(&As[row * LIN_LD + c4]) = ra[r];
        }
#pragma unroll
        for (int r = 0; r < NT; ++r) {
            const int idx = tid + r * 256, row = idx >> 3, c4 = (idx & 7) * 4;
            *reinterpret_cast<float4*>(&Bs[row * LIN_LD + c4]) = rb[r];
        }
        __syncthreads();
        if (k0 + LIN_BK < K) fetch(k0 + LIN_BK);       // in flight during the MFMA phase
        // ---- MFMA over the chunk ----
        const float* a_row = &As[(wave * 32 + i) * LIN_LD + h * 16];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const float4 a4 = *reinterpret_cast<const float4*>(a_row + s4 * 4);
            float4 b4[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) b4[t] = *reinterpret_cast<const float4*>(&Bs[(t * 32 + i) * LIN_LD + h * 16 + s4 * 4]);
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4[t].x, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4[t].y, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4[t].z, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4[t].w, acc[t], 0, 0, 0);
        }
        __syncthreads();
    }
    // ---- epilogue: C/D layout col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) ----
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int n = n0 + t * 32 + i;
        if (n >= N) continue;
        const float bn = bias ? bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long long m = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (m >= M) continue;
            float y = acc[t][r] + bn;
            if (row_bias) y += row_bias[(m / rows_per_group) * N + n];
            if (act == ACT_GELU) y = gelu_erf(y);
            if (R) y += R[m * ldr + n];
            Y[m * ldy + n] = y;
        }
    }
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

void launch_linear(hipStream_t s, const float* X, int64_t ldx, const float* W, const float* bias, const float* R,
                   int64_t ldr, float* Y, int64_t ldy, int64_t M, int N, int K, int act, const float* row_bias,
                   int64_t rows_per_group, int64_t ldw) {
    if (M <= 0 || N <= 0) return;
    if (ldw == 0) ldw = K;
    static const bool use_split = []() { const char* e = getenv("MCR_LINEAR3"); return !(e && e[0] == '0'); }();   // dev A/B knob
    if (use_split && linear3_applicable(X, ldx, W, ldw, M, N, K)) {
        launch_linear3(s, X, ldx, W, bias, R, ldr, Y, ldy, M, N, K, act, row_bias, rows_per_group, ldw);
        return;
    }
    const int vec_x = (K % 4 == 0) && (ldx % 4 == 0) && aligned16(X);
    const int vec_w = (K % 4 == 0) && (ldw % 4 == 0) && aligned16(W);
    const long long mb = cdiv(M, LIN_BM);
    // widest column tile (X is then read once), but never wider than N, and narrower when the problem is too
    // small to give every CU a few blocks otherwise
    int nt = 8;
    while (nt > 1 && (nt / 2) * 32 >= N) nt >>= 1;
    while (nt > 1 && mb * cdiv(N, nt * 32) < 512) nt >>= 1;
    dim3 grid((unsigned)mb, (unsigned)cdiv(N, nt * 32));
#define MCR_LIN(NT)                                                                                                   \
    hipLaunchKernelGGL((linear_kernel<NT>), grid, dim3(256), 0, s, X, (long long)ldx, W, (long long)ldw, bias, row_bias, \
                       (long long)(rows_per_group > 0 ? rows_per_group : 1), R, (long long)ldr, Y, (long long)ldy,       \
                       (long long)M, N, K, act, vec_x, vec_w)
    switch (nt) {
        case 8: MCR_LIN(8); break;
        case 4: MCR_LIN(4); break;
        case 2: MCR_LIN(2); break;
        default: MCR_LIN(1); break;
    }
#undef MCR_LIN
}

// =====================================================================================================
// LayerNorm: one wave per row (E <= 512), two-pass mean / variance like PyTorch.
// =====================================================================================================
template <int EPL>   // elements per lane = ceil(E / 64)
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ X, long long ldx,
                                                        const float* __restrict__ g, const float* __restrict__ b,
                                                        float* __restrict__ Y, long long ldy, long long M, int E) {
    const long long m = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const int lane = threadIdx.x & 63;
    float v[EPL];
    float sum = 0.f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const int c = lane + e * 64;
        v[e] = c < E ? X[m * ldx + c] : 0.f;
        sum += v[e];
    }
    const float mean = wave_sum_all(sum) / (float)E;
    float sq = 0.f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const int c = lane + e * 64;
        const float d = c < E ? v[e] - mean : 0.f;
        sq = fmaf(d, d, sq);
    }
    const float var = wave_sum_all(sq) / (float)E;
    const float rstd = 1.0f / sqrtf(var + 1e-5f);
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const int c = lane + e * 64;
        if (c < E) Y[m * ldy + c] = (v[e] - mean) * rstd * g[c] + b[c];
    }
}

void launch_layernorm(hipStream_t s, const float* X, int64_t ldx, const float* g, const float* b, float* Y, int64_t ldy,
                      int64_t M, int E) {
    if (M <= 0) return;
    dim3 grid((unsigned)cdiv(M, 4));
    const int epl = (E + 63) / 64;
#define MCR_LN(EPL) \
    hipLaunchKernelGGL((layernorm_kernel<EPL>), grid, dim3(256), 0, s, X, (long long)ldx, g, b, Y, (long long)ldy, (long long)M, E)
    if (epl <= 2) MCR_LN(2);
    else if (epl <= 4) MCR_LN(4);
    else MCR_LN(8);
#undef MCR_LN
}

// =====================================================================================================
// Attention, short sequences (L <= 16 tokens: the k=16 neighbourhoods of SconeOcc).  One thread per
// (sequence, head, query row); a block stages SPB sequences' packed QKV rows in LDS.
// =====================================================================================================
template <int L, int H, int DQ, int DV, int SPB>
__global__ __launch_bounds__(SPB* H* L) void attention_small_kernel(const float* __restrict__ qkv, long long ldq,
                                                                   float* __restrict__ out, long long ldo, long long S) {
    constexpr int W = 2 * H * DQ + H * DV;          // packed row width
    constexpr int WP = W + 1;                       // +1: rows land on different banks
    __shared__ float s_qkv[SPB * L * WP];
    const long long s0 = (long long)blockIdx.x * SPB;
    for (int idx = threadIdx.x; idx < SPB * L * W; idx += SPB * H * L) {
        const int r = idx / W, c = idx - r * W;
        const long long row = s0 * L + r;
        s_qkv[r * WP + c] = row < S * L ? qkv[row * ldq + c] : 0.f;
    }
    __syncthreads();
    const int sl = threadIdx.x / (H * L);           // sequence within block
    const int hh = (threadIdx.x / L) % H;
    const int qi = threadIdx.x % L;
    if (s0 + sl >= S) return;
    const float* base = &s_qkv[sl * L * WP];
    float q[DQ];
#pragma unroll
    for (int d = 0; d < DQ; ++d) q[d] = base[qi * WP + hh * DQ + d];
    float sc[L];
    float mx = -__builtin_inff();
    const float scale = 1.0f / sqrtf((float)DQ);
#pragma unroll
    for (int j = 0; j < L; ++j) {
        float a = 0.f;
#pragma unroll
        for (int d = 0; d < DQ; ++d) a = fmaf(q[d], base[j * WP + H * DQ + hh * DQ + d], a);
        sc[j] = a * scale;
        mx = fmaxf(mx, sc[j]);
    }
    float den = 0.f;
#pragma unroll
    for (int j = 0; j < L; ++j) {
        sc[j] = __expf(sc[j] - mx);
        den += sc[j];
    }
    const float inv = 1.0f / den;
    float o[DV];
#pragma unroll
    for (int c = 0; c < DV; ++c) o[c] = 0.f;
#pragma unroll
    for (int j = 0; j < L; ++j) {
        const float p = sc[j] * inv;
#pragma unroll
        for (int c = 0; c < DV; ++c) o[c] = fmaf(p, base[j * WP + 2 * H * DQ + hh * DV + c], o[c]);
    }
    float* orow = out + ((s0 + sl) * L + qi) * ldo + hh * DV;
#pragma unroll
    for (int c = 0; c < DV; ++c) orow[c] = o[c];
}

// =====================================================================================================
// Attention, long sequences (flash-style, online softmax).  Block = 64 query rows x KS key-splits (one wave per
// split); K/V tiles are staged in LDS and broadcast-read; the KS partial (m, l, o) states are merged through
// LDS.  grid = (ceil(L/64), H, S).
// =====================================================================================================
template <int DQ, int DV, int KS>
__global__ __launch_bounds__(64 * KS) void attention_flash_kernel(const float* __restrict__ qkv, long long ldq,
                                                                  float* __restrict__ out, long long ldo, int L, int H) {
    constexpr int TK = 128;                          // keys per tile
    constexpr int KW = DQ + DV;
    __shared__ __attribute__((aligned(16))) float s_kv[TK * KW > KS * 64 * (DV + 2) ? TK * KW : KS * 64 * (DV + 2)];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int hh = blockIdx.y;
    const long long seq0 = (long long)blockIdx.z * L;
    const int qi = blockIdx.x * 64 + lane;
    const bool valid = qi < L;
    const int koff = H * DQ + hh * DQ, voff = 2 * H * DQ + hh * DV;

    float q[DQ];
    const float scale = 1.0f / sqrtf((float)DQ);
    {
        const float* qp = qkv + (seq0 + (valid ? qi : 0)) * ldq + hh * DQ;
#pragma unroll
        for (int d = 0; d < DQ; ++d) q[d] = qp[d] * scale;       // fold 1/sqrt(d) into q
    }
    float m = -__builtin_inff(), l = 0.f;
    float o[DV];
#pragma unroll
    for (int c = 0; c < DV; ++c) o[c] = 0.f;

    for (int t0 = 0; t0 < L; t0 += TK) {
        const int nt = min(TK, L - t0);
        __syncthreads();
        for (int idx = threadIdx.x; idx < nt * KW; idx += 64 * KS) {
            const int r = idx / KW, c = idx - r * KW;
            const float* rowp = qkv + (seq0 + t0 + r) * ldq;
            s_kv[idx] = c < DQ ? rowp[koff + c] : rowp[voff + (c - DQ)];
        }
        __syncthreads();
        // this wave's keys of the tile: j = wave, wave+KS, ...  in groups of 8
        for (int j0 = wave; j0 < nt; j0 += 8 * KS) {
            float sc[8];
            float cmax = m;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int j = j0 + u * KS;
                float a = -__builtin_inff();
                if (j < nt) {
                    a = 0.f;
#pragma unroll
                    for (int d = 0; d < DQ; ++d) a = fmaf(q[d], s_kv[j * KW + d], a);
                }
                sc[u] = a;
                cmax = fmaxf(cmax, a);
            }
            const float alpha = __expf(m - cmax);        // m = -inf on the first group -> alpha = 0, o = l = 0 anyway
            m = cmax;
            l *= alpha;
#pragma unroll
            for (int c = 0; c < DV; ++c) o[c] *= alpha;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int j = j0 + u * KS;
                if (j < nt) {
                    const float p = __expf(sc[u] - m);
                    l += p;
#pragma unroll
                    for (int c = 0; c < DV; ++c) o[c] = fmaf(p, s_kv[j * KW + DQ + c], o[c]);
                }
            }
        }
    }
    // ---- merge the KS partial states ----
    __syncthreads();
    float* mg = s_kv;                                 // [KS][64][DV + 2]
    {
        float* p = mg + (wave * 64 + lane) * (DV + 2);
        p[0] = m; p[1] = l;
#pragma unroll
        for (int c = 0; c < DV; ++c) p[2 + c] = o[c];
    }
    __syncthreads();
    if (wave != 0 || !valid) return;
    float M = -__builtin_inff();
#pragma unroll
    for (int w = 0; w < KS; ++w) M = fmaxf(M, mg[(w * 64 + lane) * (DV + 2)]);
    float Lsum = 0.f;
#pragma unroll
    for (int c = 0; c < DV; ++c) o[c] = 0.f;
#pragma unroll
    for (int w = 0; w < KS; ++w) {
        const float* p = mg + (w * 64 + lane) * (DV + 2);
        const float f = p[1] > 0.f ? __expf(p[0] - M) : 0.f;
        Lsum = fmaf(p[1], f, Lsum);
#pragma unroll
        for (int c = 0; c < DV; ++c) o[c] = fmaf(p[2 + c], f, o[c]);
    }
    const float inv = 1.0f / Lsum;
    float* orow = out + (seq0 + qi) * ldo + hh * DV;
#pragma unroll
    for (int c = 0; c < DV; ++c) orow[c] = o[c] * inv;
}

void launch_attention(hipStream_t s, const float* qkv, int64_t ldq, float* out, int64_t ldo, int64_t S, int L, int H,
                      int DQK, int DV) {
    if (S <= 0 || L <= 0) return;
    const int dq = DQK / H, dv = DV / H;
    if (L == 16 && H == 4 && dq == 8 && dv == 32) {
        constexpr int SPB = 4;
        hipLaunchKernelGGL((attention_small_kernel<16, 4, 8, 32, SPB>), dim3((unsigned)cdiv(S, SPB)), dim3(SPB * 4 * 16), 0,
                           s, qkv, (long long)ldq, out, (long long)ldo, (long long)S);
        return;
    }
    constexpr int KS = 4;
    dim3 grid((unsigned)cdiv(L, 64), (unsigned)H, (unsigned)S);
    if (dq == 8 && dv == 32)
        hipLaunchKernelGGL((attention_flash_kernel<8, 32, KS>), grid, dim3(64 * KS), 0, s, qkv, (long long)ldq, out,
                           (long long)ldo, L, H);
    else if (dq == 16 && dv == 64)
        hipLaunchKernelGGL((attention_flash_kernel<16, 64, KS>), grid, dim3(64 * KS), 0, s, qkv, (long long)ldq, out,
                           (long long)ldo, L, H);
    else
        set_error("launch_attention: unsupported head dims dq=%d dv=%d", dq, dv);
}

// =====================================================================================================
// pooling / broadcast / copy helpers
// =====================================================================================================
// grid = (S, ceil(E/64)); block = 64 columns x 4 row lanes
template <bool BROADCAST>
__global__ __launch_bounds__(256) void pool_kernel(const float* __restrict__ X, long long ldx, float* __restrict__ Y,
                                                   long long ldy, int L, int E) {
    __shared__ float s_max[4][64];
    __shared__ float s_sum[4][64];
    const int cl = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int c = blockIdx.y * 64 + cl;
    const long long row0 = (long long)blockIdx.x * L;
    float mx = -__builtin_inff(), sm = 0.f;
    if (c < E)
        for (int r = g; r < L; r += 4) {
            const float v = X[(row0 + r) * ldx + c];
            mx = fmaxf(mx, v);
            sm += v;
        }
    s_max[g][cl] = mx;
    s_sum[g][cl] = sm;
    __syncthreads();
    mx = fmaxf(fmaxf(s_max[0][cl], s_max[1][cl]), fmaxf(s_max[2][cl], s_max[3][cl]));
    sm = (s_sum[0][cl] + s_sum[1][cl]) + (s_sum[2][cl] + s_sum[3][cl]);
    if (c >= E) return;
    if (BROADCAST) {
        for (int r = g; r < L; r += 4) Y[(row0 + r) * ldy + c] = mx;
    } else if (g == 0) {
        Y[(long long)blockIdx.x * ldy + c] = mx;
        Y[(long long)blockIdx.x * ldy + E + c] = sm / (float)L;
    }
}

void launch_colmax_broadcast(hipStream_t s, const float* X, int64_t ldx, float* Y, int64_t ldy, int64_t S, int L, int E) {
    if (S <= 0) return;
    hipLaunchKernelGGL((pool_kernel<true>), dim3((unsigned)S, (unsigned)cdiv(E, 64)), dim3(256), 0, s, X, (long long)ldx, Y,
                       (long long)ldy, L, E);
}

void launch_pool_max_avg(hipStream_t s, const float* X, int64_t ldx, float* Y, int64_t ldy, int64_t S, int L, int E) {
    if (S <= 0) return;
    hipLaunchKernelGGL((pool_kernel<false>), dim3((unsigned)S, (unsigned)cdiv(E, 64)), dim3(256), 0, s, X, (long long)ldx, Y,
                       (long long)ldy, L, E);
}

__global__ void copy2d_kernel(const float* __restrict__ X, long long ldx, float* __restrict__ Y, long long ldy, long long M,
                              int E) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= M * E) return;
    const long long m = idx / E;
    const int c = (int)(idx - m * E);
    Y[m * ldy + c] = X[m * ldx + c];
}

void launch_copy2d(hipStream_t s, const float* X, int64_t ldx, float* Y, int64_t ldy, int64_t M, int E) {
    if (M <= 0 || E <= 0) return;
    hipLaunchKernelGGL(copy2d_kernel, dim3((unsigned)cdiv(M * E, 256)), dim3(256), 0, s, X, (long long)ldx, Y, (long long)ldy,
                       (long long)M, E);
}

}  // namespace mcr
