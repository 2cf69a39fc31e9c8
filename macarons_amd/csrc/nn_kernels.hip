// K4-K8 building blocks of the SCONE networks for gfx950: fp32-MFMA linear with fused epilogues, LayerNorm,
// attention (short-sequence and flash-style long-sequence), pooling / broadcast helpers.
//
// Reference semantics (macarons/networks/Attention.py): nn.Linear, nn.GELU() (erf form), nn.LayerNorm (eps
// 1e-5), attention() :8-36 (softmax(QK^T / sqrt(d)) V; mask=None in every call site of the hot path).
// All arithmetic is fp32; matrix products use v_mfma_f32_32x32x2_f32, which is bit-for-bit an fp32 fma chain
// (no TF32/bf16 anywhere), so results differ from PyTorch only by summation order.
#include "nn_kernels.h"
#include "lp_split.h"
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
constexpr int LIN_BM = 128;
// K chunk per barrier pair: 32 for the streaming shapes; 128 for the small-M GEMMs of the 2048-token networks, which are a
// handful of blocks deep in a chain of (global load -> LDS -> barrier -> MFMA) latencies: 4x fewer links per product.

// NT = 8 holds 128 accumulator registers: capping it at 256 total keeps two waves per SIMD resident (measured
// 1.84 -> 1.57 ms on the 1344 -> 512 head GEMM); pipelining the LDS fragment reads on top of that spills.
template <int NT, int LIN_BK = 32>
__global__ __launch_bounds__(256, NT == 8 ? 2 : 1) void linear_kernel(const float* __restrict__ X, long long ldx,
                                                     const float* __restrict__ W, long long ldw,
                                                     const float* __restrict__ bias, const float* __restrict__ row_bias,
                                                     long long rows_per_group, const float* __restrict__ R, long long ldr,
                                                     float* __restrict__ Y, long long ldy, long long M, int N, int K,
                                                     int act, int vec_x, int vec_w, const int* __restrict__ row_group) {
    constexpr int LIN_LD = LIN_BK + 4, F4 = LIN_BK / 4, RA = F4 / 2, RB = NT * F4 / 8;     // float4 per row; per-thread staging counts
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
    float4 ra[RA], rb[RB];
    auto fetch = [&](int k0) {
        if (vec_x) {
#pragma unroll
            for (int r = 0; r < RA; ++r) {
                const int idx = tid + r * 256, row = idx / F4, c4 = (idx % F4) * 4;
                ra[r] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (m0 + row < M && k0 + c4 < K) ra[r] = *reinterpret_cast<const float4*>(X + (m0 + row) * ldx + k0 + c4);
            }
        } else {
#pragma unroll
            for (int r = 0; r < RA; ++r) {
                const int idx = tid + r * 256, row = idx / F4, c4 = (idx % F4) * 4;
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
            for (int r = 0; r < RB; ++r) {
                const int idx = tid + r * 256, row = idx / F4, c4 = (idx % F4) * 4;
                rb[r] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (n0 + row < N && k0 + c4 < K) rb[r] = *reinterpret_cast<const float4*>(W + (long long)(n0 + row) * ldw + k0 + c4);
            }
        } else {
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                const int idx = tid + r * 256, row = idx / F4, c4 = (idx % F4) * 4;
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
        for (int r = 0; r < RA; ++r) {
            const int idx = tid + r * 256, row = idx / F4, c4 = (idx % F4) * 4;
            *reinterpret_cast<float4*>(&As[row * LIN_LD + c4]) = ra[r];
        }
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const int idx = tid + r * 256, row = idx / F4, c4 = (idx % F4) * 4;
            *reinterpret_cast<float4*>(&Bs[row * LIN_LD + c4]) = rb[r];
        }
        __syncthreads();
        if (k0 + LIN_BK < K) fetch(k0 + LIN_BK);       // in flight during the MFMA phase
        // ---- MFMA over the chunk ----
        const float* a_row = &As[(wave * 32 + i) * LIN_LD + h * (LIN_BK / 2)];
#pragma unroll
        for (int s4 = 0; s4 < LIN_BK / 8; ++s4) {
            const float4 a4 = *reinterpret_cast<const float4*>(a_row + s4 * 4);
            float4 b4[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) b4[t] = *reinterpret_cast<const float4*>(&Bs[(t * 32 + i) * LIN_LD + h * (LIN_BK / 2) + s4 * 4]);
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
            if (row_bias) y += row_bias[(row_group ? (long long)row_group[m] : m / rows_per_group) * N + n];
            if (act == ACT_GELU) y = gelu_erf(y);
            if (R) y += R[m * ldr + n];
            Y[m * ldy + n] = y;
        }
    }
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// K <= 4 (the xyz embeddings 3 -> 128 over all queries): a thread owns 4 consecutive outputs of one row.  The fmaf chain in
// ascending k from 0, then + bias, is bit for bit what the zero-padded fp32 MFMA path returns (the fp32 MFMA is an exact fmaf
// chain); that path spent 69 us on 100k x 128 outputs, this one is bound by its 51 MB of stores.
// PLANES: the result leaves as fp16 hi/lo planes Yh / Yl [M][ldy] (the operand format of the planes GEMM that consumes it: the
// SconeOcc head's x-embedding) instead of fp32 rows -- the same values split2h would make of them, without the 2 x 51 MB round trip.
template <bool PLANES>
__global__ __launch_bounds__(256) void linear_smallk_kernel(const float* __restrict__ X, long long ldx, const float* __restrict__ W,
                                                            long long ldw, const float* __restrict__ bias,
                                                            const float* __restrict__ R, long long ldr, float* __restrict__ Y,
                                                            long long ldy, long long M, int N, int K, int act,
                                                            _Float16* __restrict__ Yh, _Float16* __restrict__ Yl, int Nreal) {
    // N: outputs written per row (a multiple of 4); Nreal <= N: the layer's width -- outputs beyond it are exact zeros (PLANES: the
    // zero padding of the consuming GEMM's K)
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const int n4 = N >> 2;
    if (idx >= M * n4) return;
    const long long m = idx / n4;
    const int n = (int)(idx - m * n4) * 4;
    float x[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < K; ++k) x[k] = X[m * ldx + k];
    float y[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float acc = 0.f;
        if (n + j < Nreal) {
            for (int k = 0; k < K; ++k) acc = fmaf(x[k], W[(long long)(n + j) * ldw + k], acc);
            acc += bias ? bias[n + j] : 0.f;
            if (act == ACT_GELU) acc = gelu_erf(acc);
            if (R) acc += R[m * ldr + n + j];
        }
        y[j] = acc;
    }
    if (PLANES) {
        uint2 hi, lo;
        split2h(y[0], y[1], hi.x, lo.x);
        split2h(y[2], y[3], hi.y, lo.y);
        *reinterpret_cast<uint2*>(Yh + m * ldy + n) = hi;
        *reinterpret_cast<uint2*>(Yl + m * ldy + n) = lo;
    } else {
        *reinterpret_cast<float4*>(Y + m * ldy + n) = make_float4(y[0], y[1], y[2], y[3]);
    }
}

// The planes form at many rows (the xyz / xyz+occupancy embeddings of a batch of clouds, the query embedding of the SconeOcc head:
// 60k ... 100k rows x 128 outputs).  The kernel above re-reads 16 weights per thread through the vector memory path and calls libm's
// branchy erff: 79 us for 61 440 x 128 outputs.  Here a thread keeps the weights and the bias of its 4 outputs in registers and walks
// the rows of its block (up to LSK_ROWS per block -- fewer when that leaves the chip short of blocks; a row's result does not depend on
// it --, 256 / (N / 4) at a time); the fmaf chain (ascending k from 0, then + bias) is the one above,
// the GELU is l3_gelu -- the exact-erf GELU of every other epilogue of the planes path (|erf error| <= 1.5e-7).
constexpr int LSK_ROWS = 64;
__global__ __launch_bounds__(256) void linear_smallk_rows_kernel(const float* __restrict__ X, long long ldx, const float* __restrict__ W,
                                                                 long long ldw, const float* __restrict__ bias, long long ldy, long long M,
                                                                 int N, int K, int act, _Float16* __restrict__ Yh, _Float16* __restrict__ Yl,
                                                                 int Nreal, int rows) {
    const int qpr = N >> 2, rpp = 256 / qpr;               // quads per row; rows per pass
    const int q = threadIdx.x % qpr, rl = threadIdx.x / qpr, n = q * 4;
    float w[4][4], b[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        b[j] = (bias && n + j < Nreal) ? bias[n + j] : 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) w[j][k] = (k < K && n + j < Nreal) ? W[(long long)(n + j) * ldw + k] : 0.f;
    }
    const long long r0 = (long long)blockIdx.x * rows;
    for (int i = rl; i < rows; i += rpp) {
        const long long m = r0 + i;
        if (m >= M) break;
        float x[4] = {0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < K; ++k) x[k] = X[m * ldx + k];
        float y[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float acc = 0.f;
            for (int k = 0; k < K; ++k) acc = fmaf(x[k], w[j][k], acc);
            acc += b[j];
            if (act == ACT_GELU) acc = l3_gelu(acc);
            y[j] = n + j < Nreal ? acc : 0.f;
        }
        uint2 hi, lo;
        split2h(y[0], y[1], hi.x, lo.x);
        split2h(y[2], y[3], hi.y, lo.y);
        *reinterpret_cast<uint2*>(Yh + m * ldy + n) = hi;
        *reinterpret_cast<uint2*>(Yl + m * ldy + n) = lo;
    }
}

// act(X W^T + bias) for K <= 4, written as fp16 hi/lo planes (N % 4 == 0, ldy % 4 == 0)
void launch_linear_smallk_planes(hipStream_t s, const float* X, int64_t ldx, const float* W, const float* bias, void* Yh, void* Yl,
                                 int64_t ldy, int64_t M, int N, int K, int act, int Np) {
    if (M <= 0 || N <= 0) return;
    const int Nw = Np > N ? Np : N;                        // outputs written per row
    static const bool rows_on = []() { const char* e = getenv("MCR_SMALLK_ROWS"); return !(e && e[0] == '0'); }();
    if (rows_on && Nw % 4 == 0 && Nw / 4 <= 256 && 256 % (Nw / 4) == 0 && K <= 4) {   // (the GELU differs from the form below by <= 2e-7: every size takes this one)
        const int rpp = 256 / (Nw / 4);
        int rows = LSK_ROWS;
        while (rows > rpp && cdiv(M, rows) < 1024) rows >>= 1;
        rows = std::max(rows, rpp);
        hipLaunchKernelGGL(linear_smallk_rows_kernel, dim3((unsigned)cdiv(M, rows)), dim3(256), 0, s, X, (long long)ldx, W, (long long)K, bias,
                           (long long)ldy, (long long)M, Nw, K, act, (_Float16*)Yh, (_Float16*)Yl, N, rows);
        return;
    }
    hipLaunchKernelGGL(linear_smallk_kernel<true>, dim3((unsigned)cdiv(M * (Nw / 4), 256)), dim3(256), 0, s, X, (long long)ldx, W, (long long)K,
                       bias, (const float*)nullptr, 0ll, (float*)nullptr, (long long)ldy, (long long)M, Nw, K, act, (_Float16*)Yh, (_Float16*)Yl, N);
}

void launch_linear(hipStream_t s, const float* X, int64_t ldx, const float* W, const float* bias, const float* R,
                   int64_t ldr, float* Y, int64_t ldy, int64_t M, int N, int K, int act, const float* row_bias,
                   int64_t rows_per_group, int64_t ldw, int64_t route_rows, const int* row_group) {
    if (M <= 0 || N <= 0) return;
    if (ldw == 0) ldw = K;
    static const bool use_split = []() { const char* e = getenv("MCR_LINEAR3"); return !(e && e[0] == '0'); }();   // dev A/B knob
    // route_rows < 0: "one sequence of -route_rows rows, matrix path chosen on the layer's SHAPE alone" (the 2048-token encoders
    // on the split-precision variants: a batch of 30 sequences and a single one take the same kernel family, whose column-tile
    // width -- performance only -- follows M)
    const bool by_shape = route_rows < 0;
    if (by_shape) route_rows = -route_rows;
    if (use_split && (by_shape ? linear3_shape_ok(X, ldx, W, ldw, N, K) : linear3_applicable(X, ldx, W, ldw, route_rows > 0 ? route_rows : M, N, K))) {
        launch_linear3(s, X, ldx, W, bias, R, ldr, Y, ldy, M, N, K, act, row_bias, rows_per_group, ldw, row_group);
        return;
    }
    if (K <= 4 && N % 4 == 0 && ldy % 4 == 0 && aligned16(Y) && !row_bias && M * (N / 4) >= 65536) {
        hipLaunchKernelGGL(linear_smallk_kernel<false>, dim3((unsigned)cdiv(M * (N / 4), 256)), dim3(256), 0, s, X, (long long)ldx, W, (long long)ldw,
                           bias, R, (long long)ldr, Y, (long long)ldy, (long long)M, N, K, act, (_Float16*)nullptr, (_Float16*)nullptr, N);
        return;
    }
    const int vec_x = (K % 4 == 0) && (ldx % 4 == 0) && aligned16(X);
    const int vec_w = (K % 4 == 0) && (ldw % 4 == 0) && aligned16(W);
    const long long mb = cdiv(M, LIN_BM);
    // The K-chunk depth fixes the summation order (the two lane halves of the 32x32x2 MFMA own the two halves of a chunk), so it
    // is chosen on the ROUTING rows (one cloud), never on M: deep 128-wide chunks when one cloud's problem is a latency chain
    // (few blocks), 32-wide otherwise.  The column tile nt only groups independent accumulators and may follow M.
    const long long mb_r = cdiv(route_rows > 0 ? route_rows : M, LIN_BM);
    const bool deep_k = K >= 128 && K % 4 == 0 && mb_r * cdiv(N, 64) <= 512;
    // widest column tile (X is then read once), but never wider than N, and narrower when the problem is too
    // small to give every CU a few blocks otherwise
    int nt = deep_k ? 2 : 8;
    while (nt > 1 && (nt / 2) * 32 >= N) nt >>= 1;
    while (nt > 1 && mb * cdiv(N, nt * 32) < 512) nt >>= 1;
    dim3 grid((unsigned)mb, (unsigned)cdiv(N, nt * 32));
    if (deep_k) {
        if (nt == 2)
            hipLaunchKernelGGL((linear_kernel<2, 128>), grid, dim3(256), 0, s, X, (long long)ldx, W, (long long)ldw, bias, row_bias,
                               (long long)(rows_per_group > 0 ? rows_per_group : 1), R, (long long)ldr, Y, (long long)ldy, (long long)M,
                               N, K, act, vec_x, vec_w, row_group);
        else
            hipLaunchKernelGGL((linear_kernel<1, 128>), grid, dim3(256), 0, s, X, (long long)ldx, W, (long long)ldw, bias, row_bias,
                               (long long)(rows_per_group > 0 ? rows_per_group : 1), R, (long long)ldr, Y, (long long)ldy, (long long)M,
                               N, K, act, vec_x, vec_w, row_group);
        return;
    }
#define MCR_LIN(NT)                                                                                                   \
    hipLaunchKernelGGL((linear_kernel<NT>), grid, dim3(256), 0, s, X, (long long)ldx, W, (long long)ldw, bias, row_bias, \
                       (long long)(rows_per_group > 0 ? rows_per_group : 1), R, (long long)ldr, Y, (long long)ldy,       \
                       (long long)M, N, K, act, vec_x, vec_w, row_group)
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

// The same normalisation, the result written as fp16 hi/lo planes (hi = fp16(y), lo = fp16(y - hi): lp_split.h's split of the
// planes GEMMs): exactly the planes a split of the fp32 rows would give, without the fp32 round trip through HBM.
template <int EPL>
__global__ __launch_bounds__(256) void layernorm_planes_kernel(const float* __restrict__ X, long long ldx, const float* __restrict__ g,
                                                               const float* __restrict__ b, _Float16* __restrict__ Yh,
                                                               _Float16* __restrict__ Yl, long long ldp, long long M, int E) {
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
        if (c < E) {
            const float y = (v[e] - mean) * rstd * g[c] + b[c];
            const _Float16 hi = (_Float16)y;
            Yh[m * ldp + c] = hi;
            if (Yl) Yl[m * ldp + c] = (_Float16)(y - (float)hi);          // (NULL: the single-plane variant 7 keeps fp16(y) alone)
        }
    }
}

void launch_layernorm_planes(hipStream_t s, const float* X, int64_t ldx, const float* g, const float* b, void* Yh, void* Yl, int64_t ldp,
                             int64_t M, int E) {
    if (M <= 0) return;
    dim3 grid((unsigned)cdiv(M, 4));
    const int epl = (E + 63) / 64;
#define MCR_LNP(EPL) \
    hipLaunchKernelGGL((layernorm_planes_kernel<EPL>), grid, dim3(256), 0, s, X, (long long)ldx, g, b, (_Float16*)Yh, (_Float16*)Yl, \
                       (long long)ldp, (long long)M, E)
    if (epl <= 2) MCR_LNP(2);
    else if (epl <= 4) MCR_LNP(4);
    else MCR_LNP(8);
#undef MCR_LNP
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
// mask (optional; Attention.py:24-27): byte (sequence s, head h, query q, key k) at mask[s * ms + h * mh + q * mq + k]; where it is 0 the
// score is REPLACED by -1e3 before the 1/sqrt(d) scale (not -inf: a fully masked row attends uniformly, as upstream).
struct AttnMask { const unsigned char* p; long long ms, mh, mq; };
template <int L, int H, int DQ, int DV, int SPB>
__global__ __launch_bounds__(SPB* H* L) void attention_small_kernel(const float* __restrict__ qkv, long long ldq,
                                                                   float* __restrict__ out, long long ldo, long long S, AttnMask mk) {
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
        if (mk.p && mk.p[(s0 + sl) * mk.ms + hh * mk.mh + qi * mk.mq + j] == 0) a = -1e3f;
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

// =====================================================================================================
// attention, long sequences, on the fp32 matrix pipe (v_mfma_f32_16x16x4_f32 = exact fp32 fma chains; same math as
// attention_flash_kernel, which stays as the VALU reference path, env MCR_ATTN_MFMA=0).
//   block = 4 waves x 16 queries of one (sequence, head); keys/values stream through LDS in tiles of 64.
//   Per 16-key sub-tile a wave computes S^T = K Q^T (A = K rows from LDS, B = Q^T held in registers, pre-scaled by
//   1/sqrt(d)): lane (qi = l & 15, g = l >> 4) then owns S[qi][j0 + 4g + r], r = 0..3.  Online softmax per query over the
//   4 registers x 4 lane groups (two xor-shuffles per 64 keys).  O += P V uses the k index j = 4g + s, so the
//   probabilities are the A operand straight from their registers; O lives in C layout (lane (c, g) owns queries 4g + r),
//   so the per-query rescale factors are fetched across lane groups (4 ds_bpermute per 64 keys).
// grid = (ceil(L/64), H, S)
// =====================================================================================================
// lens (optional, device, one int per sequence): only the first min(L, lens[seq]) rows of a sequence are keys -- the padded
// SconeVis batches of the sync-free NBV step (the number of unique sampled points never reaches the host); rows beyond it
// still get an output (never read).
// SPLIT (one or two long sequences: L / 64 * H blocks do not fill the chip): grid.z = 2 * S, the two blocks of a (query tile,
// head, sequence) take the two halves of the keys and write UNNORMALISED outputs (part 0 -> out, part 1 -> part1 [T, H*DV]) plus
// their (running max, sum) per query -> ml [2][T][H][2]; attention_combine_kernel merges them.
// PVH (the fp16-split variant 6): O += P V on v_mfma_f32_16x16x16_f16 with both operands as fp16 hi/lo pairs (P in [0, 1] split in
// registers -- its 4 values per lane ARE the A fragment; V split once per tile while it is staged, stored as [16-column block][key][16]
// fp16 images and read back TRANSPOSED by ds_read_b64_tr_b16, which hands lane (c, g) the 4 keys 4g.. of column c = the B fragment):
// three v_mfma_f32_16x16x32_f16 of ~17 pipe cycles per 32 keys x 16 columns instead of eight fp32 ones of 32 (the P V product is
// 80 % of the kernel's matrix work; S = Q K^T stays exact fp32).  A tile holding a value outside the fp16 range (|v| >= 32768, inf, NaN) is detected
// while it is staged (block-wide OR folded into the tile barrier) and takes the fp32 path: no range restriction, no flag.
// QG (1 or 2): 16-query groups per wave.  A batch of sequences (8 clouds of a scene batch, the 30 neighbour cameras of a MACARONS
// decision) fills the chip with 128-query blocks too: every K / V tile staged (fetch, fp16 split, LDS commit: half of the kernel's
// vector instructions at QG = 1) and every K / V fragment read then serves 32 queries of a wave instead of 16.  Each query's
// arithmetic -- scores, 64-key softmax steps, the order of the P V products -- is the same instruction sequence whatever QG is:
// the result does not depend on it (tests: batch == single sequence, bit for bit).
// (Measured and not kept: K / V tiles double-buffered in LDS for one barrier per tile instead of two -- no change; the split of P and V
// by v_fma_mix -- no change: the kernel waits, it is not issue-bound.  Occupancy is what moves it: att_occ below.)
#ifndef MCR_ATT_OCC_16_1
#define MCR_ATT_OCC_16_1 2
#endif
#ifndef MCR_ATT_OCC_16_2
#define MCR_ATT_OCC_16_2 2
#endif
#ifndef MCR_ATT_OCC_8_1
#define MCR_ATT_OCC_8_1 3
#endif
#ifndef MCR_ATT_OCC_8_2
#define MCR_ATT_OCC_8_2 3
#endif
constexpr int att_occ(int dq, int qg) { return dq == 16 ? (qg == 1 ? MCR_ATT_OCC_16_1 : MCR_ATT_OCC_16_2) : (qg == 1 ? MCR_ATT_OCC_8_1 : MCR_ATT_OCC_8_2); }
template <int DQ, int DV, bool SPLIT, bool PVH, bool MASK = false, int QG = 1>
__global__ __launch_bounds__(256, att_occ(DQ, QG)) void attention_mfma_kernel(const float* __restrict__ qkv, long long ldq,
                                                             float* __restrict__ out, long long ldo, int L, int H,
                                                             const int* __restrict__ lens, float* __restrict__ part1,
                                                             float* __restrict__ ml, AttnMask mk = AttnMask{nullptr, 0, 0, 0}) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
    constexpr int TK = 64, LDK = DQ + 1, LDV = DV + 4, NT = DV / 16, KQ = DQ / 4, QB = 64 * QG;
    __shared__ __attribute__((aligned(16))) float s_k[TK * LDK];
    __shared__ __attribute__((aligned(16))) float s_v[TK * LDV];
    // PVH: the bytes of a V buffer hold the fp16 images of V instead: hi | lo, each [NT][64 keys][16 columns] (2 * NT * 2 KB <= the fp32 tile);
    // the K buffer likewise: hi | lo, each [64 keys][DQ]
    static_assert(2 * NT * TK * 16 * 2 <= TK * LDV * 4, "fp16 V images do not fit the fp32 tile");
    static_assert(2 * TK * DQ * 2 <= TK * LDK * 4, "fp16 K images do not fit the fp32 tile");
    static_assert(DQ == 8 || DQ == 16, "score fragments are laid out for 8 or 16 query dimensions");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, g = lane >> 4;
    const int hh = blockIdx.y;
    const int seq = SPLIT ? blockIdx.z >> 1 : blockIdx.z, part = SPLIT ? blockIdx.z & 1 : 0;
    const long long seq0 = (long long)seq * L;
    const int Lk_all = lens ? max(1, min(L, __builtin_amdgcn_readfirstlane(lens[seq]))) : L;             // number of keys
    const int kmid = min(Lk_all, ((Lk_all / 2 + TK - 1) / TK) * TK);                                    // tile-aligned cut
    const int kb = SPLIT && part ? kmid : 0, Lk = SPLIT && !part ? kmid : Lk_all;                        // this block's keys [kb, Lk)
    const int q0 = blockIdx.x * QB + wave * 16 * QG;    // + 16 qg: the wave's query groups
    const int koff = H * DQ + hh * DQ, voff = 2 * H * DQ + hh * DV;
    // scores are kept in units of log 2: softmax weights are exp2(s' - max s'), s' = s log2(e) -- one v_exp_f32 per weight, no multiply
    const float scale = 1.4426950408889634f / sqrtf((float)DQ);

    float qb[QG][KQ];                                   // B operand of S^T on the fp32 pipe: Q[q0 + 16 qg + li][4s + g] * scale
    f16x8 qh[QG];                                       // ... and on fp16 pairs (PVH), see "scores" below
    bool q_half[QG];                                    // is the query group inside the fp16 range? (wave-uniform)
#pragma unroll
    for (int qg = 0; qg < QG; ++qg) {
        const int qi = min(q0 + 16 * qg + li, L - 1);
        const float* qp = qkv + (seq0 + qi) * ldq + hh * DQ;
#pragma unroll
        for (int sk = 0; sk < KQ; ++sk) qb[qg][sk] = qp[4 * sk + g] * scale;
        q_half[qg] = false;
        if (PVH) {
            const int d0 = DQ == 16 ? 8 * (g & 1) : 0;  // this lane group's 8 query dimensions
            float4 a = *reinterpret_cast<const float4*>(qp + d0), b = *reinterpret_cast<const float4*>(qp + d0 + 4);
            a.x *= scale; a.y *= scale; a.z *= scale; a.w *= scale; b.x *= scale; b.y *= scale; b.z *= scale; b.w *= scale;
            const float mx = fmaxf(fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))),
                                   fmaxf(fmaxf(fabsf(b.x), fabsf(b.y)), fmaxf(fabsf(b.z), fabsf(b.w))));
            q_half[qg] = !__any(!(mx < 32768.f));
            const Split2 s2 = split8h(a, b);
            const bool lo_part = DQ == 16 ? g >= 2 : (g & 1);
            qh[qg] = __builtin_bit_cast(f16x8, lo_part ? s2.lo : s2.hi);
        }
    }
    float m[QG], l[QG];                                 // running max (shared by the 4 lanes of a query), this lane's part of the sum
    f32x4 o[QG][NT];
#pragma unroll
    for (int qg = 0; qg < QG; ++qg) {
        m[qg] = -__builtin_inff(); l[qg] = 0.f;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) o[qg][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    // staging: the [64 keys x DQ] K slice is one float4 of the first 64 DQ / 4 threads, the [64 keys x DV] V slice DV / 16 float4 of every
    // thread (K or V is a property of the slot, not of the thread: no divergent commit); a tile's loads are all issued one MFMA phase
    // before they are committed to LDS (the per-element copy loop of the VALU kernel serialises ~20 load latencies per tile and
    // is what bounds it)
    constexpr int KF4 = TK * DQ / 4, C4K = DQ / 4, C4V = DV / 4, PV_ = TK * C4V / 256;
    static_assert(KF4 <= 256 && TK * C4V % 256 == 0, "staging slots");
    const bool k_thread = (int)threadIdx.x < KF4;       // wave-uniform (KF4 = 128 or 256)
    const int k_r = threadIdx.x / C4K, k_c4 = threadIdx.x - k_r * C4K;
    float4 stage_k, stage_v[PV_];
    auto fetch = [&](int t0) {
        stage_k = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k_thread && t0 + k_r < Lk) stage_k = *reinterpret_cast<const float4*>(qkv + (seq0 + t0 + k_r) * ldq + koff + 4 * k_c4);
#pragma unroll
        for (int p = 0; p < PV_; ++p) {
            const int idx = threadIdx.x + p * 256, r = idx / C4V, c4 = idx - r * C4V;
            stage_v[p] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (t0 + r < Lk) stage_v[p] = *reinterpret_cast<const float4*>(qkv + (seq0 + t0 + r) * ldq + voff + 4 * c4);
        }
    };
    auto out_of_half_range = [&]() -> int {            // does this thread's share of the K / V tile leave the fp16 range?
        float mx = fmaxf(fmaxf(fabsf(stage_k.x), fabsf(stage_k.y)), fmaxf(fabsf(stage_k.z), fabsf(stage_k.w)));
        int big = !(mx < 32768.f);                        // (NaN compares false: caught)
#pragma unroll
        for (int p = 0; p < PV_; ++p) {
            mx = fmaxf(fmaxf(fabsf(stage_v[p].x), fabsf(stage_v[p].y)), fmaxf(fabsf(stage_v[p].z), fabsf(stage_v[p].w)));
            big |= !(mx < 32768.f);
        }
        return big;
    };
    _Float16* s_vh = reinterpret_cast<_Float16*>(s_v);
    _Float16* s_vl = s_vh + NT * TK * 16;
    _Float16* s_kh = reinterpret_cast<_Float16*>(s_k);
    _Float16* s_kl = s_kh + TK * DQ;
    auto commit = [&](const bool half_v) {
        if (k_thread) {
            if (half_v) {
                uint2 hi, lo;
                split2h(stage_k.x, stage_k.y, hi.x, lo.x);
                split2h(stage_k.z, stage_k.w, hi.y, lo.y);
                *reinterpret_cast<uint2*>(s_kh + k_r * DQ + 4 * k_c4) = hi;
                *reinterpret_cast<uint2*>(s_kl + k_r * DQ + 4 * k_c4) = lo;
            } else {
                float* d = s_k + k_r * LDK + 4 * k_c4;         // odd row stride (bank-conflict-free fragment reads): scalar stores
                d[0] = stage_k.x; d[1] = stage_k.y; d[2] = stage_k.z; d[3] = stage_k.w;
            }
        }
#pragma unroll
        for (int p = 0; p < PV_; ++p) {
            const int idx = threadIdx.x + p * 256, r = idx / C4V, c4 = idx - r * C4V;
            if (half_v) {
                const int col = 4 * c4, off = ((col >> 4) * TK + r) * 16 + (col & 15);    // [16-column block][key][16]
                uint2 hi, lo;
                split2h(stage_v[p].x, stage_v[p].y, hi.x, lo.x);
                split2h(stage_v[p].z, stage_v[p].w, hi.y, lo.y);
                *reinterpret_cast<uint2*>(s_vh + off) = hi;
                *reinterpret_cast<uint2*>(s_vl + off) = lo;
            } else {
                *reinterpret_cast<float4*>(s_v + r * LDV + 4 * c4) = stage_v[p];
            }
        }
    };
    fetch(kb);
    for (int t0 = kb; t0 < Lk; t0 += TK) {
        // the previous tile is consumed (barrier); PVH: the same barrier tells every thread whether the tile fits the fp16 range
        bool half_v = false;
        if (PVH) half_v = !__syncthreads_or(out_of_half_range());
        else __syncthreads();
        commit(half_v);
        __syncthreads();
        if (t0 + TK < Lk) fetch(t0 + TK);
        const float* vbase = s_v + (4 * g) * LDV + li;       // V[sub*16 + 4g + sk][nt*16 + li]
        float vb[2][4][NT];
        f32x4 st[QG][4];
        // ---- scores of the 64 keys of the tile ----
        if (PVH && half_v) {
            // S^T = K Q^T on fp16 pairs, all four cross terms of (k_hi + k_lo)(q_hi + q_lo) through the k = 32 of v_mfma_f32_16x16x32_f16:
            // DQ = 16: lane groups 0, 1 carry the dimensions 0..7, 8..15 of q_hi, groups 2, 3 those of q_lo (B); A is K_lo, then K_hi, with
            // the same 16 dimensions in both halves of k -> two MFMAs per 16 keys x 16 queries (was four exact-fp32 ones of twice the cycles);
            // DQ = 8: B = q_hi | q_lo | q_hi | q_lo, A = k_hi | k_hi | k_lo | k_lo by lane group -> one MFMA
            f16x8 kf[4][DQ == 16 ? 2 : 1];
#pragma unroll
            for (int sub = 0; sub < 4; ++sub) {
                if (DQ == 16) {
                    kf[sub][0] = *reinterpret_cast<const f16x8*>(s_kl + (sub * 16 + li) * DQ + 8 * (g & 1));
                    kf[sub][DQ == 16 ? 1 : 0] = *reinterpret_cast<const f16x8*>(s_kh + (sub * 16 + li) * DQ + 8 * (g & 1));
                } else {
                    kf[sub][0] = *reinterpret_cast<const f16x8*>((g >= 2 ? s_kl : s_kh) + (sub * 16 + li) * DQ);
                }
            }
#pragma unroll
            for (int qg = 0; qg < QG; ++qg) {
                if (q_half[qg]) {
#pragma unroll
                    for (int sub = 0; sub < 4; ++sub) {
                        st[qg][sub] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[sub][0], qh[qg], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                        if (DQ == 16) st[qg][sub] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[sub][DQ == 16 ? 1 : 0], qh[qg], st[qg][sub], 0, 0, 0);
                    }
                } else {                                  // a query outside the fp16 range: fp32 pipe on k_hi + k_lo (never in the networks' range)
#pragma unroll
                    for (int sub = 0; sub < 4; ++sub) {
                        st[qg][sub] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int sk = 0; sk < KQ; ++sk) {
                            const int at = (sub * 16 + li) * DQ + 4 * sk + g;
                            st[qg][sub] = __builtin_amdgcn_mfma_f32_16x16x4f32((float)s_kh[at] + (float)s_kl[at], qb[qg][sk], st[qg][sub], 0, 0, 0);
                        }
                    }
                }
            }
        } else {
            // fp32 pipe (all A fragments first: LLVM otherwise issues every ds_read right in front of its MFMA and each one waits out
            // the LDS latency)
            const float* kbase = s_k + li * LDK + g;             // K[sub*16 + li][4 sk + g]
            float ka[4][KQ];
#pragma unroll
            for (int sub = 0; sub < 4; ++sub)
#pragma unroll
                for (int sk = 0; sk < KQ; ++sk) ka[sub][sk] = kbase[sub * 16 * LDK + 4 * sk];
#pragma unroll
            for (int sk = 0; sk < 4; ++sk)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) vb[0][sk][nt] = vbase[sk * LDV + nt * 16];
#pragma unroll
            for (int qg = 0; qg < QG; ++qg)
#pragma unroll
                for (int sub = 0; sub < 4; ++sub) {
                    st[qg][sub] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int sk = 0; sk < KQ; ++sk)
                        st[qg][sub] = __builtin_amdgcn_mfma_f32_16x16x4f32(ka[sub][sk], qb[qg][sk], st[qg][sub], 0, 0, 0);
                }
        }
        float ar[QG][4];
#pragma unroll
        for (int qg = 0; qg < QG; ++qg) {
            if (MASK) {                                   // Attention.py:24-27: masked pairs score -1e3 (then / sqrt(d)), not -inf
                const unsigned char* mrow = mk.p + (long long)seq * mk.ms + hh * mk.mh + (long long)min(q0 + 16 * qg + li, L - 1) * mk.mq + t0;
#pragma unroll
                for (int sub = 0; sub < 4; ++sub)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = sub * 16 + 4 * g + r;
                        if (t0 + key < L && mrow[key] == 0) st[qg][sub][r] = -1e3f * scale;
                    }
            }
            if (t0 + TK > Lk) {                           // only the last tile of a sequence has keys past its end (block-uniform)
#pragma unroll
                for (int sub = 0; sub < 4; ++sub)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (t0 + sub * 16 + 4 * g + r >= Lk) st[qg][sub][r] = -__builtin_inff();
            }
            float tmax = -__builtin_inff();
#pragma unroll
            for (int sub = 0; sub < 4; ++sub)
#pragma unroll
                for (int r = 0; r < 4; ++r) tmax = fmaxf(tmax, st[qg][sub][r]);
            tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
            const float m_new = fmaxf(m[qg], tmax);
            const float alpha = __builtin_amdgcn_exp2f(m[qg] - m_new);   // m = -inf on the first tile -> 0 (o = l = 0 anyway)
            m[qg] = m_new;
            float psum = 0.f;
#pragma unroll
            for (int sub = 0; sub < 4; ++sub)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    st[qg][sub][r] = __builtin_amdgcn_exp2f(st[qg][sub][r] - m_new);
                    psum += st[qg][sub][r];
                }
            l[qg] = fmaf(l[qg], alpha, psum);
            // ---- rescale O (rows = queries 4g + r live in lane group g); after the first tiles the maxima rarely move: a wave whose 16
            // queries all keep theirs skips the exchange and the multiplications by 1 (same bits) ----
            if (__any(alpha != 1.f)) {
#pragma unroll
                for (int r = 0; r < 4; ++r) ar[qg][r] = __shfl(alpha, 4 * g + r, 64);   // alpha of query 4g + r (any lane group holds it)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[qg][nt][r] *= ar[qg][r];
            }
        }
        // ---- accumulate P V ----
        if (PVH && half_v) {
            // P (this lane: queries li, keys 4g + r of every 16-key sub-tile) is the A fragment as it stands; V fragments by transpose read
            const _Float16* vh0 = s_vh + (4 * g) * 16 + li * 4;
            const _Float16* vl0 = s_vl + (4 * g) * 16 + li * 4;
            // two 16-key sub-tiles per v_mfma_f32_16x16x32_f16 (the double-rate shape of gfx950; the 16x16x16 one runs at the fp32
            // MFMA's cycle count): k index 8 g + r <-> key 4 g + r of sub-tile 2 q (r < 4) or 2 q + 1 (r >= 4), on both operands
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                f16x8 p_hi[QG], p_lo[QG];
#pragma unroll
                for (int qg = 0; qg < QG; ++qg) {
                    uint4 ph, pl;
                    split2h(st[qg][2 * q][0], st[qg][2 * q][1], ph.x, pl.x);
                    split2h(st[qg][2 * q][2], st[qg][2 * q][3], ph.y, pl.y);
                    split2h(st[qg][2 * q + 1][0], st[qg][2 * q + 1][1], ph.z, pl.z);
                    split2h(st[qg][2 * q + 1][2], st[qg][2 * q + 1][3], ph.w, pl.w);
                    p_hi[qg] = __builtin_bit_cast(f16x8, ph); p_lo[qg] = __builtin_bit_cast(f16x8, pl);
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const _Float16* bh = vh0 + (nt * TK + q * 32) * 16;
                    const _Float16* bl = vl0 + (nt * TK + q * 32) * 16;
                    const s16x4 h0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)bh), h1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(bh + 256));
                    const s16x4 l0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)bl), l1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(bl + 256));
                    const f16x8 v_hi = __builtin_bit_cast(f16x8, s16x8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]});
                    const f16x8 v_lo = __builtin_bit_cast(f16x8, s16x8{l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]});
#pragma unroll
                    for (int qg = 0; qg < QG; ++qg) {
                        o[qg][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(p_lo[qg], v_hi, o[qg][nt], 0, 0, 0);      // smallest terms first
                        o[qg][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(p_hi[qg], v_lo, o[qg][nt], 0, 0, 0);
                        o[qg][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(p_hi[qg], v_hi, o[qg][nt], 0, 0, 0);
                    }
                }
            }
        } else {
#pragma unroll
            for (int sub = 0; sub < 4; ++sub) {
                if (sub + 1 < 4) {
#pragma unroll
                    for (int sk = 0; sk < 4; ++sk)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) vb[(sub + 1) & 1][sk][nt] = vbase[((sub + 1) * 16 + sk) * LDV + nt * 16];
                }
#pragma unroll
                for (int qg = 0; qg < QG; ++qg)
#pragma unroll
                    for (int sk = 0; sk < 4; ++sk)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            o[qg][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(st[qg][sub][sk], vb[sub & 1][sk][nt], o[qg][nt], 0, 0, 0);
                if (sub + 1 < 4) __builtin_amdgcn_sched_group_barrier(0x100, 4 * NT, 0);     // next sub-tile's V fragments ...
                __builtin_amdgcn_sched_group_barrier(0x008, 4 * NT * QG, 0);                 // ... then this one's MFMAs
            }
        }
    }
    // ---- normalise and write: lane (c = li, g) owns O[q0 + 16 qg + 4g + r][nt*16 + li] ----
#pragma unroll
    for (int qg = 0; qg < QG; ++qg) {
        float lq = l[qg];
        lq += __shfl_xor(lq, 16, 64);
        lq += __shfl_xor(lq, 32, 64);
        const int qbase = q0 + 16 * qg;
        if (SPLIT) {
            if (g == 0 && qbase + li < L) {
                float* p = ml + (((long long)part * gridDim.z / 2 * L + seq0 + qbase + li) * H + hh) * 2;
                p[0] = m[qg]; p[1] = lq;
            }
            float* dst = part ? part1 : out;
            const long long ld = part ? (long long)H * DV : ldo;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qi = qbase + 4 * g + r;
                if (qi < L) {
                    float* orow = dst + (seq0 + qi) * ld + hh * DV;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) orow[nt * 16 + li] = o[qg][nt][r];
                }
            }
        } else {
            const float inv = 1.0f / lq;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float ir = __shfl(inv, 4 * g + r, 64);
                const int qi = qbase + 4 * g + r;
                if (qi < L) {
                    float* orow = out + (seq0 + qi) * ldo + hh * DV;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) orow[nt * 16 + li] = o[qg][nt][r] * ir;
                }
            }
        }
    }
}


// out[t, h, :] = (w0 o0 + w1 o1) / (w0 l0 + w1 l1),  w_p = exp(m_p - max(m0, m1)); a part without keys has m = -inf, l = 0
// Ph / Pl (optional, row stride ldp halves): the result leaves as fp16 hi/lo planes INSTEAD of fp32 rows -- the input of a planes GEMM
// (the out-projection of an encoder on the fp16-split matrix path), split where it is produced
__global__ void attention_combine_kernel(float* __restrict__ out, long long ldo, const float* __restrict__ part1,
                                         const float* __restrict__ ml, long long T, int H, int DV, _Float16* __restrict__ Ph,
                                         _Float16* __restrict__ Pl, long long ldp) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int E = H * DV;
    if (idx >= T * E) return;
    const long long t = idx / E;
    const int c = (int)(idx - t * E), h = c / DV;
    const float* p0 = ml + (t * H + h) * 2;
    const float* p1 = ml + ((T + t) * H + h) * 2;
    const float m0 = p0[0], l0 = p0[1], m1 = p1[0], l1 = p1[1];
    const float M = fmaxf(m0, m1);
    const float w0 = __builtin_amdgcn_exp2f(m0 - M), w1 = l1 > 0.f ? __builtin_amdgcn_exp2f(m1 - M) : 0.f;   // the maxima are in units of log 2
    const float y = (w0 * out[t * ldo + c] + w1 * part1[t * E + c]) / (w0 * l0 + w1 * l1);
    if (Ph) {
        const _Float16 hi = (_Float16)y;
        Ph[t * ldp + c] = hi;
        Pl[t * ldp + c] = (_Float16)(y - (float)hi);
    } else {
        out[t * ldo + c] = y;
    }
}

void launch_attention_combine(hipStream_t s, float* out, int64_t ldo, const float* part1, const float* ml, int64_t T, int H, int dv,
                              void* planes_h, void* planes_l, int64_t ldp) {
    hipLaunchKernelGGL(attention_combine_kernel, dim3((unsigned)cdiv(T * H * dv, 256)), dim3(256), 0, s, out, (long long)ldo, part1, ml,
                       (long long)T, H, dv, (_Float16*)planes_h, (_Float16*)planes_l, (long long)ldp);
}

size_t attention_split_floats(int64_t S, int L, int H, int DV) { return (size_t)S * L * DV + (size_t)4 * S * L * H; }

void launch_attention(hipStream_t s, const float* qkv, int64_t ldq, float* out, int64_t ldo, int64_t S, int L, int H,
                      int DQK, int DV, const int* lens, float* split_ws, size_t split_ws_floats, bool split_by_length, bool pv_half,
                      const unsigned char* mask, int64_t mask_seq_stride, int64_t mask_head_stride, int64_t mask_query_stride,
                      void* planes_h, void* planes_l, int64_t ldp, bool* planes_done) {
    if (planes_done) *planes_done = false;
    if (S <= 0 || L <= 0) return;
    const int dq = DQK / H, dv = DV / H;
    const AttnMask mk{mask, (long long)mask_seq_stride, (long long)mask_head_stride, (long long)mask_query_stride};
    if (!lens && L == 16 && H == 4 && dq == 8 && dv == 32) {
        constexpr int SPB = 4;
        hipLaunchKernelGGL((attention_small_kernel<16, 4, 8, 32, SPB>), dim3((unsigned)cdiv(S, SPB)), dim3(SPB * 4 * 16), 0,
                           s, qkv, (long long)ldq, out, (long long)ldo, (long long)S, mk);
        return;
    }
    constexpr int KS = 4;
    dim3 grid((unsigned)cdiv(L, 64), (unsigned)H, (unsigned)S);
    static const bool use_mfma = []() { const char* e = getenv("MCR_ATTN_MFMA"); return !(e && e[0] == '0'); }();   // dev A/B knob
    const bool al16 = aligned16(qkv) && ldq % 4 == 0 && (H * dq) % 4 == 0;
    // one or two long sequences leave half the chip idle (L = 2048, 4 heads: 128 blocks): split the keys over two blocks
    const bool split = split_ws && split_ws_floats >= attention_split_floats(S, L, H, DV) && L >= 512 &&
                       2 * S <= 65535 && (split_by_length || (int64_t)grid.x * H * S <= 256);
    if (use_mfma && al16 && ((dq == 8 && dv == 32) || (dq == 16 && dv == 64))) {
#define MCR_ATT(DQ_, DV_, SPLIT_, GRID_, P1_, ML_)                                                                                    \
    do {                                                                                                                               \
        if (mask)                                                                                                                      \
            hipLaunchKernelGGL((attention_mfma_kernel<DQ_, DV_, SPLIT_, false, true>), GRID_, dim3(256), 0, s, qkv, (long long)ldq,     \
                               out, (long long)ldo, L, H, lens, P1_, ML_, mk);                                                         \
        else if (pv_half)                                                                                                              \
            hipLaunchKernelGGL((attention_mfma_kernel<DQ_, DV_, SPLIT_, true>), GRID_, dim3(256), 0, s, qkv, (long long)ldq, out,       \
                               (long long)ldo, L, H, lens, P1_, ML_);                                                                  \
        else                                                                                                                           \
            hipLaunchKernelGGL((attention_mfma_kernel<DQ_, DV_, SPLIT_, false>), GRID_, dim3(256), 0, s, qkv, (long long)ldq, out,      \
                               (long long)ldo, L, H, lens, P1_, ML_);                                                                  \
    } while (0)
#define MCR_ATT2(DQ_, DV_, SPLIT_, GRID_, P1_, ML_)                                                                                   \
    do {                                                                                                                               \
        if (pv_half)                                                                                                                   \
            hipLaunchKernelGGL((attention_mfma_kernel<DQ_, DV_, SPLIT_, true, false, 2>), GRID_, dim3(256), 0, s, qkv, (long long)ldq,  \
                               out, (long long)ldo, L, H, lens, P1_, ML_);                                                             \
        else                                                                                                                           \
            hipLaunchKernelGGL((attention_mfma_kernel<DQ_, DV_, SPLIT_, false, false, 2>), GRID_, dim3(256), 0, s, qkv, (long long)ldq, \
                               out, (long long)ldo, L, H, lens, P1_, ML_);                                                             \
    } while (0)
        // a batch of sequences fills the chip with 128-query blocks (two 16-query groups per wave: half the staging and fragment
        // reads per query; same bits): >= 2 blocks per CU; MCR_ATTN_QG2=0: always 64-query blocks (A/B)
        static const bool qg2_on = []() { const char* e = getenv("MCR_ATTN_QG2"); return !(e && e[0] == '0'); }();
        const dim3 grid2((unsigned)cdiv(L, 128), (unsigned)H, (unsigned)(split ? 2 * S : S));
        const bool qg2 = qg2_on && !mask && (int64_t)grid2.x * grid2.y * grid2.z >= 512;
        if (split) {
            float* part1 = split_ws;
            float* ml = split_ws + (size_t)S * L * DV;
            const dim3 g2(grid.x, grid.y, (unsigned)(2 * S));
            if (qg2) {
                if (dq == 8) MCR_ATT2(8, 32, true, grid2, part1, ml);
                else MCR_ATT2(16, 64, true, grid2, part1, ml);
            } else if (dq == 8) {
                MCR_ATT(8, 32, true, g2, part1, ml);
            } else {
                MCR_ATT(16, 64, true, g2, part1, ml);
            }
            launch_attention_combine(s, out, ldo, part1, ml, S * L, H, dv, planes_h, planes_l, ldp);
            if (planes_done) *planes_done = planes_h != nullptr;
        } else if (qg2) {
            if (dq == 8) MCR_ATT2(8, 32, false, grid2, (float*)nullptr, (float*)nullptr);
            else MCR_ATT2(16, 64, false, grid2, (float*)nullptr, (float*)nullptr);
        } else if (dq == 8) {
            MCR_ATT(8, 32, false, grid, (float*)nullptr, (float*)nullptr);
        } else {
            MCR_ATT(16, 64, false, grid, (float*)nullptr, (float*)nullptr);
        }
#undef MCR_ATT2
#undef MCR_ATT
        return;
    }
    if (lens || mask) {
        set_error("launch_attention: per-sequence lengths / masks need the MFMA kernel (16-byte aligned qkv, head dims (8,32) or (16,64))");
        return;
    }
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
// grid = (S, ceil(E/64)); block = 64 columns x 16 row lanes (the long sequences here are one or two per launch, so the
// rows have to be spread inside the block: with 4 row lanes the 2048-row reductions of an NBV step took 0.27 ms)
constexpr int POOL_RL = 16;
template <bool BROADCAST>
__global__ __launch_bounds__(64 * POOL_RL) void pool_kernel(const float* __restrict__ X, long long ldx, float* __restrict__ Y,
                                                            long long ldy, int L, int E, const int* __restrict__ lens) {
    __shared__ float s_max[POOL_RL][64];
    __shared__ float s_sum[POOL_RL][64];
    const int cl = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int c = blockIdx.y * 64 + cl;
    const long long row0 = (long long)blockIdx.x * L;
    const int Lr = lens ? max(1, min(L, lens[blockIdx.x])) : L;          // rows that take part in the reduction
    float mx = -__builtin_inff(), sm = 0.f;
    if (c < E) {
        // 8 rows in flight per thread (a one-row loop is a chain of exposed load latencies: 128 of them at L = 2048); the sum
        // keeps its row order
        int r = g;
        for (; r + 7 * POOL_RL < Lr; r += 8 * POOL_RL) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = X[(row0 + r + j * POOL_RL) * ldx + c];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                mx = fmaxf(mx, v[j]);
                sm += v[j];
            }
        }
        for (; r < Lr; r += POOL_RL) {
            const float v = X[(row0 + r) * ldx + c];
            mx = fmaxf(mx, v);
            sm += v;
        }
    }
    s_max[g][cl] = mx;
    s_sum[g][cl] = sm;
    __syncthreads();
    mx = s_max[0][cl];
    sm = s_sum[0][cl];
#pragma unroll
    for (int k = 1; k < POOL_RL; ++k) {
        mx = fmaxf(mx, s_max[k][cl]);
        sm += s_sum[k][cl];
    }
    if (c >= E) return;
    if (BROADCAST) {
        for (int r = g; r < L; r += POOL_RL) Y[(row0 + r) * ldy + c] = mx;
    } else if (g == 0) {
        Y[(long long)blockIdx.x * ldy + c] = mx;
        Y[(long long)blockIdx.x * ldy + E + c] = sm / (float)Lr;
    }
}

// The same reductions for LONG sequences (L >= POOL_LONG_MIN rows; a launch holds 1 ... 41 of them): 16 columns x 64 row lanes per block,
// grid = (S, ceil(E/16)).  With 64 columns per block a single cloud of 2048 tokens was TWO blocks, each a chain of 16 rounds of exposed
// load latency (17 us of a 0.33 ms SconeVis forward); here it is E/16 blocks of 4 rounds.  The choice is made on L alone, so a cloud's
// mean has the same summation tree at every batch size (the max is exact in any order).  `ex` (BROADCAST only): ex_cols <= 16 columns of a
// second [S*L, ex_ld] array copied to ex_dst (leading dimension ldy) by the first column block: SconeVis's raw-input columns, without a
// launch of their own.
void launch_copy2d(hipStream_t s, const float* X, int64_t ldx, float* Y, int64_t ldy, int64_t M, int E);
constexpr int POOL_LONG_MIN = 512, POOL_LONG_RL = 64;
template <bool BROADCAST>
__global__ __launch_bounds__(16 * POOL_LONG_RL) void pool_long_kernel(const float* __restrict__ X, long long ldx, float* __restrict__ Y,
                                                                      long long ldy, int L, int E, const int* __restrict__ lens,
                                                                      const float* __restrict__ ex, int ex_ld, int ex_cols,
                                                                      float* __restrict__ ex_dst) {
    __shared__ float s_max[16][16];
    __shared__ float s_sum[16][16];
    const int cl = threadIdx.x & 15, g = threadIdx.x >> 4, w = threadIdx.x >> 6;
    const int c = blockIdx.y * 16 + cl;
    const long long row0 = (long long)blockIdx.x * L;
    const int Lr = lens ? max(1, min(L, lens[blockIdx.x])) : L;
    float mx = -__builtin_inff(), sm = 0.f;
    if (c < E) {
        int r = g;
        for (; r + 7 * POOL_LONG_RL < Lr; r += 8 * POOL_LONG_RL) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = X[(row0 + r + j * POOL_LONG_RL) * ldx + c];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                mx = fmaxf(mx, v[j]);
                sm += v[j];
            }
        }
        for (; r < Lr; r += POOL_LONG_RL) {
            const float v = X[(row0 + r) * ldx + c];
            mx = fmaxf(mx, v);
            sm += v;
        }
    }
    // the 4 row lanes of a wave (lanes cl, cl+16, cl+32, cl+48), then the 16 waves in order
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    sm += __shfl_xor(sm, 16);
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    sm += __shfl_xor(sm, 32);
    if ((threadIdx.x & 63) < 16) {
        s_max[w][cl] = mx;
        s_sum[w][cl] = sm;
    }
    __syncthreads();
    mx = s_max[0][cl];
    sm = s_sum[0][cl];
#pragma unroll
    for (int k = 1; k < 16; ++k) {
        mx = fmaxf(mx, s_max[k][cl]);
        sm += s_sum[k][cl];
    }
    if (BROADCAST) {
        if (c < E)
            for (int r = g; r < L; r += POOL_LONG_RL) Y[(row0 + r) * ldy + c] = mx;
        if (ex && blockIdx.y == 0 && cl < ex_cols)
            for (int r = g; r < L; r += POOL_LONG_RL) ex_dst[(row0 + r) * ldy + cl] = ex[(row0 + r) * ex_ld + cl];
    } else if (g == 0 && c < E) {
        Y[(long long)blockIdx.x * ldy + c] = mx;
        Y[(long long)blockIdx.x * ldy + E + c] = sm / (float)Lr;
    }
}

void launch_colmax_broadcast(hipStream_t s, const float* X, int64_t ldx, float* Y, int64_t ldy, int64_t S, int L, int E,
                             const int* lens, const float* ex, int ex_ld, int ex_cols, float* ex_dst) {
    if (S <= 0) return;
    if (L >= POOL_LONG_MIN && ex_cols <= 16) {
        hipLaunchKernelGGL((pool_long_kernel<true>), dim3((unsigned)S, (unsigned)cdiv(E, 16)), dim3(16 * POOL_LONG_RL), 0, s, X,
                           (long long)ldx, Y, (long long)ldy, L, E, lens, ex, ex_ld, ex_cols, ex_dst);
        return;
    }
    hipLaunchKernelGGL((pool_kernel<true>), dim3((unsigned)S, (unsigned)cdiv(E, 64)), dim3(64 * POOL_RL), 0, s, X, (long long)ldx, Y,
                       (long long)ldy, L, E, lens);
    if (ex) launch_copy2d(s, ex, ex_ld, ex_dst, ldy, S * (int64_t)L, ex_cols);
}

void launch_pool_max_avg(hipStream_t s, const float* X, int64_t ldx, float* Y, int64_t ldy, int64_t S, int L, int E, const int* lens) {
    if (S <= 0) return;
    if (L >= POOL_LONG_MIN) {
        hipLaunchKernelGGL((pool_long_kernel<false>), dim3((unsigned)S, (unsigned)cdiv(E, 16)), dim3(16 * POOL_LONG_RL), 0, s, X,
                           (long long)ldx, Y, (long long)ldy, L, E, lens, (const float*)nullptr, 0, 0, (float*)nullptr);
        return;
    }
    hipLaunchKernelGGL((pool_kernel<false>), dim3((unsigned)S, (unsigned)cdiv(E, 64)), dim3(64 * POOL_RL), 0, s, X, (long long)ldx, Y,
                       (long long)ldy, L, E, lens);
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
