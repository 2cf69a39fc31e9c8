// K4-split — nn.Linear on the split-precision ("bf16 x 6") matrix path for the large GEMMs of the SCONE networks
// (the SconeOcc head: lin1 1344 -> 512, lin2, xe2, xe3 over all Q queries; SconeOcc.py:320-347).
//
// Same contract as linear_kernel<NT> (nn_kernels.hip): Y = act(X W^T + bias (+ row_bias)) (+ R), fp32 in / fp32 out.
// Every fp32 operand is split EXACTLY into three bf16 pieces while its tile is staged into LDS (x = hi + mid + lo,
// lp_split.h) and the product keeps hh, hm, mh, mm, hl, lh on v_mfma_f32_32x32x16_bf16 with fp32 accumulation: the
// result is fp32-class (dropped terms <= 2^-24 relative), at 6 x 32 matrix-pipe cycles per 32x32x16 block instead of the
// 8 x 64 of v_mfma_f32_32x32x2_f32.
//
// Block = 4 waves = 128 rows x 128 columns (L3G_NT = 4 column tiles per wave), K in chunks of 32.  LDS per chunk: A planes
// 3 x 128 x 32 bf16 = 24 KB + W planes 24 KB = 48 KB -> three blocks per CU: the other blocks' MFMA phases cover this
// block's split/stage phase and the LDS latency in front of its MFMA groups (measured on the SconeOcc head at Q = 100k,
// us: lin1 1047 / lin2 240 / xe3 263 / xe2 121 against 1065 / 260 / 366 / 155 with 256-column blocks, two per CU).  Plane rows are 64 B; 16-byte chunks are XOR-swizzled by (row >> 2) & 3 so the fragment reads
// (lane = row, one ds_read_b128 per plane) are conflict-free.  The next chunk's global loads are issued before the
// MFMA phase and split + written after it.
#include "lp_split.h"

namespace mcr {

constexpr int L3G_BM = 128, L3G_BK = 32;
// column tiles per wave: 4 (128 columns, 48 KB LDS, 3 blocks/CU) for the big layers; 2 and 1 (64 / 32 columns) give a 2048-row
// problem enough blocks.  The tile width only groups independent accumulators: every output element sees the same k order and the
// same six products per k16 step whatever L3G_NT is, so the result does not depend on it (nor on M).

// uint4 index of chunk c (8 bf16) of row `row` in a plane image [rows][4 chunks]
__device__ __forceinline__ int l3g_chunk(int row, int c) { return row * 4 + (c ^ ((row >> 2) & 3)); }

template <int L3G_NT>
__global__ __launch_bounds__(256, L3G_NT == 4 ? 3 : 4) void linear3_kernel(const float* __restrict__ X, long long ldx, const float* __restrict__ W,
                                                        long long ldw, const float* __restrict__ bias,
                                                        const float* __restrict__ row_bias, long long rows_per_group,
                                                        const float* __restrict__ R, long long ldr, float* __restrict__ Y,
                                                        long long ldy, long long M, int N, int K, int act,
                                                        const int* __restrict__ row_group, int xcd_order) {
    constexpr int L3G_BN = 32 * L3G_NT;
    __shared__ __attribute__((aligned(16))) uint4 As[3][L3G_BM * 4];
    __shared__ __attribute__((aligned(16))) uint4 Bs[3][L3G_BN * 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // XCD-aware block order (1-D grid, padded to groups of 8 row blocks): workgroup b runs on XCD b % 8, and the column blocks of one
    // 128-row block -- which read the SAME activation rows -- take consecutive slots of ONE XCD: the rows come from HBM once and
    // from that XCD's L2 afterwards
    const int ncb = (N + L3G_BN - 1) / L3G_BN;
    long long m0;
    int n0;
    if (xcd_order) {
        const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
        m0 = ((long long)(q / ncb) * 8 + xcd) * L3G_BM;
        n0 = (q % ncb) * L3G_BN;
    } else {                                               // row block fastest (few row blocks: one 2048-token sequence)
        const long long mb = (M + L3G_BM - 1) / L3G_BM;
        m0 = (long long)(blockIdx.x % mb) * L3G_BM;
        n0 = (int)(blockIdx.x / mb) * L3G_BN;
    }
    if (m0 >= M) return;
    const int i = lane & 31, h = lane >> 5;

    f32x16 acc[L3G_NT];
#pragma unroll
    for (int t = 0; t < L3G_NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // staging: a thread owns 16-byte plane chunks = 8 consecutive k of one row: 2 chunks of A, 4 of W per K-chunk
    constexpr int WR = (L3G_NT + 1) / 2;           // W chunk groups of 256 threads (NT = 1: the first 128 threads of one group)
    float4 ra[2][2], rb[WR][2];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int idx = tid + r * 256, row = idx >> 2, c = idx & 3;
            ra[r][0] = ra[r][1] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m0 + row < M && k0 + c * 8 < K) {
                const float* p = X + (m0 + row) * ldx + k0 + c * 8;
                ra[r][0] = *reinterpret_cast<const float4*>(p);
                ra[r][1] = *reinterpret_cast<const float4*>(p + 4);
            }
        }
#pragma unroll
        for (int r = 0; r < WR; ++r) {
            const int idx = tid + r * 256, row = idx >> 2, c = idx & 3;
            rb[r][0] = rb[r][1] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < L3G_BN && n0 + row < N && k0 + c * 8 < K) {
                const float* p = W + (long long)(n0 + row) * ldw + k0 + c * 8;
                rb[r][0] = *reinterpret_cast<const float4*>(p);
                rb[r][1] = *reinterpret_cast<const float4*>(p + 4);
            }
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < K; k0 += L3G_BK) {
        // ---- split the staged chunk into planes and write it to LDS ----
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int idx = tid + r * 256, row = idx >> 2, c = idx & 3;
            const Split3 sp = split8(ra[r][0], ra[r][1]);
            As[0][l3g_chunk(row, c)] = sp.hi; As[1][l3g_chunk(row, c)] = sp.mid; As[2][l3g_chunk(row, c)] = sp.lo;
        }
#pragma unroll
        for (int r = 0; r < WR; ++r) {
            const int idx = tid + r * 256, row = idx >> 2, c = idx & 3;
            if (row < L3G_BN) {
                const Split3 sp = split8(rb[r][0], rb[r][1]);
                Bs[0][l3g_chunk(row, c)] = sp.hi; Bs[1][l3g_chunk(row, c)] = sp.mid; Bs[2][l3g_chunk(row, c)] = sp.lo;
            }
        }
        __syncthreads();
        if (k0 + L3G_BK < K) fetch(k0 + L3G_BK);       // in flight during the MFMA phase
        // ---- MFMA over the chunk: two k16 steps ----
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int ca = l3g_chunk(wave * 32 + i, 2 * s + h);
            const uint4 a_hi = As[0][ca], a_mid = As[1][ca], a_lo = As[2][ca];
#pragma unroll
            for (int t = 0; t < L3G_NT; ++t) {
                const int cb = l3g_chunk(t * 32 + i, 2 * s + h);
                const uint4 b_hi = Bs[0][cb], b_mid = Bs[1][cb], b_lo = Bs[2][cb];
                acc[t] = mfma_bf(a_lo, b_hi, acc[t]);      // smallest terms first
                acc[t] = mfma_bf(a_hi, b_lo, acc[t]);
                acc[t] = mfma_bf(a_mid, b_mid, acc[t]);
                acc[t] = mfma_bf(a_mid, b_hi, acc[t]);
                acc[t] = mfma_bf(a_hi, b_mid, acc[t]);
                acc[t] = mfma_bf(a_hi, b_hi, acc[t]);
            }
        }
        __syncthreads();
    }
    // ---- epilogue: C/D layout col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) ----
#pragma unroll
    for (int t = 0; t < L3G_NT; ++t) {
        const int n = n0 + t * 32 + i;
        if (n >= N) continue;
        const float bn = bias ? bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long long m = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (m >= M) continue;
            float y = acc[t][r] + bn;
            if (row_bias) y += row_bias[(row_group ? (long long)row_group[m] : m / rows_per_group) * N + n];
            if (act == ACT_GELU) y = 0.5f * y * (1.f + erff(y * 0.70710678118654752440f));
            if (R) y += R[m * ldr + n];
            Y[m * ldy + n] = y;
        }
    }
}

// true when the split-precision kernel applies: 16-byte aligned rows of 8-float groups, a problem big enough to fill the chip
constexpr int L3G_BN_BIG = 128;
bool linear3_shape_ok(const float* X, int64_t ldx, const float* W, int64_t ldw, int N, int K) {
    auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    return K % 8 == 0 && K >= 64 && ldx % 4 == 0 && ldw % 4 == 0 && al(X) && al(W) && N >= 128;
}
bool linear3_applicable(const float* X, int64_t ldx, const float* W, int64_t ldw, int64_t M, int N, int K) {
    return linear3_shape_ok(X, ldx, W, ldw, N, K) && cdiv(M, L3G_BM) * cdiv(N, L3G_BN_BIG) >= 256;
}

void launch_linear3(hipStream_t s, const float* X, int64_t ldx, const float* W, const float* bias, const float* R, int64_t ldr,
                    float* Y, int64_t ldy, int64_t M, int N, int K, int act, const float* row_bias, int64_t rows_per_group,
                    int64_t ldw, const int* row_group) {
    const int64_t mb = cdiv(M, L3G_BM);
    const long long rpg = rows_per_group > 0 ? rows_per_group : 1;
    // widest column tile that still gives the chip ~2 blocks per CU (performance only: see the note on L3G_NT above); the short-K
    // GEMMs of the encoders (K = 256 / 512: eight or sixteen chunks between an exposed first load and the epilogue) run better on
    // 64-column blocks, four per CU, than on 128-column ones, three per CU (SconeVis on 30 x 2048 tokens: 3.80 vs 4.11 ms); the
    // head's K = 1344 the other way round.  MCR_L3_SHORTK=0: 128-column blocks whatever K is (A/B)
    static const bool shortk = []() { const char* e = getenv("MCR_L3_SHORTK"); return !(e && e[0] == '0'); }();
    int nt = shortk && K <= 512 ? 2 : 4;
    while (nt > 1 && mb * cdiv(N, nt * 32) < 512) nt >>= 1;
    // XCD-aware block order for the batch-sized launches (SconeVis on 30 x 2048 tokens: 3.77 -> 3.53 ms); MCR_L3_XCD=0: never (A/B)
    static const bool xcd_on = []() { const char* e = getenv("MCR_L3_XCD"); return !(e && e[0] == '0'); }();
    static const int xcd_min = []() { const char* e = getenv("MCR_L3_XCD_MIN"); return e ? atoi(e) : 64; }();
    const int xo = xcd_on && mb >= xcd_min;
#define MCR_L3(NT)                                                                                                              \
    hipLaunchKernelGGL((linear3_kernel<NT>), dim3((unsigned)((xo ? cdiv(mb, 8) * 8 : mb) * cdiv(N, NT * 32))), dim3(256), 0, s, X, \
                       (long long)ldx, W, (long long)ldw, bias, row_bias, rpg, R, (long long)ldr, Y, (long long)ldy,            \
                       (long long)M, N, K, act, row_group, xo)
    if (nt == 4) MCR_L3(4);
    else if (nt == 2) MCR_L3(2);
    else MCR_L3(1);
#undef MCR_L3
}

}  // namespace mcr
