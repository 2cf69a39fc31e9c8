// K4-split/fp16 — nn.Linear on the two-term fp16 split (three MFMAs per fp32 product) for the large GEMMs of the SconeOcc
// head (lin1 1344 -> 512, lin2, xe2, xe3 over all Q queries; SconeOcc.py:320-347).  Same contract as linear_kernel<NT>:
// Y = act(X W^T + bias (+ row_bias)) (+ R), fp32 in / fp32 out.
//
// Numerics as in local_pct6.hip: x = hi + lo with hi = fp16(x), lo = fp16(x - hi) (22 significant bits); a product keeps
// x_lo w_hi + x_hi w_lo + x_hi w_hi on v_mfma_f32_32x32x16_f16 with fp32 accumulation.  The weights are split ONCE per call by
// split_weights_kernel (times 2^8, so the low plane of every weight above 5e-4 is a normal fp16 number; undone in the epilogue)
// into planes [2][N][K/8] -- the bf16 kernel (linear3.hip) re-split the same W tile in each of its 782 row blocks -- and the
// activations while their tile is staged into LDS.  Range: |x| < 65504, |w| < 255 (linear3.hip covers the whole fp32 range).
//
// Block = 4 waves = 128 rows x 128 columns, K in chunks of 32; LDS per chunk: X planes 2 x 128 x 32 fp16 = 16 KB + W planes
// 16 KB = 32 KB -> three blocks per CU.  16-byte chunks are XOR-swizzled by (row >> 2) & 3: conflict-free fragment reads.
#include "lp_split.h"

namespace mcr {

constexpr int LH_BM = 128, LH_BK = 32, LH_NT = 4, LH_BN = 32 * LH_NT;
constexpr float LH_WSCALE = 256.0f, LH_WSCALE_INV = 1.0f / 256.0f;

__device__ __forceinline__ int lh_chunk(int row, int c) { return row * 4 + (c ^ ((row >> 2) & 3)); }

// W [N, K] (row stride ldw) -> Wp[plane][n][K/8] (uint4 = 8 fp16) of W * 2^8
__global__ void split_weights_kernel(const float* __restrict__ W, long long ldw, uint4* __restrict__ Wp, int N, int K8) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long long)N * K8) return;
    const int n = (int)(gid / K8), c = (int)(gid % K8);
    const float* p = W + (long long)n * ldw + c * 8;
    float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    a.x *= LH_WSCALE; a.y *= LH_WSCALE; a.z *= LH_WSCALE; a.w *= LH_WSCALE;
    b.x *= LH_WSCALE; b.y *= LH_WSCALE; b.z *= LH_WSCALE; b.w *= LH_WSCALE;
    const Split2 s = split8h(a, b);
    Wp[gid] = s.hi;
    Wp[(long long)N * K8 + gid] = s.lo;
}

__global__ __launch_bounds__(256, 3) void linear3h_kernel(const float* __restrict__ X, long long ldx, const uint4* __restrict__ Wp,
                                                         const float* __restrict__ bias, const float* __restrict__ row_bias,
                                                         long long rows_per_group, const float* __restrict__ R, long long ldr,
                                                         float* __restrict__ Y, long long ldy, long long M, int N, int K, int act,
                                                         float wscale_inv, const int* __restrict__ row_group) {
    __shared__ __attribute__((aligned(16))) uint4 As[2][LH_BM * 4];
    __shared__ __attribute__((aligned(16))) uint4 Bs[2][LH_BN * 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long m0 = (long long)blockIdx.x * LH_BM;
    const int n0 = blockIdx.y * LH_BN;
    const int i = lane & 31, h = lane >> 5;
    const int K8 = K >> 3;
    const long long plane = (long long)N * K8;

    f32x16 acc[LH_NT];
#pragma unroll
    for (int t = 0; t < LH_NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // staging: a thread owns 16-byte plane chunks = 8 consecutive k of one row: 2 chunks of X (split here), 2 of W (both planes, ready)
    float4 ra[2][2];
    uint4 rb[2][2];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int idx = tid + r * 256, row = idx >> 2, c = idx & 3;
            ra[r][0] = ra[r][1] = make_float4(0.f, 0.f, 0.f, 0.f);
            rb[r][0] = rb[r][1] = make_uint4(0, 0, 0, 0);
            if (m0 + row < M && k0 + c * 8 < K) {
                const float* p = X + (m0 + row) * ldx + k0 + c * 8;
                ra[r][0] = *reinterpret_cast<const float4*>(p);
                ra[r][1] = *reinterpret_cast<const float4*>(p + 4);
            }
            if (n0 + row < N && k0 + c * 8 < K) {
                const long long g = (long long)(n0 + row) * K8 + (k0 >> 3) + c;
                rb[r][0] = Wp[g];
                rb[r][1] = Wp[plane + g];
            }
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < K; k0 += LH_BK) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int idx = tid + r * 256, row = idx >> 2, c = idx & 3;
            const Split2 sp = split8h(ra[r][0], ra[r][1]);
            As[0][lh_chunk(row, c)] = sp.hi; As[1][lh_chunk(row, c)] = sp.lo;
            Bs[0][lh_chunk(row, c)] = rb[r][0]; Bs[1][lh_chunk(row, c)] = rb[r][1];
        }
        __syncthreads();
        if (k0 + LH_BK < K) fetch(k0 + LH_BK);         // in flight during the MFMA phase
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int ca = lh_chunk(wave * 32 + i, 2 * s + h);
            const uint4 a_hi = As[0][ca], a_lo = As[1][ca];
            uint4 b_hi[LH_NT], b_lo[LH_NT];
#pragma unroll
            for (int t = 0; t < LH_NT; ++t) {
                const int cb = lh_chunk(t * 32 + i, 2 * s + h);
                b_hi[t] = Bs[0][cb]; b_lo[t] = Bs[1][cb];
            }
#pragma unroll
            for (int t = 0; t < LH_NT; ++t) acc[t] = mfma_h(a_lo, b_hi[t], acc[t]);      // smallest terms first; 4 independent chains
#pragma unroll
            for (int t = 0; t < LH_NT; ++t) acc[t] = mfma_h(a_hi, b_lo[t], acc[t]);
#pragma unroll
            for (int t = 0; t < LH_NT; ++t) acc[t] = mfma_h(a_hi, b_hi[t], acc[t]);
        }
        __syncthreads();
    }
    // ---- epilogue: C/D layout col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) ----
#pragma unroll
    for (int t = 0; t < LH_NT; ++t) {
        const int n = n0 + t * 32 + i;
        if (n >= N) continue;
        const float bn = bias ? bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long long m = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (m >= M) continue;
            float y = fmaf(acc[t][r], wscale_inv, bn);
            if (row_bias) y += row_bias[(row_group ? (long long)row_group[m] : m / rows_per_group) * N + n];
            if (act == ACT_GELU) y = 0.5f * y * (1.f + erff(y * 0.70710678118654752440f));
            if (R) y += R[m * ldr + n];
            Y[m * ldy + n] = y;
        }
    }
}

bool linear3h_applicable(const float* X, int64_t ldx, const float* W, int64_t ldw, int64_t M, int N, int K) {
    auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    return K % 8 == 0 && K >= 64 && ldx % 4 == 0 && ldw % 4 == 0 && al(X) && al(W) && N >= 128 &&
           cdiv(M, LH_BM) * cdiv(N, LH_BN) >= 256;
}
// W [N, K] (+ bias [N]) -> zero-padded planes Wp[plane][Np][Kp/8] of W * 2^8 and bias_p [Np]: a layer whose width is not a multiple
// of the planes GEMM's granules (the embeddings' 125 / 126 inner width; Np % 4 == 0, Kp % 32 == 0) joins it with exact zeros
__global__ void pad_weights_kernel(const float* __restrict__ W, long long ldw, const float* __restrict__ bias, uint4* __restrict__ Wp,
                                   float* __restrict__ bias_p, int N, int K, int Np, int Kp8) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long long)Np * Kp8) return;
    const int n = (int)(gid / Kp8), c = (int)(gid % Kp8);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (n < N && c * 8 + e < K) ? W[(long long)n * ldw + c * 8 + e] * LH_WSCALE : 0.f;
    const Split2 sp = split8h(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]));
    Wp[gid] = sp.hi;
    Wp[(long long)Np * Kp8 + gid] = sp.lo;
    if (c == 0) bias_p[n] = (bias && n < N) ? bias[n] : 0.f;
}
void launch_pad_weights(hipStream_t s, const float* W, int64_t ldw, const float* bias, void* planes, float* bias_p, int N, int K, int Np, int Kp) {
    hipLaunchKernelGGL(pad_weights_kernel, dim3((unsigned)cdiv((int64_t)Np * (Kp / 8), 256)), dim3(256), 0, s, W, (long long)ldw, bias,
                       reinterpret_cast<uint4*>(planes), bias_p, N, K, Np, Kp / 8);
}

size_t linear3h_planes_bytes(int N, int K) { return ((size_t)2 * N * (K / 8) * sizeof(uint4) + 255) & ~(size_t)255; }

void launch_split_weights(hipStream_t s, const float* W, int64_t ldw, void* planes, int N, int K) {
    const int K8 = K / 8;
    hipLaunchKernelGGL(split_weights_kernel, dim3((unsigned)cdiv((int64_t)N * K8, 256)), dim3(256), 0, s, W, (long long)ldw,
                       reinterpret_cast<uint4*>(planes), N, K8);
}

// planes: scratch of linear3h_planes_bytes(N, K) bytes (16-byte aligned) that receives the split weights -- or, with
// presplit_inv_scale > 0, planes the HOST already built (networks/packing.py: pack_head_planes, cached per parameter version;
// fp16 hi/lo of W * 2^e with e chosen per matrix like the local-transformer blobs): no split launch, the epilogue multiplies by
// presplit_inv_scale = 2^-e.
void launch_linear3h(hipStream_t s, const float* X, int64_t ldx, const float* W, int64_t ldw, void* planes, const float* bias,
                     const float* R, int64_t ldr, float* Y, int64_t ldy, int64_t M, int N, int K, int act, const float* row_bias,
                     int64_t rows_per_group, float presplit_inv_scale, const int* row_group) {
    const int K8 = K / 8;
    uint4* Wp = reinterpret_cast<uint4*>(planes);
    if (!(presplit_inv_scale > 0.f))
        hipLaunchKernelGGL(split_weights_kernel, dim3((unsigned)cdiv((int64_t)N * K8, 256)), dim3(256), 0, s, W, (long long)ldw, Wp, N, K8);
    dim3 grid((unsigned)cdiv(M, LH_BM), (unsigned)cdiv(N, LH_BN));
    hipLaunchKernelGGL(linear3h_kernel, grid, dim3(256), 0, s, X, (long long)ldx, Wp, bias, row_bias,
                       (long long)(rows_per_group > 0 ? rows_per_group : 1), R, (long long)ldr, Y, (long long)ldy, (long long)M, N,
                       K, act, presplit_inv_scale > 0.f ? presplit_inv_scale : LH_WSCALE_INV, row_group);
}

}  // namespace mcr
