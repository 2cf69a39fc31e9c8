// K1 — brute-force k-nearest surface points per query, for gfx950.
//
// Replaces macarons/utility/utils.py:1497-1509 get_knn_points  (torch.cdist -> topk(largest=False) ->
// pytorch3d knn_gather, which materialises the dense [B,Q,M] distance matrix: 4 GB at Q=100k, M=10k) and the
// offset step of macarons/networks/SconeOcc.py:297-298 (neighbours minus the query).
//
// One lane owns one query and keeps its k best (d2, index) pairs sorted in VGPRs; surface points stream
// through LDS in tiles (one broadcast ds_read_b128 per candidate per wave).  Convention (shared with
// oracle/knn.py, see there why the reference's own tie order is unspecified):
//   d2 = (dx*dx + dy*dy) + dz*dz in fp32 with every product and sum rounded (no FMA contraction),
//   ascending by (d2, index): ties go to the lower index;  dists = sqrt(d2), correctly rounded.
#include "common.h"

namespace mcr {

constexpr int KNN_BLOCK = 256;
constexpr int KNN_TILE = 2048;     // surface points per LDS tile (32 KB as float4)

template <int K, bool OFFSETS>
__global__ __launch_bounds__(KNN_BLOCK) void knn_kernel(const float* __restrict__ X, const float* __restrict__ pc,
                                                        long long* __restrict__ out_idx, float* __restrict__ out_dist,
                                                        float* __restrict__ out_pts, int Q, int M) {
    __shared__ float4 s_pc[KNN_TILE];
    const int b = blockIdx.y;
    const int q = blockIdx.x * KNN_BLOCK + threadIdx.x;
    const bool valid = q < Q;
    const float* xq = X + ((size_t)b * Q + (valid ? q : Q - 1)) * 3;
    const float qx = xq[0], qy = xq[1], qz = xq[2];
    const float* pcb = pc + (size_t)b * M * 3;

    float bd[K];
    int bi[K];
#pragma unroll
    for (int j = 0; j < K; ++j) { bd[j] = __builtin_inff(); bi[j] = 0x7fffffff; }

    for (int t0 = 0; t0 < M; t0 += KNN_TILE) {
        const int nt = min(KNN_TILE, M - t0);
        __syncthreads();
        for (int i = threadIdx.x; i < nt; i += KNN_BLOCK) {
            const float* p = pcb + (size_t)(t0 + i) * 3;
            s_pc[i] = make_float4(p[0], p[1], p[2], 0.f);
        }
        __syncthreads();
        for (int i = 0; i < nt; ++i) {
            const float4 p = s_pc[i];
            const float dx = qx - p.x, dy = qy - p.y, dz = qz - p.z;
            const float d2 = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
            if (d2 < bd[K - 1]) {                    // strict: an equal distance never displaces an earlier index
                bd[K - 1] = d2;
                bi[K - 1] = t0 + i;
#pragma unroll
                for (int j = K - 1; j > 0; --j) {
                    const bool sw = bd[j] < bd[j - 1];
                    const float dlo = sw ? bd[j] : bd[j - 1], dhi = sw ? bd[j - 1] : bd[j];
                    const int ilo = sw ? bi[j] : bi[j - 1], ihi = sw ? bi[j - 1] : bi[j];
                    bd[j - 1] = dlo; bd[j] = dhi;
                    bi[j - 1] = ilo; bi[j] = ihi;
                }
            }
        }
    }
    if (!valid) return;
    const size_t o = ((size_t)b * Q + q) * K;
#pragma unroll
    for (int j = 0; j < K; ++j) {
        out_idx[o + j] = (long long)bi[j];
        out_dist[o + j] = __fsqrt_rn(bd[j]);
        const float* p = pcb + (size_t)bi[j] * 3;
        const float ox = OFFSETS ? p[0] - qx : p[0], oy = OFFSETS ? p[1] - qy : p[1], oz = OFFSETS ? p[2] - qz : p[2];
        out_pts[(o + j) * 3 + 0] = ox;
        out_pts[(o + j) * 3 + 1] = oy;
        out_pts[(o + j) * 3 + 2] = oz;
    }
}

template <int K>
static void launch_knn(bool offsets, dim3 grid, hipStream_t s, const float* X, const float* pc, long long* idx, float* dist,
                       float* pts, int Q, int M) {
    if (offsets)
        hipLaunchKernelGGL((knn_kernel<K, true>), grid, dim3(KNN_BLOCK), 0, s, X, pc, idx, dist, pts, Q, M);
    else
        hipLaunchKernelGGL((knn_kernel<K, false>), grid, dim3(KNN_BLOCK), 0, s, X, pc, idx, dist, pts, Q, M);
}

}  // namespace mcr

using namespace mcr;

extern "C" int mcr_knn_points(const float* X, const float* pc, int64_t* idx, float* dists, float* pts, int64_t B,
                              int64_t Q, int64_t M, int k, int subtract_query, void* stream) {
    MCR_REQUIRE(X && pc && idx && dists && pts, "mcr_knn_points: null pointer");
    MCR_REQUIRE(B > 0 && Q > 0 && M > 0, "mcr_knn_points: empty problem B=%ld Q=%ld M=%ld", (long)B, (long)Q, (long)M);
    MCR_REQUIRE(k <= M, "mcr_knn_points: k=%d exceeds the number of points M=%ld (torch.topk would raise)", k, (long)M);
    MCR_REQUIRE(B <= 65535 && Q < (1ll << 31) && M < (1ll << 31), "mcr_knn_points: problem too large");
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((unsigned)cdiv(Q, KNN_BLOCK), (unsigned)B);
    long long* i64 = (long long*)idx;
    switch (k) {
        case 1: launch_knn<1>(subtract_query, grid, s, X, pc, i64, dists, pts, (int)Q, (int)M); break;
        case 4: launch_knn<4>(subtract_query, grid, s, X, pc, i64, dists, pts, (int)Q, (int)M); break;
        case 8: launch_knn<8>(subtract_query, grid, s, X, pc, i64, dists, pts, (int)Q, (int)M); break;
        case 16: launch_knn<16>(subtract_query, grid, s, X, pc, i64, dists, pts, (int)Q, (int)M); break;
        case 32: launch_knn<32>(subtract_query, grid, s, X, pc, i64, dists, pts, (int)Q, (int)M); break;
        default: MCR_REQUIRE(false, "mcr_knn_points: k must be one of 1,4,8,16,32 (got %d)", k);
    }
    MCR_LAUNCH_CHECK("knn_kernel");
    return 0;
}
