// K1-sorted — the brute-force k-nearest-neighbour scan of knn.hip on SPATIALLY SORTED queries and candidates: same result, bit for
// bit, for large problems (replaces macarons/utility/utils.py:1497-1509 get_knn_points + the offset step of SconeOcc.py:297-298
// exactly like knn.hip does).
//
// Why sorting pays.  knn.hip is bound by its selection machinery, not by the distances: with unrelated queries in a wave and the
// cloud in arbitrary order, SOME lane of the 64 accepts one of the 4 candidates of a batch 60-99 % of the time, so nearly every
// batch takes the slow path (predicated queue pushes, flushes that run as long as their busiest lane).  When the 64 queries of a
// wave are neighbours in space (queries grouped by cell of a 16^3 grid, cells in Morton order) and the candidates stream in the
// same order, the lanes accept in the same few stretches of the stream and reject everything else TOGETHER: a replay of the
// kernel's decisions on the bench clouds gives 4 % of the batches on the slow path (60 % unsorted) for 150-200 instead of 140
// insertions in the busiest lane.  Unlike the shelved wave-uniform tile culling (tools/experiments/knn_culled.hip) nothing is
// skipped and nothing is staged per wave: the block-shared LDS tiles, the broadcast reads and the 2-way candidate split of
// knn.hip stay as they are.
//
// Preparation (counting sort: bounding box, histogram, scan, scatter; ~15 us per array, no host synchronisation): the queries
// once per call (a permutation: results are written back at the original query index), every cloud once (x, y, z, original index
// as float4 -- one 16-byte load per candidate instead of three scalar ones).
//
// Exactness: d2 = (dx*dx + dy*dy) + dz*dz with every product and sum rounded (built with -ffp-contract=off, like knn.hip and
// oracle/knn.py); candidates no longer arrive in index order, so the list is kept ascending by (d2, ORIGINAL index) with
// lexicographic comparisons and a candidate stays one while d2 <= the current k-th distance (an equal distance with a lower
// index still wins) -- the result is the same (d2, index)-lexicographic top-k whatever the order.
// MCR_HIPCC_FLAGS: -ffp-contract=off
#include "common.h"
#include "nn_kernels.h"

namespace mcr {

constexpr int KS_G = 16, KS_CELLS = KS_G * KS_G * KS_G;     // grid of the counting sort
constexpr int KS_BLOCK = 256;
constexpr int KS_WAVES = KS_BLOCK / MCR_WAVE;
constexpr int KS_SPLIT = 2;                        // waves sharing one 64-query tile
constexpr int KS_QT = KS_WAVES / KS_SPLIT;         // query tiles per workgroup
constexpr int KS_TILE = 1536;                      // candidates per LDS tile (24 KB as float4); multiple of 16
constexpr int KS_QCAP = 8;                         // per-lane queue of accepted candidates (LDS, [slot][thread])
constexpr int KS_SEED = 128;                       // candidates on either side of the block's own cell that are scanned first

// ---- preparation ------------------------------------------------------------------------------------------------------------
// bounding box of n points (row stride 3 floats): one block; box[0..2] = min, box[3..5] = max; also clears the cell counters
__global__ __launch_bounds__(1024) void ks_bbox_kernel(const float* __restrict__ p, int n, float* __restrict__ box, int* __restrict__ count) {
    __shared__ float s[6][16];
    for (int i = threadIdx.x; i < KS_CELLS; i += 1024) count[i] = 0;
    float mn[3] = {3e38f, 3e38f, 3e38f}, mx[3] = {-3e38f, -3e38f, -3e38f};
    for (int i = threadIdx.x; i < n; i += 1024)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v = p[(size_t)i * 3 + c];
            mn[c] = fminf(mn[c], v);
            mx[c] = fmaxf(mx[c], v);
        }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            mn[c] = fminf(mn[c], __shfl_xor(mn[c], o, 64));
            mx[c] = fmaxf(mx[c], __shfl_xor(mx[c], o, 64));
        }
        if ((threadIdx.x & 63) == 0) { s[c][threadIdx.x >> 6] = mn[c]; s[3 + c][threadIdx.x >> 6] = mx[c]; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        float v = s[threadIdx.x][0];
        for (int w = 1; w < 16; ++w) v = threadIdx.x < 3 ? fminf(v, s[threadIdx.x][w]) : fmaxf(v, s[threadIdx.x][w]);
        box[threadIdx.x] = v;
    }
}

__device__ __forceinline__ unsigned ks_spread(unsigned v) {           // 4 bits -> every third bit
    v = (v | (v << 8)) & 0x0000F00Fu;
    v = (v | (v << 4)) & 0x000C30C3u;
    v = (v | (v << 2)) & 0x00249249u;
    return v;
}
// Morton index of the point's grid cell (any cell is correct: the cells only order the work; NaN / degenerate extents -> cell 0)
__device__ __forceinline__ int ks_cell(const float* __restrict__ p, const float* __restrict__ box) {
    unsigned c[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float ext = box[3 + a] - box[a];
        const float t = ext > 0.f ? (p[a] - box[a]) / ext * (float)KS_G : 0.f;
        c[a] = (unsigned)(t >= 0.f ? (t < (float)KS_G ? (int)t : KS_G - 1) : 0);
    }
    return (int)(ks_spread(c[0]) | (ks_spread(c[1]) << 1) | (ks_spread(c[2]) << 2));
}

__global__ void ks_hist_kernel(const float* __restrict__ p, int n, const float* __restrict__ box, int* __restrict__ code,
                               int* __restrict__ count) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c = ks_cell(p + (size_t)i * 3, box);
    code[i] = c;
    atomicAdd(count + c, 1);
}

// exclusive scan of the KS_CELLS counters in place (one block of 1024 threads x 4 cells)
__global__ __launch_bounds__(1024) void ks_scan_kernel(int* __restrict__ count) {
    __shared__ int s[1024];
    int v[4], t = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = count[threadIdx.x * 4 + e]; t += v[e]; }
    s[threadIdx.x] = t;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int u = threadIdx.x >= o ? s[threadIdx.x - o] : 0;
        __syncthreads();
        s[threadIdx.x] += u;
        __syncthreads();
    }
    int run = s[threadIdx.x] - t;
#pragma unroll
    for (int e = 0; e < 4; ++e) { count[threadIdx.x * 4 + e] = run; run += v[e]; }
}

// perm[sorted position] = original index (queries) and / or sorted[sorted position] = (x, y, z, original index) (clouds); the
// order inside a cell is arbitrary (atomics): the result does not depend on it
__global__ void ks_scatter_kernel(const float* __restrict__ p, const int* __restrict__ code, int n, int* __restrict__ cursor,
                                  int* __restrict__ perm, float4* __restrict__ sorted) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int pos = atomicAdd(cursor + code[i], 1);
    if (perm) perm[pos] = i;
    if (sorted) sorted[pos] = make_float4(p[(size_t)i * 3], p[(size_t)i * 3 + 1], p[(size_t)i * 3 + 2], __builtin_bit_cast(float, i));
}

// ---- search -----------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool ks_less(float d, int i, float bd, int bi) { return d < bd || (d == bd && i < bi); }

// Insert (d2, idx) into the list kept ascending by (d2, idx): slot j takes its upper neighbour if that one must move down, the
// new element if it lands here, else keeps its value.
template <int K>
__device__ __forceinline__ void ks_insert(float (&bd)[K], int (&bi)[K], float d2, int idx) {
    bool lands_or_below = ks_less(d2, idx, bd[K - 1], bi[K - 1]);
    if (lands_or_below) {
#pragma unroll
        for (int j = K - 1; j > 0; --j) {
            const bool up_moves = ks_less(d2, idx, bd[j - 1], bi[j - 1]);
            const float nd = up_moves ? bd[j - 1] : d2;
            const int ni = up_moves ? bi[j - 1] : idx;
            bd[j] = lands_or_below ? nd : bd[j];
            bi[j] = lands_or_below ? ni : bi[j];
            lands_or_below = up_moves;
        }
        bd[0] = lands_or_below ? d2 : bd[0];
        bi[0] = lands_or_below ? idx : bi[0];
    }
}

__device__ __forceinline__ float ks_sqrt_cr(float x) {              // correctly rounded sqrt, as in knn.hip
    if (!(x > 0.f)) return x;
    float y = __builtin_amdgcn_sqrtf(x);
    const float up = __builtin_bit_cast(float, __builtin_bit_cast(int, y) + 1);
    const float dn = __builtin_bit_cast(float, __builtin_bit_cast(int, y) - 1);
    const double m_up = 0.5 * ((double)y + (double)up), m_dn = 0.5 * ((double)y + (double)dn);
    const double xd = (double)x;
    if (xd > m_up * m_up) y = up;
    else if (xd < m_dn * m_dn) y = dn;
    return y;
}

__device__ __forceinline__ float ks_d2(float qx, float qy, float qz, const float4 p) {
    const float dx = qx - p.x, dy = qy - p.y, dz = qz - p.z;
    const float xx = dx * dx, yy = dy * dy, zz = dz * dz;
    const float s = xx + yy;
    return s + zz;
}

// grid = ceil(Q / 128); block = 2 query tiles x 2 waves (the two waves of a tile scan every other candidate); a wave owns 64
// consecutive queries of the SORTED order and writes their results at the original indices
template <int K, bool OFFSETS>
__global__ __launch_bounds__(KS_BLOCK) void knn_sorted_kernel(const float* __restrict__ X, const int* __restrict__ qperm,
                                                              const float* __restrict__ pc, const float4* __restrict__ sorted,
                                                              const float* __restrict__ cbox, const int* __restrict__ cell_end,
                                                              long long* __restrict__ out_idx, float* __restrict__ out_dist,
                                                              float* __restrict__ out_pts, int Q, int M) {
    // 40 KB: [tile 24 KB][queue distances 8 KB][queue indices 8 KB]; the first 32 KB are reused as the merge buffer
    __shared__ __attribute__((aligned(16))) char smem[KS_TILE * 16 + 2 * KS_QCAP * KS_BLOCK * 4];
    float4* s_pc = reinterpret_cast<float4*>(smem);
    float* s_qd = reinterpret_cast<float*>(smem + KS_TILE * 16);
    int* s_qi = reinterpret_cast<int*>(smem + KS_TILE * 16 + KS_QCAP * KS_BLOCK * 4);
    static_assert(KS_WAVES * K * MCR_WAVE * 8 <= KS_TILE * 16 + KS_QCAP * KS_BLOCK * 4, "merge buffer does not fit");
    const int lane = threadIdx.x & (MCR_WAVE - 1);
    const int wave = threadIdx.x / MCR_WAVE;
    const int qt = wave / KS_SPLIT, part = wave % KS_SPLIT;
    const int pos = (blockIdx.x * KS_QT + qt) * MCR_WAVE + lane;
    const bool valid = pos < Q;
    const int q = qperm[valid ? pos : Q - 1];
    const float* xq = X + (size_t)q * 3;
    const float qx = xq[0], qy = xq[1], qz = xq[2];

    float bd[K];
    int bi[K];
#pragma unroll
    for (int j = 0; j < K; ++j) { bd[j] = __builtin_inff(); bi[j] = 0x7fffffff; }

    // accepted candidates go through a small per-lane LDS queue and are inserted in batches (see knn.hip); the filter threshold
    // tau is the (possibly stale, hence larger) current k-th distance: never a false reject
    float tau = __builtin_inff();
    int cnt = 0;
    float* q_d = s_qd + threadIdx.x;
    int* q_i = s_qi + threadIdx.x;
    auto flush = [&]() {
        int maxc = cnt;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) maxc = max(maxc, __shfl_xor(maxc, o, 64));
        for (int sidx = 0; sidx < maxc; ++sidx)
            if (sidx < cnt) ks_insert<K>(bd, bi, q_d[sidx * KS_BLOCK], q_i[sidx * KS_BLOCK]);
        tau = bd[K - 1];
        cnt = 0;
    };
    auto push = [&](float d, int idx) {
        if (d <= tau) {                                 // <=: an equal distance with a lower index still displaces the k-th entry
            q_d[cnt * KS_BLOCK] = d;
            q_i[cnt * KS_BLOCK] = idx;
            ++cnt;
        }
    };

    // ---- seed: the 2 x KS_SEED candidates around the block's own grid cell first.  Streaming the sorted cloud front to back
    // approaches a far-away block's neighbourhood monotonically -- every candidate closer than the last, every one of them
    // inserted (445 insertions in the busiest lane of an average wave, 900 in the worst, against 140 unsorted); after this
    // seed the k-th distances are near their final values and the sweep below inserts what the unsorted scan inserts.
    __shared__ int s_seed;
    if (threadIdx.x == 0) {
        const int code = ks_cell(X + (size_t)q * 3, cbox);
        const int pos = code > 0 ? cell_end[code - 1] : 0;           // first candidate of that cell (cell_end = scatter cursors)
        s_seed = max(0, min(((M + 15) & ~15) - 2 * KS_SEED, (pos - KS_SEED) & ~15));
    }
    __syncthreads();
    const int s0 = s_seed, s1 = s0 + 2 * KS_SEED;
    auto scan = [&](int n_pad, int skip0, int skip1, int base) {      // candidates s_pc[0 .. n_pad); positions base + i in [skip0, skip1) are skipped
        for (int i = part; i < n_pad; i += 4 * KS_SPLIT) {
            const int g0 = base + i - part;                          // both waves of a tile test the same 8-aligned position
            if (g0 >= skip0 && g0 < skip1) continue;
            const float4 p0 = s_pc[i], p1 = s_pc[i + KS_SPLIT], p2 = s_pc[i + 2 * KS_SPLIT], p3 = s_pc[i + 3 * KS_SPLIT];
            const float d0 = ks_d2(qx, qy, qz, p0), d1 = ks_d2(qx, qy, qz, p1);
            const float d2 = ks_d2(qx, qy, qz, p2), d3 = ks_d2(qx, qy, qz, p3);
            if (__any(fminf(fminf(d0, d1), fminf(d2, d3)) <= tau)) {
                push(d0, __builtin_bit_cast(int, p0.w));
                push(d1, __builtin_bit_cast(int, p1.w));
                push(d2, __builtin_bit_cast(int, p2.w));
                push(d3, __builtin_bit_cast(int, p3.w));
                if (__any(cnt > KS_QCAP - 4)) flush();
            }
        }
    };
    const float4 far = make_float4(3e18f, 3e18f, 3e18f, __builtin_bit_cast(float, 0x7fffffff));     // d2 = +inf: never accepted
    if (s0 >= 0 && M > 2 * KS_SEED) {
        for (int i = threadIdx.x; i < 2 * KS_SEED; i += KS_BLOCK) s_pc[i] = s0 + i < M ? sorted[s0 + i] : far;
        __syncthreads();
        scan(2 * KS_SEED, 0, 0, s0);
        flush();
    }
    const bool seeded = s0 >= 0 && M > 2 * KS_SEED;
    for (int t0 = 0; t0 < M; t0 += KS_TILE) {
        const int nt = min(KS_TILE, M - t0);
        const int nt_pad = (nt + 15) & ~15;
        __syncthreads();
        for (int i = threadIdx.x; i < nt_pad; i += KS_BLOCK) s_pc[i] = i < nt ? sorted[t0 + i] : far;
        __syncthreads();
        scan(nt_pad, seeded ? s0 : 0, seeded ? s1 : 0, t0);
    }
    flush();
    // ---- merge of the tile's KS_SPLIT sorted lists (lexicographic on (d2, index)) ----
    __syncthreads();
    float* m_d = reinterpret_cast<float*>(smem);                        // [wave][K][lane]
    int* m_i = reinterpret_cast<int*>(smem) + KS_WAVES * K * MCR_WAVE;
#pragma unroll
    for (int j = 0; j < K; ++j) {
        m_d[(wave * K + j) * MCR_WAVE + lane] = bd[j];
        m_i[(wave * K + j) * MCR_WAVE + lane] = bi[j];
    }
    __syncthreads();
    if (part != 0 || !valid) return;
    m_d += qt * KS_SPLIT * K * MCR_WAVE;
    m_i += qt * KS_SPLIT * K * MCR_WAVE;
    int head[KS_SPLIT];
#pragma unroll
    for (int w = 0; w < KS_SPLIT; ++w) head[w] = 0;
    const size_t o = (size_t)q * K;
    for (int j = 0; j < K; ++j) {
        float best_d = __builtin_inff();
        int best_i = 0x7fffffff, best_w = 0;
#pragma unroll
        for (int w = 0; w < KS_SPLIT; ++w) {
            const int h = head[w] < K ? head[w] : K - 1;
            const float d = head[w] < K ? m_d[(w * K + h) * MCR_WAVE + lane] : __builtin_inff();
            const int id = head[w] < K ? m_i[(w * K + h) * MCR_WAVE + lane] : 0x7fffffff;
            const bool better = d < best_d || (d == best_d && id < best_i);
            best_d = better ? d : best_d;
            best_i = better ? id : best_i;
            best_w = better ? w : best_w;
        }
#pragma unroll
        for (int w = 0; w < KS_SPLIT; ++w) head[w] += (best_w == w) ? 1 : 0;
        if (out_idx) out_idx[o + j] = (long long)best_i;
        if (out_dist) out_dist[o + j] = ks_sqrt_cr(best_d);
        const float* p = pc + (size_t)best_i * 3;
        out_pts[(o + j) * 3 + 0] = OFFSETS ? p[0] - qx : p[0];
        out_pts[(o + j) * 3 + 1] = OFFSETS ? p[1] - qy : p[1];
        out_pts[(o + j) * 3 + 2] = OFFSETS ? p[2] - qz : p[2];
    }
}

// ---- host side ------------------------------------------------------------------------------------------------------------------
static size_t ks_al(size_t b) { return (b + 255) & ~(size_t)255; }

// worth the two counting sorts: enough queries for the lanes of a wave to be neighbours, enough candidates to scan
bool knn_sorted_applicable(int64_t Q, int64_t M, int k) {
    return (k == 1 || k == 4 || k == 8 || k == 16) && Q >= 8192 && M >= 512 && M < (1 << 24) && Q < (1ll << 31) && k <= M;
}
size_t knn_sorted_query_workspace_bytes(int64_t Q) { return 256 + ks_al(Q * sizeof(int)) * 2 + ks_al(KS_CELLS * sizeof(int)); }
size_t knn_sorted_cloud_workspace_bytes(int64_t M) { return 256 + ks_al(M * sizeof(int)) + ks_al(KS_CELLS * sizeof(int)) + ks_al(M * sizeof(float4)); }

// Sort the queries X [Q,3] of one cloud by grid cell; qws = knn_sorted_query_workspace_bytes(Q) bytes.  *qperm (device) maps
// sorted position -> original index and stays valid as long as qws is untouched.
void knn_sorted_prepare_queries(hipStream_t s, const float* X, int64_t Q, void* qws, const int** qperm) {
    char* w = (char*)qws;
    float* box = (float*)w; w += 256;
    int* code = (int*)w; w += ks_al(Q * sizeof(int));
    int* perm = (int*)w; w += ks_al(Q * sizeof(int));
    int* count = (int*)w;
    hipLaunchKernelGGL(ks_bbox_kernel, dim3(1), dim3(1024), 0, s, X, (int)Q, box, count);
    hipLaunchKernelGGL(ks_hist_kernel, dim3((unsigned)cdiv(Q, 256)), dim3(256), 0, s, X, (int)Q, box, code, count);
    hipLaunchKernelGGL(ks_scan_kernel, dim3(1), dim3(1024), 0, s, count);
    hipLaunchKernelGGL(ks_scatter_kernel, dim3((unsigned)cdiv(Q, 256)), dim3(256), 0, s, X, code, (int)Q, count, perm, (float4*)nullptr);
    *qperm = perm;
}

// k nearest points of ONE cloud pc [M,3] for the prepared queries; cws = knn_sorted_cloud_workspace_bytes(M) bytes
void knn_sorted_search(hipStream_t s, const float* X, const int* qperm, int64_t Q, const float* pc, int64_t M, int k, void* cws,
                       int64_t* idx, float* dists, float* pts, bool offsets) {
    char* w = (char*)cws;
    float* box = (float*)w; w += 256;
    int* code = (int*)w; w += ks_al(M * sizeof(int));
    int* count = (int*)w; w += ks_al(KS_CELLS * sizeof(int));
    float4* sorted = (float4*)w;
    hipLaunchKernelGGL(ks_bbox_kernel, dim3(1), dim3(1024), 0, s, pc, (int)M, box, count);
    hipLaunchKernelGGL(ks_hist_kernel, dim3((unsigned)cdiv(M, 256)), dim3(256), 0, s, pc, (int)M, box, code, count);
    hipLaunchKernelGGL(ks_scan_kernel, dim3(1), dim3(1024), 0, s, count);
    hipLaunchKernelGGL(ks_scatter_kernel, dim3((unsigned)cdiv(M, 256)), dim3(256), 0, s, pc, code, (int)M, count, (int*)nullptr, sorted);
    const dim3 grid((unsigned)cdiv(Q, MCR_WAVE * KS_QT));
    long long* i64 = (long long*)idx;
#define MCR_KS(KK)                                                                                                                    \
    if (offsets) hipLaunchKernelGGL((knn_sorted_kernel<KK, true>), grid, dim3(KS_BLOCK), 0, s, X, qperm, pc, sorted, box, count, i64, dists, pts, (int)Q, (int)M); \
    else hipLaunchKernelGGL((knn_sorted_kernel<KK, false>), grid, dim3(KS_BLOCK), 0, s, X, qperm, pc, sorted, box, count, i64, dists, pts, (int)Q, (int)M)
    switch (k) {
        case 1: MCR_KS(1); break;
        case 4: MCR_KS(4); break;
        case 8: MCR_KS(8); break;
        default: MCR_KS(16); break;
    }
#undef MCR_KS
}

}  // namespace mcr
