// K1' — k nearest surface points with WAVE-LEVEL TILE CULLING: same result as knn.hip (the brute-force scan), bit for bit,
// for a fraction of the distance evaluations.  Replaces macarons/utility/utils.py:1497-1509 get_knn_points (+ the offset step
// of SconeOcc.py:297-298) when the cloud is large enough to pay for the preparation.
//
//   * the QUERIES are grouped by cell of a 16^3 grid over their bounding box, cells in Morton order (counting sort: histogram,
//     scan, scatter): the 64 queries of a wave are neighbours in space and share one bounding box;
//   * the SURFACE POINTS are grouped the same way and cut into tiles of 64 consecutive points, each with its bounding box;
//   * a wave first scans the tile nearest to its box (its k-th distances become finite), then walks all tiles and skips a tile
//     when the squared distance between the two boxes exceeds the largest current k-th distance in the wave.  The test is
//     wave-uniform (no divergence); a visited tile is staged through a wave-private LDS buffer and read back as broadcasts.
//
// Exactness.  d2 = (dx*dx + dy*dy) + dz*dz with every product and sum rounded (this file is built with -ffp-contract=off, like
// knn.hip and oracle/knn.py); the result is ordered by (d2, ORIGINAL index).  Candidates are not visited in index order here,
// so the list insertion compares (d2, index) pairs lexicographically, and a candidate is a candidate while d2 <= the current
// k-th distance (an equal distance with a lower index still wins).  The box-to-box bound is evaluated with the same operation
// sequence on component gaps that are <= every pair's component difference; fp32 rounding is monotone, so the bound never
// exceeds any pair's computed d2: a skipped tile cannot hold a neighbour.  A skip needs bound > k-th distance (strict).
// MCR_HIPCC_FLAGS: -ffp-contract=off
#include "common.h"

namespace mcr {

constexpr int KC_G = 16, KC_CELLS = KC_G * KC_G * KC_G;     // grid of the counting sort
constexpr int KC_TILE = 64;                                 // surface points per tile
constexpr int KC_QCAP = 8;                                  // per-lane queue of accepted candidates (LDS)
constexpr int KC_BLOCK = 256;

// ---- preparation ------------------------------------------------------------------------------------------------------------
// bounding box of n points (row stride 3 floats): one block, no atomics; box[0..2] = min, box[3..5] = max
__global__ __launch_bounds__(1024) void kc_bbox_kernel(const float* __restrict__ p, int n, float* __restrict__ box) {
    __shared__ float s[6][16];
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

__device__ __forceinline__ unsigned kc_spread(unsigned v) {           // 4 bits -> every third bit
    v = (v | (v << 8)) & 0x0000F00Fu;
    v = (v | (v << 4)) & 0x000C30C3u;
    v = (v | (v << 2)) & 0x00249249u;
    return v;
}
__device__ __forceinline__ int kc_cell(const float* __restrict__ p, const float* __restrict__ box) {
    unsigned c[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float ext = box[3 + a] - box[a];
        const float t = ext > 0.f ? (p[a] - box[a]) / ext * (float)KC_G : 0.f;
        c[a] = (unsigned)min(KC_G - 1, max(0, (int)t));
    }
    return (int)(kc_spread(c[0]) | (kc_spread(c[1]) << 1) | (kc_spread(c[2]) << 2));      // Morton order of the cells
}

__global__ void kc_hist_kernel(const float* __restrict__ p, int n, const float* __restrict__ box, int* __restrict__ code,
                               int* __restrict__ count) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c = kc_cell(p + (size_t)i * 3, box);
    code[i] = c;
    atomicAdd(count + c, 1);
}

// exclusive scan of the KC_CELLS counters in place (one block of 1024 threads x 4 cells)
__global__ __launch_bounds__(1024) void kc_scan_kernel(int* __restrict__ count) {
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

__global__ void kc_scatter_kernel(const int* __restrict__ code, int n, int* __restrict__ cursor, int* __restrict__ perm) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    perm[atomicAdd(cursor + code[i], 1)] = i;           // order inside a cell is arbitrary: the result does not depend on it
}

// one wave per tile: the tile's points as float4 (x, y, z, original index) and its bounding box; the tail is padded with
// far-away points (d2 = +inf, index INT_MAX: never inserted)
__global__ __launch_bounds__(64) void kc_tiles_kernel(const float* __restrict__ pc, const int* __restrict__ perm, int M,
                                                      float4* __restrict__ sorted, float* __restrict__ tbox) {
    const int t = blockIdx.x, lane = threadIdx.x, i = t * KC_TILE + lane;
    float4 v = make_float4(3e18f, 3e18f, 3e18f, __builtin_bit_cast(float, 0x7fffffff));
    float mn[3] = {3e38f, 3e38f, 3e38f}, mx[3] = {-3e38f, -3e38f, -3e38f};
    if (i < M) {
        const int o = perm[i];
        const float* p = pc + (size_t)o * 3;
        v = make_float4(p[0], p[1], p[2], __builtin_bit_cast(float, o));
        mn[0] = mx[0] = p[0]; mn[1] = mx[1] = p[1]; mn[2] = mx[2] = p[2];
    }
    sorted[i] = v;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            mn[c] = fminf(mn[c], __shfl_xor(mn[c], o, 64));
            mx[c] = fmaxf(mx[c], __shfl_xor(mx[c], o, 64));
        }
    }
    if (lane == 0) {
        float* b = tbox + (size_t)t * 8;
        b[0] = mn[0]; b[1] = mn[1]; b[2] = mn[2]; b[3] = mx[0]; b[4] = mx[1]; b[5] = mx[2]; b[6] = 0.f; b[7] = 0.f;
    }
}

// ---- search -----------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool kc_less(float d, int i, float bd, int bi) { return d < bd || (d == bd && i < bi); }

// Insert (d2, idx) into the list kept ascending by (d2, idx): slot j takes its upper neighbour if that one must move down, the
// new element if it lands here, else keeps its value.
template <int K>
__device__ __forceinline__ void kc_insert(float (&bd)[K], int (&bi)[K], float d2, int idx) {
    bool lands_or_below = kc_less(d2, idx, bd[K - 1], bi[K - 1]);
    if (lands_or_below) {
#pragma unroll
        for (int j = K - 1; j > 0; --j) {
            const bool up_moves = kc_less(d2, idx, bd[j - 1], bi[j - 1]);
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

__device__ __forceinline__ float kc_sqrt_cr(float x) {              // correctly rounded sqrt, as in knn.hip
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

__device__ __forceinline__ float kc_d2(float qx, float qy, float qz, const float4 p) {
    const float dx = qx - p.x, dy = qy - p.y, dz = qz - p.z;
    const float xx = dx * dx, yy = dy * dy, zz = dz * dz;
    const float s = xx + yy;
    return s + zz;
}

// squared distance between two boxes with the operation sequence of kc_d2 on the component gaps (see the header)
__device__ __forceinline__ float kc_box_d2(const float (&qmn)[3], const float (&qmx)[3], const float* __restrict__ b) {
    float g[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float a = b[c] - qmx[c], e = qmn[c] - b[3 + c];          // tile entirely above / below the queries on this axis
        g[c] = fmaxf(0.f, fmaxf(a, e));
    }
    const float xx = g[0] * g[0], yy = g[1] * g[1], zz = g[2] * g[2];
    const float s = xx + yy;
    return s + zz;
}

// grid = ceil(Q / 256): a wave owns the 64 consecutive queries of the cell-sorted order.
// Candidate tiles travel global -> registers (one point per lane, coalesced 1 KB) -> a wave-private LDS buffer -> broadcast
// reads; the next surviving tile is requested while the current one is scanned.  (Scalar loads straight from global were tried
// first: 160 KB of tiles thrash the 16 KB scalar cache and every 64-byte line cost an exposed L2 round trip -- 2.5 ms.)
constexpr int KC_MAXT = 1024;                               // tiles per cloud (M <= 65536)
#ifdef KC_STATS             // dev only (tools/build_variant.py ... -DKC_STATS): tiles scanned / flushes / inserts per launch
__device__ unsigned long long kc_stats[4];
__device__ unsigned int kc_wave_rec[4 * 4096];         // per wave: cycles, tiles scanned, insert iterations, list length
#define KC_COUNT(i, v) do { if (lane == 0) atomicAdd(&kc_stats[i], (unsigned long long)(v)); } while (0)
#else
#define KC_COUNT(i, v) do { } while (0)
#endif
template <int K, bool OFFSETS>
__global__ __launch_bounds__(KC_BLOCK) void knn_culled_kernel(const float* __restrict__ X, const int* __restrict__ qperm,
                                                              const float* __restrict__ pc, const float4* __restrict__ sorted,
                                                              const float* __restrict__ tbox, int n_tiles,
                                                              long long* __restrict__ out_idx, float* __restrict__ out_dist,
                                                              float* __restrict__ out_pts, int Q) {
    __shared__ float s_qd[KC_QCAP * KC_BLOCK];
    __shared__ int s_qi[KC_QCAP * KC_BLOCK];
    __shared__ __attribute__((aligned(16))) float4 s_tile[KC_BLOCK / 64][2][KC_TILE];      // wave-private double buffer
    __shared__ unsigned short s_list[KC_BLOCK / 64][KC_MAXT];                              // surviving tiles of each wave
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pos = blockIdx.x * KC_BLOCK + threadIdx.x;
#ifdef KC_STATS
    const long long kc_t0 = clock64();
    unsigned kc_scans = 0, kc_ins = 0;
#endif
    const bool valid = pos < Q;
    const int q = qperm[valid ? pos : Q - 1];
    const float qx = X[(size_t)q * 3], qy = X[(size_t)q * 3 + 1], qz = X[(size_t)q * 3 + 2];
    // the wave's bounding box (every lane holds a real query: the tail repeats the last one)
    float qmn[3] = {qx, qy, qz}, qmx[3] = {qx, qy, qz};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            qmn[c] = fminf(qmn[c], __shfl_xor(qmn[c], o, 64));
            qmx[c] = fmaxf(qmx[c], __shfl_xor(qmx[c], o, 64));
        }
    }

    float bd[K];
    int bi[K];
#pragma unroll
    for (int j = 0; j < K; ++j) { bd[j] = __builtin_inff(); bi[j] = 0x7fffffff; }
    float tau = __builtin_inff(), tau_wave = __builtin_inff();
    int cnt = 0;
    float* q_d = s_qd + threadIdx.x;
    int* q_i = s_qi + threadIdx.x;
    auto flush = [&]() {
#ifdef KC_PROBE_NOFLUSH
        cnt = 0; tau = 0.05f; tau_wave = 0.05f; return;
#endif
        int maxc = cnt;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) maxc = max(maxc, __shfl_xor(maxc, o, 64));
        KC_COUNT(1, 1); KC_COUNT(2, maxc);
#ifdef KC_STATS
        kc_ins += maxc;
#endif
        for (int sidx = 0; sidx < maxc; ++sidx)
            if (sidx < cnt) kc_insert<K>(bd, bi, q_d[sidx * KC_BLOCK], q_i[sidx * KC_BLOCK]);
        tau = bd[K - 1];
        cnt = 0;
        float tw = tau;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) tw = fmaxf(tw, __shfl_xor(tw, o, 64));
        tau_wave = tw;                                  // the largest k-th distance in the wave (same value in every lane)
    };
    auto push = [&](float d, int idx) {
        if (d <= tau) {                                 // <=: an equal distance with a lower index still displaces the k-th entry
            q_d[cnt * KC_BLOCK] = d;
            q_i[cnt * KC_BLOCK] = idx;
            ++cnt;
        }
    };
    // scan the 64 candidates of the tile staged in buffer `buf` (broadcast reads, 4 per iteration)
    auto scan = [&](int buf) {
#ifdef KC_PROBE_NOSCAN
        return;
#endif
        KC_COUNT(0, 1);
#ifdef KC_STATS
        ++kc_scans;
#endif
        const float4* tp = s_tile[wave][buf];
#pragma unroll 2
        for (int j = 0; j < KC_TILE; j += 4) {
            const float4 p0 = tp[j], p1 = tp[j + 1], p2 = tp[j + 2], p3 = tp[j + 3];
            const float d0 = kc_d2(qx, qy, qz, p0), d1 = kc_d2(qx, qy, qz, p1);
            const float d2 = kc_d2(qx, qy, qz, p2), d3 = kc_d2(qx, qy, qz, p3);
            if (__any(fminf(fminf(d0, d1), fminf(d2, d3)) <= tau)) {
                push(d0, __builtin_bit_cast(int, p0.w));
                push(d1, __builtin_bit_cast(int, p1.w));
                push(d2, __builtin_bit_cast(int, p2.w));
                push(d3, __builtin_bit_cast(int, p3.w));
                if (__any(cnt > KC_QCAP - 4)) flush();
            }
        }
    };
    auto stage = [&](int buf, float4 v) {               // registers -> the wave's LDS buffer; LDS ops of a wave execute in order
        s_tile[wave][buf][lane] = v;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };

    // ---- phase 1: the tile nearest to the wave's box (and its two neighbours in the sorted order) first, so that the k-th
    // distances are finite before the sweep ----
    float best = __builtin_inff();
    int best_t = 0;
    for (int t = lane; t < n_tiles; t += 64) {
        const float lb = kc_box_d2(qmn, qmx, tbox + (size_t)t * 8);
        if (lb < best) { best = lb; best_t = t; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const int ot = __shfl_xor(best_t, o, 64);
        if (ob < best || (ob == best && ot < best_t)) { best = ob; best_t = ot; }
    }
    const int t_first = __builtin_amdgcn_readfirstlane(best_t);
    const int t_lo = max(0, t_first - 1), t_hi = min(n_tiles - 1, t_first + 1);
    {
        float4 nxt = sorted[(size_t)t_lo * KC_TILE + lane];
        for (int t = t_lo; t <= t_hi; ++t) {
            stage(t & 1, nxt);
            if (t < t_hi) nxt = sorted[(size_t)(t + 1) * KC_TILE + lane];
            scan(t & 1);
        }
    }
    flush();
    // ---- phase 2: the tiles that can still hold a neighbour of some query of the wave, compacted into a list ----
    int n_list = 0;
    for (int t0 = 0; t0 < n_tiles; t0 += 64) {
        const int t = t0 + lane;
        const bool keep = t < n_tiles && !(t >= t_lo && t <= t_hi) && !(kc_box_d2(qmn, qmx, tbox + (size_t)min(t, n_tiles - 1) * 8) > tau_wave);
        const unsigned long long m = __ballot(keep);
        if (keep) s_list[wave][n_list + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)t;
        n_list += __popcll(m);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    KC_COUNT(3, n_list);
    if (n_list > 0) {
        int t = s_list[wave][0];
        float4 nxt = sorted[(size_t)t * KC_TILE + lane];
        for (int i = 0; i < n_list; ++i) {
            const int tn = i + 1 < n_list ? s_list[wave][i + 1] : 0;
            // the k-th distances have shrunk since the list was built: test again before paying for the scan
            const bool skip = kc_box_d2(qmn, qmx, tbox + (size_t)t * 8) > tau_wave;
            if (!skip) stage(i & 1, nxt);
            if (i + 1 < n_list) nxt = sorted[(size_t)tn * KC_TILE + lane];
            if (!skip) scan(i & 1);
            t = tn;
        }
    }
    flush();
#ifdef KC_STATS
    {
        const int w = blockIdx.x * 4 + wave;
        if (lane == 0 && w < 4096) {
            kc_wave_rec[4 * w] = (unsigned)(clock64() - kc_t0); kc_wave_rec[4 * w + 1] = kc_scans; kc_wave_rec[4 * w + 2] = kc_ins;
            kc_wave_rec[4 * w + 3] = (unsigned)n_list;
        }
    }
#endif
    if (!valid) return;
#if defined(KC_PROBE_NOFLUSH) || defined(KC_PROBE_NOSCAN)
    out_dist[(size_t)q * K] = bd[0] + tau;             // timing probes: keep the work alive, no gather through garbage indices
    return;
#endif
    const size_t o = (size_t)q * K;
#pragma unroll
    for (int j = 0; j < K; ++j) {
        out_idx[o + j] = (long long)bi[j];
        out_dist[o + j] = kc_sqrt_cr(bd[j]);
        const float* p = pc + (size_t)bi[j] * 3;
        out_pts[(o + j) * 3 + 0] = OFFSETS ? p[0] - qx : p[0];
        out_pts[(o + j) * 3 + 1] = OFFSETS ? p[1] - qy : p[1];
        out_pts[(o + j) * 3 + 2] = OFFSETS ? p[2] - qz : p[2];
    }
}

// ---- host side ------------------------------------------------------------------------------------------------------------------
static size_t kc_al(size_t b) { return (b + 255) & ~(size_t)255; }

// counting sort of n points by grid cell: perm[sorted position] = original index.  scratch: box (8 floats), code [n], count [KC_CELLS]
static int kc_cell_sort(hipStream_t s, const float* p, int n, float* box, int* code, int* count, int* perm) {
    if (int e = check_hip(hipMemsetAsync(count, 0, KC_CELLS * sizeof(int), s), "knn_culled: memset")) return e;
    hipLaunchKernelGGL(kc_bbox_kernel, dim3(1), dim3(1024), 0, s, p, n, box);
    hipLaunchKernelGGL(kc_hist_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, s, p, n, box, code, count);
    hipLaunchKernelGGL(kc_scan_kernel, dim3(1), dim3(1024), 0, s, count);
    hipLaunchKernelGGL(kc_scatter_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, s, code, n, count, perm);
    return 0;
}

size_t knn_culled_query_workspace_bytes(int64_t Q) { return kc_al(Q * sizeof(int)) * 2 + kc_al(KC_CELLS * sizeof(int)) + 256; }
size_t knn_culled_cloud_workspace_bytes(int64_t M) {
    const int64_t nt = cdiv(M, KC_TILE);
    return kc_al(M * sizeof(int)) * 2 + kc_al(KC_CELLS * sizeof(int)) + kc_al(nt * KC_TILE * sizeof(float4)) + kc_al(nt * 8 * sizeof(float)) + 256;
}
bool knn_culled_applicable(int64_t Q, int64_t M, int k) { return k == 16 && M >= 1024 && Q >= 4096 && M < (1 << 24) && Q < (1ll << 31); }

// Group the queries of ONE cloud (X [Q,3]); qws = knn_culled_query_workspace_bytes(Q) bytes.  Returns the permutation (device).
int knn_culled_prepare_queries(hipStream_t s, const float* X, int64_t Q, void* qws, const int** qperm_out) {
    char* w = (char*)qws;
    float* box = (float*)w; w += 256;
    int* code = (int*)w; w += kc_al(Q * sizeof(int));
    int* perm = (int*)w; w += kc_al(Q * sizeof(int));
    int* count = (int*)w;
    if (int e = kc_cell_sort(s, X, (int)Q, box, code, count, perm)) return e;
    *qperm_out = perm;
    return 0;
}

// k = 16 nearest points of ONE cloud pc [M,3] for the prepared queries; cws = knn_culled_cloud_workspace_bytes(M) bytes
int knn_culled_search(hipStream_t s, const float* X, const int* qperm, int64_t Q, const float* pc, int64_t M, void* cws,
                      int64_t* idx, float* dists, float* pts, bool offsets) {
    char* w = (char*)cws;
    float* box = (float*)w; w += 256;
    int* code = (int*)w; w += kc_al(M * sizeof(int));
    int* perm = (int*)w; w += kc_al(M * sizeof(int));
    int* count = (int*)w; w += kc_al(KC_CELLS * sizeof(int));
    const int nt = (int)cdiv(M, KC_TILE);
    float4* sorted = (float4*)w; w += kc_al((size_t)nt * KC_TILE * sizeof(float4));
    float* tbox = (float*)w;
    if (int e = kc_cell_sort(s, pc, (int)M, box, code, count, perm)) return e;
    hipLaunchKernelGGL(kc_tiles_kernel, dim3((unsigned)nt), dim3(64), 0, s, pc, perm, (int)M, sorted, tbox);
    dim3 grid((unsigned)cdiv(Q, KC_BLOCK));
    if (offsets)
        hipLaunchKernelGGL((knn_culled_kernel<16, true>), grid, dim3(KC_BLOCK), 0, s, X, qperm, pc, sorted, tbox, nt,
                           (long long*)idx, dists, pts, (int)Q);
    else
        hipLaunchKernelGGL((knn_culled_kernel<16, false>), grid, dim3(KC_BLOCK), 0, s, X, qperm, pc, sorted, tbox, nt,
                           (long long*)idx, dists, pts, (int)Q);
    return 0;
}

#ifdef KC_STATS
extern "C" int mcr_dev_knn_wave_rec(unsigned int* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(kc_wave_rec), sizeof(unsigned) * 4 * 4096); }
extern "C" int mcr_dev_knn_stats(unsigned long long* out, int reset) {
    int e = (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(kc_stats), sizeof(unsigned long long) * 4);
    if (reset) { unsigned long long z[4] = {0, 0, 0, 0}; e |= (int)hipMemcpyToSymbol(HIP_SYMBOL(kc_stats), z, sizeof(z)); }
    return e;
}
#endif

}  // namespace mcr

using namespace mcr;

extern "C" {

size_t mcr_knn_points_culled_workspace_bytes(int64_t Q, int64_t M) {
    return knn_culled_query_workspace_bytes(Q) + knn_culled_cloud_workspace_bytes(M);
}

int mcr_knn_points_culled(const float* X, const float* pc, int64_t* idx, float* dists, float* pts, int64_t B, int64_t Q,
                          int64_t M, int k, int subtract_query, void* workspace, size_t workspace_bytes, void* stream) {
    MCR_REQUIRE(X && pc && idx && dists && pts && workspace, "mcr_knn_points_culled: null pointer");
    MCR_REQUIRE(B > 0 && Q > 0 && M > 0, "mcr_knn_points_culled: empty problem");
    MCR_REQUIRE(k == 16, "mcr_knn_points_culled: k must be 16 (got %d); use mcr_knn_points", k);
    MCR_REQUIRE(k <= M, "mcr_knn_points_culled: k=%d exceeds the number of points M=%ld", k, (long)M);
    MCR_REQUIRE(Q < (1ll << 31) && M < (1 << 24), "mcr_knn_points_culled: problem too large");
    MCR_REQUIRE(workspace_bytes >= mcr_knn_points_culled_workspace_bytes(Q, M), "mcr_knn_points_culled: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    char* qws = (char*)workspace;
    char* cws = qws + knn_culled_query_workspace_bytes(Q);
    for (int64_t b = 0; b < B; ++b) {
        const int* qperm = nullptr;
        if (int e = knn_culled_prepare_queries(s, X + b * Q * 3, Q, qws, &qperm)) return e;
        if (int e = knn_culled_search(s, X + b * Q * 3, qperm, Q, pc + b * M * 3, M, cws, idx + b * Q * k, dists + b * Q * k,
                                      pts + b * Q * k * 3, subtract_query != 0))
            return e;
    }
    MCR_LAUNCH_CHECK("knn_culled_kernel");
    return 0;
}

}  // extern "C"
