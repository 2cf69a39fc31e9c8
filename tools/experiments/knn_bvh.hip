// K1-tree — k nearest surface points through a bounding-box tree over the cloud: the same result as knn.hip (the brute-force
// scan), bit for bit, for ~1/30 of the distance evaluations at M = 10 240.  Replaces macarons/utility/utils.py:1497-1509
// get_knn_points (+ the offset step of macarons/networks/SconeOcc.py:297-298) whenever the cloud and the query set are large
// enough to pay for the build (knn_bvh_applicable); knn.hip stays the small-problem path and the bit-exactness yardstick.
//
// Build (5 small launches per cloud, ~20 us at M = 10 240, no host synchronisation):
//   bounding box -> cell of a 32^3 grid per point (cells in Morton order) -> counting sort (histogram, scan, scatter) ->
//   LEAVES of 8 consecutive points of that order (x, y, z, original index), padded to a power-of-two leaf count with far-away
//   points -> an implicit complete binary tree of axis-aligned boxes, bottom up (node n has children 2n, 2n+1; leaves at
//   n = L .. 2L-1).  The order of the points inside a cell is arbitrary (atomics); the result does not depend on it.
// Search: ONE LANE PER QUERY walks the tree with its own stack (LDS, one column per thread), nearer child first, and skips a
//   subtree when the squared distance from the query to its box exceeds the lane's current k-th distance.  Every lane prunes
//   with its own bound: no wave-level amplification of the candidate set (the shelved wave-uniform tile culling of
//   tools/experiments/knn_culled.hip scanned a third of the cloud per wave and lost to the plain scan).  Queries need no
//   particular order.
//
// Exactness.  d2 = (dx*dx + dy*dy) + dz*dz with every product and sum rounded (this file is built with -ffp-contract=off, like
// knn.hip and oracle/knn.py); the result is ordered by (d2, ORIGINAL index) -- candidates arrive in tree order, so the list
// insertion compares (d2, index) pairs lexicographically and a candidate stays a candidate while d2 <= the current k-th distance
// (an equal distance with a lower index still wins).  The box bound uses the same operation sequence on component gaps that are
// <= every contained point's component difference; fp32 rounding is monotone, so the bound never exceeds the computed d2 of a
// point inside the box: a skipped subtree cannot hold a neighbour.  A skip needs bound > k-th distance (strict).
// MCR_HIPCC_FLAGS: -ffp-contract=off
#include "common.h"

namespace mcr {

constexpr int KB_G = 32, KB_CELLS = KB_G * KB_G * KB_G;     // grid of the counting sort
constexpr int KB_LEAF = 8;                                  // points per leaf
constexpr int KB_BLOCK = 256;
constexpr int KB_MAX_M = 1 << 20;

// ---- build ---------------------------------------------------------------------------------------------------------------------
// bounding box of n points (row stride 3 floats) -> box[0..2] = min, box[3..5] = max; also clears the cell counters
__global__ __launch_bounds__(1024) void kb_bbox_kernel(const float* __restrict__ p, int n, float* __restrict__ box,
                                                       int* __restrict__ count) {
    __shared__ float s[6][16];
    for (int i = threadIdx.x; i < KB_CELLS; i += 1024) count[i] = 0;
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

__device__ __forceinline__ unsigned kb_spread(unsigned v) {           // 10 bits -> every third bit
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}
// Morton index of the grid cell; a non-finite coordinate or a degenerate extent lands in cell 0 of that axis (any cell is
// correct: the cells only order the points)
__device__ __forceinline__ int kb_cell(const float* __restrict__ p, const float* __restrict__ box) {
    unsigned c[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float ext = box[3 + a] - box[a];
        const float t = ext > 0.f ? (p[a] - box[a]) / ext * (float)KB_G : 0.f;
        c[a] = (unsigned)(t >= 0.f ? (t < (float)KB_G ? (int)t : KB_G - 1) : 0);
    }
    return (int)(kb_spread(c[0]) | (kb_spread(c[1]) << 1) | (kb_spread(c[2]) << 2));
}

__global__ void kb_hist_kernel(const float* __restrict__ p, int n, const float* __restrict__ box, int* __restrict__ code,
                               int* __restrict__ count) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c = kb_cell(p + (size_t)i * 3, box);
    code[i] = c;
    atomicAdd(count + c, 1);
}

// exclusive scan of the KB_CELLS counters in place (one block of 1024 threads x 32 cells)
__global__ __launch_bounds__(1024) void kb_scan_kernel(int* __restrict__ count) {
    __shared__ int s[1024];
    constexpr int PER = KB_CELLS / 1024;
    int t = 0;
    for (int e = 0; e < PER; ++e) t += count[threadIdx.x * PER + e];
    s[threadIdx.x] = t;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int u = threadIdx.x >= o ? s[threadIdx.x - o] : 0;
        __syncthreads();
        s[threadIdx.x] += u;
        __syncthreads();
    }
    int run = s[threadIdx.x] - t;
    for (int e = 0; e < PER; ++e) {
        const int v = count[threadIdx.x * PER + e];
        count[threadIdx.x * PER + e] = run;
        run += v;
    }
}

__global__ void kb_scatter_kernel(const float* __restrict__ p, const int* __restrict__ code, int n, int* __restrict__ cursor,
                                  float4* __restrict__ sorted) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* q = p + (size_t)i * 3;
    sorted[atomicAdd(cursor + code[i], 1)] = make_float4(q[0], q[1], q[2], __builtin_bit_cast(float, i));
}

// nodes[2 n] = box min, nodes[2 n + 1] = box max of tree node n; leaves L .. 2L-1 from their 8 points (the tail of `sorted` is
// padded with far-away points: d2 = +inf, index INT_MAX, never inserted; an all-padding leaf gets an inverted box whose bound is
// +inf), inner nodes level by level.  One block: the tree of a 10k-point cloud has 2048 leaves.
__global__ __launch_bounds__(1024) void kb_tree_kernel(float4* __restrict__ sorted, int M, int L, float4* __restrict__ nodes) {
    for (int l = threadIdx.x; l < L; l += 1024) {
        float mn[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()};
        float mx[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
#pragma unroll
        for (int j = 0; j < KB_LEAF; ++j) {
            const int i = l * KB_LEAF + j;
            if (i < M) {
                const float4 v = sorted[i];
                mn[0] = fminf(mn[0], v.x); mn[1] = fminf(mn[1], v.y); mn[2] = fminf(mn[2], v.z);
                mx[0] = fmaxf(mx[0], v.x); mx[1] = fmaxf(mx[1], v.y); mx[2] = fmaxf(mx[2], v.z);
            } else {
                sorted[i] = make_float4(3e18f, 3e18f, 3e18f, __builtin_bit_cast(float, 0x7fffffff));
            }
        }
        nodes[2 * (L + l)] = make_float4(mn[0], mn[1], mn[2], 0.f);
        nodes[2 * (L + l) + 1] = make_float4(mx[0], mx[1], mx[2], 0.f);
    }
    __syncthreads();
    for (int lvl = L >> 1; lvl >= 1; lvl >>= 1) {
        for (int n = lvl + threadIdx.x; n < 2 * lvl; n += 1024) {
            const float4 a = nodes[4 * n], b = nodes[4 * n + 2], c = nodes[4 * n + 1], d = nodes[4 * n + 3];
            nodes[2 * n] = make_float4(fminf(a.x, b.x), fminf(a.y, b.y), fminf(a.z, b.z), 0.f);
            nodes[2 * n + 1] = make_float4(fmaxf(c.x, d.x), fmaxf(c.y, d.y), fmaxf(c.z, d.z), 0.f);
        }
        __syncthreads();
    }
}

// ---- search --------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool kb_less(float d, int i, float bd, int bi) { return d < bd || (d == bd && i < bi); }

// Insert (d2, idx) into the list kept ascending by (d2, idx): slot j takes its upper neighbour if that one must move down, the
// new element if it lands here, else keeps its value.
template <int K>
__device__ __forceinline__ void kb_insert(float (&bd)[K], int (&bi)[K], float d2, int idx) {
    bool lands_or_below = kb_less(d2, idx, bd[K - 1], bi[K - 1]);
    if (lands_or_below) {
#pragma unroll
        for (int j = K - 1; j > 0; --j) {
            const bool up_moves = kb_less(d2, idx, bd[j - 1], bi[j - 1]);
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

__device__ __forceinline__ float kb_sqrt_cr(float x) {              // correctly rounded sqrt, as in knn.hip
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

__device__ __forceinline__ float kb_d2(float qx, float qy, float qz, const float4 p) {
    const float dx = qx - p.x, dy = qy - p.y, dz = qz - p.z;
    const float xx = dx * dx, yy = dy * dy, zz = dz * dz;
    const float s = xx + yy;
    return s + zz;
}
// squared distance from the query to a box, with the operation sequence of kb_d2 on the component gaps (see the header)
__device__ __forceinline__ float kb_box_d2(float qx, float qy, float qz, const float4 mn, const float4 mx) {
    const float gx = fmaxf(0.f, fmaxf(mn.x - qx, qx - mx.x));
    const float gy = fmaxf(0.f, fmaxf(mn.y - qy, qy - mx.y));
    const float gz = fmaxf(0.f, fmaxf(mn.z - qz, qz - mx.z));
    const float xx = gx * gx, yy = gy * gy, zz = gz * gz;
    const float s = xx + yy;
    return s + zz;
}

// grid = ceil(Q / 256); dynamic LDS = depth x 256 x 8 bytes (stack of (node, bound) per thread, depth = log2(L) + 1)
template <int K, bool OFFSETS>
__global__ __launch_bounds__(KB_BLOCK) void knn_bvh_kernel(const float* __restrict__ X, const float* __restrict__ pc,
                                                           const float4* __restrict__ sorted, const float4* __restrict__ nodes,
                                                           int L, int depth, long long* __restrict__ out_idx,
                                                           float* __restrict__ out_dist, float* __restrict__ out_pts, int Q) {
    extern __shared__ int kb_stack[];
    int* st_n = kb_stack + threadIdx.x;
    float* st_b = reinterpret_cast<float*>(kb_stack + depth * KB_BLOCK) + threadIdx.x;
    const int q = blockIdx.x * KB_BLOCK + threadIdx.x;
    const bool valid = q < Q;
    const float* xq = X + (size_t)(valid ? q : Q - 1) * 3;
    const float qx = xq[0], qy = xq[1], qz = xq[2];

    float bd[K];
    int bi[K];
#pragma unroll
    for (int j = 0; j < K; ++j) { bd[j] = __builtin_inff(); bi[j] = 0x7fffffff; }

    int sp = 0, node = 1;
    bool active = valid;
    // next subtree from the stack that can still hold a neighbour (the k-th distance has shrunk since it was pushed)
    auto pop = [&]() {
        active = false;
        while (sp > 0) {
            --sp;
            if (!(st_b[sp * KB_BLOCK] > bd[K - 1])) {
                node = st_n[sp * KB_BLOCK];
                active = true;
                break;
            }
        }
    };
    if (L == 1) node = 1;                                   // a single leaf is the root
    while (active) {
        while (active && node < L) {                       // inner node: nearer child first, the other one on the stack
            const int c0 = 2 * node;
            const float b0 = kb_box_d2(qx, qy, qz, nodes[2 * c0], nodes[2 * c0 + 1]);
            const float b1 = kb_box_d2(qx, qy, qz, nodes[2 * c0 + 2], nodes[2 * c0 + 3]);
            const float tau = bd[K - 1];
            const bool p0 = !(b0 > tau), p1 = !(b1 > tau);
            if (p0 && p1) {
                const bool first0 = b0 <= b1;
                st_n[sp * KB_BLOCK] = first0 ? c0 + 1 : c0;
                st_b[sp * KB_BLOCK] = first0 ? b1 : b0;
                ++sp;
                node = first0 ? c0 : c0 + 1;
            } else if (p0) {
                node = c0;
            } else if (p1) {
                node = c0 + 1;
            } else {
                pop();
            }
        }
        if (active) {                                       // leaf: its 8 points
            const float4* lp = sorted + (size_t)(node - L) * KB_LEAF;
            float4 p[KB_LEAF];
#pragma unroll
            for (int j = 0; j < KB_LEAF; ++j) p[j] = lp[j];
#pragma unroll
            for (int j = 0; j < KB_LEAF; ++j) kb_insert<K>(bd, bi, kb_d2(qx, qy, qz, p[j]), __builtin_bit_cast(int, p[j].w));
            pop();
        }
    }
    if (!valid) return;
    const size_t o = (size_t)q * K;
#pragma unroll
    for (int j = 0; j < K; ++j) {
        out_idx[o + j] = (long long)bi[j];
        out_dist[o + j] = kb_sqrt_cr(bd[j]);
        const float* p = pc + (size_t)bi[j] * 3;
        out_pts[(o + j) * 3 + 0] = OFFSETS ? p[0] - qx : p[0];
        out_pts[(o + j) * 3 + 1] = OFFSETS ? p[1] - qy : p[1];
        out_pts[(o + j) * 3 + 2] = OFFSETS ? p[2] - qz : p[2];
    }
}

// ---- host side ------------------------------------------------------------------------------------------------------------------
static size_t kb_al(size_t b) { return (b + 255) & ~(size_t)255; }
static int kb_leaves(int64_t M) {
    int L = 1;
    while ((int64_t)L * KB_LEAF < M) L <<= 1;
    return L;
}

size_t knn_bvh_workspace_bytes(int64_t M) {
    if (M <= 0 || M > KB_MAX_M) return 0;
    const int L = kb_leaves(M);
    return 256 + kb_al(M * sizeof(int)) + kb_al(KB_CELLS * sizeof(int)) + kb_al((size_t)L * KB_LEAF * sizeof(float4)) +
           kb_al((size_t)4 * L * sizeof(float4));
}
// worth the build: enough queries to amortise five small launches, enough points for the pruning to matter
bool knn_bvh_applicable(int64_t Q, int64_t M, int k) {
    return (k == 1 || k == 4 || k == 8 || k == 16) && M >= 512 && M <= KB_MAX_M && Q >= 1024 && Q < (1ll << 31) && k <= M;
}

template <int K>
static void kb_launch_search(bool offsets, hipStream_t s, const float* X, const float* pc, const float4* sorted, const float4* nodes,
                             int L, int depth, long long* idx, float* dist, float* pts, int Q) {
    const dim3 grid((unsigned)cdiv(Q, KB_BLOCK));
    const size_t lds = (size_t)depth * KB_BLOCK * 8;
    if (offsets)
        hipLaunchKernelGGL((knn_bvh_kernel<K, true>), grid, dim3(KB_BLOCK), lds, s, X, pc, sorted, nodes, L, depth, idx, dist, pts, Q);
    else
        hipLaunchKernelGGL((knn_bvh_kernel<K, false>), grid, dim3(KB_BLOCK), lds, s, X, pc, sorted, nodes, L, depth, idx, dist, pts, Q);
}

// k nearest points of ONE cloud pc [M,3] for the queries X [Q,3]; ws = knn_bvh_workspace_bytes(M) bytes
int knn_bvh_search(hipStream_t s, const float* X, int64_t Q, const float* pc, int64_t M, int k, void* ws, int64_t* idx,
                   float* dists, float* pts, bool offsets) {
    char* w = (char*)ws;
    float* box = (float*)w; w += 256;
    int* code = (int*)w; w += kb_al(M * sizeof(int));
    int* count = (int*)w; w += kb_al(KB_CELLS * sizeof(int));
    const int L = kb_leaves(M);
    float4* sorted = (float4*)w; w += kb_al((size_t)L * KB_LEAF * sizeof(float4));
    float4* nodes = (float4*)w;
    int depth = 2;                                          // stack entries: one per level below the root, + 1
    for (int l = L; l > 1; l >>= 1) ++depth;
    hipLaunchKernelGGL(kb_bbox_kernel, dim3(1), dim3(1024), 0, s, pc, (int)M, box, count);
    hipLaunchKernelGGL(kb_hist_kernel, dim3((unsigned)cdiv(M, 256)), dim3(256), 0, s, pc, (int)M, box, code, count);
    hipLaunchKernelGGL(kb_scan_kernel, dim3(1), dim3(1024), 0, s, count);
    hipLaunchKernelGGL(kb_scatter_kernel, dim3((unsigned)cdiv(M, 256)), dim3(256), 0, s, pc, code, (int)M, count, sorted);
    hipLaunchKernelGGL(kb_tree_kernel, dim3(1), dim3(1024), 0, s, sorted, (int)M, L, nodes);
    long long* i64 = (long long*)idx;
    switch (k) {
        case 1: kb_launch_search<1>(offsets, s, X, pc, sorted, nodes, L, depth, i64, dists, pts, (int)Q); break;
        case 4: kb_launch_search<4>(offsets, s, X, pc, sorted, nodes, L, depth, i64, dists, pts, (int)Q); break;
        case 8: kb_launch_search<8>(offsets, s, X, pc, sorted, nodes, L, depth, i64, dists, pts, (int)Q); break;
        default: kb_launch_search<16>(offsets, s, X, pc, sorted, nodes, L, depth, i64, dists, pts, (int)Q); break;
    }
    return 0;
}

}  // namespace mcr
