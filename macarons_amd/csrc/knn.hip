// K1 — brute-force k-nearest surface points per query, for gfx950.
//
// Replaces macarons/utility/utils.py:1497-1509 get_knn_points  (torch.cdist -> topk(largest=False) ->
// pytorch3d knn_gather, which materialises the dense [B,Q,M] distance matrix: 4 GB at Q=100k, M=10k) and the
// offset step of macarons/networks/SconeOcc.py:297-298 (neighbours minus the query).
//
// One lane owns one query and keeps its k best (d2, index) pairs sorted in VGPRs; surface points stream
// through LDS in tiles (one broadcast ds_read_b128 per candidate per wave).  A workgroup = 2 query tiles of 64 x 2
// waves per tile; the two waves of a tile each scan every other candidate (Q/64 waves alone cannot fill 1024 SIMDs)
// and their sorted lists are merged through LDS at the end.  A 4-way split was measured first: every extra split
// re-pays the ~16 ln(M/16) list insertions per lane, and its 6252 waves at 3 blocks/CU needed 2.03 rounds of the chip
// = 3; this shape needs 40 KB of LDS (4 blocks/CU) and 782 blocks = 0.76 rounds at Q = 100k with 35 % fewer
// instructions.  Candidates are taken 4 at a time (independent distance computations, one branch per batch).  Convention (shared with
// oracle/knn.py, see there why the reference's own tie order is unspecified):
//   d2 = (dx*dx + dy*dy) + dz*dz in fp32 with every product and sum rounded (no FMA contraction),
//   ascending by (d2, index): ties go to the lower index;  dists = sqrt(d2), correctly rounded.
// MCR_HIPCC_FLAGS: -ffp-contract=off
#include "common.h"
#include <cstdlib>

namespace mcr {

constexpr int KNN_BLOCK = 256;
constexpr int KNN_WAVES = KNN_BLOCK / MCR_WAVE;
constexpr int KNN_SPLIT = 2;                       // waves sharing one 64-query tile
constexpr int KNN_QT = KNN_WAVES / KNN_SPLIT;      // query tiles per workgroup
constexpr int KNN_TILE = 1536;     // surface points per LDS tile (24 KB as float4); multiple of 16
constexpr int KNN_QCAP = 8;        // per-lane queue of accepted candidates (LDS, [slot][thread])

// Insert (d2, idx) into the ascending list: slot j takes its upper neighbour if that one must move down,
// the new element if it lands here, else keeps its value (one v_cmp + four v_cndmask per slot, no branches).
template <int K>
__device__ __forceinline__ void knn_insert(float (&bd)[K], int (&bi)[K], float d2, int idx) {
    bool lands_or_below = d2 < bd[K - 1];    // strict: an equal distance never displaces an earlier index
    if (lands_or_below) {
#pragma unroll
        for (int j = K - 1; j > 0; --j) {
            const bool up_moves = d2 < bd[j - 1];
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

// Correctly rounded fp32 sqrt (the device sqrtf / fp64 sqrt paths measured 1 ulp off in ~5% of cases):
// start from v_sqrt_f32 (<= 1 ulp) and move to a neighbour if x lies beyond the midpoint, squared exactly in fp64.
__device__ __forceinline__ float sqrt_cr(float x) {
    if (!(x > 0.f)) return x;                                   // 0 (and NaN) pass through
    float y = __builtin_amdgcn_sqrtf(x);
    const float up = __builtin_bit_cast(float, __builtin_bit_cast(int, y) + 1);
    const float dn = __builtin_bit_cast(float, __builtin_bit_cast(int, y) - 1);
    const double m_up = 0.5 * ((double)y + (double)up), m_dn = 0.5 * ((double)y + (double)dn);
    const double xd = (double)x;
    if (xd > m_up * m_up) y = up;
    else if (xd < m_dn * m_dn) y = dn;
    return y;
}

__device__ __forceinline__ float knn_d2(float qx, float qy, float qz, const float4 p) {
    // every product and sum rounded separately (file is built with -ffp-contract=off; matches oracle/knn.py)
    const float dx = qx - p.x, dy = qy - p.y, dz = qz - p.z;
    const float xx = dx * dx, yy = dy * dy, zz = dz * dz;
    const float s = xx + yy;
    return s + zz;
}

// grid = (ceil(Q/128), B); block = 2 query tiles x 2 waves
template <int K, bool OFFSETS>
__global__ __launch_bounds__(KNN_BLOCK) void knn_kernel(const float* __restrict__ X, const float* __restrict__ pc,
                                                        long long* __restrict__ out_idx, float* __restrict__ out_dist,
                                                        float* __restrict__ out_pts, int Q, int M, const int4* __restrict__ blocks,
                                                        const long long* __restrict__ pc_off) {
    // 40 KB: [tile 24 KB][queue distances 8 KB][queue indices 8 KB]; the first 32 KB are reused as the merge buffer
    __shared__ __attribute__((aligned(16))) char smem[KNN_TILE * 16 + 2 * KNN_QCAP * KNN_BLOCK * 4];
    float4* s_pc = reinterpret_cast<float4*>(smem);
    float* s_qd = reinterpret_cast<float*>(smem + KNN_TILE * 16);
    int* s_qi = reinterpret_cast<int*>(smem + KNN_TILE * 16 + KNN_QCAP * KNN_BLOCK * 4);
    static_assert(KNN_WAVES * K * MCR_WAVE * 8 <= KNN_TILE * 16 + KNN_QCAP * KNN_BLOCK * 4, "merge buffer does not fit");
    const int b = blockIdx.y;
    const int lane = threadIdx.x & (MCR_WAVE - 1);
    const int wave = threadIdx.x / MCR_WAVE;
    const int qt = wave / KNN_SPLIT, part = wave % KNN_SPLIT;
    // segmented form (blocks != NULL): workgroup i serves queries [blocks[i].y, blocks[i].y + blocks[i].z) (at most 128 rows of X,
    // all of job blocks[i].x) against that job's own candidate cloud pc[pc_off[job] .. pc_off[job + 1]) -- many clouds of different
    // sizes in one launch (the per-cell clouds of the occupancy-field pass); neighbour indices are relative to the job's cloud
    int q_first = blockIdx.x * KNN_QT * MCR_WAVE, q_end = Q;
    const float* pcb = pc + (size_t)b * M * 3;
    if (blocks) {
        const int4 bk = blocks[blockIdx.x];
        q_first = bk.y; q_end = bk.y + bk.z;
        pcb = pc + (size_t)pc_off[bk.x] * 3;
        M = (int)(pc_off[bk.x + 1] - pc_off[bk.x]);
    }
    const int q = q_first + qt * MCR_WAVE + lane;
    const bool valid = q < q_end;
    const float* xq = X + ((size_t)b * Q + (valid ? q : q_end - 1)) * 3;
    const float qx = xq[0], qy = xq[1], qz = xq[2];

    float bd[K];
    int bi[K];
#pragma unroll
    for (int j = 0; j < K; ++j) { bd[j] = __builtin_inff(); bi[j] = 0x7fffffff; }

    // Accepted candidates are first pushed to a small per-lane LDS queue (a predicated ds_write) and inserted
    // into the sorted register list in batches: the ~90-instruction insertion then runs once per ~KNN_QCAP
    // accepted candidates of the fastest-filling lane instead of once per candidate any lane accepts.  The
    // filter threshold tau is the (possibly stale, hence larger) current k-th distance: never a false reject.
    float tau = __builtin_inff();
    int cnt = 0;
    float* q_d = s_qd + threadIdx.x;
    int* q_i = s_qi + threadIdx.x;
    auto flush = [&]() {
        int maxc = cnt;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) maxc = max(maxc, __shfl_xor(maxc, o, 64));
        for (int sidx = 0; sidx < maxc; ++sidx)
            if (sidx < cnt) knn_insert<K>(bd, bi, q_d[sidx * KNN_BLOCK], q_i[sidx * KNN_BLOCK]);
        tau = bd[K - 1];
        cnt = 0;
    };
    auto push = [&](float d, int idx) {
        if (d < tau) {
            q_d[cnt * KNN_BLOCK] = d;
            q_i[cnt * KNN_BLOCK] = idx;
            ++cnt;
        }
    };

    for (int t0 = 0; t0 < M; t0 += KNN_TILE) {
        const int nt = min(KNN_TILE, M - t0);
        const int nt_pad = (nt + 15) & ~15;
        __syncthreads();
        for (int i = threadIdx.x; i < nt_pad; i += KNN_BLOCK) {
            if (i < nt) {
                const float* p = pcb + (size_t)(t0 + i) * 3;
                s_pc[i] = make_float4(p[0], p[1], p[2], 0.f);
            } else {
                s_pc[i] = make_float4(3e18f, 3e18f, 3e18f, 0.f);        // d2 = +inf: never accepted
            }
        }
        __syncthreads();
        // the wave scans candidates i = part, part + SPLIT, ... of the tile, 4 per iteration
        for (int i = part; i < nt_pad; i += 4 * KNN_SPLIT) {
            const float4 p0 = s_pc[i], p1 = s_pc[i + KNN_SPLIT], p2 = s_pc[i + 2 * KNN_SPLIT], p3 = s_pc[i + 3 * KNN_SPLIT];
            const float d0 = knn_d2(qx, qy, qz, p0), d1 = knn_d2(qx, qy, qz, p1);
            const float d2 = knn_d2(qx, qy, qz, p2), d3 = knn_d2(qx, qy, qz, p3);
            // once the lists have warmed up almost every batch is rejected by every lane: one wave-uniform test on the batch
            // minimum then skips the four predicated pushes
            if (__any(fminf(fminf(d0, d1), fminf(d2, d3)) < tau)) {
                push(d0, t0 + i);
                push(d1, t0 + i + KNN_SPLIT);
                push(d2, t0 + i + 2 * KNN_SPLIT);
                push(d3, t0 + i + 3 * KNN_SPLIT);
                if (__any(cnt > KNN_QCAP - 4)) flush();
            }
        }
    }
    flush();
    // ---- merge of the tile's KNN_SPLIT sorted lists (lexicographic on (d2, index)) ----------------------------
    __syncthreads();
    float* m_d = reinterpret_cast<float*>(smem);                        // [wave][K][lane]
    int* m_i = reinterpret_cast<int*>(smem) + KNN_WAVES * K * MCR_WAVE;
#pragma unroll
    for (int j = 0; j < K; ++j) {
        m_d[(wave * K + j) * MCR_WAVE + lane] = bd[j];
        m_i[(wave * K + j) * MCR_WAVE + lane] = bi[j];
    }
    __syncthreads();
    if (part != 0 || !valid) return;
    m_d += qt * KNN_SPLIT * K * MCR_WAVE;                               // this tile's lists
    m_i += qt * KNN_SPLIT * K * MCR_WAVE;
    int head[KNN_SPLIT];
#pragma unroll
    for (int w = 0; w < KNN_SPLIT; ++w) head[w] = 0;
    const size_t o = ((size_t)b * Q + q) * K;
    for (int j = 0; j < K; ++j) {
        float best_d = __builtin_inff();
        int best_i = 0x7fffffff, best_w = 0;
#pragma unroll
        for (int w = 0; w < KNN_SPLIT; ++w) {
            const int h = head[w] < K ? head[w] : K - 1;
            const float d = head[w] < K ? m_d[(w * K + h) * MCR_WAVE + lane] : __builtin_inff();
            const int id = head[w] < K ? m_i[(w * K + h) * MCR_WAVE + lane] : 0x7fffffff;
            const bool better = d < best_d || (d == best_d && id < best_i);
            best_d = better ? d : best_d;
            best_i = better ? id : best_i;
            best_w = better ? w : best_w;
        }
#pragma unroll
        for (int w = 0; w < KNN_SPLIT; ++w) head[w] += (best_w == w) ? 1 : 0;
        if (out_idx) out_idx[o + j] = (long long)best_i;
        if (out_dist) out_dist[o + j] = sqrt_cr(best_d);
        const float* p = pcb + (size_t)best_i * 3;
        out_pts[(o + j) * 3 + 0] = OFFSETS ? p[0] - qx : p[0];
        out_pts[(o + j) * 3 + 1] = OFFSETS ? p[1] - qy : p[1];
        out_pts[(o + j) * 3 + 2] = OFFSETS ? p[2] - qz : p[2];
    }
}

// ---------------------------------------------------------------------------------------------------------
// K1-mfma: the same search with the distances on the matrix pipe.  A wave owns 32 queries; for 32 candidates at a time two
// v_mfma_f32_32x32x2_f32 evaluate  s[c][q] = |p_c|^2 - 2 x_q . p_c  (A row c = (p.x, p.y, p.z, |p|^2), B column q =
// (-2x.x, -2x.y, -2x.z, 1); the fp32 MFMA is bit for bit a k-ordered fmaf chain) -- 1024 candidate-query pairs per 128 pipe
// cycles, 3.7x the vector form (11 instructions per 64 pairs), and the vector ALU is left with the selection.  Lane (q, h) gets 16
// of the 32 candidates (rows (r & 3) + 8 (r >> 2) + 4 h) of ITS query, so a query's list is kept in two halves (merged at the
// end, like the two-wave split of the vector kernel).  s + |x|^2 is NOT the convention's distance (different rounding): it only
// FILTERS.  A candidate passes when s < tau - |x|^2 + eps (tau = the lane's current, possibly stale, k-th EXACT distance; eps =
// 2^-20 (|x| + max|p|)^2 covers both forms' rounding 16 times over, so no true neighbour is ever rejected); its index goes to
// the lane's LDS queue; at a flush the exact (dx^2 + dy^2) + dz^2 is recomputed from the coordinates and inserted with the same
// strict compare -- the output is bit-identical to the vector kernel's (and oracle/knn.py's).
constexpr int KM_TILE = 1024;      // candidates per LDS tile (SoA x | y | z | |p|^2: 16 KB)
constexpr int KM_QCAP = 32;        // queued candidate indices per lane (32 KB as [slot][thread])

template <int K, bool OFFSETS>
__global__ __launch_bounds__(KNN_BLOCK) void knn_mfma_kernel(const float* __restrict__ X, const float* __restrict__ pc,
                                                             long long* __restrict__ out_idx, float* __restrict__ out_dist,
                                                             float* __restrict__ out_pts, int Q, int M, const int4* __restrict__ blocks,
                                                             const long long* __restrict__ pc_off) {
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    __shared__ __attribute__((aligned(16))) float s_p[4][KM_TILE];           // x | y | z | |p|^2
    __shared__ int s_q[KM_QCAP * KNN_BLOCK];                                  // queue; reused as the merge buffer
    __shared__ unsigned s_pmax;
    static_assert(2 * K * 128 * 2 <= KM_QCAP * KNN_BLOCK, "merge buffer does not fit");
    const int b = blockIdx.y;
    const int lane = threadIdx.x & (MCR_WAVE - 1), wave = threadIdx.x / MCR_WAVE;
    const int j = lane & 31, h = lane >> 5;
    int q_first = blockIdx.x * 128, q_end = Q;
    const float* pcb = pc + (size_t)b * M * 3;
    if (blocks) {                                  // segmented form: see knn_kernel
        const int4 bk = blocks[blockIdx.x];
        q_first = bk.y; q_end = bk.y + bk.z;
        pcb = pc + (size_t)pc_off[bk.x] * 3;
        M = (int)(pc_off[bk.x + 1] - pc_off[bk.x]);
    }
    const int q = q_first + wave * 32 + j;
    const bool valid = q < q_end;
    const float* xq = X + ((size_t)b * Q + (valid ? q : q_end - 1)) * 3;
    const float qx = xq[0], qy = xq[1], qz = xq[2];
    const float q2 = (qx * qx + qy * qy) + qz * qz, qn = sqrtf(q2);
    const float b0 = h ? -2.f * qy : -2.f * qx, b1 = h ? 1.f : -2.f * qz;

    float bd[K];
    int bi[K];
#pragma unroll
    for (int r = 0; r < K; ++r) { bd[r] = __builtin_inff(); bi[r] = 0x7fffffff; }
    float tau = __builtin_inff(), thr = __builtin_inff(), eps = 0.f;
    int cnt = 0, t0 = 0;
    f32x16 zero;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero[r] = 0.f;
    asm volatile("" : "+v"(zero));                 // keep it in registers (not rematerialised as 16 moves per tile)
    int* qi = s_q + threadIdx.x;
    auto flush = [&]() {                           // exact distances of the queued candidates (their tile is still in LDS), insertion
        int maxc = cnt;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) maxc = max(maxc, __shfl_xor(maxc, o, 64));
        for (int sidx = 0; sidx < maxc; ++sidx)
            if (sidx < cnt) {
                const int idx = qi[sidx * KNN_BLOCK], li = idx - t0;
                const float d = knn_d2(qx, qy, qz, make_float4(s_p[0][li], s_p[1][li], s_p[2][li], 0.f));
                knn_insert<K>(bd, bi, d, idx);
            }
        tau = bd[K - 1];
        thr = (tau - q2) + eps;
        cnt = 0;
    };
    for (t0 = 0; t0 < M; t0 += KM_TILE) {
        const int nt = min(KM_TILE, M - t0);
        const int nt_pad = (nt + 31) & ~31;
        __syncthreads();                           // the previous tile (fragments and flushes) is done with
        if (threadIdx.x == 0) s_pmax = 0u;
        __syncthreads();
        float pm = 0.f;
        for (int i = threadIdx.x; i < nt_pad; i += KNN_BLOCK) {
            float x = 3e18f, y = 3e18f, z = 3e18f;                            // padding: far away, never passes the filter
            if (i < nt) { const float* p = pcb + (size_t)(t0 + i) * 3; x = p[0]; y = p[1]; z = p[2]; }
            const float pn = (x * x + y * y) + z * z;
            s_p[0][i] = x; s_p[1][i] = y; s_p[2][i] = z; s_p[3][i] = pn;
            if (i < nt) pm = fmaxf(pm, pn);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) pm = fmaxf(pm, __shfl_xor(pm, o, 64));
        if (lane == 0) atomicMax(&s_pmax, __builtin_bit_cast(unsigned, pm));  // non-negative floats order like their bit patterns
        __syncthreads();
        {
            const float pmax = sqrtf(__builtin_bit_cast(float, s_pmax)), e = qn + pmax;
            eps = 9.5367431640625e-07f * (e * e);                             // 2^-20 (|x| + max |p|)^2
            thr = (tau - q2) + eps;
        }
        const float* a0p = s_p[h ? 1 : 0] + j;
        const float* a1p = s_p[h ? 3 : 2] + j;
        for (int c0 = 0; c0 < nt_pad; c0 += 32) {
            f32x16 acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0p[c0], b0, zero, 0, 0, 0);     // C = a register set that stays zero
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1p[c0], b1, acc, 0, 0, 0);
            // Some lane of the wave passes in nearly every tile (64 lanes x 16 candidates), so the per-tile selection has to be cheap.
            // The lane's 16-bit mask of passing candidates is built branch-free from the SIGN of s - thr, two instructions per
            // candidate (v_sub_f32, v_alignbit_b32 shifting the sign bit in; bit 15 - r = candidate r; s == thr may pass as -0:
            // harmless, the exact compare follows); then one push per SET BIT in a loop that runs while any lane has bits left
            // (one or two rounds).  Sixteen predicated pushes cost ~12 instructions each (exec-mask juggling per branch): 0.36 of
            // 0.64 ms at Q = 100k, M = 10 240.
            unsigned mask = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                mask = __builtin_amdgcn_alignbit(mask, __builtin_bit_cast(unsigned, acc[r] - thr), 31);
            if (__any(mask != 0u)) {
                const int base = t0 + c0 + 4 * h;
                while (__any(mask != 0u)) {
                    if (mask != 0u) {
                        const int pbit = 31 - __builtin_clz(mask), r = 15 - pbit;    // highest bit = lowest candidate first: ties keep index order
                        qi[cnt * KNN_BLOCK] = base + (r & 3) + 8 * (r >> 2);
                        ++cnt;
                        mask &= ~(1u << pbit);
                    }
                }
                if (__any(cnt > KM_QCAP - 16)) flush();
            }
        }
        flush();
    }
    // ---- merge of the two half lists of every query (lexicographic on (d2, index)) ----------------------------
    __syncthreads();
    float* m_d = reinterpret_cast<float*>(s_q);                         // [half][K][128 queries of the block]
    int* m_i = s_q + 2 * K * 128;
    const int ql = wave * 32 + j;
#pragma unroll
    for (int r = 0; r < K; ++r) {
        m_d[(h * K + r) * 128 + ql] = bd[r];
        m_i[(h * K + r) * 128 + ql] = bi[r];
    }
    __syncthreads();
    // every lane takes part from here on (the staging below has barriers); the merge itself runs on the h == 0 lanes
    int head[2] = {0, 0};
    const size_t o = ((size_t)b * Q + q) * K;
    int mi[K];                                     // merged neighbour indices of this lane's query (h == 0 lanes)
    if (h == 0) {
        for (int r = 0; r < K; ++r) {
            float best_d = __builtin_inff();
            int best_i = 0x7fffffff, best_w = 0;
#pragma unroll
            for (int w = 0; w < 2; ++w) {
                const int hd = head[w] < K ? head[w] : K - 1;
                const float d = head[w] < K ? m_d[(w * K + hd) * 128 + ql] : __builtin_inff();
                const int id = head[w] < K ? m_i[(w * K + hd) * 128 + ql] : 0x7fffffff;
                const bool better = d < best_d || (d == best_d && id < best_i);
                best_d = better ? d : best_d;
                best_i = better ? id : best_i;
                best_w = better ? w : best_w;
            }
            head[0] += best_w == 0 ? 1 : 0;
            head[1] += best_w == 1 ? 1 : 0;
            mi[r] = best_i;
            if (valid) {
                if (out_idx) out_idx[o + r] = (long long)best_i;
                if (out_dist) out_dist[o + r] = sqrt_cr(best_d);
            }
        }
    }
    // ---- the neighbour coordinates leave through LDS: a block's 128 queries x K x 3 floats are ONE contiguous range of out_pts,
    // written as whole 16-byte chunks by all 256 threads (a lane storing its own 48 floats 12 bytes at a time at a 192-byte
    // stride was ~65 us of every launch: more than the whole search at M = 126)
    __syncthreads();                               // the merge buffer is read: s_p / s_q may be reused
    float* st = reinterpret_cast<float*>(s_q);     // [128 queries][K][3]  (K = 16: 24 KB <= 32 KB)
    if (h == 0) {
#pragma unroll
        for (int r = 0; r < K; ++r) {
            const int id = mi[r] == 0x7fffffff ? 0 : mi[r];
            const float* p = pcb + (size_t)id * 3;
            st[(ql * K + r) * 3 + 0] = OFFSETS ? p[0] - qx : p[0];
            st[(ql * K + r) * 3 + 1] = OFFSETS ? p[1] - qy : p[1];
            st[(ql * K + r) * 3 + 2] = OFFSETS ? p[2] - qz : p[2];
        }
    }
    __syncthreads();
    const int n_valid = max(0, min(128, q_end - q_first));                      // queries of this block
    const int n_f = n_valid * K * 3;
    float* dst = out_pts + ((size_t)b * Q + q_first) * K * 3;
    if ((reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
        for (int c = threadIdx.x; c * 4 + 3 < n_f; c += KNN_BLOCK) reinterpret_cast<float4*>(dst)[c] = reinterpret_cast<const float4*>(st)[c];
        for (int e = (n_f & ~3) + threadIdx.x; e < n_f; e += KNN_BLOCK) dst[e] = st[e];
    } else {
        for (int e = threadIdx.x; e < n_f; e += KNN_BLOCK) dst[e] = st[e];
    }
}

template <int K>
static void launch_knn(bool offsets, dim3 grid, hipStream_t s, const float* X, const float* pc, long long* idx, float* dist,
                       float* pts, int Q, int M, const int4* blocks = nullptr, const long long* pc_off = nullptr) {
    static const bool use_mfma = []() { const char* e = getenv("MCR_KNN_MFMA"); return !(e && e[0] == '0'); }();   // dev A/B knob
    if (use_mfma) {                                // distances on the matrix pipe (bit-identical results); same grid: 128 queries per block
        if (offsets)
            hipLaunchKernelGGL((knn_mfma_kernel<K, true>), grid, dim3(KNN_BLOCK), 0, s, X, pc, idx, dist, pts, Q, M, blocks, pc_off);
        else
            hipLaunchKernelGGL((knn_mfma_kernel<K, false>), grid, dim3(KNN_BLOCK), 0, s, X, pc, idx, dist, pts, Q, M, blocks, pc_off);
        return;
    }
    if (offsets)
        hipLaunchKernelGGL((knn_kernel<K, true>), grid, dim3(KNN_BLOCK), 0, s, X, pc, idx, dist, pts, Q, M, blocks, pc_off);
    else
        hipLaunchKernelGGL((knn_kernel<K, false>), grid, dim3(KNN_BLOCK), 0, s, X, pc, idx, dist, pts, Q, M, blocks, pc_off);
}

}  // namespace mcr

using namespace mcr;

// Segmented k = 16 search with the query offsets (SconeOcc's use): n_blocks workgroups, workgroup i = blocks[i] = (job, first query
// row, number of rows <= 128, 0); job j's candidates are pc[pc_off[j] .. pc_off[j+1]).  Every job needs >= 16 candidates.
namespace mcr {
void launch_knn16_segmented(hipStream_t s, const float* X, const float* pc, const long long* pc_off, const int* blocks,
                            int64_t n_blocks, int64_t T, float* offsets_out) {
    if (n_blocks <= 0) return;
    launch_knn<16>(true, dim3((unsigned)n_blocks, 1), s, X, pc, nullptr, nullptr, offsets_out, (int)T, 0,
                   reinterpret_cast<const int4*>(blocks), pc_off);
}
constexpr int knn_block_rows() { return MCR_WAVE * KNN_QT; }
int knn_rows_per_block() { return knn_block_rows(); }
}  // namespace mcr

extern "C" int mcr_knn_points(const float* X, const float* pc, int64_t* idx, float* dists, float* pts, int64_t B,
                              int64_t Q, int64_t M, int k, int subtract_query, void* stream) {
    MCR_REQUIRE(X && pc && pts, "mcr_knn_points: null pointer");
    MCR_REQUIRE(B > 0 && Q > 0 && M > 0, "mcr_knn_points: empty problem B=%ld Q=%ld M=%ld", (long)B, (long)Q, (long)M);
    MCR_REQUIRE(k <= M, "mcr_knn_points: k=%d exceeds the number of points M=%ld (torch.topk would raise)", k, (long)M);
    MCR_REQUIRE(B <= 65535 && Q < (1ll << 31) && M < (1ll << 31), "mcr_knn_points: problem too large");
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((unsigned)cdiv(Q, MCR_WAVE * KNN_QT), (unsigned)B);
    long long* i64 = (long long*)idx;
    switch (k) {
        case 1: launch_knn<1>(subtract_query, grid, s, X, pc, i64, dists, pts, (int)Q, (int)M); break;
        case 4: launch_knn<4>(subtract_query, grid, s, X, pc, i64, dists, pts, (int)Q, (int)M); break;
        case 8: launch_knn<8>(subtract_query, grid, s, X, pc, i64, dists, pts, (int)Q, (int)M); break;
        case 16: launch_knn<16>(subtract_query, grid, s, X, pc, i64, dists, pts, (int)Q, (int)M); break;
        
        default: MCR_REQUIRE(false, "mcr_knn_points: k must be one of 1,4,8,16 (got %d)", k);
    }
    MCR_LAUNCH_CHECK("knn_kernel");
    return 0;
}
