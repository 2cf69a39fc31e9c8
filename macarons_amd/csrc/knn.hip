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
#include "nn_kernels.h"
#include <algorithm>
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
                                                             const long long* __restrict__ pc_off, int n_split = 1,
                                                             unsigned long long* __restrict__ part_out = nullptr, int seg_cand = 0) {
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    __shared__ __attribute__((aligned(16))) float s_p[4][KM_TILE];           // x | y | z | |p|^2
    __shared__ int s_q[KM_QCAP * KNN_BLOCK];                                  // queue; reused as the merge buffer
    __shared__ unsigned s_pmax;
    static_assert(2 * K * 128 * 2 <= KM_QCAP * KNN_BLOCK, "merge buffer does not fit");
    // segmented form with n_split > 1 (few query blocks against large clouds: the ragged occupancy pass has ~200 blocks for 256 CUs):
    // grid.y = n_split, workgroup (i, sp) scans the sp-th slice of its job's candidates and leaves its 16 best (d2 bits << 32 | index)
    // keys per query in part_out [n_split][Q][K]; knn_split_merge_kernel takes the K smallest keys of the n_split lists -- the same
    // (d2, index) order, so the same neighbours in the same order as the unsplit scan
    const int b = blocks ? 0 : blockIdx.y, sp = blocks ? blockIdx.y : 0;
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
    int c_first = 0;
    if (n_split > 1) {                             // this workgroup's candidates [c_first, M)
        // seg_cand > 0: a job's candidates are cut into slices of ~seg_cand (at most n_split of them): the jobs of a ragged pass hold
        // 1 000 ... 15 000 candidates, and with one slice count for all of them the launch waited for the largest clouds' blocks
        if (seg_cand > 0) {
            n_split = min(n_split, max(1, (M + seg_cand - 1) / seg_cand));
            if (sp >= n_split) return;
        }
        const int chunk = (((M + n_split - 1) / n_split) + 31) & ~31;
        c_first = min(M, sp * chunk);
        M = min(M, c_first + chunk);
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
    for (t0 = c_first; t0 < M; t0 += KM_TILE) {
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
                if (part_out)
                    part_out[((size_t)sp * Q + q) * K + r] = ((unsigned long long)__builtin_bit_cast(unsigned, best_d) << 32) | (unsigned)best_i;
                if (out_idx) out_idx[o + r] = (long long)best_i;
                if (out_dist) out_dist[o + r] = sqrt_cr(best_d);
            }
        }
    }
    if (part_out) return;                          // (block-uniform) the coordinates are written by the merge
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

// ---------------------------------------------------------------------------------------------------------
// K1-grid (default for 1024 <= M <= 16384, k = 16): the same exact search, but a wave only LOOKS at the candidates near its queries.
//
// The brute-force kernels evaluate all Q M pairs (1.02 G at Q = 100k, M = 10 240) although a query's 16 neighbours sit within
// ~7 % of the cloud's extent.  Here (i) the candidates are counting-sorted by the Hilbert index of their cell in a 16^3 grid over
// the cloud's bounding box (kg_build_cloud_kernel: one workgroup per cloud, LDS counters) into an array of float4 (x, y, z, original
// index), cut into SUB-TILES of 32 consecutive candidates with their bounding boxes; (ii) the queries are counting-sorted the same
// way on a 32^3 grid over THEIR bounding box (kg_q*: once per SconeOcc forward, shared by the three scales), so that a wave's 32
// queries are neighbours in space; (iii) a wave computes the boxes of its queries (four groups of eight), seeds its lists from the
// sub-tile nearest to them, and then visits only sub-tiles whose box-to-box lower bound lb does not exceed R^2 = the largest current
// 16th distance among its queries (KG_ROUNDS rounds, lb <= 0.02 R^2 ... <= R^2: roughly near to far, the lists -- and R --
// tightening in between).  A sub-tile is
// processed like the brute-force MFMA kernel processes 32 candidates (two v_mfma_f32_32x32x2_f32 filter, exact recompute + insert
// at the flush).  Candidates no longer arrive in index order: list entries are 64-bit keys (d2 bits << 32 | index), compared
// lexicographically -- the documented convention (ascending (d2, index)) without relying on the scan order.
//
// Exactness: a candidate in a skipped sub-tile has (real) distance^2 >= lb > R^2 >= the true 16th distance of every query of
// the wave (any 16 candidates bound it from above); lb is compared after a relative guard of 1e-5 (its own fp32 rounding is
// ~2e-7).  Everything that is visited goes through the same filter + exact fp32 (dx^2 + dy^2) + dz^2 + insert as before, so the
// output is bit-identical to knn_kernel / knn_mfma_kernel and oracle/knn.py (tests/test_knn_gpu.py, incl. heavy ties).
constexpr int KG_GP_BITS = 4, KG_GQ_BITS = 5, KG_GP = 1 << KG_GP_BITS, KG_GQ = 1 << KG_GQ_BITS;      // cells per axis: candidate grid, query grid
constexpr int KG_MIN_M = 1024, KG_MAX_M = 16384;
constexpr int KG_BUILD_BLOCK = 1024;

struct KgCloud { float lo[3], inv[3]; float pmax2; int n_sub; };       // one per cloud (32 bytes)
#ifdef KG_DEBUG
__device__ unsigned kg_trace[8192 * 16];          // dev instrumentation (-DKG_DEBUG, tools/time_knn.py): per wave [0..4] sub-tiles, insert rounds, flushes, max / mean insertions per lane; [8..] cycles per phase
#define KG_COUNT(i, v) do { if (lane == 0 && blockIdx.x < 8192) kg_trace[blockIdx.x * 16 + (i)] = (unsigned)(v); } while (0)
#define KG_LOCAL(...) __VA_ARGS__
#define KG_T0() unsigned long long kg_t = __builtin_readcyclecounter()
#define KG_T(i) do { const unsigned long long n_ = __builtin_readcyclecounter(); KG_COUNT(8 + i, n_ - kg_t); kg_t = n_; } while (0)
#else
#define KG_COUNT(i, v) do { } while (0)
#define KG_LOCAL(...)
#define KG_T0() do { } while (0)
#define KG_T(i) do { } while (0)
#endif

__device__ __forceinline__ unsigned kg_spread3(unsigned v) {            // 10 bits -> every third bit
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}
__device__ __forceinline__ int kg_cell(float p, float lo, float inv, int G) {
    const int c = (int)((p - lo) * inv);
    return min(max(c, 0), G - 1);
}
// Hilbert index of cell (cx, cy, cz) in a (2^BITS)^3 grid (Skilling's transpose construction): consecutive indices are
// face-adjacent cells, so 32 consecutive sorted queries / candidates always form one compact snake.  (A Morton order jumps across
// the grid at every block boundary: 1 % of the waves straddled a jump, owned a box spanning both sides and ran 3-8x longer than
// the median wave.)
template <int BITS>
__device__ __forceinline__ unsigned kg_hilbert(unsigned x0, unsigned x1, unsigned x2) {
    unsigned X[3] = {x0, x1, x2};
#pragma unroll
    for (unsigned Q = 1u << (BITS - 1); Q > 1; Q >>= 1) {
        const unsigned P = Q - 1;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (X[i] & Q) X[0] ^= P;
            else { const unsigned t = (X[0] ^ X[i]) & P; X[0] ^= t; X[i] ^= t; }
        }
    }
    X[1] ^= X[0]; X[2] ^= X[1];
    unsigned t = 0;
#pragma unroll
    for (unsigned Q = 1u << (BITS - 1); Q > 1; Q >>= 1)
        if (X[2] & Q) t ^= Q - 1;
    return (kg_spread3(X[0] ^ t) << 2) | (kg_spread3(X[1] ^ t) << 1) | kg_spread3(X[2] ^ t);
}
template <int BITS>
__device__ __forceinline__ unsigned kg_code(float x, float y, float z, const float* lo, const float* inv) {
    constexpr int G = 1 << BITS;
    return kg_hilbert<BITS>((unsigned)kg_cell(x, lo[0], inv[0], G), (unsigned)kg_cell(y, lo[1], inv[1], G), (unsigned)kg_cell(z, lo[2], inv[2], G));
}
__device__ __forceinline__ int kg_ord(float f) {                        // float -> int with the same order
    const int i = __builtin_bit_cast(int, f);
    return i >= 0 ? i : i ^ 0x7fffffff;
}
__device__ __forceinline__ float kg_unord(int i) { return __builtin_bit_cast(float, i >= 0 ? i : i ^ 0x7fffffff); }

// exclusive scan of n = per * KG_BUILD_BLOCK ints in place (LDS or global), by one workgroup of KG_BUILD_BLOCK threads
template <int PER>
__device__ __forceinline__ void kg_block_scan(int* v, int* s_wave /* [16] */) {      // v 16-byte aligned, PER % 4 == 0
    static_assert(PER % 4 == 0, "whole int4 per thread");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int4* v4 = reinterpret_cast<int4*>(v) + tid * (PER / 4);
    int loc[PER], sum = 0;
#pragma unroll
    for (int e = 0; e < PER / 4; ++e) {
        const int4 x = v4[e];
        loc[4 * e] = sum; sum += x.x; loc[4 * e + 1] = sum; sum += x.y; loc[4 * e + 2] = sum; sum += x.z; loc[4 * e + 3] = sum; sum += x.w;
    }
    int inc = sum;                                 // inclusive wave scan of the thread sums
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; ++w) base += s_wave[w];
    base += inc - sum;
#pragma unroll
    for (int e = 0; e < PER / 4; ++e) v4[e] = make_int4(base + loc[4 * e], base + loc[4 * e + 1], base + loc[4 * e + 2], base + loc[4 * e + 3]);
    __syncthreads();
}

// one workgroup per cloud (of up to three candidate sets x B clouds in one launch): bounding box, counting sort by Hilbert cell,
// padded float4 array, sub-tile boxes
struct KgBuildArgs {
    const float* pc[3]; float4* cand[3]; float4* boxes[3]; KgCloud* hdr[3];
    int M[3]; int B;
};
__global__ __launch_bounds__(KG_BUILD_BLOCK) void kg_build_cloud_kernel(const KgBuildArgs args) {
    __shared__ __attribute__((aligned(16))) int s_cnt[KG_GP * KG_GP * KG_GP];
    __shared__ float s_red[7][16];
    __shared__ int s_wave[16];
    __shared__ float s_lo[3], s_inv[3];
    const int which = blockIdx.x / args.B, b = blockIdx.x - which * args.B;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int M = args.M[which], Mpad = (M + 31) & ~31, n_sub = Mpad / 32;
    const float* pcb = args.pc[which] + (size_t)b * M * 3;
    float4* cb = args.cand[which] + (size_t)b * Mpad;
    float4* bb = args.boxes[which] + (size_t)b * n_sub * 2;
    KgCloud* hdr = args.hdr[which] + b;
    float mn[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()}, mx[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()}, pm = 0.f;
    for (int i = tid; i < M; i += KG_BUILD_BLOCK) {
        const float x = pcb[(size_t)i * 3], y = pcb[(size_t)i * 3 + 1], z = pcb[(size_t)i * 3 + 2];
        mn[0] = fminf(mn[0], x); mn[1] = fminf(mn[1], y); mn[2] = fminf(mn[2], z);
        mx[0] = fmaxf(mx[0], x); mx[1] = fmaxf(mx[1], y); mx[2] = fmaxf(mx[2], z);
        pm = fmaxf(pm, (x * x + y * y) + z * z);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { mn[a] = fminf(mn[a], __shfl_xor(mn[a], o, 64)); mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], o, 64)); }
        pm = fmaxf(pm, __shfl_xor(pm, o, 64));
    }
    if (lane == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { s_red[a][wave] = mn[a]; s_red[3 + a][wave] = mx[a]; }
        s_red[6][wave] = pm;
    }
    for (int i = tid; i < KG_GP * KG_GP * KG_GP; i += KG_BUILD_BLOCK) s_cnt[i] = 0;
    __syncthreads();
    if (tid < 3) {
        float lo = s_red[tid][0], hi = s_red[3 + tid][0];
        for (int w = 1; w < 16; ++w) { lo = fminf(lo, s_red[tid][w]); hi = fmaxf(hi, s_red[3 + tid][w]); }
        const float ext = hi - lo;
        s_lo[tid] = lo;
        s_inv[tid] = (ext > 0.f && ext < __builtin_inff()) ? (float)KG_GP / ext : 0.f;
        hdr->lo[tid] = s_lo[tid]; hdr->inv[tid] = s_inv[tid];
    }
    if (tid == 3) {
        float p = s_red[6][0];
        for (int w = 1; w < 16; ++w) p = fmaxf(p, s_red[6][w]);
        hdr->pmax2 = p;
        hdr->n_sub = n_sub;
    }
    __syncthreads();
    for (int i = tid; i < M; i += KG_BUILD_BLOCK)
        atomicAdd(&s_cnt[kg_code<KG_GP_BITS>(pcb[(size_t)i * 3], pcb[(size_t)i * 3 + 1], pcb[(size_t)i * 3 + 2], s_lo, s_inv)], 1);
    __syncthreads();
    kg_block_scan<KG_GP * KG_GP * KG_GP / KG_BUILD_BLOCK>(s_cnt, s_wave);
    for (int i = tid; i < M; i += KG_BUILD_BLOCK) {
        const float x = pcb[(size_t)i * 3], y = pcb[(size_t)i * 3 + 1], z = pcb[(size_t)i * 3 + 2];
        const int pos = atomicAdd(&s_cnt[kg_code<KG_GP_BITS>(x, y, z, s_lo, s_inv)], 1);
        cb[pos] = make_float4(x, y, z, __builtin_bit_cast(float, i));
    }
    for (int i = M + tid; i < Mpad; i += KG_BUILD_BLOCK) cb[i] = make_float4(3e18f, 3e18f, 3e18f, __builtin_bit_cast(float, 0x7fffffff));
    __syncthreads();                               // the sorted array (written by this workgroup) is visible to it
    // sub-tile boxes: one half wave per sub-tile, lane = candidate (padding rows stay out of the box)
    const int j = lane & 31, h = lane >> 5;
    for (int t = wave * 2 + h; t < ((n_sub + 1) & ~1); t += 32) {
        float lo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()}, hi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
        if (t < n_sub && t * 32 + j < M) {
            const float4 p = cb[t * 32 + j];
            lo[0] = hi[0] = p.x; lo[1] = hi[1] = p.y; lo[2] = hi[2] = p.z;
        }
#pragma unroll
        for (int o = 16; o > 0; o >>= 1)
#pragma unroll
            for (int a = 0; a < 3; ++a) { lo[a] = fminf(lo[a], __shfl_xor(lo[a], o, 64)); hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], o, 64)); }
        if (j == 0 && t < n_sub) {
            bb[2 * t] = make_float4(lo[0], lo[1], lo[2], 0.f);
            bb[2 * t + 1] = make_float4(hi[0], hi[1], hi[2], 0.f);
        }
    }
}

// ---- query order: qperm[b][sorted position] = query row, sorted by the Hilbert index of the query's cell (32^3 over the queries' box)
constexpr int KG_QCELLS = KG_GQ * KG_GQ * KG_GQ;
__global__ void kg_qinit_kernel(int* __restrict__ hist, int* __restrict__ qbox) {     // grid (KG_QCELLS / 1024, B)
    const int b = blockIdx.y;
    hist[(size_t)b * KG_QCELLS + blockIdx.x * 1024 + threadIdx.x] = 0;
    if (blockIdx.x == 0 && threadIdx.x < 6) qbox[b * 8 + threadIdx.x] = threadIdx.x < 3 ? 0x7fffffff : (int)0x80000000;
}
__global__ __launch_bounds__(256) void kg_qbbox_kernel(const float* __restrict__ X, int Q, int* __restrict__ qbox) {      // grid (blocks, B)
    __shared__ float s_red[6][4];
    const int b = blockIdx.y;
    const float* xb = X + (size_t)b * Q * 3;
    float mn[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()}, mx[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    for (int i = blockIdx.x * 256 + threadIdx.x; i < Q; i += gridDim.x * 256)
#pragma unroll
        for (int a = 0; a < 3; ++a) { const float v = xb[(size_t)i * 3 + a]; mn[a] = fminf(mn[a], v); mx[a] = fmaxf(mx[a], v); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
        for (int a = 0; a < 3; ++a) { mn[a] = fminf(mn[a], __shfl_xor(mn[a], o, 64)); mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], o, 64)); }
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int a = 0; a < 3; ++a) { s_red[a][threadIdx.x >> 6] = mn[a]; s_red[3 + a][threadIdx.x >> 6] = mx[a]; }
    __syncthreads();
    if (threadIdx.x < 3) atomicMin(&qbox[b * 8 + threadIdx.x], kg_ord(fminf(fminf(s_red[threadIdx.x][0], s_red[threadIdx.x][1]), fminf(s_red[threadIdx.x][2], s_red[threadIdx.x][3]))));
    else if (threadIdx.x < 6) atomicMax(&qbox[b * 8 + threadIdx.x], kg_ord(fmaxf(fmaxf(s_red[threadIdx.x][0], s_red[threadIdx.x][1]), fmaxf(s_red[threadIdx.x][2], s_red[threadIdx.x][3]))));
}
__global__ __launch_bounds__(256) void kg_qcell_kernel(const float* __restrict__ X, int Q, const int* __restrict__ qbox, int* __restrict__ hist,
                                                       int* __restrict__ code, int* __restrict__ rank) {                 // grid (ceil(Q/256), B)
    const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= Q) return;
    float lo[3], inv[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        lo[a] = kg_unord(qbox[b * 8 + a]);
        const float ext = kg_unord(qbox[b * 8 + 3 + a]) - lo[a];
        inv[a] = (ext > 0.f && ext < __builtin_inff()) ? (float)KG_GQ / ext : 0.f;
    }
    const float* x = X + ((size_t)b * Q + i) * 3;
    const int c = (int)kg_code<KG_GQ_BITS>(x[0], x[1], x[2], lo, inv);
    code[(size_t)b * Q + i] = c;
    rank[(size_t)b * Q + i] = atomicAdd(&hist[(size_t)b * KG_QCELLS + c], 1);
}
__global__ __launch_bounds__(KG_BUILD_BLOCK) void kg_qscan_kernel(int* __restrict__ hist) {                               // grid (B)
    __shared__ int s_wave[16];
    kg_block_scan<KG_QCELLS / KG_BUILD_BLOCK>(hist + (size_t)blockIdx.x * KG_QCELLS, s_wave);
}
__global__ __launch_bounds__(256) void kg_qscatter_kernel(int Q, const int* __restrict__ hist, const int* __restrict__ code,
                                                          const int* __restrict__ rank, int* __restrict__ qperm) {          // grid (ceil(Q/256), B)
    const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= Q) return;
    qperm[(size_t)b * Q + hist[(size_t)b * KG_QCELLS + code[(size_t)b * Q + i]] + rank[(size_t)b * Q + i]] = i;
}

// Insert key = (d2 bits << 32 | index) into the ascending list of 64-bit keys held as (hi = d2 bits, lo = index) register pairs.
// With c_j = (k < key_j) (one 64-bit compare per slot, kept in scalar registers): slot j takes key_{j-1} if c_{j-1}, k if c_j only,
// else stays.  The high words do not need the masks: hi'_j = med3(hi_{j-1}, k_hi, hi_j) (the list is sorted; on a tie of the high
// words the candidates for the slot have equal high words anyway) -- 16 compares + 16 v_med3_u32 + 32 v_cndmask per insertion
// (the plain select form compiled to two compares, four selects and two wait states per slot).
__device__ __forceinline__ unsigned kg_umed3(unsigned a, unsigned b, unsigned c) {
    unsigned r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
template <int K>
__device__ __forceinline__ void kg_insert(unsigned (&khi)[K], unsigned (&klo)[K], unsigned hi, unsigned lo) {
    const unsigned long long k = ((unsigned long long)hi << 32) | lo;
    bool c[K];
#pragma unroll
    for (int j = 0; j < K; ++j) c[j] = k < (((unsigned long long)khi[j] << 32) | klo[j]);
#pragma unroll
    for (int j = K - 1; j > 0; --j) {
        klo[j] = c[j - 1] ? klo[j - 1] : (c[j] ? lo : klo[j]);
        khi[j] = kg_umed3(khi[j - 1], hi, khi[j]);
    }
    klo[0] = c[0] ? lo : klo[0];
    khi[0] = min(khi[0], hi);
}

// grid = (ceil(Q / 32), B); ONE wave per workgroup = 32 queries (sorted positions), lane (j, h) as in knn_mfma_kernel: waves
// never wait for each other (their work differs by 2x with the geometry) and leave the CU as they finish.  Everything a wave touches
// repeatedly sits in its 9 KB of LDS: the lower bounds of all sub-tiles (computed once), a ring of the last KG_RING sub-tiles it
// filtered (a flush re-reads candidate coordinates from LDS, not from L2) and the queue of accepted ring positions; the next
// sub-tile's 32 candidates are requested from L2 before the current one is filtered.
//
// Heavy groups.  Queries far from the cloud (the middle of a hollow object) have hundreds of sub-tiles within their 16th distance: a
// few waves out of thousands then run 5-8x longer than the median one and the launch waits for them (measured on an ellipsoid
// shell: mean 45 sub-tiles per wave, the heaviest 313; kernel 520 us for 100 us of average work).  A wave that finds more than
// KG_HEAVY sub-tiles behind it and in the band of the next visiting round therefore PARKS its group: it stores its lists, its lower bounds and the band's
// lower limit in a slot of the scratch buffer and exits.  A second launch (MODE 1) gives every parked group KG_SPLIT waves, each
// taking every KG_SPLIT-th remaining sub-tile with the parked 16th distances as its starting thresholds (so it only keeps genuine
// improvements); the eight waves are ONE workgroup and fold their lists pairwise through LDS (three steps of 16 insertions), wave
// 0 adds the parked lists and writes the group's output.  The launch has a fixed grid and exits at once for unused slots: no host
// round trip.  A group that finds no free slot simply carries on.
constexpr int KG_RING = 6;                         // sub-tiles between two flushes at most
constexpr int KG_ROUNDS = 6;                       // visiting rounds: sub-tiles in (roughly) increasing distance from the wave's queries
__device__ __forceinline__ float kg_round_frac(int r) { return r == 0 ? 0.02f : (r == 1 ? 0.06f : (r == 2 ? 0.15f : (r == 3 ? 0.3f : 0.55f))); }
constexpr int KG_MAX_SUB = KG_MAX_M / 32;          // 512
constexpr int KG_LDS_LB = KG_MAX_SUB * 4, KG_LDS_RING = KG_RING * 32 * 16, KG_LDS_QUEUE = KM_QCAP * 64 * 2;
constexpr int KG_LDS_BYTES = KG_LDS_LB + KG_LDS_RING + KG_LDS_QUEUE;            // 2 + 3 + 4 KB
constexpr int KG_HEAVY = 80, KG_SPLIT = 8, KG_MAXDEF = 384;
// scratch of one parked group: header | lower bounds | its lists [half][K = 16][32 queries] as 64-bit keys
struct KgSlot { int group, b; float prev; int seed; };
constexpr size_t KG_SLOT_LISTS = 2 * 16 * 32 * 8;                                // 8 KB
constexpr size_t KG_SLOT_BYTES = 256 + KG_MAX_SUB * 4 + KG_SLOT_LISTS;
__device__ __forceinline__ KgSlot* kg_slot_hdr(char* scratch, int slot) { return reinterpret_cast<KgSlot*>(scratch + (size_t)slot * KG_SLOT_BYTES); }
__device__ __forceinline__ float* kg_slot_lb(char* scratch, int slot) { return reinterpret_cast<float*>(scratch + (size_t)slot * KG_SLOT_BYTES + 256); }
__device__ __forceinline__ unsigned long long* kg_slot_lists(char* scratch, int slot) {
    return reinterpret_cast<unsigned long long*>(scratch + (size_t)slot * KG_SLOT_BYTES + 256 + KG_MAX_SUB * 4);
}

// the end of a group's search: merge the two half lists of every query, write indices / distances / neighbour coordinates
template <int K, bool OFFSETS, bool BLOCK_IS_WAVE>
__device__ __forceinline__ void kg_output(const unsigned (&khi)[K], const unsigned (&klo)[K], char* s_mem, int* s_row, int lane, int b, int Q,
                                          int q, bool valid, float qx, float qy, float qz, const float* pcb, long long* out_idx,
                                          float* out_dist, float* out_pts) {
    const int j = lane & 31, h = lane >> 5;
    auto sync = [&]() { if (BLOCK_IS_WAVE) __syncthreads(); else { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } };
    sync();                                        // (one wave: the barriers only order the LDS traffic)
    unsigned long long* m_k = reinterpret_cast<unsigned long long*>(s_mem);     // [half][K][32 queries]: 8 KB
#pragma unroll
    for (int r = 0; r < K; ++r) m_k[(h * K + r) * 32 + j] = ((unsigned long long)khi[r] << 32) | klo[r];
    if (h == 0) s_row[j] = valid ? q : -1;
    sync();
    int head[2] = {0, 0};
    const size_t o = ((size_t)b * Q + q) * K;
    int mi[K];
    if (h == 0) {
        for (int r = 0; r < K; ++r) {
            const unsigned long long k0 = head[0] < K ? m_k[(0 * K + min(head[0], K - 1)) * 32 + j] : ~0ull;
            const unsigned long long k1 = head[1] < K ? m_k[(1 * K + min(head[1], K - 1)) * 32 + j] : ~0ull;
            const bool first = k0 <= k1;           // equal only for two empty slots
            const unsigned long long kb = first ? k0 : k1;
            head[0] += first ? 1 : 0;
            head[1] += first ? 0 : 1;
            mi[r] = (int)(unsigned)kb;
            if (valid) {
                if (out_idx) out_idx[o + r] = (long long)mi[r];
                if (out_dist) out_dist[o + r] = sqrt_cr(__builtin_bit_cast(float, (unsigned)(kb >> 32)));
            }
        }
    }
    // neighbour coordinates through LDS; a query's K x 3 floats are one contiguous, 16-byte aligned run of out_pts
    sync();
    float* st = reinterpret_cast<float*>(s_mem);   // [32 queries][K][3]
    if (h == 0) {
#pragma unroll
        for (int r = 0; r < K; ++r) {
            const int id = mi[r] == 0x7fffffff ? 0 : mi[r];
            const float* p = pcb + (size_t)id * 3;
            st[(j * K + r) * 3 + 0] = OFFSETS ? p[0] - qx : p[0];
            st[(j * K + r) * 3 + 1] = OFFSETS ? p[1] - qy : p[1];
            st[(j * K + r) * 3 + 2] = OFFSETS ? p[2] - qz : p[2];
        }
    }
    sync();
    static_assert((K * 3) % 4 == 0, "a query's output row must be whole 16-byte chunks");
    constexpr int CH = K * 3 / 4;                  // chunks per query row
    float* dst = out_pts + (size_t)b * Q * K * 3;
    const bool aligned = (reinterpret_cast<uintptr_t>(dst) & 15) == 0;
    for (int c = lane; c < 32 * CH; c += 64) {
        const int row = c / CH, ch = c - row * CH, qr = s_row[row];
        if (qr < 0) continue;
        float* d = dst + (size_t)qr * K * 3 + ch * 4;
        const float4 v = reinterpret_cast<const float4*>(st)[c];
        if (aligned) *reinterpret_cast<float4*>(d) = v;
        else { d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w; }
    }
}

// MODE 0: grid (ceil(Q / 32), B) x 64 threads, the search proper.  MODE 1: grid (KG_MAXDEF) x 512 threads: wave `part` of parked
// group `slot`.
template <int K, bool OFFSETS, int MODE>
__global__ __launch_bounds__(MODE ? 64 * KG_SPLIT : 64, MODE ? 2 : 4) void knn_grid_kernel(const float* __restrict__ X, const float* __restrict__ pc, long long pc_stride,
                                                         const int* __restrict__ qperm, const float4* __restrict__ cand,
                                                         const float4* __restrict__ boxes, const KgCloud* __restrict__ hdr,
                                                         long long cand_stride, long long box_stride, int n_sub,
                                                         long long* __restrict__ out_idx, float* __restrict__ out_dist,
                                                         float* __restrict__ out_pts, int Q, int* __restrict__ def_count,
                                                         char* __restrict__ def_scratch) {
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    __shared__ __attribute__((aligned(16))) char s_all[(MODE ? KG_SPLIT : 1) * KG_LDS_BYTES];     // per wave: [lb | ring | queue]; reused for merges / staging
    __shared__ int s_row[32];
    static_assert(2 * K * 32 * 8 <= KG_LDS_BYTES && 32 * K * 3 * 4 <= KG_LDS_BYTES, "merge / staging buffers do not fit");
    static_assert(K == 16, "the parked-group scratch layout is written for k = 16");
    const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
    const int part = MODE ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : 0;
    char* s_mem = s_all + part * KG_LDS_BYTES;
    int b = blockIdx.y, group = blockIdx.x, slot = 0;
    float prev = -1.f;
    if (MODE == 1) {
        slot = blockIdx.x;
        if (slot >= min(*def_count, KG_MAXDEF)) return;
        const KgSlot* sl = kg_slot_hdr(def_scratch, slot);
        group = sl->group; b = sl->b; prev = sl->prev;
    }
    KG_T0();
    KG_LOCAL(int dbg_tiles = 0; int dbg_rounds = 0; int dbg_flushes = 0; int dbg_mine = 0);
    float* s_lb = reinterpret_cast<float*>(s_mem);
    float4* s_ring = reinterpret_cast<float4*>(s_mem + KG_LDS_LB);
    unsigned short* qi = reinterpret_cast<unsigned short*>(s_mem + KG_LDS_LB + KG_LDS_RING) + lane;       // [slot][lane]
    const float4* cb = cand + (size_t)b * cand_stride;
    const float4* bb = boxes + (size_t)b * box_stride;
    const float* pcb = pc + (size_t)b * pc_stride;
    const float pmax2 = hdr[b].pmax2;              // (needed late: the load overlaps the set-up)
    const int sp = group * 32 + j;
    const bool valid = sp < Q;
    const int q = qperm[(size_t)b * Q + (valid ? sp : Q - 1)];       // a lane without a query repeats the last one
    const float* xq = X + ((size_t)b * Q + q) * 3;
    const float qx = xq[0], qy = xq[1], qz = xq[2];
    float tau_parked = __builtin_inff();           // MODE 1: the query's 16th distance when its group was parked
    int s0 = -1;                                   // MODE 0: the seed sub-tile
    KG_T(0);
    if (MODE == 0) {
        // the boxes of the wave's queries, one per group of 8 consecutive sorted positions (tighter than one box of all 32)
        float glo[4][3], ghi[4][3];
        {
            float lo[3] = {qx, qy, qz}, hi[3] = {qx, qy, qz};
#pragma unroll
            for (int o = 4; o > 0; o >>= 1)
#pragma unroll
                for (int a = 0; a < 3; ++a) { lo[a] = fminf(lo[a], __shfl_xor(lo[a], o, 64)); hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], o, 64)); }
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    glo[g][a] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, lo[a]), g * 8));
                    ghi[g][a] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, hi[a]), g * 8));
                }
        }
        // ---- lower bounds of all sub-tiles (guarded: see the header), once; the nearest one seeds the lists
        float best = __builtin_inff();
        int best_t = 0;
        auto lower_bound = [&](const float4 tl, const float4 th) -> float {        // min over the four query boxes of the box-to-box distance^2
            float m = __builtin_inff();
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float gx = fmaxf(0.f, fmaxf(tl.x - ghi[g][0], glo[g][0] - th.x));
                const float gy = fmaxf(0.f, fmaxf(tl.y - ghi[g][1], glo[g][1] - th.y));
                const float gz = fmaxf(0.f, fmaxf(tl.z - ghi[g][2], glo[g][2] - th.z));
                m = fminf(m, (gx * gx + gy * gy) + gz * gz);
            }
            return m * 0.99999f;
        };
        for (int t0 = 0; t0 < n_sub; t0 += 256) {      // four independent box loads in flight
            float4 tl[4], th[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int t = min(t0 + u * 64 + lane, n_sub - 1);
                tl[u] = bb[2 * t]; th[u] = bb[2 * t + 1];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int t = t0 + u * 64 + lane;
                if (t < n_sub) {
                    const float lb = lower_bound(tl[u], th[u]);
                    s_lb[t] = lb;
                    if (lb < best) { best = lb; best_t = t; }
                }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ob = __shfl_xor(best, o, 64);
            const int ot = __shfl_xor(best_t, o, 64);
            if (ob < best || (ob == best && ot < best_t)) { best = ob; best_t = ot; }
        }
        s0 = best_t;
    } else {
        const float* lbp = kg_slot_lb(def_scratch, slot);
        for (int t = lane; t < n_sub; t += 64) s_lb[t] = lbp[t];
        const unsigned long long* L = kg_slot_lists(def_scratch, slot);
        const unsigned t0_ = (unsigned)(L[(0 * K + K - 1) * 32 + j] >> 32), t1_ = (unsigned)(L[(1 * K + K - 1) * 32 + j] >> 32);
        tau_parked = __builtin_bit_cast(float, min(t0_, t1_));
    }
    KG_T(1);
    const float q2 = (qx * qx + qy * qy) + qz * qz, qn = sqrtf(q2);
    const float b0 = h ? -2.f * qy : -2.f * qx, b1 = h ? 1.f : -2.f * qz;
    float eps;
    {
        const float e = qn + sqrtf(pmax2);
        eps = 9.5367431640625e-07f * (e * e);                         // 2^-20 (|x| + max |p|)^2
    }
    unsigned khi[K], klo[K];
#pragma unroll
    for (int r = 0; r < K; ++r) { khi[r] = 0x7f800000u; klo[r] = 0x7fffffffu; }      // (+inf, no index)
    float tau = tau_parked, thr = __builtin_inff(), R2 = __builtin_inff();
    int cnt = 0, staged = 0, visited = 0;
    f32x16 zero;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero[r] = 0.f;
    asm volatile("" : "+v"(zero));
    auto flush = [&]() {                           // exact distances of the queued candidates (ring positions), insertion, new thresholds
        int maxc = cnt;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) maxc = max(maxc, __shfl_xor(maxc, o, 64));
        KG_LOCAL(dbg_rounds += maxc; dbg_flushes += 1; dbg_mine += cnt);
        for (int sidx = 0; sidx < maxc; ++sidx)
            if (sidx < cnt) {
                const float4 p = s_ring[qi[sidx * 64]];
                const float d = knn_d2(qx, qy, qz, p);
                kg_insert<K>(khi, klo, __builtin_bit_cast(unsigned, d), __builtin_bit_cast(unsigned, p.w));
            }
        tau = fminf(__builtin_bit_cast(float, khi[K - 1]), tau_parked);
        thr = (tau - q2) + eps;
        cnt = 0; staged = 0;
        float tq = fminf(tau, __shfl_xor(tau, 32, 64));              // either half list bounds the query's 16th distance
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) tq = fmaxf(tq, __shfl_xor(tq, o, 64));
        R2 = tq;
    };
    auto fetch = [&](int t) -> float4 { return cb[t * 32 + j]; };
    auto process = [&](const float4 p) {           // p = candidate j of the sub-tile (both lane halves hold it)
        KG_LOCAL(dbg_tiles += 1);
        if (h == 0) s_ring[staged * 32 + j] = p;
        const float pn = (p.x * p.x + p.y * p.y) + p.z * p.z;
        f32x16 acc = __builtin_amdgcn_mfma_f32_32x32x2f32(h ? p.y : p.x, b0, zero, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(h ? pn : p.z, b1, acc, 0, 0, 0);
        unsigned mask = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r)
            mask = __builtin_amdgcn_alignbit(mask, __builtin_bit_cast(unsigned, acc[r] - thr), 31);
        const int base = staged * 32 + 4 * h;
        while (__any(mask != 0u)) {
            if (mask != 0u) {
                const int pbit = 31 - __builtin_clz(mask), r = 15 - pbit;
                qi[cnt * 64] = (unsigned short)(base + (r & 3) + 8 * (r >> 2));
                ++cnt;
                mask &= ~(1u << pbit);
            }
        }
        ++staged; ++visited;
        __builtin_amdgcn_wave_barrier();           // (the ring rows written above are read by other lanes of this wave in flush)
        if (staged == KG_RING || __any(cnt > KM_QCAP - 16)) flush();
    };
    // one pass over the sub-tiles with lower < lb <= upper (never the seed); take(ordinal) says whether this wave visits the ordinal-th
    // of them; a sub-tile whose lb exceeds the CURRENT bound R2 when it comes up is dropped
    auto visit = [&](const float lower, const float upper, auto take) {
        int c = -64, seen = 0;
        unsigned long long m = 0;
        auto next = [&]() -> int {                 // next sub-tile of the pass, -1 at the end
            for (;;) {
                while (m == 0) {
                    c += 64;
                    if (c >= n_sub) return -1;
                    const int t = c + lane;
                    const bool ok = t < n_sub && t != s0;
                    const float lb = ok ? s_lb[t] : __builtin_inff();
                    m = __ballot(ok && lb > lower && lb <= upper);
                }
                const int bit = __ffsll((long long)m) - 1;
                m &= m - 1;
                if (take(seen++)) return c + bit;
            }
        };
        int t_cur = next();
        float4 p_cur = fetch(max(t_cur, 0));
        while (t_cur >= 0) {
            const int t_nxt = next();
            const float4 p_nxt = fetch(max(t_nxt, 0));
            if (!(s_lb[t_cur] > R2)) process(p_cur);
            t_cur = t_nxt;
            p_cur = p_nxt;
        }
    };
    KG_T(2);
    if (MODE == 0) {
        process(fetch(s0));                        // the seed fills both half lists (16 candidates each)
        // ---- rounds: sub-tiles with prev < lb <= limit (limit fixed per round: no sub-tile is visited twice)
#pragma unroll 1
        for (int round = 0; round < KG_ROUNDS; ++round) {
            flush();
            const float limit = round == KG_ROUNDS - 1 ? R2 : kg_round_frac(round) * R2;
            if (!(limit > prev)) continue;
            if (def_count) {                       // heavy so far + the band ahead: park the group (see the header)
                int band = 0;
                for (int c = 0; c < n_sub; c += 64) {
                    const int t = c + lane;
                    const float lb = t < n_sub && t != s0 ? s_lb[t] : __builtin_inff();
                    band += __popcll(__ballot(lb > prev && lb <= limit));
                }
                if (visited + band > KG_HEAVY) {
                    int sl = 0;
                    if (lane == 0) sl = atomicAdd(def_count, 1);
                    sl = __builtin_amdgcn_readfirstlane(sl);
                    if (sl < KG_MAXDEF) {
                        KgSlot* hd = kg_slot_hdr(def_scratch, sl);
                        if (lane == 0) { hd->group = group; hd->b = b; hd->prev = prev; hd->seed = s0; }
                        float* lbp = kg_slot_lb(def_scratch, sl);
                        for (int t = lane; t < n_sub; t += 64) lbp[t] = s_lb[t];
                        unsigned long long* L = kg_slot_lists(def_scratch, sl);
#pragma unroll
                        for (int r = 0; r < K; ++r) L[(h * K + r) * 32 + j] = ((unsigned long long)khi[r] << 32) | klo[r];
                        return;
                    }
                }
            }
            visit(prev, limit, [](int) { return true; });
            prev = limit;
        }
        flush();
        KG_T(3);
        KG_LOCAL(KG_COUNT(0, dbg_tiles); KG_COUNT(1, dbg_rounds); KG_COUNT(2, dbg_flushes);
                 int mx_ = dbg_mine, sm_ = dbg_mine;
                 for (int o_ = 32; o_ > 0; o_ >>= 1) { mx_ = max(mx_, __shfl_xor(mx_, o_, 64)); sm_ += __shfl_xor(sm_, o_, 64); }
                 KG_COUNT(3, mx_); KG_COUNT(4, sm_ / 64));
        kg_output<K, OFFSETS, true>(khi, klo, s_mem, s_row, lane, b, Q, q, valid, qx, qy, qz, pcb, out_idx, out_dist, out_pts);
        KG_T(4);
    } else {
        s0 = kg_slot_hdr(def_scratch, slot)->seed; // (the parked wave's seed: already in its lists)
        flush();                                   // thresholds from the parked 16th distances
        const float upper = R2;
        visit(prev, upper, [&](int ordinal) { return ordinal % KG_SPLIT == part; });
        flush();
        // ---- fold the KG_SPLIT list sets pairwise: waves [n, 2n) hand theirs to waves [0, n) through LDS; every step inserts at most
        // K keys per lane (ascending: stop when no lane's entry beats its 16th any more)
        auto absorb = [&](const unsigned long long* src) {       // src: [half][K][32 queries]
            for (int r = 0; r < K; ++r) {
                const unsigned long long k = src[(h * K + r) * 32 + j];
                if (!__any(k < (((unsigned long long)khi[K - 1] << 32) | klo[K - 1]))) break;
                kg_insert<K>(khi, klo, (unsigned)(k >> 32), (unsigned)k);
            }
        };
#pragma unroll 1
        for (int n = KG_SPLIT / 2; n >= 1; n >>= 1) {
            __syncthreads();                       // every wave is out of its own LDS area (search / previous step)
            if (part >= n && part < 2 * n) {
                unsigned long long* dst = reinterpret_cast<unsigned long long*>(s_all + (part - n) * KG_LDS_BYTES);
#pragma unroll
                for (int r = 0; r < K; ++r) dst[(h * K + r) * 32 + j] = ((unsigned long long)khi[r] << 32) | klo[r];
            }
            __syncthreads();
            if (part < n) absorb(reinterpret_cast<const unsigned long long*>(s_mem));
        }
        if (part != 0) return;                     // (no barrier below involves the other waves: kg_output's are wave-local in effect)
        absorb(kg_slot_lists(def_scratch, slot));
        kg_output<K, OFFSETS, false>(khi, klo, s_mem, s_row, lane, b, Q, q, valid, qx, qy, qz, pcb, out_idx, out_dist, out_pts);
    }
}

// Second pass of the split segmented search: workgroup i = blocks[i] (128 threads = its <= 128 query rows); a thread merges the n_split
// ascending key lists of its query (K smallest, lexicographic = (d2, index)) and the block writes the neighbours' offsets from the
// query through LDS as one contiguous range (like knn_mfma_kernel's epilogue).
template <int K>
__global__ __launch_bounds__(64) void knn_split_merge_kernel(const float* __restrict__ X, const float* __restrict__ pc,
                                                             const long long* __restrict__ pc_off, const int4* __restrict__ blocks,
                                                             const unsigned long long* __restrict__ part, int n_split, int Q,
                                                             float* __restrict__ out_pts, int seg_cand = 0) {
    // grid = (query blocks, 2): 64 threads = one half of a block's <= 128 query rows.  The key lists of the half (n_split x 64 x K keys:
    // contiguous per slice) are copied to the LDS in one round of independent loads first -- read straight from global memory, the K
    // selection rounds were K dependent round trips to L2 / HBM (27-45 us for a merge of 23 k queries).
    extern __shared__ __attribute__((aligned(16))) unsigned long long sm_keys[];      // [n_split][64][K + 1] keys, then 64 x K x 3 floats
    constexpr int KP = K + 1;                                                          // (+1: a thread's lists start on different banks)
    const int4 bk = blocks[blockIdx.x];
    const int r0 = blockIdx.y * 64;
    const int q_first = bk.y + r0, n_valid = max(0, min(64, bk.z - r0));
    if (n_valid <= 0) return;
    const float* pcb = pc + (size_t)pc_off[bk.x] * 3;
    const int t = threadIdx.x;
    if (seg_cand > 0) n_split = min(n_split, max(1, (int)((pc_off[bk.x + 1] - pc_off[bk.x] + seg_cand - 1) / seg_cand)));   // (the search's slice count)
    float* st = reinterpret_cast<float*>(sm_keys + (size_t)n_split * 64 * KP);
    // a slice's lists of the 64 rows are one contiguous run of 64 x K keys: 16 bytes per lane and load, eight loads in flight per lane
    // and slice (the run is 16-byte aligned: K * 8 bytes per row); rows beyond n_valid stay unread
    static_assert(K == 16, "the staging below moves 8 x 64 x 16 bytes per slice");
    for (int w = 0; w < n_split; ++w) {
        const uint4* src = reinterpret_cast<const uint4*>(part + ((size_t)w * Q + q_first) * K);
        uint4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int c = u * 64 + t;                                     // 16-byte piece c = (row c / 8, keys 2 (c % 8), + 1)
            v[u] = (c >> 3) < n_valid ? src[c] : make_uint4(~0u, ~0u, ~0u, ~0u);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int c = u * 64 + t;
            unsigned long long* d = sm_keys + (w * 64 + (c >> 3)) * KP + 2 * (c & 7);
            d[0] = (unsigned long long)v[u].x | ((unsigned long long)v[u].y << 32);
            d[1] = (unsigned long long)v[u].z | ((unsigned long long)v[u].w << 32);
        }
    }
    __syncthreads();
    if (t < n_valid) {
        const int q = q_first + t;
        const float qx = X[(size_t)q * 3], qy = X[(size_t)q * 3 + 1], qz = X[(size_t)q * 3 + 2];
        // the lists' current keys live in registers (slot w of cur[] is only ever indexed by unrolled loops); the K winners first, then
        // their points in K independent gathers
        unsigned long long cur[8];
        int head[8];
#pragma unroll
        for (int w = 0; w < 8; ++w) {
            head[w] = 0;
            cur[w] = w < n_split ? sm_keys[(w * 64 + t) * KP] : ~0ull;
        }
        int ids[K];
#pragma unroll
        for (int r = 0; r < K; ++r) {
            unsigned long long best = cur[0];
            int best_w = 0;
#pragma unroll
            for (int w = 1; w < 8; ++w)
                if (cur[w] < best) { best = cur[w]; best_w = w; }
#pragma unroll
            for (int w = 0; w < 8; ++w)
                if (w == best_w) {
                    ++head[w];
                    cur[w] = head[w] < K ? sm_keys[(w * 64 + t) * KP + head[w]] : ~0ull;
                }
            const unsigned bi = (unsigned)best;
            ids[r] = bi == 0x7fffffffu ? 0 : (int)bi;                     // (an unfilled slot: cannot happen with >= K candidates)
        }
        float px[K], py[K], pz[K];
#pragma unroll
        for (int r = 0; r < K; ++r) {
            const float* p = pcb + (size_t)ids[r] * 3;
            px[r] = p[0]; py[r] = p[1]; pz[r] = p[2];
        }
#pragma unroll
        for (int r = 0; r < K; ++r) {
            st[(t * K + r) * 3 + 0] = px[r] - qx;
            st[(t * K + r) * 3 + 1] = py[r] - qy;
            st[(t * K + r) * 3 + 2] = pz[r] - qz;
        }
    }
    __syncthreads();
    const int n_f = n_valid * K * 3;
    float* dst = out_pts + (size_t)q_first * K * 3;
    if ((reinterpret_cast<uintptr_t>(dst) & 15) == 0 && (reinterpret_cast<uintptr_t>(st) & 15) == 0) {
        for (int c = t; c * 4 + 3 < n_f; c += 64) reinterpret_cast<float4*>(dst)[c] = reinterpret_cast<const float4*>(st)[c];
        for (int e = (n_f & ~3) + t; e < n_f; e += 64) dst[e] = st[e];
    } else {
        for (int e = t; e < n_f; e += 64) dst[e] = st[e];
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
// split_ws (optional, knn16_segmented_split_floats(T) floats): lets a launch with few query blocks split every job's candidates over
// up to KNN_SEG_SPLIT workgroups per block (+ one merge launch); same neighbours in the same order.  MCR_KNN_SEG_SPLIT=0: never (A/B)
#ifndef MCR_KNN_SEG_SPLIT_MAX
#define MCR_KNN_SEG_SPLIT_MAX 8
#endif
#ifndef MCR_KNN_SEG_TARGET
#define MCR_KNN_SEG_TARGET 768
#endif
constexpr int KNN_SEG_SPLIT = MCR_KNN_SEG_SPLIT_MAX;
size_t knn16_segmented_split_floats(int64_t T) { return (size_t)KNN_SEG_SPLIT * T * 16 * 2; }
void launch_knn16_segmented(hipStream_t s, const float* X, const float* pc, const long long* pc_off, const int* blocks,
                            int64_t n_blocks, int64_t T, float* offsets_out, float* split_ws, int slice) {
    if (n_blocks <= 0) return;
    static const bool split_on = []() { const char* e = getenv("MCR_KNN_SEG_SPLIT"); return !(e && e[0] == '0'); }();
    static const bool use_mfma = []() { const char* e = getenv("MCR_KNN_MFMA"); return !(e && e[0] == '0'); }();
    // four waves per block: ~3 waves per SIMD need 768 blocks
    // slice: candidates per slice of a job (0: the launch is not split); MCR_KNN_SEG_CAND=0: one slice count for every job instead (A/B)
    static const bool per_job = []() { const char* e = getenv("MCR_KNN_SEG_CAND"); return !(e && e[0] == '0'); }();
    const int seg_cand = per_job ? slice : 0;
    int n_split = 1;
    if (split_on && use_mfma && split_ws && slice > 0)
        n_split = (int)std::min<int64_t>(KNN_SEG_SPLIT, std::max<int64_t>(1, (seg_cand > 0 ? 4 * MCR_KNN_SEG_TARGET : MCR_KNN_SEG_TARGET) / n_blocks));
    if (n_split > 1) {
        unsigned long long* part = reinterpret_cast<unsigned long long*>(split_ws);
        hipLaunchKernelGGL((knn_mfma_kernel<16, true>), dim3((unsigned)n_blocks, (unsigned)n_split), dim3(KNN_BLOCK), 0, s, X, pc,
                           (long long*)nullptr, (float*)nullptr, offsets_out, (int)T, 0, reinterpret_cast<const int4*>(blocks), pc_off, n_split, part,
                           seg_cand);
        const size_t merge_lds = (size_t)n_split * 64 * 17 * 8 + 64 * 16 * 3 * 4;                    // <= 81 920 bytes at 8 slices
        // the opt-in to > 64 KB of dynamic LDS is a per-DEVICE attribute of the function: set once for every device this process drives
        static bool lds_set[64] = {};
        int dev_id = 0;
        if (hipGetDevice(&dev_id) != hipSuccess || dev_id < 0 || dev_id >= 64) dev_id = 0;
        if (!lds_set[dev_id]) {
            if (hipFuncSetAttribute((const void*)knn_split_merge_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    8 * 64 * 17 * 8 + 64 * 16 * 3 * 4) != hipSuccess) {
                // (the HIP error stays pending: the caller's MCR_LAUNCH_CHECK returns it -- offsets_out was not written)
                set_error("launch_knn16_segmented: cannot reserve %d bytes of LDS for the merge kernel on device %d", 8 * 64 * 17 * 8 + 64 * 16 * 3 * 4,
                          dev_id);
                return;
            }
            lds_set[dev_id] = true;
        }
        hipLaunchKernelGGL((knn_split_merge_kernel<16>), dim3((unsigned)n_blocks, 2), dim3(64), merge_lds, s, X, pc, pc_off,
                           reinterpret_cast<const int4*>(blocks), (const unsigned long long*)part, n_split, (int)T, offsets_out, seg_cand);
        return;
    }
    launch_knn<16>(true, dim3((unsigned)n_blocks, 1), s, X, pc, nullptr, nullptr, offsets_out, (int)T, 0,
                   reinterpret_cast<const int4*>(blocks), pc_off);
}
constexpr int knn_block_rows() { return MCR_WAVE * KNN_QT; }
int knn_rows_per_block() { return knn_block_rows(); }
}  // namespace mcr

namespace mcr {
static inline size_t kg_al(size_t n) { return (n + 255) & ~(size_t)255; }
bool knn_grid_applicable(int64_t M, int k) {
    static const bool on = []() { const char* e = getenv("MCR_KNN_GRID"); return !(e && e[0] == '0'); }();      // dev A/B knob
    return on && k == 16 && M >= KG_MIN_M && M <= KG_MAX_M;
}
size_t knn_grid_query_bytes(int64_t B, int64_t Q) {
    return kg_al(B * 8 * 4) + kg_al((size_t)B * KG_QCELLS * 4) + 3 * kg_al((size_t)B * Q * 4);
}
size_t knn_grid_cloud_bytes(int64_t B, int64_t M) {
    const size_t Mpad = (M + 31) & ~(size_t)31;
    return kg_al(B * sizeof(KgCloud)) + kg_al(B * Mpad * 16) + kg_al(B * (Mpad / 32) * 32);
}
// qperm [B][Q] (device, inside ws): the queries of every cloud in the order the grid kernel walks them
const int* knn_grid_order_queries(hipStream_t s, const float* X, int64_t B, int64_t Q, void* ws, void* park_ws, bool launch) {
    if (park_ws && launch) (void)hipMemsetAsync(park_ws, 0, 256, s);         // the parked-group counters of the launches that will use this order
    char* w = (char*)ws;
    int* qbox = (int*)w; w += kg_al(B * 8 * 4);
    int* hist = (int*)w; w += kg_al((size_t)B * KG_QCELLS * 4);
    int* code = (int*)w; w += kg_al((size_t)B * Q * 4);
    int* rank = (int*)w; w += kg_al((size_t)B * Q * 4);
    int* qperm = (int*)w;
    if (!launch) return qperm;                                       // an earlier call on this workspace built it (phase 2 of a split forward)
    hipLaunchKernelGGL(kg_qinit_kernel, dim3(KG_QCELLS / 1024, (unsigned)B), dim3(1024), 0, s, hist, qbox);
    hipLaunchKernelGGL(kg_qbbox_kernel, dim3((unsigned)std::min<int64_t>(cdiv(Q, 1024), 64), (unsigned)B), dim3(256), 0, s, X, (int)Q, qbox);
    hipLaunchKernelGGL(kg_qcell_kernel, dim3((unsigned)cdiv(Q, 256), (unsigned)B), dim3(256), 0, s, X, (int)Q, qbox, hist, code, rank);
    hipLaunchKernelGGL(kg_qscan_kernel, dim3((unsigned)B), dim3(KG_BUILD_BLOCK), 0, s, hist);
    hipLaunchKernelGGL(kg_qscatter_kernel, dim3((unsigned)cdiv(Q, 256), (unsigned)B), dim3(256), 0, s, (int)Q, hist, code, rank, qperm);
    return qperm;
}
// n <= 3 candidate sets (the scales of one SconeOcc forward) of B clouds each, in ONE launch; set i needs knn_grid_cloud_bytes(B, M[i])
// bytes at ws[i]
void knn_grid_build_clouds(hipStream_t s, int n, const float* const* pc, const int64_t* M, int64_t B, void* const* ws, KnnGridCloud* out) {
    KgBuildArgs args{};
    args.B = (int)B;
    for (int i = 0; i < n; ++i) {
        const int64_t Mpad = (M[i] + 31) & ~(int64_t)31;
        char* w = (char*)ws[i];
        KnnGridCloud& c = out[i];
        c.hdr = w; w += kg_al(B * sizeof(KgCloud));
        c.cand = w; w += kg_al(B * Mpad * 16);
        c.boxes = w;
        c.cand_stride = Mpad; c.box_stride = Mpad / 32 * 2;
        args.pc[i] = pc[i]; args.cand[i] = (float4*)c.cand; args.boxes[i] = (float4*)c.boxes; args.hdr[i] = (KgCloud*)c.hdr; args.M[i] = (int)M[i];
    }
    if (n > 0) hipLaunchKernelGGL(kg_build_cloud_kernel, dim3((unsigned)(n * B)), dim3(KG_BUILD_BLOCK), 0, s, args);
}
KnnGridCloud knn_grid_build_cloud(hipStream_t s, const float* pc, int64_t B, int64_t M, void* ws) {
    KnnGridCloud c;
    knn_grid_build_clouds(s, 1, &pc, &M, B, &ws, &c);
    return c;
}
// scratch of the parked groups: one counter per launch (`launch` < 64; zeroed by knn_grid_order_queries) + KG_MAXDEF slots (re-used by
// consecutive launches on the stream)
size_t knn_grid_park_bytes() { return 256 + (size_t)KG_MAXDEF * KG_SLOT_BYTES; }
// clouds b_first .. b_first + n_b - 1 of a batch whose INPUT arrays are laid out [B][...]; idx / dist / pts: the outputs of these n_b
// clouds, as mcr_knn_points writes them (k = 16).  park_ws (knn_grid_park_bytes(), may be NULL: no parking) + the launch's ordinal
void launch_knn16_grid(hipStream_t s, const float* X, const float* pc, int64_t M, const int* qperm, const KnnGridCloud& c, int64_t b_first,
                       int64_t n_b, int64_t Q, int64_t* idx, float* dist, float* pts, bool offsets, void* park_ws, int launch) {
    if (n_b <= 0 || Q <= 0) return;
    const float* Xb = X + b_first * Q * 3;
    const float* pcb = pc + b_first * M * 3;
    const int* qp = qperm + b_first * Q;
    const float4* cand = (const float4*)c.cand + b_first * c.cand_stride;
    const float4* boxes = (const float4*)c.boxes + b_first * c.box_stride;
    const KgCloud* hdr = (const KgCloud*)c.hdr + b_first;
    long long* i64 = (long long*)idx;             // the outputs are those of cloud b_first already
    float* d = dist;
    float* o = pts;
    static const bool park_on = []() { const char* e = getenv("MCR_KNN_PARK"); return !(e && e[0] == '0'); }();          // dev A/B knob
    int* def_count = park_ws && park_on && launch >= 0 && launch < 64 ? (int*)park_ws + launch : nullptr;
    char* def_scratch = park_ws ? (char*)park_ws + 256 : nullptr;
    dim3 grid((unsigned)cdiv(Q, 32), (unsigned)n_b);
    const int n_sub = (int)(c.cand_stride / 32);
#define KG_LAUNCH(OFF)                                                                                                                     \
    do {                                                                                                                                   \
        hipLaunchKernelGGL((knn_grid_kernel<16, OFF, 0>), grid, dim3(64), 0, s, Xb, pcb, (long long)(M * 3), qp, cand, boxes, hdr,         \
                           (long long)c.cand_stride, (long long)c.box_stride, n_sub, i64, d, o, (int)Q, def_count, def_scratch);           \
        if (def_count)                                                                                                                     \
            hipLaunchKernelGGL((knn_grid_kernel<16, OFF, 1>), dim3(KG_MAXDEF), dim3(64 * KG_SPLIT), 0, s, Xb, pcb, (long long)(M * 3), qp, cand,  \
                               boxes, hdr, (long long)c.cand_stride, (long long)c.box_stride, n_sub, i64, d, o, (int)Q, def_count, def_scratch); \
    } while (0)
    if (offsets) KG_LAUNCH(true);
    else KG_LAUNCH(false);
#undef KG_LAUNCH
}
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

extern "C" size_t mcr_knn_grid_workspace_bytes(int64_t B, int64_t Q, int64_t M) {
    return knn_grid_query_bytes(B, Q) + knn_grid_cloud_bytes(B, M) + knn_grid_park_bytes() + 512;
}

// mcr_knn_points with a scratch buffer: the grid-pruned search where it applies (k = 16, 1024 <= M <= 16384), the brute-force
// kernels otherwise -- identical outputs either way.
extern "C" int mcr_knn_points_grid(const float* X, const float* pc, int64_t* idx, float* dists, float* pts, int64_t B, int64_t Q, int64_t M,
                                   int k, int subtract_query, void* workspace, size_t workspace_bytes, void* stream) {
    if (!knn_grid_applicable(M, k)) return mcr_knn_points(X, pc, idx, dists, pts, B, Q, M, k, subtract_query, stream);
    MCR_REQUIRE(X && pc && pts, "mcr_knn_points_grid: null pointer");
    MCR_REQUIRE(B > 0 && Q > 0 && B <= 65535 && Q < (1ll << 31), "mcr_knn_points_grid: bad problem size B=%ld Q=%ld", (long)B, (long)Q);
    MCR_REQUIRE(workspace && workspace_bytes >= mcr_knn_grid_workspace_bytes(B, Q, M), "mcr_knn_points_grid: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    char* park = (char*)workspace + knn_grid_query_bytes(B, Q) + knn_grid_cloud_bytes(B, M);
    const int* qperm = knn_grid_order_queries(s, X, B, Q, workspace, park);
    const KnnGridCloud c = knn_grid_build_cloud(s, pc, B, M, (char*)workspace + knn_grid_query_bytes(B, Q));
    launch_knn16_grid(s, X, pc, M, qperm, c, 0, B, Q, idx, dists, pts, subtract_query != 0, park, 0);
    MCR_LAUNCH_CHECK("knn_grid_kernel");
    return 0;
}

extern "C" size_t mcr_knn_offsets_segmented_workspace_bytes(int64_t T) { return knn16_segmented_split_floats(T) * sizeof(float); }
extern "C" int mcr_knn_offsets_segmented(const float* X, const float* pc, const int64_t* pc_off, const int* blocks, int64_t n_blocks,
                                         int64_t T, float* offsets_out, void* workspace, size_t workspace_bytes, void* stream) {
    MCR_REQUIRE(X && pc && pc_off && blocks && offsets_out, "mcr_knn_offsets_segmented: null pointer");
    MCR_REQUIRE(n_blocks >= 0 && T >= 0 && T < (1ll << 31) && n_blocks <= 65535ll * 32768, "mcr_knn_offsets_segmented: bad sizes");
    float* ws = workspace && workspace_bytes >= mcr_knn_offsets_segmented_workspace_bytes(T) ? (float*)workspace : nullptr;
    launch_knn16_segmented((hipStream_t)stream, X, pc, (const long long*)pc_off, blocks, n_blocks, T, offsets_out, ws, ws != nullptr ? 2048 : 0);
    MCR_LAUNCH_CHECK("mcr_knn_offsets_segmented");
    return 0;
}

#ifdef KG_DEBUG
extern "C" int mcr_knn_grid_debug(unsigned* out) {      // out[8192 * 16]
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(kg_trace), sizeof(unsigned) * 8192 * 16) == hipSuccess ? 0 : 1;
}
#endif
