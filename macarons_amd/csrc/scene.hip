// Scene-side bookkeeping of one MACARONS decision for gfx950: the grid-cell stores (Scene.fill_cells over Cell.fill,
// macarons_utils.py:2551-2577, :2727-2737), the selection / grouping / cloud building of the occupancy-field pass
// (compute_scene_occupancy_probability_field, :1395-1540) and the per-camera prediction boxes of
// predict_coverage_gain_for_single_camera (:1631-1660), as a handful of launches per phase instead of ~200 small torch launches.
// These are HBM-bound integer / byte passes over <= a few 10^5 points: one coalesced sweep each; nothing is reshaped into a GEMM.
// MCR_HIPCC_FLAGS: -ffp-contract=off
#include "common.h"

namespace mcr {

// ---------------------------------------------------------------------------------------------------------
// K11: segmented nearest-distance in fp64 ("is any point of B within eps of each point of A", per grid cell):
//   dmin[i] = min_j |A[i] - B[j]|  over the B points of A[i]'s segment  (torch.min(torch.cdist(a.double(), b.double())),
//   macarons_utils.py:2566 Cell.fill, :3022 camera_coverage_gain, :3049 scene_coverage).  +inf for an empty B segment.
// grid = (ceil(max_a/256), n_seg): one block column per segment, B streamed through LDS.
__global__ __launch_bounds__(256) void min_dist_seg_kernel(const float* __restrict__ A, const long long* __restrict__ a_off,
                                                           const float* __restrict__ B, const long long* __restrict__ b_off,
                                                           double* __restrict__ dmin, const int* __restrict__ a_index) {
    __shared__ double sb[512 * 3];
    const int seg = blockIdx.y;
    const long long a0 = a_off[seg], a1 = a_off[seg + 1], b0 = b_off[seg], b1 = b_off[seg + 1];
    const long long i = a0 + (long long)blockIdx.x * 256 + threadIdx.x;
    if ((long long)blockIdx.x * 256 >= a1 - a0) return;
    const bool valid = i < a1;
    double ax = 0, ay = 0, az = 0;
    if (valid) {                                      // a_index: A is read through a row index (the candidates in cell order)
        const long long r = a_index ? (long long)a_index[i] : i;
        ax = A[3 * r]; ay = A[3 * r + 1]; az = A[3 * r + 2];
    }
    double best = __builtin_inf();
    for (long long t0 = b0; t0 < b1; t0 += 512) {
        const int nt = (int)min((long long)512, b1 - t0);
        __syncthreads();
        for (int k = threadIdx.x; k < nt * 3; k += 256) sb[k] = (double)B[3 * t0 + k];
        __syncthreads();
        for (int j = 0; j < nt; ++j) {
            const double dx = ax - sb[3 * j], dy = ay - sb[3 * j + 1], dz = az - sb[3 * j + 2];
            best = fmin(best, (dx * dx + dy * dy) + dz * dz);
        }
    }
    if (valid) dmin[i] = sqrt(best);
}

}  // namespace mcr

using namespace mcr;

// ---------------------------------------------------------------------------------------------------------
// Scene-grid bookkeeping of Scene.fill_cells / compute_scene_occupancy_probability_field (macarons_utils.py:2693-2737, :1434): the
// per-point cell lookup, Cell.fill's box tests and the per-cell counts -- fifteen small elementwise launches of the host code as one.
// Bit-compatible with the torch expressions it replaces:
//   d = pts - x_min;  idx = min((d - remainder(d, step)) / step, grid - 1) truncated to an integer, clamped at 0   (utils.floor_divide)
//   key = linear cell id if the point lies in the scene box (closed), strictly inside ITS cell's box (Cell.fill :2552-2557) and is
//   offered (valid), else n_cells.   box_test = 0: the cell id alone (:1434 uses the lookup without the tests).
__device__ __forceinline__ float torch_remainder(float a, float b) {       // torch.remainder on floats: fmod, then the divisor's sign
    float m = fmodf(a, b);
    if (m != 0.f && ((b < 0.f) != (m < 0.f))) m = __fadd_rn(m, b);
    return m;
}
__global__ void cell_keys_kernel(const float* __restrict__ pts, long long N, const unsigned char* __restrict__ valid,
                                 const float* __restrict__ gc, int gl, int gw, int gh, const float* __restrict__ lo_tab,
                                 const float* __restrict__ hi_tab, int box_test, int* __restrict__ key) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int g[3] = {gl, gw, gh};
    float p[3];
    int idx[3];
    bool in_scene = true;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        p[a] = pts[i * 3 + a];
        const float d = __fsub_rn(p[a], gc[a]), st = gc[6 + a];
        float q = __fdiv_rn(__fsub_rn(d, torch_remainder(d, st)), st);
        q = fminf(q, (float)(g[a] - 1));                  // (NaN-propagating in torch; a NaN coordinate fails every test below anyway)
        long long t = (long long)q;                       // .long(): truncation
        idx[a] = (int)(t < 0 ? 0 : t);
        in_scene = in_scene && p[a] >= gc[a] && p[a] <= gc[3 + a];
    }
    const int n_cells = gl * gw * gh;
    int cid = (idx[0] * gw + idx[1]) * gh + idx[2];
    if (cid >= n_cells) cid = n_cells - 1;                // (unreachable: idx[a] <= g[a] - 1)
    if (box_test) {
        bool ok = in_scene && (!valid || valid[i]);
#pragma unroll
        for (int a = 0; a < 3; ++a)
            ok = ok && (__fsub_rn(p[a], hi_tab[cid * 3 + a]) < 0.f) && (__fsub_rn(p[a], lo_tab[cid * 3 + a]) > 0.f);
        key[i] = ok ? cid : n_cells;
    } else {
        key[i] = cid;
    }
}

// counts[k] = #{i : key[i] == k} for k <= nk, offsets = their exclusive prefix sums (nk + 2 entries): ONE block (N is a few 10^5,
// nk <= 1023), no zero-initialised scratch, no second launch for the scan.  The keys are cell ids of points that arrive in image /
// cloud order, i.e. long runs of one value: LDS atomics on one address serialise (64 per wave instruction: 87 us at N = 230k with
// one atomic per key).  A thread therefore takes 32 CONSECUTIVE keys per round (eight 16-byte loads in flight), folds runs of equal
// keys in registers, and a wave whose lanes all end on the same key adds its counts up first and issues ONE atomic.
__global__ __launch_bounds__(1024) void key_histogram_kernel(const int* __restrict__ key, long long N, int nk,
                                                            long long* __restrict__ counts, long long* __restrict__ offsets) {
    __shared__ unsigned s_cnt[1024];
    __shared__ unsigned s_wave[16];
    const int tid = threadIdx.x;
    s_cnt[tid] = 0u;
    __syncthreads();
    const bool al16 = (reinterpret_cast<uintptr_t>(key) & 15) == 0;
    const long long N4 = al16 ? N / 4 : 0;                              // whole int4 groups
    const int4* key4 = reinterpret_cast<const int4*>(key);
    constexpr int G = 8;                                                // int4 groups per thread and round
    for (long long g0 = 0; g0 < N4; g0 += (long long)G * 1024) {       // (block-uniform trip count: the wave votes below are convergent)
        int4 v[G];
#pragma unroll
        for (int u = 0; u < G; ++u) {
            const long long g = g0 + (long long)tid * G + u;
            v[u] = g < N4 ? key4[g] : make_int4(-1, -1, -1, -1);
        }
        int cur = -1;                                                   // the open run (cur < 0: none)
        unsigned n = 0;
        auto take = [&](int k) {
            if (k < 0 || k > nk) return;                                // not a key: ignored
            if (k == cur) { ++n; return; }
            if (n) atomicAdd(&s_cnt[cur], n);                           // a run ended inside the thread's 32 keys: rare
            cur = k; n = 1;
        };
#pragma unroll
        for (int u = 0; u < G; ++u) { take(v[u].x); take(v[u].y); take(v[u].z); take(v[u].w); }
        // the last (usually the only) run of every lane: one atomic per wave when the lanes that have one agree on the key
        const unsigned long long has = __ballot(n > 0);
        if (has) {
            const int first = __builtin_ctzll(has);
            const int kf = __shfl(cur, first, 64);
            if (__all(n == 0 || cur == kf)) {
                unsigned tot = n;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o, 64);
                if ((tid & 63) == first) atomicAdd(&s_cnt[kf], tot);
            } else if (n) {
                atomicAdd(&s_cnt[cur], n);
            }
        }
    }
    for (long long i = N4 * 4 + tid; i < N; i += 1024) {                // the tail (and unaligned inputs): one key at a time
        const int k = key[i];
        if (k >= 0 && k <= nk) atomicAdd(&s_cnt[k], 1u);
    }
    __syncthreads();
    // exclusive scan over k = 0 .. 1023 (entries above nk are zero): lane prefix inside a wave, then the 16 wave totals
    const unsigned c = s_cnt[tid];
    unsigned incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned up = __shfl_up(incl, o, 64);
        if ((tid & 63) >= o) incl += up;
    }
    if ((tid & 63) == 63) s_wave[tid >> 6] = incl;
    __syncthreads();
    unsigned base = 0;
    for (int w = 0; w < (tid >> 6); ++w) base += s_wave[w];
    const unsigned excl = base + incl - c;
    if (tid <= nk) { counts[tid] = c; offsets[tid] = excl; }
    if (tid == nk) offsets[nk + 1] = excl + c;
}

// Cell.fill's admission (:2565-2568) on the sorted candidates: key2 = the cell of a candidate that is offered to a cell with more
// than n_point_min candidates and whose fp64 distance to every stored point exceeds the resolution, else nk
__global__ void admit_keys_kernel(const double* __restrict__ d, const int* __restrict__ key_s, const long long* __restrict__ cand,
                                  long long N, double resolution, long long n_point_min, int nk, int* __restrict__ key2) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int k = key_s[i];
    key2[i] = (k < nk && cand[k] > n_point_min && d[i] > resolution) ? k : nk;
}

// ---------------------------------------------------------------------------------------------------------
// Stable grouping of N rows by a small integer key (0 .. nk; anything else counts as nk): order[pos] = source row, the rows of one key
// contiguous and in ascending row order -- what torch.sort(key, stable=True).indices returns (a 12-launch merge sort there) -- plus
// the counts and their exclusive offsets.  A counting sort in three launches: per-tile histograms; one block turns them into
// per-(tile, key) bases; a stable scatter (rank inside a 64-key chunk by ballots, chunks of a wave in order, waves of a tile in order).
constexpr int GRP_TILE = 2048, GRP_THREADS = 256, GRP_MAXK = 1024;

__device__ __forceinline__ int grp_key(const int* __restrict__ key, long long i, int nk) {
    const int k = key[i];
    return (k < 0 || k > nk) ? nk : k;
}

__global__ __launch_bounds__(GRP_THREADS) void grp_hist_kernel(const int* __restrict__ key, long long N, int nk,
                                                               unsigned* __restrict__ tile_hist) {
    __shared__ unsigned s_cnt[GRP_MAXK];
    for (int k = threadIdx.x; k <= nk; k += GRP_THREADS) s_cnt[k] = 0u;
    __syncthreads();
    // a thread takes 8 CONSECUTIVE keys and folds runs of equal keys in registers (cell ids of points in cloud order come in runs)
    const long long base = (long long)blockIdx.x * GRP_TILE + (long long)threadIdx.x * 8;
    int cur = -1;
    unsigned n = 0;
    for (int u = 0; u < 8; ++u) {
        const long long i = base + u;
        if (i >= N) break;
        const int k = grp_key(key, i, nk);
        if (k == cur) { ++n; continue; }
        if (n) atomicAdd(&s_cnt[cur], n);
        cur = k; n = 1;
    }
    if (n) atomicAdd(&s_cnt[cur], n);
    __syncthreads();
    for (int k = threadIdx.x; k <= nk; k += GRP_THREADS) tile_hist[(size_t)blockIdx.x * (nk + 1) + k] = s_cnt[k];
}

// tile_hist[t][k] <- number of rows with key k in the tiles before t; counts / offsets over the keys.  ONE block, thread = key.
__global__ __launch_bounds__(GRP_MAXK) void grp_scan_kernel(unsigned* __restrict__ tile_hist, int n_tiles, int nk,
                                                            long long* __restrict__ counts, long long* __restrict__ offsets) {
    __shared__ unsigned s_wave[GRP_MAXK / 64];
    const int k = threadIdx.x;
    unsigned run = 0;
    if (k <= nk) {
#pragma unroll 8
        for (int t = 0; t < n_tiles; ++t) {
            const size_t a = (size_t)t * (nk + 1) + k;
            const unsigned c = tile_hist[a];
            tile_hist[a] = run;
            run += c;
        }
    }
    unsigned incl = run;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned up = __shfl_up(incl, o, 64);
        if ((k & 63) >= o) incl += up;
    }
    if ((k & 63) == 63) s_wave[k >> 6] = incl;
    __syncthreads();
    unsigned base = 0;
    for (int w = 0; w < (k >> 6); ++w) base += s_wave[w];
    const unsigned excl = base + incl - run;
    if (k <= nk) { counts[k] = run; offsets[k] = excl; }
    if (k == nk) offsets[nk + 1] = excl + run;
}

__global__ __launch_bounds__(GRP_THREADS) void grp_scatter_kernel(const int* __restrict__ key, long long N, int nk,
                                                                  const unsigned* __restrict__ tile_hist,
                                                                  const long long* __restrict__ offsets, int* __restrict__ order) {
    __shared__ unsigned s_base[GRP_THREADS / 64][GRP_MAXK];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int k = threadIdx.x; k < (GRP_THREADS / 64) * GRP_MAXK; k += GRP_THREADS) (&s_base[0][0])[k] = 0u;
    __syncthreads();
    // wave w owns the 512 consecutive keys [wbase, wbase + 512) of the tile: its own histogram first
    const long long wbase = (long long)blockIdx.x * GRP_TILE + (long long)w * (GRP_TILE / (GRP_THREADS / 64));
    int kk[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const long long i = wbase + u * 64 + lane;
        kk[u] = i < N ? grp_key(key, i, nk) : -1;
        if (kk[u] >= 0) atomicAdd(&s_base[w][kk[u]], 1u);
    }
    __syncthreads();
    for (int k = threadIdx.x; k <= nk; k += GRP_THREADS) {     // first output position of (key k, wave ww) = key offset + tiles before + waves before
        unsigned b = (unsigned)offsets[k] + tile_hist[(size_t)blockIdx.x * (nk + 1) + k];
#pragma unroll
        for (int ww = 0; ww < GRP_THREADS / 64; ++ww) {
            const unsigned c = s_base[ww][k];
            s_base[ww][k] = b;
            b += c;
        }
    }
    __syncthreads();
    const int nbits = 32 - __builtin_clz((unsigned)nk | 1u);
    const unsigned long long below_mask = (1ull << lane) - 1ull;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int k = kk[u];
        const bool act = k >= 0;
        unsigned long long m = __ballot(act);                   // -> the lanes of this chunk that hold the same key
        for (int bit = 0; bit < nbits; ++bit) {
            const bool b = (k >> bit) & 1;
            const unsigned long long bal = __ballot(act && b);
            m &= b ? bal : ~bal;
        }
        if (act) order[s_base[w][k] + (unsigned)__popcll(m & below_mask)] = (int)(wbase + u * 64 + lane);
        __builtin_amdgcn_wave_barrier();
        if (act && (m & below_mask) == 0ull) s_base[w][k] += (unsigned)__popcll(m);       // the first lane of every key group
        __builtin_amdgcn_wave_barrier();
    }
}

static size_t group_ws_bytes(long long N, int nk) { return (size_t)std::max<long long>(cdiv(N, GRP_TILE), 1) * (nk + 1) * sizeof(unsigned); }

static int group_by_key(hipStream_t s, const int* key, long long N, int nk, int* order, long long* counts, long long* offsets,
                        unsigned* tile_hist) {
    const int n_tiles = (int)cdiv(N, GRP_TILE);
    if (n_tiles > 0) {
        hipLaunchKernelGGL(grp_hist_kernel, dim3(n_tiles), dim3(GRP_THREADS), 0, s, key, N, nk, tile_hist);
        MCR_LAUNCH_CHECK("grp_hist_kernel");
    }
    hipLaunchKernelGGL(grp_scan_kernel, dim3(1), dim3(GRP_MAXK), 0, s, tile_hist, n_tiles, nk, counts, offsets);
    MCR_LAUNCH_CHECK("grp_scan_kernel");
    if (n_tiles > 0) {
        hipLaunchKernelGGL(grp_scatter_kernel, dim3(n_tiles), dim3(GRP_THREADS), 0, s, key, N, nk, tile_hist, offsets, order);
        MCR_LAUNCH_CHECK("grp_scatter_kernel");
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------------------
// Scene.fill_cells, device part.  Candidates in cell order are never materialised: the segmented nearest-distance pass and the
// admission read them through `order`.
__global__ void admit_sorted_kernel(const double* __restrict__ d, const int* __restrict__ key, const int* __restrict__ order,
                                    const long long* __restrict__ cand, long long N, double resolution, long long n_point_min, int nk,
                                    int* __restrict__ key2) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int k = key[order[i]];
    key2[i] = (k >= 0 && k < nk && cand[k] > n_point_min && d[i] > resolution) ? k : nk;       // Cell.fill :2562-2568 (fp64 compare)
}

// new store row r = row g[r] of the virtual table [old store | admitted candidates in cell order]
__global__ void fill_gather_kernel(const long long* __restrict__ g, long long n_new, const float* __restrict__ store_pts,
                                   const float* __restrict__ store_fts, long long n_store, int F, const float* __restrict__ pts,
                                   const float* __restrict__ features, const int* __restrict__ order, const int* __restrict__ order2,
                                   float* __restrict__ new_pts, float* __restrict__ new_fts) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_new) return;
    const long long s = g[r];
    const float* sp;
    const float* sf;
    if (s < n_store) {
        sp = store_pts + 3 * s;
        sf = store_fts ? store_fts + (long long)F * s : nullptr;
    } else {
        const long long src = order[order2[s - n_store]];
        sp = pts + 3 * src;
        sf = features ? features + (long long)F * src : nullptr;
    }
    new_pts[3 * r] = sp[0]; new_pts[3 * r + 1] = sp[1]; new_pts[3 * r + 2] = sp[2];
    if (new_fts)
        for (int f = 0; f < F; ++f) new_fts[(long long)F * r + f] = sf ? sf[f] : 0.f;
}

// the same with the row map evaluated here: tab = new_off | b_off | adm_off | pm_off | touched (int64, n_cells + 1 entries each);
// new row r of cell c is row pm[pm_off[c] + local] of the cell's [stored | admitted] rows when the cell was touched (its
// torch.randperm prefix), else its local-th stored row
__global__ void fill_gather_pm_kernel(const int* __restrict__ pm, const long long* __restrict__ tab, int n_cells, long long n_new,
                                      const float* __restrict__ store_pts, const float* __restrict__ store_fts, long long n_store, int F,
                                      const float* __restrict__ pts, const float* __restrict__ features, const int* __restrict__ order,
                                      const int* __restrict__ order2, float* __restrict__ new_pts, float* __restrict__ new_fts) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_new) return;
    const int n1 = n_cells + 1;
    const long long *new_off = tab, *b_off = tab + n1, *adm_off = tab + 2 * n1, *pm_off = tab + 3 * n1, *touched = tab + 4 * n1;
    int lo = 0, hi = n_cells;                              // last cell with new_off[c] <= r (empty cells share an offset: take the last)
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (new_off[mid] <= r) lo = mid; else hi = mid;
    }
    const int c = lo;
    const long long local = r - new_off[c];
    const long long val = touched[c] ? (long long)pm[pm_off[c] + local] : local;
    const long long b_len = b_off[c + 1] - b_off[c];
    const float* sp;
    const float* sf;
    if (val < b_len) {
        const long long s_ = b_off[c] + val;
        sp = store_pts + 3 * s_;
        sf = store_fts ? store_fts + (long long)F * s_ : nullptr;
    } else {
        const long long src = order[order2[adm_off[c] + (val - b_len)]];
        sp = pts + 3 * src;
        sf = features ? features + (long long)F * src : nullptr;
    }
    new_pts[3 * r] = sp[0]; new_pts[3 * r + 1] = sp[1]; new_pts[3 * r + 2] = sp[2];
    if (new_fts)
        for (int f = 0; f < F; ++f) new_fts[(long long)F * r + f] = sf ? sf[f] : 0.f;
}

// ---------------------------------------------------------------------------------------------------------
// Occupancy-field pass (compute_scene_occupancy_probability_field, macarons_utils.py:1395-1540): selection, grouping, job building.
__device__ __forceinline__ int cell_of_point(const float* __restrict__ p3, const float* __restrict__ gc, int gl, int gw, int gh) {
    const int g[3] = {gl, gw, gh};                      // the lookup of cell_keys_kernel without the box tests (:1434)
    int idx[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float d = __fsub_rn(p3[a], gc[a]), st = gc[6 + a];
        float q = __fdiv_rn(__fsub_rn(d, torch_remainder(d, st)), st);
        q = fminf(q, (float)(g[a] - 1));
        const long long t = (long long)q;
        idx[a] = (int)(t < 0 ? 0 : t);
    }
    const int cid = (idx[0] * gw + idx[1]) * gh + idx[2];
    return cid >= gl * gw * gh ? gl * gw * gh - 1 : cid;
}

__device__ __forceinline__ int upper_bound_ll(const long long* __restrict__ a, int n, long long v) {     // first i with a[i] > v
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a[mid] > v) hi = mid; else lo = mid + 1;
    }
    return lo;
}

__global__ void field_init_kernel(int* __restrict__ stored_cell, long long P, long long* __restrict__ visit, int n_visit) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < P) stored_cell[i] = -1;
    if (i < n_visit) visit[i] = 0;
}

// every stored row -> its cell (the store is flat, cells in linear order; column 0 of the features is the proxy point's index)
__global__ void field_stored_kernel(const float* __restrict__ store_fts, int F, long long n_store, const long long* __restrict__ store_off,
                                    int nk, long long P, int* __restrict__ stored_cell) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_store) return;
    const long long p = (long long)store_fts[r * F];
    if (p >= 0 && p < P) stored_cell[p] = upper_bound_ll(store_off, nk + 1, r) - 1;
}

// the admissions of a fill whose gather has not run yet (the decision reads fill and field counts back together)
__global__ void field_pending_kernel(const float* __restrict__ features, int F, const int* __restrict__ order, const int* __restrict__ order2,
                                     const int* __restrict__ key2, const long long* __restrict__ adm_off, int nk, long long N, long long P,
                                     int* __restrict__ stored_cell) {
    const long long a = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= N || a >= adm_off[nk]) return;
    const int pos = order2[a];
    const long long p = (long long)features[(long long)order[pos] * F];
    if (p >= 0 && p < P) stored_cell[p] = key2[pos];
}

__global__ void field_select_kernel(const float* __restrict__ proxy_points, long long P, const float* __restrict__ sup_occ,
                                    const float* __restrict__ oof, float* __restrict__ proba, const int* __restrict__ stored_cell,
                                    const float* __restrict__ gc, int gl, int gw, int gh, int use_mask, int* __restrict__ key_sel,
                                    int* __restrict__ key_oof, long long* __restrict__ visit) {
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const int nk = gl * gw * gh;
    const bool occ = sup_occ[p] > 0.f, in_field = oof[p] < 1.f;
    const bool seen = occ && in_field;
    if (seen) proba[p] = 0.f;                                                                     // :1431
    if (use_mask ? seen : in_field) visit[cell_of_point(proxy_points + 3 * p, gc, gl, gw, gh)] = 1;  // :1434 (same value from every writer)
    const int sc = stored_cell[p];
    key_sel[p] = (sc >= 0 && (!use_mask || occ)) ? sc : nk;
    key_oof[p] = oof[p] > 0.f ? 0 : 1;
}

__device__ __forceinline__ void to_prediction_space(const float* __restrict__ src, const float* __restrict__ xf, float* __restrict__ dst) {
    // xf = M_view (16, row-vector convention) | centre (3) | 1 / (box diagonal); the arithmetic of transform_points_kernel
    const float x = src[0], y = src[1], z = src[2], inv = xf[19];
    dst[0] = ((((x * xf[0] + y * xf[4]) + z * xf[8]) + xf[12]) - xf[16]) * inv;
    dst[1] = ((((x * xf[1] + y * xf[5]) + z * xf[9]) + xf[13]) - xf[17]) * inv;
    dst[2] = ((((x * xf[2] + y * xf[6]) + z * xf[10]) + xf[14]) - xf[18]) * inv;
}

// query row t of the pass: job = the (cell, chunk) it belongs to, source = the t-th selected proxy point of that cell
// jobs [J][4] int64: first position in rows_order, first query row, first cloud row, (unused)
__global__ void field_rows_kernel(const long long* __restrict__ jobs, int J, const float* __restrict__ xf, const int* __restrict__ rows_order,
                                  const float* __restrict__ proxy_points, long long T, int* __restrict__ rows, int* __restrict__ row_job,
                                  float* __restrict__ X_world, float* __restrict__ X_q) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    int lo = 0, hi = J;                                    // last job with first query row <= t
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (jobs[4 * mid + 1] <= t) lo = mid; else hi = mid;
    }
    const int r = rows_order[jobs[4 * lo] + (t - jobs[4 * lo + 1])];
    rows[t] = r;
    row_job[t] = lo;
    const float* src = proxy_points + 3ll * r;
    X_world[3 * t] = src[0]; X_world[3 * t + 1] = src[1]; X_world[3 * t + 2] = src[2];
    to_prediction_space(src, xf + 20 * lo, X_q + 3 * t);
}

// cloud row i of the pass: segs [n_seg][4] int64 = first source row in the surface store, first cloud row, job, (unused)
__global__ void field_cloud_kernel(const long long* __restrict__ segs, int n_seg, const float* __restrict__ xf, const float* __restrict__ S_all,
                                   long long tot, float* __restrict__ pc_all) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= tot) return;
    int lo = 0, hi = n_seg;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (segs[4 * mid + 1] <= i) lo = mid; else hi = mid;
    }
    const long long src = segs[4 * lo] + (i - segs[4 * lo + 1]);
    to_prediction_space(S_all + 3 * src, xf + 20 * segs[4 * lo + 2], pc_all + 3 * i);
}

// view harmonics of row t: move_view_state_to_view_space (scone_utils.py:863-931: column v <- bin perm[v]) then compute_view_harmonics
// (:934-960: a [98] x [98, 64] product).  One wave per row, lane = harmonic, the matrix in LDS.
__global__ __launch_bounds__(256) void field_vh_kernel(const float* __restrict__ view_states, int n_bins, const int* __restrict__ rows,
                                                       const int* __restrict__ bin_perm, const float* __restrict__ mt, long long T,
                                                       float* __restrict__ vh) {
    __shared__ float s_m[128 * 64];
    __shared__ float s_vs[4][128];
    for (int k = threadIdx.x; k < n_bins * 64; k += 256) s_m[k] = mt[k];
    __syncthreads();
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (long long t = (long long)blockIdx.x * 4 + w; t < T; t += (long long)gridDim.x * 4) {
        const long long row = rows ? (long long)rows[t] : t;
        for (int v = lane; v < n_bins; v += 64) s_vs[w][v] = view_states[row * n_bins + (bin_perm ? bin_perm[v] : v)];
        __builtin_amdgcn_wave_barrier();
        float acc = 0.f;
        for (int v = 0; v < n_bins; ++v) acc += s_vs[w][v] * s_m[v * 64 + lane];
        vh[t * 64 + lane] = acc;
        __builtin_amdgcn_wave_barrier();
    }
}

__global__ void field_scatter_kernel(const int* __restrict__ rows, const float* __restrict__ occ, long long T, float* __restrict__ proba) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < T) proba[rows[t]] = occ[t];                                                            // :1525
}

__global__ void field_tail_kernel(const int* __restrict__ oof_order, long long n_oof, const float* __restrict__ proxy_points,
                                  const float* __restrict__ proba, float* __restrict__ X_tail, float* __restrict__ occ_tail) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_oof) return;
    const long long p = oof_order[i];
    X_tail[3 * i] = proxy_points[3 * p]; X_tail[3 * i + 1] = proxy_points[3 * p + 1]; X_tail[3 * i + 2] = proxy_points[3 * p + 2];
    occ_tail[i] = proba[p];
}

// ---------------------------------------------------------------------------------------------------------
// Per-camera prediction boxes of predict_coverage_gain_for_single_camera (macarons_utils.py:1631-1660) for K cameras: centre of the
// bounding box of camera k's sampled points (its first nu[k] rows), moved to the prediction camera's view space; the camera centre
// itself goes to the normalised space too.  One block per camera.
__global__ __launch_bounds__(256) void camera_box_kernel(const float* __restrict__ res, const int* __restrict__ nu, int S,
                                                         const float* __restrict__ Mv, const float* __restrict__ cam_world, float inv_diag,
                                                         float* __restrict__ center, float* __restrict__ cam_view) {
    __shared__ float s_hi[4][3], s_lo[4][3];
    const int k = blockIdx.x, n = nu[k];
    float hi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()}, lo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()};
    for (int s = threadIdx.x; s < n && s < S; s += 256) {
        const float* p = res + ((size_t)k * S + s) * 4;
#pragma unroll
        for (int a = 0; a < 3; ++a) { hi[a] = fmaxf(hi[a], p[a]); lo[a] = fminf(lo[a], p[a]); }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], o, 64));
            lo[a] = fminf(lo[a], __shfl_xor(lo[a], o, 64));
        }
        if ((threadIdx.x & 63) == 0) { s_hi[threadIdx.x >> 6][a] = hi[a]; s_lo[threadIdx.x >> 6][a] = lo[a]; }
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    float cw[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float h = fmaxf(fmaxf(s_hi[0][a], s_hi[1][a]), fmaxf(s_hi[2][a], s_hi[3][a]));
        const float l = fminf(fminf(s_lo[0][a], s_lo[1][a]), fminf(s_lo[2][a], s_lo[3][a]));
        cw[a] = n > 0 ? (h + l) / 2.f : 0.f;                                  // empty frustum: any finite centre
    }
    const float* M = Mv + 16 * k;
    float c[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) c[j] = ((cw[0] * M[j] + cw[1] * M[4 + j]) + cw[2] * M[8 + j]) + M[12 + j];
    center[3 * k] = c[0]; center[3 * k + 1] = c[1]; center[3 * k + 2] = c[2];
    const float x = cam_world[3 * k], y = cam_world[3 * k + 1], z = cam_world[3 * k + 2];
#pragma unroll
    for (int j = 0; j < 3; ++j) cam_view[3 * k + j] = ((((x * M[j] + y * M[4 + j]) + z * M[8 + j]) + M[12 + j]) - c[j]) * inv_diag;
}

// gains[k] = mean_s( vis_u[k, inv[k,s]] * factor(|world_u[k, inv[k,s]] - cam_world[k]|) ) * volume[k]  (0 for an empty frustum): the
// Monte-Carlo duplicates (macarons_utils.py:1668-1671) are read through the inverse map instead of being gathered; the sum runs over
// the samples in order, in double, like macarons_gain_kernel.
__global__ __launch_bounds__(256) void macarons_gain_inv_kernel(const float* __restrict__ vis_u, const float* __restrict__ world_u,
                                                                const long long* __restrict__ inv, const int* __restrict__ nu,
                                                                const float* __restrict__ cam_world, const float* __restrict__ volume,
                                                                float distance_th, int mode, int S, float* __restrict__ gains) {
    __shared__ double s[4];
    const int b = blockIdx.x;
    const float cx = cam_world[3 * b], cy = cam_world[3 * b + 1], cz = cam_world[3 * b + 2];
    double acc = 0.0;
    for (int n = threadIdx.x; n < S; n += 256) {
        const long long u = inv[(size_t)b * S + n];
        const float* p = world_u + ((size_t)b * S + u) * 4;
        const float dx = p[0] - cx, dy = p[1] - cy, dz = p[2] - cz;
        const float d = sqrtf((dx * dx + dy * dy) + dz * dz);
        float f = 1.f;
        if (mode == 1) {
            const float q = d / distance_th;
            f = 1.f / (1.f + q * q);
        } else if (d > distance_th) {
            f = (distance_th * distance_th) / (d * d);
        }
        acc += (double)(vis_u[(size_t)b * S + u] * f);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) gains[b] = nu[b] > 0 ? (float)(((s[0] + s[1]) + (s[2] + s[3])) / (double)S) * volume[b] : 0.f;
}

// ---------------------------------------------------------------------------------------------------------
// K rows of S uniforms in ONE launch, bit for bit what K consecutive torch.rand(S, 1, device=...) calls return (upstream draws the
// sampling uniforms camera by camera, scone_utils.py:1052): torch's kernel gives element s of call c the first output of
// Philox4x32-10(key = seed, counter = (offset / 4 + c, 0, s, 0)) -- one thread per element while S <= grid capacity, the generator's
// offset advancing by 4 per call -- mapped to (0, 1] as 2^-32 + x * 2^-32 in fp32 (rocRAND's uniform_distribution) and 1 -> 0.
__device__ __forceinline__ void philox_round(unsigned (&c)[4], unsigned (&k)[2]) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c[0], p1 = (unsigned long long)0xCD9E8D57u * c[2];
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c[1] ^ k[0], n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c[3] ^ k[1], n3 = (unsigned)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
__global__ void philox_uniform_rows_kernel(unsigned long long seed, unsigned long long block0, int K, int S, int mode, float* __restrict__ out) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long long)K * S) return;
    const unsigned long long blk = block0 + (unsigned long long)(gid / S), sub = (unsigned long long)(gid % S);
    unsigned c[4] = {(unsigned)blk, (unsigned)(blk >> 32), (unsigned)sub, (unsigned)(sub >> 32)};
    unsigned k[2] = {(unsigned)seed, (unsigned)(seed >> 32)};
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k);
        k[0] += 0x9E3779B9u; k[1] += 0xBB67AE85u;
    }
    const float inv32 = 2.3283064e-10f;
    float v = mode == 0 ? __fadd_rn(__fmul_rn((float)c[0], inv32), inv32 / 2.0f)       // cuRAND's mapping
                        : __fadd_rn(inv32, __fmul_rn((float)c[0], inv32));              // rocRAND's
    out[gid] = v == 1.0f ? 0.0f : v;
}

extern "C" {

int mcr_min_dist_segmented(const float* A, const int64_t* a_offsets, const float* B, const int64_t* b_offsets, int64_t n_segments,
                           int64_t max_a_per_segment, double* dmin, void* stream) {
    MCR_REQUIRE(A && a_offsets && B && b_offsets && dmin, "mcr_min_dist_segmented: null pointer");
    MCR_REQUIRE(n_segments > 0 && n_segments <= 65535 && max_a_per_segment > 0, "mcr_min_dist_segmented: bad sizes");
    hipLaunchKernelGGL(min_dist_seg_kernel, dim3((unsigned)cdiv(max_a_per_segment, 256), (unsigned)n_segments), dim3(256), 0,
                       (hipStream_t)stream, A, (const long long*)a_offsets, B, (const long long*)b_offsets, dmin, (const int*)nullptr);
    MCR_LAUNCH_CHECK("min_dist_seg_kernel");
    return 0;
}

int mcr_cell_keys(const float* pts, int64_t N, const unsigned char* valid, const float* grid_consts, int grid_l, int grid_w, int grid_h,
                  const float* lo_tab, const float* hi_tab, int box_test, int* key, void* stream) {
    MCR_REQUIRE(pts && grid_consts && key && N > 0, "mcr_cell_keys: bad arguments");
    MCR_REQUIRE(grid_l > 0 && grid_w > 0 && grid_h > 0 && (long long)grid_l * grid_w * grid_h < (1 << 30), "mcr_cell_keys: bad grid");
    MCR_REQUIRE(!box_test || (lo_tab && hi_tab), "mcr_cell_keys: the box tests need the cells' bounds");
    hipLaunchKernelGGL(cell_keys_kernel, dim3((unsigned)cdiv(N, 256)), dim3(256), 0, (hipStream_t)stream, pts, (long long)N, valid, grid_consts,
                       grid_l, grid_w, grid_h, lo_tab, hi_tab, box_test, key);
    MCR_LAUNCH_CHECK("cell_keys_kernel");
    return 0;
}

int mcr_key_histogram(const int* key, int64_t N, int nk, int64_t* counts, int64_t* offsets, void* stream) {
    MCR_REQUIRE(key && counts && offsets && N >= 0 && nk >= 0 && nk <= 1023, "mcr_key_histogram: bad arguments (nk <= 1023)");
    hipLaunchKernelGGL(key_histogram_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, key, (long long)N, nk, (long long*)counts,
                       (long long*)offsets);
    MCR_LAUNCH_CHECK("key_histogram_kernel");
    return 0;
}

int mcr_admit_keys(const double* d, const int* key_s, const int64_t* cand, int64_t N, double resolution, int64_t n_point_min, int nk,
                   int* key2, void* stream) {
    MCR_REQUIRE(d && key_s && cand && key2 && N > 0 && nk >= 0, "mcr_admit_keys: bad arguments");
    hipLaunchKernelGGL(admit_keys_kernel, dim3((unsigned)cdiv(N, 256)), dim3(256), 0, (hipStream_t)stream, d, key_s, (const long long*)cand,
                       (long long)N, resolution, (long long)n_point_min, nk, key2);
    MCR_LAUNCH_CHECK("admit_keys_kernel");
    return 0;
}
size_t mcr_group_by_key_workspace_bytes(int64_t N, int nk) { return group_ws_bytes(N, nk); }

int mcr_group_by_key(const int* key, int64_t N, int nk, int* order, int64_t* counts, int64_t* offsets, void* workspace,
                     size_t workspace_bytes, void* stream) {
    MCR_REQUIRE(counts && offsets && N >= 0 && nk >= 0 && nk < GRP_MAXK, "mcr_group_by_key: bad arguments (nk <= 1023)");
    MCR_REQUIRE(N == 0 || (key && order), "mcr_group_by_key: null pointer");
    MCR_REQUIRE(N < (1ll << 31), "mcr_group_by_key: N must stay below 2^31");
    MCR_REQUIRE(workspace && workspace_bytes >= group_ws_bytes(N, nk), "mcr_group_by_key: workspace too small");
    return group_by_key((hipStream_t)stream, key, (long long)N, nk, order, (long long*)counts, (long long*)offsets, (unsigned*)workspace);
}

size_t mcr_scene_fill_workspace_bytes(int64_t N, int n_cells) { return group_ws_bytes(N, n_cells); }

int mcr_scene_fill_begin(const float* pts, int64_t N, const unsigned char* valid, const float* grid_consts, int grid_l, int grid_w,
                         int grid_h, const float* lo_tab, const float* hi_tab, const float* store_pts, const int64_t* store_off,
                         double resolution, int64_t n_point_min, int* key, int* order, double* dmin, int* key2, int* order2,
                         int64_t* counts, void* workspace, size_t workspace_bytes, void* stream) {
    MCR_REQUIRE(pts && grid_consts && lo_tab && hi_tab && store_off && key && order && dmin && key2 && order2 && counts,
                "mcr_scene_fill_begin: null pointer");
    const long long nk = (long long)grid_l * grid_w * grid_h;
    MCR_REQUIRE(N > 0 && N < (1ll << 31) && grid_l > 0 && grid_w > 0 && grid_h > 0 && nk < GRP_MAXK, "mcr_scene_fill_begin: bad sizes (< 1024 cells)");
    MCR_REQUIRE(workspace && workspace_bytes >= group_ws_bytes(N, (int)nk), "mcr_scene_fill_begin: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    long long* cand = (long long*)counts;                   // cand [nk+1] | a_off [nk+2] | adm [nk+1] | adm_off [nk+2]
    long long* a_off = cand + nk + 1;
    long long* adm = a_off + nk + 2;
    long long* adm_off = adm + nk + 1;
    hipLaunchKernelGGL(cell_keys_kernel, dim3((unsigned)cdiv(N, 256)), dim3(256), 0, s, pts, (long long)N, valid, grid_consts, grid_l, grid_w,
                       grid_h, lo_tab, hi_tab, 1, key);
    MCR_LAUNCH_CHECK("cell_keys_kernel");
    if (int e = group_by_key(s, key, N, (int)nk, order, cand, a_off, (unsigned*)workspace)) return e;
    // every candidate against the store of ITS cell (fp64 nearest distance); an empty store gives +inf
    hipLaunchKernelGGL(min_dist_seg_kernel, dim3((unsigned)cdiv(N, 256), (unsigned)nk), dim3(256), 0, s, pts, (const long long*)a_off,
                       store_pts ? store_pts : pts, (const long long*)store_off, dmin, (const int*)order);
    MCR_LAUNCH_CHECK("min_dist_seg_kernel");
    hipLaunchKernelGGL(admit_sorted_kernel, dim3((unsigned)cdiv(N, 256)), dim3(256), 0, s, dmin, key, order, (const long long*)cand,
                       (long long)N, resolution, (long long)n_point_min, (int)nk, key2);
    MCR_LAUNCH_CHECK("admit_sorted_kernel");
    return group_by_key(s, key2, N, (int)nk, order2, adm, adm_off, (unsigned*)workspace);
}

int mcr_scene_fill_gather(const int64_t* g, int64_t n_new, const float* store_pts, const float* store_fts, int64_t n_store, int F,
                          const float* pts, const float* features, const int* order, const int* order2, float* new_pts, float* new_fts,
                          void* stream) {
    MCR_REQUIRE(g && new_pts && n_new > 0 && n_store >= 0 && F >= 0, "mcr_scene_fill_gather: bad arguments");
    MCR_REQUIRE(n_store == 0 || store_pts, "mcr_scene_fill_gather: the old store is missing");
    MCR_REQUIRE(!new_fts || F > 0, "mcr_scene_fill_gather: features need F > 0");
    hipLaunchKernelGGL(fill_gather_kernel, dim3((unsigned)cdiv(n_new, 256)), dim3(256), 0, (hipStream_t)stream, (const long long*)g,
                       (long long)n_new, store_pts, store_fts, (long long)n_store, F, pts, features, order, order2, new_pts, new_fts);
    MCR_LAUNCH_CHECK("fill_gather_kernel");
    return 0;
}

int mcr_scene_fill_gather_perm(const int* pm, const int64_t* tables, int n_cells, int64_t n_new, const float* store_pts,
                               const float* store_fts, int64_t n_store, int F, const float* pts, const float* features, const int* order,
                               const int* order2, float* new_pts, float* new_fts, void* stream) {
    MCR_REQUIRE(tables && new_pts && n_new > 0 && n_store >= 0 && F >= 0 && n_cells > 0, "mcr_scene_fill_gather_perm: bad arguments");
    MCR_REQUIRE(n_store == 0 || store_pts, "mcr_scene_fill_gather_perm: the old store is missing");
    MCR_REQUIRE(!new_fts || F > 0, "mcr_scene_fill_gather_perm: features need F > 0");
    hipLaunchKernelGGL(fill_gather_pm_kernel, dim3((unsigned)cdiv(n_new, 256)), dim3(256), 0, (hipStream_t)stream, pm, (const long long*)tables,
                       n_cells, (long long)n_new, store_pts, store_fts, (long long)n_store, F, pts, features, order, order2, new_pts, new_fts);
    MCR_LAUNCH_CHECK("fill_gather_pm_kernel");
    return 0;
}

size_t mcr_field_select_workspace_bytes(int64_t P, int n_cells) { return group_ws_bytes(P, n_cells); }

int mcr_field_select(const float* proxy_points, int64_t P, const float* supervision_occ, const float* out_of_field, float* proxy_proba,
                     const float* store_fts, int F, int64_t n_store, const int64_t* store_off, const float* pend_features,
                     const int* pend_order, const int* pend_order2, const int* pend_key2, const int64_t* pend_adm_off, int64_t pend_N,
                     const float* grid_consts, int grid_l, int grid_w, int grid_h, int use_supervision_occ_mask, int* stored_cell,
                     int* key_sel, int* key_oof, int* rows_order, int* oof_order, int64_t* counts, void* workspace,
                     size_t workspace_bytes, void* stream) {
    MCR_REQUIRE(proxy_points && supervision_occ && out_of_field && proxy_proba && store_off && grid_consts && stored_cell && key_sel &&
                key_oof && rows_order && oof_order && counts, "mcr_field_select: null pointer");
    const long long nk = (long long)grid_l * grid_w * grid_h;
    MCR_REQUIRE(P > 0 && P < (1ll << 31) && nk > 0 && nk < GRP_MAXK && F > 0 && n_store >= 0, "mcr_field_select: bad sizes");
    MCR_REQUIRE(n_store == 0 || store_fts, "mcr_field_select: the store's features are missing");
    MCR_REQUIRE(!pend_features || (pend_order && pend_order2 && pend_key2 && pend_adm_off && pend_N > 0), "mcr_field_select: incomplete pending fill");
    MCR_REQUIRE(workspace && workspace_bytes >= group_ws_bytes(P, (int)nk), "mcr_field_select: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    long long* visit = (long long*)counts;                  // visit [nk+1] | sel_counts [nk+1] | sel_off [nk+2] | oof_counts [2] | oof_off [3]
    long long* sel_counts = visit + nk + 1;
    long long* sel_off = sel_counts + nk + 1;
    long long* oof_counts = sel_off + nk + 2;
    long long* oof_off = oof_counts + 2;
    hipLaunchKernelGGL(field_init_kernel, dim3((unsigned)cdiv(std::max<long long>(P, nk + 1), 256)), dim3(256), 0, s, stored_cell, (long long)P,
                       visit, (int)nk + 1);
    MCR_LAUNCH_CHECK("field_init_kernel");
    if (n_store > 0) {
        hipLaunchKernelGGL(field_stored_kernel, dim3((unsigned)cdiv(n_store, 256)), dim3(256), 0, s, store_fts, F, (long long)n_store,
                           (const long long*)store_off, (int)nk, (long long)P, stored_cell);
        MCR_LAUNCH_CHECK("field_stored_kernel");
    }
    if (pend_features) {
        hipLaunchKernelGGL(field_pending_kernel, dim3((unsigned)cdiv(pend_N, 256)), dim3(256), 0, s, pend_features, F, pend_order, pend_order2,
                           pend_key2, (const long long*)pend_adm_off, (int)nk, (long long)pend_N, (long long)P, stored_cell);
        MCR_LAUNCH_CHECK("field_pending_kernel");
    }
    hipLaunchKernelGGL(field_select_kernel, dim3((unsigned)cdiv(P, 256)), dim3(256), 0, s, proxy_points, (long long)P, supervision_occ,
                       out_of_field, proxy_proba, stored_cell, grid_consts, grid_l, grid_w, grid_h, use_supervision_occ_mask, key_sel, key_oof, visit);
    MCR_LAUNCH_CHECK("field_select_kernel");
    if (int e = group_by_key(s, key_sel, P, (int)nk, rows_order, sel_counts, sel_off, (unsigned*)workspace)) return e;
    return group_by_key(s, key_oof, P, 1, oof_order, oof_counts, oof_off, (unsigned*)workspace);
}

int mcr_field_build(const int64_t* jobs, int J, const int64_t* segs, int n_seg, const float* job_xf, const int* rows_order,
                    const float* proxy_points, const float* S_all, const float* view_states, int n_bins, const int* bin_perm,
                    const float* vh_matrix_t, int64_t T, int64_t tot, int* rows, int* row_job, float* X_world, float* X_q, float* vh,
                    float* pc_all, void* stream) {
    MCR_REQUIRE(jobs && segs && job_xf && rows_order && proxy_points && S_all && view_states && vh_matrix_t && rows && row_job && X_world &&
                X_q && vh && pc_all, "mcr_field_build: null pointer");
    MCR_REQUIRE(J > 0 && n_seg > 0 && T > 0 && tot > 0 && n_bins > 0 && n_bins <= 128, "mcr_field_build: bad sizes (n_bins <= 128)");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(field_rows_kernel, dim3((unsigned)cdiv(T, 256)), dim3(256), 0, s, (const long long*)jobs, J, job_xf, rows_order,
                       proxy_points, (long long)T, rows, row_job, X_world, X_q);
    MCR_LAUNCH_CHECK("field_rows_kernel");
    hipLaunchKernelGGL(field_cloud_kernel, dim3((unsigned)cdiv(tot, 256)), dim3(256), 0, s, (const long long*)segs, n_seg, job_xf, S_all,
                       (long long)tot, pc_all);
    MCR_LAUNCH_CHECK("field_cloud_kernel");
    hipLaunchKernelGGL(field_vh_kernel, dim3((unsigned)std::min<long long>(cdiv(T, 4), 1024)), dim3(256), 0, s, view_states, n_bins, rows, bin_perm,
                       vh_matrix_t, (long long)T, vh);
    MCR_LAUNCH_CHECK("field_vh_kernel");
    return 0;
}

int mcr_field_finish(const int* rows, const float* occ, int64_t T, float* proxy_proba, const int* oof_order, int64_t n_oof,
                     const float* proxy_points, float* X_tail, float* occ_tail, void* stream) {
    MCR_REQUIRE(proxy_proba && proxy_points && T >= 0 && n_oof >= 0, "mcr_field_finish: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    if (T > 0) {
        MCR_REQUIRE(rows && occ, "mcr_field_finish: null pointer");
        hipLaunchKernelGGL(field_scatter_kernel, dim3((unsigned)cdiv(T, 256)), dim3(256), 0, s, rows, occ, (long long)T, proxy_proba);
        MCR_LAUNCH_CHECK("field_scatter_kernel");
    }
    if (n_oof > 0) {
        MCR_REQUIRE(oof_order && X_tail && occ_tail, "mcr_field_finish: null pointer");
        hipLaunchKernelGGL(field_tail_kernel, dim3((unsigned)cdiv(n_oof, 256)), dim3(256), 0, s, oof_order, (long long)n_oof, proxy_points,
                           proxy_proba, X_tail, occ_tail);
        MCR_LAUNCH_CHECK("field_tail_kernel");
    }
    return 0;
}

int mcr_view_harmonics_rows(const float* view_states, int n_bins, const int* rows, const int* bin_perm, const float* vh_matrix_t,
                            int64_t T, float* vh, void* stream) {
    MCR_REQUIRE(view_states && vh_matrix_t && vh && T > 0 && n_bins > 0 && n_bins <= 128, "mcr_view_harmonics_rows: bad arguments");
    hipLaunchKernelGGL(field_vh_kernel, dim3((unsigned)std::min<long long>(cdiv(T, 4), 1024)), dim3(256), 0, (hipStream_t)stream, view_states,
                       n_bins, rows, bin_perm, vh_matrix_t, (long long)T, vh);
    MCR_LAUNCH_CHECK("field_vh_kernel");
    return 0;
}

int mcr_camera_boxes(const float* sampled, const int* n_unique, int64_t K, int S, const float* M_view, const float* cam_world,
                     float inv_diag, float* center, float* cam_view, void* stream) {
    MCR_REQUIRE(sampled && n_unique && M_view && cam_world && center && cam_view && K > 0 && S > 0, "mcr_camera_boxes: bad arguments");
    hipLaunchKernelGGL(camera_box_kernel, dim3((unsigned)K), dim3(256), 0, (hipStream_t)stream, sampled, n_unique, S, M_view, cam_world,
                       inv_diag, center, cam_view);
    MCR_LAUNCH_CHECK("camera_box_kernel");
    return 0;
}

int mcr_macarons_gain_indexed(const float* vis_unique, const float* world_unique, const int64_t* inverse, const int* n_unique,
                              const float* cam_world, const float* volume, float distance_th, int factor_mode, int64_t K, int S,
                              float* gains, void* stream) {
    MCR_REQUIRE(vis_unique && world_unique && inverse && n_unique && cam_world && volume && gains && K > 0 && S > 0,
                "mcr_macarons_gain_indexed: bad arguments");
    MCR_REQUIRE(factor_mode == 0 || factor_mode == 1, "mcr_macarons_gain_indexed: factor_mode must be 0 (threshold) or 1 (smooth)");
    MCR_REQUIRE(distance_th > 0.f, "mcr_macarons_gain_indexed: distance_th must be positive");
    hipLaunchKernelGGL(macarons_gain_inv_kernel, dim3((unsigned)K), dim3(256), 0, (hipStream_t)stream, vis_unique, world_unique,
                       (const long long*)inverse, n_unique, cam_world, volume, distance_th, factor_mode, S, gains);
    MCR_LAUNCH_CHECK("macarons_gain_inv_kernel");
    return 0;
}

int mcr_philox_uniform_rows(uint64_t seed, uint64_t offset, int64_t K, int S, int mapping, float* out, void* stream) {
    MCR_REQUIRE(out && K > 0 && S > 0 && S <= 65536 && (offset & 3) == 0, "mcr_philox_uniform_rows: bad arguments (offset % 4 == 0, S <= 65536)");
    hipLaunchKernelGGL(philox_uniform_rows_kernel, dim3((unsigned)cdiv(K * S, 256)), dim3(256), 0, (hipStream_t)stream,
                       (unsigned long long)seed, (unsigned long long)(offset / 4), (int)K, S, mapping, out);
    MCR_LAUNCH_CHECK("philox_uniform_rows_kernel");
    return 0;
}

}  // extern "C"
