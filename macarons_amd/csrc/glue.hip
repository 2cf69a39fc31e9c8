// Glue kernels either side of the networks for gfx950:
//   K3  view-state binning              macarons/utility/scone_utils.py:799-860  compute_view_state
//   K10 occupancy-weighted proxy sampling macarons/utility/scone_utils.py:1030-1061 sample_proxy_points
//   K2  camera-frustum / range mask     macarons/utility/macarons_utils.py:2400-2435 Camera.get_points_in_fov
//   a3  n-camera coverage gains         macarons/networks/SconeVis.py:289-301
// These are HBM-bound byte/index passes: one coalesced sweep each, no reshaping into GEMMs.
// MCR_HIPCC_FLAGS: -ffp-contract=off
#include "common.h"

namespace mcr {

// ---------------------------------------------------------------------------------------------------------
// K3: one thread per (point, view).  Literal fp32 restatement of the reference's binning (asin / acos / Python-
// style mod / its clamps and wrap-arounds, including the (-n_elev)//2 precedence quirk): the 98-bin state must
// be bit-exact away from bin boundaries.
__device__ __forceinline__ float py_mod(float a, float b) {       // torch.remainder / Python %, b > 0
    float m = fmodf(a, b);
    if (m != 0.f && m < 0.f) m += b;
    return m;
}

// Zero fill by a kernel instead of hipMemsetAsync: inside a captured hipGraph (nbv.GraphedNbvStep) the memset nodes of this
// ROCm build did not take effect on the second and later replays (view_state accumulated across replays); a kernel node does.
__global__ void zero_kernel(unsigned* __restrict__ p, long long n_words) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (long long)gridDim.x * blockDim.x) p[i] = 0u;
}
static void launch_zero(hipStream_t s, void* p, size_t bytes) {       // bytes % 4 == 0, p 4-byte aligned
    const long long n = (long long)(bytes / 4);
    if (n <= 0) return;
    hipLaunchKernelGGL(zero_kernel, dim3((unsigned)std::min<long long>(cdiv(n, 256), 2048)), dim3(256), 0, s, (unsigned*)p, n);
}

// Bin of the direction (x, y, z) on the n_elev x n_azim lattice: get_spherical_coords (CustomGeometry.py:27-45) followed by the
// index arithmetic of compute_view_state (scone_utils.py:830-849), literally, in fp32.
__device__ __forceinline__ long long view_state_bin(float x, float y, float z, int n_elev, int n_azim) {
    const float PI = 3.14159265358979323846f;
    const float r = sqrtf(x * x + y * y + z * z);
    const float yr = y / r;
    float elev = asinf(yr);
    if (yr <= -1.f) elev = -PI / 2;
    if (yr >= 1.f) elev = PI / 2;
    const float q = z / (r * cosf(elev));
    float azim = acosf(q);
    if (q <= -1.f) azim = PI;
    if (q >= 1.f) azim = 0.f;
    if (x < 0.f) azim = -azim;
    // scone_utils.py:830-849
    const float es = (float)(3.14159265358979323846 / (n_elev + 1)), as = (float)(2.0 * 3.14159265358979323846 / n_azim);
    const float me = py_mod(elev, es), ma = py_mod(azim, as);
    float ie = (elev - me) / es, ia = (azim - ma) / as;               // floor_divide (utils.py:113-117)
    if (me > (float)(3.14159265358979323846 / (n_elev + 1) / 2.0)) ie += 1.f;
    if (ma > (float)(2.0 * 3.14159265358979323846 / n_azim / 2.0)) ia += 1.f;
    const int lo_e = -((n_elev + 1) / 2);                              // Python: -n_elev // 2  (= -4 for 7)
    const int lo_a = -((n_azim + 1) / 2);                              // Python: -n_azim // 2  (= -7 for 14)
    if (ie >= (float)n_elev) ie = (float)(n_elev - 1);
    if (ie < (float)lo_e) ie = (float)lo_e;
    if (ia > (float)(n_azim / 2)) ia = (float)lo_a;
    ie += (float)(n_elev / 2);
    if (ia < 0.f) ia += (float)n_azim;
    long long idx = (long long)ie * n_azim + (long long)ia;
    const int nb = n_elev * n_azim;
    idx %= nb;
    if (idx < 0) idx += nb;                                            // Python % on a negative product
    return idx;
}

// Cloud c = p / pts_per_cloud reads its own view positions X_view[c] ([n_clouds, n_view, 3]); `rows` (optional) redirects point p to
// row rows[p] of view_state (the in-place update of a scene-wide state table, macarons_utils.py:2867-2877).
__global__ void view_state_kernel(const float* __restrict__ pts, int pts_dim, const float* __restrict__ X_view,
                                  float* __restrict__ view_state, long long n_points, long long pts_per_cloud, int n_view, int n_elev,
                                  int n_azim, const long long* __restrict__ rows) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= n_points * n_view) return;
    const long long p = gid / n_view;
    const int v = (int)(gid - p * n_view);
    const float* xv = X_view + (p / pts_per_cloud) * (3ll * n_view) + 3 * v;
    const long long idx = view_state_bin(xv[0] - pts[p * pts_dim + 0], xv[1] - pts[p * pts_dim + 1], xv[2] - pts[p * pts_dim + 2],
                                         n_elev, n_azim);
    view_state[(rows ? rows[p] : p) * (long long)(n_elev * n_azim) + idx] = 1.0f;      // idempotent (scone_utils.py:857-858)
}

// ---------------------------------------------------------------------------------------------------------
// K10: sampling.  Convention (oracle/view_state.py sample_proxy_points(exact=True)): C_i = fp64 running sum of
// the occupancies above min_occ (others contribute 0), sample u picks the first i with C_i >= u * C_last.
// 1) block sums; the LAST block to finish scans them (exclusive offsets + total)  2) search, one wave per sample
// 3) one block: sort / unique / inverse, then the gather of the unique rows (rows beyond n_unique zero-filled).
constexpr int SMP_BLOCK = 256, SMP_CHUNKS = 8;

__global__ __launch_bounds__(SMP_BLOCK) void smp_block_sums(const float* __restrict__ preds, long long pred_stride,
                                                            float min_occ, long long P, double* __restrict__ block_sums,
                                                            int n_blocks, double* __restrict__ total, unsigned* __restrict__ done,
                                                            long long ws_stride) {
    __shared__ double s[SMP_BLOCK];
    __shared__ double carry;
    __shared__ bool last;
    // cloud blockIdx.y: its own occupancies and its own slice of the scratch (ws_stride bytes apart)
    preds += (long long)blockIdx.y * P * pred_stride;
    block_sums = (double*)((char*)block_sums + blockIdx.y * ws_stride);
    total = (double*)((char*)total + blockIdx.y * ws_stride);
    done = (unsigned*)((char*)done + blockIdx.y * ws_stride);
    // a workgroup sums SMP_CHUNKS consecutive chunks of 256 occupancies (each chunk with the summation tree it always had) and takes ONE
    // ticket: with a workgroup per chunk the 391 tickets of a 100k-point cloud queued up on one address (40 of the launch's 50 us)
    float pv[SMP_CHUNKS];
#pragma unroll
    for (int c = 0; c < SMP_CHUNKS; ++c) {
        const long long i = ((long long)blockIdx.x * SMP_CHUNKS + c) * SMP_BLOCK + threadIdx.x;
        pv[c] = i < P ? preds[i * pred_stride] : 0.f;
    }
    double cs[SMP_CHUNKS];
#pragma unroll
    for (int c = 0; c < SMP_CHUNKS; ++c) {
        double v = pv[c] > min_occ ? (double)pv[c] : 0.0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        cs[c] = v;
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int c = 0; c < SMP_CHUNKS; ++c) s[c * 4 + (threadIdx.x >> 6)] = cs[c];
    }
    __syncthreads();
    if (threadIdx.x < SMP_CHUNKS) {
        const int k = blockIdx.x * SMP_CHUNKS + threadIdx.x;
        if (k < n_blocks) block_sums[k] = (s[threadIdx.x * 4] + s[threadIdx.x * 4 + 1]) + (s[threadIdx.x * 4 + 2] + s[threadIdx.x * 4 + 3]);
        __threadfence();                                              // the sums are visible before the ticket
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        last = atomicAdd(done, 1u) == (unsigned)(gridDim.x - 1);
        carry = 0.0;
    }
    __syncthreads();
    if (!last) return;
    __threadfence();
    // exclusive scan of the block sums in place (sequential chunks of 256: n_blocks is a few hundred)
    for (int base = 0; base < n_blocks; base += SMP_BLOCK) {
        const int k = base + threadIdx.x;
        const double x = k < n_blocks ? __builtin_nontemporal_load(block_sums + k) : 0.0;
        s[threadIdx.x] = x;
        __syncthreads();
        for (int o = 1; o < SMP_BLOCK; o <<= 1) {                     // Hillis-Steele inclusive scan
            const double t = threadIdx.x >= o ? s[threadIdx.x - o] : 0.0;
            __syncthreads();
            s[threadIdx.x] += t;
            __syncthreads();
        }
        if (k < n_blocks) block_sums[k] = carry + s[threadIdx.x] - x;  // exclusive
        __syncthreads();
        if (threadIdx.x == SMP_BLOCK - 1) carry += s[SMP_BLOCK - 1];
        __syncthreads();
    }
    if (threadIdx.x == 0) { *total = carry; *done = 0u; }              // the counter is left ready for the next call
}

// One wave per sample: binary search of the block on the exclusive offsets, then the 256 occupancies of that block (4 per
// lane) are scanned in the wave and the first position whose running sum reaches the target is taken.  (A thread per sample
// walking its block serially cost 42 us: 256 dependent loads.)  If rounding puts the target past the last kept point of the
// block, the search moves on to the next block (and ends on the last kept point overall).
__global__ __launch_bounds__(256) void smp_search(const float* __restrict__ preds, long long pred_stride, float min_occ, long long P,
                                                  const double* __restrict__ block_off, int n_blocks, const double* __restrict__ total,
                                                  const float* __restrict__ u, int n_sample, long long* __restrict__ picked,
                                                  long long ws_stride) {
    const int sidx = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (sidx >= n_sample) return;
    preds += (long long)blockIdx.y * P * pred_stride;
    u += (long long)blockIdx.y * n_sample;
    block_off = (const double*)((const char*)block_off + blockIdx.y * ws_stride);
    total = (const double*)((const char*)total + blockIdx.y * ws_stride);
    picked = (long long*)((char*)picked + blockIdx.y * ws_stride);
    const double target = (double)u[sidx] * (*total);
    int lo = 0, hi = n_blocks - 1;                 // last block whose exclusive offset < target (or 0)
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (block_off[mid] < target) lo = mid; else hi = mid - 1;
    }
    long long ans = -1, last_kept = -1;
    for (int blk = lo; blk < n_blocks && ans < 0; ++blk) {
        const long long i0 = (long long)blk * SMP_BLOCK + 4 * lane;
        double v[4], run = 0.0;
        bool kept[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float p = (i0 + e < P) ? preds[(i0 + e) * pred_stride] : 0.f;
            kept[e] = (i0 + e < P) && p > min_occ;
            run += kept[e] ? (double)p : 0.0;
            v[e] = run;                            // inclusive sums inside the lane's quad
        }
        double incl = run;                         // inclusive scan of the lane totals across the wave
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const double t = __shfl_up(incl, o, 64);
            if (lane >= o) incl += t;
        }
        const double base = block_off[blk] + (incl - run);
        int first = 4;                             // first kept element of the quad that reaches the target
        int lastk = -1;
#pragma unroll
        for (int e = 3; e >= 0; --e) {
            if (kept[e] && base + v[e] >= target) first = e;
            if (kept[e] && lastk < 0) lastk = e;
        }
        const unsigned long long hit = __ballot(first < 4), anyk = __ballot(lastk >= 0);
        if (hit) {
            const int src = __ffsll((long long)hit) - 1;
            ans = (long long)blk * SMP_BLOCK + 4 * src + __shfl(first, src, 64);
        }
        if (anyk) {
            const int src = 63 - __clzll((long long)anyk);
            last_kept = (long long)blk * SMP_BLOCK + 4 * src + __shfl(lastk, src, 64);
        }
    }
    if (ans < 0) {                                  // target beyond the total by rounding: the last kept point
        if (last_kept < 0) {
            for (int blk = lo - 1; blk >= 0 && last_kept < 0; --blk) {
                const long long i0 = (long long)blk * SMP_BLOCK + 4 * lane;
                int lastk = -1;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (i0 + e < P && preds[(i0 + e) * pred_stride] > min_occ) lastk = e;
                const unsigned long long anyk = __ballot(lastk >= 0);
                if (anyk) {
                    const int src = 63 - __clzll((long long)anyk);
                    last_kept = (long long)blk * SMP_BLOCK + 4 * src + __shfl(lastk, src, 64);
                }
            }
        }
        ans = last_kept;
    }
    if (lane == 0) picked[sidx] = ans;
}

// one block: bitonic sort of (picked, sample id), unique, inverse (scone_utils.py:1060-1061).  n_sample <= SMP_MAX.
constexpr int SMP_MAX = 4096;
__global__ __launch_bounds__(1024) void smp_unique(const long long* __restrict__ picked, int n_sample,
                                                   long long* __restrict__ uniq, long long* __restrict__ inverse,
                                                   int* __restrict__ n_unique, long long ws_stride,
                                                   const double* __restrict__ total, double* __restrict__ volume) {
    __shared__ long long key[SMP_MAX];
    __shared__ int rank[SMP_MAX];
    picked = (const long long*)((const char*)picked + blockIdx.x * ws_stride);     // one block per cloud
    if (volume && threadIdx.x == 0) volume[blockIdx.x] = *(const double*)((const char*)total + blockIdx.x * ws_stride);
    uniq += (long long)blockIdx.x * n_sample;
    inverse += (long long)blockIdx.x * n_sample;
    n_unique += blockIdx.x;
    int n2 = 1;
    while (n2 < n_sample) n2 <<= 1;
    for (int i = threadIdx.x; i < n2; i += 1024)
        key[i] = (i < n_sample && picked[i] >= 0) ? ((picked[i] << 13) | (long long)i)
                                                  : (0x7fffffffffffe000LL | (long long)(i & 8191));   // invalid / padding sort last
    __syncthreads();
    for (int k = 2; k <= n2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < n2; i += 1024) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const bool up = (i & k) == 0;
                    const long long a = key[i], b = key[ixj];
                    if ((a > b) == up) { key[i] = b; key[ixj] = a; }
                }
            }
            __syncthreads();
        }
    // flags -> inclusive scan (sequential per thread chunk + block scan)
    for (int i = threadIdx.x; i < n2; i += 1024)
        rank[i] = (i < n_sample && (key[i] >> 13) != (0x7fffffffffffe000LL >> 13) &&
                   (i == 0 || (key[i] >> 13) != (key[i - 1] >> 13))) ? 1 : 0;
    __syncthreads();
    for (int o = 1; o < n2; o <<= 1) {
        int t[SMP_MAX / 1024];
        int c = 0;
        for (int i = threadIdx.x; i < n2; i += 1024) t[c++] = i >= o ? rank[i - o] : 0;
        __syncthreads();
        c = 0;
        for (int i = threadIdx.x; i < n2; i += 1024) rank[i] += t[c++];
        __syncthreads();
    }
    const int nu = rank[n_sample - 1];
    for (int i = threadIdx.x; i < n_sample; i += 1024) {
        const long long id = key[i] >> 13;
        if (id == (0x7fffffffffffe000LL >> 13)) {                 // no point above min_occ: nothing sampled
            inverse[key[i] & 8191] = 0;
            continue;
        }
        const int r = rank[i] - 1;
        if (i == 0 || id != (key[i - 1] >> 13)) uniq[r] = id;
        inverse[key[i] & 8191] = r;
    }
    if (threadIdx.x == 0) *n_unique = nu;
    for (int r = nu + threadIdx.x; r < n_sample; r += 1024) uniq[r] = 0;      // padding rows
}

// res[r] = (X[uniq[r]], pred[uniq[r]]), res_h[r] = vh[uniq[r]] for r < n_unique, zeros beyond.  grid = ceil(n_sample / 16):
// 16 rows x 64 columns per block (inside the single sort block this gather was 128 serial passes of dependent loads: 0.2 ms)
__global__ __launch_bounds__(1024) void smp_gather(const long long* __restrict__ uniq, const int* __restrict__ n_unique, int n_sample,
                                                   const float* __restrict__ X, const float* __restrict__ preds, long long pred_stride,
                                                   const float* __restrict__ vh, float* __restrict__ res, float* __restrict__ res_h,
                                                   long long P, int shared_points) {
    const int r = blockIdx.x * 16 + (threadIdx.x >> 6), c = threadIdx.x & 63;
    if (r >= n_sample) return;
    const long long b = blockIdx.y;
    uniq += b * n_sample; n_unique += b; preds += b * P * pred_stride;
    if (!shared_points) X += b * P * 3;
    if (vh) { if (!shared_points) vh += b * P * 64; res_h += b * n_sample * 64; }
    res += b * n_sample * 4;
    if (r < *n_unique) {
        const long long i = uniq[r];
        if (vh) res_h[(long long)r * 64 + c] = vh[i * 64 + c];
        if (c < 3) res[r * 4 + c] = X[i * 3 + c];
        if (c == 3) res[r * 4 + 3] = preds[i * pred_stride];
    } else {
        if (vh) res_h[(long long)r * 64 + c] = 0.f;
        if (c < 4) res[r * 4 + c] = 0.f;
    }
}

// ---------------------------------------------------------------------------------------------------------
// K2: frustum + range mask.  Row-vector convention of the reference's camera transforms:
//   p_view = [x y z 1] * M_view (4x4 row-major),   p_ndc = ([x y z 1] * M_proj)[:3] / w
// mask = ndc_x in [min_x,max_x] & ndc_y in [min_y,max_y] & z_view > 0 [& |p - c| < range]   (macarons_utils.py:2420-2430)
__global__ void fov_kernel(const float* __restrict__ pts, long long P, const float* __restrict__ cam, int n_cam,
                           unsigned char* __restrict__ mask) {
    // cam record (40 floats): M_view[16], M_proj[16], ndc bounds {min_x,max_x,min_y,max_y}, center[3], range (<=0: none)
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= P * n_cam) return;
    const int c = (int)(gid / P);
    const long long p = gid - (long long)c * P;
    const float* K = cam + c * 40;
    const float x = pts[3 * p], y = pts[3 * p + 1], z = pts[3 * p + 2];
    const float zv = ((x * K[2] + y * K[6]) + z * K[10]) + K[14];
    const float* Mp = K + 16;
    const float px = ((x * Mp[0] + y * Mp[4]) + z * Mp[8]) + Mp[12];
    const float py = ((x * Mp[1] + y * Mp[5]) + z * Mp[9]) + Mp[13];
    const float pw = ((x * Mp[3] + y * Mp[7]) + z * Mp[11]) + Mp[15];
    const float nx = px / pw, ny = py / pw;
    bool m = nx >= K[32] && nx <= K[33] && ny >= K[34] && ny <= K[35] && zv > 0.f;
    if (K[39] > 0.f) {
        const float dx = x - K[36], dy = y - K[37], dz = z - K[38];
        m = m && sqrtf((dx * dx + dy * dy) + dz * dz) < K[39];
    }
    mask[gid] = m ? 1 : 0;
}


// ---------------------------------------------------------------------------------------------------------
// Signed distance of world points to the surface a depth map delimits (Camera.get_signed_distance_to_depth_maps,
// macarons_utils.py:2451-2500): z of the point in the camera's view space minus the depth map sampled at the point's projection
// with torch.nn.functional.grid_sample(mode='bilinear', padding_mode='border', align_corners=False); pixels outside `mask`
// count as `fill` (= 1.1 zfar, :2479).  cam = M_view[16] | M_full_projection[16] (row-vector convention).
__device__ __forceinline__ float depth_at(const float* __restrict__ depth, const unsigned char* __restrict__ mask, int W, int ix, int iy,
                                          float fill) {
    const long long i = (long long)iy * W + ix;
    return (mask && !mask[i]) ? fill : depth[i];
}
__device__ __forceinline__ float signed_distance(const float* __restrict__ cam, const float* __restrict__ depth,
                                                 const unsigned char* __restrict__ mask, int H, int W, float fill, float x, float y, float z) {
    const float zw = ((x * cam[3] + y * cam[7]) + z * cam[11]) + cam[15];
    const float zv = (((x * cam[2] + y * cam[6]) + z * cam[10]) + cam[14]) / zw;                     // get_points_zbuf (:2448)
    const float* Mp = cam + 16;
    const float pw = ((x * Mp[3] + y * Mp[7]) + z * Mp[11]) + Mp[15];
    const float px = (((x * Mp[0] + y * Mp[4]) + z * Mp[8]) + Mp[12]) / pw;
    const float py = (((x * Mp[1] + y * Mp[5]) + z * Mp[9]) + Mp[13]) / pw;
    const float factor = -(float)min(H, W);                                                           // :2484-2487
    const float gx = factor / (float)W * px, gy = factor / (float)H * py;
    // grid_sample: unnormalise (align_corners=False), clip to the border, bilinear
    float fx = ((gx + 1.f) * (float)W - 1.f) / 2.f, fy = ((gy + 1.f) * (float)H - 1.f) / 2.f;
    fx = fminf((float)(W - 1), fmaxf(fx, 0.f));
    fy = fminf((float)(H - 1), fmaxf(fy, 0.f));
    const float x0 = floorf(fx), y0 = floorf(fy);
    const int ix0 = (int)x0, iy0 = (int)y0, ix1 = ix0 + 1, iy1 = iy0 + 1;
    const float w_nw = (x0 + 1.f - fx) * (y0 + 1.f - fy), w_ne = (fx - x0) * (y0 + 1.f - fy);
    const float w_sw = (x0 + 1.f - fx) * (fy - y0), w_se = (fx - x0) * (fy - y0);
    float acc = depth_at(depth, mask, W, ix0, iy0, fill) * w_nw;
    if (ix1 < W) acc += depth_at(depth, mask, W, ix1, iy0, fill) * w_ne;
    if (iy1 < H) acc += depth_at(depth, mask, W, ix0, iy1, fill) * w_sw;
    if (ix1 < W && iy1 < H) acc += depth_at(depth, mask, W, ix1, iy1, fill) * w_se;
    return zv - acc;
}
__global__ void signed_distance_kernel(const float* __restrict__ pts, long long n, const float* __restrict__ cam,
                                       const float* __restrict__ depth, const unsigned char* __restrict__ mask, int H, int W, float fill,
                                       float* __restrict__ out) {
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    out[p] = signed_distance(cam, depth, mask, H, W, fill, pts[3 * p], pts[3 * p + 1], pts[3 * p + 2]);
}

// One pass over the scene's proxy points after a new depth map (testers/scene.py:402-418 = Camera.get_signed_distance_to_depth_maps
// + Scene.update_proxy_view_states + update_proxy_supervision_occ + update_proxy_out_of_field, macarons_utils.py:2817-2912), for
// the points whose fov_mask is set: signed distance d to the depth map; if d < distance_to_surface the bin of the direction to
// the camera is OR'ed into the point's view state; n_inside += 1, n_behind += (d >= -tol), supervision_occ = (n_behind / n_inside
// >= score_threshold); out_of_field = 0.  sgn (optional) receives d (untouched where the mask is clear).
__global__ void proxy_update_kernel(const float* __restrict__ pts, long long P, const unsigned char* __restrict__ fov_mask,
                                    const float* __restrict__ cam, const float* __restrict__ depth, const unsigned char* __restrict__ dmask,
                                    int H, int W, float fill, const float* __restrict__ X_cam, float distance_to_surface, float tol,
                                    float score_threshold, int n_elev, int n_azim, float* __restrict__ view_states,
                                    float* __restrict__ n_inside, float* __restrict__ n_behind, float* __restrict__ sup_occ,
                                    float* __restrict__ out_of_field, float* __restrict__ sgn) {
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P || !fov_mask[p]) return;
    const float x = pts[3 * p], y = pts[3 * p + 1], z = pts[3 * p + 2];
    const float d = signed_distance(cam, depth, dmask, H, W, fill, x, y, z);
    if (sgn) sgn[p] = d;
    if (d < distance_to_surface)
        view_states[p * (long long)(n_elev * n_azim) + view_state_bin(X_cam[0] - x, X_cam[1] - y, X_cam[2] - z, n_elev, n_azim)] = 1.0f;
    const float ni = n_inside[p] + 1.f, nbh = n_behind[p] + (d >= -tol ? 1.f : 0.f);
    n_inside[p] = ni;
    n_behind[p] = nbh;
    sup_occ[p] = (nbh / ni >= score_threshold) ? 1.f : 0.f;
    out_of_field[p] = 0.f;
}

// ---------------------------------------------------------------------------------------------------------
// filter_proxy_points (macarons/utility/scone_utils.py:1001-1027): keep the proxy points whose projection falls, in EVERY
// view camera, inside the screen-space bounding box of the projected surface cloud grown by filter_tol (strict compares).
// Same camera convention as K2: ndc = ([x y z 1] * M_proj)[:2] / w with M_proj row-major 4x4 per view.
// Pass 1: per view min/max of the projected cloud (one block per view, order-independent);  pass 2: the mask.
__device__ __forceinline__ void project_xy(const float* __restrict__ Mp, float x, float y, float z, float& nx, float& ny) {
    const float px = ((x * Mp[0] + y * Mp[4]) + z * Mp[8]) + Mp[12];
    const float py = ((x * Mp[1] + y * Mp[5]) + z * Mp[9]) + Mp[13];
    const float pw = ((x * Mp[3] + y * Mp[7]) + z * Mp[11]) + Mp[15];
    nx = px / pw;
    ny = py / pw;
}

__global__ __launch_bounds__(256) void proj_bounds_kernel(const float* __restrict__ pc, long long M, const float* __restrict__ proj,
                                                          float* __restrict__ bounds) {
    __shared__ float s[4][4];
    const float* Mp = proj + blockIdx.x * 16;
    float mnx = __builtin_inff(), mxx = -__builtin_inff(), mny = __builtin_inff(), mxy = -__builtin_inff();
    for (long long i = threadIdx.x; i < M; i += 256) {
        float nx, ny;
        project_xy(Mp, pc[3 * i], pc[3 * i + 1], pc[3 * i + 2], nx, ny);
        mnx = fminf(mnx, nx); mxx = fmaxf(mxx, nx);
        mny = fminf(mny, ny); mxy = fmaxf(mxy, ny);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mnx = fminf(mnx, __shfl_xor(mnx, o, 64)); mxx = fmaxf(mxx, __shfl_xor(mxx, o, 64));
        mny = fminf(mny, __shfl_xor(mny, o, 64)); mxy = fmaxf(mxy, __shfl_xor(mxy, o, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        float* r = s[threadIdx.x >> 6];
        r[0] = mnx; r[1] = mxx; r[2] = mny; r[3] = mxy;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float* b = bounds + blockIdx.x * 4;                  // {min_x, max_x, min_y, max_y}
        b[0] = fminf(fminf(s[0][0], s[1][0]), fminf(s[2][0], s[3][0]));
        b[1] = fmaxf(fmaxf(s[0][1], s[1][1]), fmaxf(s[2][1], s[3][1]));
        b[2] = fminf(fminf(s[0][2], s[1][2]), fminf(s[2][2], s[3][2]));
        b[3] = fmaxf(fmaxf(s[0][3], s[1][3]), fmaxf(s[2][3], s[3][3]));
    }
}

__global__ void filter_proxy_kernel(const float* __restrict__ X, long long P, const float* __restrict__ proj, int n_view,
                                    const float* __restrict__ bounds, float tol, unsigned char* __restrict__ mask) {
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const float x = X[3 * p], y = X[3 * p + 1], z = X[3 * p + 2];
    bool keep = true;
    for (int v = 0; v < n_view; ++v) {
        float nx, ny;
        project_xy(proj + v * 16, x, y, z, nx, ny);
        const float* b = bounds + v * 4;
        keep = keep && (nx < b[1] + tol) && (nx > b[0] - tol) && (ny < b[3] + tol) && (ny > b[2] - tol);
    }
    mask[p] = keep ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------------------
// Column gather of move_view_state_to_view_space (scone_utils.py:928): out[r, v] = in[r, idx[v]], v < V (V = 98 view bins).
__global__ void gather_columns_kernel(const float* __restrict__ in, const int* __restrict__ idx, float* __restrict__ out, long long rows,
                                      int V) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= rows * V) return;
    const long long r = gid / V;
    const int v = (int)(gid - r * V);
    out[gid] = in[r * V + idx[v]];
}

// ---------------------------------------------------------------------------------------------------------
// a3: gains of every ordered n-tuple of cameras: mean_n max_t vis[b, c_t, n]   (SconeVis.py:289-301)
// grid = (C^n tuples, B); one block per tuple, tree reduce in fixed order.
__global__ __launch_bounds__(256) void multi_gain_kernel(const float* __restrict__ vis, float* __restrict__ out, int C, int N,
                                                         int n_cam) {
    __shared__ double s[4];
    const int tup = blockIdx.x, b = blockIdx.y;
    int c[3];
    int t = tup;
    for (int k = n_cam - 1; k >= 0; --k) { c[k] = t % C; t /= C; }       // cartesian_prod order: last index fastest
    const float* v = vis + (size_t)b * C * N;
    double acc = 0.0;
    for (int n = threadIdx.x; n < N; n += 256) {
        float m = v[(size_t)c[0] * N + n];
        for (int k = 1; k < n_cam; ++k) m = fmaxf(m, v[(size_t)c[k] * N + n]);
        acc += (double)m;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        long long tuples = 1;
        for (int k = 0; k < n_cam; ++k) tuples *= C;
        out[(size_t)b * tuples + tup] = (float)(((s[0] + s[1]) + (s[2] + s[3])) / (double)N);
    }
}

// ---------------------------------------------------------------------------------------------------------
// MACARONS per-camera scoring helpers (macarons_utils.py:1580-1738):
// masked occupancy: occ_out[c][p] = in_fov(c, p) ? occ[p] : 0   (feeds the sampler: fov mask AND occ > min_occ)
__global__ void fov_mask_occ_kernel(const unsigned char* __restrict__ mask, const float* __restrict__ occ, long long occ_stride,
                                    float* __restrict__ occ_out, long long P, int n_cam) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= P * n_cam) return;
    const long long p = gid % P;
    occ_out[gid] = mask[gid] ? occ[p * occ_stride] : 0.f;
}

// in place: pts[i, :3] = ((pts[i, :3] 1) * M_view - center) * inv_diag     (world -> normalised prediction-view space,
// macarons_utils.py:1641-1660); row stride pts_dim
// batched form: point i belongs to cloud i / pts_per_cloud (or cloud_of[i]) with its own matrix (16 floats), centre (3) and scale
__global__ void transform_points_kernel(float* __restrict__ pts, int pts_dim, long long n, const float* __restrict__ M,
                                        const float* __restrict__ center, float inv_diag, long long pts_per_cloud,
                                        const float* __restrict__ inv_diag_c, const int* __restrict__ cloud_of) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const long long c = cloud_of ? (long long)cloud_of[i] : i / pts_per_cloud;
    M += c * 16; center += c * 3;
    if (inv_diag_c) inv_diag = inv_diag_c[c];
    float* p = pts + i * pts_dim;
    const float x = p[0], y = p[1], z = p[2];
    p[0] = ((((x * M[0] + y * M[4]) + z * M[8]) + M[12]) - center[0]) * inv_diag;
    p[1] = ((((x * M[1] + y * M[5]) + z * M[9]) + M[13]) - center[1]) * inv_diag;
    p[2] = ((((x * M[2] + y * M[6]) + z * M[10]) + M[14]) - center[2]) * inv_diag;
}

// gains[b] = mean_n( vis[b,n] * factor(|pts_world[b,n] - cam_world[b]|) ) * volume[b]; vis is scaled in place (the final
// product of macarons_utils.py:1699-1704).  factor: mode 0 = min(1, (th / d)^2) (get_distance_factor_threshold :1768-1776;
// get_distance_factor :1741-1765 is the same function with th = focal * epsilon / pixel), mode 1 = 1 / (1 + (d / th)^2)
// (get_distance_factor_smooth :1779-1788).  One block per b.
__global__ __launch_bounds__(256) void macarons_gain_kernel(float* __restrict__ vis, const float* __restrict__ pts_world,
                                                            int pts_dim, const float* __restrict__ cam_world,
                                                            const float* __restrict__ volume, float distance_th, int mode, int N,
                                                            float* __restrict__ gains) {
    __shared__ double s[4];
    const int b = blockIdx.x;
    const float cx = cam_world[3 * b], cy = cam_world[3 * b + 1], cz = cam_world[3 * b + 2];
    double acc = 0.0;
    for (int n = threadIdx.x; n < N; n += 256) {
        const float* p = pts_world + ((size_t)b * N + n) * pts_dim;
        const float dx = p[0] - cx, dy = p[1] - cy, dz = p[2] - cz;
        const float d = sqrtf((dx * dx + dy * dy) + dz * dz);
        float f = 1.f;
        if (mode == 1) {
            const float q = d / distance_th;
            f = 1.f / (1.f + q * q);
        } else if (d > distance_th) {
            f = (distance_th * distance_th) / (d * d);
        }
        const float v = vis[(size_t)b * N + n] * f;
        vis[(size_t)b * N + n] = v;
        acc += (double)v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) gains[b] = (float)(((s[0] + s[1]) + (s[2] + s[3])) / (double)N) * volume[b];
}

// K12: depth map -> world points (Camera.project_depth_in_3D macarons_utils.py:2339-2360 / utils.project_depth_back_to_3D
// utils.py:1458-1487 with pytorch3d FoVPerspectiveCameras.unproject_points(scaled_depth_input=False)):
//   ndc_x = W/m - 2 j/(m-1), ndc_y = H/m - 2 i/(m-1), m = min(W,H);  sdepth = (k22 * d + k32) / d;
//   world = ([ndc_x ndc_y sdepth 1] * Minv)[:3] / w          (Minv = inverse full projection, row-vector)
__global__ void unproject_depth_kernel(const float* __restrict__ depth, int H, int W, const float* __restrict__ cam,
                                       float* __restrict__ out, long long n_cam) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long hw = (long long)H * W;
    if (gid >= n_cam * hw) return;
    const long long c = gid / hw, pix = gid - c * hw;
    const int i = (int)(pix / W), j = (int)(pix - (long long)i * W);
    const float m = (float)min(H, W);
    const float nx = (float)W / m - ((float)j / (m - 1.f)) * 2.f;
    const float ny = (float)H / m - ((float)i / (m - 1.f)) * 2.f;
    const float* K = cam + c * 18;            // Minv[16], k22, k32
    const float d = depth[gid];
    const float sd = (K[16] * d + K[17]) / d;
    const float x = ((nx * K[0] + ny * K[4]) + sd * K[8]) + K[12];
    const float y = ((nx * K[1] + ny * K[5]) + sd * K[9]) + K[13];
    const float z = ((nx * K[2] + ny * K[6]) + sd * K[10]) + K[14];
    const float w = ((nx * K[3] + ny * K[7]) + sd * K[11]) + K[15];
    out[3 * gid + 0] = x / w;
    out[3 * gid + 1] = y / w;
    out[3 * gid + 2] = z / w;
}

}  // namespace mcr

using namespace mcr;


// ---- arg-max exchange records (multi-GPU camera sharding, testers/shapenet.py:172 = torch.max over cameras) ------------
// record[b] = (max_c gains[b,c], idx_offset + first arg-max) as two fp32 (camera indices < 2^24 are exact): one 8-byte record
// per cloud is what the ranks all-gather.  One wave per cloud; ties -> lowest index like torch.max.
// torch.max ordering: NaN beats every number (the first NaN wins), otherwise the larger value, ties to the lower index.
__device__ __forceinline__ bool best_before(float v, float i, float bv, float bi) {
    const bool vn = v != v, bn = bv != bv;
    if (vn != bn) return vn;
    if (!vn && v != bv) return v > bv;
    return i < bi;
}

__global__ __launch_bounds__(64) void best_record_kernel(const float* __restrict__ gains, int C, long long idx_offset,
                                                         float* __restrict__ rec) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const float* g = gains + (size_t)b * C;
    // start from the lane's first column, so that a row of -inf (or NaN) still yields a valid index like torch.max does
    float bv = lane < C ? g[lane] : -__builtin_inff();
    float bi = lane < C ? (float)lane : 3.0e38f;
    for (int c = lane + 64; c < C; c += 64) {
        const float v = g[c];
        if (best_before(v, (float)c, bv, bi)) { bv = v; bi = (float)c; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o, 64), oi = __shfl_xor(bi, o, 64);
        if (best_before(ov, oi, bv, bi)) { bv = ov; bi = oi; }
    }
    if (lane == 0) {
        rec[2 * b] = bv;
        rec[2 * b + 1] = (float)idx_offset + bi;               // camera indices < 2^24 are exact in fp32
    }
}

// The end of a single-rank NBV decision in one launch (testers/shapenet.py:172 + the empty-sample rule of nbv.py): per cloud b, a
// cloud nothing was sampled from (n_unique[b] < 1) gets NaN gains, a NaN maximum and index -1; otherwise torch.max over its cameras.
// record = (range flag, idx[0..B), max[0..B)) as doubles: the ONE buffer the host reads back at the end of the decision.
__global__ __launch_bounds__(64) void nbv_decide_kernel(float* __restrict__ gains, int C, const int* __restrict__ n_unique,
                                                        const int* __restrict__ range_flag, float* __restrict__ max_gain,
                                                        long long* __restrict__ nbv_idx, double* __restrict__ record, int B) {
    const int b = blockIdx.x, lane = threadIdx.x;
    float* g = gains + (size_t)b * C;
    const bool empty = n_unique && n_unique[b] < 1;
    float bv, bi;
    if (empty) {
        for (int c = lane; c < C; c += 64) g[c] = __builtin_nanf("");
        bv = __builtin_nanf(""); bi = -1.f;
    } else {
        bv = lane < C ? g[lane] : -__builtin_inff();
        bi = lane < C ? (float)lane : 3.0e38f;
        for (int c = lane + 64; c < C; c += 64) {
            const float v = g[c];
            if (best_before(v, (float)c, bv, bi)) { bv = v; bi = (float)c; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o, 64), oi = __shfl_xor(bi, o, 64);
            if (best_before(ov, oi, bv, bi)) { bv = ov; bi = oi; }
        }
    }
    if (lane == 0) {
        max_gain[b] = bv;
        nbv_idx[b] = (long long)bi;
        record[1 + b] = (double)bi;
        record[1 + B + b] = (double)bv;
        if (b == 0) record[0] = range_flag ? (double)*range_flag : 0.0;
    }
}

// recs [world, B, 2] -> (vals[b], idx[b]) of the global arg-max; ties -> lowest camera index.  A rank with an empty camera shard
// contributes (-inf, 3e38): it never wins against a rank that scored anything.
__global__ __launch_bounds__(64) void best_merge_kernel(const float* __restrict__ recs, int world, int B, float* __restrict__ vals,
                                                        long long* __restrict__ idx) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    float bv = recs[2 * b], bi = recs[2 * b + 1];
    for (int r = 1; r < world; ++r) {
        const float v = recs[((size_t)r * B + b) * 2], i = recs[((size_t)r * B + b) * 2 + 1];
        if (best_before(v, i, bv, bi)) { bv = v; bi = i; }
    }
    vals[b] = bv;
    idx[b] = (long long)bi;
}

extern "C" {

static int view_state_impl(const char* who, const float* pts, int pts_dim, const float* X_view, float* view_state, int64_t n_clouds,
                           int64_t pts_per_cloud, int n_view, int n_elev, int n_azim, const int64_t* rows, int accumulate,
                           void* stream) {
    MCR_REQUIRE(pts && X_view && view_state, "%s: null pointer", who);
    MCR_REQUIRE(pts_dim >= 3 && n_clouds > 0 && pts_per_cloud > 0 && n_view > 0 && n_elev > 0 && n_azim > 0, "%s: bad sizes", who);
    MCR_REQUIRE(!rows || accumulate, "%s: a row index needs accumulate != 0 (the table is the caller's state)", who);
    hipStream_t s = (hipStream_t)stream;
    const long long n_points = (long long)n_clouds * pts_per_cloud;
    if (!accumulate) launch_zero(s, view_state, (size_t)n_points * n_elev * n_azim * sizeof(float));
    const long long total = n_points * n_view;
    hipLaunchKernelGGL(view_state_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, s, pts, pts_dim, X_view, view_state,
                       n_points, (long long)pts_per_cloud, n_view, n_elev, n_azim, (const long long*)rows);
    MCR_LAUNCH_CHECK("view_state_kernel");
    return 0;
}

int mcr_view_state(const float* pts, int pts_dim, const float* X_view, float* view_state, int64_t n_points, int n_view,
                   int n_elev, int n_azim, void* stream) {
    return view_state_impl("mcr_view_state", pts, pts_dim, X_view, view_state, 1, n_points, n_view, n_elev, n_azim, nullptr, 0, stream);
}

int mcr_view_state_batched(const float* pts, int pts_dim, const float* X_view, float* view_state, int64_t n_clouds,
                           int64_t pts_per_cloud, int n_view, int n_elev, int n_azim, const int64_t* rows, int accumulate,
                           void* stream) {
    return view_state_impl("mcr_view_state_batched", pts, pts_dim, X_view, view_state, n_clouds, pts_per_cloud, n_view, n_elev, n_azim,
                           rows, accumulate, stream);
}

// per-cloud scratch slice: [block sums (nb) | total (2)] doubles, picked[n_sample] int64, ticket (+ padding)
static size_t smp_slice_bytes(int64_t P, int n_sample) {
    return (size_t)(cdiv(P, SMP_BLOCK) + 2) * sizeof(double) + (size_t)n_sample * sizeof(long long) + 64;
}

size_t mcr_sample_proxy_workspace_bytes(int64_t P, int n_sample) { return smp_slice_bytes(P, n_sample) + 448; }
size_t mcr_sample_proxy_batched_workspace_bytes(int64_t B, int64_t P, int n_sample) { return (size_t)B * smp_slice_bytes(P, n_sample) + 448; }

static int sample_proxy_impl(const float* X, const float* preds, int64_t pred_stride, const float* view_harmonics, int64_t B, int64_t P,
                             float min_occ, const float* u, int n_sample, float* res, float* res_harmonics, int64_t* uniq,
                             int64_t* inverse, int* n_unique, double* volume, void* workspace, size_t workspace_bytes, void* stream,
                             int shared_points) {
    MCR_REQUIRE(X && preds && u && res && uniq && inverse && n_unique && (res_harmonics || !view_harmonics),
                "mcr_sample_proxy: null pointer");
    MCR_REQUIRE(B > 0 && B <= 65535 && P > 0 && n_sample > 0 && n_sample <= SMP_MAX, "mcr_sample_proxy: need 0 < n_sample <= %d, 0 < B <= 65535",
                SMP_MAX);
    MCR_REQUIRE(P < (1ll << 49), "mcr_sample_proxy: P too large");
    MCR_REQUIRE(workspace && workspace_bytes >= mcr_sample_proxy_batched_workspace_bytes(B, P, n_sample), "mcr_sample_proxy: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const int nb = (int)cdiv(P, SMP_BLOCK);
    const long long stride = (long long)smp_slice_bytes(P, n_sample);
    double* block_sums = (double*)workspace;
    double* total = block_sums + nb;
    long long* picked = (long long*)(total + 2);
    unsigned* ticket = (unsigned*)(picked + n_sample);          // "last block scans" counter of smp_block_sums, in the caller's scratch
    if (B == 1) launch_zero(s, ticket, sizeof(unsigned));
    else launch_zero(s, workspace, (size_t)B * stride);
    hipLaunchKernelGGL(smp_block_sums, dim3((unsigned)cdiv(nb, SMP_CHUNKS), (unsigned)B), dim3(SMP_BLOCK), 0, s, preds, (long long)pred_stride, min_occ, (long long)P,
                       block_sums, nb, total, ticket, stride);
    hipLaunchKernelGGL(smp_search, dim3((unsigned)cdiv(n_sample, 4), (unsigned)B), dim3(256), 0, s, preds, (long long)pred_stride, min_occ,
                       (long long)P, block_sums, nb, total, u, n_sample, picked, stride);
    hipLaunchKernelGGL(smp_unique, dim3((unsigned)B), dim3(1024), 0, s, picked, n_sample, (long long*)uniq, (long long*)inverse, n_unique,
                       stride, total, volume);
    hipLaunchKernelGGL(smp_gather, dim3((unsigned)cdiv(n_sample, 16), (unsigned)B), dim3(1024), 0, s, (const long long*)uniq, n_unique,
                       n_sample, X, preds, (long long)pred_stride, view_harmonics, res, res_harmonics, (long long)P, shared_points);
    MCR_LAUNCH_CHECK("mcr_sample_proxy");
    return 0;
}

int mcr_sample_proxy_batched(const float* X, const float* preds, int64_t pred_stride, const float* view_harmonics, int64_t B, int64_t P,
                             float min_occ, const float* u, int n_sample, float* res, float* res_harmonics, int64_t* uniq,
                             int64_t* inverse, int* n_unique, double* volume, void* workspace, size_t workspace_bytes, void* stream) {
    return sample_proxy_impl(X, preds, pred_stride, view_harmonics, B, P, min_occ, u, n_sample, res, res_harmonics, uniq, inverse,
                             n_unique, volume, workspace, workspace_bytes, stream, 0);
}

int mcr_sample_proxy_shared(const float* X, const float* preds, int64_t pred_stride, const float* view_harmonics, int64_t B, int64_t P,
                            float min_occ, const float* u, int n_sample, float* res, float* res_harmonics, int64_t* uniq,
                            int64_t* inverse, int* n_unique, double* volume, void* workspace, size_t workspace_bytes, void* stream) {
    return sample_proxy_impl(X, preds, pred_stride, view_harmonics, B, P, min_occ, u, n_sample, res, res_harmonics, uniq, inverse,
                             n_unique, volume, workspace, workspace_bytes, stream, 1);
}

int mcr_sample_proxy(const float* X, const float* preds, int64_t pred_stride, const float* view_harmonics, int64_t P,
                     float min_occ, const float* u, int n_sample, float* res, float* res_harmonics, int64_t* uniq,
                     int64_t* inverse, int* n_unique, double* volume, void* workspace, size_t workspace_bytes, void* stream) {
    return mcr_sample_proxy_batched(X, preds, pred_stride, view_harmonics, 1, P, min_occ, u, n_sample, res, res_harmonics, uniq, inverse,
                                    n_unique, volume, workspace, workspace_bytes, stream);
}

int mcr_points_in_fov(const float* pts, int64_t P, const float* cameras, int n_cam, unsigned char* mask, void* stream) {
    MCR_REQUIRE(pts && cameras && mask && P > 0 && n_cam > 0, "mcr_points_in_fov: bad arguments");
    hipLaunchKernelGGL(fov_kernel, dim3((unsigned)cdiv(P * n_cam, 256)), dim3(256), 0, (hipStream_t)stream, pts, (long long)P,
                       cameras, n_cam, mask);
    MCR_LAUNCH_CHECK("fov_kernel");
    return 0;
}

int mcr_signed_distance_to_depth(const float* pts, int64_t n, const float* camera, const float* depth, const unsigned char* mask,
                                 int H, int W, float fill, float* sgn, void* stream) {
    MCR_REQUIRE(pts && camera && depth && sgn, "mcr_signed_distance_to_depth: null pointer");
    MCR_REQUIRE(n > 0 && H > 0 && W > 0, "mcr_signed_distance_to_depth: empty problem");
    hipLaunchKernelGGL(signed_distance_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, pts, (long long)n, camera,
                       depth, mask, H, W, fill, sgn);
    MCR_LAUNCH_CHECK("signed_distance_kernel");
    return 0;
}

int mcr_proxy_scene_update(const float* proxy_points, int64_t P, const unsigned char* fov_mask, const float* camera, const float* depth,
                           const unsigned char* depth_mask, int H, int W, float fill, const float* X_cam, float distance_to_surface,
                           float tol, float score_threshold, int n_elev, int n_azim, float* view_states, float* n_inside,
                           float* n_behind, float* supervision_occ, float* out_of_field, float* sgn, void* stream) {
    MCR_REQUIRE(proxy_points && fov_mask && camera && depth && X_cam && view_states && n_inside && n_behind && supervision_occ &&
                out_of_field, "mcr_proxy_scene_update: null pointer");
    MCR_REQUIRE(P > 0 && H > 0 && W > 0 && n_elev > 0 && n_azim > 0, "mcr_proxy_scene_update: bad sizes");
    hipLaunchKernelGGL(proxy_update_kernel, dim3((unsigned)cdiv(P, 256)), dim3(256), 0, (hipStream_t)stream, proxy_points, (long long)P,
                       fov_mask, camera, depth, depth_mask, H, W, fill, X_cam, distance_to_surface, tol,
                       score_threshold, n_elev, n_azim, view_states, n_inside, n_behind, supervision_occ, out_of_field, sgn);
    MCR_LAUNCH_CHECK("proxy_update_kernel");
    return 0;
}

int mcr_filter_proxy_points(const float* X, int64_t P, const float* pc, int64_t M, const float* proj, int n_view, float filter_tol,
                            float* bounds, unsigned char* mask, void* stream) {
    MCR_REQUIRE(X && pc && proj && bounds && mask && P > 0 && M > 0 && n_view > 0, "mcr_filter_proxy_points: bad arguments");
    hipLaunchKernelGGL(proj_bounds_kernel, dim3((unsigned)n_view), dim3(256), 0, (hipStream_t)stream, pc, (long long)M, proj, bounds);
    MCR_LAUNCH_CHECK("proj_bounds_kernel");
    hipLaunchKernelGGL(filter_proxy_kernel, dim3((unsigned)cdiv(P, 256)), dim3(256), 0, (hipStream_t)stream, X, (long long)P, proj,
                       n_view, bounds, filter_tol, mask);
    MCR_LAUNCH_CHECK("filter_proxy_kernel");
    return 0;
}

int mcr_gather_columns(const float* in, const int* idx, float* out, int64_t rows, int V, void* stream) {
    MCR_REQUIRE(in && idx && out && rows > 0 && V > 0, "mcr_gather_columns: bad arguments");
    hipLaunchKernelGGL(gather_columns_kernel, dim3((unsigned)cdiv(rows * V, 256)), dim3(256), 0, (hipStream_t)stream, in, idx, out,
                       (long long)rows, V);
    MCR_LAUNCH_CHECK("gather_columns_kernel");
    return 0;
}

int mcr_fov_mask_occ(const unsigned char* mask, const float* occ, int64_t occ_stride, float* occ_out, int64_t P, int n_cam,
                     void* stream) {
    MCR_REQUIRE(mask && occ && occ_out && P > 0 && n_cam > 0, "mcr_fov_mask_occ: bad arguments");
    hipLaunchKernelGGL(fov_mask_occ_kernel, dim3((unsigned)cdiv(P * n_cam, 256)), dim3(256), 0, (hipStream_t)stream, mask, occ,
                       (long long)occ_stride, occ_out, (long long)P, n_cam);
    MCR_LAUNCH_CHECK("fov_mask_occ_kernel");
    return 0;
}

int mcr_transform_points(float* pts, int pts_dim, int64_t n, const float* M_view, const float* center, float inv_diag,
                         void* stream) {
    MCR_REQUIRE(pts && M_view && center && pts_dim >= 3 && n > 0, "mcr_transform_points: bad arguments");
    hipLaunchKernelGGL(transform_points_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, pts, pts_dim,
                       (long long)n, M_view, center, inv_diag, (long long)n, (const float*)nullptr, (const int*)nullptr);
    MCR_LAUNCH_CHECK("transform_points_kernel");
    return 0;
}

int mcr_transform_points_batched(float* pts, int pts_dim, int64_t n_clouds, int64_t pts_per_cloud, const float* M_view,
                                 const float* center, const float* inv_diag, const int* cloud_of, int64_t n_points, void* stream) {
    MCR_REQUIRE(pts && M_view && center && inv_diag && pts_dim >= 3 && n_clouds > 0, "mcr_transform_points_batched: bad arguments");
    const long long n = cloud_of ? (long long)n_points : (long long)n_clouds * pts_per_cloud;
    MCR_REQUIRE(n > 0 && (cloud_of || pts_per_cloud > 0), "mcr_transform_points_batched: empty problem");
    hipLaunchKernelGGL(transform_points_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, pts, pts_dim, n, M_view,
                       center, 0.f, (long long)(cloud_of ? 1 : pts_per_cloud), inv_diag, cloud_of);
    MCR_LAUNCH_CHECK("transform_points_kernel");
    return 0;
}

int mcr_macarons_gain(float* vis, const float* pts_world, int pts_dim, const float* cam_world, const float* volume,
                      float distance_th, int factor_mode, int64_t B, int64_t N, float* gains, void* stream) {
    MCR_REQUIRE(vis && pts_world && cam_world && volume && gains && B > 0 && N > 0 && pts_dim >= 3, "mcr_macarons_gain: bad arguments");
    MCR_REQUIRE(factor_mode == 0 || factor_mode == 1, "mcr_macarons_gain: factor_mode must be 0 (threshold) or 1 (smooth)");
    MCR_REQUIRE(distance_th > 0.f, "mcr_macarons_gain: distance_th must be positive");
    hipLaunchKernelGGL(macarons_gain_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, vis, pts_world, pts_dim, cam_world,
                       volume, distance_th, factor_mode, (int)N, gains);
    MCR_LAUNCH_CHECK("macarons_gain_kernel");
    return 0;
}

int mcr_unproject_depth(const float* depth, int H, int W, const float* cameras, int64_t n_cam, float* world, void* stream) {
    MCR_REQUIRE(depth && cameras && world && H > 1 && W > 1 && n_cam > 0, "mcr_unproject_depth: bad arguments");
    hipLaunchKernelGGL(unproject_depth_kernel, dim3((unsigned)cdiv(n_cam * H * W, 256)), dim3(256), 0, (hipStream_t)stream, depth, H,
                       W, cameras, world, (long long)n_cam);
    MCR_LAUNCH_CHECK("unproject_depth_kernel");
    return 0;
}

int mcr_coverage_gain_multiple(const float* vis, float* gains, int64_t B, int64_t C, int64_t N, int n_cam, void* stream) {
    MCR_REQUIRE(vis && gains && B > 0 && C > 0 && N > 0, "mcr_coverage_gain_multiple: bad arguments");
    MCR_REQUIRE(n_cam == 2 || n_cam == 3, "n_cam is too large.");
    long long tuples = 1;
    for (int k = 0; k < n_cam; ++k) tuples *= C;
    MCR_REQUIRE(tuples < (1ll << 31) && B <= 65535, "mcr_coverage_gain_multiple: too many tuples");
    hipLaunchKernelGGL(multi_gain_kernel, dim3((unsigned)tuples, (unsigned)B), dim3(256), 0, (hipStream_t)stream, vis, gains, (int)C,
                       (int)N, n_cam);
    MCR_LAUNCH_CHECK("multi_gain_kernel");
    return 0;
}

int mcr_best_record(const float* gains, int64_t B, int64_t C, int64_t idx_offset, float* records, void* stream) {
    MCR_REQUIRE(gains && records && B > 0 && C > 0, "mcr_best_record: bad arguments");
    MCR_REQUIRE(idx_offset >= 0 && idx_offset + C <= (1ll << 24), "mcr_best_record: camera indices must stay below 2^24");
    hipLaunchKernelGGL(best_record_kernel, dim3((unsigned)B), dim3(64), 0, (hipStream_t)stream, gains, (int)C, (long long)idx_offset,
                       records);
    MCR_LAUNCH_CHECK("best_record_kernel");
    return 0;
}

int mcr_nbv_decide(float* gains, int64_t B, int64_t C, const int* n_unique, const int* range_flag, float* max_gain, int64_t* nbv_idx,
                   double* record, void* stream) {
    MCR_REQUIRE(gains && max_gain && nbv_idx && record && B > 0 && C > 0, "mcr_nbv_decide: bad arguments");
    MCR_REQUIRE(C <= (1ll << 24) && B <= 65535, "mcr_nbv_decide: camera indices must stay below 2^24");
    hipLaunchKernelGGL(nbv_decide_kernel, dim3((unsigned)B), dim3(64), 0, (hipStream_t)stream, gains, (int)C, n_unique, range_flag, max_gain,
                       reinterpret_cast<long long*>(nbv_idx), record, (int)B);
    MCR_LAUNCH_CHECK("nbv_decide_kernel");
    return 0;
}

int mcr_best_merge(const float* records, int world, int64_t B, float* vals, int64_t* idx, void* stream) {
    MCR_REQUIRE(records && vals && idx && world > 0 && B > 0, "mcr_best_merge: bad arguments");
    hipLaunchKernelGGL(best_merge_kernel, dim3((unsigned)cdiv(B, 64)), dim3(64), 0, (hipStream_t)stream, records, world, (int)B, vals,
                       reinterpret_cast<long long*>(idx));
    MCR_LAUNCH_CHECK("best_merge_kernel");
    return 0;
}

}  // extern "C"
