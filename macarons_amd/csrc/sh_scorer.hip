// K9 — spherical-harmonic coverage-gain scorer for gfx950 (MI355X).
//
// Replaces the op sequence of the reference's
//   macarons/networks/SconeVis.py:210-252  compute_coverage_gain   -> gains [B,C]
//   macarons/networks/SconeVis.py:164-208  compute_visibilities    -> vis   [B,C,N]
//   (== macarons/networks/Macarons.py:138-178 compute_visibility_gains)
// which materialise [B*C*N,64] SH tensors three times.  Here: one lane owns one point (its 64 SH
// coefficients live in VGPRs, turned once into the monomial coefficients of 15 polynomials in cos(polar)), cameras
// come from wave-uniform scalar loads, the dot with the 64 real SH of the ray direction is evaluated trig-free by
// Horner steps (94 VALU ops per (point,camera) pair, 106 with activation and reduction), sigmoid/relu applied, and the per-camera sum over points
// is a wave64 DPP reduction -> per-wave-tile partials -> a deterministic second-pass reduce (bit-stable
// run to run).  Bound: fp32 VALU (370 algorithmic flop / pair; N*268 B of HBM per cloud).
//
// Conventions (reference CustomGeometry.py:27-45, spherical_harmonics.py:67-140): polar axis +Y,
// azimuth from +Z toward +X; channel k = l*l + l + m; Condon-Shortley phase included.
#include "common.h"
#include "sh_consts.inc"
#include <algorithm>

namespace mcr {

constexpr int SC_BLOCK = 256;       // 4 waves; one point per lane

// Workgroups are dealt to the 8 XCDs round-robin (blockIdx % 8) and every XCD has its own L2.  Consecutive wave ranges
// share a wave-tile's coefficients at their boundary, so the ranges are handed out in XCD-major order: the blocks of one
// XCD own one contiguous stretch of the unit space and a tile cut by a block boundary is fetched into one L2 only
// (measured HBM reads per launch: 68.6 MB with the plain order, 27.6 MB with this one; algorithmic 26.8 MB).
__device__ __forceinline__ int xcd_major_block(int b, int nb) {
    const int x = b & 7, q = nb >> 3, r = nb & 7;            // XCD x holds q + (x < r) blocks
    return x * q + (x < r ? x : r) + (b >> 3);
}

__device__ __forceinline__ constexpr int shk(int l, int m) { return l * l + l + m; }

// z = sum_k Y_k(d) h_k, trig-free and in monomial form (algebra and constants: gen_sh_consts.py):
//   n = d / |d|  (one v_rsq);  x = cos(polar) = n_y;   sin(polar)^m {cos,sin}(m azim) = {Re,Im} (n_z + i n_x)^m
// so sin(polar), the azimuth normalisation 1/rho and the rho = 0 special case never appear (a ray along +-Y
// simply has n_x = n_z = 0 and every m != 0 term vanishes; the reference's acos path is ill-conditioned there).
// P_l^m / sin^m is a polynomial of degree l-m in x, so for one point the sum over l of each order m collapses into
// ONE polynomial per (m, cos|sin):  U_m(x) = sum_k a[m+k,+m] x^k,  V_m(x) = sum_k a[m+k,-m] x^k, whose coefficients
// a (64 per point, same storage as the SH coefficients) are produced once per point by load_mono_coeffs.  Then
//   z = U_0(x) + sum_{m>=1} ( Re w^m U_m(x) + Im w^m V_m(x) ),   w = n_z + i n_x
// = 49 Horner FMAs + 24 ops for the powers + 14 to combine + 7 to normalise: 94 VALU ops per (point, camera) pair --
// the floor for 64 per-point coefficients (the rescaled-recurrence form this replaces needed 130).
// Measured on MI355X (tools/ubench): a dependent v_fma_f32 chain issues every ~8.8 cycles per wave; the 15 Horner
// chains are independent; v_pk_fma_f32 is half rate and buys nothing.
__device__ __forceinline__ float sh_dot(float dx, float dy, float dz, const float (&a)[64]) {
    const float r2 = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
    const float ir = __builtin_amdgcn_rsqf(r2);
    const float nx = dx * ir, ct = dy * ir, nz = dz * ir;

    float z = a[shk(7, 0)];
#pragma unroll
    for (int l = 6; l >= 0; --l) z = fmaf(ct, z, a[shk(l, 0)]);
    float cm = nz, sm = nx;                     // (n_z + i n_x)^m
#pragma unroll
    for (int m = 1; m < 8; ++m) {
        float U = a[shk(7, m)], V = a[shk(7, -m)];
#pragma unroll
        for (int l = 6; l >= m; --l) {
            U = fmaf(ct, U, a[shk(l, m)]);
            V = fmaf(ct, V, a[shk(l, -m)]);
        }
        z = fmaf(cm, U, z);
        z = fmaf(sm, V, z);
        if (m < 7) {
            const float cn = fmaf(nz, cm, -nx * sm), sn = fmaf(nz, sm, nx * cm);
            cm = cn; sm = sn;
        }
    }
    return z;
}

// One point's 64 SH coefficients -> VGPRs as the monomial coefficients of its 15 polynomials in cos(polar):
//   a[m+k, +-m] = sum_{l = m+k, m+k+2, ... < 8} SH_MONO[m][l][k] * h[l, +-m]      (in place: a[m+k] only needs h[l >= m+k])
// SCALE multiplies every coefficient (compile-time: folded into the SH_MONO immediates): the sigmoid kernels evaluate
// -log2(e) * z directly, the argument of their v_exp_f32.
template <bool SCALED = false>
__device__ __forceinline__ void load_mono_coeffs(const float* __restrict__ h, float (&a)[64]) {
    constexpr float S = SCALED ? -1.4426950408889634f : 1.f;
    const float4* h4 = reinterpret_cast<const float4*>(h);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const float4 v = h4[j];
        a[4 * j + 0] = v.x; a[4 * j + 1] = v.y; a[4 * j + 2] = v.z; a[4 * j + 3] = v.w;
    }
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int k = 0; k + m < 8; ++k) {
            float u = a[shk(m + k, m)] * (S * SH_MONO[m][m + k][k]);
            float v = a[shk(m + k, -m)] * (S * SH_MONO[m][m + k][k]);
#pragma unroll
            for (int l = m + k + 2; l < 8; l += 2) {
                u = fmaf(a[shk(l, m)], S * SH_MONO[m][l][k], u);
                v = fmaf(a[shk(l, -m)], S * SH_MONO[m][l][k], v);
            }
            a[shk(m + k, m)] = u;
            if (m) a[shk(m + k, -m)] = v;
        }
}

// SIGMOID: zs = -log2(e) * z (the coefficients carry the factor, load_mono_coeffs<true>): 1 / (1 + exp(-z)) = 1 / (1 + 2^zs).
template <bool SIGMOID>
__device__ __forceinline__ float activate(float zs) {
    if (SIGMOID) return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(zs));
    return fmaxf(zs, 0.f);
}

// Lanes past the end of the cloud (last wave-tile only) read this row instead of a point's coefficients: a constant term of
// -1e30 and nothing else, so z = -2.8e29 and the activation is exactly 0 (sigmoid: 2^(+4e29) = inf, 1 / inf = 0; relu:
// max(z, 0) = 0) -- no per-pair masking multiply.
__device__ const float SC_BLANK_ROW[64] = {-1e30f};

// ---- work decomposition (both kernels) ----------------------------------------------------------------------
// A "wave-unit" is (wave-tile of 64 points, camera).  The U = B * ceil(N/64) * C units are split into W equal
// contiguous ranges, one per wave, with W = min(U, co-resident waves of the chip): the whole grid is resident
// and every wave finishes at the same time (measured: equal-size blocks needing 2.14 rounds cost 3).  A wave
// walks its range as segments (one wave-tile x a camera range): it loads the tile's 64x64 coefficients into
// VGPRs once per segment, then loops over the cameras (wave-uniform -> scalar loads).  No LDS, no barriers.
//
// ---- coverage-gain kernel (mean over points) ---------------------------------------------------------------
// The wave reductions of SC_R consecutive cameras can be issued together (step-major, hiding each other's DPP wait states) while
// the dot products still run one camera at a time.  Measured (100k points x 200 cameras, us per step): 58.6 / 62.0 / 65.2 / 65.3
// for SC_R = 1 / 2 / 3 / 4 -- the extra live values cost a resident wave per SIMD (80 VGPRs -> 6 waves at SC_R = 1, 90 -> 5 at 4),
// which outweighs the shorter tail; interleaving the dot products themselves (SC_G, round 1) lost the same way.  Each
// partial[b][c][wave-tile] is written by exactly one wave; the second pass adds them in a fixed order -> bit-stable results (no
// float atomics).
#ifndef SC_R_N
#define SC_R_N 1
#endif
constexpr int SC_R = SC_R_N;

template <bool SIGMOID>
__global__ __launch_bounds__(SC_BLOCK) void sh_gain_kernel(const float* __restrict__ pts, int pts_stride,
                                                           const float* __restrict__ harm,
                                                           const float* __restrict__ cams, float* __restrict__ partial,
                                                           int N, int C, int n_wtiles, long long U, int W) {
    const int lane = threadIdx.x & (MCR_WAVE - 1);
    const int w = __builtin_amdgcn_readfirstlane(xcd_major_block(blockIdx.x, gridDim.x) * (SC_BLOCK / MCR_WAVE) + threadIdx.x / MCR_WAVE);
    if (w >= W) return;
    long long u = (U * w) / W;
    const long long u_end = (U * (w + 1)) / W;
    while (u < u_end) {
        const long long bt = u / C;                          // flattened (cloud, wave-tile)
        const int c_begin = (int)(u - bt * C);
        const int c_end = (int)min((long long)C, (long long)c_begin + (u_end - u));
        const int b = (int)(bt / n_wtiles);
        const int wt = (int)(bt - (long long)b * n_wtiles);
        u += c_end - c_begin;
        const int n = wt * MCR_WAVE + lane;
        const bool valid = n < N;
        const size_t pn = (size_t)b * N + (valid ? n : N - 1);
        const float px = pts[pn * pts_stride + 0];
        const float py = pts[pn * pts_stride + 1];
        const float pz = pts[pn * pts_stride + 2];
        float hs[64];
        load_mono_coeffs<SIGMOID>(valid ? harm + pn * 64 : SC_BLANK_ROW, hs);
        const float* cam_b = cams + (size_t)b * C * 3;
        float* part_col = partial + (size_t)b * C * n_wtiles + wt;   // partial[b][:][wt]  (camera-major: the reduce reads rows)
        // Cameras are walked SC_R at a time: the SC_R dot products run one after the other (one camera's registers), the SC_R
        // wave reductions run together -- one reduction alone is six DEPENDENT DPP steps with nothing beside them (step-major
        // interleaving hides the wait states).  The camera centres are wave-uniform scalar loads; the next group's are requested
        // before this group's arithmetic (left in place, each iteration opens with an exposed scalar-cache round trip).
        float cn[SC_R][3];
#pragma unroll
        for (int c = 0; c < SC_R; ++c) {
            const int i = min(c_begin + c, c_end - 1);
            cn[c][0] = cam_b[3 * i + 0]; cn[c][1] = cam_b[3 * i + 1]; cn[c][2] = cam_b[3 * i + 2];
        }
        for (int ci = c_begin; ci < c_end; ci += SC_R) {
            float cc[SC_R][3];
#pragma unroll
            for (int c = 0; c < SC_R; ++c) {
                cc[c][0] = cn[c][0]; cc[c][1] = cn[c][1]; cc[c][2] = cn[c][2];
                const int i = min(ci + SC_R + c, c_end - 1);       // ragged tail: surplus slots repeat the last camera
                cn[c][0] = cam_b[3 * i + 0]; cn[c][1] = cam_b[3 * i + 1]; cn[c][2] = cam_b[3 * i + 2];
            }
            float sum[SC_R];
#pragma unroll
            for (int c = 0; c < SC_R; ++c) {
                // rays = X_cam - X_pts (SconeVis.py:230-231)
                const float z = sh_dot(cc[c][0] - px, cc[c][1] - py, cc[c][2] - pz, hs);
                sum[c] = activate<SIGMOID>(z);
                asm volatile("" : "+v"(sum[c]));             // one camera at a time: interleaving the dots costs a resident wave
            }
            wave_sum_to_last_multi<SC_R>(sum);
            if (lane == MCR_WAVE - 1) {
#pragma unroll
                for (int c = 0; c < SC_R; ++c)
                    if (ci + c < c_end) part_col[(size_t)(ci + c) * n_wtiles] = sum[c];
            }
        }
    }
}

// gains[b][c] = (sum_wt partial[b][c][wt]) / N ; one 256-thread block per (b,c) reading its contiguous row, fixed tree order.
__global__ __launch_bounds__(256) void sh_reduce_kernel(const float* __restrict__ partial, float* __restrict__ gains,
                                                        int n_wtiles, int C, float inv_n) {
    __shared__ double s_w[4];
    const int c = blockIdx.x, b = blockIdx.y;
    const float* p = partial + ((size_t)b * C + c) * n_wtiles;
    double acc = 0.0;
    for (int t = threadIdx.x; t < n_wtiles; t += 256) acc += (double)p[t];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) gains[(size_t)b * C + c] = (float)((s_w[0] + s_w[1]) + (s_w[2] + s_w[3])) * inv_n;
}

// ---- per-point kernel (visibilities [B,C,N]) ---------------------------------------------------------------
template <bool SIGMOID>
__global__ __launch_bounds__(SC_BLOCK) void sh_vis_kernel(const float* __restrict__ pts, int pts_stride,
                                                          const float* __restrict__ harm,
                                                          const float* __restrict__ cams, float* __restrict__ out,
                                                          int N, int C, int n_wtiles, long long U, int W) {
    const int lane = threadIdx.x & (MCR_WAVE - 1);
    const int w = __builtin_amdgcn_readfirstlane(xcd_major_block(blockIdx.x, gridDim.x) * (SC_BLOCK / MCR_WAVE) + threadIdx.x / MCR_WAVE);
    if (w >= W) return;
    long long u = (U * w) / W;
    const long long u_end = (U * (w + 1)) / W;
    while (u < u_end) {
        const long long bt = u / C;                          // flattened (cloud, wave-tile)
        const int c_begin = (int)(u - bt * C);
        const int c_end = (int)min((long long)C, (long long)c_begin + (u_end - u));
        const int b = (int)(bt / n_wtiles);
        const int wt = (int)(bt - (long long)b * n_wtiles);
        u += c_end - c_begin;
        const int n = wt * MCR_WAVE + lane;
        const bool valid = n < N;
        const size_t pn = (size_t)b * N + (valid ? n : N - 1);
        const float px = pts[pn * pts_stride + 0];
        const float py = pts[pn * pts_stride + 1];
        const float pz = pts[pn * pts_stride + 2];
        float hs[64];
        load_mono_coeffs<SIGMOID>(harm + pn * 64, hs);
        const float* cam_b = cams + (size_t)b * C * 3;
        for (int ci = c_begin; ci < c_end; ++ci) {
            const float z = sh_dot(cam_b[3 * ci + 0] - px, cam_b[3 * ci + 1] - py, cam_b[3 * ci + 2] - pz, hs);
            const float v = activate<SIGMOID>(z);
            if (valid) out[((size_t)b * C + ci) * N + n] = v;
        }
    }
}

// Waves that are co-resident on the chip for `kernel` (occupancy API: from its real VGPR allocation).
template <typename Kern>
static int resident_waves(Kern kernel, int waves_per_simd_override) {
    int dev = 0, n_cu = 256, blocks_per_cu = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
        n_cu = prop.multiProcessorCount;
    if (waves_per_simd_override > 0) return n_cu * 4 * waves_per_simd_override;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks_per_cu, kernel, SC_BLOCK, 0) != hipSuccess || blocks_per_cu <= 0)
        blocks_per_cu = 4;
    return n_cu * blocks_per_cu * (SC_BLOCK / MCR_WAVE);
}

template <typename Kern>
static int resident_waves_cached(Kern kernel, int waves_per_simd_override, int* cache) {
    if (waves_per_simd_override > 0) return resident_waves(kernel, waves_per_simd_override);
    if (!*cache) *cache = resident_waves(kernel, 0);
    return *cache;
}

}  // namespace mcr

using namespace mcr;

extern "C" {

size_t mcr_sh_coverage_gain_workspace_bytes(int64_t B, int64_t N, int64_t C) {
    return (size_t)B * (size_t)cdiv(N, MCR_WAVE) * (size_t)C * sizeof(float);
}

static int sh_gain_impl(const char* who, bool reduce, const float* pts, int pts_dim, const float* harmonics, const float* cams,
                        float* gains, int64_t B, int64_t N, int64_t C, int use_sigmoid, int waves_per_simd, void* workspace,
                        size_t workspace_bytes, void* stream) {
    MCR_REQUIRE(pts && harmonics && cams && (gains || !reduce), "%s: null pointer", who);
    MCR_REQUIRE(pts_dim >= 3, "%s: pts_dim must be >= 3 (got %d)", who, pts_dim);
    MCR_REQUIRE(B > 0 && N > 0 && C > 0, "%s: empty problem B=%ld N=%ld C=%ld", who, (long)B, (long)N, (long)C);
    MCR_REQUIRE(C <= 65535 * 64 && N < (1ll << 31) && B <= 65535, "%s: problem too large", who);
    MCR_REQUIRE(waves_per_simd >= 0 && waves_per_simd <= 16, "%s: waves_per_simd out of range", who);
    MCR_REQUIRE(workspace && workspace_bytes >= mcr_sh_coverage_gain_workspace_bytes(B, N, C), "%s: workspace too small", who);
    static int cache_sig = 0, cache_relu = 0;
    const int resident = use_sigmoid ? resident_waves_cached(sh_gain_kernel<true>, waves_per_simd, &cache_sig)
                                     : resident_waves_cached(sh_gain_kernel<false>, waves_per_simd, &cache_relu);
    const int n_wtiles = (int)cdiv(N, MCR_WAVE);
    const long long U = (long long)B * n_wtiles * C;
    const int W = (int)std::min<long long>(U, resident);
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((unsigned)cdiv(W, SC_BLOCK / MCR_WAVE));
    float* partial = (float*)workspace;
    if (use_sigmoid)
        hipLaunchKernelGGL(sh_gain_kernel<true>, grid, dim3(SC_BLOCK), 0, s, pts, pts_dim, harmonics, cams, partial,
                           (int)N, (int)C, n_wtiles, U, W);
    else
        hipLaunchKernelGGL(sh_gain_kernel<false>, grid, dim3(SC_BLOCK), 0, s, pts, pts_dim, harmonics, cams, partial,
                           (int)N, (int)C, n_wtiles, U, W);
    MCR_LAUNCH_CHECK("sh_gain_kernel");
    if (!reduce) return 0;
    hipLaunchKernelGGL(sh_reduce_kernel, dim3((unsigned)C, (unsigned)B), dim3(256), 0, s, partial, gains, n_wtiles, (int)C,
                       1.0f / (float)N);
    MCR_LAUNCH_CHECK("sh_reduce_kernel");
    return 0;
}

int mcr_sh_coverage_gain(const float* pts, int pts_dim, const float* harmonics, const float* cams, float* gains,
                         int64_t B, int64_t N, int64_t C, int use_sigmoid, int waves_per_simd, void* workspace,
                         size_t workspace_bytes, void* stream) {
    return sh_gain_impl("mcr_sh_coverage_gain", true, pts, pts_dim, harmonics, cams, gains, B, N, C, use_sigmoid, waves_per_simd,
                        workspace, workspace_bytes, stream);
}

int mcr_sh_coverage_gain_partials(const float* pts, int pts_dim, const float* harmonics, const float* cams, int64_t B, int64_t N,
                                  int64_t C, int use_sigmoid, int waves_per_simd, void* workspace, size_t workspace_bytes,
                                  void* stream) {
    return sh_gain_impl("mcr_sh_coverage_gain_partials", false, pts, pts_dim, harmonics, cams, nullptr, B, N, C, use_sigmoid,
                        waves_per_simd, workspace, workspace_bytes, stream);
}

int mcr_sh_visibilities(const float* pts, int pts_dim, const float* harmonics, const float* cams, float* vis,
                        int64_t B, int64_t N, int64_t C, int use_sigmoid, void* stream) {
    MCR_REQUIRE(pts && harmonics && cams && vis, "mcr_sh_visibilities: null pointer");
    MCR_REQUIRE(pts_dim >= 3, "mcr_sh_visibilities: pts_dim must be >= 3 (got %d)", pts_dim);
    MCR_REQUIRE(B > 0 && N > 0 && C > 0, "mcr_sh_visibilities: empty problem");
    MCR_REQUIRE(C < (1 << 24) && N < (1ll << 31), "mcr_sh_visibilities: problem too large");
    static int cache_sig = 0, cache_relu = 0;
    const int resident = use_sigmoid ? resident_waves_cached(sh_vis_kernel<true>, 0, &cache_sig)
                                     : resident_waves_cached(sh_vis_kernel<false>, 0, &cache_relu);
    const int n_wtiles = (int)cdiv(N, MCR_WAVE);
    const long long U = (long long)B * n_wtiles * C;
    const int W = (int)std::min<long long>(U, resident);
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((unsigned)cdiv(W, SC_BLOCK / MCR_WAVE));
    if (use_sigmoid)
        hipLaunchKernelGGL(sh_vis_kernel<true>, grid, dim3(SC_BLOCK), 0, s, pts, pts_dim, harmonics, cams, vis, (int)N,
                           (int)C, n_wtiles, U, W);
    else
        hipLaunchKernelGGL(sh_vis_kernel<false>, grid, dim3(SC_BLOCK), 0, s, pts, pts_dim, harmonics, cams, vis, (int)N,
                           (int)C, n_wtiles, U, W);
    MCR_LAUNCH_CHECK("sh_vis_kernel");
    return 0;
}

}  // extern "C"
