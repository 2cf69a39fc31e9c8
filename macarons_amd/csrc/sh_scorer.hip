// K9 — spherical-harmonic coverage-gain scorer for gfx950 (MI355X).
//
// Replaces the op sequence of the reference's
//   macarons/networks/SconeVis.py:210-252  compute_coverage_gain   -> gains [B,C]
//   macarons/networks/SconeVis.py:164-208  compute_visibilities    -> vis   [B,C,N]
//   (== macarons/networks/Macarons.py:138-178 compute_visibility_gains)
// which materialise [B*C*N,64] SH tensors three times.  Here: one lane owns one point (its 64 SH
// coefficients live in VGPRs, pre-scaled once by the recurrence constants), cameras are broadcast from
// LDS, the 64 real SH of the ray direction are produced by trig-free recurrences and contracted on the
// fly (~170 VALU ops per (point,camera) pair), sigmoid/relu applied, and the per-camera sum over points
// is a wave64 DPP reduction -> per-tile partials -> a deterministic second-pass reduce (bit-stable
// run to run).  Bound: fp32 VALU (370 algorithmic flop / pair; N*268 B of HBM per cloud).
//
// Conventions (reference CustomGeometry.py:27-45, spherical_harmonics.py:67-140): polar axis +Y,
// azimuth from +Z toward +X; channel k = l*l + l + m; Condon-Shortley phase included.
#include "common.h"
#include "sh_consts.inc"

namespace mcr {

constexpr int SC_BLOCK = 256;       // 4 waves; one point per lane
constexpr int SC_MAX_CHUNK = 64;    // cameras per block (LDS staging)

__device__ __forceinline__ constexpr int shk(int l, int m) { return l * l + l + m; }

// z = sum_k Y_k(d) h_k  with hs[k] = SH_LAMBDA[l][|m|] * h_k  (see gen_sh_consts.py for the algebra).
__device__ __forceinline__ float sh_dot(float dx, float dy, float dz, const float (&hs)[64]) {
    const float rho2 = fmaf(dz, dz, dx * dx);
    const float r2 = fmaf(dy, dy, rho2);
    const float ir = __builtin_amdgcn_rsqf(r2);
    const float irho = rho2 > 0.f ? __builtin_amdgcn_rsqf(rho2) : 0.f;   // ray || Y: m>0 terms vanish
    const float ct = dy * ir;                 // cos(polar)
    const float st = (rho2 * irho) * ir;      // sin(polar) = rho / r
    const float cp = dz * irho;               // cos(azim)
    const float sp = dx * irho;               // sin(azim)

    // m = 0 column
    float z = hs[0];
    {
        float r2_ = 1.f, r1_ = ct;
        z = fmaf(r1_, hs[shk(1, 0)], z);
#pragma unroll
        for (int l = 2; l < 8; ++l) {
            const float r = fmaf(ct, r1_, -SH_BP[l][0] * r2_);
            z = fmaf(r, hs[shk(l, 0)], z);
            r2_ = r1_;
            r1_ = r;
        }
    }
    // m = 1..7 : U_m = sum_l R_l^m h[l,+m], V_m = sum_l R_l^m h[l,-m];  z += cos(m p) U_m + sin(m p) V_m
    const float tc = cp + cp;
    float cm1 = 1.f, cm = cp, sm1 = 0.f, sm = sp;
    float stm = 1.f;
#pragma unroll
    for (int m = 1; m < 8; ++m) {
        stm *= st;
        float U = stm * hs[shk(m, m)];
        float V = stm * hs[shk(m, -m)];
        float r2_ = stm, r1_ = ct * stm;
        if (m < 7) {
            U = fmaf(r1_, hs[shk(m + 1, m)], U);
            V = fmaf(r1_, hs[shk(m + 1, -m)], V);
        }
#pragma unroll
        for (int l = m + 2; l < 8; ++l) {
            const float r = fmaf(ct, r1_, -SH_BP[l][m] * r2_);
            U = fmaf(r, hs[shk(l, m)], U);
            V = fmaf(r, hs[shk(l, -m)], V);
            r2_ = r1_;
            r1_ = r;
        }
        z = fmaf(cm, U, z);
        z = fmaf(sm, V, z);
        const float cn = fmaf(tc, cm, -cm1), sn = fmaf(tc, sm, -sm1);
        cm1 = cm; cm = cn; sm1 = sm; sm = sn;
    }
    return z;
}

template <bool SIGMOID>
__device__ __forceinline__ float activate(float z) {
    if (SIGMOID) {
        // 1 / (1 + exp(-z));  exp via v_exp_f32 (2^x)
        const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * z);
        return __builtin_amdgcn_rcpf(1.f + e);
    }
    return fmaxf(z, 0.f);
}

// grid = (point tiles, camera chunks, B)
template <bool PER_POINT, bool SIGMOID>
__global__ __launch_bounds__(SC_BLOCK) void sh_score_kernel(const float* __restrict__ pts, int pts_stride,
                                                            const float* __restrict__ harm,
                                                            const float* __restrict__ cams, float* __restrict__ out,
                                                            int N, int C, int chunk, int n_tiles) {
    __shared__ float s_cam[SC_MAX_CHUNK * 3];
    __shared__ float s_part[SC_BLOCK / MCR_WAVE][SC_MAX_CHUNK];

    const int tid = threadIdx.x;
    const int tile = blockIdx.x;
    const int c0 = blockIdx.y * chunk;
    const int nc = min(chunk, C - c0);
    const int b = blockIdx.z;

    if (tid < nc * 3) s_cam[tid] = cams[((size_t)b * C + c0) * 3 + tid];

    const int n = tile * SC_BLOCK + tid;
    const bool valid = n < N;
    const size_t pn = (size_t)b * N + (valid ? n : N - 1);

    const float px = pts[pn * pts_stride + 0];
    const float py = pts[pn * pts_stride + 1];
    const float pz = pts[pn * pts_stride + 2];

    float hs[64];
    {
        const float4* h4 = reinterpret_cast<const float4*>(harm + pn * 64);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float4 v = h4[j];
            hs[4 * j + 0] = v.x; hs[4 * j + 1] = v.y; hs[4 * j + 2] = v.z; hs[4 * j + 3] = v.w;
        }
#pragma unroll
        for (int l = 0; l < 8; ++l)
#pragma unroll
            for (int m = -l; m <= l; ++m) hs[shk(l, m)] *= SH_LAMBDA[l][m < 0 ? -m : m];
    }
    __syncthreads();

    const int lane = tid & (MCR_WAVE - 1);
    const int wave = tid / MCR_WAVE;

    for (int ci = 0; ci < nc; ++ci) {
        const float dx = s_cam[3 * ci + 0] - px;     // rays = X_cam - X_pts (SconeVis.py:230-231)
        const float dy = s_cam[3 * ci + 1] - py;
        const float dz = s_cam[3 * ci + 2] - pz;
        float v = activate<SIGMOID>(sh_dot(dx, dy, dz, hs));
        if (PER_POINT) {
            if (valid) out[((size_t)b * C + c0 + ci) * N + n] = v;
        } else {
            v = valid ? v : 0.f;
            const float s = wave_sum_to_last(v);
            if (lane == MCR_WAVE - 1) s_part[wave][ci] = s;
        }
    }
    if (!PER_POINT) {
        __syncthreads();
        if (tid < nc) {
            const float s = (s_part[0][tid] + s_part[1][tid]) + (s_part[2][tid] + s_part[3][tid]);
            out[((size_t)b * n_tiles + tile) * C + c0 + tid] = s;     // partial[b][tile][c]
        }
    }
}

// gains[b][c] = (sum_tile partial[b][tile][c]) / N      (fixed order -> deterministic)
__global__ void sh_reduce_kernel(const float* __restrict__ partial, float* __restrict__ gains, int n_tiles, int C,
                                 int BC, float inv_n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= BC) return;
    const int b = i / C, c = i - b * C;
    const float* p = partial + (size_t)b * n_tiles * C + c;
    double acc = 0.0;
    for (int t = 0; t < n_tiles; ++t) acc += (double)p[(size_t)t * C];
    gains[i] = (float)acc * inv_n;
}

static int pick_chunk(int64_t B, int64_t N, int64_t C, int requested) {
    if (requested > 0) return requested > SC_MAX_CHUNK ? SC_MAX_CHUNK : requested;
    // Aim for >= ~8 blocks per CU (256 CUs) so the equal-length blocks tile the chip with a small tail,
    // but keep >= 8 cameras per block so the 256 B/point coefficient load stays amortised.
    const int64_t tiles = cdiv(N, SC_BLOCK) * B;
    int chunk = SC_MAX_CHUNK;
    while (chunk > 8 && tiles * cdiv(C, chunk) < 2048) chunk /= 2;
    if (chunk > C) chunk = (int)C;
    return chunk < 1 ? 1 : chunk;
}

}  // namespace mcr

using namespace mcr;

extern "C" {

size_t mcr_sh_coverage_gain_workspace_bytes(int64_t B, int64_t N, int64_t C) {
    return (size_t)B * (size_t)cdiv(N, SC_BLOCK) * (size_t)C * sizeof(float);
}

int mcr_sh_coverage_gain(const float* pts, int pts_dim, const float* harmonics, const float* cams, float* gains,
                         int64_t B, int64_t N, int64_t C, int use_sigmoid, int cam_chunk, void* workspace,
                         size_t workspace_bytes, void* stream) {
    MCR_REQUIRE(pts && harmonics && cams && gains, "mcr_sh_coverage_gain: null pointer");
    MCR_REQUIRE(pts_dim >= 3, "mcr_sh_coverage_gain: pts_dim must be >= 3 (got %d)", pts_dim);
    MCR_REQUIRE(B > 0 && N > 0 && C > 0, "mcr_sh_coverage_gain: empty problem B=%ld N=%ld C=%ld", (long)B, (long)N, (long)C);
    MCR_REQUIRE(B <= 65535, "mcr_sh_coverage_gain: B too large");
    MCR_REQUIRE(workspace && workspace_bytes >= mcr_sh_coverage_gain_workspace_bytes(B, N, C),
                "mcr_sh_coverage_gain: workspace too small");
    const int chunk = pick_chunk(B, N, C, cam_chunk);
    const int n_tiles = (int)cdiv(N, SC_BLOCK);
    const int n_chunks = (int)cdiv(C, chunk);
    MCR_REQUIRE(n_chunks <= 65535, "mcr_sh_coverage_gain: too many camera chunks");
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(n_tiles, n_chunks, (unsigned)B);
    float* partial = (float*)workspace;
    if (use_sigmoid)
        hipLaunchKernelGGL((sh_score_kernel<false, true>), grid, dim3(SC_BLOCK), 0, s, pts, pts_dim, harmonics, cams,
                           partial, (int)N, (int)C, chunk, n_tiles);
    else
        hipLaunchKernelGGL((sh_score_kernel<false, false>), grid, dim3(SC_BLOCK), 0, s, pts, pts_dim, harmonics, cams,
                           partial, (int)N, (int)C, chunk, n_tiles);
    MCR_LAUNCH_CHECK("sh_score_kernel");
    const int BC = (int)(B * C);
    hipLaunchKernelGGL(sh_reduce_kernel, dim3((unsigned)cdiv(BC, 128)), dim3(128), 0, s, partial, gains, n_tiles, (int)C,
                       BC, 1.0f / (float)N);
    MCR_LAUNCH_CHECK("sh_reduce_kernel");
    return 0;
}

int mcr_sh_visibilities(const float* pts, int pts_dim, const float* harmonics, const float* cams, float* vis,
                        int64_t B, int64_t N, int64_t C, int use_sigmoid, void* stream) {
    MCR_REQUIRE(pts && harmonics && cams && vis, "mcr_sh_visibilities: null pointer");
    MCR_REQUIRE(pts_dim >= 3, "mcr_sh_visibilities: pts_dim must be >= 3 (got %d)", pts_dim);
    MCR_REQUIRE(B > 0 && N > 0 && C > 0, "mcr_sh_visibilities: empty problem");
    MCR_REQUIRE(B <= 65535, "mcr_sh_visibilities: B too large");
    const int chunk = pick_chunk(B, N, C, 0);
    const int n_tiles = (int)cdiv(N, SC_BLOCK);
    const int n_chunks = (int)cdiv(C, chunk);
    MCR_REQUIRE(n_chunks <= 65535, "mcr_sh_visibilities: too many camera chunks");
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(n_tiles, n_chunks, (unsigned)B);
    if (use_sigmoid)
        hipLaunchKernelGGL((sh_score_kernel<true, true>), grid, dim3(SC_BLOCK), 0, s, pts, pts_dim, harmonics, cams, vis,
                           (int)N, (int)C, chunk, n_tiles);
    else
        hipLaunchKernelGGL((sh_score_kernel<true, false>), grid, dim3(SC_BLOCK), 0, s, pts, pts_dim, harmonics, cams, vis,
                           (int)N, (int)C, chunk, n_tiles);
    MCR_LAUNCH_CHECK("sh_score_kernel<per_point>");
    return 0;
}

}  // extern "C"
