// K9 — spherical-harmonic coverage-gain scorer for gfx950 (MI355X).
//
// Replaces the op sequence of the reference's
//   macarons/networks/SconeVis.py:210-252  compute_coverage_gain   -> gains [B,C]
//   macarons/networks/SconeVis.py:164-208  compute_visibilities    -> vis   [B,C,N]
//   (== macarons/networks/Macarons.py:138-178 compute_visibility_gains)
// which materialise [B*C*N,64] SH tensors three times.  Here: one lane owns one point (its 64 SH
// coefficients live in VGPRs, turned once into the monomial coefficients of 15 polynomials in cos(polar)), cameras
// come from wave-uniform scalar loads, the dot with the 64 real SH of the ray direction is evaluated trig-free by
// Horner steps (82 VALU ops per (point,camera) pair, 94 with activation and reduction), sigmoid/relu applied, and the per-camera sum over points
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
// a (64 per point, same storage as the SH coefficients) are produced once per point by to_mono_coeffs.  Then
//   z = U_0(x) + sum_{m>=1} ( Re w^m U_m(x) + Im w^m V_m(x) ),   w = n_z + i n_x
// and the sum over the orders is itself a Horner evaluation, in w over the complex numbers (below): 49 Horner FMAs in x
// + 24 for the six complex steps + 2 for the last real part + 7 to normalise = 82 VALU ops per (point, camera) pair (94 in the
// gain kernel's loop with the ray, the activation and the wave reduction).  The form this replaces kept the powers w^m
// (4 ops per order) and combined cm U_m + sm V_m into z (2 per order, a serial 14-step chain): 94 / 106 ops, gain kernel
// 50.4 -> 46.5 us at N = 100k, C = 200 (results 1.2e-7 apart); the rescaled-recurrence form before that needed 130.
// Measured on MI355X (tools/ubench): a dependent v_fma_f32 chain issues every ~8.8 cycles per wave; the 15 Horner
// chains are independent.  A packed form ((U_m, V_m) as ONE v_pk_fma_f32 chain per order: 70 instead of 106 vector
// instructions per pair) was built and measured (sh_dot_pk_rate.hip, NOTES): a packed instruction costs two scalar ones in
// this stream -- sh_dot alone 295 -> 275 cycles per pair and SIMD at 6 waves, the kernel 50.3 -> 50.0 us -- not kept.
// z = U_0(x) + Re( sum_{m>=1} w^m (U_m - i V_m) ): the sum over the orders as ONE complex Horner evaluation in w
//   A_7 = P_7,  A_m = A_{m+1} w + P_m  (m = 6..1),  z = U_0 + Re(w A_1),   P_m = U_m(x) - i V_m(x),  A = ar - i bi
// -- 4 FMAs per order instead of 4 for the power w^m and 2 to combine, and no serial 14-step accumulation into z.
__device__ __forceinline__ float sh_dot(float dx, float dy, float dz, const float (&a)[64]) {
    const float r2 = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
    const float ir = __builtin_amdgcn_rsqf(r2);
    const float nx = dx * ir, ct = dy * ir, nz = dz * ir;
    float z = a[shk(7, 0)];
#pragma unroll
    for (int l = 6; l >= 0; --l) z = fmaf(ct, z, a[shk(l, 0)]);
    float ar = a[shk(7, 7)], bi = a[shk(7, -7)];
#pragma unroll
    for (int m = 6; m >= 1; --m) {
        float U = a[shk(7, m)], V = a[shk(7, -m)];
#pragma unroll
        for (int l = 6; l >= m; --l) {
            U = fmaf(ct, U, a[shk(l, m)]);
            V = fmaf(ct, V, a[shk(l, -m)]);
        }
        const float nr = fmaf(ar, nz, fmaf(bi, nx, U));
        const float nb = fmaf(bi, nz, fmaf(-ar, nx, V));
        ar = nr; bi = nb;
    }
    return fmaf(nz, ar, fmaf(nx, bi, z));
}

// One point's 64 SH coefficients -> VGPRs as the monomial coefficients of its 15 polynomials in cos(polar):
//   a[m+k, +-m] = sum_{l = m+k, m+k+2, ... < 8} SH_MONO[m][l][k] * h[l, +-m]      (in place: a[m+k] only needs h[l >= m+k])
// SCALE multiplies every coefficient (compile-time: folded into the SH_MONO immediates): the sigmoid kernels evaluate
// -log2(e) * z directly, the argument of their v_exp_f32.
template <bool SCALED = false>
__device__ __forceinline__ void to_mono_coeffs(float (&a)[64]) {
    constexpr float S = SCALED ? -1.4426950408889634f : 1.f;
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

// SIGMOID: zs = -log2(e) * z (the coefficients carry the factor, to_mono_coeffs<true>): 1 / (1 + exp(-z)) = 1 / (1 + 2^zs).
template <bool SIGMOID>
__device__ __forceinline__ float activate(float zs) {
    if (SIGMOID) return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(zs));
    return fmaxf(zs, 0.f);
}

// Lanes past the end of the cloud (last wave-tile only) carry the cloud's last point with the constant coefficient replaced by
// -1e30 (scaled form: +1e30), so z = -1e30 and the activation is exactly 0 (sigmoid: 2^(+1e30) = inf, 1 / inf = 0; relu:
// max(z, 0) = 0) -- no per-pair masking multiply.
template <bool SIGMOID>
__device__ __forceinline__ void blank_lane(bool valid, float (&a)[64]) {
    float big;
    asm volatile("v_mov_b32 %0, %1" : "=v"(big) : "n"(SIGMOID ? 0x7149f2ca : 0xf149f2ca));     // +-1e30, made here (not kept live through the camera loop)
    a[0] = valid ? a[0] : big;
}

// The 64 x 64 coefficients of a wave-tile -> one row per lane.  A lane reading its own 256-byte row with sixteen 16-byte loads
// makes every load instruction of the wave touch 64 different cache lines (1024 line requests for 128 lines; the L1 of the CU,
// shared by 24 such waves, cannot hold them between instructions): the fixed part of a launch was 10 us of 53.  Here four
// neighbouring lanes read one 64-byte piece of a row (16 lines per instruction, each line in two instructions), the 16 loads are
// all in flight together, and the tile is turned by quarters through a 5 KB strip of LDS that only this wave touches (DS
// operations of one wave execute in order: no barrier; rows padded to 80 bytes).  Rows past the end of the cloud repeat its last row.
struct ScStage { float4 q[MCR_WAVE][5]; };
__device__ __forceinline__ void load_tile_rows(const float* __restrict__ harm_b, int row0, int N, int lane, ScStage& st,
                                               float (&a)[64]) {
    asm volatile("" : "+v"(lane));      // addresses derived from the lane index are rebuilt per tile, not kept live through the camera loop
    const int chunk = lane & 3, rsub = lane >> 2;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 v[4][4];                                             // [column quarter][row group]
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        // uniform base + 32-bit byte offset (the scalar-base addressing form: one address register per row group)
        const float* src = reinterpret_cast<const float*>(reinterpret_cast<const char*>(harm_b) +
                                                          (unsigned)(min(row0 + 16 * i + rsub, N - 1) * 256 + 16 * chunk));
#pragma unroll
        for (int p = 0; p < 4; ++p) v[p][i] = *reinterpret_cast<const f32x4*>(src + 16 * p);
    }
    // In place: the four row groups of a quarter go out and the lane's own row comes back into the SAME registers (written as
    // one asm block with tied operands; left to the register allocator the turn needed 101 VGPRs and cost two resident waves).
    const unsigned wa = (unsigned)(size_t)&st.q[rsub][chunk], ra = (unsigned)(size_t)&st.q[lane][0];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        asm volatile("ds_write_b128 %4, %0\n\t"
                     "ds_write_b128 %4, %1 offset:%c6\n\t"
                     "ds_write_b128 %4, %2 offset:%c7\n\t"
                     "ds_write_b128 %4, %3 offset:%c8\n\t"
                     "ds_read_b128 %0, %5\n\t"
                     "ds_read_b128 %1, %5 offset:16\n\t"
                     "ds_read_b128 %2, %5 offset:32\n\t"
                     "ds_read_b128 %3, %5 offset:48\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "+v"(v[p][0]), "+v"(v[p][1]), "+v"(v[p][2]), "+v"(v[p][3])
                     : "v"(wa), "v"(ra), "n"(16 * 80), "n"(32 * 80), "n"(48 * 80)
                     : "memory");
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            a[16 * p + 4 * c + 0] = v[p][c].x; a[16 * p + 4 * c + 1] = v[p][c].y;
            a[16 * p + 4 * c + 2] = v[p][c].z; a[16 * p + 4 * c + 3] = v[p][c].w;
        }
    }
    // plain 32-bit values from here on: with the coefficients still tied to their 128-bit tuples the instruction scheduler
    // orders the camera loop differently (same instructions), 7 % slower per camera
#pragma unroll
    for (int k = 0; k < 64; ++k) asm volatile("" : "+v"(a[k]));
}

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
    __shared__ ScStage s_stage[SC_BLOCK / MCR_WAVE];
    const int lane = threadIdx.x & (MCR_WAVE - 1);
    const int w = __builtin_amdgcn_readfirstlane(xcd_major_block(blockIdx.x, gridDim.x) * (SC_BLOCK / MCR_WAVE) + threadIdx.x / MCR_WAVE);
    if (w >= W) return;
    // The wave's range of units -> (cloud b, wave-tile wt, first camera, units left): divided ONCE, then walked (the quotients of a
    // per-segment division, and the reciprocal seeds of its expansion, stayed in vector registers through the camera loop).
    int b, wt, c_begin, left;
    {
        const long long u = (U * w) / W, bt = u / C;
        b = __builtin_amdgcn_readfirstlane((int)(bt / n_wtiles));
        wt = __builtin_amdgcn_readfirstlane((int)(bt - (long long)(bt / n_wtiles) * n_wtiles));
        c_begin = __builtin_amdgcn_readfirstlane((int)(u - bt * C));
        left = __builtin_amdgcn_readfirstlane((int)((U * (w + 1)) / W - u));
    }
    const int wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x / MCR_WAVE);
    for (; left > 0; c_begin = 0, b += (wt + 1 == n_wtiles), wt = (wt + 1 == n_wtiles ? 0 : wt + 1)) {
        const int c_end = min(C, c_begin + left);
        left -= c_end - c_begin;
        const int n = wt * MCR_WAVE + lane;
        const bool valid = n < N;
        const size_t pn = (size_t)b * N + (valid ? n : N - 1);
        float hs[64];
        load_tile_rows(harm + (size_t)b * N * 64, wt * MCR_WAVE, N, lane, s_stage[wave_in_block], hs);
        const float px = pts[pn * pts_stride + 0];            // after the tile: its 64 landing registers are the peak
        const float py = pts[pn * pts_stride + 1];
        const float pz = pts[pn * pts_stride + 2];
        to_mono_coeffs<SIGMOID>(hs);
        blank_lane<SIGMOID>(valid, hs);
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

// The decision behind the gains (testers/shapenet.py:172: torch.max over the cameras): record[b] = (max_c gains[b][c], first arg-max as
// fp32) with torch.max's ordering -- NaN beats every number (the first NaN wins), ties go to the lower index.  One block per cloud.
__global__ __launch_bounds__(256) void sh_best_kernel(const float* __restrict__ gains, int C, float* __restrict__ record) {
    __shared__ float s_v[4], s_i[4];
    const int b = blockIdx.x, lane = threadIdx.x & 63;
    float bv = -__builtin_inff(), bi = 3.0e38f;
    bool first = true;
    auto before = [](float v, float i, float ov, float oi) {          // (v, i) precedes (ov, oi)
        const bool vn = v != v, on = ov != ov;
        return vn != on ? vn : (!vn && v != ov ? v > ov : i < oi);
    };
    for (int c = threadIdx.x; c < C; c += 256) {
        const float g = gains[(size_t)b * C + c];
        if (first || before(g, (float)c, bv, bi)) { bv = g; bi = (float)c; }
        first = false;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o, 64), oi = __shfl_xor(bi, o, 64);
        if (before(ov, oi, bv, bi)) { bv = ov; bi = oi; }
    }
    if (lane == 0) { s_v[threadIdx.x >> 6] = bv; s_i[threadIdx.x >> 6] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w)
            if (before(s_v[w], s_i[w], bv, bi)) { bv = s_v[w]; bi = s_i[w]; }
        record[2 * b] = bv;
        record[2 * b + 1] = bi;
    }
}

// ---- per-point kernel (visibilities [B,C,N]) ---------------------------------------------------------------
template <bool SIGMOID>
__global__ __launch_bounds__(SC_BLOCK) void sh_vis_kernel(const float* __restrict__ pts, int pts_stride,
                                                          const float* __restrict__ harm,
                                                          const float* __restrict__ cams, float* __restrict__ out,
                                                          int N, int C, int n_wtiles, long long U, int W) {
    __shared__ ScStage s_stage[SC_BLOCK / MCR_WAVE];
    const int lane = threadIdx.x & (MCR_WAVE - 1);
    const int w = __builtin_amdgcn_readfirstlane(xcd_major_block(blockIdx.x, gridDim.x) * (SC_BLOCK / MCR_WAVE) + threadIdx.x / MCR_WAVE);
    if (w >= W) return;
    // The wave's range of units -> (cloud b, wave-tile wt, first camera, units left): divided ONCE, then walked (the quotients of a
    // per-segment division, and the reciprocal seeds of its expansion, stayed in vector registers through the camera loop).
    int b, wt, c_begin, left;
    {
        const long long u = (U * w) / W, bt = u / C;
        b = __builtin_amdgcn_readfirstlane((int)(bt / n_wtiles));
        wt = __builtin_amdgcn_readfirstlane((int)(bt - (long long)(bt / n_wtiles) * n_wtiles));
        c_begin = __builtin_amdgcn_readfirstlane((int)(u - bt * C));
        left = __builtin_amdgcn_readfirstlane((int)((U * (w + 1)) / W - u));
    }
    const int wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x / MCR_WAVE);
    for (; left > 0; c_begin = 0, b += (wt + 1 == n_wtiles), wt = (wt + 1 == n_wtiles ? 0 : wt + 1)) {
        const int c_end = min(C, c_begin + left);
        left -= c_end - c_begin;
        const int n = wt * MCR_WAVE + lane;
        const bool valid = n < N;
        const size_t pn = (size_t)b * N + (valid ? n : N - 1);
        float hs[64];
        load_tile_rows(harm + (size_t)b * N * 64, wt * MCR_WAVE, N, lane, s_stage[wave_in_block], hs);
        const float px = pts[pn * pts_stride + 0];
        const float py = pts[pn * pts_stride + 1];
        const float pz = pts[pn * pts_stride + 2];
        to_mono_coeffs<SIGMOID>(hs);
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
    // N <= 2^23: load_tile_rows addresses a coefficient row as a 32-bit byte offset (row * 256) from the cloud's base
    MCR_REQUIRE(C <= 65535 * 64 && N <= (1ll << 23) && B <= 65535, "%s: problem too large (N <= 2^23 points per cloud)", who);
    MCR_REQUIRE(waves_per_simd >= 0 && waves_per_simd <= 16, "%s: waves_per_simd out of range", who);
    MCR_REQUIRE(workspace && workspace_bytes >= mcr_sh_coverage_gain_workspace_bytes(B, N, C), "%s: workspace too small", who);
    static int cache_sig = 0, cache_relu = 0;
    const int resident = use_sigmoid ? resident_waves_cached(sh_gain_kernel<true>, waves_per_simd, &cache_sig)
                                     : resident_waves_cached(sh_gain_kernel<false>, waves_per_simd, &cache_relu);
    const int n_wtiles = (int)cdiv(N, MCR_WAVE);
    const long long U = (long long)B * n_wtiles * C;
    const int W = (int)std::min<long long>(U, resident);
    MCR_REQUIRE(U / W < (1ll << 30), "%s: problem too large", who);          // a wave's unit count is an int
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

int mcr_sh_coverage_gain_best(const float* pts, int pts_dim, const float* harmonics, const float* cams, float* gains, float* record,
                              int64_t B, int64_t N, int64_t C, int use_sigmoid, int waves_per_simd, void* workspace,
                              size_t workspace_bytes, void* stream) {
    MCR_REQUIRE(record, "mcr_sh_coverage_gain_best: null pointer");
    MCR_REQUIRE(C < (1ll << 24), "mcr_sh_coverage_gain_best: camera indices must stay below 2^24");
    if (int e = sh_gain_impl("mcr_sh_coverage_gain_best", true, pts, pts_dim, harmonics, cams, gains, B, N, C, use_sigmoid, waves_per_simd,
                             workspace, workspace_bytes, stream))
        return e;
    hipLaunchKernelGGL(sh_best_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, gains, (int)C, record);
    MCR_LAUNCH_CHECK("sh_best_kernel");
    return 0;
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
    MCR_REQUIRE(C < (1 << 24) && N <= (1ll << 23), "mcr_sh_visibilities: problem too large (N <= 2^23 points per cloud)");
    static int cache_sig = 0, cache_relu = 0;
    const int resident = use_sigmoid ? resident_waves_cached(sh_vis_kernel<true>, 0, &cache_sig)
                                     : resident_waves_cached(sh_vis_kernel<false>, 0, &cache_relu);
    const int n_wtiles = (int)cdiv(N, MCR_WAVE);
    const long long U = (long long)B * n_wtiles * C;
    const int W = (int)std::min<long long>(U, resident);
    MCR_REQUIRE(U / W < (1ll << 30), "mcr_sh_visibilities: problem too large");
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
