// Micro-benchmark: pure-compute rate of sh_dot (no memory traffic), scalar-FMA form vs the packed form (-DSC_PK=1), at several
// occupancies (and -DNC=2/3: that many cameras side by side).  Build twice: hipcc --offload-arch=gfx950 -O3 -ffp-contract=fast [-DSC_PK=1] sh_dot_pk_rate.hip -o sh_dot_pk_rate[_pk]
#include "../../macarons_amd/csrc/sh_scorer.hip"
#include "../../macarons_amd/csrc/errors.hip"
#include <stdio.h>
using namespace mcr;

// The packed form: the (U_m, V_m) Horner pair of an order is ONE v_pk_fma_f32 chain, the power recurrence a packed complex
// multiply, the combination a packed accumulator (33 v_pk_fma_f32 + 7 v_pk_mul_f32 + 18 scalar instructions).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float sh_dot_pk(float dx, float dy, float dz, const float (&a)[64]) {
    const float r2 = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
    const float ir = __builtin_amdgcn_rsqf(r2);
    const float nx = dx * ir, ct = dy * ir, nz = dz * ir;
    const f32x2 ct2 = {ct, ct}, nz2 = {nz, nz}, nxm = {-nx, nx};
    float z = a[shk(7, 0)];
#pragma unroll
    for (int l = 6; l >= 0; --l) z = fmaf(ct, z, a[shk(l, 0)]);
    f32x2 w = {nz, nx};                         // (Re, Im) (n_z + i n_x)^m
    f32x2 acc = {0.f, 0.f};
#pragma unroll
    for (int m = 1; m < 8; ++m) {
        f32x2 UV = {a[shk(7, m)], a[shk(7, -m)]};
#pragma unroll
        for (int l = 6; l >= m; --l) UV = __builtin_elementwise_fma(ct2, UV, (f32x2){a[shk(l, m)], a[shk(l, -m)]});
        acc = m == 1 ? w * UV : __builtin_elementwise_fma(w, UV, acc);
        if (m < 7) {
            const f32x2 ws = __builtin_shufflevector(w, w, 1, 0);      // (Im, Re)
            w = __builtin_elementwise_fma(nz2, w, nxm * ws);            // (nz Re - nx Im, nz Im + nx Re)
        }
    }
    return z + (acc.x + acc.y);
}

// The powers form: every polynomial as a sum over precomputed powers of cos(polar) so that all but one multiply-add of the stream is
// a two-address v_fmac_f32 / v_mul_f32 (32-bit VOP2 encoding); Horner needs the three-address v_fma_f32 (64-bit VOP3 encoding).
__device__ __forceinline__ float sh_dot_pw(float dx, float dy, float dz, const float (&a)[64]) {
    const float r2 = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
    const float ir = __builtin_amdgcn_rsqf(r2);
    const float nx = dx * ir, ct = dy * ir, nz = dz * ir;
    float xp[8];
    xp[1] = ct; xp[2] = ct * ct; xp[3] = xp[2] * ct; xp[4] = xp[2] * xp[2]; xp[5] = xp[3] * xp[2]; xp[6] = xp[3] * xp[3]; xp[7] = xp[4] * xp[3];
    float z = fmaf(ct, a[shk(1, 0)], a[shk(0, 0)]);
#pragma unroll
    for (int k = 2; k < 8; ++k) z = fmaf(a[shk(k, 0)], xp[k], z);
    float mnx = -nx;
    asm volatile("" : "+v"(mnx));            // a value of its own: folded into the multiply as a modifier it needs the 64-bit encoding
    float cm = nz, sm = nx;
#pragma unroll
    for (int m = 1; m < 8; ++m) {
        if (m < 7) {
            float U = a[shk(m + 1, m)] * xp[1], V = a[shk(m + 1, -m)] * xp[1];
#pragma unroll
            for (int k = 2; k + m < 8; ++k) {
                U = fmaf(a[shk(m + k, m)], xp[k], U);
                V = fmaf(a[shk(m + k, -m)], xp[k], V);
            }
            z = fmaf(cm, U, z);
            z = fmaf(sm, V, z);
        }
        z = fmaf(cm, a[shk(m, m)], z);
        z = fmaf(sm, a[shk(m, -m)], z);
        if (m < 7) {
            float cn = mnx * sm, sn = nx * cm;
            cn = fmaf(nz, cm, cn); sn = fmaf(nz, sm, sn);
            cm = cn; sm = sn;
        }
    }
    return z;
}
#if defined(SC_PW)
#define SH_DOT sh_dot_pw
#elif defined(SC_PK)
#define SH_DOT sh_dot_pk
#else
#define SH_DOT sh_dot
#endif

#ifndef NC
#define NC 1              // cameras evaluated side by side per iteration (independent dot products for the scheduler to interleave)
#endif
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    float hs[64];
    for (int i = 0; i < 64; ++i) hs[i] = seed * (i + 1) + threadIdx.x * 1e-4f;
#pragma unroll
    for (int i = 0; i < 64; ++i) asm volatile("" : "+v"(hs[i]));
    float acc = 0.f;
    const float px = threadIdx.x * 1e-3f, py = 0.1f, pz = -0.2f;
    for (int it = 0; it < iters; it += NC) {
        float z[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) z[c] = SH_DOT(1.5f + (it + c) * 1e-3f - px, 0.3f + c - py, 1.f - pz + (it + c) * 1e-4f, hs);
#pragma unroll
        for (int c = 0; c < NC; ++c) asm volatile("" : "+v"(z[c]));
#pragma unroll
        for (int c = 0; c < NC; ++c) acc += z[c];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

void run(int blocks_per_cu) {
    float* out; (void)hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    int iters = 4000, blocks = 256 * blocks_per_cu;
    size_t lds = blocks_per_cu >= 8 ? 0 : (160 * 1024 / blocks_per_cu - 1024);   // force residency = blocks_per_cu
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int w = 0; w < 3; ++w) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), lds, 0, out, iters, 0.01f);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    double pairs = (double)blocks * 256 * iters;
    double cyc = ms * 1e-3 * 2.4e9 / ((double)blocks * 4 * iters / 1024.0);
    printf("waves/SIMD=%d: %.3f ms  %.1f Gpairs/s  %.1f cycles per pair per SIMD @2.4GHz\n", blocks_per_cu, ms, pairs / ms * 1e-6, cyc);
    (void)hipFree(out);
}

int main() {
#if defined(SC_PW)
    printf("powers form, NC=%d\n", NC);
#elif defined(SC_PK)
    printf("packed form, NC=%d\n", NC);
#else
    printf("scalar form, NC=%d\n", NC);
#endif
    for (int bpc : {1, 2, 3, 4, 5, 6}) run(bpc);
    return 0;
}
