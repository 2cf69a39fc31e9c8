// Micro-benchmark: pure-compute rate of sh_dot<NC> (no memory traffic) at several occupancies.
#include "../../macarons_amd/csrc/sh_scorer.hip"
#include "../../macarons_amd/csrc/errors.hip"
#include <stdio.h>
using namespace mcr;

template <int NC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    extern __shared__ float dummy[];
    float hs[64];
    for (int i = 0; i < 64; ++i) hs[i] = seed * (i + 1) + threadIdx.x * 1e-4f;
    float acc = 0.f;
    float px = threadIdx.x * 1e-3f, py = 0.1f, pz = -0.2f;
    for (int it = 0; it < iters; ++it) {
        float dx[NC], dy[NC], dz[NC], z[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) { dx[c] = 1.5f + it * 1e-3f + c - px; dy[c] = 0.3f * c - py; dz[c] = 1.f - pz + it * 1e-4f; }
        sh_dot<NC>(dx, dy, dz, hs, z);
#pragma unroll
        for (int c = 0; c < NC; ++c) asm volatile("" : "+v"(z[c]));
#pragma unroll
        for (int c = 0; c < NC; ++c) acc += z[c];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int NC>
void run(int blocks_per_cu) {
    float* out; (void)hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    int iters = 2000 / NC, blocks = 256 * blocks_per_cu;
    size_t lds = blocks_per_cu >= 8 ? 0 : (160 * 1024 / blocks_per_cu - 1024);   // force residency = blocks_per_cu
    (void)hipFuncSetAttribute((const void*)k<NC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int w = 0; w < 3; ++w) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<NC>), dim3(blocks), dim3(256), lds, 0, out, iters, 0.01f);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    double pairs = (double)blocks * 256 * iters * NC;
    printf("NC=%d waves/SIMD=%d: %.3f ms  %.1f Gpairs/s  (%.1f algorithmic TFLOP/s @370)\n", NC, blocks_per_cu, ms,
           pairs / ms * 1e-6, pairs * 370 / ms * 1e-9);
    (void)hipFree(out);
}

int main() {
    for (int bpc : {1, 2, 3, 4, 5}) { run<1>(bpc); run<2>(bpc); run<3>(bpc); run<4>(bpc); }
    return 0;
}
