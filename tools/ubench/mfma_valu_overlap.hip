// Micro-benchmark: can the vector ALU and the matrix pipe of a gfx950 SIMD work at the same time?
//   mode 0: every wave issues only v_mfma_f32_32x32x16_bf16 (4 independent accumulators)
//   mode 1: every wave issues only plain VALU ops (and/sub/perm mix like the bf16x3 split), ILP 8
//   mode 2: blocks alternate: one MFMA block and one VALU block resident per CU (2 waves per SIMD, one of each kind)
//   mode 3: every wave interleaves both in one instruction stream (VPM VALU ops per MFMA)
// Build: hipcc --offload-arch=gfx950 -O3 mfma_valu_overlap.hip -o mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void mfma_body(f32x16 (&acc)[4], uint4 a, uint4 b) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[t], 0, 0, 0);
}
__device__ __forceinline__ void valu_body(float (&x)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float hi = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, x[i]) & 0xffff0000u);
        x[i] = x[i] - hi;                                   // 2 ops
        x[i] = __builtin_bit_cast(float, __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, x[i]), __builtin_bit_cast(unsigned, hi), 0x07060302u) | 0x3f800000u);  // 2 ops
    }
}

template <int MODE, int VPM>
__global__ __launch_bounds__(256, 2) void k(float* out, int iters, unsigned seed) {
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = 1.0f + threadIdx.x * 1e-3f + i;
    uint4 a = make_uint4(seed, seed * 3, seed * 5, seed * 7), b = make_uint4(seed * 11, seed * 13, seed * 17, seed * 19);
    const bool mfma_wave = MODE == 0 || (MODE == 2 && ((blockIdx.x >> 8) & 1) == 0);
    if (MODE == 3) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[t], 0, 0, 0);
#pragma unroll
                for (int v = 0; v < VPM / 4; ++v) {          // 4 ops per element
                    const int i = (t * (VPM / 4) + v) & 7;
                    const float hi = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, x[i]) & 0xffff0000u);
                    x[i] = x[i] - hi;
                    x[i] = __builtin_bit_cast(float, __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, x[i]), __builtin_bit_cast(unsigned, hi), 0x07060302u) | 0x3f800000u);
                }
            }
        }
    } else if (mfma_wave) {
        for (int it = 0; it < iters; ++it) mfma_body(acc, a, b);
    } else {
        for (int it = 0; it < iters * VPM / 8; ++it) valu_body(x);       // 32 ops per call; same op count as mode 3
    }
    float s = 0;
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int VPM>
float run(int blocks_per_cu, int iters) {
    float* out; hipMalloc(&out, 256 * 2 * 256 * sizeof(float));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int w = 0; w < 3; ++w) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE, VPM>), dim3(256 * blocks_per_cu), dim3(256), 0, 0, out, iters, 12345u);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    hipFree(out);
    return ms;
}

int main() {
    const int iters = 20000;      // 4 MFMAs (= 128 matrix-pipe cycles) per iteration
    printf("iters=%d: per wave %d MFMAs (%.2f Mcycles of matrix pipe); VALU waves issue VPM ops per MFMA-equivalent\n", iters, iters * 4, iters * 4 * 32 / 1e6);
    printf("mode0 MFMA only   1 wave/SIMD: %.3f ms   2 waves/SIMD: %.3f ms\n", run<0, 4>(1, iters), run<0, 4>(2, iters));
    printf("mode1 VALU only (VPM=4: 4 ops per MFMA slot)  1 wave/SIMD: %.3f ms   2 waves/SIMD: %.3f ms\n", run<1, 4>(1, iters), run<1, 4>(2, iters));
    printf("mode1 VALU only (VPM=8)  1 wave/SIMD: %.3f ms   2 waves/SIMD: %.3f ms\n", run<1, 8>(1, iters), run<1, 8>(2, iters));
    printf("mode2 one MFMA wave + one VALU wave per SIMD  VPM=4: %.3f ms   VPM=8: %.3f ms\n", run<2, 4>(2, iters), run<2, 8>(2, iters));
    printf("mode3 interleaved in one wave (1 wave/SIMD)  VPM=4: %.3f ms   VPM=8: %.3f ms\n", run<3, 4>(1, iters), run<3, 8>(1, iters));
    printf("mode3 interleaved, 2 waves/SIMD              VPM=4: %.3f ms   VPM=8: %.3f ms\n", run<3, 4>(2, iters), run<3, 8>(2, iters));
    return 0;
}
