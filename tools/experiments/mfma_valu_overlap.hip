// Dev micro-benchmark: does VALU work overlap with the matrix pipe (a) inside one wave's instruction stream, (b) between two waves
// that share a SIMD?  hipcc --offload-arch=gfx950 -O3 mfma_valu_overlap.hip -o mfma_valu_overlap && ./mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// mode 0: every wave runs {MFMA; K independent v_fma}; mode 1: waves 0-3 MFMA only, waves 4-7 VALU only (2 waves / SIMD)
template <int K, int MODE, int NM>
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, int iters) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    f32x16 acc0 = {0}, acc1 = {0};
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 0.01f + i;
    const float m = 1.0001f, c = 0.5f;
    const bool do_m = MODE == 0 || wave < 4, do_v = MODE == 0 || wave >= 4;
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (do_m) {
#pragma unroll
            for (int q = 0; q < NM; ++q) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
                asm volatile("" : "+v"(acc0));
                if (do_v) {
#pragma unroll
                    for (int j = 0; j < K; ++j) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j % 16]) : "v"(m), "v"(c)); }
                }
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc1, 0, 0, 0);
                asm volatile("" : "+v"(acc1));
                if (do_v) {
#pragma unroll
                    for (int j = 0; j < K; ++j) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j % 16]) : "v"(m), "v"(c)); }
                }
            }
        } else if (do_v) {
#pragma unroll
            for (int q = 0; q < 2 * NM; ++q)
#pragma unroll
                for (int j = 0; j < K; ++j) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j % 16]) : "v"(m), "v"(c)); }
        }
    }
    const long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += v[i] + acc0[i] + acc1[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
}

template <int K, int MODE>
void run(const char* name, int threads) {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 64);
    hipMemset(cyc, 0, 64);
    const int iters = 2000, NM = 4;
    hipLaunchKernelGGL((k<K, MODE, NM>), dim3(256), dim3(threads), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    long long h[8]; hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    const double per = 1.0 / (iters * 2.0 * NM);
    printf("%-34s K=%2d  cycles per (MFMA + K fma) slot: wave0 %.1f  wave4 %.1f\n", name, K, h[0] * per, h[4] * per);
    hipFree(out); hipFree(cyc);
}

int main() {
    printf("one wave per SIMD, MFMA and K v_fma_f32 interleaved in the same wave\n");
    run<0, 0>("same wave", 256); run<4, 0>("same wave", 256); run<8, 0>("same wave", 256); run<12, 0>("same wave", 256);
    run<16, 0>("same wave", 256); run<24, 0>("same wave", 256); run<32, 0>("same wave", 256);
    printf("two waves per SIMD, both run MFMA + K fma\n");
    run<0, 0>("2 waves/SIMD same mix", 512); run<8, 0>("2 waves/SIMD same mix", 512); run<16, 0>("2 waves/SIMD same mix", 512);
    printf("two waves per SIMD: waves 0-3 MFMA only, waves 4-7 K fma per slot only\n");
    run<4, 1>("split roles", 512); run<8, 1>("split roles", 512); run<16, 1>("split roles", 512); run<32, 1>("split roles", 512);
    return 0;
}
