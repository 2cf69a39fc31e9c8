// Dev micro-benchmark: sustained VALU issue rate of one SIMD as a function of resident waves.  Each wave runs 16 independent
// v_fma_f32 (or v_pk_fma_f32) chains; reports shader cycles per wave-instruction per SIMD from the slowest wave of a CU.
// hipcc --offload-arch=gfx950 -O3 -w valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int PK>
__global__ __launch_bounds__(1024) void k(float* out, long long* cyc, int iters) {
    float v[16];
    f32x2 p[16];
    for (int i = 0; i < 16; ++i) { v[i] = threadIdx.x * 0.01f + i; p[i] = f32x2{v[i], v[i] + 1.f}; }
    const float m = 1.0001f, c = 0.5f;
    const f32x2 m2 = {m, m}, c2 = {c, c};
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                if (PK) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[j]) : "v"(m2), "v"(c2));
                else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j]) : "v"(m), "v"(c));
            }
    }
    const long long t1 = clock64();
    __syncthreads();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += v[i] + p[i][0] + p[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) atomicMax((unsigned long long*)cyc, (unsigned long long)(t1 - t0));   // the slowest wave (the oldest wave wins arbitration)
}

int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 2048 * 4); hipMalloc(&cyc, 64);
    const int iters = 2000;
    for (int pk = 0; pk < 2; ++pk)
        for (int wps : {1, 2, 3, 4, 6, 8}) {
            const int threads = 64 * 4 * wps;               // wps waves on each of the 4 SIMDs of a CU
            const int blocks = threads > 1024 ? 2 : 1;       // > 16 waves: two workgroups per CU
            const int tpb = threads / blocks;
            hipMemset(cyc, 0, 8);
            if (pk) hipLaunchKernelGGL(k<1>, dim3(256 * blocks), dim3(tpb), 0, 0, out, cyc, iters);
            else hipLaunchKernelGGL(k<0>, dim3(256 * blocks), dim3(tpb), 0, 0, out, cyc, iters);
            hipDeviceSynchronize();
            long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
            const double per_wave = (double)h / (iters * 64.0);
            printf("%s  %d waves/SIMD: %.2f cycles per instruction per wave -> %.2f cycles per instruction per SIMD\n",
                   pk ? "v_pk_fma_f32" : "v_fma_f32   ", wps, per_wave, per_wave / wps);
        }
    return 0;
}
