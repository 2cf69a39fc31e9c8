// Dev micro-benchmark: what the fp16 matrix pipe sustains on this part when every SIMD issues v_mfma_f32_32x32x16_f16 back to back,
// as a function of the operand data (zeros / constant random / fresh random-like bits every MFMA).  Reports achieved PFLOP/s
// (wall clock) and the shader clock during the run.   hipcc --offload-arch=gfx950 -O3 -w mfma_power.hip -o mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// MODE 0: zero operands; 1: random operands, constant; 2: operands re-scrambled (xor with a running counter) before every MFMA
template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, long long* stamp, int iters, unsigned seed) {
    u32x4 a, b;
    unsigned s = seed ^ (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u);
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return MODE == 0 ? 0u : ((s >> 3) & 0x3bff3bffu); };   // |x| < 1: no overflow
    a = u32x4{rnd(), rnd(), rnd(), rnd()};
    b = u32x4{rnd(), rnd(), rnd(), rnd()};
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t)
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const long long t0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (MODE == 2) {                                   // cheap data refresh: rotate / xor lanes' bits (stays < 1 in magnitude)
                a.x = (a.x ^ (it * 0x01010101u)) & 0x3bff3bffu; b.y = (b.y ^ (it * 0x00110011u)) & 0x3bff3bffu;
            }
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc[t], 0, 0, 0);
        }
    }
    const long long t1 = clock64(), w1 = wall_clock64();
    float r = 0;
    for (int t = 0; t < 4; ++t)
        for (int q = 0; q < 16; ++q) r += acc[t][q];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 100) { stamp[0] = t1 - t0; stamp[1] = (w1 - w0) * 10; }
}

template <int MODE>
void run(const char* name, int wps) {
    float* out; long long* st;
    hipMalloc(&out, 256 * 2 * 512 * 4); hipMalloc(&st, 16);
    const int iters = 200000, threads = 64 * 4 * wps;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, st, 1000, 1u);      // warm-up
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, st, iters, 7u);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[2]; hipMemcpy(h, st, 16, hipMemcpyDeviceToHost);
    const double flops = 256.0 * 4 * wps * (double)iters * 4 * 32768.0;
    printf("%-34s %d waves/SIMD: %7.1f ms  %.2f PFLOP/s   shader clock %.2f GHz   %.1f cycles per MFMA per SIMD\n", name, wps, ms,
           flops / (ms * 1e-3) / 1e15, (double)h[0] / h[1], (double)h[0] / ((double)iters * 4 * wps));
    hipFree(out); hipFree(st);
}

int main() {
    run<0>("zero operands", 1); run<0>("zero operands", 2);
    run<1>("random operands (constant)", 1); run<1>("random operands (constant)", 2);
    run<2>("random operands (refreshed)", 1); run<2>("random operands (refreshed)", 2);
    return 0;
}
