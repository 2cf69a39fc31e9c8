// Micro-benchmark: how many 256-thread blocks are co-resident per CU as a function of LDS bytes per block.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(256) void spin(float* out, long long cycles) {
    extern __shared__ float lds[];
    lds[threadIdx.x] = threadIdx.x;
    long long t0 = clock64();
    while (clock64() - t0 < cycles) { }
    out[blockIdx.x * 256 + threadIdx.x] = lds[(threadIdx.x + 1) & 255];
}
int main() {
    float* out; (void)hipMalloc(&out, 256 * 16 * 256 * 4);
    (void)hipFuncSetAttribute((const void*)spin, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int kb : {8, 16, 24, 32, 36, 40, 48, 54, 56, 64, 72, 80, 96, 128, 160}) {
        const int blocks = 256 * 8;
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        float ms = 0;
        for (int w = 0; w < 2; ++w) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(spin, dim3(blocks), dim3(256), kb * 1024, 0, out, 200000LL);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
        }
        // one round = 200000 cycles of the 100 MHz s_memtime clock?  report rounds relative to the 8 KB case
        static float base = 0; if (!base) base = ms;
        printf("LDS %3d KB/block: %.3f ms  -> rounds x%.2f  => ~%.1f blocks/CU resident\n", kb, ms, ms / base, 8.0 / (ms / base));
    }
    return 0;
}
