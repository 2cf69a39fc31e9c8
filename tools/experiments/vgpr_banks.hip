// Dev micro-benchmark: does a v_fma_f32 whose three source VGPRs share a register bank (index mod 4) issue slower than one whose
// sources sit in three different banks?  Four resident waves per SIMD (VALU saturated), explicit registers.
// hipcc --offload-arch=gfx950 -O3 -w vgpr_banks.hip -o vgpr_banks
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, long long* cyc, int iters) {
    float r = threadIdx.x * 0.001f;
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0)       // sources v8, v13, v18 (banks 0, 1, 2); destinations rotate
            asm volatile("v_fma_f32 v20, v8, v13, v18\n v_fma_f32 v21, v9, v14, v19\n v_fma_f32 v22, v10, v15, v16\n v_fma_f32 v23, v11, v12, v17\n"
                         "v_fma_f32 v24, v8, v13, v18\n v_fma_f32 v25, v9, v14, v19\n v_fma_f32 v26, v10, v15, v16\n v_fma_f32 v27, v11, v12, v17\n"
                         ::: "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27");
        else if (MODE == 1)  // two sources in the same bank: v8, v12 (bank 0), v17
            asm volatile("v_fma_f32 v20, v8, v12, v17\n v_fma_f32 v21, v9, v13, v18\n v_fma_f32 v22, v10, v14, v19\n v_fma_f32 v23, v11, v15, v16\n"
                         "v_fma_f32 v24, v8, v12, v17\n v_fma_f32 v25, v9, v13, v18\n v_fma_f32 v26, v10, v14, v19\n v_fma_f32 v27, v11, v15, v16\n"
                         ::: "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27");
        else                 // all three sources in the same bank: v8, v12, v16
            asm volatile("v_fma_f32 v20, v8, v12, v16\n v_fma_f32 v21, v9, v13, v17\n v_fma_f32 v22, v10, v14, v18\n v_fma_f32 v23, v11, v15, v19\n"
                         "v_fma_f32 v24, v8, v12, v16\n v_fma_f32 v25, v9, v13, v17\n v_fma_f32 v26, v10, v14, v18\n v_fma_f32 v27, v11, v15, v19\n"
                         ::: "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27");
    }
    const long long t1 = clock64();
    __syncthreads();
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) atomicMax((unsigned long long*)cyc, (unsigned long long)(t1 - t0));
}
template <int MODE> void run(const char* name) {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 8); hipMemset(cyc, 0, 8);
    const int iters = 20000;
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-44s %.2f cycles per instruction per SIMD (4 waves)\n", name, (double)h / (iters * 8.0) / 4.0);
}
int main() {
    run<0>("three source banks"); run<1>("two sources in one bank"); run<2>("three sources in one bank");
    return 0;
}
