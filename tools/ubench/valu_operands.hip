// Micro-benchmark: v_fma/v_mul issue rate vs number of VGPR source operands (gfx950).
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    constexpr int ILP = 8;
    float x[ILP], y[ILP], z[ILP];
    for (int i = 0; i < ILP; ++i) { x[i] = threadIdx.x * 1e-3f + i; y[i] = 1.f - x[i] * 1e-4f; z[i] = x[i] * 1e-5f; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < ILP; ++i) {
                if (MODE == 0) x[i] = fmaf(x[i], a, b);                 // 1 VGPR src
                if (MODE == 1) x[i] = fmaf(x[i], y[i], b);              // 2 VGPR src
                if (MODE == 2) x[i] = fmaf(y[i], z[i], x[i]);           // 3 VGPR src (v_fmac)
                if (MODE == 3) x[i] = fmaf(y[(i + 1) % ILP], z[(i + 3) % ILP], x[i]);   // 3 VGPR, shuffled banks
                if (MODE == 4) x[i] = x[i] * y[i];                      // v_mul 2 VGPR
                if (MODE == 5) x[i] = x[i] * a;                         // v_mul 1 VGPR
            }
    }
    float s = 0;
    for (int i = 0; i < ILP; ++i) s += x[i] + y[i] + z[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name) {
    float* out; (void)hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    int iters = 4000, blocks = 256 * 5;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int w = 0; w < 3; ++w) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, out, iters, 0.999f, 0.001f);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    double instr = (double)blocks * 4 * iters * 64.0;
    printf("%-28s %.3f ms  %.2f cyc/instr/SIMD(@2.4GHz)\n", name, ms, ms * 1e-3 * 2.4e9 / (instr / 1024.0));
    (void)hipFree(out);
}

int main() {
    run<0>("fma 1 VGPR src"); run<1>("fma 2 VGPR src"); run<2>("fmac 3 VGPR src");
    run<3>("fmac 3 VGPR shuffled"); run<4>("mul 2 VGPR"); run<5>("mul 1 VGPR");
    return 0;
}
