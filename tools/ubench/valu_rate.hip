// Micro-benchmark: issue rate of v_fma_f32 vs v_pk_fma_f32 on gfx950 at several occupancies / ILP.
// READ WITH CARE: at ILP >= 4 hipcc SLP-packs the independent scalar FMAs of the "fma" rows into v_pk_fma_f32 (check the ISA):
// only the ILP = 1 "fma" rows are scalar instructions.
// Build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int ILP, bool PK>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    float x[ILP]; f32x2 y[ILP];
    for (int i = 0; i < ILP; ++i) { x[i] = threadIdx.x * 1e-3f + i; y[i] = (f32x2)(x[i]); }
    f32x2 av = {a, a * 1.0001f}, bv = {b, b * 1.0001f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < ILP; ++i) {
                if (PK) y[i] = __builtin_elementwise_fma(y[i], av, bv);
                else x[i] = fmaf(x[i], a, b);
            }
    }
    float s = 0;
    for (int i = 0; i < ILP; ++i) s += PK ? (y[i].x + y[i].y) : x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int ILP, bool PK>
void run(int blocks_per_cu, size_t lds) {
    float* out; hipMalloc(&out, 256 * 256 * 64 * sizeof(float));
    int iters = 4000, blocks = 256 * blocks_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<ILP, PK>), dim3(blocks), dim3(256), lds, 0, out, iters, 0.999f, 0.001f);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double instr = (double)blocks * 4 /*waves*/ * iters * 8.0 * ILP;
    double flops = instr * 64 * 2 * (PK ? 2 : 1);
    // cycles per wave-instruction per SIMD assuming 2.4 GHz and blocks spread evenly
    double cyc = ms * 1e-3 * 2.4e9 / (instr / 1024.0);
    printf("%s ILP=%d blocks/CU=%d: %.3f ms  %.1f TFLOP/s  %.2f cyc/instr/SIMD(@2.4GHz)\n", PK ? "pk_fma" : "fma   ", ILP,
           blocks_per_cu, ms, flops / ms * 1e-9, cyc);
    hipFree(out);
}

int main() {
    for (int bpc : {1, 2, 4, 8}) {
        run<1, false>(bpc, 0); run<4, false>(bpc, 0); run<8, false>(bpc, 0);
        run<1, true>(bpc, 0); run<4, true>(bpc, 0); run<8, true>(bpc, 0);
    }
    return 0;
}
