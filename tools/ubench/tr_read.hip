// Check of ds_read_b64_tr_b16 (gfx950): with per-lane address block(g) + (l & 15) * 8 bytes, lane l gets halves
// s[(l >> 4) * 64 + j * 16 + (l & 15)], j = 0..3 -- column (l & 15) of the 16-lane group's [4][16] row-major block: the B fragment
// (k = 4 g + j, n = l & 15) of v_mfma_f32_16x16x16_f16 when the block holds V[key 4g + j][col].  hipcc --offload-arch=gfx950 -O3 tr_read.hip -o tr_read
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out) {
    __shared__ __attribute__((aligned(16))) _Float16 s[256];
    for (int i = threadIdx.x; i < 256; i += 64) s[i] = (_Float16)(float)i;
    __syncthreads();
    const int l = threadIdx.x;
    typedef __attribute__((address_space(3))) s16x4* lp;
    const s16x4 vi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(s + (l >> 4) * 64 + (l & 15) * 4));
    const f16x4 v = __builtin_bit_cast(f16x4, vi);
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (float)v[j];
    // MFMA check: A = one-hot row selector: A[m][k] = (k == m) for m < 16 -> D[m][n] = B[k = m][n]
    f16x4 a;
    for (int j = 0; j < 4; ++j) a[j] = (_Float16)(((l >> 4) * 4 + j) == (l & 15) ? 1.f : 0.f);
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x16f16(a, v, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[256 + l * 4 + r] = c[r];      // D[row 4 g + r][col l & 15]
}
int main() {
    float* d; hipMalloc(&d, 512 * 4); float h[512];
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d); hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 4; ++j) {
            if (h[l * 4 + j] != (float)((l >> 4) * 64 + j * 16 + (l & 15))) ++bad;
            if (h[256 + l * 4 + j] != (float)(((l >> 4) * 4 + j) * 16 + (l & 15))) ++bad;     // D[m][n] = V[key m][col n] = s[m * 16 + n]
        }
    printf("lane 0: %g %g %g %g | lane 17: %g %g %g %g | mismatches %d\n", h[0], h[1], h[2], h[3], h[68], h[69], h[70], h[71], bad);
    return bad != 0;
}
