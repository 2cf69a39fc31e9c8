// Check: the v_fma_mixlo/hi_f16 form of the two-term fp16 split (3 instructions per value pair) is bit-identical to the
// convert / subtract / convert form (6 per pair) on every fp32 bit pattern class.  hipcc --offload-arch=gfx950 -O3 split_mix.hip -o split_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_ref(const float a, const float b, unsigned& hi, unsigned& lo) {
    const f32x2 x = {a, b};
    const f16x2 h = __builtin_convertvector(x, f16x2);
    const f32x2 r = x - __builtin_convertvector(h, f32x2);
    const f16x2 l = __builtin_convertvector(r, f16x2);
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, l);
}
__device__ __forceinline__ void split_mix(const float a, const float b, unsigned& hi, unsigned& lo) {
    const f32x2 x = {a, b};
    const f16x2 h = __builtin_convertvector(x, f16x2);
    hi = __builtin_bit_cast(unsigned, h);
    unsigned l;
    asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(l) : "v"(a), "v"(hi));
    asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(b), "v"(hi));
    lo = l;
}
__global__ void k(const unsigned* bits, unsigned long long* bad, unsigned* first, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float a = __builtin_bit_cast(float, bits[2 * i]), b = __builtin_bit_cast(float, bits[2 * i + 1]);
    unsigned h0, l0, h1, l1;
    split_ref(a, b, h0, l0);
    split_mix(a, b, h1, l1);
    // NaN payloads may differ: compare NaN-ness for the halves that are NaN
    auto same = [](unsigned x, unsigned y) {
        for (int s = 0; s < 32; s += 16) {
            const unsigned p = (x >> s) & 0xffff, q = (y >> s) & 0xffff;
            const bool pn = (p & 0x7c00) == 0x7c00 && (p & 0x3ff), qn = (q & 0x7c00) == 0x7c00 && (q & 0x3ff);
            if (pn || qn) { if (pn != qn) return false; } else if (p != q) return false;
        }
        return true;
    };
    if (!same(h0, h1) || !same(l0, l1)) {
        if (atomicAdd(bad, 1ull) == 0) { first[0] = bits[2 * i]; first[1] = bits[2 * i + 1]; first[2] = l0; first[3] = l1; }
    }
}
int main() {
    const long long n = 1ll << 26;
    std::vector<unsigned> h(2 * n);
    unsigned long long s = 88172645463325252ull;
    for (long long i = 0; i < 2 * n; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        unsigned v = (unsigned)(s >> 16);
        const int cls = (int)(s & 7);
        if (cls == 0) v = (v & 0x807fffffu) | ((unsigned)(100 + (s >> 40) % 40) << 23);      // around 1: exponents 100..139
        else if (cls == 1) v = (v & 0x807fffffu) | ((unsigned)(86 + (s >> 40) % 30) << 23);  // fp16 subnormal range of hi / lo
        else if (cls == 2) v = (v & 0x807fffffu) | ((unsigned)(140 + (s >> 40) % 6) << 23);  // near the fp16 maximum and beyond
        h[i] = v;                                                                             // else: any bit pattern (inf / NaN / fp32 denormals)
    }
    unsigned *d; unsigned long long* bad; unsigned* first;
    hipMalloc(&d, 2 * n * 4); hipMalloc(&bad, 8); hipMalloc(&first, 16);
    hipMemcpy(d, h.data(), 2 * n * 4, hipMemcpyHostToDevice); hipMemset(bad, 0, 8);
    hipLaunchKernelGGL(k, dim3((unsigned)(n / 256)), dim3(256), 0, 0, d, bad, first, n);
    unsigned long long nb; unsigned f[4];
    hipMemcpy(&nb, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(f, first, 16, hipMemcpyDeviceToHost);
    printf("pairs %lld  mismatches %llu", n, nb);
    if (nb) printf("  first: a=%08x b=%08x lo_ref=%08x lo_mix=%08x", f[0], f[1], f[2], f[3]);
    printf("\n");
    return nb != 0;
}
