// Shared device/host helpers for the MI355X (gfx950, wave64) kernels of the SCONE coverage-gain path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MCR_WAVE 64

namespace mcr {

// ---- error plumbing (C-ABI returns int; message kept per thread) -----------------------------------
void set_error(const char* fmt, ...);
int check_hip(hipError_t e, const char* what);

#define MCR_REQUIRE(cond, ...)                       \
    do {                                             \
        if (!(cond)) {                               \
            ::mcr::set_error(__VA_ARGS__);           \
            return 1;                                \
        }                                            \
    } while (0)

#define MCR_LAUNCH_CHECK(name)                                           \
    do {                                                                 \
        if (int _e = ::mcr::check_hip(hipGetLastError(), name)) return _e; \
    } while (0)

static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- wave64 DPP reductions ---------------------------------------------------------------------------
// After wave_sum_to_last(), lane 63 holds the sum of all 64 lanes (other lanes hold partial garbage).
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
__device__ __forceinline__ float dpp_mov0(float v) {
    // old = 0, bound_ctrl = true: lanes without a source read 0.
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK,
                                                                 BANK_MASK, true));
}

__device__ __forceinline__ float wave_sum_to_last(float v) {
    v += dpp_mov0<0x111>(v);             // row_shr:1
    v += dpp_mov0<0x112>(v);             // row_shr:2
    v += dpp_mov0<0x114>(v);             // row_shr:4
    v += dpp_mov0<0x118>(v);             // row_shr:8   -> lane 15 of every row = row sum
    v += dpp_mov0<0x142, 0xa>(v);        // row_bcast:15 into rows 1,3
    v += dpp_mov0<0x143, 0xc>(v);        // row_bcast:31 into rows 2,3 -> lane 63 = wave sum
    return v;
}

// The two cross-row steps as ONE instruction each: `v_add_f32_dpp v, v, v row_bcast:15 row_mask:0xa` adds the broadcast lane in the
// enabled rows and leaves the other rows' v alone, which is what `v += dpp_mov0<0x142, 0xa>(v)` means -- but hipcc compiles
// that to v_mov 0 + v_mov_dpp + v_add (it only folds full-mask DPP moves).  Same additions in the same order as
// wave_sum_to_last().  The s_nop's are the two wait states a DPP read needs after a vector write of its source (the hazard
// recogniser does not look inside inline asm).
__device__ __forceinline__ float wave_sum_cross_rows(float v) {
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf"
        : "+v"(v));
    return v;
}

// G independent reductions, step-major: the chains hide each other's DPP wait states.
template <int G>
__device__ __forceinline__ void wave_sum_to_last_multi(float (&v)[G]) {
#pragma unroll
    for (int g = 0; g < G; ++g) v[g] += dpp_mov0<0x111>(v[g]);
#pragma unroll
    for (int g = 0; g < G; ++g) v[g] += dpp_mov0<0x112>(v[g]);
#pragma unroll
    for (int g = 0; g < G; ++g) v[g] += dpp_mov0<0x114>(v[g]);
#pragma unroll
    for (int g = 0; g < G; ++g) v[g] += dpp_mov0<0x118>(v[g]);
#pragma unroll
    for (int g = 0; g < G; ++g) v[g] = wave_sum_cross_rows(v[g]);
}

__device__ __forceinline__ float wave_max_all(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

__device__ __forceinline__ float wave_sum_all(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

}  // namespace mcr
