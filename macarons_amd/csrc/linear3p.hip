// K4-planes — nn.Linear on the two-term fp16 split with BOTH operands already stored as fp16 hi/lo planes in HBM (the SconeOcc
// head, SconeOcc.py:320-347, variant 6): three MFMAs per fp32 product like linear3h.hip, but nothing is split inside the GEMM.
//
// linear3h re-split every activation tile in each of its N/128 column blocks (4x for the 1344 -> 512 layer) and staged both
// operands through registers; here every activation is split ONCE where it is produced (the local transformers' pooled
// features, the epilogues of the previous layers) and both operands travel HBM -> LDS by global_load_lds_dwordx4: no staging
// registers, no ds_write pass, no vector work in the K loop besides the MFMAs' fragment reads.
//
// Operands: X planes Xh / Xl [M][ldx] fp16 (row-major), W planes Wh / Wl [N][ldw] fp16 (W times a power of two 2^e, built by
// networks/packing.py: pack_head_planes, or by split_weights_kernel).  Evaluated TRANSPOSED, D^T[n][m] = W[n][:] . X[m][:] (W
// tile = MFMA A operand), so a lane ends up with 4 CONSECUTIVE features n of one row m per register quad: the epilogue adds the
// bias (and the per-group row bias), applies the exact-erf GELU and stores either fp32 (float4 per quad) or, split again, the
// next layer's planes (8 bytes per plane and quad).
//
// Block = 8 waves = 128 features x 256 rows (wave w: features 32 (w & 3).., rows 128 (w >> 2)..), K in chunks of 32, THREE LDS
// stages of 48 KB (Xh, Xl: 256 rows; Wh, Wl: 128 rows) = 144 KB, one block per CU.  The DMA runs two chunks ahead: iteration k waits
// (counted: s_waitcnt vmcnt(6) = the six DMA instructions of chunk k+1 may stay in flight), passes ONE raw s_barrier, queues
// chunk k+2 into the stage chunk k-1 just left, and multiplies chunk k -- a __syncthreads() would drain the DMA queue at every
// barrier (hipcc puts vmcnt(0) in front of it) and the 128 x 128 / two-stage form of this kernel spent half its time waiting
// for L2 round trips (616 TFLOP/s executed on the 1344 -> 512 layer).  The DMA's LDS image is lane-linear, so the XOR swizzle of
// the 16-byte chunks (by (row >> 2) & 3: conflict-free ds_read_b128 fragment reads) is applied on the SOURCE address: LDS
// position (row, c) receives the row's chunk c ^ ((row >> 2) & 3).
#include "lp_split.h"

namespace mcr {

// GELU in the epilogues: l3_gelu (lp_split.h) -- the exact-erf GELU with erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, one rcp,
// one exp2, seven fma: 14 instructions), the function the fused local transformer of this numerics variant already applies.  libm's
// branchy erff was ~45 vector instructions per value with both branches executed: 43 of the 100 us of an encoder's FF1 at 30 x 2048
// tokens were its epilogue.
constexpr int LP_BK = 32;                                 // k per chunk
// output modes: fp32 rows, fp16 hi/lo planes (128 features x 256 rows per block), or LP_DOT: 256 features x 128 rows per block --
// the block then owns ALL 256 outputs of its rows and the epilogue reduces them against a vector: out[m] = act2(act(y[m][:]) . v +
// c), the last two layers of the SconeOcc head (512 -> 256 -> 1) in one launch: the 256-wide activations (102 MB at T = 100k) are
// neither written nor read back, and the 256 -> 1 layer costs no launch of its own.
constexpr int LP_F32 = 0, LP_PLANES = 1, LP_DOT = 2;
constexpr int LP_STAGE = 2 * 256 * 4 + 2 * 128 * 4;       // chunks per stage (Xh | Xl | Wh | Wl) = 3072 = 48 KB in either tile shape
constexpr int LP_STAGES = 3;
constexpr int LP_LDS_BYTES = LP_STAGES * LP_STAGE * 16;   // 147 456
// SMALL form for short K (the encoders' K = 128 ... 512, eight to sixteen chunks): 4 waves = 128 features x 128 rows, TWO stages of
// 32 KB = 64 KB, two blocks per CU -- with so few chunks a block that owns its CU cannot hide its DMA prologue and its GELU / split
// epilogue behind anything; two resident blocks cover each other.  Same products in the same order per output element: the bits do
// not depend on the form.
constexpr int LPS_STAGE = 2 * 128 * 4 + 2 * 128 * 4;      // 2048 chunks = 32 KB
constexpr int LPS_STAGES = 2;
constexpr int LPS_LDS_BYTES = LPS_STAGES * LPS_STAGE * 16;   // 65 536

typedef const __attribute__((address_space(1))) void* lp_gptr;
typedef __attribute__((address_space(3))) void* lp_lptr;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int OFF>
__device__ __forceinline__ u32x4 lds_read(unsigned addr) {           // ds_read_b128 the compiler's wait-count pass does not see
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
__device__ __forceinline__ f32x16 mfma_u(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// NP = 2: hi / lo planes, three MFMAs per product (variant 6).  NP = 1 (variant 7, the opt-in 16-bit matrix path): the HIGH planes
// alone -- one MFMA per product, half the DMA bytes and fragment reads; Xl / Wl / Yl are never touched (the LDS image keeps the
// two-plane stage layout, its low halves stay unused) and a planes epilogue rounds once (v_cvt_pk_f16_f32) instead of splitting.
template <int MODE, bool SMALL = false, int NP = 2>
__global__ __launch_bounds__(SMALL ? 256 : 512, SMALL ? 2 : 1) void linear3p_kernel(const _Float16* __restrict__ Xh, const _Float16* __restrict__ Xl, long long ldx,
                                                          const _Float16* __restrict__ Wh, const _Float16* __restrict__ Wl, long long ldw,
                                                          const float* __restrict__ bias, const float* __restrict__ row_bias,
                                                          long long rows_per_group, const int* __restrict__ row_group,
                                                          float* __restrict__ Y, _Float16* __restrict__ Yh, _Float16* __restrict__ Yl,
                                                          long long ldy, long long M, int N, int K, int act, float wscale_inv,
                                                          const float* __restrict__ dot_v, const float* __restrict__ dot_c, int act2,
                                                          const float* __restrict__ R, long long ldr) {
    static_assert(!(SMALL && MODE == LP_DOT), "the dot form keeps the large tile");
    constexpr int NW = SMALL ? 4 : 8;                                                         // waves per block
    constexpr int LP_TN = MODE == LP_DOT ? 256 : 128, LP_TM = (MODE == LP_DOT || SMALL) ? 128 : 256;   // features / activation rows per block
    constexpr int LP_XC = LP_TM * 4, LP_WC = LP_TN * 4;                                       // 16-byte chunks per X / W tile and plane
    constexpr int GX = LP_XC / (NW * 64), GW = LP_WC / (NW * 64);                             // DMA chunk groups (64 chunks) per wave and plane
    constexpr int STAGE = 2 * LP_XC + 2 * LP_WC, STAGES = SMALL ? LPS_STAGES : LP_STAGES;     // chunks per stage; stages
    constexpr int PER = NP * (GX + GW);                                                       // DMA instructions per wave and chunk
    constexpr bool PLANES_OUT = MODE == LP_PLANES;
    extern __shared__ __attribute__((aligned(16))) uint4 S[];                     // [stage][Xh | Xl | Wh | Wl]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = MODE == LP_DOT ? wave : (wave & 3), wm = (MODE == LP_DOT || SMALL) ? 0 : (wave >> 2);   // the wave's 32 features x 128 rows
    // XCD-aware block order: workgroup b runs on XCD b % 8 (its own L2).  The N / 128 column blocks of one 256-row block read the
    // SAME activation rows: they get consecutive slots of ONE XCD, so the rows come from HBM once and from that L2 afterwards
    // (with the plain (row block, column block) grid the 1344 -> 512 layer fetched its 537 MB of activations four times).
    const int ncb = (N + LP_TN - 1) / LP_TN;
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const long long rb = (long long)(q / ncb) * 8 + xcd;
    const long long m0 = rb * LP_TM;
    if (m0 >= M) return;                                   // (the grid is padded to whole groups of 8 row blocks)
    const int n0 = (q % ncb) * LP_TN;
    const int i = lane & 31, h = lane >> 5;

    // ---- DMA addressing: per stage this wave moves X chunk groups 2 wave, 2 wave + 1 (64 chunks = 16 rows each) of both X planes and
    // W chunk group `wave` of both W planes.  LDS position p = group * 64 + lane = (row = p >> 2, c = p & 3) takes the source
    // chunk c ^ ((row >> 2) & 3) of that row.
    const _Float16 *sx[GX][2], *sw[GW][2];                 // this lane's sources at k = 0: [group][plane]
#pragma unroll
    for (int g = 0; g < GX; ++g) {
        const int p = (GX * wave + g) * 64 + lane, row = p >> 2, c = (p & 3) ^ ((row >> 2) & 3);
        const long long xr = min(m0 + row, M - 1);         // rows beyond the matrix repeat its last row (results discarded)
        sx[g][0] = Xh + xr * ldx + c * 8; sx[g][1] = NP == 2 ? Xl + xr * ldx + c * 8 : sx[g][0];     // (NP == 1: Xl may be NULL, never formed)
    }
#pragma unroll
    for (int g = 0; g < GW; ++g) {
        const int p = (GW * wave + g) * 64 + lane, row = p >> 2, c = (p & 3) ^ ((row >> 2) & 3);
        const long long wr = min((long long)n0 + row, (long long)N - 1);
        sw[g][0] = Wh + wr * ldw + c * 8; sw[g][1] = NP == 2 ? Wl + wr * ldw + c * 8 : sw[g][0];
    }
    auto stage = [&](int st, int k0) {                     // 6 DMA instructions per wave
        uint4* b = S + st * STAGE;
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) {
#pragma unroll
            for (int g = 0; g < GX; ++g)
                __builtin_amdgcn_global_load_lds((lp_gptr)(sx[g][pl] + k0), (lp_lptr)(b + pl * LP_XC + (GX * wave + g) * 64), 16, 0, 0);
#pragma unroll
            for (int g = 0; g < GW; ++g)
                __builtin_amdgcn_global_load_lds((lp_gptr)(sw[g][pl] + k0), (lp_lptr)(b + 2 * LP_XC + pl * LP_WC + (GW * wave + g) * 64), 16, 0, 0);
        }
    };

    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // Fragment reads are inline-asm ds_read_b128 with hand-counted lgkmcnt waits: hipcc cannot prove that a C++ LDS read of stage
    // kc does not alias the DMA it has just queued into stage kc+2 (run-time stage indices) and puts s_waitcnt vmcnt(0) in front of
    // the first read of every chunk -- which drains the whole DMA pipeline (measured: 2.8 us per chunk against 0.7 us of MFMAs).
    // Byte addresses: row r of a tile, k16-step s, lane half h -> ((r * 4 + ((2 s + h) ^ ((r >> 2) & 3))) * 16; the tile / plane
    // offsets are immediates.  ((32 t + i) >> 2) & 3 == (i >> 2) & 3 for every 32-row tile.
    const int key = (i >> 2) & 3;
    const unsigned lds0 = (unsigned)(size_t)((lp_lptr)S);
    unsigned ax[2], aw[2];
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_) {
        const int cs = (2 * s_ + h) ^ key;
        ax[s_] = lds0 + (unsigned)(((wm * 128 + i) * 4 + cs) * 16);
        aw[s_] = lds0 + (unsigned)((2 * LP_XC + (wn * 32 + i) * 4 + cs) * 16);
    }
    const int n_chunks = K / LP_BK;
    static_assert(PER == 6 || PER == 8 || PER == 3 || PER == 4, "the counted waits below are written for 3, 4, 6 or 8 DMA instructions per chunk");
    stage(0, 0);
    if (STAGES == 3 && n_chunks > 1) stage(1, LP_BK);
    for (int kc = 0; kc < n_chunks; ++kc) {
        // chunk kc has landed (my share: the counted wait; everybody's: the barrier) and everybody is done reading chunk kc - 1.
        // Three stages: the DMA runs two chunks ahead (chunk kc + 1 may stay in flight); two stages: one chunk ahead.
        if (STAGES == 3 && kc + 1 < n_chunks) {
            if (PER == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else if (PER == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (PER == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        if (kc + STAGES - 1 < n_chunks) stage((kc + STAGES - 1) % STAGES, (kc + STAGES - 1) * LP_BK);
        const unsigned sb = (unsigned)((kc % STAGES) * STAGE * 16);
        if (NP == 1) {
            u32x4 w1[2], x1[2][4];
#pragma unroll
            for (int s_ = 0; s_ < 2; ++s_) {
                const unsigned pw = aw[s_] + sb, px = ax[s_] + sb;
                w1[s_] = lds_read<0>(pw);
                x1[s_][0] = lds_read<0 * 2048>(px); x1[s_][1] = lds_read<1 * 2048>(px);
                x1[s_][2] = lds_read<2 * 2048>(px); x1[s_][3] = lds_read<3 * 2048>(px);
            }
            asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(w1[0]), "+v"(x1[0][0]), "+v"(x1[0][1]), "+v"(x1[0][2]), "+v"(x1[0][3]));
#pragma unroll
            for (int s_ = 0; s_ < 2; ++s_) {
                if (s_ == 1) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(w1[1]), "+v"(x1[1][0]), "+v"(x1[1][1]), "+v"(x1[1][2]), "+v"(x1[1][3]));
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] = mfma_u(w1[s_], x1[s_][t], acc[t]);
            }
            continue;
        }
        u32x4 w_hi[2], w_lo[2], x_hi[2][4], x_lo[2][4];
#pragma unroll
        for (int s_ = 0; s_ < 2; ++s_) {
            const unsigned pw = aw[s_] + sb, px = ax[s_] + sb;
            w_hi[s_] = lds_read<0>(pw); w_lo[s_] = lds_read<LP_WC * 16>(pw);
            x_hi[s_][0] = lds_read<0 * 2048>(px); x_lo[s_][0] = lds_read<LP_XC * 16 + 0 * 2048>(px);
            x_hi[s_][1] = lds_read<1 * 2048>(px); x_lo[s_][1] = lds_read<LP_XC * 16 + 1 * 2048>(px);
            x_hi[s_][2] = lds_read<2 * 2048>(px); x_lo[s_][2] = lds_read<LP_XC * 16 + 2 * 2048>(px);
            x_hi[s_][3] = lds_read<3 * 2048>(px); x_lo[s_][3] = lds_read<LP_XC * 16 + 3 * 2048>(px);
        }
        // the first k16-step's ten fragments are back when at most the second step's ten reads are outstanding (LDS returns in order)
        asm volatile("s_waitcnt lgkmcnt(10)" : "+v"(w_hi[0]), "+v"(w_lo[0]), "+v"(x_hi[0][0]), "+v"(x_lo[0][0]), "+v"(x_hi[0][1]), "+v"(x_lo[0][1]),
                     "+v"(x_hi[0][2]), "+v"(x_lo[0][2]), "+v"(x_hi[0][3]), "+v"(x_lo[0][3]));
#pragma unroll
        for (int s_ = 0; s_ < 2; ++s_) {
            if (s_ == 1)
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(w_hi[1]), "+v"(w_lo[1]), "+v"(x_hi[1][0]), "+v"(x_lo[1][0]), "+v"(x_hi[1][1]),
                             "+v"(x_lo[1][1]), "+v"(x_hi[1][2]), "+v"(x_lo[1][2]), "+v"(x_hi[1][3]), "+v"(x_lo[1][3]));
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = mfma_u(w_lo[s_], x_hi[s_][t], acc[t]);          // smallest terms first; 4 independent chains
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = mfma_u(w_hi[s_], x_lo[s_][t], acc[t]);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = mfma_u(w_hi[s_], x_hi[s_][t], acc[t]);
        }
    }
    if (MODE == LP_DOT) {
        // ---- epilogue of the dot form: lane (j, h) of tile t holds features 32 wn + 8 g + 4 h + e of row m0 + 32 t + j.  Per row: the
        // wave's 32 features against v (16 in-lane terms in register order, then the other lane half), the eight waves through LDS in
        // wave order -- one fixed summation order whatever M is.
        float part[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float sacc = 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = wn * 32 + 8 * g + 4 * h;
                const float4 b4 = bias ? *reinterpret_cast<const float4*>(bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
                const float4 v4 = *reinterpret_cast<const float4*>(dot_v + n);
                float y[4] = {fmaf(acc[t][4 * g], wscale_inv, b4.x), fmaf(acc[t][4 * g + 1], wscale_inv, b4.y),
                              fmaf(acc[t][4 * g + 2], wscale_inv, b4.z), fmaf(acc[t][4 * g + 3], wscale_inv, b4.w)};
                if (act == ACT_GELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) y[e] = l3_gelu(y[e]);
                }
                sacc = fmaf(y[0], v4.x, sacc); sacc = fmaf(y[1], v4.y, sacc); sacc = fmaf(y[2], v4.z, sacc); sacc = fmaf(y[3], v4.w, sacc);
            }
            part[t] = sacc + __shfl_xor(sacc, 32, 64);     // (a + b == b + a: both halves hold the same value)
        }
        __syncthreads();                                   // every wave is done with the stages: the LDS is free
        float* red = reinterpret_cast<float*>(S);          // [8 waves][128 rows]
        if (h == 0) {
#pragma unroll
            for (int t = 0; t < 4; ++t) red[wave * 128 + t * 32 + i] = part[t];
        }
        __syncthreads();
        if (tid < 128 && m0 + tid < M) {
            float y = red[tid];
#pragma unroll
            for (int w = 1; w < 8; ++w) y += red[w * 128 + tid];
            y += dot_c ? dot_c[0] : 0.f;
            if (act2 == ACT_GELU) y = 0.5f * y * (1.f + erff(y * 0.70710678118654752440f));
            Y[m0 + tid] = y;
        }
        return;
    }
    // ---- epilogue: lane (j, h) of tile t holds features n = n0 + 32 wn + 8 g + 4 h + e (register 4 g + e) of row m0 + 128 wm + 32 t + j
#ifndef MCR_L3P_NO_TR
    // Out through the LDS (the stages are free once every wave has left the K loop).  A lane's register quad is 8 bytes of a row per
    // plane (16 bytes of fp32): a store instruction of the direct form below touches 32 rows with 16 (32) bytes each -- eight partial
    // writes per 128-byte line, issued by two waves; knocked out, those stores were 20-37 us of a 70-110 us encoder layer at 30 x 2048
    // tokens, the MFMAs 4 us.  Here the block's tile is laid out [plane][row][128 features] in the LDS (8-byte units XOR-swizzled by the
    // row: 4 lanes per bank pair, the minimum for 64 x 8 bytes) and leaves in whole rows: 16 lanes x 16 bytes = the 256 bytes of a row
    // and plane, four rows per store instruction.  Same values, same bits.
    constexpr int TR_THREADS = NW * 64;
    if (MODE == LP_PLANES && N % 8 == 0 && ldy % 8 == 0 && (((size_t)Yh | (NP == 2 ? (size_t)Yl : 0)) & 15) == 0) {
        __syncthreads();
        char* tb = reinterpret_cast<char*>(S);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int row = wm * 128 + t * 32 + i;
            const long long mr = min(m0 + row, M - 1);
            const long long grp = row_bias ? (row_group ? (long long)row_group[mr] : mr / rows_per_group) : 0;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + wn * 32 + 8 * g + 4 * h;
                if (n >= N) continue;
                float4 b4 = bias ? *reinterpret_cast<const float4*>(bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
                if (row_bias) {
                    const float4 rb = *reinterpret_cast<const float4*>(row_bias + grp * N + n);
                    b4.x += rb.x; b4.y += rb.y; b4.z += rb.z; b4.w += rb.w;
                }
                float y[4] = {fmaf(acc[t][4 * g], wscale_inv, b4.x), fmaf(acc[t][4 * g + 1], wscale_inv, b4.y),
                              fmaf(acc[t][4 * g + 2], wscale_inv, b4.z), fmaf(acc[t][4 * g + 3], wscale_inv, b4.w)};
                if (act == ACT_GELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) y[e] = l3_gelu(y[e]);
                }
                const int u = 8 * wn + 2 * g + h;
                const int off = row * 256 + ((u ^ ((row & 15) << 1)) << 3);
                if (NP == 1) {
                    *reinterpret_cast<uint2*>(tb + off) = make_uint2(pack2h(y[0], y[1]), pack2h(y[2], y[3]));
                } else {
                    uint2 hi, lo;
                    split2h(y[0], y[1], hi.x, lo.x);
                    split2h(y[2], y[3], hi.y, lo.y);
                    *reinterpret_cast<uint2*>(tb + off) = hi;
                    *reinterpret_cast<uint2*>(tb + LP_TM * 256 + off) = lo;
                }
            }
        }
        __syncthreads();
        constexpr int RPP = TR_THREADS / 16, PASSES = LP_TM / RPP;            // rows per pass; passes per plane
#pragma unroll
        for (int q = 0; q < NP * PASSES; ++q) {
            const int pl = q / PASSES, r = (q % PASSES) * RPP + (tid >> 4), pp = tid & 15;
            const uint4 v = *reinterpret_cast<const uint4*>(tb + pl * (LP_TM * 256) + r * 256 + ((pp ^ (r & 15)) << 4));
            const long long m = m0 + r;
            const int n = n0 + 8 * pp;
            if (m < M && n < N) *reinterpret_cast<uint4*>((pl ? Yl : Yh) + m * ldy + n) = v;
        }
        return;
    }
    // fp32 rows the same way: [row][128 floats], 16-byte units XOR-swizzled by the row; 32 lanes x 16 bytes = the 512 bytes of a row, the
    // residual read and the result written by the same lane in full lines
    if (MODE == LP_F32 && !row_bias && ldy % 4 == 0 && ((size_t)Y & 15) == 0 && (!R || (ldr % 4 == 0 && ((size_t)R & 15) == 0))) {
        __syncthreads();
        char* tb = reinterpret_cast<char*>(S);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int row = wm * 128 + t * 32 + i;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + wn * 32 + 8 * g + 4 * h;
                if (n >= N) continue;
                const float4 b4 = bias ? *reinterpret_cast<const float4*>(bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
                float y[4] = {fmaf(acc[t][4 * g], wscale_inv, b4.x), fmaf(acc[t][4 * g + 1], wscale_inv, b4.y),
                              fmaf(acc[t][4 * g + 2], wscale_inv, b4.z), fmaf(acc[t][4 * g + 3], wscale_inv, b4.w)};
                if (act == ACT_GELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) y[e] = l3_gelu(y[e]);
                }
                const int u = 8 * wn + 2 * g + h;
                *reinterpret_cast<float4*>(tb + row * 512 + ((u ^ (row & 31)) << 4)) = make_float4(y[0], y[1], y[2], y[3]);
            }
        }
        __syncthreads();
        constexpr int RPP = TR_THREADS / 32;
#pragma unroll
        for (int q = 0; q < LP_TM / RPP; ++q) {
            const int r = q * RPP + (tid >> 5), cc = tid & 31;
            float4 v = *reinterpret_cast<const float4*>(tb + r * 512 + ((cc ^ (r & 31)) << 4));
            const long long m = m0 + r;
            const int n = n0 + 4 * cc;
            if (m < M && n < N) {
                if (R) {                                   // residual (Attention.py:290, :298); R may be Y: read before the store
                    const float4 r4 = *reinterpret_cast<const float4*>(R + m * ldr + n);
                    v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w;
                }
                *reinterpret_cast<float4*>(Y + m * ldy + n) = v;
            }
        }
        return;
    }
#endif
    // the direct form (operands that are not 16-byte aligned, N % 8 != 0, a row bias on fp32 rows)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const long long m = m0 + wm * 128 + t * 32 + i;
        if (m >= M) continue;
        const long long grp = row_bias ? (row_group ? (long long)row_group[m] : m / rows_per_group) : 0;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n = n0 + wn * 32 + 8 * g + 4 * h;
            if (n >= N) continue;                          // N % 4 == 0: a quad is in or out as a whole
            float4 b4 = bias ? *reinterpret_cast<const float4*>(bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
            if (row_bias) {
                const float4 rb = *reinterpret_cast<const float4*>(row_bias + grp * N + n);
                b4.x += rb.x; b4.y += rb.y; b4.z += rb.z; b4.w += rb.w;
            }
            float y[4] = {fmaf(acc[t][4 * g], wscale_inv, b4.x), fmaf(acc[t][4 * g + 1], wscale_inv, b4.y),
                          fmaf(acc[t][4 * g + 2], wscale_inv, b4.z), fmaf(acc[t][4 * g + 3], wscale_inv, b4.w)};
            if (act == ACT_GELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] = l3_gelu(y[e]);
            }
            if (PLANES_OUT && NP == 1) {
                *reinterpret_cast<uint2*>(Yh + m * ldy + n) = make_uint2(pack2h(y[0], y[1]), pack2h(y[2], y[3]));
            } else if (PLANES_OUT) {
                uint2 hi, lo;
                split2h(y[0], y[1], hi.x, lo.x);
                split2h(y[2], y[3], hi.y, lo.y);
                *reinterpret_cast<uint2*>(Yh + m * ldy + n) = hi;
                *reinterpret_cast<uint2*>(Yl + m * ldy + n) = lo;
            } else {
                if (R) {                                   // residual (Attention.py:290, :298); R may be Y: read before the store
                    const float4 r4 = *reinterpret_cast<const float4*>(R + m * ldr + n);
                    y[0] += r4.x; y[1] += r4.y; y[2] += r4.z; y[3] += r4.w;
                }
                *reinterpret_cast<float4*>(Y + m * ldy + n) = make_float4(y[0], y[1], y[2], y[3]);
            }
        }
    }
}

// fp32 rows -> fp16 hi/lo planes: P[m][col0 + c] for c < E (E % 4 == 0); one thread per 4 consecutive values
__global__ void split_to_planes_kernel(const float* __restrict__ X, long long ldx, _Float16* __restrict__ Ph, _Float16* __restrict__ Pl,
                                       long long ldp, long long M, int E4) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= M * E4) return;
    const long long m = idx / E4;
    const int c = (int)(idx - m * E4) * 4;
    const float4 v = *reinterpret_cast<const float4*>(X + m * ldx + c);
    uint2 hi, lo;
    split2h(v.x, v.y, hi.x, lo.x);
    split2h(v.z, v.w, hi.y, lo.y);
    *reinterpret_cast<uint2*>(Ph + m * ldp + c) = hi;
    if (Pl) *reinterpret_cast<uint2*>(Pl + m * ldp + c) = lo;          // (NULL: the single-plane variant 7 keeps fp16(x) alone)
}

// ---- ONE-SHOT form for a few thousand rows (one cloud of 2048 tokens through an encoder: 48 ... 256 blocks of the pipelined forms, each
// waiting out eight to sixteen chunk round trips of ~1 us behind a barrier -- 17-20 us per layer for 1.6 GFLOP).  Block = 4 waves = 64
// features x 64 rows (wave: 32 x 32, one accumulator); the block's whole K slice (up to 256 per shot: X and W, both planes, 128 KB)
// is queued by DMA at once, ONE wait, ONE barrier, then 16 k16-steps of three MFMAs.  Same products in the same order per output
// element as the pipelined forms (k ascending; per k16-step w_lo x_hi, w_hi x_lo, w_hi x_hi): the bits do not depend on the form, so a
// cloud alone and the same cloud inside a batch (pipelined form) agree.  LDS image: [row][K] per plane, the 16-byte chunks of a row
// XOR-swizzled by the row (conflict-free ds_read_b128 of one chunk column over 32 rows), applied on the DMA's source address.
constexpr int LPO_T = 64, LPO_KS = 256;                                   // tile edge; k per shot
constexpr int LPO_LDS_BYTES = 4 * LPO_T * LPO_KS * 2;                     // Xh | Xl | Wh | Wl = 131 072
template <int MODE, int NP = 2>
__global__ __launch_bounds__(256, 1) void linear3p_once_kernel(const _Float16* __restrict__ Xh, const _Float16* __restrict__ Xl, long long ldx,
                                                               const _Float16* __restrict__ Wh, const _Float16* __restrict__ Wl, long long ldw,
                                                               const float* __restrict__ bias, float* __restrict__ Y, _Float16* __restrict__ Yh,
                                                               _Float16* __restrict__ Yl, long long ldy, long long M, int N, int K, int act,
                                                               float wscale_inv, const float* __restrict__ R, long long ldr) {
    static_assert(MODE == LP_F32 || MODE == LP_PLANES, "one-shot form: fp32 rows or planes out");
    extern __shared__ __attribute__((aligned(16))) uint4 S[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave & 1, wm = wave >> 1;
    const int ncb = (N + LPO_T - 1) / LPO_T;
    const long long m0 = (long long)(blockIdx.x / ncb) * LPO_T;          // column blocks of one row block are neighbours (the rows stay in L2)
    const int n0 = (blockIdx.x % ncb) * LPO_T;
    const int i = lane & 31, h = lane >> 5;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int k0 = 0; k0 < K; k0 += LPO_KS) {
        const int ks = min(LPO_KS, K - k0);                                // 128 or 256 (the launcher's precondition)
        const int csh = ks == 256 ? 5 : 4, cpr = 1 << csh;                 // chunks (16 bytes) per row: 32 at a full shot
        const int plane = LPO_T << csh;                                     // chunks per plane
        if (k0) __syncthreads();                                            // (the previous shot's fragments are consumed)
        // DMA: 64 chunks per wave instruction; position p of a plane = (row = p / cpr, c' = p % cpr) receives the row's chunk c' ^ (row & (cpr - 1))
        for (int q = wave; q < 2 * NP * cpr; q += 4) {                     // (a plane is cpr instructions; NP == 1: the two high planes)
            const int pl = NP == 2 ? q >> csh : (q >> csh) * 2, p = (q & (cpr - 1)) * 64 + lane;
            const int row = p >> csh, c = (p & (cpr - 1)) ^ (row & (cpr - 1));
            const _Float16* src;
            if (pl < 2) src = (pl ? Xl : Xh) + min(m0 + row, M - 1) * ldx + k0 + c * 8;       // rows beyond the matrix repeat its last row
            else src = (pl == 3 ? Wl : Wh) + (long long)min(n0 + row, N - 1) * ldw + k0 + c * 8;
            __builtin_amdgcn_global_load_lds((lp_gptr)src, (lp_lptr)(S + pl * plane + (q & (cpr - 1)) * 64), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const uint4* sx_h = S + ((wm * 32 + i) << csh);                      // this lane's X row / W row inside the planes
        const uint4* sw_h = S + 2 * plane + ((wn * 32 + i) << csh);
        const int key = i & (cpr - 1);                                       // (wm * 32 + i) & (cpr - 1): cpr <= 32
        for (int st4 = 0; st4 < ks / 64; ++st4) {                           // four k16-steps per trip: their twelve fragment reads go out together
            uint4 xh[4], xl[4], wh[4], wl[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = (2 * (4 * st4 + u) + h) ^ key;
                xh[u] = sx_h[c]; wh[u] = sw_h[c];
                if (NP == 2) { xl[u] = sx_h[plane + c]; wl[u] = sw_h[plane + c]; }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (NP == 2) {
                    acc = mfma_h(wl[u], xh[u], acc);                        // smallest terms first (the order of the pipelined forms)
                    acc = mfma_h(wh[u], xl[u], acc);
                }
                acc = mfma_h(wh[u], xh[u], acc);
            }
        }
    }
    // ---- epilogue: lane (j, h) holds features n = n0 + 32 wn + 8 g + 4 h + e (register 4 g + e) of row m0 + 32 wm + j
    const long long m = m0 + wm * 32 + i;
    if (m >= M) return;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int n = n0 + wn * 32 + 8 * g + 4 * h;
        if (n >= N) continue;                              // N % 4 == 0: a quad is in or out as a whole
        const float4 b4 = bias ? *reinterpret_cast<const float4*>(bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
        float y[4] = {fmaf(acc[4 * g], wscale_inv, b4.x), fmaf(acc[4 * g + 1], wscale_inv, b4.y), fmaf(acc[4 * g + 2], wscale_inv, b4.z),
                      fmaf(acc[4 * g + 3], wscale_inv, b4.w)};
        if (act == ACT_GELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = l3_gelu(y[e]);
        }
        if (MODE == LP_PLANES && NP == 1) {
            *reinterpret_cast<uint2*>(Yh + m * ldy + n) = make_uint2(pack2h(y[0], y[1]), pack2h(y[2], y[3]));
        } else if (MODE == LP_PLANES) {
            uint2 hi, lo;
            split2h(y[0], y[1], hi.x, lo.x);
            split2h(y[2], y[3], hi.y, lo.y);
            *reinterpret_cast<uint2*>(Yh + m * ldy + n) = hi;
            *reinterpret_cast<uint2*>(Yl + m * ldy + n) = lo;
        } else {
            if (R) {                                       // residual; R may be Y: read before the store
                const float4 r4 = *reinterpret_cast<const float4*>(R + m * ldr + n);
                y[0] += r4.x; y[1] += r4.y; y[2] += r4.z; y[3] += r4.w;
            }
            *reinterpret_cast<float4*>(Y + m * ldy + n) = make_float4(y[0], y[1], y[2], y[3]);
        }
    }
}

// ---- launchers: one instantiation per (mode, form, planes) ----
template <int MODE, int NP>
static bool lpo_reserve() {
    static const bool ok = hipFuncSetAttribute((const void*)linear3p_once_kernel<MODE, NP>, hipFuncAttributeMaxDynamicSharedMemorySize, LPO_LDS_BYTES) == hipSuccess;
    return ok;
}
template <int MODE, bool SMALL, int NP>
static bool lp_reserve() {
    static const bool ok = hipFuncSetAttribute((const void*)linear3p_kernel<MODE, SMALL, NP>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                               SMALL ? LPS_LDS_BYTES : LP_LDS_BYTES) == hipSuccess;
    return ok;
}

bool linear3p_applicable(int N, int K, int64_t ldx, int64_t ldw, int64_t ldy) {
    return K % LP_BK == 0 && K >= LP_BK && N % 4 == 0 && ldx % 8 == 0 && ldw % 8 == 0 && ldy % 4 == 0;
}

template <int NP>
static void launch_linear3p_np(hipStream_t s, const void* Xh, const void* Xl, int64_t ldx, const void* Wh, const void* Wl, int64_t ldw,
                               const float* bias, float* Y, void* Yh, void* Yl, int64_t ldy, int64_t M, int N, int K, int act, float wscale_inv,
                               const float* row_bias, int64_t rows_per_group, const int* row_group, const float* R, int64_t ldr) {
    const long long rpg = rows_per_group > 0 ? rows_per_group : 1;
    // short K: the two-blocks-per-CU form (same bits); MCR_L3P_SMALL=0: the large tile whatever K is (A/B), =2: the small one always
    static const int small_mode = []() { const char* e = getenv("MCR_L3P_SMALL"); return e ? atoi(e) : 1; }();
    // a few thousand rows: the one-shot form (same bits); MCR_L3P_ONCE=0: off (A/B)
    static const bool once_on = []() { const char* e = getenv("MCR_L3P_ONCE"); return !(e && e[0] == '0'); }();
    const _Float16 *xh = (const _Float16*)Xh, *xl = (const _Float16*)Xl, *wh = (const _Float16*)Wh, *wl = (const _Float16*)Wl;
    _Float16 *yh = (_Float16*)Yh, *yl = (_Float16*)Yl;
    const float* nf = nullptr;
    if (once_on && M <= 4096 && !row_bias && (K == 128 || K % LPO_KS == 0)) {
        if (!(Yh ? lpo_reserve<LP_PLANES, NP>() : lpo_reserve<LP_F32, NP>())) { set_error("launch_linear3p: cannot reserve %d bytes of LDS", LPO_LDS_BYTES); return; }
        dim3 g((unsigned)(cdiv(M, LPO_T) * cdiv(N, LPO_T)));
        if (Yh)
            hipLaunchKernelGGL((linear3p_once_kernel<LP_PLANES, NP>), g, dim3(256), LPO_LDS_BYTES, s, xh, xl, (long long)ldx, wh, wl, (long long)ldw, bias,
                               (float*)nullptr, yh, yl, (long long)ldy, (long long)M, N, K, act, wscale_inv, nf, 0ll);
        else
            hipLaunchKernelGGL((linear3p_once_kernel<LP_F32, NP>), g, dim3(256), LPO_LDS_BYTES, s, xh, xl, (long long)ldx, wh, wl, (long long)ldw, bias, Y,
                               (_Float16*)nullptr, (_Float16*)nullptr, (long long)ldy, (long long)M, N, K, act, wscale_inv, R, (long long)ldr);
        return;
    }
    if (small_mode == 2 || (small_mode == 1 && K <= 512)) {
        if (!(Yh ? lp_reserve<LP_PLANES, true, NP>() : lp_reserve<LP_F32, true, NP>())) { set_error("launch_linear3p: cannot reserve %d bytes of LDS", LPS_LDS_BYTES); return; }
        dim3 g((unsigned)(cdiv(cdiv(M, 128), 8) * 8 * cdiv(N, 128)));
        if (Yh)
            hipLaunchKernelGGL((linear3p_kernel<LP_PLANES, true, NP>), g, dim3(256), LPS_LDS_BYTES, s, xh, xl, (long long)ldx, wh, wl, (long long)ldw, bias,
                               row_bias, rpg, row_group, (float*)nullptr, yh, yl, (long long)ldy, (long long)M, N, K, act, wscale_inv, nf, nf, 0, nf, 0ll);
        else
            hipLaunchKernelGGL((linear3p_kernel<LP_F32, true, NP>), g, dim3(256), LPS_LDS_BYTES, s, xh, xl, (long long)ldx, wh, wl, (long long)ldw, bias,
                               row_bias, rpg, row_group, Y, (_Float16*)nullptr, (_Float16*)nullptr, (long long)ldy, (long long)M, N, K, act, wscale_inv,
                               nf, nf, 0, R, (long long)ldr);
        return;
    }
    if (!(Yh ? lp_reserve<LP_PLANES, false, NP>() : lp_reserve<LP_F32, false, NP>())) { set_error("launch_linear3p: cannot reserve %d bytes of LDS", LP_LDS_BYTES); return; }
    dim3 grid((unsigned)(cdiv(cdiv(M, 256), 8) * 8 * cdiv(N, 128)));             // 1-D: see the XCD-aware block order in the kernel
    if (Yh)
        hipLaunchKernelGGL((linear3p_kernel<LP_PLANES, false, NP>), grid, dim3(512), LP_LDS_BYTES, s, xh, xl, (long long)ldx, wh, wl, (long long)ldw, bias,
                           row_bias, rpg, row_group, (float*)nullptr, yh, yl, (long long)ldy, (long long)M, N, K, act, wscale_inv, nf, nf, 0, nf, 0ll);
    else
        hipLaunchKernelGGL((linear3p_kernel<LP_F32, false, NP>), grid, dim3(512), LP_LDS_BYTES, s, xh, xl, (long long)ldx, wh, wl, (long long)ldw, bias,
                           row_bias, rpg, row_group, Y, (_Float16*)nullptr, (_Float16*)nullptr, (long long)ldy, (long long)M, N, K, act, wscale_inv, nf,
                           nf, 0, R, (long long)ldr);
}

// Y (fp32, ldy floats) or Yh / Yl (fp16 planes, ldy halves) = act(X W^T * wscale_inv + bias (+ row bias)); exactly one of Y, Yh is set.
// n_planes = 1: the high planes alone (variant 7): Xl / Wl / Yl are ignored
void launch_linear3p(hipStream_t s, const void* Xh, const void* Xl, int64_t ldx, const void* Wh, const void* Wl, int64_t ldw,
                     const float* bias, float* Y, void* Yh, void* Yl, int64_t ldy, int64_t M, int N, int K, int act, float wscale_inv,
                     const float* row_bias, int64_t rows_per_group, const int* row_group, const float* R, int64_t ldr, int n_planes) {
    if (M <= 0 || N <= 0) return;
    if (n_planes == 1)
        launch_linear3p_np<1>(s, Xh, Xl, ldx, Wh, Wl, ldw, bias, Y, Yh, Yl, ldy, M, N, K, act, wscale_inv, row_bias, rows_per_group, row_group, R, ldr);
    else
        launch_linear3p_np<2>(s, Xh, Xl, ldx, Wh, Wl, ldw, bias, Y, Yh, Yl, ldy, M, N, K, act, wscale_inv, row_bias, rows_per_group, row_group, R, ldr);
}

// out[m] = act2( act(X W^T * wscale_inv + bias)[m][:] . v + c ) for a 256-feature layer (N == 256): two layers, one launch
bool linear3p_dot_applicable(int N, int K, int64_t ldx, int64_t ldw) { return N == 256 && K % LP_BK == 0 && K >= LP_BK && ldx % 8 == 0 && ldw % 8 == 0; }
template <int NP>
static void launch_linear3p_dot_np(hipStream_t s, const void* Xh, const void* Xl, int64_t ldx, const void* Wh, const void* Wl, int64_t ldw,
                                   const float* bias, int64_t M, int K, int act, float wscale_inv, const float* v, const float* c, int act2, float* out) {
    if (!lp_reserve<LP_DOT, false, NP>()) { set_error("launch_linear3p_dot: cannot reserve %d bytes of LDS", LP_LDS_BYTES); return; }
    dim3 grid((unsigned)(cdiv(cdiv(M, 128), 8) * 8));
    hipLaunchKernelGGL((linear3p_kernel<LP_DOT, false, NP>), grid, dim3(512), LP_LDS_BYTES, s, (const _Float16*)Xh, (const _Float16*)Xl, (long long)ldx,
                       (const _Float16*)Wh, (const _Float16*)Wl, (long long)ldw, bias, (const float*)nullptr, 1ll, (const int*)nullptr, out,
                       (_Float16*)nullptr, (_Float16*)nullptr, 1ll, (long long)M, 256, K, act, wscale_inv, v, c, act2, (const float*)nullptr, 0ll);
}
void launch_linear3p_dot(hipStream_t s, const void* Xh, const void* Xl, int64_t ldx, const void* Wh, const void* Wl, int64_t ldw,
                         const float* bias, int64_t M, int K, int act, float wscale_inv, const float* v, const float* c, int act2, float* out,
                         int n_planes) {
    if (M <= 0) return;
    if (n_planes == 1) launch_linear3p_dot_np<1>(s, Xh, Xl, ldx, Wh, Wl, ldw, bias, M, K, act, wscale_inv, v, c, act2, out);
    else launch_linear3p_dot_np<2>(s, Xh, Xl, ldx, Wh, Wl, ldw, bias, M, K, act, wscale_inv, v, c, act2, out);
}

void launch_split_to_planes(hipStream_t s, const float* X, int64_t ldx, void* Ph, void* Pl, int64_t ldp, int64_t M, int E) {
    if (M <= 0 || E <= 0) return;
    const int E4 = E / 4;
    hipLaunchKernelGGL(split_to_planes_kernel, dim3((unsigned)cdiv(M * E4, 256)), dim3(256), 0, s, X, (long long)ldx, (_Float16*)Ph,
                       (_Float16*)Pl, (long long)ldp, (long long)M, E4);
}

}  // namespace mcr
