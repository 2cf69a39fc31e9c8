// K6-planes -- long-sequence multi-head attention (Attention.py:8-36, 174-198; mask=None) whose packed q | k | v operand already IS a
// pair of fp16 hi/lo planes in HBM: the encoders of the fp16-split numerics variant (variant 6, L >= 512), where the QKV projection's
// epilogue (linear3p, LP_PLANES) splits every value once.  Same tiling and the same per-query arithmetic as attention_mfma_kernel's
// fp16 path (nn_kernels.hip): a block = 4 waves x QG 16-query groups of one (sequence, head), keys / values in tiles of 64,
// S^T = K Q^T and O += P V on v_mfma_f32_16x16x32_f16 with hi/lo pairs, online softmax in units of log 2 -- but
//   * nothing is converted or split while a tile is staged: both planes of a K / V tile travel HBM -> LDS by global_load_lds_dwordx4
//     (20 KB per tile at head dims (16, 64): five DMA instructions per wave), into TWO stages, one barrier per tile; the fetch of
//     tile t + 1 is queued before tile t is multiplied.  The staging of the fp32 kernel (float4 loads held in registers, range
//     test, two fp16 conversions per value, ds_write) was a quarter of its vector instructions, and vector issue is what bounds it
//     (PMC: VALU 60 % busy, matrix pipe 29 %);
//   * the 1 / sqrt(d) log2(e) factor is applied inside the exponent, exp2(fma(s, c, -m)): the Q planes are used as they are;
//   * fragment reads are inline asm with hand-counted waits (hipcc cannot prove that an LDS read of one stage does not alias the DMA
//     queued into the other and would drain the DMA queue in front of every read, cf. linear3p.hip).
// No fp32 fallback inside: a value outside the fp16 range became inf in the producer's planes and surfaces as inf / NaN in the
// network's output, where the variant's range guard (networks/packing.py: RangeGuard) sees it -- the contract of every planes GEMM.
//
// LDS image of a stage: K_hi | K_lo, each [64 keys][DQ] (DQ = 16: the two 16-byte chunks of a row swapped for rows 8..15 of every 16:
// conflict-free ds_read_b128 fragments), then V_hi | V_lo, each [DV / 16][64 keys][16 columns] (read back transposed by
// ds_read_b64_tr_b16).  The DMA's LDS image is lane-linear, so both layouts are applied on the SOURCE address.
#include "lp_split.h"
#include "nn_kernels.h"

namespace mcr {

typedef const __attribute__((address_space(1))) void* ap_gptr;
typedef __attribute__((address_space(3))) void* ap_lptr;
typedef unsigned ap_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned ap_u32x2 __attribute__((ext_vector_type(2)));
typedef float ap_f32x4 __attribute__((ext_vector_type(4)));

template <int OFF>
__device__ __forceinline__ ap_u32x4 ap_read128(unsigned addr) {
    ap_u32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
template <int OFF>
__device__ __forceinline__ ap_u32x2 ap_read_tr(unsigned addr) {       // 4 keys of one column: the B fragment of O += P V
    ap_u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}

// max over the four lanes (li, g = 0..3) that share a query: v_permlane32_swap / v_permlane16_swap exchange halves / neighbouring 16-lane
// rows between two registers in one instruction each -- no LDS round trip (two dependent ds_bpermute were ~250 cycles of every tile's
// critical path)
__device__ __forceinline__ float ap_max_lane_groups(float x) {
    const unsigned u = __builtin_bit_cast(unsigned, x);
    const auto a = __builtin_amdgcn_permlane32_swap(u, u, false, false);          // -> [lo, lo], [hi, hi]
    x = fmaxf(__builtin_bit_cast(float, a[0]), __builtin_bit_cast(float, a[1]));
    const unsigned v = __builtin_bit_cast(unsigned, x);
    const auto b = __builtin_amdgcn_permlane16_swap(v, v, false, false);          // -> rows [0, 0, 2, 2], [1, 1, 3, 3]
    return fmaxf(__builtin_bit_cast(float, b[0]), __builtin_bit_cast(float, b[1]));
}

#ifndef MCR_AP_OCC_16
#define MCR_AP_OCC_16 3
#endif
#ifndef MCR_AP_OCC_8
#define MCR_AP_OCC_8 3
#endif

// grid = (ceil(L / (64 QG)), H, S or 2 S); Ph / Pl: planes [S * L][ldp] of q | k | v.  Not SPLIT: the result leaves normalised, as planes
// Oh / Ol (row stride ldo halves) when given, else as fp32 rows of `out` (row stride ldo floats).  SPLIT: the two blocks of a (query tile,
// head, sequence) take the two halves of the keys and write unnormalised fp32 parts (part 0 -> out, part 1 -> part1 [T, H DV]) plus their
// (running max in units of log 2, sum) per query -> ml [2][T][H][2]; attention_combine_kernel merges them.
// NP = 2: hi / lo planes (variant 6).  NP = 1 (variant 7, the opt-in 16-bit matrix path): the HIGH planes alone -- the low planes are
// neither staged nor read (their LDS regions stay unused), the Q fragment's low half is zero (DQ = 16: the MFMA's k = 16 .. 31; DQ = 8:
// k = 8 .. 31), one MFMA per score tile and per P V tile instead of two / three, the soft-max weights are rounded once instead of split,
// and the result leaves as ONE plane (Ol is not written).  Scores, soft-max and accumulation stay fp32.
template <int DQ, int DV, bool SPLIT, int QG, int NP = 2>
__global__ __launch_bounds__(256, DQ == 16 ? MCR_AP_OCC_16 : MCR_AP_OCC_8) void attention_planes_kernel(
    const _Float16* __restrict__ Ph, const _Float16* __restrict__ Pl, long long ldp, float* __restrict__ out, _Float16* __restrict__ Oh,
    _Float16* __restrict__ Ol, long long ldo, int L, int H, const int* __restrict__ lens, float* __restrict__ part1, float* __restrict__ ml) {
    static_assert((DQ == 16 && DV == 64) || (DQ == 8 && DV == 32), "head dims");
    constexpr int TK = 64, NT = DV / 16, QB = 64 * QG;
    constexpr int KBYTES = TK * DQ * 2, VBYTES = TK * DV * 2, STAGE = 2 * KBYTES + 2 * VBYTES;       // 20 480 / 10 240 bytes
    constexpr int NK = KBYTES / 1024, NV = VBYTES / 1024, NI = 2 * NK + 2 * NV, NPW = (NI + 3) / 4;    // DMA instructions (1 KB each) per tile; per wave
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * STAGE];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), li = lane & 15, g = lane >> 4;
    const int hh = blockIdx.y;
    const int seq = SPLIT ? blockIdx.z >> 1 : blockIdx.z, part = SPLIT ? blockIdx.z & 1 : 0;
    const long long seq0 = (long long)seq * L;
    const int Lk_all = lens ? max(1, min(L, __builtin_amdgcn_readfirstlane(lens[seq]))) : L;             // number of keys
    const int kmid = min(Lk_all, ((Lk_all / 2 + TK - 1) / TK) * TK);                                    // tile-aligned cut
    const int kb = SPLIT && part ? kmid : 0, Lk = SPLIT && !part ? kmid : Lk_all;                        // this block's keys [kb, Lk)
    const int q0 = blockIdx.x * QB + wave * 16 * QG;    // + 16 qg: the wave's query groups
    const int koff = H * DQ + hh * DQ, voff = 2 * H * DQ + hh * DV;
    const float c2 = 1.4426950408889634f / sqrtf((float)DQ);   // scores -> units of log 2

    // ---- this wave's DMA instructions j = wave + 4 n of a tile: the plane (wave-uniform), and per lane the key row inside the tile and the
    // column (halves) of the 16 bytes it moves.  Whole tiles advance one 32-bit offset per instruction (relative to the sequence's first
    // row: L ldp < 2^31, checked by the launcher); only a sequence's last, partial tile decodes again and clamps its rows.
    const _Float16 *bh = Ph + seq0 * ldp, *bl = NP == 2 ? Pl + seq0 * ldp : bh;          // (NP == 1: Pl may be NULL, never formed)
    auto decode = [&](int j, int& row, int& col) -> bool {   // -> lo plane?
        bool lo;
        if (j < 2 * NK) {
            lo = j >= NK;
            const int p = (j - (lo ? NK : 0)) * 64 + lane;
            if (DQ == 16) { row = p >> 1; col = koff + 8 * ((p & 1) ^ ((row >> 3) & 1)); }
            else { row = p; col = koff; }
        } else {
            const int j2 = j - 2 * NK;
            lo = j2 >= NV;
            const int jj = j2 - (lo ? NV : 0), p = (jj & 1) * 64 + lane;
            row = p >> 1; col = voff + (jj >> 1) * 16 + 8 * (p & 1);
        }
        return lo;
    };
    const int ldp_i = (int)ldp, step = TK * ldp_i;
    int d_off[NPW];
#pragma unroll
    for (int n = 0; n < NPW; ++n) {
        int row, col;
        decode(wave + 4 * n, row, col);
        d_off[n] = (kb + row) * ldp_i + col;
    }
    auto fetch = [&](int st, int t0) {
        const bool whole = t0 + TK <= Lk;                 // block-uniform
#pragma unroll
        for (int n = 0; n < NPW; ++n) {
            const int j = wave + 4 * n;                   // wave-uniform
            if (j < NI) {
                const bool lo = j < 2 * NK ? j >= NK : j - 2 * NK >= NV;
                if (NP == 1 && lo) continue;              // (wave-uniform: the low planes are not staged)
                int off = d_off[n];
                if (!whole) {                             // (rows past the keys repeat the last key: finite, and their scores are masked)
                    int row, col;
                    decode(j, row, col);
                    off = min(t0 + row, Lk - 1) * ldp_i + col;
                }
                __builtin_amdgcn_global_load_lds((ap_gptr)((lo ? bl : bh) + off), (ap_lptr)(smem + st * STAGE + j * 1024), 16, 0, 0);
                d_off[n] += step;
            }
        }
    };

    // ---- Q fragments (B operand of S^T): DQ = 16: lane groups 0, 1 carry the dimensions 0..7, 8..15 of q_hi, groups 2, 3 those of q_lo;
    // DQ = 8: q_hi | q_lo | q_hi | q_lo by lane group
    f16x8 qh[QG];
#pragma unroll
    for (int qg = 0; qg < QG; ++qg) {
        const int qi = min(q0 + 16 * qg + li, L - 1);
        const bool lo_part = DQ == 16 ? g >= 2 : (g & 1);
        const _Float16* qp = ((NP == 2 && lo_part) ? bl : bh) + (long long)qi * ldp + hh * DQ + (DQ == 16 ? 8 * (g & 1) : 0);
        qh[qg] = *reinterpret_cast<const f16x8*>(qp);
        if (NP == 1 && (DQ == 16 ? g >= 2 : g >= 1)) qh[qg] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};      // k beyond the head's DQ dims: zeros
    }
    float m[QG];                                        // running max (units of log 2; lane (li, any g): query li)
    // O and the softmax denominators in C layout (lane (c, g): queries 4g + r): the denominators are one more column block of P V, with
    // V = 1 -- two MFMAs per 32 keys instead of 16 dependent vector adds per query group and tile (vector issue bounds the kernel, the
    // matrix pipe idles), and they already sit where the final division needs them
    ap_f32x4 o[QG][NT], ol[QG];
#pragma unroll
    for (int qg = 0; qg < QG; ++qg) {
        m[qg] = -__builtin_inff(); ol[qg] = ap_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) o[qg][nt] = ap_f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const f16x8 ones = {(_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f};
    // fragment addresses inside a stage (bytes)
    const unsigned lds0 = (unsigned)(size_t)((ap_lptr)smem);
    // K: DQ = 16: row (sub 16 + li), chunk (g & 1) ^ (li >> 3) of the hi plane (+ KBYTES: lo); DQ = 8: row of the plane the lane group reads
    const unsigned a_k = DQ == 16 ? lds0 + (unsigned)(li * 32 + (((g & 1) ^ (li >> 3)) * 16)) : lds0 + (unsigned)(li * 16 + ((NP == 2 && g >= 2) ? KBYTES : 0));
    // V: [nt][key][16]: lane (li, g) -> keys 4 g.., columns 4 (li & 3).. of the 16-key group; the transposing read hands lane (c, g) keys 4 g.. of column c
    const unsigned a_v = lds0 + (unsigned)(2 * KBYTES + ((4 * g) * 16 + li * 4) * 2);

    if (kb < Lk) fetch(0, kb);
    int st_i = 0;
    for (int t0 = kb; t0 < Lk; t0 += TK, st_i ^= 1) {
        // tile t0 has landed (my share: the wait; everybody's: the barrier) and everybody is done with the other stage
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#ifdef MCR_AP_EXP_NODMA      // (experiment: wrong results -- every tile multiplies the first one's bytes; prices the K / V stream)
        const unsigned sb = 0;
#else
        if (t0 + TK < Lk) fetch(st_i ^ 1, t0 + TK);
        const unsigned sb = (unsigned)(st_i * STAGE);
#endif
        // ---- scores of the 64 keys: two MFMAs (K_lo, then K_hi: smallest terms first) per 16 keys x 16 queries at DQ = 16, one at DQ = 8
        constexpr int KF = DQ == 16 ? 2 : 1;
        ap_u32x4 kf[4][KF];
        if constexpr (DQ == 16 && NP == 1) {                // (the K_hi fragments alone, in slot 0: the second MFMA is skipped; no copy of a
            kf[0][0] = ap_read128<0>(a_k + sb); kf[1][0] = ap_read128<512>(a_k + sb);      // fragment may be made before the counted wait below)
            kf[2][0] = ap_read128<1024>(a_k + sb); kf[3][0] = ap_read128<1536>(a_k + sb);
        } else if constexpr (DQ == 16) {
            kf[0][0] = ap_read128<KBYTES>(a_k + sb); kf[0][KF - 1] = ap_read128<0>(a_k + sb);
            kf[1][0] = ap_read128<KBYTES + 512>(a_k + sb); kf[1][KF - 1] = ap_read128<512>(a_k + sb);
            kf[2][0] = ap_read128<KBYTES + 1024>(a_k + sb); kf[2][KF - 1] = ap_read128<1024>(a_k + sb);
            kf[3][0] = ap_read128<KBYTES + 1536>(a_k + sb); kf[3][KF - 1] = ap_read128<1536>(a_k + sb);
        } else {
            kf[0][0] = ap_read128<0>(a_k + sb); kf[1][0] = ap_read128<256>(a_k + sb);
            kf[2][0] = ap_read128<512>(a_k + sb); kf[3][0] = ap_read128<768>(a_k + sb);
        }
        ap_u32x2 vf[NT][4], vg[NT][4];                     // [nt][hi k 0..3 | hi k 4..7 | lo k 0..3 | lo k 4..7] of the MFMA's k = 8 g + r; vg: keys 32..63
#define MCR_AP_VREAD(V_, NT_, BASE_)                                                                                                  \
    do {                                                                                                                               \
        V_[NT_][0] = ap_read_tr<(NT_) * 2048 + (BASE_)>(a_v + sb); V_[NT_][1] = ap_read_tr<(NT_) * 2048 + (BASE_) + 512>(a_v + sb);      \
        if constexpr (NP == 2) {                                                                                                       \
            V_[NT_][2] = ap_read_tr<VBYTES + (NT_) * 2048 + (BASE_)>(a_v + sb);                                                         \
            V_[NT_][3] = ap_read_tr<VBYTES + (NT_) * 2048 + (BASE_) + 512>(a_v + sb);                                                   \
        }                                                                                                                              \
    } while (0)
#define MCR_AP_VWAIT(V_)                                                                                                              \
    do {                                                                                                                               \
        if constexpr (NP == 1 && NT == 4)                                                                                              \
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(V_[0][0]), "+v"(V_[0][1]), "+v"(V_[1][0]), "+v"(V_[1][1]), "+v"(V_[NT - 2][0]),   \
                         "+v"(V_[NT - 2][1]), "+v"(V_[NT - 1][0]), "+v"(V_[NT - 1][1]));                                                  \
        else if constexpr (NP == 1)                                                                                                    \
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(V_[0][0]), "+v"(V_[0][1]), "+v"(V_[1][0]), "+v"(V_[1][1]));                       \
        else if constexpr (NT == 4)                                                                                                    \
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(V_[0][0]), "+v"(V_[0][1]), "+v"(V_[0][2]), "+v"(V_[0][3]), "+v"(V_[1][0]), "+v"(V_[1][1]), \
                         "+v"(V_[1][2]), "+v"(V_[1][3]), "+v"(V_[NT - 2][0]), "+v"(V_[NT - 2][1]), "+v"(V_[NT - 2][2]), "+v"(V_[NT - 2][3]),  \
                         "+v"(V_[NT - 1][0]), "+v"(V_[NT - 1][1]), "+v"(V_[NT - 1][2]), "+v"(V_[NT - 1][3]));                            \
        else                                                                                                                           \
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(V_[0][0]), "+v"(V_[0][1]), "+v"(V_[0][2]), "+v"(V_[0][3]), "+v"(V_[1][0]), "+v"(V_[1][1]), \
                         "+v"(V_[1][2]), "+v"(V_[1][3]));                                                                              \
    } while (0)
        if constexpr (DQ == 16 && NP == 2)
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kf[0][0]), "+v"(kf[0][KF - 1]), "+v"(kf[1][0]), "+v"(kf[1][KF - 1]), "+v"(kf[2][0]),
                         "+v"(kf[2][KF - 1]), "+v"(kf[3][0]), "+v"(kf[3][KF - 1]));
        else
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kf[0][0]), "+v"(kf[1][0]), "+v"(kf[2][0]), "+v"(kf[3][0]));
        ap_f32x4 st[QG][4];
        // (independent accumulators back to back: the K_lo products of all 4 QG sub-tiles, then the K_hi ones on top)
#pragma unroll
        for (int sub = 0; sub < 4; ++sub)
#pragma unroll
            for (int qg = 0; qg < QG; ++qg)
                st[qg][sub] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, kf[sub][0]), qh[qg], ap_f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        if constexpr (DQ == 16 && NP == 2) {
#pragma unroll
            for (int sub = 0; sub < 4; ++sub)
#pragma unroll
                for (int qg = 0; qg < QG; ++qg)
                    st[qg][sub] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, kf[sub][KF - 1]), qh[qg], st[qg][sub], 0, 0, 0);
        }
        // the V fragments of the first 32 keys are read behind the score MFMAs (whose K fragments they may replace in the register file) and
        // are back before the matrix pipe has drained: nothing stays pending across the softmax
        __builtin_amdgcn_sched_barrier(0);
        MCR_AP_VREAD(vf, 0, 0); MCR_AP_VREAD(vf, 1, 0);
        if constexpr (NT == 4) { MCR_AP_VREAD(vf, NT - 2, 0); MCR_AP_VREAD(vf, NT - 1, 0); }
        MCR_AP_VWAIT(vf);
        // ---- online softmax: lane (qi = li, g) owns S[qi][16 sub + 4 g + r] ----
#pragma unroll
        for (int qg = 0; qg < QG; ++qg) {
            if (t0 + TK > Lk) {                           // only the last tile of a sequence has keys past its end (block-uniform)
                int rem = Lk - t0, g4 = 4 * g;             // (opaque: hipcc otherwise hoists the branch's 32 adds and compares into every tile)
                asm volatile("" : "+s"(rem), "+v"(g4));
#pragma unroll
                for (int sub = 0; sub < 4; ++sub)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (g4 + (sub * 16 + r) >= rem) st[qg][sub][r] = -__builtin_inff();
            }
            float tmax = fmaxf(fmaxf(st[qg][0][0], st[qg][0][1]), st[qg][0][2]);
#pragma unroll
            for (int sub = 0; sub < 4; ++sub)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (sub * 4 + r >= 3) tmax = fmaxf(tmax, st[qg][sub][r]);
            tmax = ap_max_lane_groups(tmax);
            const float m_new = fmaxf(m[qg], tmax * c2);
            const float alpha = __builtin_amdgcn_exp2f(m[qg] - m_new);   // m = -inf on the first tile -> 0 (o = 0 anyway)
            m[qg] = m_new;
            // rescale O and the denominators (rows = queries 4g + r live in lane group g); a wave whose 16 queries all keep their maxima
            // skips the exchange and the multiplications by 1 (same bits)
            if (__any(alpha != 1.f)) {
                float ar[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) ar[r] = __shfl(alpha, 4 * g + r, 64);   // alpha of query 4g + r (any lane group holds it)
#pragma unroll
                for (int r = 0; r < 4; ++r) ol[qg][r] *= ar[r];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[qg][nt][r] *= ar[r];
            }
        }
        // ---- weights and O += P V, 32 keys at a time: p = exp2(s c - m) and its fp16 split for keys 32..63 are issued between the MFMA
        // groups of keys 0..31 -- the matrix pipe works through a group while the vector unit prepares the next operands (the phases of a
        // wave are otherwise strictly serial, and measured costs were additive: matrix + vector + staging).  Two 16-key sub-tiles per
        // MFMA: k index 8 g + r <-> key 4 g + r of sub-tile 2 q (r < 4) or 2 q + 1 (r >= 4)
        f16x8 p_hi[2][QG], p_lo[2][QG];
        auto weights = [&](const int q, const int qg, const int half) {     // half 0 / 1: sub-tile 2 q / 2 q + 1 of query group qg
            const int sub = 2 * q + half;
#ifdef MCR_AP_EXP_NOEXP          // (experiment, wrong results: no exponentials -- prices the transcendental unit)
#pragma unroll
            for (int r = 0; r < 4; ++r) st[qg][sub][r] = fmaf(st[qg][sub][r], c2, -m[qg]);
#else
#pragma unroll
            for (int r = 0; r < 4; ++r) st[qg][sub][r] = __builtin_amdgcn_exp2f(fmaf(st[qg][sub][r], c2, -m[qg]));
#endif
        };
        auto split_q = [&](const int q, const int qg) {
            if constexpr (NP == 1) {                          // rounded once: v_cvt_pk_f16_f32
                const uint4 ph1 = make_uint4(pack2h(st[qg][2 * q][0], st[qg][2 * q][1]), pack2h(st[qg][2 * q][2], st[qg][2 * q][3]),
                                             pack2h(st[qg][2 * q + 1][0], st[qg][2 * q + 1][1]), pack2h(st[qg][2 * q + 1][2], st[qg][2 * q + 1][3]));
                p_hi[q][qg] = __builtin_bit_cast(f16x8, ph1);
                return;
            }
            uint4 ph, pl;
            split2h(st[qg][2 * q][0], st[qg][2 * q][1], ph.x, pl.x);
            split2h(st[qg][2 * q][2], st[qg][2 * q][3], ph.y, pl.y);
            split2h(st[qg][2 * q + 1][0], st[qg][2 * q + 1][1], ph.z, pl.z);
            split2h(st[qg][2 * q + 1][2], st[qg][2 * q + 1][3], ph.w, pl.w);
            p_hi[q][qg] = __builtin_bit_cast(f16x8, ph); p_lo[q][qg] = __builtin_bit_cast(f16x8, pl);
        };
#pragma unroll
        for (int qg = 0; qg < QG; ++qg) { weights(0, qg, 0); weights(0, qg, 1); split_q(0, qg); }
        // first 32 keys, one 16-column block after the other: per accumulator p_lo v_hi, then p_hi v_lo, then p_hi v_hi (smallest terms
        // first).  As soon as a block's MFMAs are issued its fragments of the SECOND 32 keys (+ 1024 bytes inside an [nt] block) are read
        // into the registers they leave: the reads travel under the remaining MFMAs, and the two halves never hold 2 x 32 registers at once
        // (the kernel runs three waves per SIMD at 168 registers)
        auto pv = [&](const int q, const int nt, const ap_u32x2* v) {
            const f16x8 v_hi = __builtin_bit_cast(f16x8, ap_u32x4{v[0][0], v[0][1], v[1][0], v[1][1]});
            if constexpr (NP == 2) {
                const f16x8 v_lo = __builtin_bit_cast(f16x8, ap_u32x4{v[2][0], v[2][1], v[3][0], v[3][1]});
#pragma unroll
                for (int qg = 0; qg < QG; ++qg) o[qg][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(p_lo[q][qg], v_hi, o[qg][nt], 0, 0, 0);
#pragma unroll
                for (int qg = 0; qg < QG; ++qg) o[qg][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(p_hi[q][qg], v_lo, o[qg][nt], 0, 0, 0);
            }
#pragma unroll
            for (int qg = 0; qg < QG; ++qg) o[qg][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(p_hi[q][qg], v_hi, o[qg][nt], 0, 0, 0);
        };
        auto pl_sum = [&](const int q) {
            if constexpr (NP == 2) {
#pragma unroll
                for (int qg = 0; qg < QG; ++qg) ol[qg] = __builtin_amdgcn_mfma_f32_16x16x32_f16(p_lo[q][qg], ones, ol[qg], 0, 0, 0);
            }
#pragma unroll
            for (int qg = 0; qg < QG; ++qg) ol[qg] = __builtin_amdgcn_mfma_f32_16x16x32_f16(p_hi[q][qg], ones, ol[qg], 0, 0, 0);
        };
        // (a scheduling wall after every group: MFMAs, then the vector work that overlaps them, then the fragment reads of keys 32..63)
        pv(0, 0, vf[0]); weights(1, 0, 0); weights(1, 0, 1);
        __builtin_amdgcn_sched_barrier(0); MCR_AP_VREAD(vg, 0, 1024); __builtin_amdgcn_sched_barrier(0);
        pv(0, 1, vf[1]); split_q(1, 0);
        __builtin_amdgcn_sched_barrier(0); MCR_AP_VREAD(vg, 1, 1024); __builtin_amdgcn_sched_barrier(0);
        if constexpr (NT == 4) {
            pv(0, NT - 2, vf[NT - 2]);
            if constexpr (QG == 2) { weights(1, QG - 1, 0); weights(1, QG - 1, 1); }
            __builtin_amdgcn_sched_barrier(0); MCR_AP_VREAD(vg, NT - 2, 1024); __builtin_amdgcn_sched_barrier(0);
            pv(0, NT - 1, vf[NT - 1]);
            if constexpr (QG == 2) split_q(1, QG - 1);
            __builtin_amdgcn_sched_barrier(0); MCR_AP_VREAD(vg, NT - 1, 1024); __builtin_amdgcn_sched_barrier(0);
        } else if constexpr (QG == 2) {
            weights(1, QG - 1, 0); weights(1, QG - 1, 1); split_q(1, QG - 1);
        }
        pl_sum(0);
        pl_sum(1);
        __builtin_amdgcn_sched_barrier(0);                 // (the wait stays behind those MFMAs)
        MCR_AP_VWAIT(vg);
#ifdef MCR_AP_EXP_HALFPV         // (experiment, wrong results: the second 32 keys' products dropped -- prices the matrix pipe)
        asm volatile("" :: "v"(vg[0][0]), "v"(vg[1][0]));
#else
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) pv(1, nt, vg[nt]);
#endif
#undef MCR_AP_VREAD
#undef MCR_AP_VWAIT
    }
    // ---- normalise and write: lane (c = li, g) owns O[q0 + 16 qg + 4g + r][nt*16 + li] and the denominator of that query ----
#pragma unroll
    for (int qg = 0; qg < QG; ++qg) {
        const int qbase = q0 + 16 * qg;
        if (SPLIT) {
            float* dst = part ? part1 : out;
            const long long ld = part ? (long long)H * DV : ldo;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qi = qbase + 4 * g + r;
                const float mq = __shfl(m[qg], 4 * g + r, 64);   // the running max of query 4g + r lives with lane li = 4g + r
                if (qi < L) {
                    if (li == 0) {
                        float* p = ml + (((long long)part * gridDim.z / 2 * L + seq0 + qi) * H + hh) * 2;
                        p[0] = mq; p[1] = ol[qg][r];
                    }
                    float* orow = dst + (seq0 + qi) * ld + hh * DV;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) orow[nt * 16 + li] = o[qg][nt][r];
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float ir = 1.0f / ol[qg][r];
                const int qi = qbase + 4 * g + r;
                if (qi < L) {
                    const long long at = (seq0 + qi) * ldo + hh * DV + li;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const float y = o[qg][nt][r] * ir;
                        if (Oh) {
                            const _Float16 hi = (_Float16)y;
                            Oh[at + nt * 16] = hi;
                            if (NP == 2) Ol[at + nt * 16] = (_Float16)(y - (float)hi);
                        } else {
                            out[at + nt * 16] = y;
                        }
                    }
                }
            }
        }
    }
}

bool attention_planes_applicable(int H, int DQK, int DV, int64_t ldp) {
    const int dq = DQK / H, dv = DV / H;
    return ((dq == 16 && dv == 64) || (dq == 8 && dv == 32)) && ldp % 8 == 0 && (H * dq) % 8 == 0;
}

// See nn_kernels.h.  split_ws: attention_split_floats(S, L, H, DV) floats, used when the keys of a sequence are split over two blocks
// (out then receives fp32 parts and the combine pass writes the planes Oh / Ol, or fp32 rows of `out` when Oh is null).
void launch_attention_planes(hipStream_t s, const void* Ph_, const void* Pl_, int64_t ldp, float* out, int64_t ldo, void* Oh_, void* Ol_,
                             int64_t ldoh, int64_t S, int L, int H, int DQK, int DV, const int* lens, float* split_ws, size_t split_ws_floats,
                             int split_mode, int n_planes) {
    if (S <= 0 || L <= 0) return;
    const int dq = DQK / H, dv = DV / H;
    if ((int64_t)L * ldp >= ((int64_t)1 << 31)) { set_error("launch_attention_planes: sequence too long for 32-bit row offsets"); return; }
    if (!attention_planes_applicable(H, DQK, DV, ldp) || (reinterpret_cast<uintptr_t>(Ph_) & 15) || (n_planes != 1 && (reinterpret_cast<uintptr_t>(Pl_) & 15))) {
        set_error("launch_attention_planes: unsupported head dims / alignment (dq=%d dv=%d ldp=%lld)", dq, dv, (long long)ldp);
        return;
    }
    const _Float16 *Ph = (const _Float16*)Ph_, *Pl = (const _Float16*)Pl_;
    _Float16 *Oh = (_Float16*)Oh_, *Ol = (_Float16*)Ol_;
    // split_mode: 1 = always (when L >= 512 and the scratch is there), 0 = never, -1 = when the unsplit grid leaves CUs idle
    const int64_t blocks64 = (int64_t)cdiv(L, 64) * H * S;
    const bool can_split = n_planes != 1 && split_ws && split_ws_floats >= attention_split_floats(S, L, H, DV) && L >= 512 && 2 * S <= 65535;   // (single-plane form: never split)
    const bool split = can_split && (split_mode == 1 || (split_mode < 0 && blocks64 <= 256));
    const unsigned gz = (unsigned)(split ? 2 * S : S);
    const bool qg2 = (int64_t)cdiv(L, 128) * H * gz >= 512;
    const dim3 grid((unsigned)cdiv(L, qg2 ? 128 : 64), (unsigned)H, gz);
    float* part1 = split ? split_ws : nullptr;
    float* ml = split ? split_ws + (size_t)S * L * DV : nullptr;
#define MCR_AP(DQ_, DV_, SPLIT_, QG_)                                                                                                  \
    hipLaunchKernelGGL((attention_planes_kernel<DQ_, DV_, SPLIT_, QG_>), grid, dim3(256), 0, s, Ph, Pl, (long long)ldp, out,            \
                       SPLIT_ ? (_Float16*)nullptr : Oh, SPLIT_ ? (_Float16*)nullptr : Ol, (long long)(SPLIT_ || !Oh ? ldo : ldoh), L, H, \
                       lens, part1, ml)
    if (n_planes == 1) {                                  // variant 7: the high planes alone (never the key-split form)
#define MCR_AP1(DQ_, DV_, QG_)                                                                                                         \
    hipLaunchKernelGGL((attention_planes_kernel<DQ_, DV_, false, QG_, 1>), grid, dim3(256), 0, s, Ph, Pl, (long long)ldp, out, Oh,       \
                       (_Float16*)nullptr, (long long)(!Oh ? ldo : ldoh), L, H, lens, part1, ml)
        if (dq == 16) { if (qg2) MCR_AP1(16, 64, 2); else MCR_AP1(16, 64, 1); }
        else { if (qg2) MCR_AP1(8, 32, 2); else MCR_AP1(8, 32, 1); }
#undef MCR_AP1
    } else if (dq == 16) {
        if (split) { if (qg2) MCR_AP(16, 64, true, 2); else MCR_AP(16, 64, true, 1); }
        else { if (qg2) MCR_AP(16, 64, false, 2); else MCR_AP(16, 64, false, 1); }
    } else {
        if (split) { if (qg2) MCR_AP(8, 32, true, 2); else MCR_AP(8, 32, true, 1); }
        else { if (qg2) MCR_AP(8, 32, false, 2); else MCR_AP(8, 32, false, 1); }
    }
#undef MCR_AP
    if (split) launch_attention_combine(s, out, ldo, part1, ml, S * L, H, dv, Oh_, Ol_, ldoh);
}

}  // namespace mcr
