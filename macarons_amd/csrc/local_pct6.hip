// K5-local v6 (default) — the structure of local_pct5.hip (two workgroups per CU, residual stream in registers, LayerNorm
// outputs pre-split in LDS, one n-tile column per wave) on a TWO-TERM FP16 SPLIT: three MFMAs per fp32 product instead of
// six, and a split that costs ~3 instead of 5.5 vector instructions per element.
//
// Numerics.  Every fp32 operand x is carried as hi = fp16(x), lo = fp16(x - hi): 22 significant bits, |x - hi - lo| <=
// max(2^-22 |x|, 2^-25).  A product a w is evaluated as a_lo w_hi + a_hi w_lo + a_hi w_hi on v_mfma_f32_32x32x16_f16
// (fp16 x fp16 products are exact in the fp32 accumulator); the dropped a_lo w_lo term is <= 2^-22 |a w|.  Weights are
// multiplied by a per-matrix power of two on the host (their low plane then stays a normal fp16 number) and the exact
// inverse is applied in the epilogue FMA that adds the bias.  Measured against the fp64 oracle: see DESIGN.md / the
// test test_fused_local_transformer[6].  Range: |activation| < 65504 (LayerNorm outputs are <= sqrt(128); variant 5
// covers the whole fp32 range).
//
// LDS = 64 KB, both tiles XOR-swizzled in 16-byte chunks by (row & 15):
//           P  32 KB  x^ as planes [2][64 rows][16 chunks of 8 fp16], or q|k as fp32 [64][64] during attention
//           F  32 KB  fp32 [64][128]: raw x for the LayerNorm / v and the attention output / a half of the FF hidden
// Reference mapping in local_pct.hip (SconeOcc.py:104-130).
#include "lp_split.h"

namespace mcr {
namespace v6 {

// ---- swizzled addressing ----------------------------------------------------------------------------------------------
__device__ __forceinline__ int f_idx(int row, int col) { return row * 128 + ((((col >> 2) ^ (row & 15)) << 2) | (col & 3)); }
__device__ __forceinline__ int f_chunk(int row, int chunk) { return row * 128 + ((chunk ^ (row & 15)) << 2); }     // float index of a float4
__device__ __forceinline__ int q_idx(int row, int col) { return row * 64 + ((((col >> 2) ^ (row & 15)) << 2) | (col & 3)); }
__device__ __forceinline__ int q_chunk(int row, int chunk) { return row * 64 + ((chunk ^ (row & 15)) << 2); }
__device__ __forceinline__ int p_chunk(int plane, int row, int chunk) { return (plane * 64 + row) * 16 + (chunk ^ (row & 15)); }   // uint4 index

// The swizzled addresses are loop-invariant functions of the lane; left alone, LLVM hoists all of them out of the encoder
// loop and spills them (172 scratch stores).  Re-deriving them from an opaque copy of the lane id per phase is cheaper.
__device__ __forceinline__ int l6_opaque(int v) {
    asm volatile("" : "+v"(v));
    return v;
}

#ifndef L6_PF_N
#define L6_PF_N 3
#endif
constexpr int L6_PF = L6_PF_N;                    // k16-steps of weights in flight per wave

// acc[u][mt] (+)= A[64 x 16 S] * W^T; wave owns n-tiles {nt0 + 4u} for both m-tiles (see local_pct4.hip).
// PLANES: A comes pre-split from P (uint4 planes); else fp32 from F, split here.
template <int S, int NTW, bool INIT, bool PLANES>
__device__ __forceinline__ void l6_gemm(f32x16 (&acc)[NTW][2], const void* __restrict__ Asrc, const float* __restrict__ Wp,
                                        int nt0, int lane_) {
    const int lane = l6_opaque(lane_);
    const int i = lane & 31, h = lane >> 5, key = i & 15;
    const uint4* bp[NTW];
#pragma unroll
    for (int u = 0; u < NTW; ++u) {
        bp[u] = reinterpret_cast<const uint4*>(Wp) + (size_t)(nt0 + 4 * u) * S * 2 * 64 + lane;
        if (INIT) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[u][mt][r] = 0.f;
        }
    }
    constexpr int PF = S < L6_PF ? S : L6_PF;
    uint4 b[PF][NTW][2];
#pragma unroll
    for (int p = 0; p < PF; ++p)
#pragma unroll
        for (int u = 0; u < NTW; ++u)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) b[p][u][pl] = bp[u][((p * 2 + pl) * 64)];
    const float* F = reinterpret_cast<const float*>(Asrc);
    const uint4* P = reinterpret_cast<const uint4*>(Asrc);
    float4 ra[2][2];                                        // fp32 mode: raw A rows of the next step
    Split2 sn[2];                                           // planes mode: fragments of the next step
    auto fetch = [&](int s) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            if (PLANES) {
                const int c = (2 * s + h) ^ key;
                sn[mt].hi = P[(0 * 64 + mt * 32 + i) * 16 + c];
                sn[mt].lo = P[(1 * 64 + mt * 32 + i) * 16 + c];
            } else {
                const float* row = F + (mt * 32 + i) * 128;
                ra[mt][0] = *reinterpret_cast<const float4*>(row + (((4 * s + 2 * h) ^ key) << 2));
                ra[mt][1] = *reinterpret_cast<const float4*>(row + (((4 * s + 2 * h + 1) ^ key) << 2));
            }
        }
    };
    fetch(0);
#pragma unroll
    for (int s = 0; s < S; ++s) {
        Split2 sa[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) sa[mt] = PLANES ? sn[mt] : split8h(ra[mt][0], ra[mt][1]);
        if (s + 1 < S) fetch(s + 1);
        uint4 bc[NTW][2];
#pragma unroll
        for (int u = 0; u < NTW; ++u)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) bc[u][pl] = b[s % PF][u][pl];
        if (s + PF < S) {
#pragma unroll
            for (int u = 0; u < NTW; ++u)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) b[s % PF][u][pl] = bp[u][(((s + PF) * 2 + pl) * 64)];
        }
#pragma unroll
        for (int u = 0; u < NTW; ++u)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                acc[u][mt] = mfma_h(sa[mt].lo, bc[u][0], acc[u][mt]);       // smallest terms first
                acc[u][mt] = mfma_h(sa[mt].hi, bc[u][1], acc[u][mt]);
                acc[u][mt] = mfma_h(sa[mt].hi, bc[u][0], acc[u][mt]);
            }
    }
}

// Write one 64 x 32 column block held as C fragments (t[0], t[1] = the two m-tiles; lane (j, h) owns column j and rows
// mt*32 + (r&3) + 8(r>>2) + 4h) into a swizzled fp32 tile with row stride LD at column tile ct, through g(value).
// row & 15 = Kc(r) | 4h with Kc(r) = (r&3) + 8((r>>2)&1), so the swizzle splits into a per-lane word w and a compile-time
// XOR: one v_xor per element, the row part goes into the ds_write offset field.
template <int LD, class G>
__device__ __forceinline__ void l6_put(const f32x16 (&t)[2], float* buf, int ct, int lane_, G g) {
    const int lane = l6_opaque(lane_);
    const int j = lane & 31, h = lane >> 5;
    const int w = ((((ct << 3) | (j >> 2)) ^ (h << 2)) << 2) | (j & 3) | (h * 4 * LD);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            buf[(mt * 32 + (r & 3) + 8 * (r >> 2)) * LD + (w ^ ((((r & 3) + 8 * ((r >> 2) & 1))) << 2))] = g((float)t[mt][r]);
}

// QKV product (N = 192 = 6 n-tiles over 4 waves), balanced: every wave takes its own n-tile w for both m-tiles plus HALF
// of n-tile 4 + (w >> 1): the m-tile w & 1.  (Giving waves 0,1 two whole n-tiles and waves 2,3 one made this the longest
// phase of an encoder: 4 tile-units of matrix work on the critical path instead of 3.)  The two waves sharing n-tile 4 or
// 5 are not in lock-step on it -- one starts with its own tile's fragments in flight -- so their requests do not collide in
// the L1.  A comes pre-split from P.
__device__ __forceinline__ void l6_gemm_qkv(f32x16 (&acc)[1][2], f32x16& acch, const uint4* __restrict__ P,
                                            const float* __restrict__ Wp, int wave, int lane_) {
    constexpr int S = 8, PF = L6_PF;
    const int lane = l6_opaque(lane_);
    const int i = lane & 31, h = lane >> 5, key = i & 15, mh = wave & 1;
    const uint4* bp[2] = {reinterpret_cast<const uint4*>(Wp) + (size_t)wave * S * 2 * 64 + lane,
                          reinterpret_cast<const uint4*>(Wp) + (size_t)(4 + (wave >> 1)) * S * 2 * 64 + lane};
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[0][0][r] = 0.f; acc[0][1][r] = 0.f; acch[r] = 0.f; }
    uint4 b[PF][2][2];
#pragma unroll
    for (int p = 0; p < PF; ++p)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) b[p][u][pl] = bp[u][((p * 2 + pl) * 64)];
    Split2 sn[3];                                           // fragments of the next step: m-tile 0, m-tile 1, m-tile mh again
    auto fetch = [&](int s) {
        const int c = (2 * s + h) ^ key;
#pragma unroll
        for (int f = 0; f < 3; ++f) {
            const int row = (f < 2 ? f : mh) * 32 + i;
            sn[f].hi = P[(0 * 64 + row) * 16 + c];
            sn[f].lo = P[(1 * 64 + row) * 16 + c];
        }
    };
    fetch(0);
#pragma unroll
    for (int s = 0; s < S; ++s) {
        Split2 sa[3];
#pragma unroll
        for (int f = 0; f < 3; ++f) sa[f] = sn[f];
        if (s + 1 < S) fetch(s + 1);
        uint4 bc[2][2];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) bc[u][pl] = b[s % PF][u][pl];
        if (s + PF < S) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) b[s % PF][u][pl] = bp[u][(((s + PF) * 2 + pl) * 64)];
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            acc[0][mt] = mfma_h(sa[mt].lo, bc[0][0], acc[0][mt]);
            acc[0][mt] = mfma_h(sa[mt].hi, bc[0][1], acc[0][mt]);
            acc[0][mt] = mfma_h(sa[mt].hi, bc[0][0], acc[0][mt]);
        }
        acch = mfma_h(sa[2].lo, bc[1][0], acch);
        acch = mfma_h(sa[2].hi, bc[1][1], acch);
        acch = mfma_h(sa[2].hi, bc[1][0], acch);
    }
}

// l6_put for ONE 32 x 32 C fragment: m-tile mt (wave-uniform) of column tile ct
template <int LD, class G>
__device__ __forceinline__ void l6_put_half(const f32x16& t, float* buf, int ct, int mt, int lane_, G g) {
    const int lane = l6_opaque(lane_);
    const int j = lane & 31, h = lane >> 5;
    const int w = (((((ct << 3) | (j >> 2)) ^ (h << 2)) << 2) | (j & 3) | (h * 4 * LD)) + mt * 32 * LD;
#pragma unroll
    for (int r = 0; r < 16; ++r)
        buf[((r & 3) + 8 * (r >> 2)) * LD + (w ^ ((((r & 3) + 8 * ((r >> 2) & 1))) << 2))] = g((float)t[r]);
}

// All-reduce over the four 16-lane rows of a wave (lanes l, l^16, l^32, l^48) without the LDS crossbar: gfx950's
// v_permlane16_swap / v_permlane32_swap exchange rows / halves between two registers; fed the same value twice they return
// (this row pair's even row | odd row) resp. (lower half | upper half) replicated, so one op on the pair is the reduction.
template <class Op>
__device__ __forceinline__ float l6_rows_allreduce(float v, Op op) {
    const unsigned b = __builtin_bit_cast(unsigned, v);
    const auto r = __builtin_amdgcn_permlane16_swap(b, b, false, false);
    v = op(__builtin_bit_cast(float, (unsigned)r[0]), __builtin_bit_cast(float, (unsigned)r[1]));
    const unsigned c = __builtin_bit_cast(unsigned, v);
    const auto q = __builtin_amdgcn_permlane32_swap(c, c, false, false);
    return op(__builtin_bit_cast(float, (unsigned)q[0]), __builtin_bit_cast(float, (unsigned)q[1]));
}

// LayerNorm (eps 1e-5, affine folded into the next weights) of the 64 rows of F, written as fp16 hi/lo planes into P:
// 4 threads per row, 32 columns each; two-pass (mean, then centred variance) like torch's.
__device__ __forceinline__ void l6_norm(const float* F, uint4* P, int tid_) {
    const int tid = l6_opaque(tid_);
    const int row = tid >> 2, part = tid & 3;
    float v[32];
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const float4 q = *reinterpret_cast<const float4*>(F + f_chunk(row, part * 8 + c));
        v[4 * c] = q.x; v[4 * c + 1] = q.y; v[4 * c + 2] = q.z; v[4 * c + 3] = q.w;
        sum += (q.x + q.y) + (q.z + q.w);
    }
    sum += dpp_mov0<0xB1>(sum);          // quad_perm [1,0,3,2]: the row's 4 threads are one quad (DPP, not an LDS bpermute)
    sum += dpp_mov0<0x4E>(sum);          // quad_perm [2,3,0,1]
    const float mu = sum * (1.0f / 128.f);
    float sq = 0.f;
#pragma unroll
    for (int c = 0; c < 32; ++c) {
        v[c] -= mu;
        sq = fmaf(v[c], v[c], sq);
    }
    sq += dpp_mov0<0xB1>(sq);
    sq += dpp_mov0<0x4E>(sq);
    const float rstd = 1.0f / sqrtf(sq * (1.0f / 128.f) + 1e-5f);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const Split2 s = split8h(make_float4(v[8 * c] * rstd, v[8 * c + 1] * rstd, v[8 * c + 2] * rstd, v[8 * c + 3] * rstd),
                                make_float4(v[8 * c + 4] * rstd, v[8 * c + 5] * rstd, v[8 * c + 6] * rstd, v[8 * c + 7] * rstd));
        P[p_chunk(0, row, part * 4 + c)] = s.hi;
        P[p_chunk(1, row, part * 4 + c)] = s.lo;
    }
}

// grid = ceil(S / 4); S sequences of 16 offsets [S,16,3]; features[s*ld_feat + 0:256] = max(128) || avg(128)
__global__ __launch_bounds__(256, 2) void local_pct6_kernel(const float* __restrict__ offs, float* __restrict__ feat,
                                                          long long ld_feat, long long S,
                                                          const float* __restrict__ blob) {
    __shared__ __attribute__((aligned(16))) uint4 P[2 * 64 * 16];
    __shared__ __attribute__((aligned(16))) float F[64 * 128];
    float* Pq = reinterpret_cast<float*>(P);               // fp32 [64][64] view: q | k, and the raw xyz during the embedding
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* mats = blob;
    const float* vecs = blob + L6_MATS_TOTAL;
    const float* isc = vecs + L6_SCALES;                   // 2^-e of every matrix (the host stores W * 2^e)
    const long long s0 = (long long)blockIdx.x * L3_QPB;

    // ---- stage the 64 x 3 offsets, zero-padded to K = 16, into F[:, 0:16]; a copy of xyz into Pq[:, 0:4] ----
    if (tid < L3_T) {
        const long long seq = s0 + (tid >> 4);
        float x = 0.f, y = 0.f, z = 0.f;
        if (seq < S) {
            const float* p = offs + (seq * 16 + (tid & 15)) * 3;
            x = p[0]; y = p[1]; z = p[2];
        }
        *reinterpret_cast<float4*>(F + f_chunk(tid, 0)) = make_float4(x, y, z, 0.f);
        *reinterpret_cast<float4*>(F + f_chunk(tid, 1)) = make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(F + f_chunk(tid, 2)) = make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(F + f_chunk(tid, 3)) = make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(Pq + q_chunk(tid, 0)) = make_float4(x, y, z, 0.f);
    }
    __syncthreads();
    f32x16 acc[1][2], xres[1][2];
    // ---- Embedding (Attention.py:98-128): linear1 3->125, GELU -> F ; linear2 125->125 || xyz -> xres ----
    l6_gemm<1, 1, true, false>(acc, F, mats + l6_mat_off(0), wave, lane);
    __syncthreads();                               // the offsets in F[:, 0:16] are consumed
    {
        const float b = vecs[L3_VEC_EMB1 + wave * 32 + (lane & 31)];
        const float sc = isc[0];
        l6_put<128>(acc[0], F, wave, lane, [&](float v) { return l3_gelu(fmaf(v, sc, b)); });
    }
    __syncthreads();
    l6_gemm<8, 1, true, false>(xres, F, mats + l6_mat_off(1), wave, lane);
    {
        const int j = lane & 31, h = lane >> 5;
        const float b = vecs[L3_VEC_EMB2 + wave * 32 + j], sc = isc[1];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) xres[0][mt][r] = fmaf(xres[0][mt][r], sc, b);
        if (wave == 3 && j >= 29) {                // columns 125..127: concat the raw xyz (Attention.py:123-126)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) xres[0][mt][r] = Pq[q_idx(mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, j - 29)];
        }
    }
    __syncthreads();                               // every wave is done reading F as the A operand
    l6_put<128>(xres[0], F, wave, lane, [](float v) { return v; });

#pragma unroll 1
    for (int e = 0; e < 2; ++e) {
        const float* em = mats + l6_mat_off(2 + 6 * e);
        const float* ev = vecs + L3_VEC_ENC0 + e * L3_VEC_ENC_STRIDE;
        const float* es = isc + 2 + 6 * e;          // qkv, out, ff1a, ff1b, ff2a (= ff2b)
        // ---- norm1 (folded) -> planes ; QKV (Attention.py:186-188, 287): q|k -> Pq, v -> F ----
        // 6 n-tiles over 4 waves: one each plus half of n-tile 4 or 5 (l6_gemm_qkv).
        __syncthreads();
        l6_norm(F, P, tid);
        __syncthreads();
        {
            f32x16 aq[1][2], ah;
            l6_gemm_qkv(aq, ah, P, em, wave, lane);
            __syncthreads();                       // x^ planes consumed: P may take q|k (F's raw x died with the norm)
            // n-tiles 0,1 = q,k -> Pq column tiles 0,1; n-tiles 2..5 = v -> F column tiles 0..3
            {
                const float b0 = ev[wave * 32 + (lane & 31)], sc = es[0];
                if (wave < 2) l6_put<64>(aq[0], Pq, wave, lane, [&](float v) { return fmaf(v, sc, b0); });
                else l6_put<128>(aq[0], F, wave - 2, lane, [&](float v) { return fmaf(v, sc, b0); });
                const int nth = 4 + (wave >> 1);
                const float b1 = ev[nth * 32 + (lane & 31)];
                l6_put_half<128>(ah, F, nth - 2, wave & 1, lane, [&](float v) { return fmaf(v, sc, b1); });
            }
        }
        __syncthreads();
        // ---- attention (Attention.py:8-36) on the fp32 matrix pipe (v_mfma_f32_16x16x4_f32 = exact fp32 fma chains):
        // wave = query (16 tokens); per head  S^T = K Q^T (j rows, query rows qi as columns: lane (qi, g) then owns
        // S[qi][4g..4g+3]), softmax over j = 4 registers x the 4 lane groups, O = P V with the k index j = 4g + s so the
        // probabilities are used as the A operand straight from their registers.  The output overwrites the head's V block.
        {
            typedef float f32x4 __attribute__((ext_vector_type(4)));
            const int ln = l6_opaque(lane);
            const int r0 = wave * 16, li = ln & 15, g = ln >> 4;
            // swizzled addresses as one per-lane word XOR a compile-time constant (row & 15 = li for q|k, 4g + s for v):
            const int wq = ((r0 + li) * 64) | (li << 2) | g;                                   // Pq[row = r0+li][chunk c][g]   = wq ^ (c << 2)
            const int wv = ((r0 + 4 * g) * 128) | ((((li >> 2) | (g << 2)) << 2)) | (li & 3);    // F[row = r0+4g+s][chunk C][li&3] = s*128 + (wv ^ ((C ^ s) << 2))
            f32x4 pr[4];
#pragma unroll
            for (int hh = 0; hh < 4; ++hh) {
                f32x4 st = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int sk = 0; sk < 2; ++sk) {
                    const float kk = Pq[wq ^ ((8 + 2 * hh + sk) << 2)];                  // A[i = j][k = d]      = k[j][hh*8 + 4 sk + g]
                    const float qq = Pq[wq ^ ((2 * hh + sk) << 2)];                      // B[k = d][n = qi]     = q[qi][hh*8 + 4 sk + g]
                    st = __builtin_amdgcn_mfma_f32_16x16x4f32(kk, qq, st, 0, 0, 0);
                }
                float mx = fmaxf(fmaxf(st[0], st[1]), fmaxf(st[2], st[3]));
                mx = l6_rows_allreduce(mx, [](float a, float b) { return fmaxf(a, b); });
                mx *= 0.35355339059327376220f;                                           // scores / sqrt(8)
                float den = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    st[r] = __expf(fmaf(st[r], 0.35355339059327376220f, -mx));
                    den += st[r];
                }
                den = l6_rows_allreduce(den, [](float a, float b) { return a + b; });
                const float inv = __builtin_amdgcn_rcpf(den);
#pragma unroll
                for (int r = 0; r < 4; ++r) pr[hh][r] = st[r] * inv;
            }
            f32x4 o[4][2];
#pragma unroll
            for (int hh = 0; hh < 4; ++hh)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    o[hh][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int sk = 0; sk < 4; ++sk) {
                        const float vv = F[sk * 128 + (wv ^ (((hh * 8 + nt * 4) ^ sk) << 2))];   // B[k = g][n = c] = V[j = 4g + sk][hh*32 + nt*16 + li]
                        o[hh][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(pr[hh][sk], vv, o[hh][nt], 0, 0, 0);
                    }
                }
            // every V read of this wave's 16 rows precedes the writes (other waves own other rows)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int hh = 0; hh < 4; ++hh)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) F[r * 128 + (wv ^ (((hh * 8 + nt * 4) ^ r) << 2))] = o[hh][nt][r];
        }
        __syncthreads();
        // ---- out projection + residual (Attention.py:201-202, 290): x += att W_o^T + b ----
        l6_gemm<8, 1, true, false>(acc, F, em + L6_MAT_QKV, wave, lane);
        {
            const float b = ev[192 + wave * 32 + (lane & 31)], sc = es[1];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) xres[0][mt][r] += fmaf(acc[0][mt][r], sc, b);
        }
        __syncthreads();                           // the attention output in F is consumed
        l6_put<128>(xres[0], F, wave, lane, [](float v) { return v; });
        __syncthreads();
        // ---- norm2 (folded) -> planes ; FF 128 -> 256 (GELU) -> 128 + residual (Attention.py:293-298), two halves ----
        l6_norm(F, P, tid);
        __syncthreads();
        f32x16 accf[1][2];
        const float* w_ff1a = em + L6_MAT_QKV + L6_MAT_128 * 1;
        const float* w_ff1b = em + L6_MAT_QKV + L6_MAT_128 * 2;
        const float* w_ff2a = em + L6_MAT_QKV + L6_MAT_128 * 3;
        const float* w_ff2b = em + L6_MAT_QKV + L6_MAT_128 * 4;
        l6_gemm<8, 1, true, true>(acc, P, w_ff1a, wave, lane);
        {
            const float b = ev[192 + 128 + wave * 32 + (lane & 31)], sc = es[2];
            l6_put<128>(acc[0], F, wave, lane, [&](float v) { return l3_gelu(fmaf(v, sc, b)); });
        }
        __syncthreads();
        l6_gemm<8, 1, true, false>(accf, F, w_ff2a, wave, lane);
        l6_gemm<8, 1, true, true>(acc, P, w_ff1b, wave, lane);          // reads P only: no barrier needed before it
        __syncthreads();                           // the first hidden half in F is consumed
        {
            const float b = ev[192 + 128 + 128 + wave * 32 + (lane & 31)], sc = es[3];
            l6_put<128>(acc[0], F, wave, lane, [&](float v) { return l3_gelu(fmaf(v, sc, b)); });
        }
        __syncthreads();
        l6_gemm<8, 1, false, false>(accf, F, w_ff2b, wave, lane);
        {
            const float b = ev[192 + 128 + 256 + wave * 32 + (lane & 31)], sc = es[4];    // ff2a and ff2b share one exponent
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) xres[0][mt][r] += fmaf(accf[0][mt][r], sc, b);
        }
        __syncthreads();                           // the second hidden half in F is consumed
        l6_put<128>(xres[0], F, wave, lane, [](float v) { return v; });
    }
    // ---- final norm (folded) + linear0 128 -> 128 (SconeOcc.py:119-122) ----
    __syncthreads();
    l6_norm(F, P, tid);
    __syncthreads();
    l6_gemm<8, 1, true, true>(acc, P, mats + l6_mat_off(14), wave, lane);
    {
        const float b = vecs[L3_VEC_LIN0 + wave * 32 + (lane & 31)], sc = isc[14];
        l6_put<128>(acc[0], F, wave, lane, [&](float v) { return fmaf(v, sc, b); });
    }
    __syncthreads();
    // ---- max || avg pool over the 16 tokens of each query (SconeOcc.py:124-126) ----
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int o = tid + r * 256, q = o >> 7, c = o & 127;
        if (s0 + q < S) {
            float mx = -__builtin_inff(), sm = 0.f;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const float v = F[f_idx(q * 16 + j, c)];
                mx = fmaxf(mx, v);
                sm += v;
            }
            feat[(s0 + q) * ld_feat + c] = mx;
            feat[(s0 + q) * ld_feat + 128 + c] = sm * (1.0f / 16.f);
        }
    }
}

}  // namespace v6

void launch_local_pct6(hipStream_t s, const float* offs, float* feat, int64_t ld_feat, int64_t S, const float* blob) {
    if (S <= 0) return;
    hipLaunchKernelGGL(v6::local_pct6_kernel, dim3((unsigned)cdiv(S, L3_QPB)), dim3(256), 0, s, offs, feat, (long long)ld_feat,
                       (long long)S, blob);
}

int local_pct6_blob_floats() { return L6_BLOB_FLOATS; }

}  // namespace mcr
