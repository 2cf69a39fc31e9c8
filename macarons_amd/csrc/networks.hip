// SconeVis.forward / PCTransformer.forward / SconeOcc.forward composed from the gfx950 building blocks
// (nn_kernels.hip, knn.hip) behind single C-ABI entry points (include/macarons_hip.h).
//
// Reference (file:line, upstream tree):
//   macarons/networks/Attention.py:98-128   Embedding.forward      (linear-GELU-linear, optional cloud max, concat x)
//   macarons/networks/Attention.py:278-300  Encoder.forward        (pre-LN MHSA + residual, pre-LN FF + residual)
//   macarons/networks/SconeVis.py:121-162   SconeVis.forward
//   macarons/networks/SconeOcc.py:104-130   PCTransformer.forward
//   macarons/networks/SconeOcc.py:250-347   SconeOcc.forward
// Only the reference's default architecture hyper-parameters are implemented on the HIP path (every call site
// uses them); the Python host classes refuse other configurations loudly.
#include "nn_kernels.h"
#include <cstdlib>
#include <algorithm>
#include <map>
#include <mutex>
#include <utility>

namespace mcr {

// ---- weight tables (arrays of device pointers, order documented in include/macarons_hip.h) ------------------
struct LinW { const float* w; const float* b; };
struct EncW {                       // 12 pointers
    const float *n1g, *n1b;         // norm1
    LinW qkv;                       // rows of w_q, w_k, w_v stacked: [2*dqk + dv, E]
    LinW out;
    const float *n2g, *n2b;         // norm2
    LinW ff1, ff2;
    // optional (NULL: split per call): the four weight matrices as fp16 hi/lo planes [2][N][K] of W * 2^8, built by the host once per
    // parameter version (networks/packing.py: encoder_weight_planes) -- pointers 4 per encoder appended to the weight table
    const void *p_qkv = nullptr, *p_out = nullptr, *p_ff1 = nullptr, *p_ff2 = nullptr;
};
static void read_enc_planes(const float* const*& p, EncW& e) {
    e.p_qkv = *p++; e.p_out = *p++; e.p_ff1 = *p++; e.p_ff2 = *p++;
}
static EncW read_enc(const float* const*& p) {
    EncW e;
    e.n1g = *p++; e.n1b = *p++;
    e.qkv.w = *p++; e.qkv.b = *p++;
    e.out.w = *p++; e.out.b = *p++;
    e.n2g = *p++; e.n2b = *p++;
    e.ff1.w = *p++; e.ff1.b = *p++;
    e.ff2.w = *p++; e.ff2.b = *p++;
    return e;
}

// bump allocator over the caller's workspace (256-B aligned blocks)
struct Arena {
    char* base; size_t cap, off;
    float* f(size_t n_floats) {
        size_t bytes = (n_floats * sizeof(float) + 255) & ~(size_t)255;
        float* p = reinterpret_cast<float*>(base + off);
        off += bytes;
        return p;
    }
    bool ok() const { return off <= cap; }
};
static size_t al(size_t n_floats) { return (n_floats * sizeof(float) + 255) & ~(size_t)255; }

// 1: local_pct.hip exact-fp32 MFMA; 5: local_pct5.hip split-precision bf16 hi/mid/lo (6 MFMAs per product, whole fp32
// range); 6 (default): local_pct6.hip two-term fp16 split (3 MFMAs per product); 7 (OPT-IN, per call only -- never a process
// default): local_pct7.hip, ONE fp16 plane per matrix operand (1 MFMA per product), BASELINE.json config 3's 16-bit matrix
// path with its own stated tolerance (tests/test_variant7_gpu.py).  Each has its own blob format.
#ifdef MCR_DEV_LOCAL_PCT8      // dev builds only (tools/build_variant.py with tools/experiments/local_pct8.hip): the shelved register-resident kernel as variant 8
void launch_local_pct8(hipStream_t s, const float* offs, float* feat, int64_t ld_feat, int64_t S, const float* blob);
#define MCR_VARIANT8_OK(v) ((v) == 8)
#else
#define MCR_VARIANT8_OK(v) false
#endif
// The numerics variant is a property of a CALL, not of the process: every network entry point opens a VariantScope, which fixes the
// variant of that call on the calling thread -- the one-shot value mcr_call_variant(v) left for it (the per-call argument of the ABI:
// "the next network entry point on this thread runs on v"), else the process default (mcr_set_local_pct_variant / env
// MCR_LOCAL_PCT_VARIANT).  Two host threads, or two models on different variants, cannot see each other's choice, and a failed
// call cannot leave a flipped switch behind (round 4 flipped a process-global around the range-guard fallback).
static int g_default_variant = []() {                   // env MCR_LOCAL_PCT_VARIANT picks the start-up value (testing: whole suites on a variant)
    const char* e = getenv("MCR_LOCAL_PCT_VARIANT");
    const int v = e ? atoi(e) : 6;
    return (v == 1 || v == 5 || v == 6) ? v : 6;
}();
static thread_local int t_next_variant = 0;             // one-shot: consumed by the next VariantScope of this thread
static thread_local int t_variant = 0;                  // the variant of the entry point running on this thread
static thread_local int t_scope_depth = 0;              // (an entry point may call another: the outermost scope decides)
struct VariantScope {
    VariantScope() {
        if (t_scope_depth++ == 0) {
            t_variant = t_next_variant ? t_next_variant : g_default_variant;
            t_next_variant = 0;
        }
    }
    ~VariantScope() {
        if (--t_scope_depth == 0) t_variant = 0;
    }
};
#define g_local_pct_variant (t_variant ? t_variant : g_default_variant)
// Variant 7 shares variant 6's routes (operands as fp16 planes in HBM, the same range guard); what differs is how many planes a
// product multiplies: matrix_planes() = 1 on variant 7 (the high planes alone; low planes are neither written nor read where
// the single-plane kernels exist: the local transformers and the SconeOcc head), 2 on variant 6.
static inline bool fp16_planes_variant() { return g_local_pct_variant == 6 || g_local_pct_variant == 7; }
static inline int matrix_planes() { return g_local_pct_variant == 7 ? 1 : 2; }
// rows-per-sequence argument of launch_linear for the encoders of the 2048-token networks (SconeVis, SconeOcc's global
// transformer): on the split-precision variants (5, 6) their GEMMs take the split-precision kernel for EVERY launch size (negative
// argument = "choose on the layer's shape"; its column-tile width follows the launch, which does not change a single bit) -- one
// sequence costs the same as on the fp32 kernels, 8 or 30 sequences in one launch (scene batch, the neighbour cameras of a MACARONS
// decision) run 1.3x faster, and a sequence's result still does not depend on how many share the launch.  Variant 1 = exact fp32
// everywhere.  MCR_SMALL_SPLIT=0: the fp32 small-problem kernels on every variant (A/B).
static inline int64_t seq_route(int64_t L) {
    static const bool on = []() { const char* e = getenv("MCR_SMALL_SPLIT"); return !(e && e[0] == '0'); }();
    return on && g_local_pct_variant >= 5 ? -L : L;
}

// the wide layers behind the encoders (SconeVis fc1 / fc2, the global transformer's lin0): on the encoders' matrix path;
// MCR_HEAD_SPLIT=0: the exact-fp32 kernels, as before round 4 (A/B)
static inline int64_t head_route(int64_t L) {
    static const bool on = []() { const char* e = getenv("MCR_HEAD_SPLIT"); return !(e && e[0] == '0'); }();
    return on ? seq_route(L) : L;
}

// long-sequence attention: P V on fp16 hi/lo pairs (nn_kernels.hip: PVH) on the fp16-split variant; MCR_ATTN_PVH=0: fp32 MFMA (A/B)
static inline bool attn_pv_half() {
    static const bool on = []() { const char* e = getenv("MCR_ATTN_PVH"); return !(e && e[0] == '0'); }();
    return on && fp16_planes_variant();
}

// Encoder GEMMs of the long-sequence networks on fp16 hi/lo PLANES (variant 6, sequences of >= 512 tokens; MCR_ENC_PLANES=0: the
// bf16 x 6 kernels as on variant 5, A/B): the GEMM inputs are split once where they are produced -- LayerNorm writes planes, the FF's
// first GEMM writes planes, the attention output is split in one pass -- and every operand reaches LDS by DMA (linear3p.hip): no
// split and no staging registers inside the GEMMs, three MFMAs per product instead of six.  Chosen on the sequence length alone, so a
// cloud's result does not depend on how many clouds share the launch.  Needs |activation| < 65504 like the rest of variant 6: the
// occupancy / harmonics that come out non-finite otherwise are what the range guards look at.
static inline bool enc_planes(int L, int E) {
    static const bool on = []() { const char* e = getenv("MCR_ENC_PLANES"); return !(e && e[0] == '0'); }();
    return on && fp16_planes_variant() && L >= 512 && E % 32 == 0;
}

// The layers either side of the encoders (the embeddings' second layer, the final LayerNorm and the fc / lin0 layers behind it) on the
// same planes path; MCR_ENDS_PLANES=0: the fp32 / bf16 x 6 kernels as before (A/B)
static inline bool ends_planes(int L, int E) {
    static const bool on = []() { const char* e = getenv("MCR_ENDS_PLANES"); return !(e && e[0] == '0'); }();
    return on && enc_planes(L, E);
}

// The planes live in the encoder's own scratch (an fp32 row = two fp16 rows): h <- planes of LayerNorm(x) / fp32 attention output,
// ff <- planes of the attention output, then of the FF's hidden layer; the weights' planes (split per call, 2^8 scale: linear3h.hip)
// go to whichever of ff / qkv is idle.  L >= 512 makes every region large enough for them.
static void run_encoder_planes(hipStream_t s, const EncW& w, float* x, float* h, float* qkv, float* ff, int64_t S, int L, int E, int H,
                               const int* lens) {
    const int64_t T = S * L;
    const int dqk = E / 4, W3 = 2 * dqk + E;
    const float inv = 1.0f / 256.0f;
    // 1 on variant 7 (with head dims the planes attention covers: the default architectures): the low planes below are neither written nor read
    const int np = attention_planes_applicable(H, dqk, E, W3) ? matrix_planes() : 2;
    _Float16 *hh = reinterpret_cast<_Float16*>(h), *hl = np == 1 ? nullptr : hh + (size_t)T * E;   // planes [2][T][E] over h
    _Float16 *fh = reinterpret_cast<_Float16*>(ff);                                          // planes over ff: [2][T][E] or [2][T][2E]
    auto wsplit = [&](const float* W, int N, int K, void* dst, const void* given = nullptr) {
        if (given) return (const _Float16*)given;            // host-built planes (one split per parameter version instead of per call)
        launch_split_weights(s, W, K, dst, N, K);
        return (const _Float16*)dst;
    };
    launch_layernorm_planes(s, x, E, w.n1g, w.n1b, hh, hl, E, T, E);                         // Attention.py:287
    const _Float16* Wq = wsplit(w.qkv.w, W3, E, ff, w.p_qkv);
    _Float16* ah = reinterpret_cast<_Float16*>(qkv);                                         // the attention's result as planes [2][T][E]
    static const bool att_planes = []() { const char* e = getenv("MCR_ENC_ATT_PLANES"); return !(e && e[0] == '0'); }();   // (A/B)
    if ((att_planes || np == 1) && attention_planes_applicable(H, dqk, E, W3)) {
        // q | k | v leave the projection as planes [2][T][W3] over qkv (:186-188); the attention stages K / V tiles by DMA (:191-198) and
        // writes its result as planes over h (the LayerNorm's, consumed by then).  One block per (query tile, head, sequence) whatever S is
        // (a cloud's result must not depend on how many clouds share the launch): a batch of clouds saves the combine pass and the fp32
        // parts of the key-split form (0.23 ms of a MACARONS decision); one cloud alone pays 13 us per attention for it (41 instead of
        // 23 + 5 us, hidden beside the local transformers in an NBV step).  MCR_ENC_ATT_SPLIT=1: keys over two blocks + combine (A/B)
        _Float16 *qh_ = reinterpret_cast<_Float16*>(qkv), *ql_ = qh_ + (size_t)T * W3;
        launch_linear3p(s, hh, hl, E, Wq, Wq + (size_t)W3 * E, E, w.qkv.b, nullptr, qh_, ql_, W3, T, W3, E, ACT_NONE, inv, nullptr, 0, nullptr,
                        nullptr, 0, np);
        static const int split_mode_env = []() { const char* e = getenv("MCR_ENC_ATT_SPLIT"); return e ? atoi(e) : 0; }();
        const int split_mode = np == 1 ? 0 : split_mode_env;                                   // (the single-plane form never splits its keys)
        if (split_mode == 0) ah = hh;                                                          // (split: the combine pass writes the planes over qkv, dead by then)
        launch_attention_planes(s, qh_, ql_, W3, h, E, ah, ah + (size_t)T * E, E, S, L, H, dqk, E, lens, ff, (size_t)T * 2 * E, split_mode, np);
    } else {
        launch_linear3p(s, hh, hl, E, Wq, Wq + (size_t)W3 * E, E, w.qkv.b, qkv, nullptr, nullptr, W3, T, W3, E, ACT_NONE, inv, nullptr, 0, nullptr);   // :186-188
        // attention: fp32 parts in h / ff (key-split scratch); its combine pass writes the result straight as planes into qkv (free by then)
        bool planes_done = false;
        static const bool fuse = []() { const char* e = getenv("MCR_ENC_COMBINE_PLANES"); return !(e && e[0] == '0'); }();   // (A/B)
        launch_attention(s, qkv, W3, h, E, S, L, H, dqk, E, lens, ff, (size_t)T * 2 * E, /*split_by_length=*/true, attn_pv_half(), nullptr, 0, 0, 0,
                         fuse ? ah : nullptr, ah + (size_t)T * E, E, &planes_done);              // :191-198
        if (!planes_done) launch_split_to_planes(s, h, E, ah, ah + (size_t)T * E, E, T, E);
    }
    const _Float16* Wo = wsplit(w.out.w, E, E, ff, w.p_out);
    launch_linear3p(s, ah, ah + (size_t)T * E, E, Wo, Wo + (size_t)E * E, E, w.out.b, x, nullptr, nullptr, E, T, E, E, ACT_NONE, inv, nullptr, 0,
                    nullptr, x, E, np);                                                        // :201-202 + residual :290
    launch_layernorm_planes(s, x, E, w.n2g, w.n2b, hh, hl, E, T, E);                         // :293
    const _Float16* W1 = wsplit(w.ff1.w, 2 * E, E, qkv, w.p_ff1);
    launch_linear3p(s, hh, hl, E, W1, W1 + (size_t)2 * E * E, E, w.ff1.b, nullptr, fh, fh + (size_t)T * 2 * E, 2 * E, T, 2 * E, E, ACT_GELU, inv,
                    nullptr, 0, nullptr, nullptr, 0, np);                                      // :232 (planes out)
    const _Float16* W2 = wsplit(w.ff2.w, E, 2 * E, qkv, w.p_ff2);
    launch_linear3p(s, fh, fh + (size_t)T * 2 * E, 2 * E, W2, W2 + (size_t)E * 2 * E, 2 * E, w.ff2.b, x, nullptr, nullptr, E, T, E, 2 * E, ACT_NONE,
                    inv, nullptr, 0, nullptr, x, E, np);                                       // :235 + residual :298
}

// x <- Encoder(x)  in place.  x [T, E]; scratch h [T, E], qkv [T, 2*dqk + E], ff [T, 2E]
static void run_encoder(hipStream_t s, const EncW& w, float* x, float* h, float* qkv, float* ff, int64_t S, int L, int E,
                        int H, const int* lens = nullptr) {
    const int64_t T = S * L;
    const int dqk = E / 4, W3 = 2 * dqk + E;
    if (enc_planes(L, E)) {
        run_encoder_planes(s, w, x, h, qkv, ff, S, L, E, H, lens);
        return;
    }
    launch_layernorm(s, x, E, w.n1g, w.n1b, h, E, T, E);                                   // Attention.py:287
    // every GEMM of the networks routes (fp32 vs split precision) on the rows of ONE sequence, not on T: see launch_linear
    launch_linear(s, h, E, w.qkv.w, w.qkv.b, nullptr, 0, qkv, W3, T, W3, E, ACT_NONE, nullptr, 0, 0, seq_route(L));       // :186-188
    launch_attention(s, qkv, W3, h, E, S, L, H, dqk, E, lens, ff, (size_t)T * 2 * E, /*split_by_length=*/true,     // :191-198 (ff is free here: key-split scratch)
                     attn_pv_half());
    launch_linear(s, h, E, w.out.w, w.out.b, x, E, x, E, T, E, E, ACT_NONE, nullptr, 0, 0, seq_route(L));                 // :201-202 + residual :290
    launch_layernorm(s, x, E, w.n2g, w.n2b, h, E, T, E);                                   // :293
    launch_linear(s, h, E, w.ff1.w, w.ff1.b, nullptr, 0, ff, 2 * E, T, 2 * E, E, ACT_GELU, nullptr, 0, 0, seq_route(L));  // :232
    launch_linear(s, ff, 2 * E, w.ff2.w, w.ff2.b, x, E, x, E, T, E, 2 * E, ACT_NONE, nullptr, 0, 0, seq_route(L));        // :235 + residual :298
}

// ---- PCTransformer (SconeOcc.py:45-130): S sequences of L points (pts_dim 3), E = 128, 2 encoders, 4 heads ----
constexpr int PCT_E = 128, PCT_INNER = 125, PCT_NW = 4 + 2 * 12 + 2 + 2;
struct PctW { LinW l1, l2; EncW enc[2]; const float *ng, *nb; LinW lin0;
              const void *p_l2 = nullptr; const float* b_l2p = nullptr; const void* p_lin0 = nullptr; };   // host-built planes of the end layers (optional)
static void read_pct_end_planes(const float* const*& p, PctW& w) {
    w.p_l2 = *p++; w.b_l2p = *p++; w.p_lin0 = *p++;
}
static PctW read_pct(const float* const*& p) {
    PctW w;
    w.l1.w = *p++; w.l1.b = *p++; w.l2.w = *p++; w.l2.b = *p++;
    w.enc[0] = read_enc(p); w.enc[1] = read_enc(p);
    w.ng = *p++; w.nb = *p++;
    w.lin0.w = *p++; w.lin0.b = *p++;
    return w;
}
static size_t pct_ws_bytes(int64_t T) {
    return al(T * PCT_E) * 2 + al(T * (PCT_E + 64)) + al(T * 2 * PCT_E);
}
// feat[s*ld_feat + 0 : feature_dim] ; feature_dim = 2 * half (max || avg)
// lens (optional, device int per sequence): sequence s consists of its first lens[s] rows (zero-padded batch of clouds of
// different sizes): attention keys and the pooling stop there
static void run_pct(hipStream_t s, const PctW& w, const float* pc, float* feat, int64_t ld_feat, int64_t S, int L, int half,
                    Arena& a, const int* lens = nullptr) {
    const int64_t T = S * L;
    float* x = a.f(T * PCT_E);
    float* h = a.f(T * PCT_E);
    float* qkv = a.f(T * (PCT_E + 64));
    float* ff = a.f(T * 2 * PCT_E);
    const bool planes = ends_planes(L, PCT_E) && half % 4 == 0;
    const float inv = 1.0f / 256.0f;
    const int np = matrix_planes();                       // 1 on variant 7: the GEMMs below read / write the high planes alone
    _Float16 *hh = reinterpret_cast<_Float16*>(h), *hl = hh + (size_t)T * PCT_E;            // planes [2][T][128] over h
    // Embedding (Attention.py:98-128): linear1 3->125, GELU, linear2 125->125, concat raw input -> 128
    if (planes) {
        // the 125-wide inner layer padded with exact zeros to the planes GEMM's K = 128: linear1 writes planes, linear2 multiplies them
        // (output columns 125..127 = 0 + 0, then overwritten by the raw input)
        launch_linear_smallk_planes(s, pc, 3, w.l1.w, w.l1.b, hh, hl, PCT_E, T, PCT_INNER, 3, ACT_GELU, PCT_E);
        const _Float16* wp = (const _Float16*)w.p_l2;                                        // host-built (once per parameter version) ...
        const float* bp = w.b_l2p;
        if (!wp) {                                                                           // ... or padded here, per call
            float* bq = ff + (size_t)PCT_E * PCT_E;                                          // behind the [2][128][128] halves
            launch_pad_weights(s, w.l2.w, PCT_INNER, w.l2.b, ff, bq, PCT_INNER, PCT_INNER, PCT_E, PCT_E);
            wp = reinterpret_cast<const _Float16*>(ff); bp = bq;
        }
        launch_linear3p(s, hh, hl, PCT_E, wp, wp + (size_t)PCT_E * PCT_E, PCT_E, bp, x, nullptr, nullptr, PCT_E, T, PCT_E, PCT_E, ACT_NONE, inv,
                        nullptr, 0, nullptr, nullptr, 0, np);
    } else {
        launch_linear(s, pc, 3, w.l1.w, w.l1.b, nullptr, 0, h, PCT_INNER, T, PCT_INNER, 3, ACT_GELU, nullptr, 0, 0, L);
        launch_linear(s, h, PCT_INNER, w.l2.w, w.l2.b, nullptr, 0, x, PCT_E, T, PCT_INNER, PCT_INNER, ACT_NONE, nullptr, 0, 0, L);
    }
    launch_copy2d(s, pc, 3, x + PCT_INNER, PCT_E, T, 3);
    for (int e = 0; e < 2; ++e) run_encoder(s, w.enc[e], x, h, qkv, ff, S, L, PCT_E, 4, lens);
    if (planes) {                                                                            // SconeOcc.py:119-122 on planes
        launch_layernorm_planes(s, x, PCT_E, w.ng, w.nb, hh, np == 1 ? nullptr : hl, PCT_E, T, PCT_E);
        const _Float16* wp = (const _Float16*)w.p_lin0;
        if (!wp) { launch_split_weights(s, w.lin0.w, PCT_E, qkv, half, PCT_E); wp = reinterpret_cast<const _Float16*>(qkv); }
        launch_linear3p(s, hh, hl, PCT_E, wp, wp + (size_t)half * PCT_E, PCT_E, w.lin0.b, ff, nullptr, nullptr, half, T, half, PCT_E, ACT_NONE, inv,
                        nullptr, 0, nullptr, nullptr, 0, np);
    } else {
        launch_layernorm(s, x, PCT_E, w.ng, w.nb, h, PCT_E, T, PCT_E);                      // SconeOcc.py:119
        launch_linear(s, h, PCT_E, w.lin0.w, w.lin0.b, nullptr, 0, ff, half, T, half, PCT_E, ACT_NONE, nullptr, 0, 0, head_route(L));   // :122
    }
    launch_pool_max_avg(s, ff, half, feat, ld_feat, S, L, half, lens);                       // :124-126
}

// *flag |= 1 if any of x[0..n) is inf / NaN (the flag is the caller's: cleared by the caller, OR'ed here)
__global__ void nonfinite_flag_kernel(const float* __restrict__ x, long long n, int* __restrict__ flag) {
    bool bad = false;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        bad |= !(fabsf(x[i]) <= 3.4028234663852886e38f);
    if (__ballot(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}
static void launch_nonfinite_flag(hipStream_t s, const float* x, int64_t n, int* flag) {
    hipLaunchKernelGGL(nonfinite_flag_kernel, dim3((unsigned)std::min<int64_t>(cdiv(n, 256), 1024)), dim3(256), 0, s, x, (long long)n, flag);
}

// ---- SconeOcc head on fp16 hi/lo planes end to end (variant 6; linear3p.hip): every activation is split once where it is produced
// (local_pct6's pooled features, the GEMM epilogues), every operand reaches LDS by DMA.  Scratch (the caller's regions, same
// bytes as the fp32 layout): featP = planes [2][T][1344] fp16, h1P = [2][T][512] fp16 (first used as [2][T][256] for the
// x-embedding), h2 = fp32 [T][256] (first half first used for xe1's fp32 output, second half for its planes).
struct HeadScratch { _Float16* featP; _Float16* h1P; float* h2; void* wplanes; };
static void head_planes_weights(hipStream_t s, int which, const float* W, int64_t ldw, int N, int K, const void* const* head_planes,
                                const float* head_inv_scales, void* wplanes, const _Float16*& Wh, const _Float16*& Wl, float& inv) {
    if (head_planes && head_planes[which] && head_inv_scales[which] > 0.f) {
        Wh = (const _Float16*)head_planes[which];
        inv = head_inv_scales[which];
    } else {                                              // no host planes: split here (fixed 2^8 scale, |w| < 255)
        launch_split_weights(s, W, ldw, wplanes, N, K);
        Wh = (const _Float16*)wplanes;
        inv = 1.0f / 256.0f;
    }
    Wl = Wh + (size_t)N * K;
}
// the part of the planes head that needs nothing but the queries: x embedding 3 -> 128 -> 256 -> 512 (GELU each, SconeOcc.py:35-42)
// into columns 768.. of the feature planes, the view harmonics into columns 1280..
static void run_x_embedding_planes(hipStream_t s, const float* x, const float* view_harmonics, int64_t T, const LinW& xe1, const LinW& xe2,
                                   const LinW& xe3, const void* const* head_planes, const float* head_inv_scales, const HeadScratch& w) {
    _Float16 *fh = w.featP, *fl = w.featP + (size_t)T * 1344;
    _Float16 *hh = w.h1P;
    _Float16* x1h = reinterpret_cast<_Float16*>(w.h2 + (size_t)T * 128);     // planes [2][T][128]
    _Float16* x1l = x1h + (size_t)T * 128;
    const _Float16 *Wh, *Wl;
    float inv;
    const int np = matrix_planes();                      // 1 on variant 7: the low planes below are neither written (GEMM epilogues) nor read
    launch_linear_smallk_planes(s, x, 3, xe1.w, xe1.b, x1h, x1l, 128, T, 128, 3, ACT_GELU);       // (planes directly: no fp32 rows, no split pass)
    head_planes_weights(s, 0, xe2.w, 128, 256, 128, head_planes, head_inv_scales, w.wplanes, Wh, Wl, inv);
    launch_linear3p(s, x1h, x1l, 128, Wh, Wl, 128, xe2.b, nullptr, hh, hh + (size_t)T * 256, 256, T, 256, 128, ACT_GELU, inv, nullptr, 0, nullptr,
                    nullptr, 0, np);
    head_planes_weights(s, 1, xe3.w, 256, 512, 256, head_planes, head_inv_scales, w.wplanes, Wh, Wl, inv);
    launch_linear3p(s, hh, hh + (size_t)T * 256, 256, Wh, Wl, 256, xe3.b, nullptr, fh + 768, fl + 768, 1344, T, 512, 256, ACT_GELU, inv, nullptr, 0,
                    nullptr, nullptr, 0, np);
    if (view_harmonics) launch_split_to_planes(s, view_harmonics, 64, fh + 1280, np == 1 ? nullptr : fl + 1280, 1344, T, 64);   // (NULL: the caller splits them later)
}

// x_done: the x-embedding part has been queued elsewhere (the side stream: the join covers it)
template <class Join>
static void run_head_planes(hipStream_t s, const float* x, const float* view_harmonics, int64_t T, const LinW& xe1, const LinW& xe2,
                            const LinW& xe3, const LinW& lin1, const LinW& lin2, const LinW& lin3, const float* gbias,
                            int64_t rows_per_group, const int* row_group, const void* const* head_planes, const float* head_inv_scales,
                            const HeadScratch& w, float* out, Join join, bool x_done = false) {
    _Float16 *fh = w.featP, *fl = w.featP + (size_t)T * 1344;
    _Float16 *hh = w.h1P;
    const _Float16 *Wh, *Wl;
    float inv;
    const int np = matrix_planes();
    if (!x_done) run_x_embedding_planes(s, x, view_harmonics, T, xe1, xe2, xe3, head_planes, head_inv_scales, w);
    join();                                               // the global feature (side stream) is needed from here on
    // head MLP 1856 -> 512 -> 256 -> 1, GELU after every layer incl. the last (SconeOcc.py:334-345); the global 512 columns are gbias
    head_planes_weights(s, 2, lin1.w + 512, 1856, 512, 1344, head_planes, head_inv_scales, w.wplanes, Wh, Wl, inv);
    launch_linear3p(s, fh, fl, 1344, Wh, Wl, 1344, lin1.b, nullptr, hh, hh + (size_t)T * 512, 512, T, 512, 1344, ACT_GELU, inv, gbias,
                    rows_per_group, row_group, nullptr, 0, np);
    head_planes_weights(s, 3, lin2.w, 512, 256, 512, head_planes, head_inv_scales, w.wplanes, Wh, Wl, inv);
    static const bool fuse_tail = []() { const char* e = getenv("MCR_HEAD_FUSE_TAIL"); return !(e && e[0] == '0'); }();     // dev A/B knob
    if (fuse_tail && linear3p_dot_applicable(256, 512, 512, 512)) {
        // 512 -> 256 (GELU) -> 1 (GELU) in one launch: the block owns all 256 features of its rows and dots them with linear3.weight
        launch_linear3p_dot(s, hh, hh + (size_t)T * 512, 512, Wh, Wl, 512, lin2.b, T, 512, ACT_GELU, inv, lin3.w, lin3.b, ACT_GELU, out, np);
        return;
    }
    launch_linear3p(s, hh, hh + (size_t)T * 512, 512, Wh, Wl, 512, lin2.b, w.h2, nullptr, nullptr, 256, T, 256, 512, ACT_GELU, inv, nullptr, 0,
                    nullptr, nullptr, 0, np);
    launch_linear(s, w.h2, 256, lin3.w, lin3.b, nullptr, 0, out, 1, T, 1, 256, ACT_GELU, nullptr, 0, 0, 1);
}

}  // namespace mcr

using namespace mcr;

extern "C" {

// ---- individual blocks (used by the host mirrors of Attention.py's modules and by the block-level tests) ----
int mcr_linear(const float* X, int64_t ldx, const float* W, const float* bias, const float* residual, int64_t ldr, float* Y,
               int64_t ldy, int64_t M, int N, int K, int gelu, void* stream) {
    VariantScope variant_scope_;
    MCR_REQUIRE(X && W && Y, "mcr_linear: null pointer");
    MCR_REQUIRE(M > 0 && N > 0 && K > 0, "mcr_linear: empty problem");
    MCR_REQUIRE(ldx >= K && ldy >= N && (!residual || ldr >= N), "mcr_linear: leading dimension too small");
    launch_linear((hipStream_t)stream, X, ldx, W, bias, residual, ldr, Y, ldy, M, N, K, gelu ? ACT_GELU : ACT_NONE);
    MCR_LAUNCH_CHECK("mcr_linear");
    return 0;
}

int mcr_layernorm(const float* X, int64_t ldx, const float* gamma, const float* beta, float* Y, int64_t ldy, int64_t M, int E,
                  void* stream) {
    MCR_REQUIRE(X && gamma && beta && Y, "mcr_layernorm: null pointer");
    MCR_REQUIRE(M > 0 && E > 0 && E <= 512, "mcr_layernorm: need 0 < E <= 512 (got %d)", E);
    launch_layernorm((hipStream_t)stream, X, ldx, gamma, beta, Y, ldy, M, E);
    MCR_LAUNCH_CHECK("mcr_layernorm");
    return 0;
}

int mcr_attention(const float* qkv, int64_t ldq, float* out, int64_t ldo, int64_t S, int64_t L, int n_heads, int qk_dim,
                  int v_dim, void* stream) {
    VariantScope variant_scope_;
    MCR_REQUIRE(qkv && out, "mcr_attention: null pointer");
    MCR_REQUIRE(S > 0 && L > 0, "mcr_attention: empty problem");
    MCR_REQUIRE(n_heads == 4 && ((qk_dim == 32 && v_dim == 128) || (qk_dim == 64 && v_dim == 256)),
                "mcr_attention: supported head layouts are 4 heads with (qk,v) = (32,128) or (64,256); got %d heads (%d,%d)",
                n_heads, qk_dim, v_dim);
    MCR_REQUIRE(L == 16 || S <= 65535, "mcr_attention: too many long sequences");
    launch_attention((hipStream_t)stream, qkv, ldq, out, ldo, S, (int)L, n_heads, qk_dim, v_dim, nullptr, nullptr, 0, false, attn_pv_half());
    MCR_LAUNCH_CHECK("mcr_attention");
    return 0;
}

int mcr_attention_masked(const float* qkv, int64_t ldq, float* out, int64_t ldo, int64_t S, int64_t L, int n_heads, int qk_dim, int v_dim,
                         const unsigned char* mask, int64_t mask_seq_stride, int64_t mask_head_stride, int64_t mask_query_stride,
                         void* workspace, size_t workspace_bytes, void* stream) {
    VariantScope variant_scope_;
    MCR_REQUIRE(qkv && out && mask, "mcr_attention_masked: null pointer");
    MCR_REQUIRE(S > 0 && L > 0, "mcr_attention_masked: empty problem");
    MCR_REQUIRE(n_heads == 4 && ((qk_dim == 32 && v_dim == 128) || (qk_dim == 64 && v_dim == 256)),
                "mcr_attention_masked: supported head layouts are 4 heads with (qk,v) = (32,128) or (64,256); got %d heads (%d,%d)",
                n_heads, qk_dim, v_dim);
    MCR_REQUIRE(mask_seq_stride >= 0 && mask_head_stride >= 0 && mask_query_stride >= 0, "mcr_attention_masked: negative mask stride");
    MCR_REQUIRE(L == 16 || S <= 32767, "mcr_attention_masked: too many long sequences");
    launch_attention((hipStream_t)stream, qkv, ldq, out, ldo, S, (int)L, n_heads, qk_dim, v_dim, nullptr, (float*)workspace,
                     workspace ? workspace_bytes / sizeof(float) : 0, false, false, mask, mask_seq_stride, mask_head_stride, mask_query_stride);
    MCR_LAUNCH_CHECK("mcr_attention_masked");
    return 0;
}

size_t mcr_attention_workspace_bytes(int64_t S, int64_t L, int n_heads, int v_dim) {
    return attention_split_floats(S, (int)L, n_heads, v_dim) * sizeof(float);
}

int mcr_attention_ws(const float* qkv, int64_t ldq, float* out, int64_t ldo, int64_t S, int64_t L, int n_heads, int qk_dim,
                     int v_dim, void* workspace, size_t workspace_bytes, void* stream) {
    VariantScope variant_scope_;
    MCR_REQUIRE(qkv && out, "mcr_attention_ws: null pointer");
    MCR_REQUIRE(S > 0 && L > 0, "mcr_attention_ws: empty problem");
    MCR_REQUIRE(n_heads == 4 && ((qk_dim == 32 && v_dim == 128) || (qk_dim == 64 && v_dim == 256)),
                "mcr_attention_ws: supported head layouts are 4 heads with (qk,v) = (32,128) or (64,256); got %d heads (%d,%d)",
                n_heads, qk_dim, v_dim);
    MCR_REQUIRE(L == 16 || S <= 32767, "mcr_attention_ws: too many long sequences");
    launch_attention((hipStream_t)stream, qkv, ldq, out, ldo, S, (int)L, n_heads, qk_dim, v_dim, nullptr, (float*)workspace,
                     workspace ? workspace_bytes / sizeof(float) : 0, false, attn_pv_half());
    MCR_LAUNCH_CHECK("mcr_attention_ws");
    return 0;
}

size_t mcr_attention_planes_workspace_bytes(int64_t S, int64_t L, int n_heads, int qk_dim, int v_dim) {
    return al((size_t)S * L * (2 * qk_dim + v_dim)) + attention_split_floats(S, (int)L, n_heads, v_dim) * sizeof(float);   // planes (the bytes of the fp32 rows) + key-split scratch
}

int mcr_attention_planes(const float* qkv, int64_t ldq, float* out, int64_t ldo, int64_t S, int64_t L, int n_heads, int qk_dim, int v_dim,
                         const int* lens, int split_mode, void* workspace, size_t workspace_bytes, void* stream) {
    MCR_REQUIRE(qkv && out && workspace, "mcr_attention_planes: null pointer");
    MCR_REQUIRE(S > 0 && L > 0 && S <= 32767, "mcr_attention_planes: bad problem size");
    MCR_REQUIRE(n_heads == 4 && ((qk_dim == 32 && v_dim == 128) || (qk_dim == 64 && v_dim == 256)),
                "mcr_attention_planes: supported head layouts are 4 heads with (qk,v) = (32,128) or (64,256); got %d heads (%d,%d)",
                n_heads, qk_dim, v_dim);
    MCR_REQUIRE(workspace_bytes >= mcr_attention_planes_workspace_bytes(S, L, n_heads, qk_dim, v_dim), "mcr_attention_planes: workspace too small");
    const int W3 = 2 * qk_dim + v_dim;
    const int64_t T = S * L;
    _Float16* ph = reinterpret_cast<_Float16*>(workspace);
    _Float16* pl = ph + (size_t)T * W3;
    float* split_ws = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + al((size_t)T * W3));
    launch_split_to_planes((hipStream_t)stream, qkv, ldq, ph, pl, W3, T, W3);
    launch_attention_planes((hipStream_t)stream, ph, pl, W3, out, ldo, nullptr, nullptr, 0, S, (int)L, n_heads, qk_dim, v_dim, lens, split_ws,
                            attention_split_floats(S, (int)L, n_heads, v_dim), split_mode);
    MCR_LAUNCH_CHECK("mcr_attention_planes");
    return 0;
}

int mcr_colmax_broadcast(const float* X, int64_t ldx, float* Y, int64_t ldy, int64_t S, int64_t L, int E, void* stream) {
    MCR_REQUIRE(X && Y && S > 0 && L > 0 && E > 0, "mcr_colmax_broadcast: bad arguments");
    launch_colmax_broadcast((hipStream_t)stream, X, ldx, Y, ldy, S, (int)L, E);
    MCR_LAUNCH_CHECK("mcr_colmax_broadcast");
    return 0;
}

int mcr_pool_max_avg(const float* X, int64_t ldx, float* Y, int64_t ldy, int64_t S, int64_t L, int E, void* stream) {
    MCR_REQUIRE(X && Y && S > 0 && L > 0 && E > 0, "mcr_pool_max_avg: bad arguments");
    launch_pool_max_avg((hipStream_t)stream, X, ldx, Y, ldy, S, (int)L, E);
    MCR_LAUNCH_CHECK("mcr_pool_max_avg");
    return 0;
}

// *flag |= 1 if any of x[0 .. n) is inf / NaN: the range guard of the fp16-split matrix path for outputs that leave through an entry
// point without a flag of its own (SconeVis' harmonics)
int mcr_nonfinite_flag(const float* x, int64_t n, int* flag, void* stream) {
    MCR_REQUIRE(x && flag && n > 0, "mcr_nonfinite_flag: bad arguments");
    launch_nonfinite_flag((hipStream_t)stream, x, n, flag);
    MCR_LAUNCH_CHECK("nonfinite_flag_kernel");
    return 0;
}

int mcr_get_local_pct_variant(void);
int mcr_local_pct_blob_floats(void) { return local_pct_blob_floats(); }
int mcr_local_pct3_blob_floats(void) { return local_pct3_blob_floats(); }
int mcr_local_pct6_blob_floats(void) { return local_pct6_blob_floats(); }
int mcr_local_pct7_blob_floats(void) { return local_pct7_blob_floats(); }

int mcr_set_local_pct_variant(int v) {
    MCR_REQUIRE(v == 1 || v == 5 || v == 6 || MCR_VARIANT8_OK(v), "mcr_set_local_pct_variant: the process default must be 1, 5 or 6 (got %d; 7, the opt-in 16-bit matrix path, is per call only: mcr_call_variant)", v);
    g_default_variant = v;
    return 0;
}
int mcr_get_local_pct_variant(void) { return g_default_variant; }
int mcr_call_variant(int v) {
    MCR_REQUIRE(v == 0 || v == 1 || v == 5 || v == 6 || v == 7 || MCR_VARIANT8_OK(v),
                "mcr_call_variant: variant must be 0 (default), 1, 5, 6 or 7 (the opt-in 16-bit matrix path) (got %d)", v);
    t_next_variant = v;
    return 0;
}
static void run_local_pct(hipStream_t s, const float* offs, float* feat, int64_t ld, int64_t S, const float* blob,
                          void* feat_h = nullptr, void* feat_l = nullptr) {
    if (g_local_pct_variant == 1) launch_local_pct(s, offs, feat, ld, S, blob);
    else if (g_local_pct_variant == 5) launch_local_pct5(s, offs, feat, ld, S, blob);
#ifdef MCR_DEV_LOCAL_PCT8
    else if (g_local_pct_variant == 8) launch_local_pct8(s, offs, feat, ld, S, blob);
#endif
    else if (g_local_pct_variant == 7) launch_local_pct7(s, offs, feat, ld, S, blob, feat_h);   // ONE plane out when feat_h is set
    else launch_local_pct6(s, offs, feat, ld, S, blob, feat_h, feat_l);       // planes out (variant 6 only) when feat_h is set
}

int mcr_local_pct_forward(const float* offsets, float* features, int64_t ld_features, int64_t S, const float* blob,
                          void* stream) {
    VariantScope variant_scope_;
    MCR_REQUIRE(offsets && features && blob, "mcr_local_pct_forward: null pointer");
    MCR_REQUIRE(S > 0 && ld_features >= 256, "mcr_local_pct_forward: bad sizes");
    MCR_REQUIRE((reinterpret_cast<uintptr_t>(blob) & 15) == 0, "mcr_local_pct_forward: blob must be 16-byte aligned");
    run_local_pct((hipStream_t)stream, offsets, features, ld_features, S, blob);
    MCR_LAUNCH_CHECK("mcr_local_pct_forward");
    return 0;
}

// ---------------------------------------------------------------------------------------------------------
size_t mcr_pc_transformer_workspace_bytes(int64_t S, int64_t L) { return pct_ws_bytes(S * L) + 4096; }

int mcr_pc_transformer_forward(const float* pc, float* features, int64_t S, int64_t L, int feature_dim,
                               const float* const* weights, int n_weights, void* workspace, size_t workspace_bytes,
                               void* stream) {
    VariantScope variant_scope_;
    MCR_REQUIRE(pc && features && weights, "mcr_pc_transformer_forward: null pointer");
    MCR_REQUIRE(n_weights == PCT_NW || n_weights == PCT_NW + 8 || n_weights == PCT_NW + 11,
                "mcr_pc_transformer_forward: expected %d weight pointers (+ 8 or 11 plane pointers), got %d", PCT_NW, n_weights);
    MCR_REQUIRE(S > 0 && L > 0, "mcr_pc_transformer_forward: empty problem");
    MCR_REQUIRE(feature_dim == 256 || feature_dim == 512, "mcr_pc_transformer_forward: feature_dim must be 256 or 512");
    MCR_REQUIRE(L == 16 || S <= 65535, "mcr_pc_transformer_forward: too many long sequences");
    MCR_REQUIRE(workspace && workspace_bytes >= mcr_pc_transformer_workspace_bytes(S, L),
                "mcr_pc_transformer_forward: workspace too small");
    for (int i = 0; i < n_weights; ++i) MCR_REQUIRE(weights[i], "mcr_pc_transformer_forward: weight %d is null", i);
    const float* const* p = weights;
    PctW w = read_pct(p);
    if (n_weights >= PCT_NW + 8) { read_enc_planes(p, w.enc[0]); read_enc_planes(p, w.enc[1]); }
    if (n_weights == PCT_NW + 11) read_pct_end_planes(p, w);
    Arena a{(char*)workspace, workspace_bytes, 0};
    run_pct((hipStream_t)stream, w, pc, features, feature_dim, S, (int)L, feature_dim / 2, a);
    MCR_LAUNCH_CHECK("mcr_pc_transformer_forward");
    return 0;
}

// ---------------------------------------------------------------------------------------------------------
// SconeVis.forward (SconeVis.py:121-162): E = 256, 3 encoders, 4 heads, view_state_mode "end".
constexpr int VIS_E = 256, VIS_F = 126, VIS_NW = 4 + 3 * 12 + 2 + 6;

size_t mcr_scone_vis_workspace_bytes(int64_t B, int64_t N) {
    const int64_t T = B * N;
    return al(T * VIS_E) * 2 + al(T * (VIS_E + 128)) + al(T * 2 * VIS_E) + 4096;
}

int mcr_scone_vis_forward(const float* pts, const float* view_harmonics, float* out, int64_t B, int64_t N,
                          const float* const* weights, int n_weights, const int* lengths, void* workspace,
                          size_t workspace_bytes, void* stream) {
    VariantScope variant_scope_;
    MCR_REQUIRE(pts && view_harmonics && out && weights, "mcr_scone_vis_forward: null pointer");
    MCR_REQUIRE(n_weights == VIS_NW || n_weights == VIS_NW + 12 || n_weights == VIS_NW + 17,
                "mcr_scone_vis_forward: expected %d weight pointers (+ 12 or 17 plane pointers), got %d", VIS_NW, n_weights);
    MCR_REQUIRE(B > 0 && N > 0 && B <= 65535, "mcr_scone_vis_forward: bad problem size B=%ld N=%ld", (long)B, (long)N);
    MCR_REQUIRE(workspace && workspace_bytes >= mcr_scone_vis_workspace_bytes(B, N), "mcr_scone_vis_forward: workspace too small");
    for (int i = 0; i < n_weights; ++i) MCR_REQUIRE(weights[i], "mcr_scone_vis_forward: weight %d is null", i);
    hipStream_t s = (hipStream_t)stream;
    const float* const* p = weights;
    LinW l1{p[0], p[1]}, l2{p[2], p[3]};
    p += 4;
    EncW enc[3] = {read_enc(p), read_enc(p), read_enc(p)};
    const float *ng = *p++, *nb = *p++;
    LinW fc1{p[0], p[1]}, fc2{p[2], p[3]}, fc3{p[4], p[5]};
    p += 6;
    if (n_weights >= VIS_NW + 12)
        for (int e = 0; e < 3; ++e) read_enc_planes(p, enc[e]);
    const void *hp_l2 = nullptr, *hp_fc1 = nullptr, *hp_fc2 = nullptr, *hp_fc3 = nullptr;     // host-built planes of the end layers (optional)
    const float* hb_l2 = nullptr;
    if (n_weights == VIS_NW + 17) { hp_l2 = p[0]; hb_l2 = p[1]; hp_fc1 = p[2]; hp_fc2 = p[3]; hp_fc3 = p[4]; p += 5; }

    const int64_t T = B * N;
    Arena a{(char*)workspace, workspace_bytes, 0};
    float* x = a.f(T * VIS_E);
    float* h = a.f(T * VIS_E);
    float* qkv = a.f(T * (VIS_E + 128));
    float* ff = a.f(T * 2 * VIS_E);
    const bool planes = ends_planes((int)N, VIS_E);
    const float inv = 1.0f / 256.0f;
    const int np = matrix_planes();                       // 1 on variant 7: the GEMMs below read / write the high planes alone
    _Float16 *hh = reinterpret_cast<_Float16*>(h), *hl = hh + (size_t)T * VIS_E;               // planes [2][T][<= 256] over h
    // Embedding: 4 -> 126 GELU -> 126, || cloud-wide max (126) || raw input (4)  = 256   (Attention.py:98-128)
    if (planes) {
        // the 126-wide inner layer padded with exact zeros to K = 128: linear1 writes planes, linear2 multiplies them (its output
        // columns 126, 127 = 0 + 0 are overwritten by the cloud-wide max below)
        _Float16* x1l = hh + (size_t)T * 128;
        launch_linear_smallk_planes(s, pts, 4, l1.w, l1.b, hh, x1l, 128, T, VIS_F, 4, ACT_GELU, 128);
        const _Float16* wp = (const _Float16*)hp_l2;                                            // host-built (once per parameter version) ...
        const float* bp = hb_l2;
        if (!wp) {                                                                              // ... or padded here, per call
            float* bq = ff + (size_t)128 * 128;
            launch_pad_weights(s, l2.w, VIS_F, l2.b, ff, bq, VIS_F, VIS_F, 128, 128);
            wp = reinterpret_cast<const _Float16*>(ff); bp = bq;
        }
        launch_linear3p(s, hh, x1l, 128, wp, wp + (size_t)128 * 128, 128, bp, x, nullptr, nullptr, VIS_E, T, 128, 128, ACT_NONE, inv, nullptr, 0, nullptr,
                        nullptr, 0, np);
    } else {
        launch_linear(s, pts, 4, l1.w, l1.b, nullptr, 0, h, VIS_F, T, VIS_F, 4, ACT_GELU, nullptr, 0, 0, N);
        launch_linear(s, h, VIS_F, l2.w, l2.b, nullptr, 0, x, VIS_E, T, VIS_F, VIS_F, ACT_NONE, nullptr, 0, 0, N);
    }
    launch_colmax_broadcast(s, x, VIS_E, x + VIS_F, VIS_E, B, (int)N, VIS_F, lengths, pts, 4, 4, x + 2 * VIS_F);   // (+ the raw input columns)
    for (int e = 0; e < 3; ++e) run_encoder(s, enc[e], x, h, qkv, ff, B, (int)N, VIS_E, 4, lengths);   // SconeVis.py:139-140
    if (planes) {
        // :143-152 on planes: LayerNorm -> planes; fc1 (GELU) writes columns 0..191 of the next operand's planes, the view harmonics are
        // split into columns 192..255; fc2 (GELU) writes planes; fc3 leaves fp32.  Weight planes: split per call into the idle qkv region
        launch_layernorm_planes(s, x, VIS_E, ng, nb, hh, np == 1 ? nullptr : hl, VIS_E, T, VIS_E);
        const _Float16 *w1 = (const _Float16*)hp_fc1, *w2 = (const _Float16*)hp_fc2, *w3 = (const _Float16*)hp_fc3;
        if (!w1) {
            _Float16* q1 = reinterpret_cast<_Float16*>(qkv);
            _Float16* q2 = q1 + (size_t)2 * 192 * VIS_E;
            _Float16* q3 = q2 + (size_t)2 * 128 * VIS_E;
            launch_split_weights(s, fc1.w, VIS_E, q1, 192, VIS_E);
            launch_split_weights(s, fc2.w, VIS_E, q2, 128, VIS_E);
            launch_split_weights(s, fc3.w, 128, q3, 64, 128);
            w1 = q1; w2 = q2; w3 = q3;
        }
        _Float16 *fh = reinterpret_cast<_Float16*>(ff), *fl = fh + (size_t)T * VIS_E;            // planes [2][T][256] over ff
        launch_linear3p(s, hh, hl, VIS_E, w1, w1 + (size_t)192 * VIS_E, VIS_E, fc1.b, nullptr, fh, fl, VIS_E, T, 192, VIS_E, ACT_GELU, inv, nullptr, 0, nullptr,
                        nullptr, 0, np);
        launch_split_to_planes(s, view_harmonics, 64, fh + 192, np == 1 ? nullptr : fl + 192, VIS_E, T, 64);
        _Float16* gl = hh + (size_t)T * 128;                                                       // planes [2][T][128] over h (the LayerNorm's are consumed)
        launch_linear3p(s, fh, fl, VIS_E, w2, w2 + (size_t)128 * VIS_E, VIS_E, fc2.b, nullptr, hh, gl, 128, T, 128, VIS_E, ACT_GELU, inv, nullptr, 0, nullptr,
                        nullptr, 0, np);
        launch_linear3p(s, hh, gl, 128, w3, w3 + (size_t)64 * 128, 128, fc3.b, out, nullptr, nullptr, 64, T, 64, 128, ACT_NONE, inv, nullptr, 0, nullptr,
                        nullptr, 0, np);
    } else {
        launch_layernorm(s, x, VIS_E, ng, nb, h, VIS_E, T, VIS_E);                                   // :143
        // fc1 256->192 GELU, || view_harmonics (64), fc2 256->128 GELU, fc3 128->64                  (:146-152)
        launch_linear(s, h, VIS_E, fc1.w, fc1.b, nullptr, 0, ff, VIS_E, T, 192, VIS_E, ACT_GELU, nullptr, 0, 0, head_route(N));
        launch_copy2d(s, view_harmonics, 64, ff + 192, VIS_E, T, 64);
        launch_linear(s, ff, VIS_E, fc2.w, fc2.b, nullptr, 0, h, 128, T, 128, VIS_E, ACT_GELU, nullptr, 0, 0, head_route(N));
        launch_linear(s, h, 128, fc3.w, fc3.b, nullptr, 0, out, 64, T, 64, 128, ACT_NONE, nullptr, 0, 0, N);
    }
    MCR_LAUNCH_CHECK("mcr_scone_vis_forward");
    return 0;
}

// ---------------------------------------------------------------------------------------------------------
// SconeOcc.forward (SconeOcc.py:250-347).  The host supplies the already down-sampled clouds (it consumes
// torch's CPU generator exactly like the reference: randperm at :269 and :311):
//   pc_global [B, Lg, 3]            Lg = min(M, 2048)
//   pc_scale[i] [B, M_i, 3], i < 3  the cloud seen by scale i (M_0 = M, then M_i = M_{i-1} // ds_factor)
constexpr int OCC_NW = PCT_NW * 4 + 6 + 6, OCC_CHUNK = 16384;

size_t mcr_scone_occ_workspace_bytes(int64_t B, int64_t Q, int64_t Lg) {
    const int64_t qc = std::min<int64_t>(Q, OCC_CHUNK);
    size_t local = pct_ws_bytes(qc * 16) + al(qc * 16 * 3) + al(qc * 16) + al(qc * 16 * 2) + 1024;
    local = std::max(local, al(B * Q * 16 * 3) + al(Q * 16) + al(Q * 16 * 2) + 1024);  // fused path: kNN outputs for all B x Q rows
    size_t glob = pct_ws_bytes(B * Lg);
    size_t head = al(B * Q * 1344) + al(B * Q * 512) + al(B * Q * 256) + al(B * 512) * 2;
    head += linear3h_planes_bytes(512, 1344);        // split weight planes of the largest head layer (reused layer after layer)
    // grid-pruned kNN (knn.hip: K1-grid): the query order of all clouds + one sorted candidate copy (the largest admissible cloud)
    const size_t knn_grid = knn_grid_query_bytes(B, Q) + 3 * knn_grid_cloud_bytes(B, 16384) + knn_grid_park_bytes() + 1024;
    return local + glob + head + knn_grid + 8192;          // local and global paths run concurrently (two streams): disjoint scratch
}

int mcr_knn_points(const float* X, const float* pc, int64_t* idx, float* dists, float* pts, int64_t B, int64_t Q, int64_t M,
                   int k, int subtract_query, void* stream);

// The global feature (a chain of ~20 small launches on 2048 tokens, ~0.3 ms of mostly latency) depends on nothing the local
// path produces and is needed only by the first head layer: it runs on a side stream beside the kNN / local-transformer
// launches (fork / join by events, so a captured graph keeps the structure).  MCR_OCC_OVERLAP=0 puts it back on the caller's stream.
struct OccSide { hipStream_t s = nullptr; hipEvent_t fork = nullptr, join = nullptr; };
// One side stream + fork / join event pair per (device, CALLER stream): two host streams (or threads) running SconeOcc
// concurrently never share an event pair; creation is serialised.
// slot 0: the global transformer / the cloud build; slot 1: the x embedding queued with the early part (its own stream: on slot 0 the
// cloud build of the first search would queue up behind its GEMMs)
static OccSide* occ_side(hipStream_t caller, int slot = 0) {
    static const bool on = []() { const char* e = getenv("MCR_OCC_OVERLAP"); return !(e && e[0] == '0'); }();
    if (!on) return nullptr;
    static std::mutex mu;
    static std::map<std::pair<std::pair<int, int>, hipStream_t>, OccSide> table;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    OccSide& x = table[std::make_pair(std::make_pair(dev, slot), caller)];
    if (!x.s) {
        if (hipStreamCreateWithFlags(&x.s, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&x.fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&x.join, hipEventDisableTiming) != hipSuccess) {
            x.s = nullptr;
            return nullptr;
        }
    }
    return &x;
}

int mcr_scone_occ_forward_phase(const float* pc_global, int64_t Lg, const float* const* pc_scale, const int64_t* M_scale,
                                const float* x, const float* view_harmonics, float* out, int64_t B, int64_t Q,
                                const float* const* weights, int n_weights, const float* const* local_blobs,
                                const void* const* head_planes, const float* head_inv_scales, int* range_flag, void* workspace,
                                size_t workspace_bytes, int phase, void* stream);
int mcr_scone_occ_forward(const float* pc_global, int64_t Lg, const float* const* pc_scale, const int64_t* M_scale,
                          const float* x, const float* view_harmonics, float* out, int64_t B, int64_t Q,
                          const float* const* weights, int n_weights, const float* const* local_blobs,
                          const void* const* head_planes, const float* head_inv_scales, int* range_flag, void* workspace,
                          size_t workspace_bytes, void* stream) {
    return mcr_scone_occ_forward_phase(pc_global, Lg, pc_scale, M_scale, x, view_harmonics, out, B, Q, weights, n_weights, local_blobs,
                                       head_planes, head_inv_scales, range_flag, workspace, workspace_bytes, 0, stream);
}

// The same in two calls on one stream and ONE workspace.  Phase 1 = what needs neither the view harmonics nor any hidden draw: the
// query order of the grid search, scale 0 (the whole cloud: search + local transformer) -- it reads x, pc_scale[0] and M_scale[0..2]
// only (the sizes of the down-sampled clouds are known before they are drawn) and is the first long kernel of an NBV step, so a
// caller queues it BEFORE it builds the view state, the harmonics and the down-sampled clouds: the host work of those hides behind
// it instead of leaving the GPU idle at the start of the step.  Phase 2 = the rest (global transformer, scales 1 and 2, x
// embedding, head).  phase 0 = both, in the single-call order.
int mcr_scone_occ_forward_phase(const float* pc_global, int64_t Lg, const float* const* pc_scale, const int64_t* M_scale,
                                const float* x, const float* view_harmonics, float* out, int64_t B, int64_t Q,
                                const float* const* weights, int n_weights, const float* const* local_blobs,
                                const void* const* head_planes, const float* head_inv_scales, int* range_flag, void* workspace,
                                size_t workspace_bytes, int phase, void* stream) {
    VariantScope variant_scope_;
    MCR_REQUIRE(phase >= 0 && phase <= 2, "mcr_scone_occ_forward: phase must be 0, 1 or 2");
    const bool early = phase != 2, late = phase != 1;
    MCR_REQUIRE(pc_scale && M_scale && x && weights && (!late || (pc_global && view_harmonics && out)), "mcr_scone_occ_forward: null pointer");
    MCR_REQUIRE(!head_planes || head_inv_scales, "mcr_scone_occ_forward: head_planes need head_inv_scales");
    MCR_REQUIRE(n_weights == OCC_NW || n_weights == OCC_NW + 8 || n_weights == OCC_NW + 11,
                "mcr_scone_occ_forward: expected %d weight pointers (+ 8 or 11 plane pointers), got %d", OCC_NW, n_weights);
    MCR_REQUIRE(B > 0 && Q > 0 && Lg > 0 && B <= 65535, "mcr_scone_occ_forward: bad problem size");
    MCR_REQUIRE(workspace && workspace_bytes >= mcr_scone_occ_workspace_bytes(B, Q, Lg), "mcr_scone_occ_forward: workspace too small");
    for (int i = 0; i < n_weights; ++i) MCR_REQUIRE(weights[i], "mcr_scone_occ_forward: weight %d is null", i);
    for (int i = 0; i < 3; ++i)
        MCR_REQUIRE((pc_scale[i] || (!late && i > 0)) && M_scale[i] >= 16, "mcr_scone_occ_forward: scale %d has %ld points (< k = 16)", i,
                    (long)M_scale[i]);
    hipStream_t s = (hipStream_t)stream;
    const float* const* p = weights;
    PctW wg = read_pct(p);
    const PctW wl[3] = {read_pct(p), read_pct(p), read_pct(p)};
    LinW xe1{p[0], p[1]}, xe2{p[2], p[3]}, xe3{p[4], p[5]};
    p += 6;
    LinW lin1{p[0], p[1]}, lin2{p[2], p[3]}, lin3{p[4], p[5]};
    p += 6;
    if (n_weights >= OCC_NW + 8) { read_enc_planes(p, wg.enc[0]); read_enc_planes(p, wg.enc[1]); }     // the global transformer's encoders
    if (n_weights == OCC_NW + 11) read_pct_end_planes(p, wg);                                           // ... and its end layers

    Arena head{(char*)workspace, workspace_bytes, 0};
    // per-query feature row: [ local 3x256 | x-embedding 512 | view harmonics 64 ] = 1344   (cat at SconeOcc.py:333
    // is (global 512, local 768, x 512, harmonics 64); the per-cloud global part is folded into a row bias)
    constexpr int FEAT = 1344;
    float* feat = head.f(B * Q * FEAT);
    float* h1 = head.f(B * Q * 512);
    float* h2 = head.f(B * Q * 256);
    float* gfeat = head.f(B * 512);
    float* gbias = head.f(B * 512);
    void* wplanes = head.f(linear3h_planes_bytes(512, 1344) / sizeof(float));
    const size_t knn_q_bytes = knn_grid_query_bytes(B, Q), knn_c_bytes = knn_grid_cloud_bytes(B, 16384);
    char* knn_q_ws = (char*)head.f((knn_q_bytes + 3) / 4);
    char* knn_c_ws = (char*)head.f((3 * knn_c_bytes + 3) / 4);
    char* knn_park_ws = (char*)head.f((knn_grid_park_bytes() + 3) / 4);
    const size_t glob_bytes = pct_ws_bytes(B * Lg);
    Arena garena{(char*)workspace + head.off, glob_bytes, 0};
    Arena scratch{(char*)workspace + head.off + glob_bytes, workspace_bytes - head.off - glob_bytes, 0};

    // ---- global feature (SconeOcc.py:269-277), on the side stream ----
    OccSide* side = late ? occ_side(s) : nullptr;
    hipStream_t gs = s;
    if (side && hipEventRecord(side->fork, s) == hipSuccess && hipStreamWaitEvent(side->s, side->fork, 0) == hipSuccess) gs = side->s;
    else side = nullptr;
    // every way out of this function joins the side stream again (error returns included: a dangling fork would poison a capture and
    // let side-stream work outlive the caller's workspace): the guard exists before anything is queued on the side stream, and on an
    // early return it records the join itself
    struct SideJoin {
        OccSide* side; hipStream_t s; bool recorded, joined;
        ~SideJoin() {
            if (!side || joined) return;
            if (!recorded) (void)hipEventRecord(side->join, side->s);
            (void)hipStreamWaitEvent(s, side->join, 0);
        }
    } side_join{side, s, false, false};
    // fused path: one kNN + one LDS-resident transformer launch per (cloud, scale) over ALL queries (nothing but
    // the [Q,16,3] offsets is materialised); layer-by-layer path: chunked over queries to bound its workspace.
    const bool fused_all = local_blobs && local_blobs[0] && local_blobs[1] && local_blobs[2];
    const int64_t qc = fused_all ? Q : std::min<int64_t>(Q, OCC_CHUNK);
    // variant 6 with all three fused transformers: the head runs on fp16 hi/lo planes end to end (run_head_planes); the feature
    // buffer then holds planes [2][T][1344] fp16 instead of fp32 [T][1344] (MCR_HEAD_PLANES=0: the fp32-input linear3h path)
    static const bool planes_on = []() { const char* e = getenv("MCR_HEAD_PLANES"); return !(e && e[0] == '0'); }();
    const bool planes = planes_on && fused_all && fp16_planes_variant();
    _Float16* featP = reinterpret_cast<_Float16*>(feat);
    const int64_t Tall = B * Q;
    // the x embedding of the planes head (0.3 ms of GEMMs that need only the queries) rides on the side stream behind the global
    // transformer: its workgroups fill the machine in the holes of the local path (kNN preparation, the parked kNN groups, kernel
    // tails) instead of extending the serial tail of the step.  MCR_OCC_X_SIDE=0: on the caller's stream, after the local path.
    static const bool x_side_on = []() { const char* e = getenv("MCR_OCC_X_SIDE"); return !(e && e[0] == '0'); }();
    const HeadScratch head_scratch{featP, reinterpret_cast<_Float16*>(h1), h2, wplanes};
    // ... and it needs nothing but the queries, so it is queued with the EARLY part (phase 1 / the start of the single call), where the
    // GPU is nearly idle for ~0.4 ms (query order, cloud build, the scale-0 search): behind the global transformer it only found the
    // holes between the local-transformer launches and finished 0.15 ms AFTER the last of them -- the head waited for it
    // (profiles/r04_nbv_step_breakdown.txt).  The view harmonics (known in phase 2 only) are split there.  MCR_OCC_X_EARLY=0: as before.
    static const bool x_early_on = []() { const char* e = getenv("MCR_OCC_X_EARLY"); return !(e && e[0] == '0'); }();
    OccSide* xside = planes && x_side_on && x_early_on && occ_side(s) ? occ_side(s, 1) : nullptr;
    const bool x_early = xside != nullptr;               // (the same answer in phase 1 and in phase 2 of one forward)
    const bool x_on_side = planes && side && x_side_on;
    // the caller's stream waits for the early x embedding BEHIND the early part's own kernels (end of phase 1 / before the head); every
    // way out of the function in between queues that wait too (a dangling fork would poison a capture)
    struct XJoin {
        OccSide* side; hipStream_t s; bool armed;
        bool wait() { if (!armed) return true; armed = false; return hipStreamWaitEvent(s, side->join, 0) == hipSuccess; }
        ~XJoin() { (void)wait(); }
    } x_join{xside, s, false};
    if (x_early && early) {
        MCR_REQUIRE(hipEventRecord(xside->fork, s) == hipSuccess && hipStreamWaitEvent(xside->s, xside->fork, 0) == hipSuccess,
                    "mcr_scone_occ_forward: side stream (x embedding fork)");
        run_x_embedding_planes(xside->s, x, nullptr, B * Q, xe1, xe2, xe3, head_planes, head_inv_scales, head_scratch);
        MCR_REQUIRE(hipEventRecord(xside->join, xside->s) == hipSuccess, "mcr_scone_occ_forward: side stream (x embedding record)");
        x_join.armed = true;
    }
    if (late) {
        run_pct(gs, wg, pc_global, gfeat, 512, B, (int)Lg, 256, garena);
        MCR_REQUIRE(garena.ok(), "mcr_scone_occ_forward: workspace overflow (global)");
        // its contribution to linear1: gbias[b, n] = sum_k gfeat[b, k] * W1[n, k]  (columns 0..511 of linear1.weight)
        launch_linear(gs, gfeat, 512, lin1.w, nullptr, nullptr, 0, gbias, 512, B, 512, 512, ACT_NONE, nullptr, 0, 1856, 1);
        if (x_on_side && x_early)
            launch_split_to_planes(gs, view_harmonics, 64, featP + 1280, matrix_planes() == 1 ? nullptr : featP + Tall * FEAT + 1280, 1344, B * Q, 64);
        else if (x_on_side) run_x_embedding_planes(gs, x, view_harmonics, B * Q, xe1, xe2, xe3, head_planes, head_inv_scales, head_scratch);
    }
    if (side) {
        MCR_REQUIRE(hipEventRecord(side->join, side->s) == hipSuccess, "mcr_scone_occ_forward: side stream (record)");
        side_join.recorded = true;
    }
    // ---- local multi-scale neighbourhood features (SconeOcc.py:290-311) ----
    // grid-pruned kNN for the scales it applies to (whole-Q launches only): ONE query order for all three scales
    const int* knn_qperm = nullptr;
    KnnGridCloud knn_clouds[3]{};
    int knn_slot[3] = {-1, -1, -1};
    if (qc == Q) {
        const float* g_pc[3]; int64_t g_M[3]; void* g_ws[3];
        int n_grid = 0;
        for (int sc = 0; sc < 3; ++sc)
            if (knn_grid_applicable(M_scale[sc], 16)) {
                g_pc[n_grid] = pc_scale[sc]; g_M[n_grid] = M_scale[sc]; g_ws[n_grid] = knn_c_ws + n_grid * knn_c_bytes;
                knn_slot[sc] = n_grid++;
            }
        if (n_grid) {
            // the query order belongs to phase 1 (phase 2 finds it where phase 1 left it); every scale's cloud is built in the
            // phase that searches it -- all of them in one launch on the single call
            if (phase == 0) {
                knn_qperm = knn_grid_order_queries(s, x, B, Q, knn_q_ws, knn_park_ws);
                knn_grid_build_clouds(s, n_grid, g_pc, g_M, B, g_ws, knn_clouds);
            } else {
                const int first = early ? 0 : (knn_slot[0] >= 0 ? 1 : 0), count = early ? (knn_slot[0] >= 0 ? 1 : 0) : n_grid - first;
                // phase 1 opens the step on an idle GPU: the cloud build (one workgroup per cloud, 40 us) runs on the side stream beside
                // the five small launches of the query order instead of after them
                OccSide* bside = early && count > 0 ? occ_side(s) : nullptr;
                if (bside && !(hipEventRecord(bside->fork, s) == hipSuccess && hipStreamWaitEvent(bside->s, bside->fork, 0) == hipSuccess)) bside = nullptr;
                knn_grid_build_clouds(bside ? bside->s : s, count, g_pc + first, g_M + first, B, g_ws + first, knn_clouds + first);
                const bool recorded = !bside || hipEventRecord(bside->join, bside->s) == hipSuccess;
                knn_qperm = knn_grid_order_queries(s, x, B, Q, knn_q_ws, knn_park_ws, /*launch=*/early);
                const bool joined = !bside || hipStreamWaitEvent(s, bside->join, 0) == hipSuccess;        // (waits for a stale record at worst)
                MCR_REQUIRE(recorded && joined, "mcr_scone_occ_forward: side stream (cloud build)");
            }
            MCR_LAUNCH_CHECK("knn grid preparation");
        }
    }
    for (int sc = 0; sc < 3; ++sc) {
        if (sc == 0 ? !early : !late) continue;
        const bool grid_knn = knn_slot[sc] >= 0;
        const KnnGridCloud knn_cloud = grid_knn ? knn_clouds[knn_slot[sc]] : KnnGridCloud{};
        // a batch of clouds on the fused path (config 3's scene batch): ONE search and ONE transformer launch per scale over all B x Q
        // rows instead of one per cloud -- eight brute-force searches of 256 workgroups each (one wave per SIMD: every wave waits out
        // its own latencies) become one of 2048.  The rows are the same rows: same bits.  MCR_OCC_BATCH_LOCAL=0: per cloud (A/B)
        static const bool batch_local_on = []() { const char* e = getenv("MCR_OCC_BATCH_LOCAL"); return !(e && e[0] == '0'); }();
        if (batch_local_on && B > 1 && qc == Q && (planes || (local_blobs && local_blobs[sc]))) {
            Arena a = scratch;
            float* offs = a.f(Tall * 16 * 3);
            if (a.ok()) {
                if (grid_knn) {
                    launch_knn16_grid(s, x, pc_scale[sc], M_scale[sc], knn_qperm, knn_cloud, 0, B, Q, nullptr, nullptr, offs, true, knn_park_ws, sc);
                    MCR_LAUNCH_CHECK("knn_grid_kernel");
                } else if (int e = mcr_knn_points(x, pc_scale[sc], nullptr, nullptr, offs, B, Q, M_scale[sc], 16, 1, stream))
                    return e;
                if (planes) run_local_pct(s, offs, nullptr, FEAT, Tall, local_blobs[sc], featP + sc * 256, featP + Tall * FEAT + sc * 256);
                else run_local_pct(s, offs, feat + sc * 256, FEAT, Tall, local_blobs[sc]);
                continue;
            }                                              // (workspace sized by an older caller: the per-cloud form below)
        }
        for (int64_t q0 = 0; q0 < Q; q0 += qc) {
            const int64_t nq = std::min<int64_t>(qc, Q - q0);
            for (int64_t b = 0; b < B; ++b) {
                Arena a = scratch;
                float* offs = a.f(nq * 16 * 3);
                MCR_REQUIRE(a.ok(), "mcr_scone_occ_forward: workspace overflow (kNN)");
                // only the offsets are consumed (SconeOcc.py:297-298): indices and distances are not written
                if (grid_knn) {
                    launch_knn16_grid(s, x, pc_scale[sc], M_scale[sc], knn_qperm, knn_cloud, b, 1, Q, nullptr, nullptr, offs, true, knn_park_ws,
                                      (int)(sc * B + b));
                    MCR_LAUNCH_CHECK("knn_grid_kernel");
                } else if (int e = mcr_knn_points(x + (b * Q + q0) * 3, pc_scale[sc] + b * M_scale[sc] * 3, nullptr, nullptr, offs, 1, nq,
                                                  M_scale[sc], 16, 1, stream))
                    return e;
                if (planes)
                    run_local_pct(s, offs, nullptr, FEAT, nq, local_blobs[sc], featP + (b * Q + q0) * FEAT + sc * 256,
                                  featP + Tall * FEAT + (b * Q + q0) * FEAT + sc * 256);
                else if (local_blobs && local_blobs[sc])      // fused LDS-resident kernel (local_pct.hip)
                    run_local_pct(s, offs, feat + (b * Q + q0) * FEAT + sc * 256, FEAT, nq, local_blobs[sc]);
                else                                      // layer-by-layer path through HBM
                    run_pct(s, wl[sc], offs, feat + (b * Q + q0) * FEAT + sc * 256, FEAT, nq, 16, 128, a);
                MCR_REQUIRE(a.ok(), "mcr_scone_occ_forward: workspace overflow (local)");
            }
        }
    }
    MCR_REQUIRE(x_join.wait(), "mcr_scone_occ_forward: side stream (x embedding join)");
    if (!late) {
        MCR_LAUNCH_CHECK("mcr_scone_occ_forward (phase 1)");
        return 0;
    }
    // The large layers run on the matrix path of the selected variant -- 6: fp16 x 3 with the weights split once per call into
    // `wplanes`; 5: bf16 x 6 (exact hi/mid/lo); 1: exact fp32 MFMA -- chosen by the variant and the layer alone, never by the
    // number of rows: a query's occupancy must not depend on how many other queries share the launch (query shards of the
    // multi-GPU step, chunks, scene batches and the single call agree bit for bit).
    const int variant = g_local_pct_variant;
    const int64_t ANY_M = (int64_t)1 << 40;
    // which: 0 xe2, 1 xe3, 2 lin1 (columns 512..1855), 3 lin2 -- the order of the host's pre-split planes (variant 6)
    auto big_linear = [&](int which, const float* X_, int64_t ldx, const float* W_, int64_t ldw, const float* b_, float* Y_, int64_t ldy,
                          int64_t M_, int N_, int K_, const float* rb, int64_t rpg) {
        if ((variant == 6 || variant == 7) && linear3h_applicable(X_, ldx, W_, ldw, ANY_M, N_, K_)) {
            const bool pre = head_planes && head_planes[which] && head_inv_scales[which] > 0.f;
            launch_linear3h(s, X_, ldx, W_, ldw, pre ? const_cast<void*>(head_planes[which]) : wplanes, b_, nullptr, 0, Y_, ldy, M_, N_, K_,
                            ACT_GELU, rb, rpg, pre ? head_inv_scales[which] : 0.f);
        }
        else if (variant == 5 && linear3_applicable(X_, ldx, W_, ldw, ANY_M, N_, K_))
            launch_linear3(s, X_, ldx, W_, b_, nullptr, 0, Y_, ldy, M_, N_, K_, ACT_GELU, rb, rpg, ldw);
        else
            launch_linear(s, X_, ldx, W_, b_, nullptr, 0, Y_, ldy, M_, N_, K_, ACT_GELU, rb, rpg, ldw, /*route_rows=*/1);
    };
    const int64_t T = B * Q;
    if (planes) {
        bool join_failed = false;
        run_head_planes(s, x, view_harmonics, T, xe1, xe2, xe3, lin1, lin2, lin3, gbias, Q, nullptr, head_planes, head_inv_scales,
                        head_scratch, out, [&]() {
                            if (side) {
                                side_join.joined = true;
                                join_failed = hipStreamWaitEvent(s, side->join, 0) != hipSuccess;
                            }
                        }, x_on_side);
        MCR_REQUIRE(!join_failed, "mcr_scone_occ_forward: side stream (join)");
        if (range_flag) launch_nonfinite_flag(s, out, T, range_flag);
        MCR_LAUNCH_CHECK("mcr_scone_occ_forward");
        return 0;
    }
    // ---- x embedding 3 -> 128 -> 256 -> 512, GELU each (SconeOcc.py:35-42) ----
    launch_linear(s, x, 3, xe1.w, xe1.b, nullptr, 0, h2, 128, T, 128, 3, ACT_GELU, nullptr, 0, 0, 1);
    big_linear(0, h2, 128, xe2.w, 128, xe2.b, h1, 256, T, 256, 128, nullptr, 0);
    big_linear(1, h1, 256, xe3.w, 256, xe3.b, feat + 768, FEAT, T, 512, 256, nullptr, 0);
    launch_copy2d(s, view_harmonics, 64, feat + 1280, FEAT, T, 64);
    // ---- head MLP 1856 -> 512 -> 256 -> 1, GELU after every layer incl. the last (SconeOcc.py:334-345) ----
    if (side) {
        side_join.joined = true;
        MCR_REQUIRE(hipStreamWaitEvent(s, side->join, 0) == hipSuccess, "mcr_scone_occ_forward: side stream (join)");
    }
    big_linear(2, feat, FEAT, lin1.w + 512, 1856, lin1.b, h1, 512, T, 512, FEAT, gbias, Q);
    big_linear(3, h1, 512, lin2.w, 512, lin2.b, h2, 256, T, 256, 512, nullptr, 0);
    launch_linear(s, h2, 256, lin3.w, lin3.b, nullptr, 0, out, 1, T, 1, 256, ACT_GELU, nullptr, 0, 0, 1);
    // range guard of the fp16 split path: an activation beyond the fp16 range (|x| >= 65520) becomes inf in its high plane and
    // reaches the output as a non-finite occupancy (inf - inf in the accumulators, NaN through LayerNorm / soft-max / the mean
    // pooling); the caller re-runs on the full-range variant 5 when the flag comes back set
    if (range_flag) launch_nonfinite_flag(s, out, T, range_flag);
    MCR_LAUNCH_CHECK("mcr_scone_occ_forward");
    return 0;
}

// ---------------------------------------------------------------------------------------------------------
// Ragged SconeOcc: J independent SconeOcc.forward calls ("jobs": one surface cloud + one chunk of queries each, clouds and
// chunks of different sizes -- the per-cell passes of compute_scene_occupancy_probability_field, macarons_utils.py:1395-1540)
// as ONE launch sequence.  Job j: global cloud pc_global[j] (Lg rows, the first global_len[j] valid), neighbourhood clouds
// pc_scale[s][scale_off[s][j] .. scale_off[s][j+1]), queries = the rows r of x with row_job[r] == j (rows sorted by job).
// knn_blocks: n_blocks x int4 (job, first row, rows <= mcr_knn_rows_per_block(), 0) covering every row once.
size_t mcr_scone_occ_ragged_workspace_bytes(int64_t J, int64_t T, int64_t Lg) {
    size_t local = al(T * 16 * 3) + al(knn16_segmented_split_floats(T)) + 1024;
    size_t glob = pct_ws_bytes(J * Lg);
    size_t head = al(T * 1344) + al(T * 512) + al(T * 256) + al(J * 512) * 2 + linear3h_planes_bytes(512, 1344);
    return local + glob + head + 8192;
}
int mcr_knn_rows_per_block(void) { return knn_rows_per_block(); }

int mcr_scone_occ_forward_ragged_phase(const float* pc_global, const int* global_len, int64_t Lg, const float* const* pc_scale,
                                       const int64_t* const* scale_off, const float* x, const float* view_harmonics, const int* row_job,
                                       const int* knn_blocks, int64_t n_blocks, float* out, int64_t J, int64_t T,
                                       const float* const* weights, int n_weights, const float* const* local_blobs,
                                       const void* const* head_planes, const float* head_inv_scales, int* range_flag, void* workspace,
                                       size_t workspace_bytes, int phase, void* stream);
int mcr_scone_occ_forward_ragged(const float* pc_global, const int* global_len, int64_t Lg, const float* const* pc_scale,
                                 const int64_t* const* scale_off, const float* x, const float* view_harmonics, const int* row_job,
                                 const int* knn_blocks, int64_t n_blocks, float* out, int64_t J, int64_t T,
                                 const float* const* weights, int n_weights, const float* const* local_blobs,
                                 const void* const* head_planes, const float* head_inv_scales, int* range_flag, void* workspace,
                                 size_t workspace_bytes, void* stream) {
    return mcr_scone_occ_forward_ragged_phase(pc_global, global_len, Lg, pc_scale, scale_off, x, view_harmonics, row_job, knn_blocks, n_blocks, out,
                                              J, T, weights, n_weights, local_blobs, head_planes, head_inv_scales, range_flag, workspace,
                                              workspace_bytes, 0, stream);
}

// The same in two calls on one stream and ONE workspace: phase 1 = everything that needs none of the hidden random draws (scale 0:
// the whole clouds; the x embedding on the planes path), phase 2 = the rest (global transformer, scales 1 and 2, head).  The host
// makes the draws (~60 us of torch.randperm per job) between the two calls while the GPU works on phase 1.  phase 0 = both.
// Phase 1 reads pc_scale[0], scale_off[0], x, view_harmonics, row_job, knn_blocks, local_blobs[0]; phase 2 everything else too.
int mcr_scone_occ_forward_ragged_phase(const float* pc_global, const int* global_len, int64_t Lg, const float* const* pc_scale,
                                       const int64_t* const* scale_off, const float* x, const float* view_harmonics, const int* row_job,
                                       const int* knn_blocks, int64_t n_blocks, float* out, int64_t J, int64_t T,
                                       const float* const* weights, int n_weights, const float* const* local_blobs,
                                       const void* const* head_planes, const float* head_inv_scales, int* range_flag, void* workspace,
                                       size_t workspace_bytes, int phase, void* stream) {
    VariantScope variant_scope_;
    MCR_REQUIRE(phase >= 0 && phase <= 2, "mcr_scone_occ_forward_ragged: phase must be 0, 1 or 2");
    const bool early = phase != 2, late = phase != 1;
    MCR_REQUIRE(pc_scale && scale_off && x && view_harmonics && row_job && knn_blocks && weights && (!late || (pc_global && global_len && out)),
                "mcr_scone_occ_forward_ragged: null pointer");
    MCR_REQUIRE(n_weights == OCC_NW || n_weights == OCC_NW + 8 || n_weights == OCC_NW + 11,
                "mcr_scone_occ_forward_ragged: expected %d weight pointers (+ 8 or 11 plane pointers), got %d", OCC_NW, n_weights);
    MCR_REQUIRE(J > 0 && T > 0 && Lg > 0 && J <= 32767 && n_blocks > 0, "mcr_scone_occ_forward_ragged: bad problem size");
    MCR_REQUIRE(local_blobs && local_blobs[0] && local_blobs[1] && local_blobs[2],
                "mcr_scone_occ_forward_ragged: needs the fused local-transformer blobs");
    MCR_REQUIRE(!head_planes || head_inv_scales, "mcr_scone_occ_forward_ragged: head_planes need head_inv_scales");
    MCR_REQUIRE(workspace && workspace_bytes >= mcr_scone_occ_ragged_workspace_bytes(J, T, Lg), "mcr_scone_occ_forward_ragged: workspace too small");
    for (int i = 0; i < n_weights; ++i) MCR_REQUIRE(weights[i], "mcr_scone_occ_forward_ragged: weight %d is null", i);
    for (int i = 0; i < (late ? 3 : 1); ++i) MCR_REQUIRE(pc_scale[i] && scale_off[i], "mcr_scone_occ_forward_ragged: scale %d is null", i);
    hipStream_t s = (hipStream_t)stream;
    const float* const* p = weights;
    PctW wg = read_pct(p);
    for (int i = 0; i < 3; ++i) (void)read_pct(p);
    LinW xe1{p[0], p[1]}, xe2{p[2], p[3]}, xe3{p[4], p[5]};
    p += 6;
    LinW lin1{p[0], p[1]}, lin2{p[2], p[3]}, lin3{p[4], p[5]};
    p += 6;
    if (n_weights >= OCC_NW + 8) { read_enc_planes(p, wg.enc[0]); read_enc_planes(p, wg.enc[1]); }     // the global transformer's encoders
    if (n_weights == OCC_NW + 11) read_pct_end_planes(p, wg);                                           // ... and its end layers

    Arena head{(char*)workspace, workspace_bytes, 0};
    constexpr int FEAT = 1344;
    float* feat = head.f(T * FEAT);
    float* h1 = head.f(T * 512);
    float* h2 = head.f(T * 256);
    float* gfeat = head.f(J * 512);
    float* gbias = head.f(J * 512);
    void* wplanes = head.f(linear3h_planes_bytes(512, 1344) / sizeof(float));
    const size_t glob_bytes = pct_ws_bytes(J * Lg);
    Arena garena{(char*)workspace + head.off, glob_bytes, 0};
    Arena scratch{(char*)workspace + head.off + glob_bytes, workspace_bytes - head.off - glob_bytes, 0};
    static const bool planes_on = []() { const char* e = getenv("MCR_HEAD_PLANES"); return !(e && e[0] == '0'); }();
    const bool planes = planes_on && fp16_planes_variant();
    _Float16* featP = reinterpret_cast<_Float16*>(feat);
    const HeadScratch head_scratch{featP, reinterpret_cast<_Float16*>(h1), h2, wplanes};
    float* offs = scratch.f(T * 16 * 3);
    float* knn_split_ws = scratch.f(knn16_segmented_split_floats(T));
    MCR_REQUIRE(scratch.ok(), "mcr_scone_occ_forward_ragged: workspace overflow (kNN)");
    auto local_scale = [&](int sc) {                      // one segmented kNN + one fused transformer launch over ALL rows
        // (the whole clouds of scale 0 are the large ones: a launch with few query blocks cuts every job's candidates into slices of
        // ~2048 for more workgroups; slicing the down-sampled clouds of the coarser scales too -- 512 per slice -- measured no better)
        launch_knn16_segmented(s, x, pc_scale[sc], (const long long*)scale_off[sc], knn_blocks, n_blocks, T, offs, knn_split_ws, sc == 0 ? 2048 : 0);
        if (planes) run_local_pct(s, offs, nullptr, FEAT, T, local_blobs[sc], featP + sc * 256, featP + T * FEAT + sc * 256);
        else run_local_pct(s, offs, feat + sc * 256, FEAT, T, local_blobs[sc]);
    };
    if (early) {
        if (planes && phase == 1) run_x_embedding_planes(s, x, view_harmonics, T, xe1, xe2, xe3, head_planes, head_inv_scales, head_scratch);
        local_scale(0);
        if (!late) {
            MCR_LAUNCH_CHECK("mcr_scone_occ_forward_ragged (phase 1)");
            return 0;
        }
    }

    OccSide* side = occ_side(s);
    hipStream_t gs = s;
    if (side && hipEventRecord(side->fork, s) == hipSuccess && hipStreamWaitEvent(side->s, side->fork, 0) == hipSuccess) gs = side->s;
    else side = nullptr;
    struct SideJoin {
        OccSide* side; hipStream_t s; bool recorded, joined;
        ~SideJoin() {
            if (!side || joined) return;
            if (!recorded) (void)hipEventRecord(side->join, side->s);
            (void)hipStreamWaitEvent(s, side->join, 0);
        }
    } side_join{side, s, false, false};
    run_pct(gs, wg, pc_global, gfeat, 512, J, (int)Lg, 256, garena, global_len);
    MCR_REQUIRE(garena.ok(), "mcr_scone_occ_forward_ragged: workspace overflow (global)");
    launch_linear(gs, gfeat, 512, lin1.w, nullptr, nullptr, 0, gbias, 512, J, 512, 512, ACT_NONE, nullptr, 0, 1856, 1);
    if (side) {
        MCR_REQUIRE(hipEventRecord(side->join, side->s) == hipSuccess, "mcr_scone_occ_forward_ragged: side stream (record)");
        side_join.recorded = true;
    }
    // ---- local features of scales 1 and 2 (scale 0: above) ----
    for (int sc = 1; sc < 3; ++sc) local_scale(sc);
    if (planes) {
        bool join_failed = false;
        run_head_planes(s, x, view_harmonics, T, xe1, xe2, xe3, lin1, lin2, lin3, gbias, 0, row_job, head_planes, head_inv_scales,
                        head_scratch, out, [&]() {
                            if (side) {
                                side_join.joined = true;
                                join_failed = hipStreamWaitEvent(s, side->join, 0) != hipSuccess;
                            }
                        }, /*x_done=*/phase == 2);
        MCR_REQUIRE(!join_failed, "mcr_scone_occ_forward_ragged: side stream (join)");
        if (range_flag) launch_nonfinite_flag(s, out, T, range_flag);
        MCR_LAUNCH_CHECK("mcr_scone_occ_forward_ragged");
        return 0;
    }
    const int variant = g_local_pct_variant;
    const int64_t ANY_M = (int64_t)1 << 40;
    auto big_linear = [&](int which, const float* X_, int64_t ldx, const float* W_, int64_t ldw, const float* b_, float* Y_, int64_t ldy,
                          int64_t M_, int N_, int K_, const float* rb, const int* rg) {
        if ((variant == 6 || variant == 7) && linear3h_applicable(X_, ldx, W_, ldw, ANY_M, N_, K_)) {
            const bool pre = head_planes && head_planes[which] && head_inv_scales[which] > 0.f;
            launch_linear3h(s, X_, ldx, W_, ldw, pre ? const_cast<void*>(head_planes[which]) : wplanes, b_, nullptr, 0, Y_, ldy, M_, N_, K_,
                            ACT_GELU, rb, 0, pre ? head_inv_scales[which] : 0.f, rg);
        } else if (variant == 5 && linear3_applicable(X_, ldx, W_, ldw, ANY_M, N_, K_))
            launch_linear3(s, X_, ldx, W_, b_, nullptr, 0, Y_, ldy, M_, N_, K_, ACT_GELU, rb, 0, ldw, rg);
        else
            launch_linear(s, X_, ldx, W_, b_, nullptr, 0, Y_, ldy, M_, N_, K_, ACT_GELU, rb, 0, ldw, /*route_rows=*/1, rg);
    };
    launch_linear(s, x, 3, xe1.w, xe1.b, nullptr, 0, h2, 128, T, 128, 3, ACT_GELU, nullptr, 0, 0, 1);
    big_linear(0, h2, 128, xe2.w, 128, xe2.b, h1, 256, T, 256, 128, nullptr, nullptr);
    big_linear(1, h1, 256, xe3.w, 256, xe3.b, feat + 768, FEAT, T, 512, 256, nullptr, nullptr);
    launch_copy2d(s, view_harmonics, 64, feat + 1280, FEAT, T, 64);
    if (side) {
        side_join.joined = true;
        MCR_REQUIRE(hipStreamWaitEvent(s, side->join, 0) == hipSuccess, "mcr_scone_occ_forward_ragged: side stream (join)");
    }
    big_linear(2, feat, FEAT, lin1.w + 512, 1856, lin1.b, h1, 512, T, 512, FEAT, gbias, row_job);
    big_linear(3, h1, 512, lin2.w, 512, lin2.b, h2, 256, T, 256, 512, nullptr, nullptr);
    launch_linear(s, h2, 256, lin3.w, lin3.b, nullptr, 0, out, 1, T, 1, 256, ACT_GELU, nullptr, 0, 0, 1);
    if (range_flag) launch_nonfinite_flag(s, out, T, range_flag);
    MCR_LAUNCH_CHECK("mcr_scone_occ_forward_ragged");
    return 0;
}

}  // extern "C"
