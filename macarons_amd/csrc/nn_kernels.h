// Device-side building blocks of the SCONE networks (K4-K8 of SURVEY §2.3) for gfx950.
// Launch helpers are declared here and defined in nn_kernels.hip; networks.hip composes them.
#pragma once
#include "common.h"

namespace mcr {

enum Act { ACT_NONE = 0, ACT_GELU = 1 };

// Y[m, n] = act( sum_k X[m*ldx + k] * W[n*K + k] + bias[n] ) (+ R[m*ldr + n]);   m < M, n < N
// nn.Linear semantics (W is [N,K] row-major).  fp32 MFMA (v_mfma_f32_32x32x2_f32), exact-fp32 products.
// row_bias (optional): extra bias per group of rows, row_bias[(m / rows_per_group) * N + n]  (used to fold the
// per-cloud global feature of SconeOcc into the head's first layer without materialising the concat).
// route_rows (0 = M): the row count the fp32 / split-precision routing decision is taken on.  The networks pass the rows of ONE
// cloud / sequence, so that a cloud's numerics do not depend on how many other clouds share the launch (a scene batch, a
// query shard of a multi-GPU step and the single-cloud call then agree bit for bit); the tiling (nt) may still follow M: every
// fp32 tiling accumulates k in the same order.
void launch_linear(hipStream_t s, const float* X, int64_t ldx, const float* W, const float* bias, const float* R,
                   int64_t ldr, float* Y, int64_t ldy, int64_t M, int N, int K, int act,
                   const float* row_bias = nullptr, int64_t rows_per_group = 0, int64_t ldw = 0, int64_t route_rows = 0,
                   const int* row_group = nullptr);   // row_group (optional, device int per row): the row_bias group of row m instead of m / rows_per_group

// Split-precision (exact bf16 hi/mid/lo, six MFMAs per product) variant for the large GEMMs (linear3.hip); launch_linear
// routes to it when linear3_applicable().
bool linear3_shape_ok(const float* X, int64_t ldx, const float* W, int64_t ldw, int N, int K);
bool linear3_applicable(const float* X, int64_t ldx, const float* W, int64_t ldw, int64_t M, int N, int K);
void launch_linear3(hipStream_t s, const float* X, int64_t ldx, const float* W, const float* bias, const float* R, int64_t ldr,
                    float* Y, int64_t ldy, int64_t M, int N, int K, int act, const float* row_bias, int64_t rows_per_group,
                    int64_t ldw, const int* row_group = nullptr);

// Two-term fp16 split variant (linear3h.hip, three MFMAs per product): the weights are split once per call into `planes`
// (linear3h_planes_bytes(N, K) bytes of scratch).  Range |x| < 65504, |w| < 255.
bool linear3h_applicable(const float* X, int64_t ldx, const float* W, int64_t ldw, int64_t M, int N, int K);
size_t linear3h_planes_bytes(int N, int K);
void launch_linear3h(hipStream_t s, const float* X, int64_t ldx, const float* W, int64_t ldw, void* planes, const float* bias,
                     const float* R, int64_t ldr, float* Y, int64_t ldy, int64_t M, int N, int K, int act, const float* row_bias,
                     int64_t rows_per_group, float presplit_inv_scale = 0.f, const int* row_group = nullptr);

// Row LayerNorm (eps 1e-5, affine): Y[m, :E] = (X[m, :E] - mean) * rstd * g + b     (Attention.py:274,292)
void launch_layernorm(hipStream_t s, const float* X, int64_t ldx, const float* g, const float* b, float* Y, int64_t ldy,
                      int64_t M, int E);

// Multi-head self-attention core on a packed QKV buffer (Attention.py:8-36,174-198; mask=None, no dropout):
//   qkv[m, 0:DQK | DQK:2DQK | 2DQK:2DQK+DV], head h owns channels [h*d,(h+1)*d); scores / sqrt(dqk_per_head);
//   out[m, h*dv:(h+1)*dv].   Sequences are S consecutive blocks of L rows.
//   lens (optional, device int per sequence): keys = the first min(L, lens[s]) rows (padded variable-length batches).
// mask (optional, bytes; Attention.py:24-27): pair (sequence s, head h, query q, key k) is masked where
//   mask[s * mask_seq_stride + h * mask_head_stride + q * mask_query_stride + k] == 0: its score becomes -1e3 BEFORE the 1/sqrt(d)
//   scale (so a fully masked query attends uniformly, as upstream); strides of 0 broadcast (a [S, L] key mask: query stride 0).
// split_ws (optional, attention_split_floats(S, L, H, DV) floats): lets one or two long sequences split their keys over two blocks
// (L >= 512 and at most 256 blocks otherwise); split_by_length: split whenever L >= 512, whatever S -- the networks use this so
// that a cloud's result does not depend on how many clouds share the launch (the two forms differ by summation order, ~1e-6).
void launch_attention(hipStream_t s, const float* qkv, int64_t ldq, float* out, int64_t ldo, int64_t S, int L, int H,
                      int DQK, int DV, const int* lens = nullptr, float* split_ws = nullptr, size_t split_ws_floats = 0,
                      bool split_by_length = false, bool pv_half = false, const unsigned char* mask = nullptr,
                      int64_t mask_seq_stride = 0, int64_t mask_head_stride = 0, int64_t mask_query_stride = 0,
                      void* planes_h = nullptr, void* planes_l = nullptr, int64_t ldp = 0, bool* planes_done = nullptr);
// planes_h / planes_l (optional): when the key-split form runs, its combine pass writes the result as fp16 hi/lo planes (row stride ldp
// halves) INSTEAD of fp32 rows of `out`; *planes_done says whether that happened (else `out` holds fp32 rows as usual)
size_t attention_split_floats(int64_t S, int L, int H, int DV);
// the merge pass of the key-split form: out <- (w0 out + w1 part1) / (w0 l0 + w1 l1) per (row, head), as fp32 rows or as planes
void launch_attention_combine(hipStream_t s, float* out, int64_t ldo, const float* part1, const float* ml, int64_t T, int H, int dv,
                              void* planes_h, void* planes_l, int64_t ldp);
// The same attention on a packed q | k | v operand that already is a pair of fp16 hi/lo planes (attention_planes.hip; row stride ldp
// halves): K / V tiles by LDS DMA, nothing split inside.  Result: planes Oh / Ol (row stride ldoh halves) when given, else fp32 rows of
// out (row stride ldo floats; out is also the scratch of the key-split form's first part).  split_mode: 1 = split the keys over two
// blocks whenever L >= 512 and split_ws is there, 0 = never, -1 = only when the unsplit grid leaves CUs idle.
bool attention_planes_applicable(int H, int DQK, int DV, int64_t ldp);
void launch_attention_planes(hipStream_t s, const void* Ph, const void* Pl, int64_t ldp, float* out, int64_t ldo, void* Oh, void* Ol,
                             int64_t ldoh, int64_t S, int L, int H, int DQK, int DV, const int* lens, float* split_ws, size_t split_ws_floats,
                             int split_mode, int n_planes = 2);     // n_planes = 1: the high planes alone (variant 7); Pl / Ol unused

// Column max over the L rows of each of S sequences, broadcast into a column slice of every row:
//   Y[(s*L + r)*ldy + c] = max_r' X[(s*L + r')*ldx + c], c < E          (Embedding global feature, Attention.py:117-121)
// `ex` (optional): ex_cols columns of a second [S*L, ex_ld] array copied to ex_dst (leading dimension ldy) on the way -- inside the
// same launch for the long sequences (L >= 512, ex_cols <= 16), by a copy launch otherwise.
void launch_colmax_broadcast(hipStream_t s, const float* X, int64_t ldx, float* Y, int64_t ldy, int64_t S, int L, int E,
                             const int* lens = nullptr, const float* ex = nullptr, int ex_ld = 0, int ex_cols = 0, float* ex_dst = nullptr);

// PCTransformer tail (SconeOcc.py:123-126): per sequence, max over rows then mean over rows:
//   Y[s*ldy + c] = max_r X[(s*L+r)*ldx + c],  Y[s*ldy + E + c] = mean_r X[...]
void launch_pool_max_avg(hipStream_t s, const float* X, int64_t ldx, float* Y, int64_t ldy, int64_t S, int L, int E,
                         const int* lens = nullptr);   // lens: rows of sequence s that take part (padded batches)

// Strided 2-D copy: Y[m*ldy + c] = X[m*ldx + c], c < E
void launch_copy2d(hipStream_t s, const float* X, int64_t ldx, float* Y, int64_t ldy, int64_t M, int E);

// Fused per-query local PCTransformer (local_pct.hip): offs [S,16,3] -> feat[s*ld_feat + 0:256] (max || avg).
// `blob` is the host-packed parameter image of one local transformer (local_pct_blob_floats() floats).
void launch_local_pct(hipStream_t s, const float* offs, float* feat, int64_t ld_feat, int64_t S, const float* blob);
void launch_local_pct5(hipStream_t s, const float* offs, float* feat, int64_t ld_feat, int64_t S, const float* blob);   // bf16 hi/mid/lo blob
// fp16 hi/lo blob; feat_h / feat_l (optional): write the features as fp16 hi/lo planes (row stride ld_feat halves) instead of fp32
void launch_local_pct6(hipStream_t s, const float* offs, float* feat, int64_t ld_feat, int64_t S, const float* blob,
                       void* feat_h = nullptr, void* feat_l = nullptr);
// variant 7 (opt-in 16-bit matrix path): ONE fp16 plane per operand (local_pct7.hip); feat_h (optional): the features as one fp16 plane
void launch_local_pct7(hipStream_t s, const float* offs, float* feat, int64_t ld_feat, int64_t S, const float* blob, void* feat_h = nullptr);

// Head GEMM on operands that already are fp16 hi/lo planes in HBM (linear3p.hip): Y fp32 or Yh / Yl planes = act(X W^T 2^-e + bias ...)
bool linear3p_applicable(int N, int K, int64_t ldx, int64_t ldw, int64_t ldy);
void launch_linear3p(hipStream_t s, const void* Xh, const void* Xl, int64_t ldx, const void* Wh, const void* Wl, int64_t ldw,
                     const float* bias, float* Y, void* Yh, void* Yl, int64_t ldy, int64_t M, int N, int K, int act, float wscale_inv,
                     const float* row_bias, int64_t rows_per_group, const int* row_group, const float* R = nullptr, int64_t ldr = 0,
                     int n_planes = 2);                  // n_planes = 1: the high planes alone, one MFMA per product (variant 7); Xl / Wl / Yl unused
// LayerNorm whose output leaves as fp16 hi/lo planes (row stride ldp halves): the input of a planes GEMM, split where it is produced
void launch_layernorm_planes(hipStream_t s, const float* X, int64_t ldx, const float* g, const float* b, void* Yh, void* Yl, int64_t ldp,
                             int64_t M, int E);
// (Np > N: columns N .. Np - 1 of the planes are written as zeros -- a padded K for the consuming GEMM)
void launch_linear_smallk_planes(hipStream_t s, const float* X, int64_t ldx, const float* W, const float* bias, void* Yh, void* Yl,
                                 int64_t ldy, int64_t M, int N, int K, int act, int Np = 0);
bool linear3p_dot_applicable(int N, int K, int64_t ldx, int64_t ldw);
void launch_linear3p_dot(hipStream_t s, const void* Xh, const void* Xl, int64_t ldx, const void* Wh, const void* Wl, int64_t ldw,
                         const float* bias, int64_t M, int K, int act, float wscale_inv, const float* v, const float* c, int act2, float* out,
                         int n_planes = 2);
void launch_split_to_planes(hipStream_t s, const float* X, int64_t ldx, void* Ph, void* Pl, int64_t ldp, int64_t M, int E);
void launch_split_weights(hipStream_t s, const float* W, int64_t ldw, void* planes, int N, int K);   // linear3h.hip: [2][N][K] fp16 of W * 2^8
// the same with zero padding to [2][Np][Kp] (Np % 4 == 0, Kp % 32 == 0) and the bias padded to bias_p [Np]
void launch_pad_weights(hipStream_t s, const float* W, int64_t ldw, const float* bias, void* planes, float* bias_p, int N, int K, int Np, int Kp);
// segmented kNN-16 with query offsets (knn.hip): see launch_knn16_segmented there
void launch_knn16_segmented(hipStream_t s, const float* X, const float* pc, const long long* pc_off, const int* blocks,
                            int64_t n_blocks, int64_t T, float* offsets_out, float* split_ws = nullptr, int slice = 0);
size_t knn16_segmented_split_floats(int64_t T);
int knn_rows_per_block();
// grid-pruned exact kNN-16 (knn.hip: K1-grid): query order once per query set, one sorted copy per candidate cloud
struct KnnGridCloud { const void* cand; const void* boxes; const void* hdr; int64_t cand_stride, box_stride; };
bool knn_grid_applicable(int64_t M, int k);
size_t knn_grid_query_bytes(int64_t B, int64_t Q);
size_t knn_grid_cloud_bytes(int64_t B, int64_t M);
size_t knn_grid_park_bytes();
const int* knn_grid_order_queries(hipStream_t s, const float* X, int64_t B, int64_t Q, void* ws, void* park_ws, bool launch = true);
KnnGridCloud knn_grid_build_cloud(hipStream_t s, const float* pc, int64_t B, int64_t M, void* ws);
void knn_grid_build_clouds(hipStream_t s, int n, const float* const* pc, const int64_t* M, int64_t B, void* const* ws, KnnGridCloud* out);
void launch_knn16_grid(hipStream_t s, const float* X, const float* pc, int64_t M, const int* qperm, const KnnGridCloud& c, int64_t b_first,
                       int64_t n_b, int64_t Q, int64_t* idx, float* dist, float* pts, bool offsets, void* park_ws, int launch);
int local_pct_blob_floats();
int local_pct3_blob_floats();
int local_pct6_blob_floats();
int local_pct7_blob_floats();

}  // namespace mcr
