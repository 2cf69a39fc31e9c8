/* macarons_hip.h — C ABI of libmacarons_hip.so: the MI355X (gfx950) kernels of the SCONE
 * coverage-gain hot path of MACARONS.
 *
 * The reference (Anttwo/MACARONS) is pure Python/PyTorch: it has no FFI.  Its boundary for this path
 * is the Python class surface of macarons/networks/{SconeVis,SconeOcc,Macarons}.py, which
 * macarons_amd/networks mirrors.  THIS header is the layer below it: plain pointers and sizes, no
 * torch types.  Every entry point cites the reference op sequence (file:line, relative to the
 * upstream tree) it replaces.
 *
 * Conventions
 *   - all pointers are DEVICE pointers (HIP, gfx950) to contiguous row-major fp32 unless stated;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls are asynchronous;
 *   - return 0 on success; non-zero on error, message via mcr_last_error() (thread-local);
 *   - no entry point allocates device memory: scratch is passed in (`*_workspace_bytes` gives size).
 */
#ifndef MACARONS_HIP_H
#define MACARONS_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

const char* mcr_last_error(void);
int mcr_abi_version(void);
const char* mcr_target_arch(void);

/* ---- K9: SH coverage-gain scorer -----------------------------------------------------------------
 * Replaces SconeVis.compute_coverage_gain (macarons/networks/SconeVis.py:210-252):
 *   gains[b,c] = mean_n act( sum_k Y_k(dir(cams[b,c] - pts[b,n,:3])) * harmonics[b,n,k] )
 * with the real SH basis of macarons/utility/spherical_harmonics.py:111-157 in the angle convention
 * of macarons/utility/CustomGeometry.py:27-45; act = sigmoid (use_sigmoid) or relu (SconeVis.py:242-245).
 *   pts [B,N,pts_dim] (pts_dim >= 3, only xyz read — SconeVis.py:224), harmonics [B,N,64],
 *   cams [B,C,3], gains [B,C].  waves_per_simd: grid sizing override, 0 = auto (whole grid co-resident).
 * Deterministic (two-pass reduce, no float atomics). */
size_t mcr_sh_coverage_gain_workspace_bytes(int64_t B, int64_t N, int64_t C);
int mcr_sh_coverage_gain(const float* pts, int pts_dim, const float* harmonics, const float* cams, float* gains,
                         int64_t B, int64_t N, int64_t C, int use_sigmoid, int waves_per_simd, void* workspace,
                         size_t workspace_bytes, void* stream);
/* First stage of mcr_sh_coverage_gain alone (sh_gain_kernel: per-(wave tile, camera) partial sums left in `workspace`), so that
 * the dominant kernel can be timed by itself (bench.py roofline); same arguments minus `gains`. */
int mcr_sh_coverage_gain_partials(const float* pts, int pts_dim, const float* harmonics, const float* cams, int64_t B, int64_t N,
                                  int64_t C, int use_sigmoid, int waves_per_simd, void* workspace, size_t workspace_bytes,
                                  void* stream);

/* mcr_sh_coverage_gain + the decision behind it (testers/shapenet.py:172: torch.max over the cameras) in one call: record [B,2] =
 * (max_c gains[b,c], first arg-max camera as fp32; a NaN gain wins like torch.max; C < 2^24). */
int mcr_sh_coverage_gain_best(const float* pts, int pts_dim, const float* harmonics, const float* cams, float* gains, float* record,
                              int64_t B, int64_t N, int64_t C, int use_sigmoid, int waves_per_simd, void* workspace,
                              size_t workspace_bytes, void* stream);

/* Replaces SconeVis.compute_visibilities (SconeVis.py:164-208) == Macarons.compute_visibility_gains
 * (macarons/networks/Macarons.py:138-178): the same without the mean.  vis [B,C,N]. */
int mcr_sh_visibilities(const float* pts, int pts_dim, const float* harmonics, const float* cams, float* vis,
                        int64_t B, int64_t N, int64_t C, int use_sigmoid, void* stream);

/* ---- K1: k nearest surface points per query ---------------------------------------------------------
 * Replaces get_knn_points (macarons/utility/utils.py:1497-1509: torch.cdist + topk(largest=False) +
 * pytorch3d.ops.knn_gather) and, with subtract_query != 0, the offset step of SconeOcc.forward
 * (macarons/networks/SconeOcc.py:297-298).
 *   X [B,Q,3] queries, pc [B,M,3] surface points ->
 *   idx [B,Q,k] int64 (ascending distance; ties -> lower index), dists [B,Q,k], pts [B,Q,k,3]
 *   (neighbour coordinates, minus the query if subtract_query).  k in {1,4,8,16}, k <= M.  idx and / or dists may be NULL
 *   (not written: SconeOcc only consumes the offsets). */
int mcr_knn_points(const float* X, const float* pc, int64_t* idx, float* dists, float* pts, int64_t B, int64_t Q,
                   int64_t M, int k, int subtract_query, void* stream);

/* The same search with a scratch buffer (mcr_knn_grid_workspace_bytes(B, Q, M) bytes, device): for k = 16 and 1024 <= M <= 16384 the
 * candidates are counting-sorted into a 16^3 Morton grid, the queries into a 32^3 one, and a wave visits only the 32-candidate
 * sub-tiles whose bounding box can still hold one of its queries' 16 nearest points -- outputs IDENTICAL to mcr_knn_points (same
 * exact fp32 distances, same (distance, index) order); any other shape runs mcr_knn_points itself.  Nothing upstream corresponds to
 * the pruning: torch.cdist + topk (utils.py:1497-1509) is brute force. */
size_t mcr_knn_grid_workspace_bytes(int64_t B, int64_t Q, int64_t M);
int mcr_knn_points_grid(const float* X, const float* pc, int64_t* idx, float* dists, float* pts, int64_t B, int64_t Q, int64_t M,
                        int k, int subtract_query, void* workspace, size_t workspace_bytes, void* stream);

/* The k = 16 search of J independent jobs in one launch (the ragged occupancy pass, macarons_utils.py:1395-1540: every (cell, chunk)
 * job runs utils.py:1497-1509 + SconeOcc.py:297-298 on its own cloud): X [T,3] query rows sorted by job; pc = the jobs' clouds
 * back to back, job j = rows pc_off[j] .. pc_off[j+1] (device int64 [J+1], every job >= 16 points); blocks (device int32
 * [n_blocks,4]) = (job, first query row, rows <= mcr_knn_rows_per_block(), 0), one workgroup each; offsets_out [T,16,3] = neighbour
 * minus query, neighbours in mcr_knn_points' order.  workspace (optional, mcr_knn_offsets_segmented_workspace_bytes(T) bytes): lets a
 * launch with few blocks split every job's candidates over several workgroups (same result). */
size_t mcr_knn_offsets_segmented_workspace_bytes(int64_t T);
int mcr_knn_offsets_segmented(const float* X, const float* pc, const int64_t* pc_off, const int* blocks, int64_t n_blocks, int64_t T,
                              float* offsets_out, void* workspace, size_t workspace_bytes, void* stream);

/* ---- K4/K5 building blocks (macarons/networks/Attention.py) -------------------------------------------
 * mcr_linear: nn.Linear (+ optional exact-erf GELU, + optional residual add):
 *   Y[m*ldy+n] = act(sum_k X[m*ldx+k] * W[n*K+k] + bias[n]) + residual[m*ldr+n]     (Attention.py:98-103,186-188,232-235)
 * mcr_layernorm: nn.LayerNorm(E), eps 1e-5 (Attention.py:274,292).
 * mcr_attention: attention() of Attention.py:8-36 with the head split of :174-198 on a packed row
 *   [q (qk_dim) | k (qk_dim) | v (v_dim)], head h owning channels [h*d,(h+1)*d); mask = None; S sequences of
 *   L consecutive rows; scores are divided by sqrt(qk_dim / n_heads).  Supported: 4 heads, (32,128) | (64,256).
 * mcr_colmax_broadcast: Embedding's cloud-wide max feature (Attention.py:117-121).
 * mcr_pool_max_avg: PCTransformer's max || avg pooling over the sequence (SconeOcc.py:123-126). */
int mcr_linear(const float* X, int64_t ldx, const float* W, const float* bias, const float* residual, int64_t ldr, float* Y,
               int64_t ldy, int64_t M, int N, int K, int gelu, void* stream);
int mcr_layernorm(const float* X, int64_t ldx, const float* gamma, const float* beta, float* Y, int64_t ldy, int64_t M, int E,
                  void* stream);
int mcr_attention(const float* qkv, int64_t ldq, float* out, int64_t ldo, int64_t S, int64_t L, int n_heads, int qk_dim,
                  int v_dim, void* stream);
/* Same with a scratch buffer of mcr_attention_workspace_bytes(S, L, n_heads, v_dim) bytes: one or two long sequences (too few
 * blocks to fill the chip) split their keys over two blocks and merge the partial soft-maxes (results differ from mcr_attention
 * by summation order only, ~1e-6). */
size_t mcr_attention_workspace_bytes(int64_t S, int64_t L, int n_heads, int v_dim);
int mcr_attention_ws(const float* qkv, int64_t ldq, float* out, int64_t ldo, int64_t S, int64_t L, int n_heads, int qk_dim,
                     int v_dim, void* workspace, size_t workspace_bytes, void* stream);
/* attention(q, k, v, mask) of Attention.py:8-36 WITH a mask (:24-27): where the mask byte of (sequence s, head h, query q, key k) --
 * mask[s * mask_seq_stride + h * mask_head_stride + q * mask_query_stride + k] -- is 0 the score is replaced by -1e3 BEFORE the
 * division by sqrt(d) (upstream's masked_fill rule: a fully masked query attends uniformly over the keys; it is not -inf).  Strides
 * of 0 broadcast: a [S, L] key mask has query and head stride 0, upstream's [S, 1, L, L] has head stride 0.  workspace: as
 * mcr_attention_ws (may be NULL).  fp32 P V (the fp16-split P V of the unmasked path is not taken). */
int mcr_attention_masked(const float* qkv, int64_t ldq, float* out, int64_t ldo, int64_t S, int64_t L, int n_heads, int qk_dim,
                         int v_dim, const unsigned char* mask, int64_t mask_seq_stride, int64_t mask_head_stride,
                         int64_t mask_query_stride, void* workspace, size_t workspace_bytes, void* stream);
/* The long-sequence attention of the encoders on the fp16-split matrix path (variant 6, sequences of >= 512 tokens), as one call: the
 * packed rows are split ONCE into fp16 hi/lo planes (inside the networks the QKV projection's epilogue writes them), K / V tiles then
 * reach LDS by DMA and both products run on fp16 pairs (attention_planes.hip).  Needs |q|, |k|, |v| < 65504 (an inf / NaN output tells).
 * lens (optional, device int per sequence): keys = the first min(L, lens[s]) rows.  split_mode: 1 = keys of every sequence over two
 * blocks (L >= 512), 0 = never, -1 = only when the unsplit grid leaves CUs idle.  workspace: mcr_attention_planes_workspace_bytes. */
size_t mcr_attention_planes_workspace_bytes(int64_t S, int64_t L, int n_heads, int qk_dim, int v_dim);
int mcr_attention_planes(const float* qkv, int64_t ldq, float* out, int64_t ldo, int64_t S, int64_t L, int n_heads, int qk_dim,
                         int v_dim, const int* lens, int split_mode, void* workspace, size_t workspace_bytes, void* stream);
int mcr_colmax_broadcast(const float* X, int64_t ldx, float* Y, int64_t ldy, int64_t S, int64_t L, int E, void* stream);
int mcr_pool_max_avg(const float* X, int64_t ldx, float* Y, int64_t ldy, int64_t S, int64_t L, int E, void* stream);

/* ---- network forwards -----------------------------------------------------------------------------------
 * Weight tables are arrays of device pointers to contiguous fp32 tensors in nn.Module layout ([out,in] weights):
 *   ENCODER (12): norm1.weight, norm1.bias, qkv.weight, qkv.bias, out.weight, out.bias, norm2.weight, norm2.bias,
 *                 ff.linear1.weight, ff.linear1.bias, ff.linear2.weight, ff.linear2.bias
 *                 where qkv = rows of mhsa.w_q, mhsa.w_k, mhsa.w_v stacked ([2*qk_dim + E, E]) and likewise the bias.
 *   PCT (32):     embedding.linear1.{weight,bias}, embedding.linear2.{weight,bias}, ENCODER x2, norm.{weight,bias},
 *                 linear0.{weight,bias}
 *   SCONE_VIS (48): embedding.linear1.{w,b}, embedding.linear2.{w,b}, ENCODER x3, norm.{w,b}, fc1.{w,b}, fc2.{w,b}, fc3.{w,b}
 *   SCONE_OCC (140): PCT global_transformer, PCT local_transformers.{0,1,2}, x_embedding.linear{1,2,3}.{w,b},
 *                 linear1.{w,b}, linear2.{w,b}, linear3.{w,b}
 *   PLANES (optional tail; n_weights = 32 + 8 / 48 + 12 / 140 + 8): per ENCODER of the long-sequence networks (PCT's two, SCONE_VIS's
 *                 three, SCONE_OCC's global_transformer's two) four more pointers -- qkv.weight, out.weight, ff.linear1.weight,
 *                 ff.linear2.weight as fp16 hi/lo planes [2][N][K] of W * 2^8 (hi = fp16(x), lo = fp16(x - hi);
 *                 networks/packing.py: encoder_weight_planes) -- for the planes GEMMs of variant 6 (sequences of >= 512 tokens).
 *                 Without the tail the weights are split by one small launch per GEMM and call.
 *   END PLANES (optional, behind PLANES; n_weights = 32 + 11 / 48 + 17 / 140 + 11): the layers either side of the encoders on the same
 *                 path -- PCT / SCONE_OCC's global_transformer: embedding.linear2.weight zero-padded to planes [2][128][128],
 *                 its bias zero-padded to fp32 [128], linear0.weight planes; SCONE_VIS: the same two for its embedding.linear2
 *                 (126 x 126 -> 128 x 128), then fc1, fc2, fc3 planes (networks/packing.py: padded_weight_planes, weight_planes).
 *
 * mcr_pc_transformer_forward: PCTransformer.forward (macarons/networks/SconeOcc.py:104-130); pc [S,L,3] ->
 *   features [S, feature_dim] (max || avg), feature_dim in {256, 512}.
 * mcr_scone_vis_forward: SconeVis.forward (macarons/networks/SconeVis.py:121-162); pts [B,N,4], view_harmonics
 *   [B,N,64] -> out [B,N,64].  Default architecture only (pts_embedding_dim 256, 4 heads, 3 encoders,
 *   view_state_mode "end", global feature, concatenated input).  lengths (optional, DEVICE int32 [B], may be NULL): cloud b
 *   consists of its first min(N, lengths[b]) rows only -- the cloud-wide max of the embedding and the attention keys stop
 *   there; rows beyond still produce (meaningless) outputs.  This is how the padded output of mcr_sample_proxy is consumed
 *   without its count ever reaching the host (the reference slices with the count on the host, testers/shapenet.py:146-157).
 * mcr_scone_occ_forward: SconeOcc.forward (macarons/networks/SconeOcc.py:250-347) given the clouds the reference
 *   would obtain from its torch.randperm draws (:269, :311), which the HOST performs so the RNG stream matches:
 *   pc_global [B,Lg,3]; pc_scale[i] [B,M_scale[i],3] for the 3 neighbourhood scales; x [B,Q,3];
 *   view_harmonics [B,Q,64] -> out [B,Q,1].  pc_scale / M_scale are HOST arrays of 3 entries.
 *   head_planes / head_inv_scales (optional, HOST arrays of 4; variant 6 only): the fp16 hi/lo planes of the four large head
 *   matrices -- x_embedding.linear2 [256,128], x_embedding.linear3 [512,256], linear1[:, 512:1856] [512,1344], linear2
 *   [256,512] -- each times a power of two 2^e, laid out [plane][n][K/8][8 fp16], and 2^-e (networks/packing.py:
 *   pack_head_planes); NULL: the kernel splits the weights itself on every call (fixed 2^8 scale: needs |w| < 255).
 *   range_flag (optional, DEVICE int, cleared by the caller): set to 1 when an occupancy comes out non-finite -- on the fp16
 *   split path (variant 6, |activation| < 65504) that is what an out-of-range activation turns into; the host mirror then
 *   re-runs the call on variant 5 (whole fp32 range). */
size_t mcr_pc_transformer_workspace_bytes(int64_t S, int64_t L);
int mcr_pc_transformer_forward(const float* pc, float* features, int64_t S, int64_t L, int feature_dim,
                               const float* const* weights, int n_weights, void* workspace, size_t workspace_bytes,
                               void* stream);
size_t mcr_scone_vis_workspace_bytes(int64_t B, int64_t N);
int mcr_scone_vis_forward(const float* pts, const float* view_harmonics, float* out, int64_t B, int64_t N,
                          const float* const* weights, int n_weights, const int* lengths, void* workspace,
                          size_t workspace_bytes, void* stream);
size_t mcr_scone_occ_workspace_bytes(int64_t B, int64_t Q, int64_t Lg);
int mcr_scone_occ_forward(const float* pc_global, int64_t Lg, const float* const* pc_scale, const int64_t* M_scale,
                          const float* x, const float* view_harmonics, float* out, int64_t B, int64_t Q,
                          const float* const* weights, int n_weights, const float* const* local_blobs,
                          const void* const* head_planes, const float* head_inv_scales, int* range_flag, void* workspace,
                          size_t workspace_bytes, void* stream);

/* mcr_scone_occ_forward in two calls on one stream and one workspace (phase 0 = the single call).
 *   phase 1: what needs neither view_harmonics nor a hidden draw -- the query order of the grid search and scale 0 (the whole
 *            cloud: k-NN + local transformer).  Reads x, pc_scale[0], M_scale[0..2]; pc_global, pc_scale[1..2], view_harmonics and
 *            out may be NULL.  It is the first long kernel of an NBV step (testers/shapenet.py:126-144): queued before the host builds
 *            the view state, the harmonics and the down-sampled clouds, it hides their launch latency.
 *   phase 2: the rest (global transformer, scales 1 and 2, x embedding, head) -> out; same arguments as the single call. */
int mcr_scone_occ_forward_phase(const float* pc_global, int64_t Lg, const float* const* pc_scale, const int64_t* M_scale,
                                const float* x, const float* view_harmonics, float* out, int64_t B, int64_t Q,
                                const float* const* weights, int n_weights, const float* const* local_blobs,
                                const void* const* head_planes, const float* head_inv_scales, int* range_flag, void* workspace,
                                size_t workspace_bytes, int phase, void* stream);

/* Ragged SconeOcc: J independent SconeOcc.forward calls ("jobs" = one surface cloud + one chunk of queries, all of different
 * sizes: the per-cell passes of compute_scene_occupancy_probability_field, macarons/utility/macarons_utils.py:1395-1540, which
 * upstream runs one after the other from a Python loop over the grid cells) in ONE launch sequence.
 *   pc_global [J,Lg,3]: job j's global down-sample (its first global_len[j] rows; the rest padding), global_len DEVICE int[J];
 *   pc_scale[s] (HOST array of 3 device pointers): the neighbourhood clouds of scale s of all jobs back to back, job j =
 *     rows [scale_off[s][j], scale_off[s][j+1]) (scale_off: HOST array of 3 DEVICE int64[J+1]); every job needs >= 16 points per scale;
 *   x [T,3], view_harmonics [T,64]: the queries of all jobs back to back, sorted by job; row_job DEVICE int[T];
 *   knn_blocks DEVICE int[n_blocks*4] = (job, first row, rows <= mcr_knn_rows_per_block(), 0) per workgroup of the segmented
 *     kNN, covering every row exactly once;  out [T].  Needs the fused local-transformer blobs; other arguments as above. */
int mcr_knn_rows_per_block(void);
size_t mcr_scone_occ_ragged_workspace_bytes(int64_t J, int64_t T, int64_t Lg);
int mcr_scone_occ_forward_ragged(const float* pc_global, const int* global_len, int64_t Lg, const float* const* pc_scale,
                                 const int64_t* const* scale_off, const float* x, const float* view_harmonics, const int* row_job,
                                 const int* knn_blocks, int64_t n_blocks, float* out, int64_t J, int64_t T,
                                 const float* const* weights, int n_weights, const float* const* local_blobs,
                                 const void* const* head_planes, const float* head_inv_scales, int* range_flag, void* workspace,
                                 size_t workspace_bytes, void* stream);

/* The same as two calls on one stream with ONE workspace: phase 1 = what needs none of the hidden torch.randperm draws (the scale-0
 * search and local transformer over the whole clouds; the x embedding on the fp16-planes path) -- it reads pc_scale[0], scale_off[0],
 * x, view_harmonics, row_job, knn_blocks, local_blobs[0] only (the other arrays may hold anything, pc_global / global_len / out may be
 * NULL); phase 2 = the rest.  The host makes the draws between the two calls while the GPU works on phase 1 (a MACARONS decision
 * otherwise idles ~2.5 ms there).  phase 0 = both = mcr_scone_occ_forward_ragged. */
int mcr_scone_occ_forward_ragged_phase(const float* pc_global, const int* global_len, int64_t Lg, const float* const* pc_scale,
                                       const int64_t* const* scale_off, const float* x, const float* view_harmonics, const int* row_job,
                                       const int* knn_blocks, int64_t n_blocks, float* out, int64_t J, int64_t T,
                                       const float* const* weights, int n_weights, const float* const* local_blobs,
                                       const void* const* head_planes, const float* head_inv_scales, int* range_flag, void* workspace,
                                       size_t workspace_bytes, int phase, void* stream);

/* Fused per-query local PCTransformer (the FLOP majority of SconeOcc.forward, SconeOcc.py:293-304 + :104-130):
 *   offsets [S,16,3] (kNN neighbours minus the query) -> features[s*ld_features + 0:256] = max(128) || avg(128).
 * `blob`: 16-byte-aligned device image of ONE local transformer's parameters, mcr_local_pct_blob_floats() floats,
 * laid out as macarons_amd/networks/packing.py builds it (MFMA-fragment order, LayerNorm folded into the
 * following linear layer).  mcr_scone_occ_forward uses the fused kernel for scale i when local_blobs != NULL and
 * local_blobs[i] != NULL (HOST array of 3 device pointers), else the layer-by-layer path. */
int mcr_local_pct_blob_floats(void);
int mcr_local_pct3_blob_floats(void);
int mcr_local_pct6_blob_floats(void);
int mcr_local_pct7_blob_floats(void);
/* Kernel variant behind mcr_local_pct_forward / the fused path of mcr_scone_occ_forward.  The blob must have been packed
 * for the selected variant (macarons_amd/networks/packing.py):
 *   1: exact-fp32 MFMA, one workgroup/CU (local_pct.hip; mcr_local_pct_blob_floats() floats);
 *   5: split precision, every fp32 operand as exact bf16 hi/mid/lo, six MFMAs per product, two workgroups per CU, valid
 *      for the whole fp32 range (local_pct5.hip; mcr_local_pct3_blob_floats() floats);
 *   6 (default): two-term fp16 split (hi + lo, 22 significant bits), three MFMAs per product, same structure
 *      (local_pct6.hip; mcr_local_pct6_blob_floats() floats); needs |activation| < 65504;
 *   7 (OPT-IN, per call only: mcr_call_variant(7); mcr_set_local_pct_variant refuses it): BASELINE.json config 3's 16-bit matrix
 *      path -- ONE fp16 plane per matrix operand, one MFMA per product, fp32 accumulation; LayerNorm statistics, soft-max, GELU,
 *      pooling, SH math and every reduction stay fp32 (local_pct7.hip; mcr_local_pct7_blob_floats() floats; the SconeOcc head's
 *      planes GEMMs read the high planes alone).  NOT within the 1e-4 of variants 1 / 5 / 6: occupancies within 2e-3 relative
 *      (measured, tests/test_variant7_gpu.py); same range rule as variant 6. */
/* The variant is a property of a CALL: mcr_call_variant(v) is the per-call argument -- the NEXT network entry point (mcr_linear,
 * mcr_attention*, mcr_local_pct_forward, mcr_pc_transformer_forward, mcr_scone_vis_forward, mcr_scone_occ_forward*) called on THIS
 * thread runs on variant v (one-shot, thread-local; 0 = the default again).  mcr_set_local_pct_variant only sets the process DEFAULT
 * (start-up value: env MCR_LOCAL_PCT_VARIANT, else 6) that calls without a one-shot take; no call ever changes it. */
int mcr_set_local_pct_variant(int variant);
int mcr_get_local_pct_variant(void);
int mcr_call_variant(int variant);
/* Range guard of variant 6 for outputs that leave through an entry point without a flag argument (mcr_scone_vis_forward's harmonics):
 * *flag |= 1 if any of x[0 .. n) is inf / NaN -- what an activation beyond the fp16 range of the split matrix path turns into.  On
 * variant 6 the encoders of sequences of >= 512 tokens (SconeVis, SconeOcc's global transformer; Attention.py:278-300) run their
 * GEMMs on fp16 hi/lo planes (LayerNorm / GELU epilogues write the planes, operands reach LDS by DMA; env MCR_ENC_PLANES=0: the
 * full-range bf16 x 6 kernels of variant 5). */
int mcr_nonfinite_flag(const float* x, int64_t n, int* flag, void* stream);
int mcr_local_pct_forward(const float* offsets, float* features, int64_t ld_features, int64_t S, const float* blob,
                          void* stream);

/* ---- glue either side of the networks --------------------------------------------------------
 * mcr_view_state: compute_view_state (macarons/utility/scone_utils.py:799-860): for every (point, view) the
 *   elevation/azimuth of (X_view[v] - pts[p,:3]) is binned on the n_elev x n_azim lattice with the reference's
 *   exact clamp / wrap rules; view_state [n_points, n_elev*n_azim] fp32 0/1 (zeroed here, then set).
 * mcr_sample_proxy: sample_proxy_points (scone_utils.py:1030-1061) with the uniforms `u` supplied by the host:
 *   keeps points with preds > min_occ, draws n_sample points with probability proportional to preds (inverse
 *   CDF in fp64: first i with C_i >= u * C_last), then unique (sorted) + inverse.  Outputs are padded to
 *   n_sample rows (rows >= *n_unique are zero-filled); *n_unique (device int) says how many are valid.  uniq holds
 *   ORIGINAL point indices.  view_harmonics may be NULL (then res_harmonics is not written: a caller that only holds a
 *   shard of the harmonics recomputes the rows of the sampled points from uniq).  Three launches: block sums + scan of them by the last block to finish, one wave per sample
 *   for the search, one block for sort / unique / inverse / gather.
 *   res [n_sample,4] = (X, pred), res_harmonics [n_sample,64].  preds may be strided (pred_stride floats).
 *   volume (optional device double): sum of the kept occupancies (fov_proxy_volume of macarons_utils.py:1620).
 * mcr_points_in_fov: Camera.get_points_in_fov (macarons/utility/macarons_utils.py:2400-2435) for n_cam cameras
 *   at once.  Each camera is 40 floats: M_view[16] and M_proj[16] (row-major 4x4, row-vector convention
 *   p' = [x y z 1] M, as pytorch3d's get_world_to_view_transform / get_full_projection_transform matrices),
 *   {min_ndc_x, max_ndc_x, min_ndc_y, max_ndc_y}, camera centre[3], range (<= 0: no range test).
 *   mask [n_cam, P] bytes (0/1).
 * mcr_coverage_gain_multiple: the tuple stage of SconeVis.compute_coverage_gain_multiple (SconeVis.py:289-301):
 *   vis [B,C,N] -> gains [B, C^n_cam] over ordered tuples in torch.cartesian_prod order, n_cam in {2,3}. */
int mcr_view_state(const float* pts, int pts_dim, const float* X_view, float* view_state, int64_t n_points, int n_view,
                   int n_elev, int n_azim, void* stream);
/* Batched / in-place form.  n_clouds clouds of pts_per_cloud points each; cloud c is binned against ITS OWN view positions
 * X_view[c] ([n_clouds, n_view, 3]: every object of a scene batch has its own trajectory, testers/shapenet.py:33-37).
 * accumulate != 0: view_state is not cleared first -- bins are OR'ed into the caller's 0/1 table, which is what the reference's
 * `view_states[mask] += compute_view_state(...)` followed by torch.heaviside(., 0) does (macarons_utils.py:2867-2877); with
 * rows != NULL (device int64 [n_clouds*pts_per_cloud], needs accumulate) point p updates row rows[p] of the table, i.e. the
 * masked in-place update of the scene-wide state table in one launch. */
int mcr_view_state_batched(const float* pts, int pts_dim, const float* X_view, float* view_state, int64_t n_clouds,
                           int64_t pts_per_cloud, int n_view, int n_elev, int n_azim, const int64_t* rows, int accumulate,
                           void* stream);
size_t mcr_sample_proxy_workspace_bytes(int64_t P, int n_sample);
int mcr_sample_proxy(const float* X, const float* preds, int64_t pred_stride, const float* view_harmonics, int64_t P,
                     float min_occ, const float* u, int n_sample, float* res, float* res_harmonics, int64_t* uniq,
                     int64_t* inverse, int* n_unique, double* volume, void* workspace, size_t workspace_bytes, void* stream);
/* The same for B independent clouds in one launch sequence (a scene batch: every tensor gains a leading B; preds is
 * [B, P] with element stride pred_stride; u [B, n_sample]; n_unique int[B]; volume double[B] or NULL): every cloud draws from its
 * own CDF with its own uniforms, exactly as B calls of mcr_sample_proxy would. */
size_t mcr_sample_proxy_batched_workspace_bytes(int64_t B, int64_t P, int n_sample);
int mcr_sample_proxy_batched(const float* X, const float* preds, int64_t pred_stride, const float* view_harmonics, int64_t B,
                             int64_t P, float min_occ, const float* u, int n_sample, float* res, float* res_harmonics,
                             int64_t* uniq, int64_t* inverse, int* n_unique, double* volume, void* workspace,
                             size_t workspace_bytes, void* stream);
/* B distributions over ONE shared point set: X [P,3] and view_harmonics [P,64] are common, only preds [B,P] and u [B,n_sample]
 * differ -- the K neighbour cameras of a MACARONS decision, each sampling the scene's proxy points inside its own frustum
 * (preds = mcr_fov_mask_occ rows; macarons_utils.py:1603-1624).  Workspace as for the batched form. */
int mcr_sample_proxy_shared(const float* X, const float* preds, int64_t pred_stride, const float* view_harmonics, int64_t B,
                            int64_t P, float min_occ, const float* u, int n_sample, float* res, float* res_harmonics,
                            int64_t* uniq, int64_t* inverse, int* n_unique, double* volume, void* workspace,
                            size_t workspace_bytes, void* stream);
int mcr_points_in_fov(const float* pts, int64_t P, const float* cameras, int n_cam, unsigned char* mask, void* stream);
/* filter_proxy_points (macarons/utility/scone_utils.py:1001-1027; call site testers/shapenet.py:122): mask[p] = 1 iff in every
 * view v the projection ([x y z 1] * proj[v])[:2] / w of X[p] lies strictly inside the bounding box of the projected surface
 * cloud pc grown by filter_tol.  proj = n_view row-major 4x4 full-projection matrices (row-vector convention, as
 * pytorch3d's get_full_projection_transform().get_matrix()); bounds = n_view*4 floats of scratch, left holding
 * {min_x, max_x, min_y, max_y} per view. */
/* Column gather of move_view_state_to_view_space (macarons/utility/scone_utils.py:863-931, line :928; callers
 * macarons_utils.py:1356,1488): out[r, v] = in[r, idx[v]]; idx = the V bin indices the host derives from the camera rotation
 * (device int32, every entry in [0, V)). */
int mcr_gather_columns(const float* in, const int* idx, float* out, int64_t rows, int V, void* stream);
int mcr_filter_proxy_points(const float* X, int64_t P, const float* pc, int64_t M, const float* proj, int n_view, float filter_tol,
                            float* bounds, unsigned char* mask, void* stream);
int mcr_coverage_gain_multiple(const float* vis, float* gains, int64_t B, int64_t C, int64_t N, int n_cam, void* stream);

/* The end of a single-rank NBV decision in one launch: gains [B,C] (in/out) -> max_gain [B], nbv_idx [B] (int64), torch.max over
 * the cameras (testers/shapenet.py:172: first maximum, NaN wins); a cloud with n_unique[b] < 1 (nothing sampled: the reference
 * fails there, scone_utils.py:1052-1061) gets NaN gains, a NaN maximum and index -1.  n_unique / range_flag may be NULL.
 * record: 1 + 2 B doubles = (*range_flag or 0, nbv_idx[0..B), max_gain[0..B)) -- the one buffer a caller reads back. */
int mcr_nbv_decide(float* gains, int64_t B, int64_t C, const int* n_unique, const int* range_flag, float* max_gain, int64_t* nbv_idx,
                   double* record, void* stream);

/* Arg-max exchange of the camera-sharded decision (the reference takes torch.max over all cameras on one GPU,
 * macarons/testers/shapenet.py:172; ties -> first = lowest camera index):
 * mcr_best_record: records[b] = (max_c gains[b,c], idx_offset + argmax) as two fp32 (indices < 2^24 are exact) -- the 8-byte
 *   per-cloud record each rank contributes to one all-gather;
 * mcr_best_merge: records [world,B,2] -> vals[b], idx[b] (int64) of the global maximum. */
int mcr_best_record(const float* gains, int64_t B, int64_t C, int64_t idx_offset, float* records, void* stream);
int mcr_best_merge(const float* records, int world, int64_t B, float* vals, int64_t* idx, void* stream);

/* MACARONS per-camera scoring (predict_coverage_gain_for_single_camera, macarons/utility/macarons_utils.py:1580-1738):
 * mcr_fov_mask_occ: occ_out[c,p] = mask[c,p] ? occ[p] : 0  (frustum mask AND'ed into the sampler's occupancy, :1603-1613)
 * mcr_transform_points: in place pts[i,:3] = (([x y z 1] M_view)[:3] - center) * inv_diag   (:1641-1660)
 * mcr_macarons_gain: vis[b,n] *= factor(d = |pts_world[b,n]-cam_world[b]|), gains[b] = mean_n vis[b,n] * volume[b] (:1699-1704);
 *   factor_mode 0: min(1, (th/d)^2) = get_distance_factor_threshold (:1768-1776) and, with th = focal * epsilon / pixel_size,
 *   get_distance_factor (:1741-1765); factor_mode 1: 1 / (1 + (d/th)^2) = get_distance_factor_smooth (:1779-1788). */
int mcr_fov_mask_occ(const unsigned char* mask, const float* occ, int64_t occ_stride, float* occ_out, int64_t P, int n_cam,
                     void* stream);
int mcr_transform_points(float* pts, int pts_dim, int64_t n, const float* M_view, const float* center, float inv_diag,
                         void* stream);
/* All clouds in one launch: cloud c (pts_per_cloud consecutive rows, or, with cloud_of != NULL, the n_points rows whose
 * cloud_of[i] == c -- ragged clouds) uses M_view[c] (16 floats), center[c] (3 floats) and inv_diag[c]: the K neighbour cameras'
 * sampled sets, or the per-cell clouds of the occupancy-field pass, go to their prediction spaces together. */
int mcr_transform_points_batched(float* pts, int pts_dim, int64_t n_clouds, int64_t pts_per_cloud, const float* M_view,
                                 const float* center, const float* inv_diag, const int* cloud_of, int64_t n_points, void* stream);
int mcr_macarons_gain(float* vis, const float* pts_world, int pts_dim, const float* cam_world, const float* volume,
                      float distance_th, int factor_mode, int64_t B, int64_t N, float* gains, void* stream);

/* ---- scene-side point bookkeeping ---------------------------------------------------------
 * mcr_min_dist_segmented (K11): dmin[i] = min_j |A[i] - B[j]| in fp64 over the B points of A[i]'s segment (grid cell);
 *   replaces torch.min(torch.cdist(a.double(), b.double())) of Cell.fill / camera_coverage_gain / scene_coverage
 *   (macarons/utility/macarons_utils.py:2566, :3022, :3049).  Offsets are CSR int64 [n_segments+1]; +inf if B is empty.
 * mcr_unproject_depth (K12): Camera.project_depth_in_3D / utils.project_depth_back_to_3D (macarons_utils.py:2339-2360,
 *   utils.py:1458-1487): depth [n_cam,H,W] -> world [n_cam,H*W,3]; each camera = 18 floats: inverse full projection
 *   matrix (row-major, row-vector convention) + k22, k32 of the projection matrix (scaled-depth conversion). */
/* mcr_signed_distance_to_depth: Camera.get_signed_distance_to_depth_maps (macarons_utils.py:2451-2500) for one camera / depth
 *   map: sgn[i] = z_view(pts[i]) - bilinear(depth)(projection of pts[i]), grid_sample(bilinear, border, align_corners=False)
 *   semantics; camera = 32 floats M_view[16] | M_full_projection[16] (row-major, row-vector convention); depth [H,W];
 *   mask [H,W] bytes or NULL: pixels with mask == 0 count as `fill` (the reference writes 1.1 zfar there, :2479).
 * mcr_proxy_scene_update: the proxy-point bookkeeping of one MACARONS step after a new depth map (testers/scene.py:402-418:
 *   signed distances, Scene.update_proxy_view_states :2817-2877, update_proxy_supervision_occ :2888-2912,
 *   update_proxy_out_of_field :2879-2886) fused in one pass over the P proxy points; fov_mask [P] bytes selects the points in
 *   the current frustum (mcr_points_in_fov); X_cam = 3 device floats (the camera centre); per-point state tensors are updated
 *   in place: view_states [P, n_elev*n_azim] (0/1; the bin towards the camera is OR'ed in where the signed distance is below
 *   distance_to_surface), n_inside / n_behind / supervision_occ / out_of_field [P].  sgn [P] (optional) receives the signed
 *   distances of the masked points. */
int mcr_signed_distance_to_depth(const float* pts, int64_t n, const float* camera, const float* depth, const unsigned char* mask,
                                 int H, int W, float fill, float* sgn, void* stream);
int mcr_proxy_scene_update(const float* proxy_points, int64_t P, const unsigned char* fov_mask, const float* camera, const float* depth,
                           const unsigned char* depth_mask, int H, int W, float fill, const float* X_cam, float distance_to_surface,
                           float tol, float score_threshold, int n_elev, int n_azim, float* view_states, float* n_inside,
                           float* n_behind, float* supervision_occ, float* out_of_field, float* sgn, void* stream);
int mcr_min_dist_segmented(const float* A, const int64_t* a_offsets, const float* B, const int64_t* b_offsets, int64_t n_segments,
                           int64_t max_a_per_segment, double* dmin, void* stream);
int mcr_unproject_depth(const float* depth, int H, int W, const float* cameras, int64_t n_cam, float* world, void* stream);

/* ---- scene-grid bookkeeping fused (Scene.fill_cells macarons_utils.py:2727-2737 over Cell.fill :2551-2577; the cell lookup of
 * compute_scene_occupancy_probability_field :1434) --------------------------------------------------------------------------------
 * mcr_cell_keys: key[i] = linear id ((i_l * grid_w + i_w) * grid_h + i_h) of the cell point i falls in by upstream's floor rule
 *   (utils.floor_divide :113-117 on pts - x_min, capped at grid - 1); box_test != 0: n_cells instead when the point is outside the
 *   scene box [x_min, x_max] (closed, get_pts_in_bounding_box :2676-2691), not STRICTLY inside its cell's box (lo_tab / hi_tab
 *   [n_cells, 3]: Cell.fill's two masks) or not offered (valid[i] == 0; valid may be NULL).  grid_consts = x_min[3] x_max[3] step[3].
 * mcr_key_histogram: counts[k] = #{i: key[i] == k}, k = 0 .. nk, and their exclusive prefix sums offsets[0 .. nk + 1]  (nk <= 1023).
 * mcr_admit_keys: Cell.fill's admission on candidates sorted by cell: key2[i] = key_s[i] if the cell has more than n_point_min
 *   candidates (cand[key]) and the candidate's fp64 distance d[i] to the cell's stored points exceeds `resolution`, else nk. */
int mcr_cell_keys(const float* pts, int64_t N, const unsigned char* valid, const float* grid_consts, int grid_l, int grid_w, int grid_h,
                  const float* lo_tab, const float* hi_tab, int box_test, int* key, void* stream);
int mcr_key_histogram(const int* key, int64_t N, int nk, int64_t* counts, int64_t* offsets, void* stream);
int mcr_admit_keys(const double* d, const int* key_s, const int64_t* cand, int64_t N, double resolution, int64_t n_point_min, int nk,
                   int* key2, void* stream);

/* ---- one MACARONS decision without the host glue (testers/scene.py:391-454; macarons/utility/macarons_utils.py:2727-2737 Scene.fill_cells
 * over :2551-2577 Cell.fill, :1395-1540 compute_scene_occupancy_probability_field, :1631-1704 predict_coverage_gain_for_single_camera):
 * each phase of the loop body is a handful of launches behind ONE call, the host reads two small count tables back.
 *
 * mcr_group_by_key: stable grouping of N rows by key[i] in 0 .. nk (anything else counts as nk; nk <= 1023): order[pos] = source row --
 *   rows of one key contiguous, ascending row index inside a key, i.e. torch.sort(key, stable=True).indices -- with counts[0 .. nk] and
 *   their exclusive offsets[0 .. nk + 1].  (The reference compacts with boolean masks cell by cell; the sort replaced it in round 3.)
 *
 * mcr_scene_fill_begin: the device part of Scene.fill_cells for ALL cells: key[i] = cell of point i (mcr_cell_keys, box_test = 1),
 *   order = candidates grouped by cell, dmin[i] = fp64 distance of the i-th candidate IN CELL ORDER to the store of its cell
 *   (store_pts: every cell's stored points, cells in linear order; store_off [n_cells + 1] their offsets), key2 / order2 = the admitted
 *   candidates (Cell.fill :2562-2568) grouped by cell.  counts = cand [n_cells+1] | a_off [n_cells+2] | adm [n_cells+1] | adm_off [n_cells+2].
 * mcr_scene_fill_gather: new store row r = row g[r] of the virtual table [old store (n_store rows) | admitted candidates in cell order]
 *   (the host builds g from the cells' torch.randperm draws, :2573); features ride along (F floats per row; NULL = none).
 *
 * mcr_field_select: which proxy points take part in the occupancy pass and where (:1428-1442): stored_cell[p] = cell whose store holds
 *   proxy point p (features column 0 = proxy index; `pend_*` = the arrays of a mcr_scene_fill_begin whose gather has not run yet, NULL =
 *   none), proxy_proba <- 0 where seen (:1431), visit[c] = 1 for cells holding a seen point (:1434), rows_order = selected points
 *   grouped by storing cell (ascending index), oof_order = never-seen points first (ascending index).
 *   counts = visit [n+1] | sel_counts [n+1] | sel_off [n+2] | oof_counts [2] | oof_off [3], n = n_cells.
 * mcr_field_build: the (cell, chunk) jobs of the pass from host tables -- jobs [J][4] int64 = first position in rows_order, first query row,
 *   first cloud row, unused; segs [n_seg][4] int64 = first row in S_all (the surface store), first cloud row, job, unused; job_xf [J][20] =
 *   world->view matrix (16, row-vector), box centre in view space (3), 1 / (neighbourhood size x cell diagonal) (:1468-1478) --:
 *   rows [T], row_job [T], X_world [T,3] = the selected proxy points, X_q [T,3] / pc_all [tot,3] = queries / surface clouds in their
 *   jobs' prediction spaces (:1479-1484), vh [T,64] = view harmonics of the rows (mcr_view_harmonics_rows).
 * mcr_view_harmonics_rows: vh[t, k] = sum_v view_states[rows[t], bin_perm[v]] * vh_matrix_t[v, k]: move_view_state_to_view_space
 *   (scone_utils.py:863-931) + compute_view_harmonics (:934-960) of selected rows (rows / bin_perm may be NULL = identity).
 * mcr_field_finish: proxy_proba[rows[t]] = occ[t] (:1525), then the field's tail: the n_oof never-seen points with their stored
 *   probability (:1531-1537).
 *
 * mcr_camera_boxes: per neighbour camera k, centre of the bounding box of its first n_unique[k] sampled points (sampled [K,S,4]) in
 *   the prediction camera's view space (:1631-1641) and the camera centre in the normalised prediction space (:1655-1659).
 * mcr_macarons_gain_indexed: gains[k] = mean_s vis_unique[k, inverse[k,s]] * factor(|world_unique[k, inverse[k,s]] - cam_world[k]|)
 *   * volume[k], 0 when n_unique[k] == 0 (:1668-1704; the Monte-Carlo duplicates read through the inverse map).
 * mcr_philox_uniform_rows: out [K,S] = what K consecutive torch.rand(S, 1, device=...) calls return for a device generator at
 *   (seed, offset): the per-camera sampling uniforms of scone_utils.py:1052 in one launch (mapping 1 = rocRAND's (0,1] map, 0 = cuRAND's). */
size_t mcr_group_by_key_workspace_bytes(int64_t N, int nk);
int mcr_group_by_key(const int* key, int64_t N, int nk, int* order, int64_t* counts, int64_t* offsets, void* workspace,
                     size_t workspace_bytes, void* stream);
size_t mcr_scene_fill_workspace_bytes(int64_t N, int n_cells);
int mcr_scene_fill_begin(const float* pts, int64_t N, const unsigned char* valid, const float* grid_consts, int grid_l, int grid_w,
                         int grid_h, const float* lo_tab, const float* hi_tab, const float* store_pts, const int64_t* store_off,
                         double resolution, int64_t n_point_min, int* key, int* order, double* dmin, int* key2, int* order2,
                         int64_t* counts, void* workspace, size_t workspace_bytes, void* stream);
int mcr_scene_fill_gather(const int64_t* g, int64_t n_new, const float* store_pts, const float* store_fts, int64_t n_store, int F,
                          const float* pts, const float* features, const int* order, const int* order2, float* new_pts, float* new_fts,
                          void* stream);
/* The same with the row map evaluated on the device: pm = the touched cells' torch.randperm prefixes back to back (int32), tables =
 * new_off | b_off | adm_off | pm_off | touched, int64, n_cells + 1 entries each (offsets of the new store, of the old store, of the
 * admitted candidates, of pm; touched[c] != 0: cell c drew a permutation, else it keeps its rows in order). */
int mcr_scene_fill_gather_perm(const int* pm, const int64_t* tables, int n_cells, int64_t n_new, const float* store_pts,
                               const float* store_fts, int64_t n_store, int F, const float* pts, const float* features, const int* order,
                               const int* order2, float* new_pts, float* new_fts, void* stream);
size_t mcr_field_select_workspace_bytes(int64_t P, int n_cells);
int mcr_field_select(const float* proxy_points, int64_t P, const float* supervision_occ, const float* out_of_field, float* proxy_proba,
                     const float* store_fts, int F, int64_t n_store, const int64_t* store_off, const float* pend_features,
                     const int* pend_order, const int* pend_order2, const int* pend_key2, const int64_t* pend_adm_off, int64_t pend_N,
                     const float* grid_consts, int grid_l, int grid_w, int grid_h, int use_supervision_occ_mask, int* stored_cell,
                     int* key_sel, int* key_oof, int* rows_order, int* oof_order, int64_t* counts, void* workspace,
                     size_t workspace_bytes, void* stream);
int mcr_field_build(const int64_t* jobs, int J, const int64_t* segs, int n_seg, const float* job_xf, const int* rows_order,
                    const float* proxy_points, const float* S_all, const float* view_states, int n_bins, const int* bin_perm,
                    const float* vh_matrix_t, int64_t T, int64_t tot, int* rows, int* row_job, float* X_world, float* X_q, float* vh,
                    float* pc_all, void* stream);
int mcr_view_harmonics_rows(const float* view_states, int n_bins, const int* rows, const int* bin_perm, const float* vh_matrix_t,
                            int64_t T, float* vh, void* stream);
int mcr_field_finish(const int* rows, const float* occ, int64_t T, float* proxy_proba, const int* oof_order, int64_t n_oof,
                     const float* proxy_points, float* X_tail, float* occ_tail, void* stream);
int mcr_camera_boxes(const float* sampled, const int* n_unique, int64_t K, int S, const float* M_view, const float* cam_world,
                     float inv_diag, float* center, float* cam_view, void* stream);
int mcr_macarons_gain_indexed(const float* vis_unique, const float* world_unique, const int64_t* inverse, const int* n_unique,
                              const float* cam_world, const float* volume, float distance_th, int factor_mode, int64_t K, int S,
                              float* gains, void* stream);
int mcr_philox_uniform_rows(uint64_t seed, uint64_t offset, int64_t K, int S, int mapping, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MACARONS_HIP_H */
