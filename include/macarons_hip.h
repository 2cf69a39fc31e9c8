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
 *   (neighbour coordinates, minus the query if subtract_query).  k in {1,4,8,16}, k <= M. */
int mcr_knn_points(const float* X, const float* pc, int64_t* idx, float* dists, float* pts, int64_t B, int64_t Q,
                   int64_t M, int k, int subtract_query, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MACARONS_HIP_H */
