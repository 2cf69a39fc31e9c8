// torch.ops.macarons.* -- the operator layer SURVEY §8(b) asks for below the Python class boundary, as a C++ TORCH_LIBRARY
// extension: at::Tensor in / out, launches on torch's current HIP stream (c10::hip::getCurrentHIPStream...()), errors through TORCH_CHECK.  Every operator is a
// thin shim over the C ABI of include/macarons_hip.h (libmacarons_hip.so, linked): validation, output / scratch allocation
// through torch's caching allocator, one call.  Registered on the CUDA (= HIP on ROCm) dispatch key only: there is no CPU kernel.
// Reference op sequences (file:line, upstream tree) are cited per operator in include/macarons_hip.h.
#include <ATen/ATen.h>
#include <ATen/CPUGeneratorImpl.h>
#include <ATen/core/MT19937RNGEngine.h>
// torch on ROCm keeps the device type "cuda": the guard / stream accessors are the Masquerading-As-CUDA flavours of c10::hip
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <torch/library.h>
#include <hip/hip_runtime.h>
#include <immintrin.h>

#include <algorithm>
#include <limits>
#include <optional>
#include <set>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <exception>
#include <functional>
#include <map>
#include <mutex>
#include <thread>
#include <tuple>
#include <vector>

#include "macarons_hip.h"

namespace {

void* stream_of(const at::Tensor& t) { return (void*)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.device().index()).stream(); }

at::Tensor f32(const at::Tensor& t, const char* name) {
    TORCH_CHECK(t.is_cuda(), "macarons::", name, " must live on a HIP device (the MI355X hot path has no CPU fallback)");
    TORCH_CHECK(t.scalar_type() == at::kFloat, "macarons::", name, " must be float32");
    return t.contiguous();
}
void ok(int rc, const char* what) { TORCH_CHECK(rc == 0, what, " failed: ", mcr_last_error()); }
at::Tensor scratch(const at::Tensor& like, size_t bytes) {
    return at::empty({(int64_t)std::max<size_t>(bytes, 16)}, like.options().dtype(at::kByte));
}
std::vector<const float*> pointers(const c10::List<at::Tensor>& ts, std::vector<at::Tensor>& keep, const char* name) {
    std::vector<const float*> p;
    for (const at::Tensor& t : ts) {
        keep.push_back(f32(t, name));
        p.push_back(keep.back().data_ptr<float>());
    }
    return p;
}

// SconeVis.compute_coverage_gain (SconeVis.py:210-252): pts [B,N,3|4], harmonics [B,N,64], cams [B,C,3] -> [B,C]; with want_record also
// the decision of testers/shapenet.py:172 (torch.max over the cameras) as record [B,2] = (max gain, first arg-max camera as fp32)
std::tuple<at::Tensor, at::Tensor> sh_gain_impl(const at::Tensor& pts_, const at::Tensor& harm_, const at::Tensor& cams_, bool use_sigmoid,
                                               bool want_record) {
    const at::Tensor pts = f32(pts_, "pts"), harm = f32(harm_, "harmonics"), cams = f32(cams_, "cams");
    TORCH_CHECK(pts.dim() == 3 && harm.dim() == 3 && cams.dim() == 3, "sh_coverage_gain: pts [B,N,P], harmonics [B,N,64], cams [B,C,3]");
    const int64_t B = pts.size(0), N = pts.size(1), P = pts.size(2), C = cams.size(1);
    TORCH_CHECK(harm.size(0) == B && harm.size(1) == N && harm.size(2) == 64 && cams.size(0) == B && cams.size(2) == 3,
                "sh_coverage_gain: shape mismatch");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(pts.device());
    at::Tensor gains = at::empty({B, C}, pts.options());
    at::Tensor record = want_record ? at::empty({B, 2}, pts.options()) : at::Tensor();
    at::Tensor ws = scratch(pts, mcr_sh_coverage_gain_workspace_bytes(B, N, C));
    if (want_record)
        ok(mcr_sh_coverage_gain_best(pts.data_ptr<float>(), (int)P, harm.data_ptr<float>(), cams.data_ptr<float>(), gains.data_ptr<float>(),
                                     record.data_ptr<float>(), B, N, C, use_sigmoid ? 1 : 0, 0, ws.data_ptr(), (size_t)ws.numel(),
                                     stream_of(pts)), "mcr_sh_coverage_gain_best");
    else
        ok(mcr_sh_coverage_gain(pts.data_ptr<float>(), (int)P, harm.data_ptr<float>(), cams.data_ptr<float>(), gains.data_ptr<float>(), B, N, C,
                                use_sigmoid ? 1 : 0, 0, ws.data_ptr(), (size_t)ws.numel(), stream_of(pts)), "mcr_sh_coverage_gain");
    return std::make_tuple(gains, record);
}

at::Tensor sh_coverage_gain(const at::Tensor& pts, const at::Tensor& harm, const at::Tensor& cams, bool use_sigmoid) {
    return std::get<0>(sh_gain_impl(pts, harm, cams, use_sigmoid, false));
}

std::tuple<at::Tensor, at::Tensor> sh_coverage_gain_best(const at::Tensor& pts, const at::Tensor& harm, const at::Tensor& cams, bool use_sigmoid) {
    return sh_gain_impl(pts, harm, cams, use_sigmoid, true);
}

// SconeVis.compute_visibilities (SconeVis.py:164-208) -> [B,C,N]
at::Tensor sh_visibilities(const at::Tensor& pts_, const at::Tensor& harm_, const at::Tensor& cams_, bool use_sigmoid) {
    const at::Tensor pts = f32(pts_, "pts"), harm = f32(harm_, "harmonics"), cams = f32(cams_, "cams");
    TORCH_CHECK(pts.dim() == 3 && harm.dim() == 3 && cams.dim() == 3, "sh_visibilities: pts [B,N,P], harmonics [B,N,64], cams [B,C,3]");
    const int64_t B = pts.size(0), N = pts.size(1), P = pts.size(2), C = cams.size(1);
    TORCH_CHECK(harm.size(0) == B && harm.size(1) == N && harm.size(2) == 64 && cams.size(0) == B && cams.size(2) == 3,
                "sh_visibilities: shape mismatch");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(pts.device());
    at::Tensor vis = at::empty({B, C, N}, pts.options());
    ok(mcr_sh_visibilities(pts.data_ptr<float>(), (int)P, harm.data_ptr<float>(), cams.data_ptr<float>(), vis.data_ptr<float>(), B, N, C,
                           use_sigmoid ? 1 : 0, stream_of(pts)), "mcr_sh_visibilities");
    return vis;
}

// get_knn_points + the offset step (utils.py:1497-1509, SconeOcc.py:297-298) -> (offsets [B,Q,k,3], dists [B,Q,k], idx int64)
std::tuple<at::Tensor, at::Tensor, at::Tensor> knn_gather_offset(const at::Tensor& x_, const at::Tensor& pc_, int64_t k) {
    const at::Tensor x = f32(x_, "x"), pc = f32(pc_, "pc");
    TORCH_CHECK(x.dim() == 3 && pc.dim() == 3 && x.size(2) == 3 && pc.size(2) == 3 && pc.size(0) == x.size(0), "knn_gather_offset: x [B,Q,3], pc [B,M,3]");
    const int64_t B = x.size(0), Q = x.size(1), M = pc.size(1);
    c10::hip::HIPGuardMasqueradingAsCUDA guard(x.device());
    at::Tensor idx = at::empty({B, Q, k}, x.options().dtype(at::kLong)), d = at::empty({B, Q, k}, x.options()), pts = at::empty({B, Q, k, 3}, x.options());
    at::Tensor ws = scratch(x, mcr_knn_grid_workspace_bytes(B, Q, M));                   // grid-pruned search where it applies
    ok(mcr_knn_points_grid(x.data_ptr<float>(), pc.data_ptr<float>(), idx.data_ptr<int64_t>(), d.data_ptr<float>(), pts.data_ptr<float>(), B, Q, M,
                           (int)k, 1, ws.data_ptr(), (size_t)ws.numel(), stream_of(x)), "mcr_knn_points_grid");
    return {pts, d, idx};
}

// Camera.get_points_in_fov (macarons_utils.py:2400-2435): pts [P,3], camera records [n_cam,40] -> bool [n_cam,P]
at::Tensor points_in_fov(const at::Tensor& pts_, const at::Tensor& cams_) {
    const at::Tensor pts = f32(pts_, "pts"), cams = f32(cams_, "cameras");
    TORCH_CHECK(pts.dim() == 2 && pts.size(1) == 3 && cams.dim() == 2 && cams.size(1) == 40, "points_in_fov: pts [P,3], cameras [n_cam,40]");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(pts.device());
    at::Tensor mask = at::empty({cams.size(0), pts.size(0)}, pts.options().dtype(at::kByte));
    ok(mcr_points_in_fov(pts.data_ptr<float>(), pts.size(0), cams.data_ptr<float>(), (int)cams.size(0), mask.data_ptr<uint8_t>(), stream_of(pts)),
       "mcr_points_in_fov");
    return mask.to(at::kBool);
}

// compute_view_state (scone_utils.py:799-860) -> [B,Q,n_elev*n_azim]
at::Tensor view_state(const at::Tensor& pts_, const at::Tensor& xv_, int64_t n_elev, int64_t n_azim) {
    const at::Tensor pts = f32(pts_, "pts"), xv = f32(xv_, "X_view");
    TORCH_CHECK(pts.dim() == 3 && pts.size(2) >= 3 && xv.dim() == 2 && xv.size(1) == 3, "view_state: pts [B,Q,>=3], X_view [n_view,3]");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(pts.device());
    at::Tensor out = at::empty({pts.size(0), pts.size(1), n_elev * n_azim}, pts.options());
    ok(mcr_view_state(pts.data_ptr<float>(), (int)pts.size(2), xv.data_ptr<float>(), out.data_ptr<float>(), pts.size(0) * pts.size(1),
                      (int)xv.size(0), (int)n_elev, (int)n_azim, stream_of(pts)), "mcr_view_state");
    return out;
}

// compute_view_harmonics (scone_utils.py:934-960) as one product with the constant [n_harmonics, n_bins] matrix
at::Tensor view_harmonics(const at::Tensor& vs_, const at::Tensor& mat_) {
    const at::Tensor vs = f32(vs_, "view_state"), mat = f32(mat_, "matrix");
    TORCH_CHECK(mat.dim() == 2 && vs.size(-1) == mat.size(1), "view_harmonics: view_state [..., n_bins], matrix [n_harmonics, n_bins]");
    const int64_t K = mat.size(1), N = mat.size(0), M = vs.numel() / K;
    c10::hip::HIPGuardMasqueradingAsCUDA guard(vs.device());
    std::vector<int64_t> shape(vs.sizes().begin(), vs.sizes().end());
    shape.back() = N;
    at::Tensor out = at::empty(shape, vs.options());
    ok(mcr_linear(vs.data_ptr<float>(), K, mat.data_ptr<float>(), nullptr, nullptr, N, out.data_ptr<float>(), N, M, (int)N, (int)K, 0,
                  stream_of(vs)), "mcr_linear");
    return out;
}

// sample_proxy_points (scone_utils.py:1030-1061) -> (points+occupancy [n_u,4], harmonics [n_u,64], inverse [n], unique idx [n_u]);
// one host read-back of the unique count (the reference's torch.unique synchronises at the same point)
std::tuple<at::Tensor, at::Tensor, at::Tensor, at::Tensor> sample_proxy(const at::Tensor& X_, const at::Tensor& probs_, const at::Tensor& vh_,
                                                                        const at::Tensor& u_, double min_occ) {
    const at::Tensor X = f32(X_, "X"), probs = f32(probs_, "probs"), vh = f32(vh_, "view_harmonics"), u = f32(u_, "u");
    TORCH_CHECK(X.dim() == 2 && X.size(1) == 3 && probs.numel() == X.size(0) && vh.dim() == 2 && vh.size(0) == X.size(0) && vh.size(1) == 64,
                "sample_proxy: X [P,3], probs [P], view_harmonics [P,64]");
    const int64_t P = X.size(0), n = u.numel();
    c10::hip::HIPGuardMasqueradingAsCUDA guard(X.device());
    at::Tensor res = at::empty({n, 4}, X.options()), resh = at::empty({n, 64}, X.options());
    at::Tensor uniq = at::empty({n}, X.options().dtype(at::kLong)), inv = at::empty({n}, X.options().dtype(at::kLong));
    at::Tensor nu = at::zeros({1}, X.options().dtype(at::kInt));
    at::Tensor ws = scratch(X, mcr_sample_proxy_workspace_bytes(P, (int)n));
    ok(mcr_sample_proxy(X.data_ptr<float>(), probs.data_ptr<float>(), 1, vh.data_ptr<float>(), P, (float)min_occ, u.data_ptr<float>(), (int)n,
                        res.data_ptr<float>(), resh.data_ptr<float>(), uniq.data_ptr<int64_t>(), inv.data_ptr<int64_t>(), nu.data_ptr<int>(),
                        nullptr, ws.data_ptr(), (size_t)ws.numel(), stream_of(X)), "mcr_sample_proxy");
    const int64_t k = nu.item<int>();
    return {res.narrow(0, 0, k), resh.narrow(0, 0, k), inv, uniq.narrow(0, 0, k)};
}

// SconeVis.forward (SconeVis.py:121-162): pts [B,N,4], view_harmonics [B,N,64], the 48 weight tensors -> [B,N,64]
at::Tensor scone_vis_forward(const at::Tensor& pts_, const at::Tensor& vh_, c10::List<at::Tensor> weights) {
    const at::Tensor pts = f32(pts_, "pts"), vh = f32(vh_, "view_harmonics");
    TORCH_CHECK(pts.dim() == 3 && pts.size(2) == 4 && vh.dim() == 3 && vh.size(0) == pts.size(0) && vh.size(1) == pts.size(1) && vh.size(2) == 64,
                "scone_vis_forward: pts [B,N,4], view_harmonics [B,N,64]");
    const int64_t B = pts.size(0), N = pts.size(1);
    std::vector<at::Tensor> keep;
    const std::vector<const float*> w = pointers(weights, keep, "weights");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(pts.device());
    at::Tensor out = at::empty({B, N, 64}, pts.options());
    at::Tensor ws = scratch(pts, mcr_scone_vis_workspace_bytes(B, N));
    ok(mcr_scone_vis_forward(pts.data_ptr<float>(), vh.data_ptr<float>(), out.data_ptr<float>(), B, N, w.data(), (int)w.size(), nullptr,
                             ws.data_ptr(), (size_t)ws.numel(), stream_of(pts)), "mcr_scone_vis_forward");
    return out;
}

// SconeOcc.forward (SconeOcc.py:250-347) after the three down-sampling draws: pc_global [B,Lg,3], the three scale clouds,
// x [B,Q,3], view_harmonics [B,Q,64], the 140 weight tensors, the three packed local-transformer blobs (may be empty) -> [B,Q,1]
at::Tensor scone_occ_forward(const at::Tensor& pcg_, c10::List<at::Tensor> pc_scales, const at::Tensor& x_, const at::Tensor& vh_,
                             c10::List<at::Tensor> weights, c10::List<at::Tensor> local_blobs) {
    const at::Tensor pcg = f32(pcg_, "pc_global"), x = f32(x_, "x"), vh = f32(vh_, "view_harmonics");
    TORCH_CHECK(pc_scales.size() == 3, "scone_occ_forward: three scale clouds");
    TORCH_CHECK(x.dim() == 3 && x.size(2) == 3 && vh.dim() == 3 && vh.size(2) == 64 && pcg.dim() == 3, "scone_occ_forward: x [B,Q,3], view_harmonics [B,Q,64]");
    const int64_t B = x.size(0), Q = x.size(1), Lg = pcg.size(1);
    std::vector<at::Tensor> keep;
    const std::vector<const float*> sc = pointers(pc_scales, keep, "pc_scales"), w = pointers(weights, keep, "weights"),
                                    bl = pointers(local_blobs, keep, "local_blobs");
    TORCH_CHECK(bl.empty() || bl.size() == 3, "scone_occ_forward: local_blobs must be empty or hold the three packed transformers");
    int64_t Ms[3];
    for (int i = 0; i < 3; ++i) Ms[i] = keep[i].size(1);
    c10::hip::HIPGuardMasqueradingAsCUDA guard(x.device());
    at::Tensor out = at::empty({B, Q, 1}, x.options());
    at::Tensor ws = scratch(x, mcr_scone_occ_workspace_bytes(B, Q, Lg));
    ok(mcr_scone_occ_forward(pcg.data_ptr<float>(), Lg, sc.data(), Ms, x.data_ptr<float>(), vh.data_ptr<float>(), out.data_ptr<float>(), B, Q,
                             w.data(), (int)w.size(), bl.empty() ? nullptr : bl.data(), nullptr, nullptr, nullptr, ws.data_ptr(),
                             (size_t)ws.numel(), stream_of(x)), "mcr_scone_occ_forward");
    return out;
}

// ---- hidden draws of the reference, batched (host side; no device work) -----------------------------------------------------------
// The reference draws torch.randperm on the CPU default generator deep inside its loops: once per touched grid cell in Cell.fill
// (macarons_utils.py:2573) and three times per SconeOcc.forward (SconeOcc.py:269, :311).  A MACARONS decision makes ~200 of them;
// from Python every one is a dispatcher round trip plus index bookkeeping in numpy.  These two operators make the SAME at::randperm
// calls in the SAME order on the SAME generator -- the stream of draws is the reference's, bit for bit -- and return the index arrays
// the launches need, ready to upload.

// Only a PREFIX of most of these permutations is used (2048 of up to 27 000 points for the global transformer, 1 / ds of them for the
// coarser scales), and at::randperm's Fisher-Yates loop fixes position i in iteration i: the first k outputs are final after k
// iterations.  So the loop is restated here (randperm_cpu's branch for n < 2^32 / 20: z = random() % (n - i), swap(r[i], r[i + z]) for
// i < n - 1, random() = the next 32-bit output of the generator's mt19937 engine), run for the iterations that matter, and the engine
// is ADVANCED over the outputs the remaining iterations would have consumed (whole state blocks by the twist alone, no tempering, no
// division, no swap): the prefix and the generator state afterwards are at::randperm's, bit for bit (tests/test_draws_cpu.py holds
// both operators to loops of torch.randperm), at a third of the host time.
// The state transition on 8 words at a time (AVX2, chosen at run time; the recurrence s[k] = s[k + 397 mod 624] ^ twist(s[k], s[k + 1])
// reaches back 227 words, so eight consecutive k are independent): a decision's draws consume ~0.8 M generator outputs, most of
// them only skipped: the draws of the bench decision 395 -> 310 us of host time on the GPU box (the decision itself within noise: the
// GPU is not waiting for them any more).
__attribute__((target("avx2"))) static void mt_next_state_avx2(uint32_t* s) {
    const __m256i um = _mm256_set1_epi32((int)at::UMASK), lm = _mm256_set1_epi32((int)at::LMASK), one = _mm256_set1_epi32(1),
                  ma = _mm256_set1_epi32((int)at::MATRIX_A), zero = _mm256_setzero_si256();
    auto step = [&](int k, int src) __attribute__((target("avx2"))) {
        const __m256i u = _mm256_loadu_si256((const __m256i*)(s + k)), v = _mm256_loadu_si256((const __m256i*)(s + k + 1)),
                      m = _mm256_loadu_si256((const __m256i*)(s + src));
        const __m256i y = _mm256_or_si256(_mm256_and_si256(u, um), _mm256_and_si256(v, lm));
        const __m256i a = _mm256_and_si256(_mm256_sub_epi32(zero, _mm256_and_si256(v, one)), ma);
        _mm256_storeu_si256((__m256i*)(s + k), _mm256_xor_si256(_mm256_xor_si256(m, _mm256_srli_epi32(y, 1)), a));
    };
    auto scalar = [&](int k, int src, int nxt) {
        const uint32_t u = s[k], v = s[nxt];
        s[k] = s[src] ^ ((((u & at::UMASK) | (v & at::LMASK)) >> 1) ^ (v & 1 ? at::MATRIX_A : 0));
    };
    int k = 0;
    for (; k + 8 <= 227; k += 8) step(k, k + 397);         // k + 397 + 7 <= 623
    for (; k < 227; ++k) scalar(k, k + 397, k + 1);
    for (; k + 8 <= 623; k += 8) step(k, k - 227);         // (k + 8 <= 623: the vector's last successor is s[k + 8], still old)
    for (; k < 623; ++k) scalar(k, k - 227, k + 1);
    scalar(623, 396, 0);
}

struct Mt19937 {
    at::mt19937_data_pod d;
    static inline uint32_t twist(uint32_t u, uint32_t v) { return (((u & at::UMASK) | (v & at::LMASK)) >> 1) ^ (v & 1 ? at::MATRIX_A : 0); }
    void next_state() {                                    // at::mt19937_engine::next_state
        uint32_t* p = d.state_.data();
        d.left_ = at::MERSENNE_STATE_N;
        d.next_ = 0;
        static const bool avx2 = __builtin_cpu_supports("avx2") && !getenv("MCR_MT_SCALAR");      // (16 words at a time: no faster on Zen 5)
        if (avx2) { mt_next_state_avx2(p); return; }
        for (int j = at::MERSENNE_STATE_N - at::MERSENNE_STATE_M + 1; --j; p++) *p = p[at::MERSENNE_STATE_M] ^ twist(p[0], p[1]);
        for (int j = at::MERSENNE_STATE_M; --j; p++) *p = p[at::MERSENNE_STATE_M - at::MERSENNE_STATE_N] ^ twist(p[0], p[1]);
        *p = p[at::MERSENNE_STATE_M - at::MERSENNE_STATE_N] ^ twist(p[0], d.state_[0]);
    }
    inline uint32_t next() {                               // at::mt19937_engine::operator()
        if (--d.left_ == 0) next_state();
        uint32_t y = d.state_[d.next_++];
        y ^= (y >> 11);
        y ^= (y << 7) & 0x9d2c5680;
        y ^= (y << 15) & 0xefc60000;
        y ^= (y >> 18);
        return y;
    }
    void discard(int64_t n) {                              // n calls of next() whose values nobody looks at
        while (n > 0) {
            const int64_t avail = d.left_ - 1;             // calls before the one that regenerates the state
            if (n <= avail) {
                d.left_ -= (int)n;
                d.next_ += (uint32_t)n;
                return;
            }
            n -= avail + 1;                                // ... those, and the regenerating call itself (it consumes state_[0])
            next_state();
            d.next_ = 1;
        }
    }
};

struct CpuDraws {                                          // the default CPU generator's engine, held under its mutex for a batch of draws
    at::CPUGeneratorImpl* gen;
    std::unique_lock<std::mutex> lock;
    Mt19937 mt;
    std::vector<int32_t> r;                                // identity permutation, restored after every draw
    std::vector<int32_t> touched;
    CpuDraws() : gen(at::get_generator_or_default<at::CPUGeneratorImpl>(std::nullopt, at::detail::getDefaultCPUGenerator())), lock(gen->mutex_) {
        mt.d = gen->engine().data();
    }
    ~CpuDraws() {
        at::mt19937 e = gen->engine();
        e.set_data(mt.d);
        gen->set_engine(e);
    }
    // out[t] = base + randperm(n)[t] for t < min(k, n)
    void prefix(int64_t n, int64_t k, int64_t base, int64_t* out) {
        TORCH_CHECK(n >= 0 && n < (int64_t)(std::numeric_limits<uint32_t>::max() / 20), "randperm prefix: n out of the 32-bit branch of at::randperm");
        if ((int64_t)r.size() < n) {
            const size_t o = r.size();
            r.resize((size_t)n);
            for (size_t i = o; i < (size_t)n; ++i) r[i] = (int32_t)i;
        }
        k = std::min(k, n);
        const int64_t iters = std::min(k, std::max<int64_t>(n - 1, 0));   // position i is final after iteration i (the last one after n - 2)
        touched.clear();
        for (int64_t i = 0; i < iters; ++i) {
            const int64_t z = (int64_t)(mt.next() % (uint32_t)(n - i));      // (n < 2^32 / 20: the 32-bit division gives the same value)
            const int32_t sav = r[i];
            r[i] = r[z + i];
            r[z + i] = sav;
            touched.push_back((int32_t)(z + i));
        }
        for (int64_t t = 0; t < k; ++t) out[t] = base + r[t];
        for (int64_t i = 0; i < iters; ++i) r[i] = (int32_t)i;
        for (int32_t p : touched) r[p] = p;
        if (n - 1 > iters) mt.discard(n - 1 - iters);
    }
};

// randperm(n[i])[:keep[i]] for every i, concatenated.
at::Tensor randperm_prefixes(c10::IntArrayRef n, c10::IntArrayRef keep) {
    TORCH_CHECK(n.size() == keep.size(), "randperm_prefixes: one prefix length per draw");
    int64_t total = 0;
    for (size_t i = 0; i < n.size(); ++i) total += std::min<int64_t>(n[i], keep[i]);
    at::Tensor out = at::empty({total}, at::kLong);
    int64_t* o = out.data_ptr<int64_t>();
    CpuDraws draws;
    for (size_t i = 0; i < n.size(); ++i) {
        draws.prefix(n[i], keep[i], 0, o);
        o += std::min<int64_t>(n[i], keep[i]);
    }
    return out;
}

// The three draws of J SconeOcc.forward calls in job order -- randperm(m0)[:Lg], randperm(m0)[:m1], randperm(m1)[:m2] per job
// (m0 = the job's cloud, m1 = m0 // ds, m2 = m1 // ds) -- as the index arrays of SconeOcc.forward_ragged over the concatenated clouds:
//   [ g_idx (J*Lg: row of `pc` per global-transformer token, padded with the cloud's first row) | idx1 (rows of pc of scale 1) |
//     idx2 (rows of scale 1's cloud of scale 2) | off1 (J+1) | off2 (J+1) | g_len (J) ]     one int64 tensor.
at::Tensor scone_occ_draws(c10::IntArrayRef m0, c10::IntArrayRef m1, c10::IntArrayRef m2, int64_t Lg) {
    const int64_t J = (int64_t)m0.size();
    TORCH_CHECK((int64_t)m1.size() == J && (int64_t)m2.size() == J && Lg > 0, "scone_occ_draws: one (m0, m1, m2) per job");
    int64_t n1 = 0, n2 = 0;
    for (int64_t j = 0; j < J; ++j) {
        TORCH_CHECK(m0[j] > 0 && m1[j] <= m0[j] && m2[j] <= m1[j] && m1[j] >= 0 && m2[j] >= 0, "scone_occ_draws: bad sizes");
        n1 += m1[j]; n2 += m2[j];
    }
    at::Tensor out = at::empty({J * Lg + n1 + n2 + 2 * (J + 1) + J}, at::kLong);
    int64_t* g = out.data_ptr<int64_t>();
    int64_t* i1 = g + J * Lg;
    int64_t* i2 = i1 + n1;
    int64_t* off1 = i2 + n2;
    int64_t* off2 = off1 + J + 1;
    int64_t* glen = off2 + J + 1;
    int64_t c0 = 0, a1 = 0, a2 = 0;
    off1[0] = off2[0] = 0;
    CpuDraws draws;
    for (int64_t j = 0; j < J; ++j) {
        const int64_t n0 = std::min<int64_t>(m0[j], Lg);
        draws.prefix(m0[j], Lg, c0, g + j * Lg);                         // SconeOcc.py:269
        for (int64_t t = n0; t < Lg; ++t) g[j * Lg + t] = c0;
        glen[j] = n0;
        draws.prefix(m0[j], m1[j], c0, i1 + a1);                         // :311, scale 0 -> 1
        draws.prefix(m1[j], m2[j], a1, i2 + a2);                         // :311, scale 1 -> 2 (rows of scale 1's cloud)
        c0 += m0[j]; a1 += m1[j]; a2 += m2[j];
        off1[j + 1] = a1; off2[j + 1] = a2;
    }
    return out;
}

// The bin permutation of move_view_state_to_view_space (scone_utils.py:863-931) for a world->view rotation given as a tensor: the same
// ATen operators in the same order as macarons_amd.utility.scone_utils.view_space_bin_indices / CustomGeometry.get_spherical_coords
// (so the same bits: torch's own CPU kernels evaluate every element), without ~40 Python dispatcher round trips (200 -> ~60 us on the
// critical path of a MACARONS decision, whose GPU waits for this host work).  x_ref [n_elev * n_azim, 3] fp32 CPU: the lattice's unit
// directions; r [3, 3] fp32 CPU.  -> int64 [n_elev * n_azim]
at::Tensor view_space_bins(const at::Tensor& x_ref, const at::Tensor& r, int64_t n_elev, int64_t n_azim) {
    TORCH_CHECK(x_ref.device().is_cpu() && r.device().is_cpu() && x_ref.scalar_type() == at::kFloat && r.scalar_type() == at::kFloat,
                "view_space_bins: CPU fp32 tensors");
    const double pi = 3.141592653589793;
    const at::Tensor X = at::matmul(x_ref, r.view({3, 3}).t()).reshape({-1, 3});
    // get_spherical_coords (CustomGeometry.py:27-45)
    const at::Tensor r_x = at::linalg_norm(X, c10::nullopt, at::IntArrayRef{1});
    const at::Tensor x0 = X.select(1, 0), x1 = X.select(1, 1), x2 = X.select(1, 2);
    const at::Tensor yr = at::div(x1, r_x);
    at::Tensor elev = at::asin(yr);
    elev = at::where(at::le(yr, -1), at::full_like(elev, -pi / 2), elev);
    elev = at::where(at::ge(yr, 1), at::full_like(elev, pi / 2), elev);
    const at::Tensor q = at::div(x2, at::mul(r_x, at::cos(elev)));
    at::Tensor azim = at::acos(q);
    azim = at::where(at::le(q, -1), at::full_like(azim, pi), azim);
    azim = at::where(at::ge(q, 1), at::zeros_like(azim), azim);
    azim = at::where(at::lt(x0, 0), at::neg(azim), azim);
    // view_space_bin_indices (scone_utils.py:901-926)
    const double elev_step = pi / (double)(n_elev + 1), azim_step = 2 * pi / (double)n_azim;
    auto fd = [](const at::Tensor& a, double st) { return at::div(at::sub(a, at::remainder(a, st)), st); };     // utils.floor_divide
    at::Tensor ie = fd(elev, elev_step), ia = fd(azim, azim_step);
    ie = at::add(ie, at::gt(at::remainder(elev, elev_step), elev_step / 2.).to(ie.scalar_type()));
    ia = at::add(ia, at::gt(at::remainder(azim, azim_step), azim_step / 2.).to(ia.scalar_type()));
    ie = at::clamp(ie, -(double)(n_elev / 2), (double)(n_elev / 2));
    ia = at::where(at::gt(ia, n_azim / 2), at::full_like(ia, -(double)(n_azim / 2)), ia);
    ie = at::add(ie, n_elev / 2);
    ia = at::where(at::lt(ia, 0), at::add(ia, n_azim), ia);
    return at::add(at::mul(ie.to(at::kLong), n_azim), ia.to(at::kLong));
}

// Host data -> device tensor WITHOUT stalling the host (macarons_amd.ops.h2d's job, natively): a copy from pageable memory is a
// stream-ordered BLOCKING copy (the host waits for every kernel queued before it); here the bytes are staged in a small pool of re-used
// pinned buffers (power-of-two sizes; a buffer is taken again once the event behind its last copy has fired) and copied asynchronously
// on torch's current stream.  ~8 us per call instead of the 30-45 us of the same steps through a dozen Python-level torch calls; a
// MACARONS decision makes ten of them, six on its host-bound front section.
struct PinSlot { void* p; size_t cap; hipEvent_t ev; };
at::Tensor h2d(const at::Tensor& src, int64_t device_index) {
    TORCH_CHECK(src.device().is_cpu(), "h2d: a CPU tensor");
    const at::Tensor t = src.contiguous();
    const c10::Device dev(c10::DeviceType::CUDA, (c10::DeviceIndex)device_index);
    c10::hip::HIPGuardMasqueradingAsCUDA guard(dev);
    at::Tensor out = at::empty(t.sizes(), t.options().device(dev));
    const size_t nbytes = t.nbytes();
    if (!nbytes) return out;
    static std::mutex mu;
    static std::map<int64_t, std::vector<PinSlot>> pools;
    std::lock_guard<std::mutex> lock(mu);
    std::vector<PinSlot>& pool = pools[device_index];
    PinSlot* slot = nullptr;
    for (PinSlot& e : pool)
        if (e.cap >= nbytes && (!slot || e.cap < slot->cap) && hipEventQuery(e.ev) == hipSuccess) slot = &e;
    if (!slot) {
        size_t cap = 4096;
        while (cap < nbytes) cap <<= 1;
        PinSlot e{nullptr, cap, nullptr};
        TORCH_CHECK(hipHostMalloc(&e.p, cap, hipHostMallocDefault) == hipSuccess, "h2d: hipHostMalloc of ", cap, " bytes failed");
        TORCH_CHECK(hipEventCreateWithFlags(&e.ev, hipEventDisableTiming) == hipSuccess, "h2d: hipEventCreate failed");
        pool.push_back(e);
        slot = &pool.back();
    }
    std::memcpy(slot->p, t.data_ptr(), nbytes);
    hipStream_t s = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(dev.index()).stream();
    TORCH_CHECK(hipMemcpyAsync(out.data_ptr(), slot->p, nbytes, hipMemcpyHostToDevice, s) == hipSuccess, "h2d: hipMemcpyAsync failed");
    TORCH_CHECK(hipEventRecord(slot->ev, s) == hipSuccess, "h2d: hipEventRecord failed");
    return out;
}

// The (cell, chunk) job tables of the occupancy-field pass (macarons_utils.compute_scene_occupancy_probability_field; upstream walks the
// cells in Python, macarons_utils.py:1443-1478) from the counts the decision has just read back -- the ~30 numpy calls of the Python
// restatement (100 us between the read-back and the first launch behind it, with the GPU idle) as one loop.  Integer work only: the
// tables are those of the numpy code, element for element (tests/test_draws_cpu.py).
//   hostc: int64 CPU = [visit (n_cells) | pad | counts (n_cells) | pad | sel_off (n_cells + 2) | n_oof]   (the layout of mcr_field_select)
//   s_off [n_cells + 1]: the surface store's cell offsets; nbm [n_cells, 27]: neighbour cells, -1 padded; xf_all [n_cells, 20] fp32;
//   perm int32: the view-space bin permutation
// -> (raw uint8 = job table [J,4] int64 | segment table [n_seg,4] int64 | xf [J,20] fp32 | perm,
//     meta int64 = [J, n_seg, T, tot, n_oof | job_q (J) | job_m (J) | q_start (J+1) | m_start (J+1)])
std::tuple<at::Tensor, at::Tensor> field_jobs(const at::Tensor& hostc, const at::Tensor& s_off, const at::Tensor& nbm, const at::Tensor& xf_all,
                                              const at::Tensor& perm, int64_t n_cells, int64_t chunk, int64_t k_for_knn) {
    TORCH_CHECK(hostc.device().is_cpu() && hostc.scalar_type() == at::kLong && hostc.is_contiguous() && hostc.numel() >= 3 * n_cells + 5,
                "field_jobs: hostc");
    TORCH_CHECK(s_off.scalar_type() == at::kLong && s_off.numel() == n_cells + 1 && nbm.scalar_type() == at::kLong && nbm.numel() == n_cells * 27 &&
                xf_all.scalar_type() == at::kFloat && xf_all.numel() == n_cells * 20 && perm.scalar_type() == at::kInt && chunk > 0,
                "field_jobs: table shapes");
    const int64_t* hc = hostc.data_ptr<int64_t>();
    const int64_t *visit = hc, *counts = hc + n_cells + 1, *sel_off = hc + 2 * n_cells + 2;
    const int64_t n_oof = hc[3 * n_cells + 4];
    const at::Tensor s_off_c = s_off.contiguous(), nbm_c = nbm.contiguous(), xf_c = xf_all.contiguous(), perm_c = perm.contiguous();
    const int64_t *so = s_off_c.data_ptr<int64_t>(), *nb = nbm_c.data_ptr<int64_t>();
    std::vector<int64_t> m_cell(n_cells, 0);
    for (int64_t c = 0; c < n_cells; ++c)
        for (int k = 0; k < 27; ++k) {
            const int64_t q = nb[c * 27 + k];
            if (q >= 0) m_cell[c] += so[q + 1] - so[q];
        }
    std::vector<int64_t> job_cell, job_lo;
    for (int64_t c = 0; c < n_cells; ++c)
        if (visit[c] != 0 && m_cell[c] > 2 * 2 * k_for_knn && counts[c] > 0)                       // :1455-1456
            for (int64_t lo = 0; lo < counts[c]; lo += chunk) { job_cell.push_back(c); job_lo.push_back(lo); }
    const int64_t J = (int64_t)job_cell.size();
    int64_t n_seg = 0;
    for (int64_t j = 0; j < J; ++j)
        for (int k = 0; k < 27; ++k) {
            const int64_t q = nb[job_cell[j] * 27 + k];
            n_seg += q >= 0 && so[q + 1] - so[q] > 0;
        }
    at::Tensor meta = at::empty({5 + 2 * J + 2 * (J + 1)}, at::kLong);
    int64_t* me = meta.data_ptr<int64_t>();
    int64_t *job_q = me + 5, *job_m = job_q + J, *q_start = job_m + J, *m_start = q_start + J + 1;
    const int64_t perm_bytes = perm_c.numel() * 4;
    at::Tensor raw = at::empty({32 * (J + n_seg) + 80 * J + perm_bytes}, at::kByte);
    int64_t* jt = reinterpret_cast<int64_t*>(raw.data_ptr<uint8_t>());
    int64_t* st = jt + 4 * J;
    float* xf = reinterpret_cast<float*>(st + 4 * n_seg);
    q_start[0] = m_start[0] = 0;
    int64_t seg = 0, seg_dst = 0;
    const float* xa = xf_c.data_ptr<float>();
    for (int64_t j = 0; j < J; ++j) {
        const int64_t c = job_cell[j];
        job_q[j] = std::min<int64_t>(chunk, counts[c] - job_lo[j]);
        job_m[j] = m_cell[c];
        q_start[j + 1] = q_start[j] + job_q[j];
        m_start[j + 1] = m_start[j] + job_m[j];
        jt[4 * j] = sel_off[c] + job_lo[j]; jt[4 * j + 1] = q_start[j]; jt[4 * j + 2] = m_start[j]; jt[4 * j + 3] = 0;
        for (int k = 0; k < 27; ++k) {                                                            // its cell's non-empty neighbours, ascending
            const int64_t q = nb[c * 27 + k];
            if (q < 0) continue;
            const int64_t len = so[q + 1] - so[q];
            if (len <= 0) continue;
            st[4 * seg] = so[q]; st[4 * seg + 1] = seg_dst; st[4 * seg + 2] = j; st[4 * seg + 3] = 0;
            seg_dst += len; ++seg;
        }
        std::memcpy(xf + 20 * j, xa + 20 * c, 80);
    }
    std::memcpy(reinterpret_cast<uint8_t*>(xf + 20 * J), perm_c.data_ptr<int>(), perm_bytes);
    me[0] = J; me[1] = n_seg; me[2] = q_start[J]; me[3] = m_start[J]; me[4] = n_oof;
    return {raw, meta};
}

// The host tables SconeOcc.forward_ragged uploads before its first launch, as one loop instead of a dozen numpy calls:
//   [ off0 (J+1): cloud offsets | blocks (n_blocks x 4): (job, first query row, rows <= rows_per_block, 0) | row_job (T, optional) ]  int64
at::Tensor ragged_tables(c10::IntArrayRef cloud_sizes, c10::IntArrayRef query_sizes, int64_t rows_per_block, bool with_row_job) {
    const int64_t J = (int64_t)cloud_sizes.size();
    TORCH_CHECK((int64_t)query_sizes.size() == J && rows_per_block > 0, "ragged_tables: one (cloud, query) size per job");
    int64_t nb = 0, T = 0;
    for (int64_t j = 0; j < J; ++j) { nb += (query_sizes[j] + rows_per_block - 1) / rows_per_block; T += query_sizes[j]; }
    at::Tensor out = at::empty({J + 1 + 4 * nb + (with_row_job ? T : 0)}, at::kLong);
    int64_t* off0 = out.data_ptr<int64_t>();
    int64_t* blk = off0 + J + 1;
    int64_t* rj = blk + 4 * nb;
    off0[0] = 0;
    int64_t q0 = 0, b = 0;
    for (int64_t j = 0; j < J; ++j) {
        off0[j + 1] = off0[j] + cloud_sizes[j];
        for (int64_t r = 0; r < query_sizes[j]; r += rows_per_block, ++b) {
            blk[4 * b] = j; blk[4 * b + 1] = q0 + r; blk[4 * b + 2] = std::min<int64_t>(rows_per_block, query_sizes[j] - r); blk[4 * b + 3] = 0;
        }
        if (with_row_job) std::fill(rj + q0, rj + q0 + query_sizes[j], j);
        q0 += query_sizes[j];
    }
    return out;
}

// The count-independent host geometry of the occupancy-field pass (macarons_utils._field_prepare) as ONE call: every cell's
// prediction-box transform (world->view matrix | box centre in view space | 1 / (neighbourhood size x cell diagonal); :1468-1478) and
// the view-space bin permutation -- the same ATen operators in the same order as the Python restatement (same bits), callable from a
// worker thread (the dispatcher releases the GIL) while the main thread queues the decision's first launches.
std::tuple<at::Tensor, at::Tensor> field_prepare(const at::Tensor& mv, const at::Tensor& centers, const at::Tensor& diag, const at::Tensor& x_ref,
                                                 double pns, int64_t n_elev, int64_t n_azim) {
    TORCH_CHECK(mv.device().is_cpu() && mv.scalar_type() == at::kFloat && mv.numel() == 16 && centers.device().is_cpu() && diag.device().is_cpu(),
                "field_prepare: CPU fp32 tensors");
    const at::Tensor M = mv.reshape({4, 4});
    const int64_t n = centers.size(0);
    const at::Tensor cen_h = at::matmul(at::cat({centers, at::ones({n, 1}, centers.options())}, 1), M).slice(1, 0, 3);
    const at::Tensor inv_h = at::mul(at::reciprocal(at::mul(diag, pns)), 1.0).to(at::kFloat);
    const at::Tensor xf_all = at::cat({M.reshape({1, 16}).expand({n, -1}), cen_h, inv_h.view({n, 1})}, 1).contiguous();
    const at::Tensor perm = view_space_bins(x_ref, M.slice(0, 0, 3).slice(1, 0, 3).contiguous(), n_elev, n_azim).to(at::kInt);
    return {xf_all, perm};
}

// field_prepare on a worker thread of this library (no Python, no GIL): field_prepare_async queues the job and returns a ticket at once,
// field_prepare_wait hands the result over (blocking until it is there; an exception of the job is rethrown here).  The decision starts
// the job before its first launch and collects it in front of the read-back: ~170 us of host geometry off its critical path.
struct PrepWorker {
    std::mutex mu;
    std::condition_variable cv_job, cv_done;
    std::deque<std::pair<int64_t, std::function<std::tuple<at::Tensor, at::Tensor>()>>> jobs;
    std::map<int64_t, std::tuple<at::Tensor, at::Tensor>> results;
    std::map<int64_t, std::exception_ptr> errors;
    std::set<int64_t> open;                                               // tickets handed out and not collected yet
    int64_t next_ticket = 1;
    std::thread th;
    bool started = false;
    void run() {
        c10::InferenceMode no_grad;
        for (;;) {
            std::pair<int64_t, std::function<std::tuple<at::Tensor, at::Tensor>()>> job;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_job.wait(lk, [&] { return !jobs.empty(); });
                job = std::move(jobs.front());
                jobs.pop_front();
            }
            std::tuple<at::Tensor, at::Tensor> out;
            std::exception_ptr err;
            try { out = job.second(); } catch (...) { err = std::current_exception(); }
            {
                std::lock_guard<std::mutex> lk(mu);
                if (err) errors[job.first] = err; else results[job.first] = std::move(out);
            }
            cv_done.notify_all();
        }
    }
};

PrepWorker& prep_worker() {
    static PrepWorker* w = new PrepWorker();                               // (never destroyed: the thread outlives static teardown)
    return *w;
}

int64_t field_prepare_async(const at::Tensor& mv, const at::Tensor& centers, const at::Tensor& diag, const at::Tensor& x_ref,
                            double pns, int64_t n_elev, int64_t n_azim) {
    PrepWorker& w = prep_worker();
    std::lock_guard<std::mutex> lk(w.mu);
    if (!w.started) { w.th = std::thread([&w] { w.run(); }); w.th.detach(); w.started = true; }
    const int64_t ticket = w.next_ticket++;
    w.open.insert(ticket);
    const at::Tensor a = mv.detach().clone(), b = centers, c = diag, d = x_ref;      // (mv may be a view of a record the caller rewrites)
    w.jobs.emplace_back(ticket, [a, b, c, d, pns, n_elev, n_azim] { return field_prepare(a, b, c, d, pns, n_elev, n_azim); });
    w.cv_job.notify_one();
    return ticket;
}

std::tuple<at::Tensor, at::Tensor> field_prepare_wait(int64_t ticket) {
    PrepWorker& w = prep_worker();
    std::unique_lock<std::mutex> lk(w.mu);
    TORCH_CHECK(w.open.erase(ticket) == 1, "field_prepare_wait: unknown or already collected ticket ", ticket);
    w.cv_done.wait(lk, [&] { return w.results.count(ticket) || w.errors.count(ticket); });
    if (w.errors.count(ticket)) {
        std::exception_ptr e = w.errors[ticket];
        w.errors.erase(ticket);
        std::rethrow_exception(e);
    }
    std::tuple<at::Tensor, at::Tensor> out = std::move(w.results[ticket]);
    w.results.erase(ticket);
    return out;
}

}  // namespace

TORCH_LIBRARY(macarons, m) {
    m.def("field_prepare_async(Tensor mv, Tensor centers, Tensor diag, Tensor x_ref, float pns, int n_elev, int n_azim) -> int", &field_prepare_async);
    m.def("field_prepare_wait(int ticket) -> (Tensor, Tensor)", &field_prepare_wait);
    m.def("field_prepare(Tensor mv, Tensor centers, Tensor diag, Tensor x_ref, float pns, int n_elev, int n_azim) -> (Tensor, Tensor)", &field_prepare);
    m.def("ragged_tables(int[] cloud_sizes, int[] query_sizes, int rows_per_block, bool with_row_job) -> Tensor", &ragged_tables);
    m.def("field_jobs(Tensor hostc, Tensor s_off, Tensor nbm, Tensor xf_all, Tensor perm, int n_cells, int chunk, int k_for_knn) -> (Tensor, Tensor)", &field_jobs);
    m.def("h2d(Tensor src, int device) -> Tensor", &h2d);
    m.def("view_space_bins(Tensor x_ref, Tensor r, int n_elev, int n_azim) -> Tensor", &view_space_bins);
    m.def("randperm_prefixes(int[] n, int[] keep) -> Tensor", &randperm_prefixes);
    m.def("scone_occ_draws(int[] m0, int[] m1, int[] m2, int Lg) -> Tensor", &scone_occ_draws);
    m.def("sh_coverage_gain(Tensor pts, Tensor harmonics, Tensor cams, bool use_sigmoid) -> Tensor");
    m.def("sh_coverage_gain_best(Tensor pts, Tensor harmonics, Tensor cams, bool use_sigmoid) -> (Tensor, Tensor)");
    m.def("sh_visibilities(Tensor pts, Tensor harmonics, Tensor cams, bool use_sigmoid) -> Tensor");
    m.def("knn_gather_offset(Tensor x, Tensor pc, int k) -> (Tensor, Tensor, Tensor)");
    m.def("points_in_fov(Tensor pts, Tensor cameras) -> Tensor");
    m.def("view_state(Tensor pts, Tensor X_view, int n_elev, int n_azim) -> Tensor");
    m.def("view_harmonics(Tensor view_state, Tensor matrix) -> Tensor");
    m.def("sample_proxy(Tensor X, Tensor probs, Tensor view_harmonics, Tensor u, float min_occ) -> (Tensor, Tensor, Tensor, Tensor)");
    m.def("scone_vis_forward(Tensor pts, Tensor view_harmonics, Tensor[] weights) -> Tensor");
    m.def("scone_occ_forward(Tensor pc_global, Tensor[] pc_scales, Tensor x, Tensor view_harmonics, Tensor[] weights, Tensor[] local_blobs) -> Tensor");
}

TORCH_LIBRARY_IMPL(macarons, CUDA, m) {
    m.impl("sh_coverage_gain", &sh_coverage_gain);
    m.impl("sh_coverage_gain_best", &sh_coverage_gain_best);
    m.impl("sh_visibilities", &sh_visibilities);
    m.impl("knn_gather_offset", &knn_gather_offset);
    m.impl("points_in_fov", &points_in_fov);
    m.impl("view_state", &view_state);
    m.impl("view_harmonics", &view_harmonics);
    m.impl("sample_proxy", &sample_proxy);
    m.impl("scone_vis_forward", &scone_vis_forward);
    m.impl("scone_occ_forward", &scone_occ_forward);
}
