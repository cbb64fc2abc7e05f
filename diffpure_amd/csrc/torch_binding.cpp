// torch.ops.diffpure_hip.*: the hot operators of libdiffpure_hip.so registered with PyTorch's dispatcher from C++
// (TORCH_LIBRARY), the "Native op ABI" row of SURVEY.md section 8b / BASELINE.json's "registered as torch extensions".
//
// Every operator is a thin, allocation-only shim over ONE C-ABI entry point of include/diffpure_hip.h: tensors in,
// tensors out, outputs from torch's caching allocator, the launch on the CURRENT HIP stream of the tensor's device under a
// device guard (what nn.DataParallel's per-replica threads need), TORCH_CHECK -> c10::Error -> Python RuntimeError with the
// library's message.  Implementations exist for the CUDA (= HIP on ROCm) dispatch key only: there is no CPU kernel.
// The column statistics a convolution's epilogue reduces for the GroupNorm that follows are an EXPLICIT second return
// (conv2d_nhwc_stats / conv2d_h2_stats), consumed by group_norm_stats_from_cols.
//
// Built by diffpure_amd/build.py into libdiffpure_torch.so (host C++ only; links libdiffpure_hip.so); loaded by
// diffpure_amd/torch_ops.py with torch.ops.load_library.
#include <ATen/ATen.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <c10/core/DeviceGuard.h>
#include <torch/library.h>

#include "../../include/diffpure_hip.h"

namespace {

using at::Tensor;

// On a ROCm build of PyTorch HIP devices carry the DeviceType "cuda": the stream comes from the masquerading accessor and the
// device guard is the generic one (it dispatches on the tensor's device type).
void* cur_stream(const Tensor& t) { return c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.device().index()).stream(); }

void chk_f32(const Tensor& t, const char* name, int64_t dim = -1) {
    TORCH_CHECK(t.is_cuda(), "diffpure_hip: ", name, " must be a GPU tensor (there is no CPU kernel)");
    TORCH_CHECK(t.scalar_type() == at::kFloat && t.is_contiguous(), "diffpure_hip: ", name, " must be contiguous float32");
    TORCH_CHECK(dim < 0 || t.dim() == dim, "diffpure_hip: ", name, " must have ", dim, " dimensions");
}
void chk_h(const Tensor& t, const char* name) {
    TORCH_CHECK(t.is_cuda() && t.scalar_type() == at::kHalf && t.is_contiguous(), "diffpure_hip: ", name,
                " must be a contiguous fp16 GPU tensor (operand format of dp_conv2d_nhwc_h2)");
}
const float* opt_ptr(const c10::optional<Tensor>& t, const char* name) {
    if (!t.has_value() || !t->defined()) return nullptr;
    chk_f32(*t, name);
    return t->data_ptr<float>();
}
#define DP_CALL(expr) TORCH_CHECK((expr) == 0, "diffpure_hip: ", dp_last_error())

// ---- convolution -------------------------------------------------------------------------------------------------
std::tuple<Tensor, Tensor> conv2d_nhwc_impl(const Tensor& x, const Tensor& wp, const c10::optional<Tensor>& bias, int64_t n_out,
                                            int64_t ksize, bool want_stats) {
    chk_f32(x, "x", 4);
    chk_f32(wp, "wp", 2);
    TORCH_CHECK(wp.size(0) == ksize * ksize * x.size(3) && wp.size(1) >= n_out, "diffpure_hip: weight panel does not match");
    c10::DeviceGuard guard(x.device());
    const int64_t B = x.size(0), H = x.size(1), W = x.size(2);
    Tensor out = at::empty({B, H, W, n_out}, x.options());
    Tensor cols;
    int tile_rows = 0;
    if (want_stats) cols = at::zeros({(B * H * W + 511) / 512 * 8, 2, n_out}, x.options());   // whole 512-row tiles: see ops._colstats_alloc
    DP_CALL(dp_conv2d_nhwc(x.data_ptr<float>(), (int)x.size(3), nullptr, 0, (int)B, (int)H, (int)W, (int)ksize, (int)ksize,
                           wp.data_ptr<float>(), (int)wp.size(1), (int)n_out, opt_ptr(bias, "bias"), nullptr, 0, nullptr, 0, 1.f,
                           out.data_ptr<float>(), (int)n_out, 0, want_stats ? cols.data_ptr<float>() : nullptr,
                           want_stats ? &tile_rows : nullptr, cur_stream(x)));
    return {out, cols};
}
Tensor conv2d_nhwc(const Tensor& x, const Tensor& wp, const c10::optional<Tensor>& bias, int64_t n_out, int64_t ksize) {
    return std::get<0>(conv2d_nhwc_impl(x, wp, bias, n_out, ksize, false));
}
std::tuple<Tensor, Tensor> conv2d_nhwc_stats(const Tensor& x, const Tensor& wp, const c10::optional<Tensor>& bias, int64_t n_out,
                                             int64_t ksize) {
    return conv2d_nhwc_impl(x, wp, bias, n_out, ksize, true);
}

std::tuple<Tensor, Tensor> conv2d_h2_impl(const Tensor& xh, const Tensor& wh, const c10::optional<Tensor>& bias, int64_t n_out,
                                          int64_t ksize, int64_t passes, int64_t w_fmt, bool want_stats) {
    chk_h(xh, "xh");
    chk_h(wh, "wh");
    TORCH_CHECK(w_fmt == 0 || w_fmt == 1, "diffpure_hip: w_fmt 0 (hi|lo weights) or 1 (plain fp16 weights)");
    // plain fp16 panels are stored in whole 32-row blocks (ops.order_conv_weight_w16): roundup(n_out, 32) rows
    TORCH_CHECK(xh.dim() == 4 && wh.dim() == 2 && wh.size(0) == (w_fmt ? (n_out + 31) / 32 * 32 : n_out),
                "diffpure_hip: conv2d_h2 operand shapes");
    const int64_t C = wh.size(1) / ((w_fmt ? 1 : 2) * ksize * ksize);
    const int a_fmt = xh.size(3) == C ? 1 : 0;                 // plain fp16 ("h1") or hi|lo octets ("h2")
    TORCH_CHECK(a_fmt == 1 || xh.size(3) == 2 * C, "diffpure_hip: activation operand does not match the weight panel");
    if (passes == 0) passes = w_fmt ? 1 : (a_fmt ? 2 : 3);
    c10::DeviceGuard guard(xh.device());
    const int64_t B = xh.size(0), H = xh.size(1) - 2, W = xh.size(2) - 2;
    auto fopt = xh.options().dtype(at::kFloat);
    Tensor out = at::empty({B, H, W, n_out}, fopt);
    Tensor cols, work;
    int tile_rows = 0;
    if (want_stats) cols = at::zeros({(B * H * W + 511) / 512 * 8, 2, n_out}, fopt);   // whole 512-row tiles: see ops._colstats_alloc
    const long long wbytes = dp_conv2d_nhwc_h2_workspace((int)B, (int)H, (int)W, (int)ksize, (int)C, (int)n_out);
    if (wbytes) work = at::empty({wbytes / 4}, fopt);
    DP_CALL(dp_conv2d_nhwc_h2(xh.data_ptr(), (int)C, (int)B, (int)H, (int)W, (int)ksize, wh.data_ptr(), (int)n_out,
                              opt_ptr(bias, "bias"), nullptr, 0, nullptr, 0, 1.f, out.data_ptr<float>(), (int)n_out,
                              want_stats ? cols.data_ptr<float>() : nullptr, want_stats ? &tile_rows : nullptr,
                              wbytes ? work.data_ptr() : nullptr, wbytes, (int)passes, a_fmt, (int)w_fmt, /*out_fmt=*/0, /*res_fmt=*/0,
                              /*1x1 K-segments: none*/ nullptr, 0, nullptr, 0, cur_stream(xh)));
    return {out, cols};
}
Tensor conv2d_h2(const Tensor& xh, const Tensor& wh, const c10::optional<Tensor>& bias, int64_t n_out, int64_t ksize, int64_t passes,
                 int64_t w_fmt) {
    return std::get<0>(conv2d_h2_impl(xh, wh, bias, n_out, ksize, passes, w_fmt, false));
}
std::tuple<Tensor, Tensor> conv2d_h2_stats(const Tensor& xh, const Tensor& wh, const c10::optional<Tensor>& bias, int64_t n_out,
                                           int64_t ksize, int64_t passes, int64_t w_fmt) {
    return conv2d_h2_impl(xh, wh, bias, n_out, ksize, passes, w_fmt, true);
}

// ---- GroupNorm ---------------------------------------------------------------------------------------------------
Tensor group_norm_stats_from_cols(const Tensor& cols, int64_t batch, int64_t hw, int64_t groups, double eps) {
    chk_f32(cols, "cols", 3);
    c10::DeviceGuard guard(cols.device());
    Tensor stats = at::empty({batch, groups, 2}, cols.options());
    DP_CALL(dp_gn_finalize_cols(cols.data_ptr<float>(), (int)cols.size(2), 64, nullptr, 0, 0, (int)batch, (int)hw, (int)groups,
                                (float)eps, stats.data_ptr<float>(), cur_stream(cols)));
    return stats;
}
Tensor group_norm_silu(const Tensor& x, const Tensor& gamma, const Tensor& beta, int64_t groups, double eps, bool act,
                       int64_t out_fmt, const c10::optional<Tensor>& stats_in) {
    chk_f32(x, "x", 4);
    chk_f32(gamma, "gamma", 1);
    chk_f32(beta, "beta", 1);
    TORCH_CHECK(out_fmt >= 0 && out_fmt <= 2, "diffpure_hip: out_fmt 0 (fp32), 1 (h2) or 2 (h1)");
    c10::DeviceGuard guard(x.device());
    const int64_t B = x.size(0), H = x.size(1), W = x.size(2), C = x.size(3);
    void* s = cur_stream(x);
    Tensor stats;
    if (stats_in.has_value() && stats_in->defined()) {
        chk_f32(*stats_in, "stats", 3);
        stats = *stats_in;
    } else {   // slab partition = a function of the image size only (results do not depend on the batch sharding)
        const int64_t ns = std::max<int64_t>(1, std::min<int64_t>(H * W / 256, 128));
        Tensor partial = at::empty({B, ns, groups, 2}, x.options());
        stats = at::empty({B, groups, 2}, x.options());
        DP_CALL(dp_gn_stats(x.data_ptr<float>(), (int)C, nullptr, 0, (int)B, (int)(H * W), (int)groups, (int)ns,
                            partial.data_ptr<float>(), s));
        DP_CALL(dp_gn_finalize(partial.data_ptr<float>(), (int)B, (int)ns, (int)groups, H * W * (C / groups), (float)eps,
                               stats.data_ptr<float>(), s));
    }
    Tensor y = out_fmt == 0 ? at::empty({B, H, W, C}, x.options())
                            : at::empty({B, H + 2, W + 2, out_fmt == 1 ? 2 * C : C}, x.options().dtype(at::kHalf));
    DP_CALL(dp_gn_apply(x.data_ptr<float>(), (int)C, nullptr, 0, (int)B, (int)H, (int)W, (int)groups, stats.data_ptr<float>(),
                        gamma.data_ptr<float>(), beta.data_ptr<float>(), nullptr, nullptr, 0, act ? 1 : 0, 0, (int)out_fmt,
                        y.data_ptr(), nullptr, nullptr, s));
    return y;
}

// ---- attention -----------------------------------------------------------------------------------------------------
Tensor attention(const Tensor& qkv, int64_t n_heads, bool legacy_layout) {
    chk_f32(qkv, "qkv", 3);
    c10::DeviceGuard guard(qkv.device());
    const int64_t B = qkv.size(0), T = qkv.size(1), C = qkv.size(2) / 3, d = C / n_heads;
    void* s = cur_stream(qkv);
    Tensor out = at::empty({B, T, C}, qkv.options());
    if ((d == 64 && T % 64 == 0) || (d == 256 && T % 128 == 0)) {       // flash-style kernel: the T x T scores never leave the registers
        Tensor work = at::empty({3 * B * T * C}, qkv.options());
        DP_CALL(dp_attention_fused(qkv.data_ptr<float>(), /*qkv_fmt=*/0, (int)B, (int)T, (int)C, (int)n_heads, legacy_layout ? 0 : 1,
                                   out.data_ptr<float>(), /*out_fmt=*/0, /*W=*/0, work.data_ptr(), s));
        return out;
    }
    const int64_t oq = 0, ok = legacy_layout ? d : C, ov = legacy_layout ? 2 * d : 2 * C, sh = legacy_layout ? 3 * d : d;
    const float* base = qkv.data_ptr<float>();
    Tensor scores = at::empty({B * n_heads, T, T}, qkv.options());
    const int c3 = (int)(3 * C);
    DP_CALL(dp_gemm_strided(base + oq, c3, T * 3 * C, sh, 0, base + ok, c3, T * 3 * C, sh, 1, scores.data_ptr<float>(), (int)T,
                            n_heads * T * T, T * T, (int)T, (int)T, (int)d, (int)B, (int)n_heads, 1.0f / std::sqrt((float)d), s));
    DP_CALL(dp_softmax_rows(scores.data_ptr<float>(), B * n_heads * T, (int)T, s));
    DP_CALL(dp_gemm_strided(scores.data_ptr<float>(), (int)T, n_heads * T * T, T * T, 0, base + ov, c3, T * 3 * C, sh, 0,
                            out.data_ptr<float>(), (int)C, T * C, d, (int)T, (int)d, (int)T, (int)B, (int)n_heads, 1.0f, s));
    return out;
}

// ---- solver step, resize -------------------------------------------------------------------------------------------
Tensor em_step(const Tensor& x, const Tensor& eps, double nhb, double gg, double sc, bool div, double h, double g, double sqrt_h,
               int64_t seed, int64_t sample0, int64_t step) {
    chk_f32(x, "x", 4);
    chk_f32(eps, "eps", 4);
    TORCH_CHECK(eps.size(0) == x.size(0) && eps.size(1) == x.size(1) && eps.size(2) == x.size(2) && eps.size(3) >= x.size(3),
                "diffpure_hip: em_step shapes");
    c10::DeviceGuard guard(x.device());
    Tensor out = at::empty_like(x);
    DP_CALL(dp_em_step(x.data_ptr<float>(), eps.data_ptr<float>(), (int)eps.size(3), (int)x.size(0), (int)(x.size(1) * x.size(2)),
                       (int)x.size(3), (float)nhb, (float)gg, (float)sc, div ? 1 : 0, (float)h, (float)g, (float)sqrt_h, nullptr,
                       (unsigned long long)seed, (long long)sample0, (int)step, out.data_ptr<float>(), cur_stream(x)));
    return out;
}
Tensor resize_affine(const Tensor& x, int64_t ho, int64_t wo, double shift, double scale, bool in_nhwc, bool out_nhwc) {
    chk_f32(x, "x", 4);
    c10::DeviceGuard guard(x.device());
    const int64_t B = x.size(0), C = in_nhwc ? x.size(3) : x.size(1), Hi = in_nhwc ? x.size(1) : x.size(2),
                  Wi = in_nhwc ? x.size(2) : x.size(3);
    Tensor y = out_nhwc ? at::empty({B, ho, wo, C}, x.options()) : at::empty({B, C, ho, wo}, x.options());
    DP_CALL(dp_resize_affine(x.data_ptr<float>(), (int)B, (int)C, (int)Hi, (int)Wi, in_nhwc ? 1 : 0, (float)shift, (float)scale,
                             y.data_ptr<float>(), (int)ho, (int)wo, out_nhwc ? 1 : 0, cur_stream(x)));
    return y;
}

}  // namespace

TORCH_LIBRARY(diffpure_hip, m) {
    m.def("conv2d_nhwc(Tensor x, Tensor wp, Tensor? bias, int n_out, int ksize) -> Tensor");
    m.def("conv2d_nhwc_stats(Tensor x, Tensor wp, Tensor? bias, int n_out, int ksize) -> (Tensor, Tensor)");
    m.def("conv2d_h2(Tensor xh, Tensor wh, Tensor? bias, int n_out, int ksize, int passes=0, int w_fmt=0) -> Tensor");
    m.def("conv2d_h2_stats(Tensor xh, Tensor wh, Tensor? bias, int n_out, int ksize, int passes=0, int w_fmt=0) -> (Tensor, Tensor)");
    m.def("group_norm_stats_from_cols(Tensor cols, int batch, int hw, int groups, float eps) -> Tensor");
    m.def("group_norm_silu(Tensor x, Tensor gamma, Tensor beta, int groups, float eps, bool act, int out_fmt, Tensor? stats=None) -> Tensor");
    m.def("attention(Tensor qkv, int n_heads, bool legacy_layout) -> Tensor");
    m.def("em_step(Tensor x, Tensor eps, float nhb, float gg, float sc, bool div, float h, float g, float sqrt_h, int seed, "
          "int sample0, int step) -> Tensor");
    m.def("resize_affine(Tensor x, int ho, int wo, float shift, float scale, bool in_nhwc, bool out_nhwc) -> Tensor");
}

TORCH_LIBRARY_IMPL(diffpure_hip, CUDA, m) {      // CUDA dispatch key = HIP devices on a ROCm build; no CPU implementation
    m.impl("conv2d_nhwc", conv2d_nhwc);
    m.impl("conv2d_nhwc_stats", conv2d_nhwc_stats);
    m.impl("conv2d_h2", conv2d_h2);
    m.impl("conv2d_h2_stats", conv2d_h2_stats);
    m.impl("group_norm_stats_from_cols", group_norm_stats_from_cols);
    m.impl("group_norm_silu", group_norm_silu);
    m.impl("attention", attention);
    m.impl("em_step", em_step);
    m.impl("resize_affine", resize_affine);
}
