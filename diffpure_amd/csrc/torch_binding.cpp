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
#include <torch/csrc/autograd/custom_function.h>

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

// ABI 8 (round 6): the write-bound stem kernel (csrc/stem.hip): x [B, H, W, 3] fp32, w16 = ops.pack_stem_weight -> (out, column records)
std::tuple<Tensor, Tensor> conv2d_stem(const Tensor& x, const Tensor& w16, const c10::optional<Tensor>& bias, bool out_f16, bool want_stats) {
    chk_f32(x, "x", 4);
    chk_h(w16, "w16");
    TORCH_CHECK(w16.dim() == 3 && w16.size(0) == 2 && w16.size(2) == 32, "diffpure_hip: conv2d_stem wants the [2, N, 32] fp16 (hi | lo) panel");
    const int64_t B = x.size(0), H = x.size(1), W = x.size(2), N = w16.size(1);
    TORCH_CHECK(dp_conv2d_stem_ok((int)x.size(3), (int)B, (int)H, (int)W, (int)N), "diffpure_hip: conv2d_stem does not serve this shape");
    c10::DeviceGuard guard(x.device());
    Tensor out = at::empty({B, H, W, N}, x.options().dtype(out_f16 ? at::kHalf : at::kFloat));
    Tensor cols;
    int tile_rows = 0;
    if (want_stats) cols = at::zeros({(B * H * W + 511) / 512 * 8, 2, N}, x.options());
    DP_CALL(dp_conv2d_stem(x.data_ptr<float>(), (int)x.size(3), (int)B, (int)H, (int)W, w16.data_ptr(), (int)N, opt_ptr(bias, "bias"),
                           out.data_ptr(), out_f16 ? 1 : 0, want_stats ? cols.data_ptr<float>() : nullptr, want_stats ? &tile_rows : nullptr,
                           cur_stream(x)));
    return {out, cols};
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


// ==== ABI 6 / 7 (rounds 4-5): the operators of the fp16 residual stream, and the backward entry points ============================
const void* opt_h_ptr(const c10::optional<Tensor>& t, const char* name) {
    if (!t.has_value() || !t->defined()) return nullptr;
    chk_h(*t, name);
    return t->data_ptr();
}

// conv2d_h2_ex: the whole epilogue contract of dp_conv2d_nhwc_h2 - bias, per-sample time-embedding rows, fp32 or plain-fp16
// residual, scale, fp32 or plain-fp16 output, column records, and the 1x1 K-segments of a fused skip convolution.
std::tuple<Tensor, Tensor> conv2d_h2_ex(const Tensor& xh, const Tensor& wh, const c10::optional<Tensor>& bias, const c10::optional<Tensor>& temb,
                                        const c10::optional<Tensor>& res, const c10::optional<Tensor>& seg1, const c10::optional<Tensor>& seg2,
                                        int64_t n_out, int64_t ksize, int64_t passes, int64_t w_fmt, double scale, bool out_f16, bool want_stats) {
    chk_h(xh, "xh");
    chk_h(wh, "wh");
    TORCH_CHECK(w_fmt == 0 || w_fmt == 1, "diffpure_hip: w_fmt 0 (hi|lo weights) or 1 (plain fp16 weights)");
    TORCH_CHECK(xh.dim() == 4 && wh.dim() == 2 && wh.size(0) == (w_fmt ? (n_out + 31) / 32 * 32 : n_out), "diffpure_hip: conv2d_h2_ex operand shapes");
    const int64_t B = xh.size(0), H = xh.size(1) - 2, W = xh.size(2) - 2;
    int64_t sc1 = 0, sc2 = 0;
    const void* sp1 = opt_h_ptr(seg1, "seg1");
    const void* sp2 = opt_h_ptr(seg2, "seg2");
    if (sp1) {
        TORCH_CHECK(w_fmt == 1 && seg1->dim() == 4 && seg1->size(0) == B && seg1->size(1) == H && seg1->size(2) == W, "diffpure_hip: seg1 is [B,H,W,Cs] fp16 on the fp16 x fp16 path");
        sc1 = seg1->size(3);
    }
    if (sp2) {
        TORCH_CHECK(sp1 && seg2->dim() == 4 && seg2->size(0) == B && seg2->size(1) == H && seg2->size(2) == W, "diffpure_hip: seg2 only after seg1, [B,H,W,Cs] fp16");
        sc2 = seg2->size(3);
    }
    const int64_t C = (wh.size(1) / (w_fmt ? 1 : 2) - sc1 - sc2) / (ksize * ksize);
    const int a_fmt = xh.size(3) == C ? 1 : 0;
    TORCH_CHECK(a_fmt == 1 || xh.size(3) == 2 * C, "diffpure_hip: activation operand does not match the weight panel");
    if (passes == 0) passes = w_fmt ? 1 : (a_fmt ? 2 : 3);
    c10::DeviceGuard guard(xh.device());
    auto fopt = xh.options().dtype(at::kFloat);
    Tensor out = at::empty({B, H, W, n_out}, out_f16 ? xh.options() : fopt);
    const float* tptr = nullptr;
    int tstride = 0;
    if (temb.has_value() && temb->defined()) {
        TORCH_CHECK(temb->is_cuda() && temb->scalar_type() == at::kFloat && temb->dim() == 2 && temb->stride(1) == 1 && temb->size(1) >= n_out &&
                        (temb->size(0) == 1 || temb->size(0) == B), "diffpure_hip: temb is [1 | B, >= n_out] fp32 rows (a column view of a wider table is fine)");
        tptr = temb->data_ptr<float>();
        tstride = temb->size(0) == 1 ? 0 : (int)temb->stride(0);
    }
    const void* rptr = nullptr;
    int rfmt = 0;
    if (res.has_value() && res->defined()) {
        TORCH_CHECK(res->is_cuda() && res->is_contiguous() && res->dim() == 4 && res->size(0) == B && res->size(1) == H && res->size(2) == W && res->size(3) == n_out,
                    "diffpure_hip: res is a contiguous [B,H,W,n_out] tensor");
        TORCH_CHECK(res->scalar_type() == at::kFloat || res->scalar_type() == at::kHalf, "diffpure_hip: res is fp32 or plain fp16");
        rfmt = res->scalar_type() == at::kHalf ? 1 : 0;
        rptr = res->data_ptr();
    }
    Tensor cols, work;
    int tile_rows = 0;
    if (want_stats) cols = at::zeros({(B * H * W + 511) / 512 * 8, 2, n_out}, fopt);
    const long long wbytes = dp_conv2d_nhwc_h2_workspace((int)B, (int)H, (int)W, (int)ksize, (int)C, (int)n_out);
    if (wbytes) work = at::empty({wbytes / 4}, fopt);
    DP_CALL(dp_conv2d_nhwc_h2(xh.data_ptr(), (int)C, (int)B, (int)H, (int)W, (int)ksize, wh.data_ptr(), (int)n_out, opt_ptr(bias, "bias"), tptr, tstride,
                              rptr, rptr ? (int)n_out : 0, (float)scale, out.data_ptr(), (int)n_out, want_stats ? cols.data_ptr<float>() : nullptr,
                              want_stats ? &tile_rows : nullptr, wbytes ? work.data_ptr() : nullptr, wbytes, (int)passes, a_fmt, (int)w_fmt,
                              out_f16 ? 1 : 0, rfmt, sp1, (int)sc1, sp2, (int)sc2, cur_stream(xh)));
    return {out, cols};
}

// gn_apply_h16: GroupNorm-apply (+FiLM) (+SiLU) (+2x resample) over PLAIN fp16 tensors (dp_gn_apply_h16): out_fmt 2 = the zero-bordered
// "h1" operand, 3 = a plain fp16 tensor; gamma == None: no normalisation (the resampler of an identity skip); raw: second output.
std::tuple<Tensor, Tensor> gn_apply_h16(const Tensor& x, const c10::optional<Tensor>& x2, const c10::optional<Tensor>& stats,
                                        const c10::optional<Tensor>& gamma, const c10::optional<Tensor>& beta, const c10::optional<Tensor>& fscale,
                                        const c10::optional<Tensor>& fshift, int64_t groups, bool act, int64_t resample, int64_t out_fmt, bool raw) {
    chk_h(x, "x");
    TORCH_CHECK(x.dim() == 4 && (out_fmt == 2 || out_fmt == 3) && resample >= 0 && resample <= 2, "diffpure_hip: gn_apply_h16 arguments");
    const int64_t B = x.size(0), H = x.size(1), W = x.size(2), C1 = x.size(3);
    int64_t C2 = 0;
    const void* x2p = opt_h_ptr(x2, "x2");
    if (x2p) {
        TORCH_CHECK(x2->dim() == 4 && x2->size(0) == B && x2->size(1) == H && x2->size(2) == W, "diffpure_hip: x2 shape");
        C2 = x2->size(3);
    }
    const int64_t C = C1 + C2, Ho = resample == 1 ? 2 * H : (resample == 2 ? H / 2 : H), Wo = resample == 1 ? 2 * W : (resample == 2 ? W / 2 : W);
    // FiLM rows [1 | B, C]: the two may be column views of one [R, 2C] table (unit column stride, a common row stride)
    const float* fs = nullptr;
    const float* fh = nullptr;
    int fstride = 0;
    if (fscale.has_value() && fscale->defined()) {
        TORCH_CHECK(fshift.has_value() && fshift->defined(), "diffpure_hip: FiLM scale without shift");
        for (const Tensor* t : {&*fscale, &*fshift})
            TORCH_CHECK(t->is_cuda() && t->scalar_type() == at::kFloat && t->dim() == 2 && t->stride(1) == 1 && t->size(1) == C &&
                            (t->size(0) == 1 || t->size(0) == B), "diffpure_hip: FiLM rows are fp32 [1 | B, C] with unit column stride");
        TORCH_CHECK(fscale->stride(0) == fshift->stride(0) && fscale->size(0) == fshift->size(0), "diffpure_hip: FiLM scale and shift share a row stride");
        fs = fscale->data_ptr<float>();
        fh = fshift->data_ptr<float>();
        fstride = fscale->size(0) == 1 ? 0 : (int)fscale->stride(0);
    }
    c10::DeviceGuard guard(x.device());
    Tensor y = out_fmt == 2 ? at::zeros({B, Ho + 2, Wo + 2, C}, x.options()) : at::empty({B, Ho, Wo, C}, x.options());
    Tensor yr;
    if (raw) {
        TORCH_CHECK(out_fmt == 2, "diffpure_hip: the raw second output comes with the operand form (out_fmt 2)");
        // resample 0: the raw input in operand form; 1 | 2 (ABI 8): the resampled raw input as a plain tensor (identity skip of an up / down block)
        yr = resample == 0 ? at::zeros_like(y) : at::empty({B, Ho, Wo, C}, x.options());
    }
    DP_CALL(dp_gn_apply_h16(x.data_ptr(), (int)C1, x2p, (int)C2, (int)B, (int)H, (int)W, (int)groups, opt_ptr(stats, "stats"), opt_ptr(gamma, "gamma"),
                            opt_ptr(beta, "beta"), fs, fh, fstride, act ? 1 : 0, (int)resample, (int)out_fmt, y.data_ptr(), raw ? yr.data_ptr() : nullptr,
                            cur_stream(x)));
    return {y, yr};
}

// attention_fused: the flash-style kernel on fp32 qkv (three passes) or plain-fp16 qkv (qkv_fmt 1: ONE pass, Q / K read in place);
// operand_w > 0: the result is the zero-bordered fp16 operand [B, T / W + 2, W + 2, C] of the proj_out convolution.
Tensor attention_fused(const Tensor& qkv, int64_t n_heads, bool legacy_layout, int64_t operand_w) {
    TORCH_CHECK(qkv.is_cuda() && qkv.is_contiguous() && qkv.dim() == 3 && (qkv.scalar_type() == at::kFloat || qkv.scalar_type() == at::kHalf),
                "diffpure_hip: qkv is a contiguous [B, T, 3C] fp32 or fp16 GPU tensor");
    const int64_t B = qkv.size(0), T = qkv.size(1), C = qkv.size(2) / 3, d = C / n_heads;
    TORCH_CHECK((d == 64 && T % 64 == 0) || (d == 256 && T % 128 == 0), "diffpure_hip: the fused kernel covers head dimension 64 (T % 64 == 0) and 256 (T % 128 == 0)");
    const bool f16 = qkv.scalar_type() == at::kHalf;
    c10::DeviceGuard guard(qkv.device());
    Tensor work = f16 ? at::empty({B * T * C}, qkv.options()) : at::empty({3 * B * T * C}, qkv.options());
    Tensor out;
    if (operand_w > 0) {
        TORCH_CHECK(T % operand_w == 0, "diffpure_hip: T must be a multiple of the image width");
        out = at::zeros({B, T / operand_w + 2, operand_w + 2, C}, qkv.options().dtype(at::kHalf));     // the caller zeroes the border (header)
    } else {
        out = at::empty({B, T, C}, qkv.options().dtype(at::kFloat));
    }
    DP_CALL(dp_attention_fused(qkv.data_ptr(), f16 ? 1 : 0, (int)B, (int)T, (int)C, (int)n_heads, legacy_layout ? 0 : 1, out.data_ptr(), operand_w > 0 ? 1 : 0,
                               (int)operand_w, work.data_ptr(), cur_stream(qkv)));
    return out;
}

// round_weights: fp32 masters -> plain fp16 panels, in place on `work` (round to nearest, or unbiased stochastic rounding keyed by
// Philox(seed, key, element): precision "f16sr" re-rounds every panel of a network with one launch before every UNet call).
void round_weights(const Tensor& master, Tensor work, bool stochastic, int64_t seed, int64_t key) {
    chk_f32(master, "master", 1);
    TORCH_CHECK(work.is_cuda() && work.scalar_type() == at::kHalf && work.is_contiguous() && work.numel() == master.numel() && master.numel() % 8 == 0,
                "diffpure_hip: work is the fp16 twin of master (a multiple of 8 elements)");
    c10::DeviceGuard guard(master.device());
    DP_CALL(dp_round_weights(master.data_ptr<float>(), work.data_ptr(), master.numel(), stochastic ? 1 : 0, (unsigned long long)seed, (long long)key,
                             cur_stream(master)));
}

// ---- backward entry points ----------------------------------------------------------------------------------------------
// group_norm_silu_bwd: input gradient of group_norm_silu(out_fmt 0): the one-pass kernel where the shape fits, else stats + apply.
Tensor group_norm_silu_bwd(const Tensor& x, const Tensor& stats, const Tensor& gamma, const Tensor& beta, const Tensor& dy, int64_t groups, bool act) {
    chk_f32(x, "x", 4);
    chk_f32(dy, "dy", 4);
    chk_f32(stats, "stats", 3);
    chk_f32(gamma, "gamma", 1);
    chk_f32(beta, "beta", 1);
    TORCH_CHECK(dy.sizes() == x.sizes(), "diffpure_hip: dy has the shape of x");
    c10::DeviceGuard guard(x.device());
    const int B = (int)x.size(0), H = (int)x.size(1), W = (int)x.size(2), C = (int)x.size(3), G = (int)groups;
    void* s = cur_stream(x);
    Tensor dx = at::empty_like(x);
    if (dp_gn_bwd_fused_ok(H, W, C, 0, G, 0)) {
        DP_CALL(dp_gn_bwd_fused(x.data_ptr<float>(), C, nullptr, 0, /*x_fmt=*/0, B, H, W, G, stats.data_ptr<float>(), gamma.data_ptr<float>(), beta.data_ptr<float>(),
                                nullptr, nullptr, 0, act ? 1 : 0, 0, dy.data_ptr<float>(), 0, dx.data_ptr(), nullptr, nullptr, nullptr, 1.f, s));
        return dx;
    }
    const int ns = (int)std::max<int64_t>(1, std::min<int64_t>((int64_t)H * W / 256, 128));
    Tensor partial = at::empty({B, ns, G, 2}, x.options()), sums = at::empty({B, G, 2}, x.options());
    DP_CALL(dp_gn_bwd_stats(x.data_ptr<float>(), C, nullptr, 0, /*x_fmt=*/0, B, H, W, G, stats.data_ptr<float>(), gamma.data_ptr<float>(), beta.data_ptr<float>(), nullptr,
                            nullptr, 0, act ? 1 : 0, 0, nullptr, dy.data_ptr<float>(), ns, partial.data_ptr<float>(), sums.data_ptr<float>(), s));
    DP_CALL(dp_gn_bwd_apply(x.data_ptr<float>(), C, nullptr, 0, /*x_fmt=*/0, B, H, W, G, stats.data_ptr<float>(), gamma.data_ptr<float>(), beta.data_ptr<float>(), nullptr,
                            nullptr, 0, act ? 1 : 0, 0, nullptr, dy.data_ptr<float>(), sums.data_ptr<float>(), 0, dx.data_ptr(), nullptr, nullptr, nullptr, 1.f, s));
    return dx;
}
Tensor group_norm_stats(const Tensor& x, int64_t groups, double eps) {
    chk_f32(x, "x", 4);
    c10::DeviceGuard guard(x.device());
    const int64_t B = x.size(0), H = x.size(1), W = x.size(2), C = x.size(3);
    const int64_t ns = std::max<int64_t>(1, std::min<int64_t>(H * W / 256, 128));
    Tensor partial = at::empty({B, ns, groups, 2}, x.options()), stats = at::empty({B, groups, 2}, x.options());
    void* s = cur_stream(x);
    DP_CALL(dp_gn_stats(x.data_ptr<float>(), (int)C, nullptr, 0, (int)B, (int)(H * W), (int)groups, (int)ns, partial.data_ptr<float>(), s));
    DP_CALL(dp_gn_finalize(partial.data_ptr<float>(), (int)B, (int)ns, (int)groups, H * W * (C / groups), (float)eps, stats.data_ptr<float>(), s));
    return stats;
}

// attention_bwd: d qkv for attention(qkv) given d out; the probabilities are recomputed here (as the engines' taped forward does).
Tensor attention_bwd(const Tensor& qkv, const Tensor& dout, int64_t n_heads, bool legacy_layout) {
    chk_f32(qkv, "qkv", 3);
    chk_f32(dout, "dout", 3);
    c10::DeviceGuard guard(qkv.device());
    const int64_t B = qkv.size(0), T = qkv.size(1), C = qkv.size(2) / 3, d = C / n_heads;
    TORCH_CHECK(dout.size(0) == B && dout.size(1) == T && dout.size(2) == C, "diffpure_hip: dout is [B, T, C]");
    const int64_t oq = 0, ok = legacy_layout ? d : C, ov = legacy_layout ? 2 * d : 2 * C, sh = legacy_layout ? 3 * d : d;
    void* s = cur_stream(qkv);
    const float* q0 = qkv.data_ptr<float>();
    const float* do0 = dout.data_ptr<float>();
    const int c3 = (int)(3 * C), t = (int)T, dd = (int)d, b = (int)B, nh = (int)n_heads, c = (int)C;
    const long long zb = n_heads * T * T, zh = T * T, sq = T * 3 * C;
    const float sc = 1.0f / std::sqrt((float)d);
    Tensor probs = at::empty({B * n_heads, T, T}, qkv.options()), dp = at::empty({B * n_heads, T, T}, qkv.options()), dqkv = at::empty_like(qkv);
    float* g0 = dqkv.data_ptr<float>();
    DP_CALL(dp_gemm_strided(q0 + oq, c3, sq, sh, 0, q0 + ok, c3, sq, sh, 1, probs.data_ptr<float>(), t, zb, zh, t, t, dd, b, nh, sc, s));
    DP_CALL(dp_softmax_rows(probs.data_ptr<float>(), B * n_heads * T, t, s));
    DP_CALL(dp_gemm_strided(probs.data_ptr<float>(), t, zb, zh, 1, do0, c, T * C, d, 0, g0 + ov, c3, sq, sh, t, dd, t, b, nh, 1.0f, s));           // dV = P^T dO
    DP_CALL(dp_gemm_strided(do0, c, T * C, d, 0, q0 + ov, c3, sq, sh, 1, dp.data_ptr<float>(), t, zb, zh, t, t, dd, b, nh, 1.0f, s));               // dP = dO V^T
    DP_CALL(dp_softmax_bwd_rows(probs.data_ptr<float>(), dp.data_ptr<float>(), B * n_heads * T, t, s));
    DP_CALL(dp_gemm_strided(dp.data_ptr<float>(), t, zb, zh, 0, q0 + ok, c3, sq, sh, 0, g0 + oq, c3, sq, sh, t, dd, t, b, nh, sc, s));              // dQ = dS K / sqrt(d)
    DP_CALL(dp_gemm_strided(dp.data_ptr<float>(), t, zb, zh, 1, q0 + oq, c3, sq, sh, 0, g0 + ok, c3, sq, sh, t, dd, t, b, nh, sc, s));              // dK = dS^T Q / sqrt(d)
    return dqkv;
}
Tensor resize_affine_bwd(const Tensor& dy, int64_t hi, int64_t wi, double scale, bool in_nhwc, bool out_nhwc) {
    chk_f32(dy, "dy", 4);
    c10::DeviceGuard guard(dy.device());
    const int64_t B = dy.size(0), C = out_nhwc ? dy.size(3) : dy.size(1), Ho = out_nhwc ? dy.size(1) : dy.size(2), Wo = out_nhwc ? dy.size(2) : dy.size(3);
    Tensor dx = in_nhwc ? at::empty({B, hi, wi, C}, dy.options()) : at::empty({B, C, hi, wi}, dy.options());
    DP_CALL(dp_resize_affine_bwd(dy.data_ptr<float>(), (int)B, (int)C, (int)Ho, (int)Wo, out_nhwc ? 1 : 0, (float)scale, dx.data_ptr<float>(), (int)hi,
                                 (int)wi, in_nhwc ? 1 : 0, cur_stream(dy)));
    return dx;
}
// conv2d_nhwc_dgrad: input gradient of conv2d_nhwc - the SAME kernel on the flipped / transposed weight panel (built here from the
// forward panel with ATen views; the engines pack it once at load: ops.dgrad_weight).
Tensor conv2d_nhwc_dgrad(const Tensor& dy, const Tensor& wp, int64_t c_in, int64_t ksize) {
    chk_f32(dy, "dy", 4);
    chk_f32(wp, "wp", 2);
    const int64_t n_out = dy.size(3);
    TORCH_CHECK(wp.size(0) == ksize * ksize * c_in && wp.size(1) >= n_out, "diffpure_hip: weight panel does not match");
    // wp[(ky*K+kx)*I + i][o]  ->  wd[((K-1-ky)*K + (K-1-kx))*O + o][i], columns padded to a multiple of 4
    Tensor w4 = wp.narrow(1, 0, n_out).view({ksize, ksize, c_in, n_out}).flip({0, 1}).permute({0, 1, 3, 2}).reshape({ksize * ksize * n_out, c_in});
    const int64_t ld = (c_in + 3) / 4 * 4;
    Tensor wd = at::zeros({ksize * ksize * n_out, ld}, wp.options());
    wd.narrow(1, 0, c_in).copy_(w4);
    return conv2d_nhwc(dy, wd, c10::nullopt, c_in, ksize);
}

// ---- autograd formulas (Autograd dispatch key): dL/dx of the differentiable-in-x operators, weights are constants -----------------
using torch::autograd::AutogradContext;
using torch::autograd::variable_list;

struct Conv2dNhwcFn : public torch::autograd::Function<Conv2dNhwcFn> {
    static Tensor forward(AutogradContext* ctx, const Tensor& x, const Tensor& wp, const c10::optional<Tensor>& bias, int64_t n_out, int64_t ksize) {
        at::AutoDispatchBelowADInplaceOrView g;
        ctx->save_for_backward({wp});
        ctx->saved_data["c_in"] = x.size(3);
        ctx->saved_data["ksize"] = ksize;
        static auto op = c10::Dispatcher::singleton().findSchemaOrThrow("diffpure_hip::conv2d_nhwc", "").typed<Tensor(const Tensor&, const Tensor&, const c10::optional<Tensor>&, int64_t, int64_t)>();
        return op.call(x, wp, bias, n_out, ksize);
    }
    static variable_list backward(AutogradContext* ctx, variable_list grads) {
        auto saved = ctx->get_saved_variables();
        static auto op = c10::Dispatcher::singleton().findSchemaOrThrow("diffpure_hip::conv2d_nhwc_dgrad", "").typed<Tensor(const Tensor&, const Tensor&, int64_t, int64_t)>();
        Tensor dx = op.call(grads[0].contiguous(), saved[0], ctx->saved_data["c_in"].toInt(), ctx->saved_data["ksize"].toInt());
        return {dx, Tensor(), Tensor(), Tensor(), Tensor()};
    }
};
Tensor conv2d_nhwc_autograd(const Tensor& x, const Tensor& wp, const c10::optional<Tensor>& bias, int64_t n_out, int64_t ksize) {
    return Conv2dNhwcFn::apply(x, wp, bias, n_out, ksize);
}

struct GroupNormSiluFn : public torch::autograd::Function<GroupNormSiluFn> {
    static Tensor forward(AutogradContext* ctx, const Tensor& x, const Tensor& gamma, const Tensor& beta, int64_t groups, double eps, bool act, int64_t out_fmt,
                          const c10::optional<Tensor>& stats_in) {
        at::AutoDispatchBelowADInplaceOrView g;
        static auto st_op = c10::Dispatcher::singleton().findSchemaOrThrow("diffpure_hip::group_norm_stats", "").typed<Tensor(const Tensor&, int64_t, double)>();
        static auto op = c10::Dispatcher::singleton().findSchemaOrThrow("diffpure_hip::group_norm_silu", "")
                             .typed<Tensor(const Tensor&, const Tensor&, const Tensor&, int64_t, double, bool, int64_t, const c10::optional<Tensor>&)>();
        Tensor stats = (stats_in.has_value() && stats_in->defined()) ? *stats_in : st_op.call(x, groups, eps);
        ctx->save_for_backward({x, stats, gamma, beta});
        ctx->saved_data["groups"] = groups;
        ctx->saved_data["act"] = act;
        ctx->saved_data["out_fmt"] = out_fmt;
        return op.call(x, gamma, beta, groups, eps, act, out_fmt, stats);
    }
    static variable_list backward(AutogradContext* ctx, variable_list grads) {
        TORCH_CHECK(ctx->saved_data["out_fmt"].toInt() == 0, "diffpure_hip: group_norm_silu is differentiable in its fp32 output form (out_fmt 0)");
        auto sv = ctx->get_saved_variables();
        static auto op = c10::Dispatcher::singleton().findSchemaOrThrow("diffpure_hip::group_norm_silu_bwd", "")
                             .typed<Tensor(const Tensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor&, int64_t, bool)>();
        Tensor dx = op.call(sv[0], sv[1], sv[2], sv[3], grads[0].contiguous(), ctx->saved_data["groups"].toInt(), ctx->saved_data["act"].toBool());
        return {dx, Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor()};
    }
};
Tensor group_norm_silu_autograd(const Tensor& x, const Tensor& gamma, const Tensor& beta, int64_t groups, double eps, bool act, int64_t out_fmt,
                                const c10::optional<Tensor>& stats_in) {
    return GroupNormSiluFn::apply(x, gamma, beta, groups, eps, act, out_fmt, stats_in);
}

struct AttentionFn : public torch::autograd::Function<AttentionFn> {
    static Tensor forward(AutogradContext* ctx, const Tensor& qkv, int64_t n_heads, bool legacy_layout) {
        at::AutoDispatchBelowADInplaceOrView g;
        ctx->save_for_backward({qkv});
        ctx->saved_data["n_heads"] = n_heads;
        ctx->saved_data["legacy"] = legacy_layout;
        static auto op = c10::Dispatcher::singleton().findSchemaOrThrow("diffpure_hip::attention", "").typed<Tensor(const Tensor&, int64_t, bool)>();
        return op.call(qkv, n_heads, legacy_layout);
    }
    static variable_list backward(AutogradContext* ctx, variable_list grads) {
        auto sv = ctx->get_saved_variables();
        static auto op = c10::Dispatcher::singleton().findSchemaOrThrow("diffpure_hip::attention_bwd", "").typed<Tensor(const Tensor&, const Tensor&, int64_t, bool)>();
        return {op.call(sv[0], grads[0].contiguous(), ctx->saved_data["n_heads"].toInt(), ctx->saved_data["legacy"].toBool()), Tensor(), Tensor()};
    }
};
Tensor attention_autograd(const Tensor& qkv, int64_t n_heads, bool legacy_layout) { return AttentionFn::apply(qkv, n_heads, legacy_layout); }

struct ResizeAffineFn : public torch::autograd::Function<ResizeAffineFn> {
    static Tensor forward(AutogradContext* ctx, const Tensor& x, int64_t ho, int64_t wo, double shift, double scale, bool in_nhwc, bool out_nhwc) {
        at::AutoDispatchBelowADInplaceOrView g;
        ctx->saved_data["hi"] = in_nhwc ? x.size(1) : x.size(2);
        ctx->saved_data["wi"] = in_nhwc ? x.size(2) : x.size(3);
        ctx->saved_data["scale"] = scale;
        ctx->saved_data["in_nhwc"] = in_nhwc;
        ctx->saved_data["out_nhwc"] = out_nhwc;
        static auto op = c10::Dispatcher::singleton().findSchemaOrThrow("diffpure_hip::resize_affine", "").typed<Tensor(const Tensor&, int64_t, int64_t, double, double, bool, bool)>();
        return op.call(x, ho, wo, shift, scale, in_nhwc, out_nhwc);
    }
    static variable_list backward(AutogradContext* ctx, variable_list grads) {
        static auto op = c10::Dispatcher::singleton().findSchemaOrThrow("diffpure_hip::resize_affine_bwd", "").typed<Tensor(const Tensor&, int64_t, int64_t, double, bool, bool)>();
        Tensor dx = op.call(grads[0].contiguous(), ctx->saved_data["hi"].toInt(), ctx->saved_data["wi"].toInt(), ctx->saved_data["scale"].toDouble(),
                            ctx->saved_data["in_nhwc"].toBool(), ctx->saved_data["out_nhwc"].toBool());
        return {dx, Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor()};
    }
};
Tensor resize_affine_autograd(const Tensor& x, int64_t ho, int64_t wo, double shift, double scale, bool in_nhwc, bool out_nhwc) {
    return ResizeAffineFn::apply(x, ho, wo, shift, scale, in_nhwc, out_nhwc);
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
    // ABI 6 / 7 (rounds 4-5): the fp16 residual stream
    m.def("conv2d_h2_ex(Tensor xh, Tensor wh, Tensor? bias, Tensor? temb, Tensor? res, Tensor? seg1, Tensor? seg2, int n_out, int ksize, int passes=0, "
          "int w_fmt=0, float scale=1.0, bool out_f16=False, bool want_stats=False) -> (Tensor, Tensor)");
    m.def("gn_apply_h16(Tensor x, Tensor? x2, Tensor? stats, Tensor? gamma, Tensor? beta, Tensor? film_scale, Tensor? film_shift, int groups, bool act, "
          "int resample=0, int out_fmt=2, bool raw=False) -> (Tensor, Tensor)");
    m.def("attention_fused(Tensor qkv, int n_heads, bool legacy_layout, int operand_w=0) -> Tensor");
    m.def("round_weights(Tensor master, Tensor(a!) work, bool stochastic, int seed, int key) -> ()");
    // ABI 8 (round 6)
    m.def("conv2d_stem(Tensor x, Tensor w16, Tensor? bias, bool out_f16=False, bool want_stats=False) -> (Tensor, Tensor)");
    // backward entry points (dL/dx; weights are constants on this path)
    m.def("group_norm_stats(Tensor x, int groups, float eps) -> Tensor");
    m.def("group_norm_silu_bwd(Tensor x, Tensor stats, Tensor gamma, Tensor beta, Tensor dy, int groups, bool act) -> Tensor");
    m.def("attention_bwd(Tensor qkv, Tensor dout, int n_heads, bool legacy_layout) -> Tensor");
    m.def("resize_affine_bwd(Tensor dy, int hi, int wi, float scale, bool in_nhwc, bool out_nhwc) -> Tensor");
    m.def("conv2d_nhwc_dgrad(Tensor dy, Tensor wp, int c_in, int ksize) -> Tensor");
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
    m.impl("conv2d_h2_ex", conv2d_h2_ex);
    m.impl("gn_apply_h16", gn_apply_h16);
    m.impl("attention_fused", attention_fused);
    m.impl("round_weights", round_weights);
    m.impl("conv2d_stem", conv2d_stem);
    m.impl("group_norm_stats", group_norm_stats);
    m.impl("group_norm_silu_bwd", group_norm_silu_bwd);
    m.impl("attention_bwd", attention_bwd);
    m.impl("resize_affine_bwd", resize_affine_bwd);
    m.impl("conv2d_nhwc_dgrad", conv2d_nhwc_dgrad);
}

// torch.autograd through the operators: conv2d_nhwc, group_norm_silu (fp32 output form), attention and resize_affine carry dL/dx
// formulas built from the backward entry points above (what SDE_Adv_Model's adaptive attacks need from a drop-in operator,
// eval_sde_adv.py:74-89 + guided_diffusion/unet.py:196-234, 295-302); every other operator is inference-only.
TORCH_LIBRARY_IMPL(diffpure_hip, Autograd, m) {
    m.impl("conv2d_nhwc", conv2d_nhwc_autograd);
    m.impl("group_norm_silu", group_norm_silu_autograd);
    m.impl("attention", attention_autograd);
    m.impl("resize_affine", resize_affine_autograd);
}
