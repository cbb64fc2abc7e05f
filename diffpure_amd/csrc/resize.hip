// The two steps either side of the purifier inside SDE_Adv_Model.forward
// (/root/reference/eval_sde_adv.py:73-89), each as ONE pass over the image:
//   before:  x in [0,1], NCHW, classifier resolution --F.interpolate(bilinear, align_corners=False)-->
//            diffusion resolution --(x - 0.5) * 2--> NHWC state of the SDE loop
//   after :  purified NHWC state --F.interpolate--> classifier resolution --(x + 1) * 0.5--> NCHW
// i.e.  y = (bilinear(x) + shift) * scale  with a free choice of input / output layout, and its adjoint
// (dL/dx from dL/dy) for the adaptive-attack gradient path.  HBM-bound: 4 B read + 4 B written per
// element (3-channel images: 0.8 MB per 256^2 sample), negligible next to the UNet calls, but the
// torch version is three kernels and two extra round trips (interpolate, affine, permute).
//
// Source-index rule = ATen's area_pixel_compute_source_index for align_corners=False, non-cubic:
//   src = (in / out) * (dst + 0.5) - 0.5, clamped below at 0;  i0 = (int)src, i1 = i0 + (i0 < in - 1),
//   weights (1 - frac, frac);  value = h0*(w0*v00 + w1*v01) + h1*(w0*v10 + w1*v11).
#include "dp_common.h"

namespace {

struct ResizeArgs {
    const float* x;
    float* y;
    int B, C, Hi, Wi, Ho, Wo;
    int in_nhwc, out_nhwc;
    float shift, scale;
    float rh, rw;       // Hi / Ho, Wi / Wo
};

struct Tap {
    int i0, i1;
    float w0, w1;
};

__device__ __forceinline__ Tap tap_of(int dst, float r, int n_in) {
    float src = r * ((float)dst + 0.5f) - 0.5f;
    if (src < 0.f) src = 0.f;
    Tap t;
    t.i0 = (int)src;
    if (t.i0 > n_in - 1) t.i0 = n_in - 1;       // cannot happen for r = in/out; guards rounding
    t.i1 = t.i0 + (t.i0 < n_in - 1 ? 1 : 0);
    t.w1 = src - (float)t.i0;
    t.w0 = 1.f - t.w1;
    return t;
}

__device__ __forceinline__ size_t at(const ResizeArgs& p, bool nhwc, int H, int W, int b, int c, int y, int x) {
    return nhwc ? (((size_t)b * H + y) * W + x) * p.C + c : (((size_t)b * p.C + c) * H + y) * W + x;
}

// one thread per output pixel, all channels
__global__ void resize_affine_kernel(ResizeArgs p) {
    const long long total = (long long)p.B * p.Ho * p.Wo;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int ox = (int)(i % p.Wo);
        const long long t = i / p.Wo;
        const int oy = (int)(t % p.Ho), b = (int)(t / p.Ho);
        const Tap ty = tap_of(oy, p.rh, p.Hi), tx = tap_of(ox, p.rw, p.Wi);
        for (int c = 0; c < p.C; ++c) {
            const float v00 = p.x[at(p, p.in_nhwc, p.Hi, p.Wi, b, c, ty.i0, tx.i0)];
            const float v01 = p.x[at(p, p.in_nhwc, p.Hi, p.Wi, b, c, ty.i0, tx.i1)];
            const float v10 = p.x[at(p, p.in_nhwc, p.Hi, p.Wi, b, c, ty.i1, tx.i0)];
            const float v11 = p.x[at(p, p.in_nhwc, p.Hi, p.Wi, b, c, ty.i1, tx.i1)];
            const float v = ty.w0 * (tx.w0 * v00 + tx.w1 * v01) + ty.w1 * (tx.w0 * v10 + tx.w1 * v11);
            p.y[at(p, p.out_nhwc, p.Ho, p.Wo, b, c, oy, ox)] = (v + p.shift) * p.scale;
        }
    }
}

// weight with which output index `dst` reads input index `src_i` (0 when it does not)
__device__ __forceinline__ float pull_weight(int dst, float r, int n_in, int src_i) {
    const Tap t = tap_of(dst, r, n_in);
    return (t.i0 == src_i ? t.w0 : 0.f) + (t.i1 == src_i ? t.w1 : 0.f);
}

// Adjoint in GATHER form (no float atomics: deterministic): one thread per INPUT pixel sums, in a fixed
// order, the output pixels that read it.  x/Hi/Wi/in_nhwc describe dx, y/Ho/Wo/out_nhwc describe dy.
__global__ void resize_affine_bwd_kernel(ResizeArgs p) {
    const long long total = (long long)p.B * p.Hi * p.Wi;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int ix = (int)(i % p.Wi);
        const long long t = i / p.Wi;
        const int iy = (int)(t % p.Hi), b = (int)(t / p.Hi);
        // outputs whose source coordinate lies in (iy - 1, iy + 1): a superset, filtered exactly by pull_weight
        int oy_lo = (int)floorf(((float)iy - 0.5f) / p.rh - 0.5f) - 1, oy_hi = (int)ceilf(((float)iy + 1.5f) / p.rh - 0.5f) + 1;
        int ox_lo = (int)floorf(((float)ix - 0.5f) / p.rw - 0.5f) - 1, ox_hi = (int)ceilf(((float)ix + 1.5f) / p.rw - 0.5f) + 1;
        oy_lo = max(oy_lo, 0); ox_lo = max(ox_lo, 0);
        oy_hi = min(oy_hi, p.Ho - 1); ox_hi = min(ox_hi, p.Wo - 1);
        for (int c = 0; c < p.C; ++c) {
            float acc = 0.f;
            for (int oy = oy_lo; oy <= oy_hi; ++oy) {
                const float wy = pull_weight(oy, p.rh, p.Hi, iy);
                if (wy == 0.f) continue;
                float row = 0.f;
                for (int ox = ox_lo; ox <= ox_hi; ++ox) {
                    const float wx = pull_weight(ox, p.rw, p.Wi, ix);
                    if (wx != 0.f) row += wx * p.y[at(p, p.out_nhwc, p.Ho, p.Wo, b, c, oy, ox)];
                }
                acc += wy * row;
            }
            const_cast<float*>(p.x)[at(p, p.in_nhwc, p.Hi, p.Wi, b, c, iy, ix)] = acc * p.scale;
        }
    }
}

inline unsigned grid_for(long long items) {
    long long g = (items + 255) / 256;
    return (unsigned)(g < 1 ? 1 : (g > 65535 ? 65535 : g));
}

}  // namespace

extern "C" int dp_resize_affine(const float* x, int B, int C, int Hi, int Wi, int in_nhwc, float shift, float scale, float* y,
                                int Ho, int Wo, int out_nhwc, void* stream) {
    DP_REQUIRE(x && y && B > 0 && C > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0, "dp_resize_affine: bad args");
    ResizeArgs p{x, y, B, C, Hi, Wi, Ho, Wo, in_nhwc, out_nhwc, shift, scale, (float)Hi / (float)Ho, (float)Wi / (float)Wo};
    hipLaunchKernelGGL(resize_affine_kernel, dim3(grid_for((long long)B * Ho * Wo)), dim3(256), 0, (hipStream_t)stream, p);
    DP_LAUNCH_CHECK("resize_affine");
    return 0;
}

extern "C" int dp_resize_affine_bwd(const float* dy, int B, int C, int Ho, int Wo, int out_nhwc, float scale, float* dx, int Hi,
                                    int Wi, int in_nhwc, void* stream) {
    DP_REQUIRE(dy && dx && B > 0 && C > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0, "dp_resize_affine_bwd: bad args");
    ResizeArgs p{dx, const_cast<float*>(dy), B, C, Hi, Wi, Ho, Wo, in_nhwc, out_nhwc, 0.f, scale, (float)Hi / (float)Ho,
                 (float)Wi / (float)Wo};
    hipLaunchKernelGGL(resize_affine_bwd_kernel, dim3(grid_for((long long)B * Hi * Wi)), dim3(256), 0, (hipStream_t)stream, p);
    DP_LAUNCH_CHECK("resize_affine_bwd");
    return 0;
}
