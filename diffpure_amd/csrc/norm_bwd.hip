// Backward (input gradient only) of the fused GroupNorm(+FiLM)(+SiLU)(+2x resample) operator, used by
// the adjoint-ODE path: dL/dx is all an adaptive attack needs, parameter gradients are never formed
// (the reference's odeint_adjoint integrates 106.6 M parameter adjoints nobody reads).
//
// forward (csrc/norm.hip):  xh = (x - mean) * rstd ; u = (xh*gamma + beta) * (1 + fs) + fh ;
//                           a = act(u) ; y = resample(a)
// backward:                 da = resample^T(dy) ; du = da * act'(u) ; dxh = du * (1 + fs) * gamma
//                           dx = rstd * ( dxh - mean_g(dxh) - xh * mean_g(dxh * xh) )
// Same three-launch, atomics-free structure as the forward: slab partial sums of (dxh, dxh*xh) per
// (sample, group) in a fixed order, a double-precision combine, then one elementwise pass.
// The adjoint of the resampler is folded into the load of dy: forward nearest-x2 -> sum of the 2x2
// block; forward mean-2x2 -> 0.25 * dy[y/2][x/2]; the FIR resamplers of `fir: True` networks (modes 3 / 4,
// csrc/norm.hip fir_up2 / fir_down2) -> the transposed 4-tap stencils of fir_adjoint below.
#include "dp_common.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

struct BwdArgs {
    const float* x1;
    const float* x2;
    int C1, C2, B, H, W, G;      // H, W: INPUT resolution of the forward operator
    const float* stats;          // [B][G][2] mean, rstd of the forward
    const float* gamma;
    const float* beta;
    const float* fscale;
    const float* fshift;
    int film_stride, act, resample;
    const float* dy;             // [B][Ho][Wo][C]
    int C4, cpg, Ho, Wo;
    // stats pass
    int nsplit, ppb, cpg4;
    float* partial;
    // apply pass
    const float* sums;           // [B][G][2] = (mean_g(dxh), mean_g(dxh*xh))
    float* dx1;
    float* dx2;
    int out_fmt;
    float fir[4];                // resample 3 / 4: the forward's 1-D taps k[0..3]
};

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

// Transposes of the FIR x2 resamplers (forward per axis, zero outside the image - csrc/norm.hip):
//   up   : out[2i] = 2 (k3 x[i-1] + k1 x[i]),  out[2i+1] = 2 (k2 x[i] + k0 x[i+1])
//          => dx[i] = 2 (k0 dy[2i-1] + k1 dy[2i] + k2 dy[2i+1] + k3 dy[2i+2])                   (4 x 4 taps in 2-D)
//   down : out[i] = sum_j k[3-j] x[2i+j-1]
//          => dx[2n] = k2 dy[n] + k0 dy[n-1],  dx[2n+1] = k1 dy[n] + k3 dy[n+1]                 (2 x 2 taps in 2-D)
// for channel quad c of INPUT pixel (y, x) of sample b; dy is [B][Ho][Wo][C], out-of-range dy counts as zero.
__device__ __forceinline__ f32x4 fir_adjoint(const float* dy, int b, int Ho, int Wo, int C, int c, int y, int x, int mode,
                                             const float* k) {
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
    if (mode == 3) {
#pragma unroll
        for (int ty = 0; ty < 4; ++ty) {
            const int oy = 2 * y - 1 + ty;
            if ((unsigned)oy >= (unsigned)Ho) continue;
#pragma unroll
            for (int tx = 0; tx < 4; ++tx) {
                const int ox = 2 * x - 1 + tx;
                if ((unsigned)ox >= (unsigned)Wo) continue;
                const f32x4 s = ld4(dy + (((size_t)b * Ho + oy) * Wo + ox) * C + c);
                const float w = (2.f * k[ty]) * (2.f * k[tx]);
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = fmaf(w, s[j], o[j]);
            }
        }
        return o;
    }
    const int ny = y >> 1, nx = x >> 1;
    const int oy2[2] = {ny, (y & 1) ? ny + 1 : ny - 1}, ox2[2] = {nx, (x & 1) ? nx + 1 : nx - 1};
    const float wy[2] = {(y & 1) ? k[1] : k[2], (y & 1) ? k[3] : k[0]}, wx[2] = {(x & 1) ? k[1] : k[2], (x & 1) ? k[3] : k[0]};
#pragma unroll
    for (int ty = 0; ty < 2; ++ty) {
        if ((unsigned)oy2[ty] >= (unsigned)Ho) continue;
#pragma unroll
        for (int tx = 0; tx < 2; ++tx) {
            if ((unsigned)ox2[tx] >= (unsigned)Wo) continue;
            const f32x4 s = ld4(dy + (((size_t)b * Ho + oy2[ty]) * Wo + ox2[tx]) * C + c);
            const float w = wy[ty] * wx[tx];
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = fmaf(w, s[j], o[j]);
        }
    }
    return o;
}

// da (gradient w.r.t. the pre-resample activation) for channel quad c of input pixel (b, y, x).
// FIR: the kernels are instantiated with and without the FIR stencils - compiled into the common instantiation, the sixteen-tap
// stencil (inlined twice into gn_bwd_apply_kernel) cost EVERY network 196 bytes of scratch per lane and doubled that kernel's time
template <bool FIR>
__device__ __forceinline__ f32x4 load_da(const BwdArgs& p, int b, int y, int x, int c) {
    const int C = p.C4 * 4;
    if (p.resample == 0) return ld4(p.dy + (((size_t)b * p.Ho + y) * p.Wo + x) * C + c);
    if (p.resample == 1) {  // forward was nearest x2
        const float* r0 = p.dy + (((size_t)b * p.Ho + 2 * y) * p.Wo + 2 * x) * C + c;
        const f32x4 a = ld4(r0), b2 = ld4(r0 + C), c2 = ld4(r0 + (size_t)p.Wo * C), d = ld4(r0 + (size_t)p.Wo * C + C);
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (a[j] + b2[j]) + (c2[j] + d[j]);
        return o;
    }
    if constexpr (FIR) return fir_adjoint(p.dy, b, p.Ho, p.Wo, C, c, y, x, p.resample, p.fir);
    f32x4 v = ld4(p.dy + (((size_t)b * p.Ho + (y >> 1)) * p.Wo + (x >> 1)) * C + c);  // forward was mean 2x2
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] *= 0.25f;
    return v;
}

// returns dxh and xh for one channel quad of one input pixel
template <bool FIR>
__device__ __forceinline__ void quad_grad(const BwdArgs& p, int b, size_t pix, int y, int x, int c, f32x4& dxh, f32x4& xh) {
    const f32x4 xv = (c < p.C1) ? ld4(p.x1 + pix * p.C1 + c) : ld4(p.x2 + pix * p.C2 + (c - p.C1));
    const int g = c / p.cpg;
    const float mean = p.stats[(b * p.G + g) * 2], rstd = p.stats[(b * p.G + g) * 2 + 1];
    const f32x4 ga = ld4(p.gamma + c), be = ld4(p.beta + c);
    f32x4 m = {1.f, 1.f, 1.f, 1.f}, fh = {0.f, 0.f, 0.f, 0.f};
    if (p.fscale) {
        const f32x4 fs = ld4(p.fscale + (size_t)b * p.film_stride + c);
        fh = ld4(p.fshift + (size_t)b * p.film_stride + c);
#pragma unroll
        for (int j = 0; j < 4; ++j) m[j] = 1.f + fs[j];
    }
    const f32x4 da = load_da<FIR>(p, b, y, x, c);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        xh[j] = (xv[j] - mean) * rstd;
        float du = da[j];
        if (p.act) {
            const float u = (xh[j] * ga[j] + be[j]) * m[j] + fh[j];
            const float sg = dp_sigmoid_f(u);
            du *= sg * (1.f + u * (1.f - sg));
        }
        dxh[j] = du * m[j] * ga[j];
    }
}

template <bool FIR>
__global__ void gn_bwd_stats_kernel(BwdArgs p) {
    __shared__ float red_s[1024];
    __shared__ float red_q[1024];
    const int t = threadIdx.x;
    const int b = blockIdx.x / p.nsplit, sp = blockIdx.x - b * p.nsplit;
    const int HW = p.H * p.W;
    const int per = (HW + p.nsplit - 1) / p.nsplit;
    const int p0 = sp * per, p1 = min(HW, p0 + per);
    const int pl = t / p.C4, cq = t - pl * p.C4;
    float s = 0.f, q = 0.f;
    for (int px = p0 + pl; px < p1; px += p.ppb) {
        const int y = px / p.W, x = px - y * p.W;
        f32x4 dxh, xh;
        quad_grad<FIR>(p, b, (size_t)b * HW + px, y, x, cq * 4, dxh, xh);
        s += (dxh[0] + dxh[1]) + (dxh[2] + dxh[3]);
        q += (dxh[0] * xh[0] + dxh[1] * xh[1]) + (dxh[2] * xh[2] + dxh[3] * xh[3]);
    }
    red_s[t] = s;
    red_q[t] = q;
    __syncthreads();
    if (t < p.G) {
        double ds = 0.0, dq = 0.0;
        for (int l = 0; l < p.ppb; ++l)
            for (int k = 0; k < p.cpg4; ++k) {
                const int idx = l * p.C4 + t * p.cpg4 + k;
                ds += red_s[idx];
                dq += red_q[idx];
            }
        float* dst = p.partial + ((size_t)(b * p.nsplit + sp) * p.G + t) * 2;
        dst[0] = (float)ds;
        dst[1] = (float)dq;
    }
}

__global__ void reduce_partials_kernel(const float* partial, int B, int nsplit, int G, double scale, float* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * G) return;
    const int b = i / G, g = i - b * G;
    double s = 0.0, q = 0.0;
    for (int sp = 0; sp < nsplit; ++sp) {
        const float* src = partial + ((size_t)(b * nsplit + sp) * G + g) * 2;
        s += src[0];
        q += src[1];
    }
    out[i * 2] = (float)(s * scale);
    out[i * 2 + 1] = (float)(q * scale);
}

// one work item = 8 channels of one input pixel (so that the h2 output form is one 32-byte block)
template <bool FIR>
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(BwdArgs p) {
    const int C = p.C4 * 4, C8 = C / 8;
    const int BORDER = p.out_fmt ? 1 : 0;
    const int Hq = p.H + 2 * BORDER, Wq = p.W + 2 * BORDER;
    const long long total = (long long)p.B * Hq * Wq * C8;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c8 = (int)(i % C8);
        const long long qpix = i / C8;
        const int qx = (int)(qpix % Wq);
        const long long t2 = qpix / Wq;
        const int qy = (int)(t2 % Hq), b = (int)(t2 / Hq);
        const int x = qx - BORDER, y = qy - BORDER;
        if (p.out_fmt && ((unsigned)x >= (unsigned)p.W || (unsigned)y >= (unsigned)p.H)) {
            half8 z;
#pragma unroll
            for (int j = 0; j < 8; ++j) z[j] = (_Float16)0.f;
            if (p.out_fmt == 2) {
                *reinterpret_cast<half8*>(reinterpret_cast<char*>(p.dx1) + ((size_t)qpix * C8 + c8) * 16) = z;
                continue;
            }
            half8* dst = reinterpret_cast<half8*>(reinterpret_cast<char*>(p.dx1) + ((size_t)qpix * C8 + c8) * 32);
            dst[0] = z;
            dst[1] = z;
            continue;
        }
        const size_t pix = ((size_t)b * p.H + y) * p.W + x;
        f32x4 o[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int c = c8 * 8 + h * 4;
            f32x4 dxh, xh;
            quad_grad<FIR>(p, b, pix, y, x, c, dxh, xh);
            const int g = c / p.cpg;
            const float m1 = p.sums[(b * p.G + g) * 2], m2 = p.sums[(b * p.G + g) * 2 + 1];
            const float rstd = p.stats[(b * p.G + g) * 2 + 1];
#pragma unroll
            for (int j = 0; j < 4; ++j) o[h][j] = rstd * (dxh[j] - m1 - xh[j] * m2);
        }
        if (p.out_fmt == 2) {       // plain fp16 operand ("h1") of a one-pass fp16 x fp16 dgrad convolution
            half8 hv;
#pragma unroll
            for (int j = 0; j < 8; ++j) hv[j] = (_Float16)o[j >> 2][j & 3];
            *reinterpret_cast<half8*>(reinterpret_cast<char*>(p.dx1) + ((size_t)qpix * C8 + c8) * 16) = hv;
        } else if (p.out_fmt) {
            half8 hi, lo;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float v = o[j >> 2][j & 3];
                hi[j] = (_Float16)v;
                lo[j] = (_Float16)(v - (float)hi[j]);
            }
            half8* dst = reinterpret_cast<half8*>(reinterpret_cast<char*>(p.dx1) + ((size_t)qpix * C8 + c8) * 32);
            dst[0] = hi;
            dst[1] = lo;
        } else {
            const int c = c8 * 8;
            float* d = (c < p.C1) ? p.dx1 + pix * p.C1 + c : p.dx2 + pix * p.C2 + (c - p.C1);
            *reinterpret_cast<f32x4*>(d) = o[0];
            *reinterpret_cast<f32x4*>(d + 4) = o[1];
        }
    }
}

struct Fir4 { float k[4]; };

// adjoint of the plain 2x resamplers (x-branch of a resampling ResBlock)
__global__ void resample_bwd_kernel(const float* dy, int B, int Ho, int Wo, int C4, int mode, Fir4 fir, float* dx) {
    const bool up = mode == 1 || mode == 3;
    const int H = up ? Ho / 2 : Ho * 2, W = up ? Wo / 2 : Wo * 2;
    const long long total = (long long)B * H * W * C4;
    const int C = C4 * 4;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int cq = (int)(i % C4);
        const long long pix = i / C4;
        const int x = (int)(pix % W);
        const long long t2 = pix / W;
        const int y = (int)(t2 % H), b = (int)(t2 / H);
        f32x4 o;
        if (mode == 1) {
            const float* r0 = dy + (((size_t)b * Ho + 2 * y) * Wo + 2 * x) * C + cq * 4;
            const f32x4 a = ld4(r0), b2 = ld4(r0 + C), c2 = ld4(r0 + (size_t)Wo * C), d = ld4(r0 + (size_t)Wo * C + C);
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (a[j] + b2[j]) + (c2[j] + d[j]);
        } else if (mode >= 3) {
            o = fir_adjoint(dy, b, Ho, Wo, C, cq * 4, y, x, mode, fir.k);
        } else {
            o = ld4(dy + (((size_t)b * Ho + (y >> 1)) * Wo + (x >> 1)) * C + cq * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] *= 0.25f;
        }
        *reinterpret_cast<f32x4*>(dx + (size_t)pix * C + cq * 4) = o;
    }
}

// dS = P * (dP - sum_j dP_j P_j) per row, in place on dP; one wave per row
__global__ void softmax_bwd_rows_kernel(const float* pm, float* dp, long long rows, int cols) {
    const int lane = threadIdx.x & 63;
    const long long row = blockIdx.x * 4ll + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* pr = pm + row * cols;
    float* dr = dp + row * cols;
    float s = 0.f;
    for (int c = lane; c < cols; c += 64) s += pr[c] * dr[c];
    s = wave_sum(s);
    for (int c = lane; c < cols; c += 64) dr[c] = pr[c] * (dr[c] - s);
}

__global__ void add_kernel(const float* a, const float* b, float* out, long long n4) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const f32x4 x = ld4(a + i * 4), y = ld4(b + i * 4);
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = x[j] + y[j];
        *reinterpret_cast<f32x4*>(out + i * 4) = o;
    }
}

inline unsigned grid_cap(long long items, int block, int cap) {
    long long g = (items + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (unsigned)g;
}

int fill_common(BwdArgs& p, const char* fn, const float* x1, int C1, const float* x2, int C2, int B, int H, int W, int G,
                const float* stats, const float* gamma, const float* beta, const float* fscale, const float* fshift,
                int film_stride, int act, int resample, const float* fir4, const float* dy) {
    const int C = C1 + C2;
    DP_REQUIRE(x1 && stats && gamma && beta && dy && B > 0 && H > 0 && W > 0 && G > 0, "%s: bad args", fn);
    DP_REQUIRE(C2 == 0 || x2, "%s: x2 missing", fn);
    DP_REQUIRE(C % (4 * G) == 0 && C1 % 8 == 0 && C % 8 == 0, "%s: need C %% (4*G) == 0 and C1, C %% 8 == 0 (C=%d+%d, G=%d)", fn, C1, C2, G);
    DP_REQUIRE((fscale == nullptr) == (fshift == nullptr), "%s: FiLM scale and shift come together", fn);
    DP_REQUIRE(resample >= 0 && resample <= 4, "%s: resample mode %d", fn, resample);
    DP_REQUIRE((resample != 2 && resample != 4) || (H % 2 == 0 && W % 2 == 0), "%s: 2x down-sampling needs even H, W", fn);
    DP_REQUIRE(resample < 3 || fir4, "%s: the FIR resampling modes (3, 4) need the 4 filter taps", fn);
    DP_REQUIRE(dp_aligned16(x1) && (C2 == 0 || dp_aligned16(x2)) && dp_aligned16(dy), "%s: misaligned tensor", fn);
    p = BwdArgs{};
    p.x1 = x1; p.x2 = x2; p.C1 = C1; p.C2 = C2; p.B = B; p.H = H; p.W = W; p.G = G;
    p.stats = stats; p.gamma = gamma; p.beta = beta; p.fscale = fscale; p.fshift = fshift;
    p.film_stride = film_stride; p.act = act; p.resample = resample; p.dy = dy;
    p.C4 = C / 4; p.cpg = C / G;
    const bool up = resample == 1 || resample == 3, down = resample == 2 || resample == 4;
    p.Ho = up ? 2 * H : (down ? H / 2 : H);
    p.Wo = up ? 2 * W : (down ? W / 2 : W);
    if (resample >= 3)
        for (int i = 0; i < 4; ++i) p.fir[i] = fir4[i];
    return 0;
}

}  // namespace

extern "C" int dp_gn_bwd_stats(const float* x1, int C1, const float* x2, int C2, int B, int H, int W, int G,
                               const float* stats, const float* gamma, const float* beta, const float* fscale,
                               const float* fshift, int film_stride, int act, int resample, const float* fir4, const float* dy,
                               int nsplit, float* partial, float* sums, void* stream) {
    BwdArgs p;
    if (int rc = fill_common(p, "dp_gn_bwd_stats", x1, C1, x2, C2, B, H, W, G, stats, gamma, beta, fscale, fshift,
                             film_stride, act, resample, fir4, dy)) return rc;
    DP_REQUIRE(partial && sums && nsplit > 0, "dp_gn_bwd_stats: scratch missing");
    DP_REQUIRE(p.C4 <= 1024 && G <= p.C4, "dp_gn_bwd_stats: C too wide");
    p.nsplit = nsplit; p.partial = partial; p.cpg4 = p.C4 / G;
    p.ppb = p.C4 >= 256 ? 1 : 256 / p.C4;
    const int block = p.C4 * p.ppb;
    hipStream_t s = (hipStream_t)stream;
    if (resample >= 3) hipLaunchKernelGGL(gn_bwd_stats_kernel<true>, dim3((unsigned)(B * nsplit)), dim3(block), 0, s, p);
    else hipLaunchKernelGGL(gn_bwd_stats_kernel<false>, dim3((unsigned)(B * nsplit)), dim3(block), 0, s, p);
    const double inv = 1.0 / ((double)H * W * p.cpg);
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((B * G + 255) / 256), dim3(256), 0, s, partial, B, nsplit, G, inv, sums);
    DP_LAUNCH_CHECK("gn_bwd_stats");
    return 0;
}

extern "C" int dp_gn_bwd_apply(const float* x1, int C1, const float* x2, int C2, int B, int H, int W, int G,
                               const float* stats, const float* gamma, const float* beta, const float* fscale,
                               const float* fshift, int film_stride, int act, int resample, const float* fir4, const float* dy,
                               const float* sums, int out_fmt, void* dx1, float* dx2, void* stream) {
    BwdArgs p;
    if (int rc = fill_common(p, "dp_gn_bwd_apply", x1, C1, x2, C2, B, H, W, G, stats, gamma, beta, fscale, fshift,
                             film_stride, act, resample, fir4, dy)) return rc;
    DP_REQUIRE(sums && dx1 && (C2 == 0 || dx2), "dp_gn_bwd_apply: output missing");
    DP_REQUIRE(out_fmt == 0 || ((out_fmt == 1 || out_fmt == 2) && C2 == 0), "dp_gn_bwd_apply: operand output (1 = h2, 2 = h1) needs a single source");
    p.sums = sums; p.dx1 = (float*)dx1; p.dx2 = dx2; p.out_fmt = out_fmt;
    const int border = out_fmt ? 1 : 0;
    const long long total = (long long)B * (H + 2 * border) * (W + 2 * border) * (p.C4 / 2);
    if (resample >= 3) hipLaunchKernelGGL(gn_bwd_apply_kernel<true>, dim3(grid_cap(total, 256, 4096)), dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(gn_bwd_apply_kernel<false>, dim3(grid_cap(total, 256, 4096)), dim3(256), 0, (hipStream_t)stream, p);
    DP_LAUNCH_CHECK("gn_bwd_apply");
    return 0;
}

extern "C" int dp_resample_bwd(const float* dy, int B, int Ho, int Wo, int C, int mode, const float* fir4, float* dx, void* stream) {
    DP_REQUIRE(dy && dx && B > 0 && Ho > 0 && Wo > 0 && C % 4 == 0 && mode >= 1 && mode <= 4, "dp_resample_bwd: bad args");
    const bool up = mode == 1 || mode == 3;
    DP_REQUIRE(!up || (Ho % 2 == 0 && Wo % 2 == 0), "dp_resample_bwd: odd output size");
    DP_REQUIRE(mode < 3 || fir4, "dp_resample_bwd: the FIR resampling modes (3, 4) need the 4 filter taps");
    Fir4 fir{};
    if (mode >= 3)
        for (int i = 0; i < 4; ++i) fir.k[i] = fir4[i];
    const long long total = (long long)B * (up ? Ho / 2 : Ho * 2) * (up ? Wo / 2 : Wo * 2) * (C / 4);
    hipLaunchKernelGGL(resample_bwd_kernel, dim3(grid_cap(total, 256, 4096)), dim3(256), 0, (hipStream_t)stream, dy, B, Ho, Wo,
                       C / 4, mode, fir, dx);
    DP_LAUNCH_CHECK("resample_bwd");
    return 0;
}

extern "C" int dp_softmax_bwd_rows(const float* p, float* dp, long long rows, int cols, void* stream) {
    DP_REQUIRE(p && dp && rows > 0 && cols > 0, "dp_softmax_bwd_rows: bad args");
    const long long grid = (rows + 3) / 4;
    DP_REQUIRE(grid < (1ll << 31), "dp_softmax_bwd_rows: too many rows");
    hipLaunchKernelGGL(softmax_bwd_rows_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, p, dp, rows, cols);
    DP_LAUNCH_CHECK("softmax_bwd_rows");
    return 0;
}

extern "C" int dp_add(const float* a, const float* b, float* out, long long n, void* stream) {
    DP_REQUIRE(a && b && out && n > 0 && n % 4 == 0, "dp_add: n must be a positive multiple of 4");
    DP_REQUIRE(dp_aligned16(a) && dp_aligned16(b) && dp_aligned16(out), "dp_add: misaligned");
    hipLaunchKernelGGL(add_kernel, dim3(grid_cap(n / 4, 256, 4096)), dim3(256), 0, (hipStream_t)stream, a, b, out, n / 4);
    DP_LAUNCH_CHECK("add");
    return 0;
}
