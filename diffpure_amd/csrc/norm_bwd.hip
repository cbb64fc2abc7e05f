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
#include "dp_tune.h"

// Round 6: the compiler's own mul + add contraction is OFF in this file and every fused multiply-add is spelled `fmaf` in ONE set of
// helpers (gnb_quad, gnb_out, gnb_add below) that the statistics pass, the generic and the lean apply pass and the one-pass kernel share:
// left to itself the compiler fused `dxh - m1 - xh * m2` (and half a dozen products of the chain) in one kernel and not in its twin, so two
// forms of the same pass differed in the last bit of a third of their elements.
#pragma clang fp contract(off)

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

struct BwdArgs {
    const float* x1;
    const float* x2;
    int x_fmt;                   // 0: x1 / x2 are fp32; 1: plain fp16 (the taped forward of the fp16 x fp16 modes keeps the fp16 residual stream)
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
    // apply pass, fp32 output only (round 5; the one-pass kernel had it since round 4): a second gradient arriving at the same tensors -
    // the skip branch of a ResBlock - is added in the same pass, dx += add_scale * add, instead of by a dp_add / dp_axpby launch
    const float* add1;
    const float* add2;
    float add_scale;
    // round 6 (DP_GNB_NT): non-temporal hints on the streaming accesses of the three-launch form - bit 0 the loads of x / dy / the addend
    // (un-resampled dy only), bit 1 the stores of dx.  Same bits.  Set by tensor size: a gradient map far beyond the 256 MB of last-level
    // cache gains nothing from occupying it between the statistics pass and the apply pass.
    int nt;
};

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 ld4s(const float* p, int nt) {     // a streaming quad: with the non-temporal hint when nt & 1
    return (nt & 1) ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p)) : *reinterpret_cast<const f32x4*>(p);
}
__device__ __forceinline__ void st4s(float* p, f32x4 v, int nt) {
    if (nt & 2) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p));
    else *reinterpret_cast<f32x4*>(p) = v;
}
typedef _Float16 half4q __attribute__((ext_vector_type(4)));
// a channel quad of the forward's INPUT tensor in its stored format (element index e from the tensor's base)
__device__ __forceinline__ f32x4 ld4x(const float* base, size_t e, int x_fmt, int nt = 0) {
    if (x_fmt) {
        const half4q* src = reinterpret_cast<const half4q*>(reinterpret_cast<const _Float16*>(base) + e);
        const half4q h = (nt & 1) ? __builtin_nontemporal_load(src) : *src;
        return f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
    }
    return ld4s(base + e, nt);
}

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
    if (p.resample == 0) return ld4s(p.dy + (((size_t)b * p.Ho + y) * p.Wo + x) * C + c, p.nt);
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

// The arithmetic of the pass, spelled once.  xh = (x - mean) * rstd ; u = fma(fma(xh, gamma, beta), m, fh) ;
// du = da * (sg * fma(u, 1 - sg, 1)) ; dxh = (du * m) * gamma ; dx = rstd * fma(-xh, m2, dxh - m1) ; dx = fma(add_scale, addend, dx)
__device__ __forceinline__ void gnb_quad(bool act, f32x4 xv, f32x4 da, float mean, float rstd, f32x4 ga, f32x4 be, f32x4 m, f32x4 fh, f32x4& dxh,
                                         f32x4& xh) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        xh[j] = (xv[j] - mean) * rstd;
        float du = da[j];
        if (act) {
            const float u = fmaf(fmaf(xh[j], ga[j], be[j]), m[j], fh[j]);
            const float sg = dp_sigmoid_f(u);
            du *= sg * fmaf(u, 1.f - sg, 1.f);
        }
        dxh[j] = (du * m[j]) * ga[j];
    }
}
__device__ __forceinline__ f32x4 gnb_out(f32x4 dxh, f32x4 xh, float rstd, float m1, float m2) {
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = rstd * fmaf(-xh[j], m2, dxh[j] - m1);
    return o;
}
__device__ __forceinline__ f32x4 gnb_add(f32x4 o, float scale, f32x4 a) {
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = fmaf(scale, a[j], o[j]);
    return o;
}

// returns dxh and xh for one channel quad of one input pixel
template <bool FIR>
__device__ __forceinline__ void quad_grad(const BwdArgs& p, int b, size_t pix, int y, int x, int c, f32x4& dxh, f32x4& xh) {
    const f32x4 xv = (c < p.C1) ? ld4x(p.x1, pix * p.C1 + c, p.x_fmt, p.nt) : ld4x(p.x2, pix * p.C2 + (c - p.C1), p.x_fmt, p.nt);
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
    gnb_quad(p.act != 0, xv, da, mean, rstd, ga, be, m, fh, dxh, xh);
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

// LEAN statistics pass (round 6; see the lean apply pass below): no resampling, plain fp16 tape.  Same thread geometry, same per-thread
// accumulation order and the same reduction as gn_bwd_stats_kernel - hence the same partial sums bit for bit - but the quad's constants
// (statistics, gamma, beta, FiLM) are loaded ONCE, nothing divides inside the loop, and two pixels are in flight per iteration.
template <bool ACT>
__global__ void gn_bwd_stats_lean_kernel(BwdArgs p) {
    __shared__ float red_s[1024];
    __shared__ float red_q[1024];
    const int t = threadIdx.x;
    const int b = blockIdx.x / p.nsplit, sp = blockIdx.x - b * p.nsplit;
    const int HW = p.H * p.W;
    const int per = (HW + p.nsplit - 1) / p.nsplit;
    const int p0 = sp * per, p1 = min(HW, p0 + per);
    const int pl = t / p.C4, cq = t - pl * p.C4;
    const int c = cq * 4, C = p.C4 * 4;
    const bool first = c < p.C1;
    const int Cs = first ? p.C1 : p.C2, cs = first ? c : c - p.C1;
    const _Float16* xs = reinterpret_cast<const _Float16*>(first ? p.x1 : p.x2) + (size_t)b * HW * Cs + cs;
    const float* dys = p.dy + (size_t)b * HW * C + c;
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
    auto ldx = [&](int px) {
        const half4q* src = reinterpret_cast<const half4q*>(xs + (size_t)px * Cs);
        const half4q h = (p.nt & 1) ? __builtin_nontemporal_load(src) : *src;
        return f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
    };
    float s = 0.f, q = 0.f;
    auto acc = [&](f32x4 xv, f32x4 da) {
        f32x4 dxh, xh;
        gnb_quad(ACT, xv, da, mean, rstd, ga, be, m, fh, dxh, xh);
        s += (dxh[0] + dxh[1]) + (dxh[2] + dxh[3]);
        q += (dxh[0] * xh[0] + dxh[1] * xh[1]) + (dxh[2] * xh[2] + dxh[3] * xh[3]);
    };
    int px = p0 + pl;
    for (; px + p.ppb < p1; px += 2 * p.ppb) {              // two pixels in flight, accumulated in pixel order
        const f32x4 xa = ldx(px), xb = ldx(px + p.ppb);
        const f32x4 da = ld4s(dys + (size_t)px * C, p.nt), db = ld4s(dys + (size_t)(px + p.ppb) * C, p.nt);
        acc(xa, da);
        acc(xb, db);
    }
    if (px < p1) acc(ldx(px), ld4s(dys + (size_t)px * C, p.nt));
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
            o[h] = gnb_out(dxh, xh, rstd, m1, m2);
        }
        if (p.out_fmt == 2) {       // plain fp16 operand ("h1") of a one-pass fp16 x fp16 dgrad convolution
            half8 hv;
#pragma unroll
            for (int j = 0; j < 8; ++j) hv[j] = (_Float16)o[j >> 2][j & 3];
            half8* dsth = reinterpret_cast<half8*>(reinterpret_cast<char*>(p.dx1) + ((size_t)qpix * C8 + c8) * 16);
            if (p.nt & 2) __builtin_nontemporal_store(hv, dsth);
            else *dsth = hv;
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
            const bool first = c < p.C1;
            const size_t e = first ? pix * p.C1 + c : pix * p.C2 + (c - p.C1);
            const float* ad = first ? p.add1 : p.add2;
            if (ad) {
                const f32x4 a0 = ld4s(ad + e, p.nt), a1 = ld4s(ad + e + 4, p.nt);
                o[0] = gnb_add(o[0], p.add_scale, a0);
                o[1] = gnb_add(o[1], p.add_scale, a1);
            }
            float* d = (first ? p.dx1 : p.dx2) + e;
            st4s(d, o[0], p.nt);
            st4s(d + 4, o[1], p.nt);
        }
    }
}


// ---- LEAN apply pass (round 6) ------------------------------------------------------------------------------------------------------
// The generic apply kernel above spends most of its issue slots OUTSIDE the gradient arithmetic: three 64-bit div / mod pairs per work item
// to find (sample, pixel, channel octet), a re-load of gamma / beta / FiLM / statistics / group sums for every item, and the format
// branches of every operand - ~420 vector instructions per 64 bytes of HBM traffic at 106 registers, i.e. a kernel that is VALU-bound below
// 5 TB/s (the 256^2 maps of the guided UNet's adjoint ran it at 2.8-3.9 TB/s: 11.3 % of the ImageNet adjoint step).  This form serves the hot
// case - no resampling, plain fp16 tape (x_fmt 1), fewer than 2^31 items, a channel-octet count that divides the thread count of the grid -
// with the SAME arithmetic (gnb_quad / gnb_out / gnb_add are shared with it: identical bits, tests/test_gpu_grad.py) and
//   * a thread keeps ONE channel octet for its whole life: gamma / beta once, the per-sample values (mean, rstd, the two group sums, the
//     FiLM rows) re-loaded only when its pixel walk crosses into the next sample;
//   * the pixel walk is incremental in (sample, y, x) - no division in the loop;
//   * OUTF / ADD are template parameters; the zero border of the operand form is written by a short second loop.
// OUTF: 0 = fp32 dx (two sources, optional addends), 2 = the zero-bordered plain-fp16 operand (single source)
template <int OUTF, bool ADD, bool ACT>
__global__ __launch_bounds__(256) void gn_bwd_apply_lean_kernel(BwdArgs p, int C8, int PS, int dB, int dY, int dX) {
    const int gid = blockIdx.x * 256 + threadIdx.x;
    const int c8 = gid % C8, pl = gid / C8;                    // this thread's channel octet and its first pixel (of B * H * W)
    const int c = c8 * 8, C = C8 * 8;
    const int HW = p.H * p.W;
    const int total = p.B * HW;
    const bool first = c < p.C1;
    const int Cs = first ? p.C1 : p.C2, cs = first ? c : c - p.C1;      // source tensor of this octet (the split C1 is a multiple of 8)
    const _Float16* xs = reinterpret_cast<const _Float16*>(first ? p.x1 : p.x2);
    const f32x4 ga0 = ld4(p.gamma + c), ga1 = ld4(p.gamma + c + 4), be0 = ld4(p.beta + c), be1 = ld4(p.beta + c + 4);
    const int g0 = c / p.cpg, g1 = (c + 4) / p.cpg;
    int b = pl / HW, rem = pl - b * HW;
    int y = rem / p.W, x = rem - y * p.W;
    int cur_b = -1;
    float mean0 = 0.f, rstd0 = 0.f, mean1 = 0.f, rstd1 = 0.f, s10 = 0.f, s20 = 0.f, s11 = 0.f, s21 = 0.f;
    f32x4 m0 = {1.f, 1.f, 1.f, 1.f}, m1 = m0, fh0 = {0.f, 0.f, 0.f, 0.f}, fh1 = fh0;
    for (int px = pl; px < total; px += PS) {
        if (b != cur_b) {                                      // crossed into another sample: its statistics, group sums and FiLM rows
            cur_b = b;
            const float* st = p.stats + (size_t)b * p.G * 2;
            const float* su = p.sums + (size_t)b * p.G * 2;
            mean0 = st[g0 * 2]; rstd0 = st[g0 * 2 + 1]; mean1 = st[g1 * 2]; rstd1 = st[g1 * 2 + 1];
            s10 = su[g0 * 2]; s20 = su[g0 * 2 + 1]; s11 = su[g1 * 2]; s21 = su[g1 * 2 + 1];
            if (p.fscale) {
                const f32x4 a0 = ld4(p.fscale + (size_t)b * p.film_stride + c), a1 = ld4(p.fscale + (size_t)b * p.film_stride + c + 4);
                fh0 = ld4(p.fshift + (size_t)b * p.film_stride + c);
                fh1 = ld4(p.fshift + (size_t)b * p.film_stride + c + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    m0[j] = 1.f + a0[j];
                    m1[j] = 1.f + a1[j];
                }
            }
        }
        const size_t pix = (size_t)px;
        const half8* xp = reinterpret_cast<const half8*>(xs + pix * Cs + cs);
        const half8 xh8 = (p.nt & 1) ? __builtin_nontemporal_load(xp) : *xp;
        const float* dyp = p.dy + pix * C + c;
        const f32x4 d0 = ld4s(dyp, p.nt), d1 = ld4s(dyp + 4, p.nt);
        f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
        if (ADD) {
            const float* ad = (first ? p.add1 : p.add2) + pix * Cs + cs;
            a0 = ld4s(ad, p.nt);
            a1 = ld4s(ad + 4, p.nt);
        }
        const f32x4 x0 = {(float)xh8[0], (float)xh8[1], (float)xh8[2], (float)xh8[3]}, x1 = {(float)xh8[4], (float)xh8[5], (float)xh8[6], (float)xh8[7]};
        f32x4 dxh0, xh0, dxh1, xh1, o0, o1;
        gnb_quad(ACT, x0, d0, mean0, rstd0, ga0, be0, m0, fh0, dxh0, xh0);
        gnb_quad(ACT, x1, d1, mean1, rstd1, ga1, be1, m1, fh1, dxh1, xh1);
        o0 = gnb_out(dxh0, xh0, rstd0, s10, s20);
        o1 = gnb_out(dxh1, xh1, rstd1, s11, s21);
        if (OUTF == 2) {
            half8 hv;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                hv[j] = (_Float16)o0[j];
                hv[4 + j] = (_Float16)o1[j];
            }
            const size_t qpix = ((size_t)b * (p.H + 2) + y + 1) * (p.W + 2) + x + 1;
            half8* dst = reinterpret_cast<half8*>(reinterpret_cast<char*>(p.dx1) + (qpix * C8 + c8) * 16);
            if (p.nt & 2) __builtin_nontemporal_store(hv, dst);
            else *dst = hv;
        } else {
            if (ADD) {
                o0 = gnb_add(o0, p.add_scale, a0);
                o1 = gnb_add(o1, p.add_scale, a1);
            }
            float* d = (first ? p.dx1 : p.dx2) + pix * Cs + cs;
            st4s(d, o0, p.nt);
            st4s(d + 4, o1, p.nt);
        }
        // the next pixel of this thread: px + PS = (dB samples, dY rows, dX columns) further, carried without a division
        x += dX;
        if (x >= p.W) { x -= p.W; ++y; }
        y += dY;
        if (y >= p.H) { y -= p.H; ++b; }
        b += dB;
    }
    if (OUTF == 2) {                                           // zero border of the operand: frame pixels of every sample, this grid's octets
        const int Wq = p.W + 2, Hq = p.H + 2;
        const int nb = 2 * Wq + 2 * p.H;
        half8 z;
#pragma unroll
        for (int j = 0; j < 8; ++j) z[j] = (_Float16)0.f;
        for (int f = pl; f < p.B * nb; f += PS) {
            const int bb = f / nb, r = f - bb * nb;
            int yy, xx;
            if (r < Wq) { yy = 0; xx = r; }
            else if (r < 2 * Wq) { yy = Hq - 1; xx = r - Wq; }
            else { const int k = r - 2 * Wq; yy = 1 + (k >> 1); xx = (k & 1) ? Wq - 1 : 0; }
            const size_t qpix = ((size_t)bb * Hq + yy) * Wq + xx;
            *reinterpret_cast<half8*>(reinterpret_cast<char*>(p.dx1) + (qpix * C8 + c8) * 16) = z;
        }
    }
}


// ---- ONE-PASS form for small feature maps (round 4; CIFAR-10 sizes) ---------------------------------------------------------
// The three launches above read x and dy twice (statistics pass, apply pass).  When all pixels of a block of whole groups of one
// sample fit the registers of a workgroup - HW x CB <= 512 threads x ITEMS quads - ONE workgroup per (sample, channel block) reads
// them ONCE: every thread keeps (dxh, xh) of its ITEMS <= 8 channel quads (one quad column, pixels pl, pl + PL, ...), the group sums go
// through LDS in a fixed order (double, as reduce_partials), and dx is formed from the registers.  The same values up to the
// summation order of the two group sums (which is again a function of the sample's shape only: results do not depend on the batch).
// `add1` / `add2` (optional, fp32 output only): a second gradient that arrives at the same tensor - the identity / 1x1 skip branch
// of a ResBlock - is added in the same pass instead of an `add` launch of its own.
struct FusedArgs {
    BwdArgs b;
    const float* add1;
    const float* add2;
    float add_scale;            // dx += add_scale * add
    int CB, QB, PL;             // channels / quads per block, pixel lanes (512 / QB)
    int xcd_pair;               // DP_XCD_MAP (round 6): neighbouring channel blocks on the same XCD (see the kernel)
    int lean;                   // DP_GNB_LEAN: 1 = constants once + all loads up front (round 6), 0 = rounds 4-5's item-by-item loop (A/B; same bits)
};

template <int ITEMS>
__global__ __launch_bounds__(512) void gn_bwd_fused_kernel(FusedArgs a) {
    const BwdArgs& p = a.b;
    __shared__ float red_s[512];
    __shared__ float red_q[512];
    __shared__ double part[2 * 512];                    // run sums of the second reduction level
    __shared__ float gsum[2 * 64];                      // (m1, m2) of the block's groups (CB / cpg <= 64)
    const int t = threadIdx.x;
    const int nblk = (p.C4 * 4) / a.CB;
    // round 6: workgroups are dealt to the 8 XCDs round-robin by blockIdx, and two NEIGHBOURING channel blocks of a sample share every
    // 128-byte line of dy / addend / dx when a block is 16 channels (64 bytes per pixel): dealt in order they land on different XCDs and
    // each L2 fetches the whole line for half of it.  Inside every group of 16 workgroups, physical r and r + 8 (the same XCD) take
    // logical blocks 2k and 2k + 1.  (speed only; a.xcd_pair = 0: identity)
    int bid = blockIdx.x;
    if (a.xcd_pair && (int)(gridDim.x - (gridDim.x & 15)) > bid) {
        const int r = bid & 15;
        bid = (bid & ~15) + 2 * (r & 7) + (r >> 3);
    }
    // (four neighbours per XCD - the fp16 x lines are shared four ways - measured +0.2 %: noise; pairs: +0.9 % on the CIFAR adjoint)
    const int b = bid / nblk, cb = bid - b * nblk;
    const int HW = p.H * p.W;
    const int q = t % a.QB, pl = t / a.QB;
    const int c = cb * a.CB + q * 4;                    // this thread's channel quad (one group: cpg % 4 == 0)
    f32x4 dxh[ITEMS], xh[ITEMS];
    f32x4 adv[ITEMS];                                   // round 6: the skip branch's gradient, fetched WITH x and dy (lean path) instead of inside the store loop
    const float* adp = nullptr;                         // this thread's addend column (fp32 output only)
    if (p.out_fmt == 0) {
        const bool first_ = c < p.C1;
        const float* ab = first_ ? a.add1 : a.add2;
        if (ab) adp = ab + (first_ ? c : c - p.C1);
    }
    const int adC = c < p.C1 ? p.C1 : p.C2;
    float s = 0.f, sq = 0.f;
    if (!a.lean) {
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) {
            const int px = pl + k * a.PL;
            if (px < HW) {
                const int y = px / p.W, x = px - y * p.W;
                quad_grad<false>(p, b, (size_t)b * HW + px, y, x, c, dxh[k], xh[k]);
                s += (dxh[k][0] + dxh[k][1]) + (dxh[k][2] + dxh[k][3]);
                sq += (dxh[k][0] * xh[k][0] + dxh[k][1] * xh[k][1]) + (dxh[k][2] * xh[k][2] + dxh[k][3] * xh[k][3]);
            }
        }
    } else {   // round 6: the quad's constants are loaded ONCE (a thread keeps its channel quad and its sample), and ALL of the thread's x / dy
        // loads are issued before the first use (rounds 4-5 went through quad_grad per item: constants re-loaded behind every `px < HW`
        // branch, a division per item, one item's loads waited for before the next item's were issued).  Same arithmetic (gnb_quad).
        const bool first = c < p.C1;
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
        f32x4 xv[ITEMS], da[ITEMS];
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) {
            const int px = pl + k * a.PL;
            if (px < HW) {
                const size_t pix = (size_t)b * HW + px;
                xv[k] = first ? ld4x(p.x1, pix * p.C1 + c, p.x_fmt) : ld4x(p.x2, pix * p.C2 + (c - p.C1), p.x_fmt);
                if (p.resample == 0) {
                    da[k] = ld4(p.dy + pix * (size_t)(p.C4 * 4) + c);
                } else {
                    const int y = px / p.W, x = px - y * p.W;
                    da[k] = load_da<false>(p, b, y, x, c);
                }
                if (adp) adv[k] = ld4(adp + pix * adC);
            }
        }
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) {
            const int px = pl + k * a.PL;
            if (px < HW) {
                gnb_quad(p.act != 0, xv[k], da[k], mean, rstd, ga, be, m, fh, dxh[k], xh[k]);
                s += (dxh[k][0] + dxh[k][1]) + (dxh[k][2] + dxh[k][3]);
                sq += (dxh[k][0] * xh[k][0] + dxh[k][1] * xh[k][1]) + (dxh[k][2] * xh[k][2] + dxh[k][3] * xh[k][3]);
            }
        }
    }
    red_s[t] = s;
    red_q[t] = sq;
    __syncthreads();
    // (sum, sum.x) per group in double, in a fixed two-level order: a group's n = 512 / gpb thread partials (entry e = lane-row l *
    // cpg4 + k) are cut into P = min(16, n) runs, one thread per run, then one thread per group adds the P run sums.
    const int cpg4 = p.cpg / 4, gpb = a.CB / p.cpg;     // quads per group, groups per block
    const int n = 512 / gpb, P = n < 16 ? n : 16, run = n / P;
    if (t < gpb * P) {
        const int g = t / P, j = t - g * P;
        double ds = 0.0, dq = 0.0;
        for (int e = j * run; e < (j + 1) * run; ++e) {
            const int l = e / cpg4, k = e - l * cpg4;
            const int idx = l * a.QB + g * cpg4 + k;
            ds += red_s[idx];
            dq += red_q[idx];
        }
        part[2 * t] = ds;
        part[2 * t + 1] = dq;
    }
    __syncthreads();
    if (t < gpb) {
        double ds = 0.0, dq = 0.0;
        for (int j = 0; j < P; ++j) {
            ds += part[2 * (t * P + j)];
            dq += part[2 * (t * P + j) + 1];
        }
        const double inv = 1.0 / ((double)HW * p.cpg);
        gsum[2 * t] = (float)(ds * inv);
        gsum[2 * t + 1] = (float)(dq * inv);
    }
    __syncthreads();
    const int gl = (q * 4) / p.cpg;                     // the quad's group inside the block
    const float m1 = gsum[2 * gl], m2 = gsum[2 * gl + 1];
    const float rstd = p.stats[(b * p.G + c / p.cpg) * 2 + 1];
    const int C = p.C4 * 4;
    const int BORDER = p.out_fmt ? 1 : 0;
    const int Wq = p.W + 2 * BORDER, Hq = p.H + 2 * BORDER;
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        const int px = pl + k * a.PL;
        if (px >= HW) continue;
        f32x4 o = gnb_out(dxh[k], xh[k], rstd, m1, m2);
        const size_t pix = (size_t)b * HW + px;
        if (p.out_fmt == 0) {
            const bool first = c < p.C1;
            float* d = first ? p.dx1 + pix * p.C1 + c : p.dx2 + pix * p.C2 + (c - p.C1);
            const float* ad = first ? (a.add1 ? a.add1 + pix * p.C1 + c : nullptr) : (a.add2 ? a.add2 + pix * p.C2 + (c - p.C1) : nullptr);
            if (ad) {
                o = gnb_add(o, a.add_scale, a.lean ? adv[k] : ld4(ad));
            }
            *reinterpret_cast<f32x4*>(d) = o;
        } else {
            const int y = px / p.W, x = px - y * p.W;
            const size_t qpix = ((size_t)b * Hq + y + 1) * Wq + x + 1;
            typedef _Float16 half4v __attribute__((ext_vector_type(4)));
            if (p.out_fmt == 2) {                       // plain fp16 operand: this quad's 8 bytes
                half4v h;
#pragma unroll
                for (int j = 0; j < 4; ++j) h[j] = (_Float16)o[j];
                *reinterpret_cast<half4v*>(reinterpret_cast<char*>(p.dx1) + (qpix * C + c) * 2) = h;
            } else {                                    // h2: octet = [8 hi | 8 lo]; this quad is half of an octet
                half4v hi, lo;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    hi[j] = (_Float16)o[j];
                    lo[j] = (_Float16)(o[j] - (float)hi[j]);
                }
                char* base = reinterpret_cast<char*>(p.dx1) + (qpix * C + (c & ~7)) * 4 + (c & 4) * 2;
                *reinterpret_cast<half4v*>(base) = hi;
                *reinterpret_cast<half4v*>(base + 16) = lo;
            }
        }
    }
    if (BORDER) {       // the zero border of the operand: this block's channels of the sample's frame pixels
        const int nb = 2 * Wq + 2 * p.H;                // frame pixels
        const int esz = p.out_fmt == 2 ? 2 : 4;         // bytes per channel in the operand
        for (int i = t; i < nb * a.QB; i += 512) {
            const int f = i / a.QB, qq = i - f * a.QB;
            int y, x;
            if (f < Wq) { y = 0; x = f; }
            else if (f < 2 * Wq) { y = Hq - 1; x = f - Wq; }
            else { const int r = f - 2 * Wq; y = 1 + (r >> 1); x = (r & 1) ? Wq - 1 : 0; }
            const size_t qpix = ((size_t)b * Hq + y) * Wq + x;
            const int cc = cb * a.CB + qq * 4;
            char* base = reinterpret_cast<char*>(p.dx1);
            if (p.out_fmt == 2) {
                *reinterpret_cast<unsigned long long*>(base + (qpix * C + cc) * 2) = 0ull;
            } else {
                char* o8 = base + (qpix * C + (cc & ~7)) * 4 + (cc & 4) * 2;
                *reinterpret_cast<unsigned long long*>(o8) = 0ull;
                *reinterpret_cast<unsigned long long*>(o8 + 16) = 0ull;
            }
            (void)esz;
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

// the same pass with both rows in registers (round 6; cols = 64 NE, the lane -> column map and the summation order of the kernel above:
// same bits): P and dP are read once, dS written once
template <int NE>
__global__ void softmax_bwd_rows_reg_kernel(const float* pm, float* dp, long long rows) {
    const int lane = threadIdx.x & 63;
    const long long row = blockIdx.x * 4ll + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* pr = pm + row * (64 * NE);
    float* dr = dp + row * (64 * NE);
    float pv[NE], dv[NE];
#pragma unroll
    for (int k = 0; k < NE; ++k) {
        pv[k] = pr[lane + 64 * k];
        dv[k] = dr[lane + 64 * k];
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NE; ++k) s += pv[k] * dv[k];
    s = wave_sum(s);
#pragma unroll
    for (int k = 0; k < NE; ++k) dr[lane + 64 * k] = pv[k] * (dv[k] - s);
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

// DP_GNB_NT: -1 (default) by size - both hints from 800 MB of x + dy up -, 0 never, 1 | 2 | 3 forced
int gnb_nt(int B, int H, int W, int C, int x_fmt) {
    const int t = dp_tune(DP_T_GNB_NT);
    if (t >= 0) return t & 3;
    const double bytes = (double)B * H * W * C * (4.0 + (x_fmt ? 2.0 : 4.0));
    return bytes >= 800e6 ? 3 : 0;
}

int fill_common(BwdArgs& p, const char* fn, const void* x1, int C1, const void* x2, int C2, int x_fmt, int B, int H, int W, int G,
                const float* stats, const float* gamma, const float* beta, const float* fscale, const float* fshift,
                int film_stride, int act, int resample, const float* fir4, const float* dy) {
    const int C = C1 + C2;
    DP_REQUIRE(x1 && stats && gamma && beta && dy && B > 0 && H > 0 && W > 0 && G > 0, "%s: bad args", fn);
    DP_REQUIRE(C2 == 0 || x2, "%s: x2 missing", fn);
    DP_REQUIRE(x_fmt == 0 || x_fmt == 1, "%s: x_fmt 0 (fp32) or 1 (plain fp16)", fn);
    DP_REQUIRE(C % (4 * G) == 0 && C1 % 8 == 0 && C % 8 == 0, "%s: need C %% (4*G) == 0 and C1, C %% 8 == 0 (C=%d+%d, G=%d)", fn, C1, C2, G);
    DP_REQUIRE((fscale == nullptr) == (fshift == nullptr), "%s: FiLM scale and shift come together", fn);
    DP_REQUIRE(resample >= 0 && resample <= 4, "%s: resample mode %d", fn, resample);
    DP_REQUIRE((resample != 2 && resample != 4) || (H % 2 == 0 && W % 2 == 0), "%s: 2x down-sampling needs even H, W", fn);
    DP_REQUIRE(resample < 3 || fir4, "%s: the FIR resampling modes (3, 4) need the 4 filter taps", fn);
    DP_REQUIRE(dp_aligned16(x1) && (C2 == 0 || dp_aligned16(x2)) && dp_aligned16(dy), "%s: misaligned tensor", fn);
    p = BwdArgs{};
    p.x1 = static_cast<const float*>(x1); p.x2 = static_cast<const float*>(x2); p.x_fmt = x_fmt; p.C1 = C1; p.C2 = C2; p.B = B; p.H = H; p.W = W; p.G = G;
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

extern "C" int dp_gn_bwd_stats(const void* x1, int C1, const void* x2, int C2, int x_fmt, int B, int H, int W, int G,
                               const float* stats, const float* gamma, const float* beta, const float* fscale,
                               const float* fshift, int film_stride, int act, int resample, const float* fir4, const float* dy,
                               int nsplit, float* partial, float* sums, void* stream) {
    BwdArgs p;
    if (int rc = fill_common(p, "dp_gn_bwd_stats", x1, C1, x2, C2, x_fmt, B, H, W, G, stats, gamma, beta, fscale, fshift,
                             film_stride, act, resample, fir4, dy)) return rc;
    DP_REQUIRE(partial && sums && nsplit > 0, "dp_gn_bwd_stats: scratch missing");
    DP_REQUIRE(p.C4 <= 1024 && G <= p.C4, "dp_gn_bwd_stats: C too wide");
    p.nsplit = nsplit; p.partial = partial; p.cpg4 = p.C4 / G;
    p.nt = gnb_nt(B, H, W, C1 + C2, x_fmt) & 1;
    p.ppb = p.C4 >= 256 ? 1 : 256 / p.C4;
    const int block = p.C4 * p.ppb;
    hipStream_t s = (hipStream_t)stream;
    if (dp_tune(DP_T_GNB_LEAN) != 0 && resample == 0 && x_fmt == 1) {       // the lean form: same partial sums, bit for bit
        if (act) hipLaunchKernelGGL(gn_bwd_stats_lean_kernel<true>, dim3((unsigned)(B * nsplit)), dim3(block), 0, s, p);
        else hipLaunchKernelGGL(gn_bwd_stats_lean_kernel<false>, dim3((unsigned)(B * nsplit)), dim3(block), 0, s, p);
    } else if (resample >= 3) hipLaunchKernelGGL(gn_bwd_stats_kernel<true>, dim3((unsigned)(B * nsplit)), dim3(block), 0, s, p);
    else hipLaunchKernelGGL(gn_bwd_stats_kernel<false>, dim3((unsigned)(B * nsplit)), dim3(block), 0, s, p);
    const double inv = 1.0 / ((double)H * W * p.cpg);
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((B * G + 255) / 256), dim3(256), 0, s, partial, B, nsplit, G, inv, sums);
    DP_LAUNCH_CHECK("gn_bwd_stats");
    return 0;
}

extern "C" int dp_gn_bwd_apply(const void* x1, int C1, const void* x2, int C2, int x_fmt, int B, int H, int W, int G,
                               const float* stats, const float* gamma, const float* beta, const float* fscale,
                               const float* fshift, int film_stride, int act, int resample, const float* fir4, const float* dy,
                               const float* sums, int out_fmt, void* dx1, float* dx2, const float* add1, const float* add2, float add_scale,
                               void* stream) {
    BwdArgs p;
    if (int rc = fill_common(p, "dp_gn_bwd_apply", x1, C1, x2, C2, x_fmt, B, H, W, G, stats, gamma, beta, fscale, fshift,
                             film_stride, act, resample, fir4, dy)) return rc;
    DP_REQUIRE(sums && dx1 && (C2 == 0 || dx2), "dp_gn_bwd_apply: output missing");
    DP_REQUIRE(out_fmt == 0 || ((out_fmt == 1 || out_fmt == 2) && C2 == 0), "dp_gn_bwd_apply: operand output (1 = h2, 2 = h1) needs a single source");
    DP_REQUIRE((!add1 && !add2) || out_fmt == 0, "dp_gn_bwd_apply: an addend needs the fp32 output form");
    DP_REQUIRE((!add1 || dp_aligned16(add1)) && (!add2 || (C2 > 0 && dp_aligned16(add2))), "dp_gn_bwd_apply: addend");
    p.sums = sums; p.dx1 = (float*)dx1; p.dx2 = dx2; p.out_fmt = out_fmt;
    p.add1 = add1; p.add2 = add2; p.add_scale = add_scale;
    p.nt = gnb_nt(B, H, W, C1 + C2, x_fmt);
    const int border = out_fmt ? 1 : 0;
    const long long total = (long long)B * (H + 2 * border) * (W + 2 * border) * (p.C4 / 2);
    {   // the lean form (DP_GNB_LEAN, default 1): un-resampled, fp16 tape, fp32 or plain-fp16-operand output - identical bits
        const int C8 = p.C4 / 2;
        const long long items = (long long)B * H * W * C8;
        if (dp_tune(DP_T_GNB_LEAN) != 0 && resample == 0 && x_fmt == 1 && (out_fmt == 0 || out_fmt == 2) && items < (1ll << 31) &&
            (long long)B * (H + 2) * (W + 2) * C8 < (1ll << 31) && (long long)B * H * W < (1ll << 30)) {
            // both addends or none (a concatenated source with ONE addend goes to the generic kernel)
            const bool add = add1 != nullptr;
            if (!(C2 > 0 && ((add1 != nullptr) != (add2 != nullptr))) && !(C2 == 0 && add2 != nullptr)) {
                // workgroups: at most 4096, and a thread count that is a multiple of C8, so that a thread keeps its octet for life
                long long wg = (items + 255) / 256;
                if (wg > 4096) wg = 4096;
                while (wg > 0 && (wg * 256) % C8 != 0) --wg;
                if (wg > 0) {
                    const int PS = (int)(wg * 256 / C8);
                    const int HW = H * W;
                    const int dB = PS / HW, remp = PS - dB * HW, dY = remp / W, dX = remp - dY * W;
                    const dim3 g((unsigned)wg), blk(256);
                    hipStream_t st_ = (hipStream_t)stream;
#define GNB_LEAN(OUTF_, ADD_)                                                                                                      \
    do {                                                                                                                         \
        if (act) hipLaunchKernelGGL((gn_bwd_apply_lean_kernel<OUTF_, ADD_, true>), g, blk, 0, st_, p, C8, PS, dB, dY, dX);       \
        else hipLaunchKernelGGL((gn_bwd_apply_lean_kernel<OUTF_, ADD_, false>), g, blk, 0, st_, p, C8, PS, dB, dY, dX);          \
    } while (0)
                    if (out_fmt == 2) GNB_LEAN(2, false);
                    else if (add) GNB_LEAN(0, true);
                    else GNB_LEAN(0, false);
#undef GNB_LEAN
                    DP_LAUNCH_CHECK("gn_bwd_apply_lean");
                    return 0;
                }
            }
        }
    }
    if (resample >= 3) hipLaunchKernelGGL(gn_bwd_apply_kernel<true>, dim3(grid_cap(total, 256, 4096)), dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(gn_bwd_apply_kernel<false>, dim3(grid_cap(total, 256, 4096)), dim3(256), 0, (hipStream_t)stream, p);
    DP_LAUNCH_CHECK("gn_bwd_apply");
    return 0;
}

// Channel block of the one-pass form for this tensor shape, or 0 when the three-launch form has to serve (a function of the
// shape only): the largest CB = whole groups, a multiple of 8 channels, dividing C, with 512 % (CB / 4) == 0 and
// HW * CB <= 512 threads * 8 quads * 4, the registers of one workgroup (16 quads per thread spill).
static int gn_bwd_fused_block(int HW, int C, int G, int C1) {
    const int cpg = C / G;
    if (cpg % 4 != 0 || HW <= 0) return 0;
    // (round 6 measured the opposite preference - blocks of at most FOUR quads per thread, 103 registers, two workgroups per CU instead of
    //  one at 175 - on the CIFAR adjoint: 150.7 -> 149.3 images/s at t = 20: the 8-channel blocks halve the contiguous run per pixel (32 B of
    //  dy), and that costs more than the second resident workgroup hides.  Not kept.)
    int best = 0;
    for (int cb = cpg; cb <= C && cb <= 256; cb += cpg) {
        if (C % cb != 0 || cb % 8 != 0 || 512 % (cb / 4) != 0) continue;
        if ((long long)HW * cb > 512ll * 8 * 4) continue;
        if (C1 % cb != 0 && C1 != C) continue;          // a block never straddles the two sources
        best = cb;
    }
    return best;
}

extern "C" int dp_gn_bwd_fused_ok(int H, int W, int C1, int C2, int G, int resample) {
    if (resample < 0 || resample > 2 || G <= 0 || (C1 + C2) % G != 0) return 0;
    return gn_bwd_fused_block(H * W, C1 + C2, G, C1) != 0 ? 1 : 0;
}

extern "C" int dp_gn_bwd_fused(const void* x1, int C1, const void* x2, int C2, int x_fmt, int B, int H, int W, int G,
                               const float* stats, const float* gamma, const float* beta, const float* fscale,
                               const float* fshift, int film_stride, int act, int resample, const float* dy,
                               int out_fmt, void* dx1, float* dx2, const float* add1, const float* add2, float add_scale, void* stream) {
    FusedArgs a{};
    if (int rc = fill_common(a.b, "dp_gn_bwd_fused", x1, C1, x2, C2, x_fmt, B, H, W, G, stats, gamma, beta, fscale, fshift,
                             film_stride, act, resample, nullptr, dy)) return rc;
    DP_REQUIRE(resample <= 2, "dp_gn_bwd_fused: the FIR resampling modes take the three-launch form");
    DP_REQUIRE(dx1 && (C2 == 0 || dx2), "dp_gn_bwd_fused: output missing");
    DP_REQUIRE(out_fmt == 0 || ((out_fmt == 1 || out_fmt == 2) && C2 == 0 && !add1 && !add2),
               "dp_gn_bwd_fused: operand output (1 = h2, 2 = h1) needs a single source and takes no addend");
    DP_REQUIRE((!add1 || dp_aligned16(add1)) && (!add2 || (C2 > 0 && dp_aligned16(add2))), "dp_gn_bwd_fused: addend");
    const int C = C1 + C2;
    a.CB = gn_bwd_fused_block(H * W, C, G, C1);
    DP_REQUIRE(a.CB != 0, "dp_gn_bwd_fused: shape %dx%d x %d channels / %d groups does not fit one workgroup per channel block (ask dp_gn_bwd_fused_ok)", H, W, C, G);
    a.QB = a.CB / 4;
    a.PL = 512 / a.QB;
    a.lean = dp_tune(DP_T_GNB_LEAN) != 0;
    a.xcd_pair = dp_tune(DP_T_XCD_MAP) != 0;
    a.add1 = add1; a.add2 = add2; a.add_scale = add_scale;
    a.b.dx1 = (float*)dx1; a.b.dx2 = dx2; a.b.out_fmt = out_fmt;
    const int items = (H * W + a.PL - 1) / a.PL;
    const dim3 g((unsigned)(B * (C / a.CB))), blk(512);
    hipStream_t s = (hipStream_t)stream;
    if (items <= 2) hipLaunchKernelGGL(gn_bwd_fused_kernel<2>, g, blk, 0, s, a);
    else if (items <= 4) hipLaunchKernelGGL(gn_bwd_fused_kernel<4>, g, blk, 0, s, a);
    else hipLaunchKernelGGL(gn_bwd_fused_kernel<8>, g, blk, 0, s, a);
    DP_LAUNCH_CHECK("gn_bwd_fused");
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
    if (cols == 1024) hipLaunchKernelGGL(softmax_bwd_rows_reg_kernel<16>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, p, dp, rows);
    else if (cols == 256) hipLaunchKernelGGL(softmax_bwd_rows_reg_kernel<4>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, p, dp, rows);
    else if (cols == 64) hipLaunchKernelGGL(softmax_bwd_rows_reg_kernel<1>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, p, dp, rows);
    else hipLaunchKernelGGL(softmax_bwd_rows_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, p, dp, rows, cols);
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
