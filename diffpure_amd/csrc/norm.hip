// GroupNorm over NHWC activations, split in three HBM-bound launches:
//   stats    : every workgroup owns one (sample, pixel-slab); each thread owns ONE channel quad
//              (float4, coalesced 16 B/lane along C) and strides over the slab's pixels, so the
//              whole read is full 64-B..8-KB contiguous rows; per-group sums are combined through
//              LDS in a fixed order (no float atomics -> bitwise reproducible for any sharding).
//   finalize : (mean, rstd) per (sample, group), slabs combined in double.
//   apply    : y = resample(act(FiLM(norm(x)))), float4 in / float4 out, with the channel
//              concatenation of two sources and the 2x nearest-up / 2x2-mean-down of the
//              resampling ResBlocks folded into the same pass.
#include <stdlib.h>

#include "dp_common.h"
#include "dp_tune.h"

namespace {

struct StatsArgs {
    const float* x1;
    const float* x2;
    int C1, C2, B, HW, G, nsplit;
    float* partial;
    int C4, ppb, cpg4;
};

__global__ void gn_stats_kernel(StatsArgs p) {
    __shared__ float red_s[1024];
    __shared__ float red_q[1024];
    const int t = threadIdx.x;
    const int b = blockIdx.x / p.nsplit, sp = blockIdx.x - b * p.nsplit;
    const int per = (p.HW + p.nsplit - 1) / p.nsplit;
    const int p0 = sp * per, p1 = min(p.HW, p0 + per);
    const int pl = t / p.C4, cq = t - pl * p.C4;
    const int c = cq * 4;
    float s = 0.f, q = 0.f;
    if (pl < p.ppb) {
        const bool first = c < p.C1;
        const float* base = first ? (p.x1 + (size_t)b * p.HW * p.C1 + c) : (p.x2 + (size_t)b * p.HW * p.C2 + (c - p.C1));
        const int Cs = first ? p.C1 : p.C2;
        for (int px = p0 + pl; px < p1; px += p.ppb) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(base + (size_t)px * Cs);
            s += (v[0] + v[1]) + (v[2] + v[3]);
            q += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
        }
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

__global__ void gn_finalize_kernel(const float* partial, int B, int nsplit, int G, double inv_count, float eps,
                                   float* stats) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * G) return;
    const int b = i / G, g = i - b * G;
    double s = 0.0, q = 0.0;
    for (int sp = 0; sp < nsplit; ++sp) {
        const float* src = partial + ((size_t)(b * nsplit + sp) * G + g) * 2;
        s += src[0];
        q += src[1];
    }
    const double mean = s * inv_count;
    double var = q * inv_count - mean * mean;
    if (var < 0.0) var = 0.0;
    stats[i * 2] = (float)mean;
    stats[i * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
}

// (mean, rstd) per (sample, group) from the per-column (sum, sumsq) partials the convolution epilogues
// leave behind: source s has C_s channels and one [2][C_s] record per tile of tr_s output rows;
// a sample owns HW / tr_s consecutive tiles.  One workgroup per (sample, group); fixed-order tree.
__global__ void gn_finalize_cols_kernel(const float* cs1, int C1, int tr1, const float* cs2, int C2, int tr2, int HW,
                                        int G, float eps, float* stats) {
    __shared__ double rs[256];
    __shared__ double rq[256];
    const int b = blockIdx.x / G, g = blockIdx.x - b * G;
    const int C = C1 + C2, cpg = C / G;
    const int ch0 = g * cpg, ch1 = ch0 + cpg;
    double s = 0.0, q = 0.0;
    // the group's channels are contiguous inside a record: threads sweep (record, channel) pairs so that
    // consecutive lanes read consecutive floats (a group may straddle the two sources)
    for (int src = 0; src < 2; ++src) {
        const float* base = src ? cs2 : cs1;
        const int Cs = src ? C2 : C1, tr = src ? tr2 : tr1, coff = src ? C1 : 0;
        const int lo = max(ch0, coff) - coff, hi = min(ch1, coff + Cs) - coff;   // channel range inside this source
        if (!base || hi <= lo) continue;
        const int n = hi - lo, tps = HW / tr;
        const float* rec0 = base + (size_t)b * tps * 2 * Cs + lo;
        for (int idx = threadIdx.x; idx < tps * n; idx += blockDim.x) {
            const int t = idx / n, c = idx - t * n;
            const float* rec = rec0 + (size_t)t * 2 * Cs;
            s += rec[c];
            q += rec[Cs + c];
        }
    }
    rs[threadIdx.x] = s;
    rq[threadIdx.x] = q;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            rs[threadIdx.x] += rs[threadIdx.x + o];
            rq[threadIdx.x] += rq[threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double inv = 1.0 / ((double)HW * cpg);
        const double mean = rs[0] * inv;
        double var = rq[0] * inv - mean * mean;
        if (var < 0.0) var = 0.0;
        stats[(b * G + g) * 2] = (float)mean;
        stats[(b * G + g) * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

// Measured and not kept: a line-wide form of the kernel above for single-source statistics (one workgroup per (sample, 32-channel
// block), the lanes of a wave along 32 consecutive channels of two records, so that every 128-byte line of the record array is
// used whole instead of 32 .. 64 bytes of it).  Isolated, back to back at 256^2 x 256, B=64: 45 us instead of 82 us per launch; in
// the purification itself (rocprofv3, 10 100 launches) 20.7 us average instead of 18.9 us: a quarter of the workgroups, each with
// a four times longer dependent sweep - the launch is latency-bound, not byte-bound.
// GroupNorm-apply WITHOUT a finalize launch ("fold", small feature maps): instead of reading (mean, rstd) that
// gn_finalize_cols_kernel wrote, every workgroup of an apply kernel reduces the column records of ITS sample itself - a sample
// of <= FOLD_MAX_TILES record tiles is a few KB..tens of KB of L2-resident floats, against one more dependent launch
// (11-19 us each, 122 per NCSN++ call = 5 % of a CIFAR purification).  Thread c sums channel c over the sample's tiles in
// double (coalesced: consecutive threads, consecutive floats), group g then adds its C/G channel sums in order.  The values
// are a function of the sample's records only - batch composition and sharding cannot change them.
struct FoldSrc {
    const float* cs1;   // [tiles][2][C1] records of source 1 (null: no fold, statistics come from `stats`)
    const float* cs2;   // ... of source 2 (channel concat) or null
    int tr1, tr2;       // output rows per record
    int HW;             // pixels per sample of the INPUT tensor the statistics describe
    float eps;
};
constexpr int FOLD_MAX_TILES = 16;

// (mean, rstd) of every group of sample b from its column records -> LDS (behind the 2 C doubles of scratch)
__device__ __forceinline__ float* gn_fold_reduce(const FoldSrc& f, int b, int C1, int C2, int G, double* sh) {
    const int C = C1 + C2, cpg = C / G;
    float* st = reinterpret_cast<float*>(sh + 2 * C);
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const bool second = c >= C1;
        const float* base = second ? f.cs2 : f.cs1;
        const int Cs = second ? C2 : C1, tr = second ? f.tr2 : f.tr1, cc = second ? c - C1 : c;
        const int tps = f.HW / tr;
        const float* rec = base + (size_t)b * tps * 2 * Cs + cc;
        double s = 0.0, q = 0.0;
        for (int t = 0; t < tps; ++t) {
            s += rec[(size_t)t * 2 * Cs];
            q += rec[(size_t)t * 2 * Cs + Cs];
        }
        sh[c] = s;
        sh[C + c] = q;
    }
    __syncthreads();
    if ((int)threadIdx.x < G) {
        double s = 0.0, q = 0.0;
        for (int j = 0; j < cpg; ++j) {
            s += sh[threadIdx.x * cpg + j];
            q += sh[C + threadIdx.x * cpg + j];
        }
        const double inv = 1.0 / ((double)f.HW * cpg);
        const double mean = s * inv;
        double var = q * inv - mean * mean;
        if (var < 0.0) var = 0.0;
        st[threadIdx.x * 2] = (float)mean;
        st[threadIdx.x * 2 + 1] = (float)(1.0 / sqrt(var + (double)f.eps));
    }
    __syncthreads();
    return st;
}

// -> pointer to this sample's [G][2] (mean, rstd): global memory, or the workgroup's LDS copy it has just computed
__device__ __forceinline__ const float* gn_stats_of(const FoldSrc& f, const float* stats, int b, int C1, int C2, int G, double* sh) {
    if (!f.cs1) return stats + (size_t)b * G * 2;
    return gn_fold_reduce(f, b, C1, C2, G, sh);
}

// gn_finalize_cols for small feature maps (a sample of <= FOLD_MAX_TILES record tiles): ONE workgroup per sample reduces all G
// groups with the arithmetic of gn_stats_of - B workgroups instead of B * G (8192 workgroups of a 256-wide tree for a few hundred
// floats each at CIFAR sizes: the launch was pure overhead, 11 us x 122 per NCSN++ call).
__global__ __launch_bounds__(256) void gn_finalize_cols_sample_kernel(FoldSrc f, int C1, int C2, int G, float* stats) {
    extern __shared__ double fold_lds[];
    const int b = blockIdx.x;
    const float* st = gn_fold_reduce(f, b, C1, C2, G, fold_lds);
    if ((int)threadIdx.x < 2 * G) stats[(size_t)b * G * 2 + threadIdx.x] = st[threadIdx.x];
}

struct ApplyArgs {
    const float* x1;
    const float* x2;
    int C1, C2, B, H, W, G;
    const float* stats;
    const float* gamma;
    const float* beta;
    const float* fscale;
    const float* fshift;
    int film_stride, act, resample;
    float* y;
    int C4, cpg, Ho, Wo;
    char* y_raw;     // optional second output (H2 kernels, resample == 0): the UN-normalised input in bordered h2 form
    float fir[4];    // resample 3 / 4: the 1-D FIR taps k[0..3] (sum 1) of upfirdn2d's separable filter
    FoldSrc fold;
};

// FIR resampling of score_sde's `fir: True` networks (up_or_down_sampling.py:203-265 -> upfirdn2d, op/upfirdn2d_kernel.cu:107-207)
// for a 4-tap separable filter k (sum 1), zero outside the image, evaluated at one output position from an accessor
// v(y, x) of the (already activated) source pixels:
//   up x2   (upsample_2d: kernel 2k per axis, zero insertion, pad (2, 1), correlation with the flipped kernel):
//           out[2i] = 2 (k[3] x[i-1] + k[1] x[i]),  out[2i+1] = 2 (k[2] x[i] + k[0] x[i+1])      per axis
//   down x2 (downsample_2d: pad (1, 1)):  out[i] = sum_j k[3-j] x[2i + j - 1]                     per axis
template <class F>
__device__ __forceinline__ f32x4 fir_up2(const ApplyArgs& p, int oy, int ox, F v) {
    const int iy = oy >> 1, ix = ox >> 1;
    const int ya = (oy & 1) ? iy : iy - 1, xa = (ox & 1) ? ix : ix - 1;
    const float wy[2] = {2.f * ((oy & 1) ? p.fir[2] : p.fir[3]), 2.f * ((oy & 1) ? p.fir[0] : p.fir[1])};
    const float wx[2] = {2.f * ((ox & 1) ? p.fir[2] : p.fir[3]), 2.f * ((ox & 1) ? p.fir[0] : p.fir[1])};
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
            const int y = ya + dy, x = xa + dx;
            if ((unsigned)y >= (unsigned)p.H || (unsigned)x >= (unsigned)p.W) continue;
            const f32x4 s = v(y, x);
            const float w = wy[dy] * wx[dx];
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = fmaf(w, s[j], o[j]);
        }
    return o;
}
template <class F>
__device__ __forceinline__ f32x4 fir_down2(const ApplyArgs& p, int oy, int ox, F v) {
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dy = 0; dy < 4; ++dy)
#pragma unroll
        for (int dx = 0; dx < 4; ++dx) {
            const int y = 2 * oy + dy - 1, x = 2 * ox + dx - 1;
            if ((unsigned)y >= (unsigned)p.H || (unsigned)x >= (unsigned)p.W) continue;
            const f32x4 s = v(y, x);
            const float w = p.fir[3 - dy] * p.fir[3 - dx];
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = fmaf(w, s[j], o[j]);
        }
    return o;
}

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x4 gn_load(const ApplyArgs& p, size_t pix, int c) {
    const float* src = (c < p.C1) ? (p.x1 + pix * p.C1 + c) : (p.x2 + pix * p.C2 + (c - p.C1));
    return *reinterpret_cast<const f32x4*>(src);
}

// h2 output, lane-contiguous version: a thread owns ONE channel QUAD (4 channels) of the row's pixels, so that a
// wave's load is 64 x 16 contiguous bytes (one instruction, whole cache lines) and so is its store.  An h2 octet
// (8 channels: 16 B of hi, then 16 B of lo) is built by the two lanes that own its quads: they exchange their 4
// activated values (one DPP swap each), the even lane stores the octet's hi half, the odd lane its lo half - byte
// offset 16 * quad either way.  Same arithmetic per element as gn_apply_kernel<true, ACT> (bit-identical output).
// FMT = 2 ("h1": plain fp16, 2 bytes per element, the operand of the two-pass / one-pass convolutions): a lane simply
// stores its own quad as 4 fp16 (8 bytes at byte offset 8 * quad) - same values as the hi halves of the h2 form.
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
// FIR: instantiated with and without the upfirdn2d stencils of `fir: True` networks - compiled into the common instantiation they
// cost every network 45 registers per lane (141 instead of 96: three resident waves per SIMD instead of five in an HBM-bound kernel)
template <bool ACT, int FMT, bool FIR>
__global__ __launch_bounds__(256) void gn_apply_h2q_kernel(ApplyArgs p, int CQT, int slots) {
    const int CQ = p.C4;                                 // quads per pixel
    const int Hq = p.Ho + 2, Wq = p.Wo + 2;
    const int b = blockIdx.x / Hq, qy = blockIdx.x - b * Hq;
    const int oy = qy - 1;
    const bool zrow = (unsigned)oy >= (unsigned)p.Ho;
    extern __shared__ double fold_lds[];
    const float* st = p.gamma ? gn_stats_of(p.fold, p.stats, b, p.C1, p.C2, p.G, fold_lds) : nullptr;
    const int slot = threadIdx.x / CQT;
    const bool odd = threadIdx.x & 1;                    // CQT is even: lane parity == quad parity
    const size_t orow = ((size_t)b * Hq + qy) * Wq;
    for (int cq = threadIdx.x - slot * CQT; cq < CQ; cq += CQT) {
        const int c = cq * 4;
        f32x4 a = {1.f, 1.f, 1.f, 1.f}, d = {0.f, 0.f, 0.f, 0.f};       // y = x*a + d before act
        if (p.gamma) {
            const int g = c / p.cpg;
            const float mean = st[g * 2], rstd = st[g * 2 + 1];
            const f32x4 ga = *reinterpret_cast<const f32x4*>(p.gamma + c);
            const f32x4 be = *reinterpret_cast<const f32x4*>(p.beta + c);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                a[j] = rstd * ga[j];
                d[j] = be[j] - mean * a[j];
            }
        }
        if (p.fscale) {
            const f32x4 fs = *reinterpret_cast<const f32x4*>(p.fscale + (size_t)b * p.film_stride + c);
            const f32x4 fh = *reinterpret_cast<const f32x4*>(p.fshift + (size_t)b * p.film_stride + c);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float m = 1.f + fs[j];
                a[j] *= m;
                d[j] = d[j] * m + fh[j];
            }
        }
        auto xf = [&](f32x4 v) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float u = v[j] * a[j] + d[j];
                v[j] = ACT ? dp_silu_f(u) : u;
            }
            return v;
        };
        // the 16 bytes this lane owns of the octet made of (even lane's quad, odd lane's quad)
        auto half_of_octet = [&](f32x4 mine) {
            f32x4 other;
#pragma unroll
            for (int j = 0; j < 4; ++j) other[j] = __shfl_xor(mine[j], 1, 64);
            half8 out;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float v = (j < 4) == !odd ? mine[j & 3] : other[j & 3];      // even lane: [mine | other]; odd: [other | mine]
                const _Float16 hi = (_Float16)v;
                out[j] = odd ? (_Float16)(v - (float)hi) : hi;
            }
            return out;
        };
        // store this lane's share of pixel `opix` of the operand tensor `base`
        auto put = [&](char* base, size_t opix, f32x4 v) {
            if constexpr (FMT == 1) {
                *reinterpret_cast<half8*>(base + (opix * CQ + cq) * 16) = half_of_octet(v);
            } else {
                half4 h;
#pragma unroll
                for (int j = 0; j < 4; ++j) h[j] = (_Float16)v[j];
                *reinterpret_cast<half4*>(base + (opix * CQ + cq) * 8) = h;
            }
        };
        int qx = slot;
        if (p.resample == 0 && !zrow) {
            // interior pixels four at a time: all four loads are issued before the first use
            const size_t irow = ((size_t)b * p.H + oy) * p.W;
            for (; qx + 3 * slots < Wq; qx += 4 * slots) {
                f32x4 rv[4];
                bool in[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int ox = qx + k * slots - 1;
                    in[k] = (unsigned)ox < (unsigned)p.Wo;
                    rv[k] = in[k] ? gn_load(p, irow + ox, c) : f32x4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const size_t opix = orow + qx + k * slots;
                    f32x4 o = xf(rv[k]);
                    if (!in[k]) o = f32x4{0.f, 0.f, 0.f, 0.f};          // border pixel of the row: zeros
                    put(reinterpret_cast<char*>(p.y), opix, o);
                    if (p.y_raw) put(p.y_raw, opix, rv[k]);
                }
            }
        }
        for (; qx < Wq; qx += slots) {
            const int ox = qx - 1;
            const size_t opix = orow + qx;
            if (zrow || (unsigned)ox >= (unsigned)p.Wo) {
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                put(reinterpret_cast<char*>(p.y), opix, z);
                if (p.y_raw) put(p.y_raw, opix, z);
                continue;
            }
            f32x4 o, raw;
            if (p.resample == 0) {
                raw = gn_load(p, ((size_t)b * p.H + oy) * p.W + ox, c);
                o = xf(raw);
            } else if (p.resample == 1) {
                raw = gn_load(p, ((size_t)b * p.H + (oy >> 1)) * p.W + (ox >> 1), c);
                o = xf(raw);
            } else if (FIR && p.resample >= 3) {
                auto src = [&](int y, int x) { return xf(gn_load(p, ((size_t)b * p.H + y) * p.W + x, c)); };
                o = p.resample == 3 ? fir_up2(p, oy, ox, src) : fir_down2(p, oy, ox, src);
                raw = o;
            } else {
                const size_t r0 = ((size_t)b * p.H + 2 * oy) * p.W + 2 * ox;
                const f32x4 v00 = xf(gn_load(p, r0, c)), v01 = xf(gn_load(p, r0 + 1, c));
                const f32x4 v10 = xf(gn_load(p, r0 + p.W, c)), v11 = xf(gn_load(p, r0 + p.W + 1, c));
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = ((v00[j] + v01[j]) + (v10[j] + v11[j])) * 0.25f;
                raw = o;
            }
            put(reinterpret_cast<char*>(p.y), opix, o);
            if (p.y_raw) put(p.y_raw, opix, raw);     // resample == 0 here: the un-normalised input in operand form
        }
    }
}

// One work item = VEC channels of one OUTPUT pixel: VEC = 4 (fp32 out, one float4) or 8 (h2 out:
// 8 fp16 hi | 8 fp16 lo = 32 bytes, the operand format of csrc/igemm_h2.hip).
// H2 output carries a one-pixel zero border ([B][Ho+2][Wo+2][C]): the operand format of
// csrc/igemm_h2.hip, whose loader then needs no bounds tests.
// One workgroup = one (bordered) output row of one sample; a thread owns ONE channel vector (its
// normalisation / FiLM coefficients are formed once, not per pixel) and walks the row's pixels in
// steps of `slots`, so the inner loop has no integer division: load, fma, SiLU, convert, store.
// Lanes run along the channels: a wave touches 64 * VEC * 4 contiguous bytes per step.
template <bool H2, bool ACT, bool FIR>
__global__ __launch_bounds__(256) void gn_apply_kernel(ApplyArgs p, int CVT, int slots) {      // CVT * slots <= 256 threads
    constexpr int VEC = H2 ? 8 : 4, NQ = VEC / 4;
    constexpr int BORDER = H2 ? 1 : 0;
    const int CV = p.C4 * 4 / VEC;
    const int Hq = p.Ho + 2 * BORDER, Wq = p.Wo + 2 * BORDER;
    const int b = blockIdx.x / Hq, qy = blockIdx.x - b * Hq;
    const int oy = qy - BORDER;
    const bool zrow = H2 && (unsigned)oy >= (unsigned)p.Ho;
    extern __shared__ double fold_lds[];
    const float* st = p.gamma ? gn_stats_of(p.fold, p.stats, b, p.C1, p.C2, p.G, fold_lds) : nullptr;
    const int slot = threadIdx.x / CVT;
    const size_t orow = ((size_t)b * Hq + qy) * Wq;     // first pixel of this row in the (bordered) output
    for (int cv = threadIdx.x - slot * CVT; cv < CV; cv += CVT) {
        f32x4 a[NQ], d[NQ];                             // y = x*a + d before act
#pragma unroll
        for (int qd = 0; qd < NQ; ++qd) {
            const int c = cv * VEC + qd * 4;
            a[qd] = f32x4{1.f, 1.f, 1.f, 1.f};
            d[qd] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (p.gamma) {
                const int g = c / p.cpg;
                const float mean = st[g * 2], rstd = st[g * 2 + 1];
                const f32x4 ga = *reinterpret_cast<const f32x4*>(p.gamma + c);
                const f32x4 be = *reinterpret_cast<const f32x4*>(p.beta + c);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    a[qd][j] = rstd * ga[j];
                    d[qd][j] = be[j] - mean * a[qd][j];
                }
            }
            if (p.fscale) {
                const f32x4 fs = *reinterpret_cast<const f32x4*>(p.fscale + (size_t)b * p.film_stride + c);
                const f32x4 fh = *reinterpret_cast<const f32x4*>(p.fshift + (size_t)b * p.film_stride + c);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float m = 1.f + fs[j];
                    a[qd][j] *= m;
                    d[qd][j] = d[qd][j] * m + fh[j];
                }
            }
        }
        for (int qx = slot; qx < Wq; qx += slots) {
            const int ox = qx - BORDER;
            const size_t opix = orow + qx;
            if (H2 && (zrow || (unsigned)ox >= (unsigned)p.Wo)) {
                half8 z;
#pragma unroll
                for (int j = 0; j < 8; ++j) z[j] = (_Float16)0.f;
                half8* dst = reinterpret_cast<half8*>(reinterpret_cast<char*>(p.y) + (opix * CV + cv) * 32);
                dst[0] = z;
                dst[1] = z;
                if (p.y_raw) {
                    half8* dr = reinterpret_cast<half8*>(p.y_raw + (opix * CV + cv) * 32);
                    dr[0] = z;
                    dr[1] = z;
                }
                continue;
            }
            f32x4 o[NQ];
            f32x4 raw[NQ];
#pragma unroll
            for (int qd = 0; qd < NQ; ++qd) {
                const int c = cv * VEC + qd * 4;
                auto xf = [&](size_t pix) {
                    f32x4 v = gn_load(p, pix, c);
                    raw[qd] = v;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float u = v[j] * a[qd][j] + d[qd][j];
                        v[j] = ACT ? dp_silu_f(u) : u;
                    }
                    return v;
                };
                if (p.resample == 0) {
                    o[qd] = xf(((size_t)b * p.H + oy) * p.W + ox);
                } else if (p.resample == 1) {
                    o[qd] = xf(((size_t)b * p.H + (oy >> 1)) * p.W + (ox >> 1));
                } else if (FIR && p.resample >= 3) {
                    auto src = [&](int y, int x) { return xf(((size_t)b * p.H + y) * p.W + x); };
                    o[qd] = p.resample == 3 ? fir_up2(p, oy, ox, src) : fir_down2(p, oy, ox, src);
                } else {
                    const size_t r0 = ((size_t)b * p.H + 2 * oy) * p.W + 2 * ox;
                    const f32x4 v00 = xf(r0), v01 = xf(r0 + 1), v10 = xf(r0 + p.W), v11 = xf(r0 + p.W + 1);
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[qd][j] = ((v00[j] + v01[j]) + (v10[j] + v11[j])) * 0.25f;
                }
            }
            if constexpr (H2) {
                half8 hi, lo;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float v = o[j >> 2][j & 3];
                    hi[j] = (_Float16)v;
                    lo[j] = (_Float16)(v - (float)hi[j]);
                }
                half8* dst = reinterpret_cast<half8*>(reinterpret_cast<char*>(p.y) + (opix * CV + cv) * 32);
                dst[0] = hi;
                dst[1] = lo;
                if (p.y_raw) {       // resample == 0 here: raw[] holds this very pixel
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float v = raw[j >> 2][j & 3];
                        hi[j] = (_Float16)v;
                        lo[j] = (_Float16)(v - (float)hi[j]);
                    }
                    half8* dr = reinterpret_cast<half8*>(p.y_raw + (opix * CV + cv) * 32);
                    dr[0] = hi;
                    dr[1] = lo;
                }
            } else {
                *reinterpret_cast<f32x4*>(p.y + opix * (p.C4 * 4) + cv * 4) = o[0];   // BORDER == 0: opix is the pixel
            }
        }
    }
}

// fp16 in -> "h1" operand out: GroupNorm-apply over a tensor the producing convolution stored as plain fp16 (dp_conv2d_nhwc_h2
// out_fmt 1: [B][H][W][C] fp16, no border) - the second GroupNorm of a ResBlock (FiLM + SiLU, no resampling, one source).
// 4 HBM bytes per element instead of 6.  A thread owns one channel OCTET: 16-byte loads, 16-byte stores, lanes along the
// channels.  Per element the arithmetic is gn_apply_h2q_kernel's on the (exactly representable) fp32 value of the fp16
// input: identical bytes to dp_gn_apply(out_fmt 2) of the up-converted tensor.
struct Apply16Args {
    const _Float16* x;
    int C, B, H, W, G;
    const float* stats;
    const float* gamma;
    const float* beta;
    const float* fscale;
    const float* fshift;
    int film_stride;
    char* y;
    int cpg;
    FoldSrc fold;
};

template <bool ACT>
__global__ __launch_bounds__(256) void gn_apply_f16in_kernel(Apply16Args p, int COT, int slots) {
    const int CO = p.C / 8;
    const int Hq = p.H + 2, Wq = p.W + 2;
    const int b = blockIdx.x / Hq, qy = blockIdx.x - b * Hq;
    const int oy = qy - 1;
    const bool zrow = (unsigned)oy >= (unsigned)p.H;
    extern __shared__ double fold_lds[];
    const float* st = gn_stats_of(p.fold, p.stats, b, p.C, 0, p.G, fold_lds);
    const int slot = threadIdx.x / COT;
    const size_t orow = ((size_t)b * Hq + qy) * Wq;
    half8 zero8;
#pragma unroll
    for (int j = 0; j < 8; ++j) zero8[j] = (_Float16)0.f;
    for (int co = threadIdx.x - slot * COT; co < CO; co += COT) {
        float a[8], d[8];                                  // y = x*a + d before act
#pragma unroll
        for (int qd = 0; qd < 2; ++qd) {
            const int c = co * 8 + qd * 4;
            const int g = c / p.cpg;
            const float mean = st[g * 2], rstd = st[g * 2 + 1];
            const f32x4 ga = *reinterpret_cast<const f32x4*>(p.gamma + c);
            const f32x4 be = *reinterpret_cast<const f32x4*>(p.beta + c);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                a[qd * 4 + j] = rstd * ga[j];
                d[qd * 4 + j] = be[j] - mean * a[qd * 4 + j];
            }
            if (p.fscale) {
                const f32x4 fs = *reinterpret_cast<const f32x4*>(p.fscale + (size_t)b * p.film_stride + c);
                const f32x4 fh = *reinterpret_cast<const f32x4*>(p.fshift + (size_t)b * p.film_stride + c);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float m = 1.f + fs[j];
                    a[qd * 4 + j] *= m;
                    d[qd * 4 + j] = d[qd * 4 + j] * m + fh[j];
                }
            }
        }
        auto xf = [&](half8 v) {
            half8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float u = (float)v[j] * a[j] + d[j];
                o[j] = (_Float16)(ACT ? dp_silu_f(u) : u);
            }
            return o;
        };
        half8* yrow = reinterpret_cast<half8*>(p.y) + orow * CO + co;
        if (zrow) {
            for (int qx = slot; qx < Wq; qx += slots) yrow[(size_t)qx * CO] = zero8;
            continue;
        }
        const half8* xrow = reinterpret_cast<const half8*>(p.x) + ((size_t)b * p.H + oy) * p.W * CO + co;
        int qx = slot;
        for (; qx + 3 * slots < Wq; qx += 4 * slots) {       // four pixels at a time: all loads issued before the first use
            half8 rv[4];
            bool in[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int ox = qx + k * slots - 1;
                in[k] = (unsigned)ox < (unsigned)p.W;
                rv[k] = in[k] ? xrow[(size_t)ox * CO] : zero8;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) yrow[(size_t)(qx + k * slots) * CO] = in[k] ? xf(rv[k]) : zero8;
        }
        for (; qx < Wq; qx += slots) {
            const int ox = qx - 1;
            yrow[(size_t)qx * CO] = (unsigned)ox < (unsigned)p.W ? xf(xrow[(size_t)ox * CO]) : zero8;
        }
    }
}

}  // namespace

extern "C" int dp_gn_stats(const float* x1, int C1, const float* x2, int C2, int B, int HW, int G, int nsplit,
                           float* partial, void* stream) {
    const int C = C1 + C2;
    DP_REQUIRE(x1 && partial && B > 0 && HW > 0 && G > 0 && nsplit > 0, "dp_gn_stats: bad args");
    DP_REQUIRE(C2 == 0 || x2, "dp_gn_stats: x2 missing");
    DP_REQUIRE(C % (4 * G) == 0 && C1 % 4 == 0, "dp_gn_stats: need C %% (4*G) == 0 and C1 %% 4 == 0 (C=%d, G=%d)", C, G);
    DP_REQUIRE(C / 4 <= 1024 && G <= C / 4, "dp_gn_stats: C=%d too wide", C);
    DP_REQUIRE(dp_aligned16(x1) && (C2 == 0 || dp_aligned16(x2)), "dp_gn_stats: misaligned input");
    StatsArgs p{x1, x2, C1, C2, B, HW, G, nsplit, partial, C / 4, 1, C / 4 / G};
    p.ppb = p.C4 >= 256 ? 1 : 256 / p.C4;
    const int block = p.C4 * p.ppb;
    DP_REQUIRE(block >= G && block <= 1024, "dp_gn_stats: internal block size %d", block);
    hipLaunchKernelGGL(gn_stats_kernel, dim3((unsigned)(B * nsplit)), dim3(block), 0, (hipStream_t)stream, p);
    DP_LAUNCH_CHECK("gn_stats");
    return 0;
}

extern "C" int dp_gn_finalize(const float* partial, int B, int nsplit, int G, long long count, float eps, float* stats,
                              void* stream) {
    DP_REQUIRE(partial && stats && B > 0 && nsplit > 0 && G > 0 && count > 0, "dp_gn_finalize: bad args");
    hipLaunchKernelGGL(gn_finalize_kernel, dim3((B * G + 255) / 256), dim3(256), 0, (hipStream_t)stream, partial, B,
                       nsplit, G, 1.0 / (double)count, eps, stats);
    DP_LAUNCH_CHECK("gn_finalize");
    return 0;
}

extern "C" int dp_gn_finalize_cols(const float* cs1, int C1, int tile_rows1, const float* cs2, int C2, int tile_rows2, int B,
                                   int HW, int G, float eps, float* stats, void* stream) {
    DP_REQUIRE(cs1 && stats && B > 0 && HW > 0 && G > 0 && C1 > 0 && C2 >= 0 && (C2 == 0 || cs2), "dp_gn_finalize_cols: bad args");
    DP_REQUIRE((C1 + C2) % G == 0, "dp_gn_finalize_cols: C must be a multiple of G");
    DP_REQUIRE(tile_rows1 > 0 && HW % tile_rows1 == 0 && (C2 == 0 || (tile_rows2 > 0 && HW % tile_rows2 == 0)),
               "dp_gn_finalize_cols: a convolution tile must not straddle two samples (HW %% tile_rows != 0)");
    const bool small = HW / tile_rows1 <= FOLD_MAX_TILES && (C2 == 0 || HW / tile_rows2 <= FOLD_MAX_TILES) && G <= 128 &&
                       dp_tune(DP_T_GN_FINALIZE_SAMPLE) != 0;
    if (small) {     // one workgroup per sample (same values: tests/test_gpu_ops.py compares the two bit for bit)
        const FoldSrc f{cs1, cs2, tile_rows1, C2 ? tile_rows2 : 1, HW, eps};
        hipLaunchKernelGGL(gn_finalize_cols_sample_kernel, dim3((unsigned)B), dim3(256), (size_t)(C1 + C2) * 16 + (size_t)G * 8,
                           (hipStream_t)stream, f, C1, C2, G, stats);
        DP_LAUNCH_CHECK("gn_finalize_cols_sample");
        return 0;
    }
    hipLaunchKernelGGL(gn_finalize_cols_kernel, dim3((unsigned)(B * G)), dim3(256), 0, (hipStream_t)stream, cs1, C1, tile_rows1,
                       cs2, C2, C2 ? tile_rows2 : 1, HW, G, eps, stats);
    DP_LAUNCH_CHECK("gn_finalize_cols");
    return 0;
}

extern "C" int dp_gn_apply(const float* x1, int C1, const float* x2, int C2, int B, int H, int W, int G,
                           const float* stats, const float* gamma, const float* beta, const float* fscale,
                           const float* fshift, int film_stride, int act, int resample, int out_fmt, void* y,
                           void* y_raw, const float* fir4, const float* cs1, int tile_rows1, const float* cs2, int tile_rows2,
                           float eps, void* stream) {
    const int C = C1 + C2;
    DP_REQUIRE(x1 && y && B > 0 && H > 0 && W > 0, "dp_gn_apply: bad args");
    DP_REQUIRE(C2 == 0 || x2, "dp_gn_apply: x2 missing");
    DP_REQUIRE(C % 4 == 0 && C1 % 4 == 0, "dp_gn_apply: channel counts must be multiples of 4");
    DP_REQUIRE(!gamma || (beta && (stats || cs1) && G > 0 && C % (4 * G) == 0), "dp_gn_apply: need beta, stats (or column records) and C %% (4*G) == 0");
    DP_REQUIRE(!cs1 || (gamma && !stats && tile_rows1 > 0 && (H * W) % tile_rows1 == 0 && (H * W) / tile_rows1 <= FOLD_MAX_TILES &&
                        (C2 == 0 || (cs2 && tile_rows2 > 0 && (H * W) % tile_rows2 == 0 && (H * W) / tile_rows2 <= FOLD_MAX_TILES)) && G <= 256),
               "dp_gn_apply: folded statistics need the column records of every source, whole record tiles per sample and at most %d "
               "tiles per sample (H*W = %d)", FOLD_MAX_TILES, H * W);
    DP_REQUIRE((fscale == nullptr) == (fshift == nullptr), "dp_gn_apply: FiLM scale and shift come together");
    DP_REQUIRE(resample >= 0 && resample <= 4, "dp_gn_apply: resample mode %d", resample);
    DP_REQUIRE((resample != 2 && resample != 4) || (H % 2 == 0 && W % 2 == 0), "dp_gn_apply: 2x down-sampling needs even H, W");
    DP_REQUIRE(resample < 3 || fir4, "dp_gn_apply: the FIR resampling modes (3, 4) need the 4 filter taps");
    DP_REQUIRE(dp_aligned16(x1) && (C2 == 0 || dp_aligned16(x2)) && dp_aligned16(y), "dp_gn_apply: misaligned tensor");
    DP_REQUIRE(!fscale || (film_stride % 4 == 0 && dp_aligned16(fscale) && dp_aligned16(fshift)), "dp_gn_apply: misaligned FiLM rows");
    DP_REQUIRE(out_fmt == 0 || ((out_fmt == 1 || out_fmt == 2) && C % 8 == 0 && C1 % 8 == 0), "dp_gn_apply: out_fmt %d needs channel counts that are multiples of 8", out_fmt);
    DP_REQUIRE(!y_raw || (out_fmt != 0 && resample == 0), "dp_gn_apply: the raw operand output needs out_fmt=1|2 and no resampling");
    ApplyArgs p{x1, x2, C1, C2, B, H, W, G, stats, gamma, beta, fscale, fshift, film_stride, act, resample, (float*)y,
                C / 4, gamma ? C / G : C, (resample == 1 || resample == 3) ? 2 * H : ((resample == 2 || resample == 4) ? H / 2 : H),
                (resample == 1 || resample == 3) ? 2 * W : ((resample == 2 || resample == 4) ? W / 2 : W), (char*)y_raw, {0.f, 0.f, 0.f, 0.f},
                FoldSrc{cs1, cs2, tile_rows1, C2 ? tile_rows2 : 1, H * W, eps}};
    const size_t shm = cs1 ? (size_t)C * 16 + (size_t)G * 8 : 0;      // folded statistics: 2 C doubles + G (mean, rstd) pairs
    if (resample >= 3)
        for (int i = 0; i < 4; ++i) p.fir[i] = fir4[i];
    const int CV = out_fmt ? C / 8 : C / 4;
    const int CVT = CV < 256 ? CV : 256, slots = 256 / CVT;
    const unsigned rows = (unsigned)(B * (out_fmt ? p.Ho + 2 : p.Ho));
#define GN_APPLY_LAUNCH(H2_, ACT_) \
    do {                                                                                                                              \
        if (resample >= 3) hipLaunchKernelGGL((gn_apply_kernel<H2_, ACT_, true>), dim3(rows), dim3(CVT * slots), shm, (hipStream_t)stream, p, CVT, slots);   \
        else hipLaunchKernelGGL((gn_apply_kernel<H2_, ACT_, false>), dim3(rows), dim3(CVT * slots), shm, (hipStream_t)stream, p, CVT, slots);              \
    } while (0)
#define GN_H2Q_LAUNCH(ACT_, FMT_)                                                                                                      \
    do {                                                                                                                              \
        if (resample >= 3) hipLaunchKernelGGL((gn_apply_h2q_kernel<ACT_, FMT_, true>), dim3(rows), dim3(CQT * qslots), shm, (hipStream_t)stream, p, CQT, qslots);  \
        else hipLaunchKernelGGL((gn_apply_h2q_kernel<ACT_, FMT_, false>), dim3(rows), dim3(CQT * qslots), shm, (hipStream_t)stream, p, CQT, qslots);             \
    } while (0)
    // h2 output: the lane-contiguous quad kernel unless DP_GN_APPLY_QUAD=0 (A/B switch; both give identical bytes)
    const bool quad = dp_tune(DP_T_GN_APPLY_QUAD) != 0;
    if (out_fmt == 2) {
        const int CQ = C / 4, CQT = CQ < 256 ? CQ : 256, qslots = 256 / CQT;
        if (act) GN_H2Q_LAUNCH(true, 2);
        else GN_H2Q_LAUNCH(false, 2);
    } else if (out_fmt && quad) {
        const int CQ = C / 4, CQT = CQ < 256 ? CQ : 256, qslots = 256 / CQT;     // C % 8 == 0: CQ and CQT are even
        if (act) GN_H2Q_LAUNCH(true, 1);
        else GN_H2Q_LAUNCH(false, 1);
    } else if (out_fmt) {
        if (act) GN_APPLY_LAUNCH(true, true);
        else GN_APPLY_LAUNCH(true, false);
    } else {
        if (act) GN_APPLY_LAUNCH(false, true);
        else GN_APPLY_LAUNCH(false, false);
    }
#undef GN_APPLY_LAUNCH
#undef GN_H2Q_LAUNCH
    DP_LAUNCH_CHECK("gn_apply");
    return 0;
}

extern "C" int dp_gn_apply_f16in(const void* x16, int C, int B, int H, int W, int G, const float* stats, const float* gamma,
                                 const float* beta, const float* fscale, const float* fshift, int film_stride, int act, void* y,
                                 const float* cs1, int tile_rows1, float eps, void* stream) {
    DP_REQUIRE(x16 && y && (stats || cs1) && gamma && beta && B > 0 && H > 0 && W > 0 && G > 0, "dp_gn_apply_f16in: bad args");
    DP_REQUIRE(!cs1 || (!stats && tile_rows1 > 0 && (H * W) % tile_rows1 == 0 && (H * W) / tile_rows1 <= FOLD_MAX_TILES && G <= 256),
               "dp_gn_apply_f16in: folded statistics need whole record tiles per sample, at most %d of them (H*W = %d)", FOLD_MAX_TILES, H * W);
    DP_REQUIRE(C % 8 == 0 && C % (4 * G) == 0, "dp_gn_apply_f16in: need C %% 8 == 0 and C %% (4*G) == 0 (C=%d, G=%d)", C, G);
    DP_REQUIRE((fscale == nullptr) == (fshift == nullptr), "dp_gn_apply_f16in: FiLM scale and shift come together");
    DP_REQUIRE(dp_aligned16(x16) && dp_aligned16(y) && dp_aligned16(gamma) && dp_aligned16(beta), "dp_gn_apply_f16in: misaligned tensor");
    DP_REQUIRE(!fscale || (film_stride % 4 == 0 && dp_aligned16(fscale) && dp_aligned16(fshift)), "dp_gn_apply_f16in: misaligned FiLM rows");
    Apply16Args p{(const _Float16*)x16, C, B, H, W, G, stats, gamma, beta, fscale, fshift, film_stride, (char*)y, C / G,
                  FoldSrc{cs1, nullptr, tile_rows1, 1, H * W, eps}};
    const size_t shm = cs1 ? (size_t)C * 16 + (size_t)G * 8 : 0;
    const int CO = C / 8, COT = CO < 256 ? CO : 256, slots = 256 / COT;
    const unsigned rows = (unsigned)(B * (H + 2));
    if (act) hipLaunchKernelGGL((gn_apply_f16in_kernel<true>), dim3(rows), dim3(COT * slots), shm, (hipStream_t)stream, p, COT, slots);
    else hipLaunchKernelGGL((gn_apply_f16in_kernel<false>), dim3(rows), dim3(COT * slots), shm, (hipStream_t)stream, p, COT, slots);
    DP_LAUNCH_CHECK("gn_apply_f16in");
    return 0;
}
