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
                                        int G, float eps, float* stats, int xcd_group) {
    __shared__ double rs[256];
    __shared__ double rq[256];
    // round 6: four NEIGHBOURING groups of a sample share every 128-byte line of a record (8 channels x 4 bytes each) and workgroups are
    // dealt to the 8 XCDs round-robin: inside every 32 workgroups, physical r, r + 8, r + 16, r + 24 (one XCD) take logical 4k .. 4k + 3
    int bid = blockIdx.x;
    if (xcd_group && (int)(gridDim.x - (gridDim.x & 31)) > bid) {
        const int r = bid & 31;
        bid = (bid & ~31) + 4 * (r & 7) + (r >> 3);
    }
    const int b = bid / G, g = bid - b * G;
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
        if (blockDim.x % n == 0) {
            // round 6: the workgroup size is a multiple of the group's width (always, for power-of-two groups), so idx % n is the thread's
            // own constant and idx / n advances by blockDim / n: the same (record, channel) pairs in the same order per thread - the same
            // sums bit for bit - without an integer division per 8 bytes loaded, four records in flight
            const int c = threadIdx.x % n, t0 = threadIdx.x / n, step = blockDim.x / n;
            const float* rc = rec0 + c;
            int t = t0;
            for (; t + 3 * step < tps; t += 4 * step) {
                float a[4], d[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float* rec = rc + (size_t)(t + k * step) * 2 * Cs;
                    a[k] = rec[0];
                    d[k] = rec[Cs];
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    s += a[k];
                    q += d[k];
                }
            }
            for (; t < tps; t += step) {
                const float* rec = rc + (size_t)t * 2 * Cs;
                s += rec[0];
                q += rec[Cs];
            }
            continue;
        }
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
// Per-sample reduction of the column records (small feature maps: a sample of <= FOLD_MAX_TILES record tiles is a few KB..tens of
// KB of L2-resident floats).  Thread c sums channel c over the sample's tiles in double (coalesced: consecutive threads,
// consecutive floats), group g then adds its C/G channel sums in order.  The values are a function of the sample's records only -
// batch composition and sharding cannot change them.  (Round 3 also let the GroupNorm-APPLY kernels run this reduction themselves
// instead of a finalize launch: bit-identical, 4.5 % slower on the CIFAR purification, removed in round 4.)
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

// gn_finalize_cols for small feature maps (a sample of <= FOLD_MAX_TILES record tiles): ONE workgroup per sample reduces all G
// groups - B workgroups instead of B * G (8192 workgroups of a 256-wide tree for a few hundred
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
    char* y_raw;     // optional second output (operand kernels, resample == 0): the UN-normalised input in bordered operand form
    float fir[4];    // resample 3 / 4: the 1-D FIR taps k[0..3] (sum 1) of upfirdn2d's separable filter
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
    const float* st = p.gamma ? p.stats + (size_t)b * p.G * 2 : nullptr;
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
                float u = fmaf(v[j], a[j], d[j]);       // (explicit: every GroupNorm-apply kernel evaluates the same fused multiply-add)
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
                const _Float16 hi = dp_to_half(v);
                out[j] = odd ? dp_to_half(v - (float)hi) : hi;
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
                for (int j = 0; j < 4; ++j) h[j] = dp_to_half(v[j]);
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

// fp32 -> fp32 form (the tape of the adjoint passes, the resampled identity skip of the fp32 residual stream): one work item =
// 4 channels of one OUTPUT pixel (one float4).  One workgroup = one output row of one sample; a thread owns ONE channel quad (its
// normalisation / FiLM coefficients are formed once, not per pixel) and walks the row's pixels in steps of `slots`, so the inner
// loop has no integer division: load, fma, SiLU, store.  Lanes run along the channels: a wave touches 64 * 16 contiguous bytes
// per step.  (The operand formats are written by gn_apply_h2q_kernel above; the octet-per-thread h2 form of round 1 is gone.)
template <bool ACT, bool FIR>
__global__ __launch_bounds__(256) void gn_apply_kernel(ApplyArgs p, int CVT, int slots) {      // CVT * slots <= 256 threads
    const int CV = p.C4;
    const int b = blockIdx.x / p.Ho, oy = blockIdx.x - b * p.Ho;
    const float* st = p.gamma ? p.stats + (size_t)b * p.G * 2 : nullptr;
    const int slot = threadIdx.x / CVT;
    const size_t orow = ((size_t)b * p.Ho + oy) * p.Wo;     // first pixel of this row in the output
    for (int cv = threadIdx.x - slot * CVT; cv < CV; cv += CVT) {
        const int c = cv * 4;
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
        auto xf = [&](size_t pix) {
            f32x4 v = gn_load(p, pix, c);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float u = fmaf(v[j], a[j], d[j]);       // (explicit: every GroupNorm-apply kernel evaluates the same fused multiply-add)
                v[j] = ACT ? dp_silu_f(u) : u;
            }
            return v;
        };
        for (int ox = slot; ox < p.Wo; ox += slots) {
            f32x4 o;
            if (p.resample == 0) {
                o = xf(((size_t)b * p.H + oy) * p.W + ox);
            } else if (p.resample == 1) {
                o = xf(((size_t)b * p.H + (oy >> 1)) * p.W + (ox >> 1));
            } else if (FIR && p.resample >= 3) {
                auto src = [&](int y, int x) { return xf(((size_t)b * p.H + y) * p.W + x); };
                o = p.resample == 3 ? fir_up2(p, oy, ox, src) : fir_down2(p, oy, ox, src);
            } else {
                const size_t r0 = ((size_t)b * p.H + 2 * oy) * p.W + 2 * ox;
                const f32x4 v00 = xf(r0), v01 = xf(r0 + 1), v10 = xf(r0 + p.W), v11 = xf(r0 + p.W + 1);
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = ((v00[j] + v01[j]) + (v10[j] + v11[j])) * 0.25f;
            }
            *reinterpret_cast<f32x4*>(p.y + (orow + ox) * (p.C4 * 4) + c) = o;
        }
    }
}

// fp16 in -> fp16 out ("h16"): GroupNorm-apply over tensors that are stored as PLAIN fp16 [B][H][W][C] (no border) - the output of a
// ResBlock's first convolution (round 3) and, since round 4, the whole residual stream of the fp16 x fp16 modes (the reference's
// own `use_fp16` torso keeps h in fp16: guided_diffusion/unet.py:626-632).  Everything gn_apply_h2q_kernel does for an fp32
// input: two sources (the skip concatenation, never materialised), optional normalisation + FiLM + SiLU, 2x nearest-up / 2x2
// mean-down resampling, and the raw input as a second operand - at 4 instead of 6 HBM bytes per element.  Output: the
// zero-bordered "h1" operand [B][Ho+2][Wo+2][C] (BORDER 1) or a plain fp16 tensor [B][Ho][Wo][C] (BORDER 0: the resampled
// identity skip of an up / down ResBlock, unet.py:245-250 - the residual of its second convolution).
// A thread owns one channel OCTET: 16-byte loads, 16-byte stores, lanes along the channels.  Per element the arithmetic is
// gn_apply_h2q_kernel's on the (exactly representable) fp32 value of the fp16 input: identical bytes to dp_gn_apply(out_fmt 2)
// of the up-converted tensor (tests/test_gpu_ops.py).
struct Apply16Args {
    const _Float16* x1;
    const _Float16* x2;
    int C1, C2, B, H, W, G;
    const float* stats;
    const float* gamma;
    const float* beta;
    const float* fscale;
    const float* fshift;
    int film_stride, resample;
    char* y;
    char* y_raw;
    int cpg, Ho, Wo;
};

// RS: the resampling mode as a template parameter (round 6) - the resampling loops keep eight loads in flight and must not cost the
// un-resampled instantiation (92 registers: five resident waves per SIMD) its occupancy
// NTM (probe, DP_GN_NT): non-temporal hints on the streaming accesses of the un-resampled path - bit 0 the loads, bit 1 the stores
template <bool ACT, int BORDER, int RS, int NTM = 0>
__global__ __launch_bounds__(256) void gn_apply_h16_kernel(Apply16Args p, int COT, int slots) {
    const int C = p.C1 + p.C2, CO = C / 8;
    const int Hq = p.Ho + 2 * BORDER, Wq = p.Wo + 2 * BORDER;
    const int b = blockIdx.x / Hq, qy = blockIdx.x - b * Hq;
    const int oy = qy - BORDER;
    const bool zrow = BORDER && (unsigned)oy >= (unsigned)p.Ho;
    const float* st = p.gamma ? p.stats + (size_t)b * p.G * 2 : nullptr;
    // pixel slot of this thread among ALL workgroups of the row: gridDim.y > 1 (round 6) cuts a row's pixels across several workgroups
    // where one workgroup per row would leave the chip short of them (small batches / low levels); `slots` counts the row's slots
    const int lslot = threadIdx.x / COT;
    const int slot = blockIdx.y * (blockDim.x / COT) + lslot;
    const size_t orow = ((size_t)b * Hq + qy) * Wq;
    half8 zero8;
#pragma unroll
    for (int j = 0; j < 8; ++j) zero8[j] = (_Float16)0.f;
    for (int co = threadIdx.x - lslot * COT; co < CO; co += COT) {
        const int c0 = co * 8;
        float a[8], d[8];                                  // y = x*a + d before act
#pragma unroll
        for (int qd = 0; qd < 2; ++qd) {
            const int c = c0 + qd * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                a[qd * 4 + j] = 1.f;
                d[qd * 4 + j] = 0.f;
            }
            if (p.gamma) {
                const int g = c / p.cpg;
                const float mean = st[g * 2], rstd = st[g * 2 + 1];
                const f32x4 ga = *reinterpret_cast<const f32x4*>(p.gamma + c);
                const f32x4 be = *reinterpret_cast<const f32x4*>(p.beta + c);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    a[qd * 4 + j] = rstd * ga[j];
                    d[qd * 4 + j] = be[j] - mean * a[qd * 4 + j];
                }
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
        // this octet's source tensor (the channel split C1 is a multiple of 8) and its octet stride per pixel
        const bool first = c0 < p.C1;
        const half8* src = first ? reinterpret_cast<const half8*>(p.x1) + co : reinterpret_cast<const half8*>(p.x2) + (co - p.C1 / 8);
        const int so = (first ? p.C1 : p.C2) / 8;
        auto xf32 = [&](half8 v, float (&o)[8]) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float u = fmaf((float)v[j], a[j], d[j]);
                o[j] = ACT ? dp_silu_fast_f(u) : u;            // every result of this kernel is stored as fp16
            }
        };
        auto xf = [&](half8 v) {
            float o[8];
            xf32(v, o);
            half8 h;
#pragma unroll
            for (int j = 0; j < 8; ++j) h[j] = dp_to_half(o[j]);
            return h;
        };
        half8* yrow = reinterpret_cast<half8*>(p.y) + orow * CO + co;
        // (RS != 0: y_raw is the un-bordered resampled tensor, addressed by `srow` below - never through the bordered geometry)
        half8* rrow = (RS == 0 && p.y_raw) ? reinterpret_cast<half8*>(p.y_raw) + orow * CO + co : nullptr;
        if (zrow) {
            for (int qx = slot; qx < Wq; qx += slots) {
                yrow[(size_t)qx * CO] = zero8;
                if (rrow) rrow[(size_t)qx * CO] = zero8;
            }
            continue;
        }
        int qx = slot;
        if constexpr (RS == 0) {
            const half8* xrow = src + ((size_t)b * p.H + oy) * p.W * so;
            for (; qx + 3 * slots < Wq; qx += 4 * slots) {       // four pixels at a time: all loads issued before the first use
                half8 rv[4];
                bool in[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int ox = qx + k * slots - BORDER;
                    in[k] = (unsigned)ox < (unsigned)p.Wo;
                    if constexpr (NTM & 1) rv[k] = in[k] ? __builtin_nontemporal_load(xrow + (size_t)ox * so) : zero8;
                    else rv[k] = in[k] ? xrow[(size_t)ox * so] : zero8;
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if constexpr (NTM & 2) __builtin_nontemporal_store(in[k] ? xf(rv[k]) : zero8, yrow + (size_t)(qx + k * slots) * CO);
                    else yrow[(size_t)(qx + k * slots) * CO] = in[k] ? xf(rv[k]) : zero8;
                    if (rrow) rrow[(size_t)(qx + k * slots) * CO] = rv[k];
                }
            }
        }
        if constexpr (RS == 0) {
            for (; qx < Wq; qx += slots) {
                const int ox = qx - BORDER;
                if ((unsigned)ox >= (unsigned)p.Wo) {
                    yrow[(size_t)qx * CO] = zero8;
                    if (rrow) rrow[(size_t)qx * CO] = zero8;
                    continue;
                }
                const half8 v = src[(((size_t)b * p.H + oy) * p.W + ox) * so];
                yrow[(size_t)qx * CO] = xf(v);
                if (rrow) rrow[(size_t)qx * CO] = v;
            }
            continue;
        }
        // ---- 2x resampling.  y_raw here (round 6) is the RESAMPLED RAW input as a plain tensor [B][Ho][Wo][C] - the identity skip of an
        // up / down ResBlock (unet.py:245-250), the residual of its second convolution - written from the values this pass has in
        // registers anyway (rounds 4-5 ran a second launch over x for it: 1.2 % of the headline step)
        if (BORDER && slot == 0) {
            yrow[0] = zero8;
            yrow[(size_t)(Wq - 1) * CO] = zero8;
        }
        half8* srow = p.y_raw ? reinterpret_cast<half8*>(p.y_raw) + ((size_t)b * p.Ho + oy) * p.Wo * CO + co : nullptr;
        if constexpr (RS == 1) {
            // nearest x2: a source pixel is normalised ONCE for its two outputs of this row; four source pixels in flight
            const half8* xrow = src + ((size_t)b * p.H + (oy >> 1)) * p.W * so;
            for (int sx = slot; sx < p.W; sx += 4 * slots) {
                half8 rv[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) rv[k] = sx + k * slots < p.W ? xrow[(size_t)(sx + k * slots) * so] : zero8;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int s1 = sx + k * slots;
                    if (s1 >= p.W) break;
                    const half8 o = xf(rv[k]);
                    yrow[(size_t)(2 * s1 + BORDER) * CO] = o;
                    yrow[(size_t)(2 * s1 + 1 + BORDER) * CO] = o;
                    if (srow) {
                        srow[(size_t)(2 * s1) * CO] = rv[k];
                        srow[(size_t)(2 * s1 + 1) * CO] = rv[k];
                    }
                }
            }
        } else if constexpr (RS == 2) {
            // 2x2 mean: two outputs (eight source loads) in flight
            const half8* xr0 = src + ((size_t)b * p.H + 2 * oy) * p.W * so;
            const half8* xr1 = xr0 + (size_t)p.W * so;
            for (int ox = slot; ox < p.Wo; ox += 2 * slots) {
                half8 rv[2][4];
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int o1 = ox + k * slots;
                    const bool in = o1 < p.Wo;
                    rv[k][0] = in ? xr0[(size_t)(2 * o1) * so] : zero8;
                    rv[k][1] = in ? xr0[(size_t)(2 * o1 + 1) * so] : zero8;
                    rv[k][2] = in ? xr1[(size_t)(2 * o1) * so] : zero8;
                    rv[k][3] = in ? xr1[(size_t)(2 * o1 + 1) * so] : zero8;
                }
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int o1 = ox + k * slots;
                    if (o1 >= p.Wo) break;
                    float v00[8], v01[8], v10[8], v11[8];
                    xf32(rv[k][0], v00);
                    xf32(rv[k][1], v01);
                    xf32(rv[k][2], v10);
                    xf32(rv[k][3], v11);
                    half8 o;
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] = dp_to_half(((v00[j] + v01[j]) + (v10[j] + v11[j])) * 0.25f);
                    yrow[(size_t)(o1 + BORDER) * CO] = o;
                    if (srow) {
#pragma unroll
                        for (int j = 0; j < 8; ++j)
                            o[j] = dp_to_half((((float)rv[k][0][j] + (float)rv[k][1][j]) + ((float)rv[k][2][j] + (float)rv[k][3][j])) * 0.25f);
                        srow[(size_t)o1 * CO] = o;
                    }
                }
            }
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
                       cs2, C2, C2 ? tile_rows2 : 1, HW, G, eps, stats, dp_tune(DP_T_XCD_MAP) != 0 ? 1 : 0);
    DP_LAUNCH_CHECK("gn_finalize_cols");
    return 0;
}

extern "C" int dp_gn_apply(const float* x1, int C1, const float* x2, int C2, int B, int H, int W, int G,
                           const float* stats, const float* gamma, const float* beta, const float* fscale,
                           const float* fshift, int film_stride, int act, int resample, int out_fmt, void* y,
                           void* y_raw, const float* fir4, void* stream) {
    const int C = C1 + C2;
    DP_REQUIRE(x1 && y && B > 0 && H > 0 && W > 0, "dp_gn_apply: bad args");
    DP_REQUIRE(C2 == 0 || x2, "dp_gn_apply: x2 missing");
    DP_REQUIRE(C % 4 == 0 && C1 % 4 == 0, "dp_gn_apply: channel counts must be multiples of 4");
    DP_REQUIRE(!gamma || (beta && stats && G > 0 && C % (4 * G) == 0), "dp_gn_apply: need beta, stats and C %% (4*G) == 0");
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
                (resample == 1 || resample == 3) ? 2 * W : ((resample == 2 || resample == 4) ? W / 2 : W), (char*)y_raw, {0.f, 0.f, 0.f, 0.f}};
    if (resample >= 3)
        for (int i = 0; i < 4; ++i) p.fir[i] = fir4[i];
    const int CQ = C / 4, CQT = CQ < 256 ? CQ : 256, qslots = 256 / CQT;     // operand formats: C % 8 == 0, so CQ and CQT are even
    const unsigned rows = (unsigned)(B * (out_fmt ? p.Ho + 2 : p.Ho));
    void* rec = nullptr;
    {   // algorithmic HBM bytes: every source element the output needs once (4 B) + every output element once
        const double in_px = resample == 1 || resample == 3 ? (double)H * W : (double)p.Ho * p.Wo * (resample ? 4 : 1);
        const double out_px = (double)(out_fmt ? (p.Ho + 2) * (p.Wo + 2) : p.Ho * p.Wo);
        dp_prof_begin(DP_PROF_GN_APPLY, 0.0, (double)B * C * (in_px * 4 + out_px * (out_fmt == 2 ? 2 : 4) * (y_raw ? 2 : 1)), (hipStream_t)stream, &rec);
    }
#define GN_APPLY_LAUNCH(ACT_) \
    do {                                                                                                                              \
        if (resample >= 3) hipLaunchKernelGGL((gn_apply_kernel<ACT_, true>), dim3(rows), dim3(CQT * qslots), 0, (hipStream_t)stream, p, CQT, qslots);   \
        else hipLaunchKernelGGL((gn_apply_kernel<ACT_, false>), dim3(rows), dim3(CQT * qslots), 0, (hipStream_t)stream, p, CQT, qslots);              \
    } while (0)
#define GN_H2Q_LAUNCH(ACT_, FMT_)                                                                                                      \
    do {                                                                                                                              \
        if (resample >= 3) hipLaunchKernelGGL((gn_apply_h2q_kernel<ACT_, FMT_, true>), dim3(rows), dim3(CQT * qslots), 0, (hipStream_t)stream, p, CQT, qslots);  \
        else hipLaunchKernelGGL((gn_apply_h2q_kernel<ACT_, FMT_, false>), dim3(rows), dim3(CQT * qslots), 0, (hipStream_t)stream, p, CQT, qslots);             \
    } while (0)
    if (out_fmt == 2) {
        if (act) GN_H2Q_LAUNCH(true, 2);
        else GN_H2Q_LAUNCH(false, 2);
    } else if (out_fmt == 1) {
        if (act) GN_H2Q_LAUNCH(true, 1);
        else GN_H2Q_LAUNCH(false, 1);
    } else {
        if (act) GN_APPLY_LAUNCH(true);
        else GN_APPLY_LAUNCH(false);
    }
#undef GN_APPLY_LAUNCH
#undef GN_H2Q_LAUNCH
    dp_prof_end(rec, (hipStream_t)stream);
    DP_LAUNCH_CHECK("gn_apply");
    return 0;
}

// GroupNorm-apply over PLAIN fp16 tensors (the fp16 residual stream / a first convolution's fp16 output): see gn_apply_h16_kernel.
// out_fmt 2 = the zero-bordered "h1" operand [B][Ho+2][Wo+2][C]; 3 = a plain fp16 tensor [B][Ho][Wo][C].  resample: 0 | 1 (2x
// nearest up) | 2 (2x2 mean down).  gamma == NULL: no normalisation (conversion / resampling only).  y_raw (out_fmt 2): resample 0 -
// the un-normalised input in operand form; resample 1 | 2 (round 6) - the RESAMPLED un-normalised input as a plain tensor [B][Ho][Wo][C].
extern "C" int dp_gn_apply_h16(const void* x1, int C1, const void* x2, int C2, int B, int H, int W, int G, const float* stats,
                               const float* gamma, const float* beta, const float* fscale, const float* fshift, int film_stride,
                               int act, int resample, int out_fmt, void* y, void* y_raw, void* stream) {
    const int C = C1 + C2;
    DP_REQUIRE(x1 && y && B > 0 && H > 0 && W > 0, "dp_gn_apply_h16: bad args");
    DP_REQUIRE(C2 == 0 || x2, "dp_gn_apply_h16: x2 missing");
    DP_REQUIRE(C % 8 == 0 && C1 % 8 == 0, "dp_gn_apply_h16: channel counts must be multiples of 8 (C1=%d, C2=%d)", C1, C2);
    DP_REQUIRE(!gamma || (beta && stats && G > 0 && C % (4 * G) == 0), "dp_gn_apply_h16: need beta, stats and C %% (4*G) == 0");
    DP_REQUIRE((fscale == nullptr) == (fshift == nullptr), "dp_gn_apply_h16: FiLM scale and shift come together");
    DP_REQUIRE(resample >= 0 && resample <= 2, "dp_gn_apply_h16: resample mode %d (0 | 1 | 2; the FIR modes run on the fp32 stream)", resample);
    DP_REQUIRE(resample != 2 || (H % 2 == 0 && W % 2 == 0), "dp_gn_apply_h16: 2x down-sampling needs even H, W");
    DP_REQUIRE(out_fmt == 2 || out_fmt == 3, "dp_gn_apply_h16: out_fmt must be 2 (bordered fp16 operand) or 3 (plain fp16), got %d", out_fmt);
    DP_REQUIRE(!y_raw || out_fmt == 2, "dp_gn_apply_h16: the raw second output comes with the operand form (out_fmt=2)");
    DP_REQUIRE(dp_aligned16(x1) && (C2 == 0 || dp_aligned16(x2)) && dp_aligned16(y) && (!y_raw || dp_aligned16(y_raw)) &&
                   (!gamma || (dp_aligned16(gamma) && dp_aligned16(beta))), "dp_gn_apply_h16: misaligned tensor");
    DP_REQUIRE(!fscale || (film_stride % 4 == 0 && dp_aligned16(fscale) && dp_aligned16(fshift)), "dp_gn_apply_h16: misaligned FiLM rows");
    Apply16Args p{(const _Float16*)x1, (const _Float16*)x2, C1, C2, B, H, W, G, stats, gamma, beta, fscale, fshift, film_stride, resample,
                  (char*)y, (char*)y_raw, gamma ? C / G : C, resample == 1 ? 2 * H : (resample == 2 ? H / 2 : H),
                  resample == 1 ? 2 * W : (resample == 2 ? W / 2 : W)};
    const int CO = C / 8, COT = CO < 256 ? CO : 256;
    const unsigned rows = (unsigned)(B * (out_fmt == 2 ? p.Ho + 2 : p.Ho));
    // workgroups per output row (round 6): one, unless that leaves fewer than DP_GN_WG workgroups (default 2048 = eight per CU) - then the
    // row's pixels are cut across up to Wq / (slots per workgroup) of them.  Elementwise: same bits for any cut.
    int ysplit = 1;
    {
        const int want = dp_tune(DP_T_GN_WG), per_wg = 256 / COT, wq = (resample == 1 ? p.W : p.Wo) + (resample == 0 && out_fmt == 2 ? 2 : 0);
        if (want > 0 && (int)rows < want) {
            ysplit = (want + (int)rows - 1) / (int)rows;
            const int most = (wq + per_wg - 1) / per_wg;
            if (ysplit > most) ysplit = most;
            if (ysplit < 1) ysplit = 1;
        }
    }
    const int slots = (256 / COT) * ysplit;
    const dim3 g(rows, (unsigned)ysplit), blk((unsigned)(COT * (256 / COT)));
    void* rec = nullptr;
    {   // algorithmic HBM bytes: every source element the output needs once (2 B) + every output element once (2 B)
        const double in_px = resample == 1 ? (double)H * W : (double)p.Ho * p.Wo * (resample ? 4 : 1);
        const double out_px = (double)(out_fmt == 2 ? (p.Ho + 2) * (p.Wo + 2) : p.Ho * p.Wo);
        const double raw_px = !y_raw ? 0.0 : (resample ? (double)p.Ho * p.Wo : out_px);
        dp_prof_begin(DP_PROF_GN_APPLY, 0.0, (double)B * C * 2.0 * (in_px + out_px + raw_px), (hipStream_t)stream, &rec);
    }
    // non-temporal hints on the streaming accesses (round 6; same bits): a tensor far beyond the 256 MB of last-level cache gains nothing
    // from occupying it on its way through - measured (tests/probes/gn_bench.py, profiles/r06/gn_bench_nt.log): 256^2 x 256 at B = 64
    // 5.15 -> 5.35 TB/s with both hints, 64^2 x 512 4.85 -> 5.69 with the store hint alone, the small levels (whose operand the next
    // convolution finds in cache) LOSE with either; headline purification 20.77 -> 20.89 images/s at t = 20 (+0.6 %).
    // DP_GN_NT: -1 (default) by tensor size, 0 never, 1 | 2 | 3 forced (bit 0 loads, bit 1 stores).
    int ntm = dp_tune(DP_T_GN_NT);
    if (ntm < 0) {
        const double in_bytes = (double)B * H * W * C * 2.0;
        ntm = in_bytes >= 400e6 ? 3 : (in_bytes >= 128e6 ? 2 : 0);
    }
#define GN_H16_LAUNCH(ACT_, BORDER_)                                                                                                  \
    do {                                                                                                                              \
        if (resample == 0 && ntm == 1) hipLaunchKernelGGL((gn_apply_h16_kernel<ACT_, BORDER_, 0, 1>), g, blk, 0, (hipStream_t)stream, p, COT, slots);      \
        else if (resample == 0 && ntm == 2) hipLaunchKernelGGL((gn_apply_h16_kernel<ACT_, BORDER_, 0, 2>), g, blk, 0, (hipStream_t)stream, p, COT, slots); \
        else if (resample == 0 && ntm == 3) hipLaunchKernelGGL((gn_apply_h16_kernel<ACT_, BORDER_, 0, 3>), g, blk, 0, (hipStream_t)stream, p, COT, slots); \
        else if (resample == 0) hipLaunchKernelGGL((gn_apply_h16_kernel<ACT_, BORDER_, 0>), g, blk, 0, (hipStream_t)stream, p, COT, slots);      \
        else if (resample == 1) hipLaunchKernelGGL((gn_apply_h16_kernel<ACT_, BORDER_, 1>), g, blk, 0, (hipStream_t)stream, p, COT, slots); \
        else hipLaunchKernelGGL((gn_apply_h16_kernel<ACT_, BORDER_, 2>), g, blk, 0, (hipStream_t)stream, p, COT, slots);                    \
    } while (0)
    if (out_fmt == 2) {
        if (act) GN_H16_LAUNCH(true, 1);
        else GN_H16_LAUNCH(false, 1);
    } else {
        if (act) GN_H16_LAUNCH(true, 0);
        else GN_H16_LAUNCH(false, 0);
    }
#undef GN_H16_LAUNCH
    dp_prof_end(rec, (hipStream_t)stream);
    DP_LAUNCH_CHECK("gn_apply_h16");
    return 0;
}
