// Block boundary of the <= 64-pixel levels in ONE launch (round 6): the split-K reduction of the producing convolution, its epilogue
// (bias, time embedding, residual, scale), the GroupNorm statistics of the result and the normalised (+FiLM) (+SiLU) zero-bordered fp16
// operand of the NEXT convolution.
//
// Where: score_sde/models/layerspp.py:242-274 (ResnetBlockBigGANpp) at 8x8 / 4x4 - 90 of the 3x3 convolutions of one NCSN++ call -,
// guided_diffusion/unet.py:244-264 at 8x8.  Rounds 2-5 ran four launches per boundary there:
//     conv (raw split-K partial sums) -> splitk_epilogue_kernel -> gn_finalize_cols_sample / gn_stats + gn_finalize -> gn_apply
// three of them glue of 5-8 us each around convolutions of 30-60 us (3.8 + 3.5 + 2.1 % of the CIFAR step in round 5's profile, plus
// the apply launches themselves).  A whole sample's slab of such a level is small - 64 x 256 or 16 x 256 values per 256-channel
// block - so one workgroup per (sample, channel block of whole groups) holds it in REGISTERS between the reduction, the statistics
// and the apply: every byte is read once and written once.
//
// Workgroup (b, cb): 256 threads = (CB / 4 channel quads) x (256 / (CB / 4) row groups); thread (q, rg) owns channels 4q .. 4q + 3 of
// rows rg, rg + RG, ... (RPT rows).  A channel block of the FIRST source [0, N) is the convolution's output:
//     v = scale * (res + temb[b] + bias + sum_s ws[s])      s = 0 .. S-1 in this order, every operation one IEEE fp32 operation
// - the arithmetic of splitk_epilogue_kernel (igemm_h2.hip) - stored as the stream tensor (fp32, or plain fp16 rounded to nearest)
// when the caller wants it; a block of the SECOND source [N, N + C2) is the other half of a skip concatenation (th.cat of
// ncsnpp.py:325 / unet.py:667, never materialised), read as it is stored.  GroupNorm statistics: per thread fp32 partial sums of its
// <= 64 values, then a fixed-order sum in double over the group's threads (through LDS), mean / rstd as gn_finalize forms them - of the
// UNROUNDED values, as the column records of the un-fused path are.  The value that is normalised is the one the stream STORES (the
// fp16-rounded one where the stream is fp16): what every later reader of the tensor - the backward pass included - sees.
// The split factor, the channel block and the reduction orders are functions of the layer shape only (never of the batch).
#include "dp_tune.h"
#include "igemm_h2.h"

#pragma clang fp contract(off)

namespace {

struct SkGnArgs {
    const float* ws;      // [S][M][N] raw partial sums of the convolution (M = B * H * W)
    int S;
    const float* bias;    // [N] or null
    const float* temb;    // [B | 1][temb_stride] or null
    int temb_stride;
    const void* res;      // [M][N] fp32 (rfmt 0) / plain fp16 (rfmt 1) or null
    int rfmt;
    float scale;
    void* out;            // [M][N] fp32 (OUT16 false) / plain fp16 (OUT16 true) or null: the stream tensor
    float* colstats;      // [B][2][N]: the 64-row column records of `out` (H * W == 64 only) or null
    const void* x2;       // second source [B][H*W][C2], fp32 (x2fmt 0) / plain fp16 (x2fmt 1), or null
    int x2fmt, C2;
    int B, H, W, N, G;
    float eps;
    const float* gamma;   // [N + C2]
    const float* beta;
    const float* fscale;  // FiLM rows [B | 1][film_stride] or null
    const float* fshift;
    int film_stride;
    int act;
    float* stats;         // [B][G][2] (mean, rstd) or null
    _Float16* y;          // [B][H+2][W+2][N + C2]: the operand, zero border included
    _Float16* y_raw;      // same shape: the UN-normalised cat(out, x2) as an operand (input of a 1x1 shortcut) or null
};

template <bool OUT16, int RPT>
__global__ __launch_bounds__(256) void splitk_gn_kernel(SkGnArgs p, int CB) {
    __shared__ float red[2][256];
    __shared__ float gst[64][2];
    __shared__ float colp[2][8][256];        // column records: [sum | sumsq][row group][channel of the block]
    const int tid = threadIdx.x;
    const int HW = p.H * p.W, C = p.N + p.C2, cpg = C / p.G;
    const int quads = CB >> 2, RG = 256 / quads;
    const int q = tid % quads, rg = tid / quads;
    const int b = blockIdx.y, c0 = blockIdx.x * CB + q * 4;      // first of the thread's four channels, in the concatenated tensor
    const bool second = c0 >= p.N;
    float v[RPT][4];        // the values as the stream stores them (fp16-rounded where it is fp16)
    float s = 0.f, qq = 0.f;
    float cs[4] = {0.f, 0.f, 0.f, 0.f}, cq[4] = {0.f, 0.f, 0.f, 0.f};
    if (!second) {
        const size_t MN = (size_t)p.B * HW * p.N;
        f32x4 bv = {0.f, 0.f, 0.f, 0.f}, tv = {0.f, 0.f, 0.f, 0.f};
        if (p.bias) bv = f32x4{p.bias[c0], p.bias[c0 + 1], p.bias[c0 + 2], p.bias[c0 + 3]};       // (rows of wider tables: scalar loads)
        if (p.temb) {
            const float* t = p.temb + (size_t)b * p.temb_stride + c0;
            tv = f32x4{t[0], t[1], t[2], t[3]};
        }
        f32x4 acc[RPT];
#pragma unroll
        for (int k = 0; k < RPT; ++k) acc[k] = *reinterpret_cast<const f32x4*>(p.ws + ((size_t)b * HW + rg + k * RG) * p.N + c0);
        for (int sp = 1; sp < p.S; ++sp) {
#pragma unroll
            for (int k = 0; k < RPT; ++k) {
                const f32x4 w = *reinterpret_cast<const f32x4*>(p.ws + sp * MN + ((size_t)b * HW + rg + k * RG) * p.N + c0);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[k][j] += w[j];
            }
        }
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            const size_t row = (size_t)b * HW + rg + k * RG;
            f32x4 r = {0.f, 0.f, 0.f, 0.f};
            if (p.res) {
                if (p.rfmt) {
                    const dp_half4 h = *reinterpret_cast<const dp_half4*>(reinterpret_cast<const _Float16*>(p.res) + row * p.N + c0);
#pragma unroll
                    for (int j = 0; j < 4; ++j) r[j] = (float)h[j];
                } else {
                    r = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.res) + row * p.N + c0);
                }
            }
            dp_half4 o16;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float u = acc[k][j] + bv[j];
                if (p.temb) u += tv[j];
                if (p.res) u += r[j];
                u *= p.scale;
                s += u;
                qq += u * u;
                cs[j] += u;
                cq[j] += u * u;
                if constexpr (OUT16) {
                    o16[j] = dp_to_half(u);
                    v[k][j] = (float)o16[j];
                } else {
                    v[k][j] = u;
                }
            }
            if (p.out) {
                if constexpr (OUT16) *reinterpret_cast<dp_half4*>(reinterpret_cast<_Float16*>(p.out) + row * p.N + c0) = o16;
                else *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + row * p.N + c0) = f32x4{v[k][0], v[k][1], v[k][2], v[k][3]};
            }
        }
    } else {
        const int c2 = c0 - p.N;
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            const size_t row = (size_t)b * HW + rg + k * RG;
            if (p.x2fmt) {
                const dp_half4 h = *reinterpret_cast<const dp_half4*>(reinterpret_cast<const _Float16*>(p.x2) + row * p.C2 + c2);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[k][j] = (float)h[j];
            } else {
                const f32x4 f = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.x2) + row * p.C2 + c2);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[k][j] = f[j];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s += v[k][j];
                qq += v[k][j] * v[k][j];
            }
        }
    }
    // ---- statistics of the groups of this channel block: fixed-order sum in double over each group's threads
    red[0][tid] = s;
    red[1][tid] = qq;
    const bool want_cols = p.colstats && !second;
    if (want_cols) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            colp[0][rg][q * 4 + j] = cs[j];
            colp[1][rg][q * 4 + j] = cq[j];
        }
    }
    __syncthreads();
    const int qpg = cpg >> 2, gpb = CB / cpg;          // quads per group, groups per channel block
    if (tid < gpb) {
        double ds = 0.0, dq = 0.0;
        for (int r = 0; r < RG; ++r)
            for (int i = 0; i < qpg; ++i) {
                ds += (double)red[0][r * quads + tid * qpg + i];
                dq += (double)red[1][r * quads + tid * qpg + i];
            }
        const double inv = 1.0 / ((double)HW * cpg);
        const double mean = ds * inv;
        double var = dq * inv - mean * mean;
        if (var < 0.0) var = 0.0;
        const float m = (float)mean, rs = (float)(1.0 / sqrt(var + (double)p.eps));
        gst[tid][0] = m;
        gst[tid][1] = rs;
        if (p.stats) {
            float* d = p.stats + ((size_t)b * p.G + (blockIdx.x * CB) / cpg + tid) * 2;
            d[0] = m;
            d[1] = rs;
        }
    }
    if (want_cols && tid < CB) {      // the sample's one 64-row record: the row groups' partial sums in the order 0 .. RG-1
        float a0 = colp[0][0][tid], a1 = colp[1][0][tid];
        for (int r = 1; r < RG; ++r) {
            a0 += colp[0][r][tid];
            a1 += colp[1][r][tid];
        }
        float* d = p.colstats + (size_t)b * 2 * p.N + blockIdx.x * CB + tid;
        d[0] = a0;
        d[p.N] = a1;
    }
    __syncthreads();
    // ---- apply: y = act(FiLM(GroupNorm(v))) as the zero-bordered fp16 operand
    const int gl = (q * 4) / cpg;
    const float mean = gst[gl][0], rstd = gst[gl][1];
    float a[4], d[4];
    {
        const f32x4 ga = *reinterpret_cast<const f32x4*>(p.gamma + c0);
        const f32x4 be = *reinterpret_cast<const f32x4*>(p.beta + c0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            a[j] = rstd * ga[j];
            d[j] = be[j] - mean * a[j];
        }
        if (p.fscale) {
            const f32x4 fs = *reinterpret_cast<const f32x4*>(p.fscale + (size_t)b * p.film_stride + c0);
            const f32x4 fh = *reinterpret_cast<const f32x4*>(p.fshift + (size_t)b * p.film_stride + c0);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float m = 1.f + fs[j];
                a[j] *= m;
                d[j] = d[j] * m + fh[j];
            }
        }
    }
    const int Wq = p.W + 2;
    _Float16* yb = p.y + (size_t)b * (p.H + 2) * Wq * C + c0;
    _Float16* rb = p.y_raw ? p.y_raw + (size_t)b * (p.H + 2) * Wq * C + c0 : nullptr;
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const int r = rg + k * RG, py = r / p.W, px = r - py * p.W;
        const size_t o = ((size_t)(py + 1) * Wq + px + 1) * C;
        dp_half4 h, hr;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float u = fmaf(v[k][j], a[j], d[j]);
            h[j] = dp_to_half(p.act ? (OUT16 ? dp_silu_fast_f(u) : dp_silu_f(u)) : u);
            hr[j] = dp_to_half(v[k][j]);
        }
        *reinterpret_cast<dp_half4*>(yb + o) = h;
        if (rb) *reinterpret_cast<dp_half4*>(rb + o) = hr;
    }
    // ---- the zero border of this channel block: top and bottom rows, left and right columns
    const dp_half4 z = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
    const int nb = 2 * Wq + 2 * p.H;
    for (int i = rg; i < nb; i += RG) {
        int py, px;
        if (i < Wq) { py = 0; px = i; }
        else if (i < 2 * Wq) { py = p.H + 1; px = i - Wq; }
        else { const int t = i - 2 * Wq; py = 1 + (t >> 1); px = (t & 1) ? p.W + 1 : 0; }
        const size_t o = ((size_t)py * Wq + px) * C;
        *reinterpret_cast<dp_half4*>(yb + o) = z;
        if (rb) *reinterpret_cast<dp_half4*>(rb + o) = z;
    }
}

int skgn_block(int N, int C2) { return (N % 256 == 0 && C2 % 256 == 0) ? 256 : 128; }

}  // namespace

// Does the fused boundary serve a (convolution output [B][H][W][N], optional second source of C2 channels, GroupNorm of G groups)?  A
// function of the layer shape only.
extern "C" int dp_splitk_gn_ok(int H, int W, int N, int C2, int G) {
    if (H <= 0 || W <= 0 || N <= 0 || C2 < 0 || G <= 0) return 0;
    const int HW = H * W, C = N + C2;
    if (HW != 64 && HW != 16) return 0;
    if (C % G != 0) return 0;
    const int cpg = C / G, CB = skgn_block(N, C2);
    if (cpg % 4 != 0 || N % CB != 0 || C2 % CB != 0 || CB % cpg != 0 || CB / cpg > 64) return 0;
    return 1;
}

extern "C" int dp_splitk_gn(const float* ws, int S, int B, int H, int W, int N, const float* bias, const float* temb, int temb_stride,
                            const void* res, int res_fmt, float scale, void* out, int out_fmt, float* colstats, const void* x2, int x2_fmt,
                            int C2, int G, float eps, const float* gamma, const float* beta, const float* fscale, const float* fshift,
                            int film_stride, int act, float* stats, void* y, void* y_raw, void* stream) {
    DP_REQUIRE(ws && S >= 1 && y && gamma && beta, "dp_splitk_gn: null pointer / no partial sums");
    DP_REQUIRE(dp_splitk_gn_ok(H, W, N, C2, G), "dp_splitk_gn: shape H=%d W=%d N=%d C2=%d G=%d is not served (dp_splitk_gn_ok)", H, W, N, C2, G);
    DP_REQUIRE(B > 0 && (long long)B * H * W * (N > C2 ? N : C2) < (1ll << 31), "dp_splitk_gn: bad batch");
    DP_REQUIRE((C2 == 0) == (x2 == nullptr), "dp_splitk_gn: x2 and C2 come together");
    DP_REQUIRE(out_fmt == 0 || out_fmt == 1, "dp_splitk_gn: out_fmt %d (0 = fp32, 1 = plain fp16)", out_fmt);
    DP_REQUIRE(!colstats || H * W == 64, "dp_splitk_gn: column records exist for 64-pixel samples only (one 64-row record per sample)");
    DP_REQUIRE((fscale == nullptr) == (fshift == nullptr), "dp_splitk_gn: FiLM scale and shift come together");
    DP_REQUIRE(dp_aligned16(ws) && dp_aligned16(gamma) && dp_aligned16(beta) && ((size_t)y & 7) == 0 && (!y_raw || ((size_t)y_raw & 7) == 0) &&
                   (!out || dp_aligned16(out)) && (!res || ((size_t)res & (res_fmt ? 7 : 15)) == 0) && (!x2 || ((size_t)x2 & (x2_fmt ? 7 : 15)) == 0),
               "dp_splitk_gn: misaligned tensor");
    DP_REQUIRE(!fscale || (film_stride % 4 == 0 && dp_aligned16(fscale) && dp_aligned16(fshift)), "dp_splitk_gn: misaligned FiLM rows");
    SkGnArgs p{ws, S, bias, temb, temb_stride, res, res_fmt, scale, out, colstats, x2, x2_fmt, C2, B, H, W, N, G, eps, gamma, beta, fscale, fshift,
               film_stride, act, stats, static_cast<_Float16*>(y), static_cast<_Float16*>(y_raw)};
    const int CB = skgn_block(N, C2), HW = H * W, rpt = HW / (256 / (CB / 4));
    const dim3 grid((unsigned)((N + C2) / CB), (unsigned)B), blk(256);
    hipStream_t s = static_cast<hipStream_t>(stream);
    void* rec = nullptr;
    dp_prof_begin(DP_PROF_GN_APPLY, 0.0, (double)B * HW * (4.0 * S * N + (res ? (res_fmt ? 2.0 : 4.0) : 0.0) * N + (out ? (out_fmt ? 2.0 : 4.0) : 0.0) * N +
                                                            (x2_fmt ? 2.0 : 4.0) * C2) + (double)B * (H + 2) * (W + 2) * (N + C2) * 2.0 * (y_raw ? 2 : 1), s, &rec);
#define SKGN_LAUNCH(RPT_)                                                                              \
    do {                                                                                               \
        if (out_fmt) hipLaunchKernelGGL((splitk_gn_kernel<true, RPT_>), grid, blk, 0, s, p, CB);       \
        else hipLaunchKernelGGL((splitk_gn_kernel<false, RPT_>), grid, blk, 0, s, p, CB);              \
    } while (0)
    switch (rpt) {
        case 16: SKGN_LAUNCH(16); break;
        case 8: SKGN_LAUNCH(8); break;
        case 4: SKGN_LAUNCH(4); break;
        case 2: SKGN_LAUNCH(2); break;
        default: dp_set_error("dp_splitk_gn: internal rows-per-thread %d", rpt); return 1;
    }
#undef SKGN_LAUNCH
    dp_prof_end(rec, s);
    DP_LAUNCH_CHECK("splitk_gn");
    return 0;
}
