// HBM-bound elementwise / row kernels of the purification loop: solver steps (fused drift +
// diffusion + in-kernel Philox noise), SiLU, axpby, sinusoidal embedding, row softmax.
// All are float4-vectorised, grid-stride, and sized to ~2048 workgroups (256 CUs x 8).
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>

#include "dp_common.h"
#include "dp_tune.h"

// ---- error plumbing (shared by every translation unit) ----------------------------------------
static thread_local char g_dp_err[512] = "";
void dp_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_dp_err, sizeof(g_dp_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* dp_last_error(void) { return g_dp_err; }
extern "C" int dp_abi_version(void) { return 8; }

// ---- tuning switches (dp_tune.h): environment read once, then only dp_set_tuning() changes a value -----------------
namespace {
struct TuneEntry { const char* name; int def; };
const TuneEntry kTune[DP_T_COUNT] = {
    {"DP_H2_PP", 2}, {"DP_H2_SW", 2}, {"DP_H2_NN", 1}, {"DP_H2_DW", 2}, {"DP_H2_DH", 2}, {"DP_H2_DH_MIN", 32}, {"DP_GN_FINALIZE_SAMPLE", 1},
    {"DIFFPURE_BATCH_INVARIANT", 0}, {"DP_GN_NT", -1}, {"DP_GN_WG", 2048}, {"DP_GNB_NT", -1}, {"DP_GNB_LEAN", 1}, {"DP_XCD_MAP", 1},
};
std::atomic<int> g_tune[DP_T_COUNT];       // relaxed: read on every launch, possibly from several host threads (DataParallel replicas)
std::once_flag g_tune_once;
void tune_load() {
    for (int k = 0; k < DP_T_COUNT; ++k) {
        const char* e = getenv(kTune[k].name);
        g_tune[k].store(e ? atoi(e) : kTune[k].def, std::memory_order_relaxed);
    }
}
}  // namespace
int dp_tune(DpTune k) {
    std::call_once(g_tune_once, tune_load);
    return g_tune[k].load(std::memory_order_relaxed);
}
extern "C" int dp_set_tuning(const char* name, int value) {
    std::call_once(g_tune_once, tune_load);
    for (int k = 0; k < DP_T_COUNT; ++k)
        if (name && strcmp(name, kTune[k].name) == 0) {
            g_tune[k].store(value, std::memory_order_relaxed);
            return 0;
        }
    dp_set_error("dp_set_tuning: unknown switch '%s'", name ? name : "(null)");
    return 1;
}
extern "C" int dp_get_tuning(const char* name, int* value) {
    std::call_once(g_tune_once, tune_load);
    for (int k = 0; k < DP_T_COUNT; ++k)
        if (name && value && strcmp(name, kTune[k].name) == 0) {
            *value = g_tune[k].load(std::memory_order_relaxed);
            return 0;
        }
    dp_set_error("dp_get_tuning: unknown switch '%s'", name ? name : "(null)");
    return 1;
}

namespace {

inline unsigned grid_for(long long work_items, int block = 256, int cap = 2048) {
    long long g = (work_items + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (unsigned)g;
}

// ---- Philox4x32-10 (Salmon et al. 2011) + Box-Muller ---------------------------------------------
struct U4 { uint32_t x, y, z, w; };

__device__ __forceinline__ U4 philox4x32_10(U4 c, uint32_t k0, uint32_t k1) {
    constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(M0, c.x), lo0 = M0 * c.x;
        const uint32_t hi1 = __umulhi(M1, c.z), lo1 = M1 * c.z;
        c = U4{hi1 ^ c.y ^ k0, lo1, hi0 ^ c.w ^ k1, lo0};
        k0 += W0;
        k1 += W1;
    }
    return c;
}

__device__ __forceinline__ float u01(uint32_t r) {  // (0,1), 24 random bits, never 0 or 1
    return ((float)(r >> 8) + 0.5f) * (1.0f / 16777216.0f);
}

// 4 standard normals for element quad q of (sample, step) under `seed`
__device__ __forceinline__ f32x4 philox_normal4(unsigned long long seed, long long sample, int step, uint32_t q) {
    U4 c{q, (uint32_t)(step + 1), (uint32_t)((unsigned long long)sample & 0xffffffffu),
         (uint32_t)((unsigned long long)sample >> 32)};
    const U4 r = philox4x32_10(c, (uint32_t)(seed & 0xffffffffu), (uint32_t)(seed >> 32));
    const float r0 = sqrtf(-2.0f * logf(u01(r.x))), r1 = sqrtf(-2.0f * logf(u01(r.z)));
    const float a0 = 6.283185307179586f * u01(r.y), a1 = 6.283185307179586f * u01(r.w);
    float s0, c0, s1, c1;
    sincosf(a0, &s0, &c0);
    sincosf(a1, &s1, &c1);
    return f32x4{r0 * c0, r0 * s0, r1 * c1, r1 * s1};
}

__global__ void philox_normal_kernel(float* out, int B, long long per_sample, unsigned long long seed,
                                     long long sample0, int step) {
    const long long q_per = per_sample / 4, total = (long long)B * q_per;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long b = i / q_per, q = i - b * q_per;
        *reinterpret_cast<f32x4*>(out + b * per_sample + q * 4) = philox_normal4(seed, sample0 + b, step, (uint32_t)q);
    }
}

// ---- solver steps ----------------------------------------------------------------------------
// One thread handles 4 consecutive state elements (flat index within a sample: e = pix*C + c).
// eps is read at [pix][eps_ld] channel c (first C channels of the network output).
struct EmArgs {
    const float* x;
    const float* eps;
    const float* noise;
    float* x_out;
    int eps_ld, B, HW, C;
    float nhb, gg, sc, h, g, sqrt_h;
    int score_div;
    unsigned long long seed;
    long long sample0;
    int step;
};

__global__ void em_step_kernel(EmArgs p) {
    const long long per = (long long)p.HW * p.C, q_per = per / 4, total = (long long)p.B * q_per;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long b = i / q_per, q = i - b * q_per;
        const long long base = b * per + q * 4;
        const f32x4 xv = *reinterpret_cast<const f32x4*>(p.x + base);
        f32x4 ev;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long long e = q * 4 + j;
            const long long pix = e / p.C;
            const int c = (int)(e - pix * p.C);
            ev[j] = p.eps[(b * p.HW + pix) * p.eps_ld + c];
        }
        f32x4 z = {0.f, 0.f, 0.f, 0.f};
        if (p.g != 0.f) {
            z = p.noise ? *reinterpret_cast<const f32x4*>(p.noise + base)
                        : philox_normal4(p.seed, p.sample0 + b, p.step, (uint32_t)q);
        }
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float score = p.score_div ? (-ev[j]) / p.sc : p.sc * ev[j];
            const float drift = p.nhb * xv[j] - p.gg * score;
            o[j] = xv[j] + (-drift) * p.h + p.g * (z[j] * p.sqrt_h);
        }
        *reinterpret_cast<f32x4*>(p.x_out + base) = o;
    }
}

struct DdpmArgs {
    const float* x;
    const float* out6;
    const float* noise;
    float* x_out;
    int B, HW, C;
    float sr, srm1, c1, c2, min_log, max_log;
    int nonzero;
    unsigned long long seed;
    long long sample0;
    int step;
};

__global__ void ddpm_step_kernel(DdpmArgs p) {
    const long long per = (long long)p.HW * p.C, q_per = per / 4, total = (long long)p.B * q_per;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long b = i / q_per, q = i - b * q_per;
        const long long base = b * per + q * 4;
        const f32x4 xv = *reinterpret_cast<const f32x4*>(p.x + base);
        f32x4 z = {0.f, 0.f, 0.f, 0.f};
        if (p.nonzero) {
            z = p.noise ? *reinterpret_cast<const f32x4*>(p.noise + base)
                        : philox_normal4(p.seed, p.sample0 + b, p.step, (uint32_t)q);
        }
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long long e = q * 4 + j;
            const long long pix = e / p.C;
            const int c = (int)(e - pix * p.C);
            const float* row = p.out6 + (b * p.HW + pix) * (2 * p.C);
            const float eps = row[c], v = row[p.C + c];
            const float frac = (v + 1.f) / 2.f;
            const float logvar = frac * p.max_log + (1.f - frac) * p.min_log;
            float x0 = p.sr * xv[j] - p.srm1 * eps;
            x0 = fminf(fmaxf(x0, -1.f), 1.f);
            const float mean = p.c1 * x0 + p.c2 * xv[j];
            o[j] = mean + (p.nonzero ? 1.f : 0.f) * expf(0.5f * logvar) * z[j];
        }
        *reinterpret_cast<f32x4*>(p.x_out + base) = o;
    }
}

// ---- misc ------------------------------------------------------------------------------------
__global__ void silu_kernel(const float* x, float* y, long long n) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        y[i] = dp_silu_f(x[i]);
}
__global__ void axpby_kernel(const float* x, float a, const float* y, float b, float* out, long long n) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        out[i] = x[i] * a + y[i] * b;
}
__global__ void temb_kernel(const float* t, int n, const float* freqs, int half, int cos_first, float* emb) {
    const int total = n * half;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int r = i / half, f = i - r * half;
        const float a = t[r] * freqs[f];
        float s, c;
        sincosf(a, &s, &c);
        float* row = emb + (size_t)r * 2 * half;
        row[f] = cos_first ? c : s;
        row[half + f] = cos_first ? s : c;
    }
}

// one wave per row; 4 rows per 256-thread workgroup
__global__ void softmax_rows_kernel(float* x, long long rows, int cols) {
    const int lane = threadIdx.x & 63;
    const long long row = blockIdx.x * 4ll + (threadIdx.x >> 6);
    if (row >= rows) return;
    float* r = x + row * cols;
    float m = -INFINITY;
    for (int c = lane; c < cols; c += 64) m = fmaxf(m, r[c]);
    m = wave_max(m);
    float s = 0.f;
    for (int c = lane; c < cols; c += 64) {
        const float e = expf(r[c] - m);
        r[c] = e;
        s += e;
    }
    s = wave_sum(s);
    const float inv = 1.0f / s;
    for (int c = lane; c < cols; c += 64) r[c] *= inv;
}

// The same pass with the row held in registers (round 6): cols = 64 NE, lane l owns columns l, l + 64, ... exactly as above - the same
// per-lane order of every max / exp / sum, hence the same bits - but the row is read ONCE (all NE loads in flight together) and written
// once, where the three-sweep form issues 3 NE loads + 2 NE stores per lane (T = 1 024: 2.5 TB/s of row traffic through L2).
template <int NE>
__global__ void softmax_rows_reg_kernel(float* x, long long rows) {
    const int lane = threadIdx.x & 63;
    const long long row = blockIdx.x * 4ll + (threadIdx.x >> 6);
    if (row >= rows) return;
    float* r = x + row * (64 * NE);
    float v[NE];
#pragma unroll
    for (int k = 0; k < NE; ++k) v[k] = r[lane + 64 * k];
    float m = -INFINITY;
#pragma unroll
    for (int k = 0; k < NE; ++k) m = fmaxf(m, v[k]);
    m = wave_max(m);
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NE; ++k) {
        v[k] = expf(v[k] - m);
        s += v[k];
    }
    s = wave_sum(s);
    const float inv = 1.0f / s;
#pragma unroll
    for (int k = 0; k < NE; ++k) r[lane + 64 * k] = v[k] * inv;
}

}  // namespace

extern "C" int dp_silu(const float* x, float* y, long long n, void* stream) {
    DP_REQUIRE(x && y && n >= 0, "dp_silu: bad args");
    if (n == 0) return 0;
    hipLaunchKernelGGL(silu_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, y, n);
    DP_LAUNCH_CHECK("silu");
    return 0;
}

extern "C" int dp_axpby(const float* x, float a, const float* y, float b, float* out, long long n, void* stream) {
    DP_REQUIRE(x && y && out && n >= 0, "dp_axpby: bad args");
    if (n == 0) return 0;
    hipLaunchKernelGGL(axpby_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, a, y, b, out, n);
    DP_LAUNCH_CHECK("axpby");
    return 0;
}

extern "C" int dp_timestep_embedding(const float* t, int n, const float* freqs, int half, int cos_first, float* emb,
                                     void* stream) {
    DP_REQUIRE(t && freqs && emb && n > 0 && half > 0, "dp_timestep_embedding: bad args");
    hipLaunchKernelGGL(temb_kernel, dim3(grid_for((long long)n * half)), dim3(256), 0, (hipStream_t)stream, t, n,
                       freqs, half, cos_first, emb);
    DP_LAUNCH_CHECK("timestep_embedding");
    return 0;
}

extern "C" int dp_softmax_rows(float* x, long long rows, int cols, void* stream) {
    DP_REQUIRE(x && rows > 0 && cols > 0, "dp_softmax_rows: bad args");
    const long long grid = (rows + 3) / 4;
    DP_REQUIRE(grid < (1ll << 31), "dp_softmax_rows: too many rows");
    if (cols == 1024) hipLaunchKernelGGL(softmax_rows_reg_kernel<16>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, x, rows);
    else if (cols == 256) hipLaunchKernelGGL(softmax_rows_reg_kernel<4>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, x, rows);
    else if (cols == 64) hipLaunchKernelGGL(softmax_rows_reg_kernel<1>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, x, rows);
    else hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, x, rows, cols);
    DP_LAUNCH_CHECK("softmax_rows");
    return 0;
}

// fp32 -> fp16 over a flat weight buffer; STOCH: unbiased stochastic rounding.  h0 = the fp16 neighbour towards zero,
// h1 = the next one away from zero (bit pattern + 1: fp16 magnitudes are ordered like their bit patterns, through the
// subnormal range as well); w goes to h1 with probability (|w| - |h0|) / (|h1| - |h0|), decided by 24 Philox bits keyed by
// (seed, key, element quad).  A value beyond the largest finite fp16 number stays at it (weights never get there).
template <bool STOCH>
__global__ void round_weights_kernel(const float* __restrict__ src, _Float16* __restrict__ dst, long long nquads,
                                     unsigned long long seed, long long key) {
    for (long long q = blockIdx.x * (long long)blockDim.x + threadIdx.x; q < nquads; q += (long long)gridDim.x * blockDim.x) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(src + q * 4);
        typedef _Float16 half4 __attribute__((ext_vector_type(4)));
        half4 o;
        if constexpr (STOCH) {
            U4 c{(uint32_t)((unsigned long long)q & 0xffffffffu), (uint32_t)((unsigned long long)q >> 32),
                 (uint32_t)((unsigned long long)key & 0xffffffffu), (uint32_t)((unsigned long long)key >> 32)};
            const U4 r = philox4x32_10(c, (uint32_t)(seed & 0xffffffffu), (uint32_t)(seed >> 32));
            const uint32_t rr[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float w = v[j], aw = fabsf(w);
                const _Float16 hn = (_Float16)aw;                               // nearest; step back if it overshot: towards zero
                unsigned short b0 = __builtin_bit_cast(unsigned short, hn);
                if ((float)hn > aw) --b0;                                       // (aw >= 0: hn > aw implies b0 >= 1; inf steps to 65504)
                const _Float16 h0 = __builtin_bit_cast(_Float16, b0);
                const float f0 = (float)h0;
                _Float16 h = h0;
                if (aw > f0 && b0 < 0x7bff) {                                   // not exactly representable, not at the top
                    const _Float16 h1 = __builtin_bit_cast(_Float16, (unsigned short)(b0 + 1));
                    // (h1 - f0) is the spacing of fp16 at this binade: a power of two between 2^-24 and 2^5, so dividing by it is exact and so is
                    // multiplying by its reciprocal, built from the exponent field (round 6: the IEEE division was ~10 of the ~40 vector
                    // instructions per element of a kernel that runs at 4.1 TB/s; same bits)
                    const float ulp = (float)h1 - f0;
                    const float inv_ulp = __uint_as_float((254u << 23) - (__float_as_uint(ulp) & 0x7f800000u));
                    const float p = (aw - f0) * inv_ulp;                        // in (0, 1)
                    if (u01(rr[j]) < p) h = h1;
                }
                o[j] = w < 0.f ? -h : h;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (_Float16)v[j];
        }
        *reinterpret_cast<half4*>(dst + q * 4) = o;
    }
}

extern "C" int dp_round_weights(const float* src, void* dst, long long n, int stochastic, unsigned long long seed, long long key,
                                void* stream) {
    DP_REQUIRE(src && dst && n > 0 && n % 8 == 0 && dp_aligned16(src) && dp_aligned16(dst), "dp_round_weights: n must be a positive multiple of 8, buffers 16-byte aligned");
    const long long nq = n / 4;
    const unsigned grid = grid_for(nq);
    if (stochastic) hipLaunchKernelGGL(round_weights_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, (_Float16*)dst, nq, seed, key);
    else hipLaunchKernelGGL(round_weights_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, (_Float16*)dst, nq, seed, key);
    DP_LAUNCH_CHECK("round_weights");
    return 0;
}

extern "C" int dp_philox_normal(float* out, int B, long long per_sample, unsigned long long seed, long long sample0,
                                int step, void* stream) {
    DP_REQUIRE(out && B > 0 && per_sample > 0 && per_sample % 4 == 0, "dp_philox_normal: per_sample must be a positive multiple of 4");
    DP_REQUIRE(per_sample / 4 < (1ll << 32) && dp_aligned16(out), "dp_philox_normal: sample too large or misaligned");
    hipLaunchKernelGGL(philox_normal_kernel, dim3(grid_for(B * (per_sample / 4))), dim3(256), 0, (hipStream_t)stream,
                       out, B, per_sample, seed, sample0, step);
    DP_LAUNCH_CHECK("philox_normal");
    return 0;
}

extern "C" int dp_em_step(const float* x, const float* eps, int eps_ld, int B, int HW, int C, float neg_half_beta,
                          float gg, float score_coef, int score_div, float h, float g, float sqrt_h,
                          const float* noise, unsigned long long seed, long long sample0, int step, float* x_out,
                          void* stream) {
    DP_REQUIRE(x && eps && x_out && B > 0 && HW > 0 && C > 0 && eps_ld >= C, "dp_em_step: bad args");
    DP_REQUIRE(((long long)HW * C) % 4 == 0, "dp_em_step: HW*C must be a multiple of 4");
    DP_REQUIRE(dp_aligned16(x) && dp_aligned16(x_out) && (!noise || dp_aligned16(noise)), "dp_em_step: misaligned state");
    EmArgs p{x, eps, noise, x_out, eps_ld, B, HW, C, neg_half_beta, gg, score_coef, h, g, sqrt_h, score_div, seed, sample0, step};
    hipLaunchKernelGGL(em_step_kernel, dim3(grid_for((long long)B * HW * C / 4)), dim3(256), 0, (hipStream_t)stream, p);
    DP_LAUNCH_CHECK("em_step");
    return 0;
}

extern "C" int dp_ddpm_step(const float* x, const float* out6, int B, int HW, int C, float sqrt_recip_ac,
                            float sqrt_recipm1_ac, float coef1, float coef2, float min_log, float max_log, int nonzero,
                            const float* noise, unsigned long long seed, long long sample0, int step, float* x_out,
                            void* stream) {
    DP_REQUIRE(x && out6 && x_out && B > 0 && HW > 0 && C > 0, "dp_ddpm_step: bad args");
    DP_REQUIRE(((long long)HW * C) % 4 == 0, "dp_ddpm_step: HW*C must be a multiple of 4");
    DP_REQUIRE(dp_aligned16(x) && dp_aligned16(x_out) && (!noise || dp_aligned16(noise)), "dp_ddpm_step: misaligned state");
    DdpmArgs p{x, out6, noise, x_out, B, HW, C, sqrt_recip_ac, sqrt_recipm1_ac, coef1, coef2, min_log, max_log, nonzero, seed, sample0, step};
    hipLaunchKernelGGL(ddpm_step_kernel, dim3(grid_for((long long)B * HW * C / 4)), dim3(256), 0, (hipStream_t)stream, p);
    DP_LAUNCH_CHECK("ddpm_step");
    return 0;
}
