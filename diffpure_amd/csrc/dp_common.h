// Shared helpers for libdiffpure_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/diffpure_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

void dp_set_error(const char* fmt, ...);
// per-launch hipEvent profiling (igemm.hip); rec is opaque
void dp_prof_begin(int kind, double flop, double bytes, hipStream_t s, void** rec);
void dp_prof_end(void* rec, hipStream_t s);
void dp_prof_set_kind(void* rec, int kind);

#define DP_REQUIRE(cond, ...)        \
    do {                             \
        if (!(cond)) {               \
            dp_set_error(__VA_ARGS__); \
            return 1;                \
        }                            \
    } while (0)

#define DP_LAUNCH_CHECK(name)                                              \
    do {                                                                   \
        hipError_t e_ = hipGetLastError();                                 \
        if (e_ != hipSuccess) {                                            \
            dp_set_error("%s launch failed: %s", name, hipGetErrorString(e_)); \
            return 2;                                                      \
        }                                                                  \
    } while (0)

static inline bool dp_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// exp(x) in ~7 instructions, ~2 ulp: exp2 on the hardware unit with the rounding error of x * log2(e) carried to
// first order (libm's expf + an IEEE division cost ~25 VALU instructions per SiLU - enough to make GroupNorm-apply
// instruction-bound instead of HBM-bound).  Overflow -> inf, underflow -> 0 as exp2 gives them.
__device__ __forceinline__ float dp_exp_f(float x) {
    const float t = x * 1.44269504088896341f;
    const float r = fmaf(x, 1.44269504088896341f, -t) + x * 1.92596299112661746e-8f;   // low bits of x * log2(e)
    const float e = __builtin_amdgcn_exp2f(t);
    // e == 0 or inf: the argument is out of range and r may be inf - inf; the limit value is e itself
    return (e > 0.f && e < 3.0e38f) ? fmaf(e, r * 0.693147180559945309f, e) : e;
}
// 1 / (1 + exp(-v)): the reciprocal unit is within 1 ulp, one Newton step makes it correctly rounded for practical purposes
__device__ __forceinline__ float dp_sigmoid_f(float v) {
    const float d = 1.0f + dp_exp_f(-v);
    float r = __builtin_amdgcn_rcpf(d);
    r = fmaf(fmaf(-d, r, 1.0f), r, r);
    return d > 3.0e38f ? 0.f : r;          // d = inf: rcp gives 0 but the Newton step would make NaN (inf * 0)
}
// (the product is pinned in a register: left free, the compiler contracts it into whatever consumes it - an add of the 2x2 mean
//  resampler, the fp16 conversion - in one kernel and not in its twin, and results that must agree bit for bit differ in the last place)
__device__ __forceinline__ float dp_silu_f(float v) {
    float r = v * dp_sigmoid_f(v);
    asm volatile("" : "+v"(r));
    return r;
}

// SiLU for results that are rounded to fp16 right away (GroupNorm-apply over the fp16 residual stream): exp2 of v * -log2(e) straight on the
// hardware unit (relative error ~|v| 2^-24, 1e-6 at |v| = 20) and the reciprocal unit's 1 ulp - 2^-11 is what survives the store.  The full
// dp_silu_f costs ~30 issue slots per element (two quarter-rate transcendentals, the carried rounding error of the exponent, a Newton step,
// range selects), which made that kernel - 4 bytes of HBM traffic per element - instruction-bound: ~12 slots here.
// v -> +inf: v.  Very negative FINITE v (below about -89): exp2 overflows to inf, rcp = 0, result -0 (the limit of SiLU).  v = -inf itself gives
// -inf * 0 = NaN (dp_silu_f returns -0 there): a normalised activation is never infinite unless the tensor already carried an inf / NaN.
__device__ __forceinline__ float dp_silu_fast_f(float v) {
    const float e = __builtin_amdgcn_exp2f(v * -1.44269504088896341f);
    float r = v * __builtin_amdgcn_rcpf(1.0f + e);
    asm volatile("" : "+v"(r));
    return r;
}

// fp32 -> fp16, round to nearest even, of the fp32 value AS ROUNDED TO fp32.  Left to itself the compiler folds a preceding multiply
// into the conversion (v_fma_mixlo_f16: ONE rounding of the exact product) wherever the product has no other fp32 use - which
// kernel variant, output format or resampling mode does so is an accident of code shape, and the double-rounding cases (1e-4 of
// the elements) then differ by one fp16 ulp between variants that must give identical bits (a batch's sharding picks the
// variant).  The empty asm pins the fp32 value in a register first.
__device__ __forceinline__ _Float16 dp_to_half(float v) {
    asm volatile("" : "+v"(v));
    return (_Float16)v;
}

// wave64 reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
