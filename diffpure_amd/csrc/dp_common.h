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
void dp_prof_begin(int kind, double flop, hipStream_t s, void** rec);
void dp_prof_end(void* rec, hipStream_t s);

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

__device__ __forceinline__ float dp_silu_f(float v) { return v / (1.0f + expf(-v)); }

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
