// Pieces shared by the 8-wave ping-pong convolution kernel (igemm_h2_pp.hip; round 2 also had a halo-tile form, removed in round 4): LDS-DMA issue with scalar
// base + lane offset, the barrier spelling, and the fused epilogue.
#pragma once
#include "igemm_h2.h"

// Every tile variant must produce the SAME bits, column records included (a batch's sharding picks the variant): products and sums stay
// separate IEEE operations in this file - left to itself the compiler contracts `cq += v * v` (and `v *= scale; cs += v`) into FMAs in one
// code shape and not in another.  Explicit fmaf() calls are unaffected.
#pragma clang fp contract(off)

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

// LDS-DMA with (scalar base + 32-bit lane offset) addressing, spelled in asm: the builtin lets the
// compiler strength-reduce the k-loop addresses back into 64-bit VGPR pointers (two VALU adds and two
// address VGPRs per DMA).  M0 = LDS destination of lane 0 (wave-uniform); lane l lands at M0 + 16 l.
// make a wave-uniform 64-bit value provably scalar for the compiler
__device__ __forceinline__ long long pp_uniform(long long v) {
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)v >> 32));
    return (long long)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ void pp_glds(unsigned voff, const char* sbase, const char* lds_dst) {
    const unsigned m0v = (unsigned)(size_t)(const __attribute__((address_space(3))) char*)lds_dst;
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 :
                 : "v"(voff), "s"(sbase), "s"(m0v)
                 : "memory");
}
#define PP_BARRIER() asm volatile("s_barrier" ::: "memory")


// ---- epilogue of a 128 x 64 wave tile (acc[4][2] MFMA tiles of 32 x 32) at rows m0 + wr*128, columns n0 + wc*64:
// bias / temb / residual / scale, store, per-column (sum, sumsq) of every 64 output rows.
// 32-row sub-sums (the lane's 16 values as two chains - even r, odd r - added, then the partner half-wave) paired even+odd: the
// tile-shape-independent order of igemm.hip.  Both sub-sums of a record live in this wave: no LDS.
// OUT16 (p.ofmt 1): the tensor is stored as plain fp16 (one 2-byte store per value; the packed form lives in igemm_sw_common.h).  A
// template parameter, chosen once per kernel by pp_epilogue: tested per store it costs a branch per element.
template <int BM, bool OUT16>
__device__ __forceinline__ void pp_epilogue_t(const ConvH2Args& p, f32x16 (&acc)[4][2], int m0, int n0, int tile_m, int wr, int wc, int lr,
                                              int lk, int HW) {
    const float* __restrict__ resp = p.res;
    const float* __restrict__ tembp = p.temb;
    float* __restrict__ outp = p.out;
    _Float16* __restrict__ outh = reinterpret_cast<_Float16*>(p.out);      // p.ofmt 1: fp16 output
    const bool hw32 = HW % 32 == 0;          // a 32-row block lies inside one sample: one temb value per block
    // One 32-row block at a time, both 32-column blocks inside it: a row's address is formed once and serves both
    // column blocks (+128 bytes), 32 residual loads are in flight per lane.  (Column block outermost makes the compiler
    // keep all 64 row addresses = 128 VGPRs live from one column block to the next next to the 128 accumulators: ~110
    // spilled values and -5 % on the whole kernel.)
    const int col0 = n0 + wc * 64 + lr;
    float bv[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) bv[j] = p.bias ? p.bias[col0 + j * 32] : 0.f;
    float cs[4][2], cq[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float cs2[2][2], cq2[2][2];          // [r & 1][j]: two chains per column (even r, odd r), the order of igemm_sw_common.h
        const int rowb = m0 + wr * 128 + i * 32 + 4 * lk;
        float rv[2][16];
        float tv[2];
        if (resp) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const size_t rrow = (size_t)(rowb + (r & 3) + 8 * (r >> 2));
                rv[0][r] = dp_conv_res(p, rrow, col0);
                rv[1][r] = dp_conv_res(p, rrow, col0 + 32);
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            tv[j] = (tembp && hw32) ? tembp[(size_t)(rowb / HW) * p.temb_stride + col0 + j * 32] : 0.f;
            cs2[0][j] = cs2[1][j] = 0.f;
            cq2[0][j] = cq2[1][j] = 0.f;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = rowb + (r & 3) + 8 * (r >> 2);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float v = acc[i][j][r] + bv[j];
                if (tembp) v += hw32 ? tv[j] : tembp[(size_t)(row / HW) * p.temb_stride + col0 + j * 32];
                if (resp) v += rv[j][r];
                v *= p.scale;
                if constexpr (OUT16) outh[(size_t)row * p.ldo + col0 + j * 32] = dp_to_half(v);
                else outp[(size_t)row * p.ldo + col0 + j * 32] = v;
                cs2[r & 1][j] += v;
                cq2[r & 1][j] += v * v;
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            cs[i][j] = cs2[0][j] + cs2[1][j];
            cq[i][j] = cq2[0][j] + cq2[1][j];
        }
        if (p.colstats) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                cs[i][j] += __shfl_xor(cs[i][j], 32, 64);
                cq[i][j] += __shfl_xor(cq[i][j], 32, 64);
            }
        }
    }
    if (p.colstats && lk == 0) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                float* d = p.colstats + (size_t)(tile_m * (BM / 64) + wr * 2 + q) * 2 * p.N + col0 + j * 32;
                d[0] = cs[2 * q][j] + cs[2 * q + 1][j];
                d[p.N] = cq[2 * q][j] + cq[2 * q + 1][j];
            }
    }
}

template <int BM>
__device__ __forceinline__ void pp_epilogue(const ConvH2Args& p, f32x16 (&acc)[4][2], int m0, int n0, int tile_m, int wr, int wc, int lr,
                                            int lk, int HW) {
    if (p.ofmt) pp_epilogue_t<BM, true>(p, acc, m0, n0, tile_m, wr, wc, lr, lk, HW);
    else pp_epilogue_t<BM, false>(p, acc, m0, n0, tile_m, wr, wc, lr, lk, HW);
}

}  // namespace
