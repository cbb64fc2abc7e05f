// Epilogue of the one-wave-per-SIMD convolution kernel (igemm_h2_sw.hip): 4 waves per workgroup, each owning a 128 x 128 wave
// tile = 4 x 4 MFMA tiles of 32 x 32 (256 accumulator registers).
#pragma once
#include "igemm_h2.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

#define SW_BARRIER() asm volatile("s_barrier" ::: "memory")

// value of the neighbouring lane (lane ^ 1) through the DPP crossbar (quad_perm [1, 0, 3, 2]): no LDS traffic
__device__ __forceinline__ float sw_swap1(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));
}

// ---- epilogue of a 128 x 128 wave tile: the arithmetic and the column-record order of pp_epilogue (igemm_pp_common.h).
// OUT16 (p.ofmt 1): the tensor is stored as plain fp16 - the value a GroupNorm-apply pass would read next anyway (it rounds
// its own output to fp16), at half the bytes for this kernel's stores and for that pass's loads.  The MFMA accumulator
// layout gives a lane ONE column (lr) of 16 rows; two neighbouring lanes exchange half of their values (rows r odd <-> r
// even) so that each stores two ADJACENT columns of 8 rows as one dword: 8 stores per 32 x 32 tile instead of 16.
template <bool OUT16>
__device__ __forceinline__ void sw_epilogue(const ConvH2Args& p, f32x16 (&acc)[4][4], int m0, int n0, int tile_m, int wr, int wc, int lr,
                                            int lk, int HW) {
    const __attribute__((address_space(1))) float* __restrict__ resp = (const __attribute__((address_space(1))) float*)p.res;
    const float* __restrict__ tembp = p.temb;
    float* __restrict__ outp = p.out;
    _Float16* __restrict__ outh = reinterpret_cast<_Float16*>(p.out);
    const bool hw32 = HW % 32 == 0;
    const int col0 = n0 + wc * 128 + lr;
    const bool odd = lr & 1;
    float bv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bv[j] = p.bias ? p.bias[col0 + j * 32] : 0.f;
#pragma unroll
    for (int q = 0; q < 2; ++q) {                   // one 64-row column record = two 32-row MFMA tiles
        float cs[2][4], cq[2][4];
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
            const int i = 2 * q + ii;
            const int rowb = m0 + wr * 128 + i * 32 + 4 * lk;
            float tv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                tv[j] = (tembp && hw32) ? tembp[(size_t)(rowb / HW) * p.temb_stride + col0 + j * 32] : 0.f;
                cs[ii][j] = 0.f;
                cq[ii][j] = 0.f;
            }
#pragma unroll
            for (int jh = 0; jh < 2; ++jh) {        // two column tiles at a time: 32 residual loads in flight per lane
                float rv[2][16];
                if (resp) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const __attribute__((address_space(1))) float* rp = resp + (size_t)(rowb + (r & 3) + 8 * (r >> 2)) * p.ldr + col0 + jh * 64;
                        rv[0][r] = rp[0];
                        rv[1][r] = rp[32];
                    }
                }
                float vv[2][16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = rowb + (r & 3) + 8 * (r >> 2);
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        const int j = jh * 2 + jj;
                        float v = acc[i][j][r] + bv[j];
                        if (tembp) v += hw32 ? tv[j] : tembp[(size_t)(row / HW) * p.temb_stride + col0 + j * 32];
                        if (resp) v += rv[jj][r];
                        v *= p.scale;
                        if constexpr (!OUT16) outp[(size_t)row * p.ldo + col0 + jh * 64 + jj * 32] = v;
                        vv[jj][r] = v;
                        cs[ii][j] += v;
                        cq[ii][j] += v * v;
                    }
                }
                if constexpr (OUT16) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) {   // rows r = 2k (kept by the even lane) and 2k + 1 = the next row (odd lane)
                        const int row = rowb + ((2 * k) & 3) + 8 * ((2 * k) >> 2) + (odd ? 1 : 0);
#pragma unroll
                        for (int jj = 0; jj < 2; ++jj) {
                            const float mine = odd ? vv[jj][2 * k + 1] : vv[jj][2 * k];
                            const float other = sw_swap1(odd ? vv[jj][2 * k] : vv[jj][2 * k + 1]);
                            // even lane: columns (lr, lr + 1) of row r = 2k; odd lane: columns (lr - 1, lr) of row r = 2k + 1
                            const dp_half2 h = {(_Float16)(odd ? other : mine), (_Float16)(odd ? mine : other)};
                            *reinterpret_cast<dp_half2*>(outh + (size_t)row * p.ldo + (col0 - (odd ? 1 : 0)) + jh * 64 + jj * 32) = h;
                        }
                    }
                }
            }
            if (p.colstats) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    cs[ii][j] += __shfl_xor(cs[ii][j], 32, 64);
                    cq[ii][j] += __shfl_xor(cq[ii][j], 32, 64);
                }
            }
        }
        if (p.colstats && lk == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float* d = p.colstats + (size_t)(tile_m * 4 + wr * 2 + q) * 2 * p.N + col0 + j * 32;
                d[0] = cs[0][j] + cs[1][j];
                d[p.N] = cq[0][j] + cq[1][j];
            }
        }
    }
}

__device__ __forceinline__ void sw_epilogue_any(const ConvH2Args& p, f32x16 (&acc)[4][4], int m0, int n0, int tile_m, int wr, int wc,
                                                int lr, int lk, int HW) {
    if (p.ofmt) sw_epilogue<true>(p, acc, m0, n0, tile_m, wr, wc, lr, lk, HW);
    else sw_epilogue<false>(p, acc, m0, n0, tile_m, wr, wc, lr, lk, HW);
}

// Measured on this epilogue and NOT kept (tests/probes/pp_ablate.py, B=64, bit-identical results):
//   * 16-byte stores and residual loads after a 4 x 4 transpose inside each lane quad (two DPP exchange stages): a quarter of the
//     memory instructions, +0.7 % without a residual, -3 % with one, -2.5 % with fp16 output - the tile's fixed cost (~18 us of
//     an 84 us K = 2304 tile) is not the number of store instructions.
}  // namespace
