// Epilogue of the fp16 x fp16 convolution kernels: igemm_h2_sw.hip (four waves, one per SIMD, 128 x 128 wave tiles = 4 x 4
// MFMA tiles of 32 x 32, NQ = 2 column records of 64 rows per wave) and igemm_h2_dw.hip (eight waves, 64 x 128 wave
// tiles = 2 x 4 MFMA tiles, NQ = 1).
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
// row0: first output row of the wave tile; colw: its first column; rec0: index of its first 64-row column record
// JP: column tiles handled per pass (16 JP residual loads in flight per lane)
// Addressing: every global access is (wave-uniform 64-bit base in SGPRs) + (32-bit per-lane byte offset): the lane offsets of
// a lane's first row is ONE register per tensor, the row of each accumulator register is a wave-uniform addend of the base; the
// per-element 64-bit multiply-adds of the first version of this epilogue (two VALU instructions and a register pair per
// element) are gone - which is also what lets the 256-register kernel keep its accumulators out of scratch.
// p.rfmt 1 (with OUT16 only: the callers send an fp16 residual under an fp32 output - which no network produces - to the generic
// tiles): the residual is plain fp16, the fp16 residual stream of the fp16 x fp16 modes.  It is read the way the output is stored -
// one dword = two adjacent columns of one row per lane, then the same lane-pair exchange - 8 loads per 32 x 32 tile instead of 16.
// A wave-uniform run-time branch inside the OUT16 instantiation (a third instantiation of the whole epilogue made the 512-register
// kernel spill its accumulators).
// RA: look-ahead of the fp16 residual loads, in column tiles.  1: a tile's 8 loads are issued when the tile is processed (eight dependent
// HBM round trips per 64 x 128 wave tile).  4 (the 8-wave kernel, whose 48 fragment registers are free by now): the loads of all four
// tiles of a 32-row block - 32 dwords per lane - fly together, two round trips per wave tile.
template <bool OUT16, int NQ, int JP, int RA>
__device__ __forceinline__ void sw_epilogue(const ConvH2Args& p, f32x16 (&acc)[2 * NQ][4], int row0, int colw, int rec0, int lr, int lk,
                                            int HW) {
    // Every tile variant must produce the SAME bits, column records included: products and sums stay separate instructions
    // here as in the other epilogues (left to itself the vectoriser pairs `cq += v * v` into v_pk_fma_f32, which skips the
    // rounding of v * v and changes the records in the last bit).
#pragma clang fp contract(off)
    typedef const __attribute__((address_space(1))) char* gptr;
    const float* __restrict__ tembp = p.temb;
    // time-embedding rows: the callers (dp_conv_sw_applies / dp_conv_dw_applies) admit a temb only when H * W % 32 == 0,
    // i.e. when the 32 rows of an MFMA tile belong to ONE sample
    const int col0 = colw + lr;
    const int odd = lr & 1;
    float bv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bv[j] = p.bias ? p.bias[col0 + j * 32] : 0.f;
    // lane offsets (bytes) inside a 32-row tile: the lane's FIRST row (4 lk) and column lr - one register per tensor; the row
    // the r-th accumulator register belongs to, 4 lk + (r & 3) + 8 (r >> 2), adds a wave-uniform (r & 3) + 8 (r >> 2) rows,
    // which goes into the scalar base of the access
    const bool res16 = OUT16 && p.rfmt != 0;
    const unsigned vr0 = res16 ? ((unsigned)(4 * lk + odd) * (unsigned)p.ldr + (unsigned)(lr - odd)) * 2u      // pair load
                               : ((unsigned)(4 * lk) * (unsigned)p.ldr + (unsigned)lr) * 4u;
    const unsigned vo0 = OUT16 ? ((unsigned)(4 * lk + odd) * (unsigned)p.ldo + (unsigned)(lr - odd)) * 2u     // pair store, below
                               : ((unsigned)(4 * lk) * (unsigned)p.ldo + (unsigned)lr) * 4u;
    const size_t ldr_b = (size_t)p.ldr * (res16 ? 2 : 4), ldo_b = (size_t)p.ldo * (OUT16 ? 2 : 4);
    auto rows_of = [](int r) { return (r & 3) + 8 * (r >> 2); };
#pragma unroll
    for (int q = 0; q < NQ; ++q) {                  // one 64-row column record = two 32-row MFMA tiles
        float cs[2][4], cq[2][4];
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
            const int i = 2 * q + ii;
            const int rowt = row0 + i * 32;         // wave-uniform
            float tv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                tv[j] = tembp ? tembp[(size_t)(rowt / HW) * p.temb_stride + col0 + j * 32] : 0.f;
                cs[ii][j] = 0.f;
                cq[ii][j] = 0.f;
            }
            dp_half2 hraw[RA == 4 ? 4 : 1][8];
            if constexpr (RA == 4) {
                if (p.res && res16) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        gptr rb = (gptr)(reinterpret_cast<const _Float16*>(p.res) + (size_t)rowt * p.ldr + colw + j * 32);
#pragma unroll
                        for (int k = 0; k < 8; ++k)
                            hraw[j][k] = *reinterpret_cast<const __attribute__((address_space(1))) dp_half2*>(rb + rows_of(2 * k) * ldr_b + vr0);
                    }
                }
            }
#pragma unroll
            for (int jh = 0; jh < 4 / JP; ++jh) {   // JP column tiles at a time
                float rv[JP][16];
                if (p.res) {
#pragma unroll
                    for (int jj = 0; jj < JP; ++jj) {
                        if (res16) {
                            gptr rb = (gptr)(reinterpret_cast<const _Float16*>(p.res) + (size_t)rowt * p.ldr + colw + (jh * JP + jj) * 32);
#pragma unroll
                            for (int k = 0; k < 8; ++k) {
                                // even lane: row r = 2k, columns (lr, lr + 1); odd lane: row 2k + 1, columns (lr - 1, lr)
                                const dp_half2 h = RA == 4 ? hraw[RA == 4 ? jh * JP + jj : 0][k]
                                                           : *reinterpret_cast<const __attribute__((address_space(1))) dp_half2*>(rb + rows_of(2 * k) * ldr_b + vr0);
                                const float mine = odd ? (float)h[1] : (float)h[0];
                                const float other = sw_swap1(odd ? (float)h[0] : (float)h[1]);
                                rv[jj][2 * k] = odd ? other : mine;
                                rv[jj][2 * k + 1] = odd ? mine : other;
                            }
                        } else {
                            gptr rb = (gptr)(p.res + (size_t)rowt * p.ldr + colw + (jh * JP + jj) * 32);
#pragma unroll
                            for (int r = 0; r < 16; ++r)
                                rv[jj][r] = *reinterpret_cast<const __attribute__((address_space(1))) float*>(rb + rows_of(r) * ldr_b + vr0);
                        }
                    }
                }
                float vv[JP][16];
#pragma unroll
                for (int jj = 0; jj < JP; ++jj) {
                    const int j = jh * JP + jj;
                    char* ob = reinterpret_cast<char*>(p.out + (size_t)rowt * p.ldo + colw + j * 32);   // fp32 output only
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float v = acc[i][j][r] + bv[j] + tv[j];
                        if (p.res) v += rv[jj][r];
                        v *= p.scale;
                        if constexpr (!OUT16) *reinterpret_cast<float*>(ob + rows_of(r) * ldo_b + vo0) = v;
                        vv[jj][r] = v;
                        cs[ii][j] += v;
                        cq[ii][j] += v * v;
                    }
                }
                if constexpr (OUT16) {
#pragma unroll
                    for (int jj = 0; jj < JP; ++jj) {
                        char* oh = reinterpret_cast<char*>(reinterpret_cast<_Float16*>(p.out) + (size_t)rowt * p.ldo + colw + (jh * JP + jj) * 32);
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            const float mine = odd ? vv[jj][2 * k + 1] : vv[jj][2 * k];
                            const float other = sw_swap1(odd ? vv[jj][2 * k] : vv[jj][2 * k + 1]);
                            const dp_half2 h = {dp_to_half(odd ? other : mine), dp_to_half(odd ? mine : other)};
                            // the even lane keeps row r = 2k and stores columns (lr, lr + 1); the odd lane keeps row 2k + 1, columns (lr - 1, lr)
                            *reinterpret_cast<dp_half2*>(oh + rows_of(2 * k) * ldo_b + vo0) = h;
                        }
                    }
                }
                // 256-register kernel: the next pass's residual loads stay behind this pass's stores (hoisted, all 4 x 16 of
                // them are live at once and the accumulators go to scratch)
                if constexpr (JP == 1) asm volatile("" ::: "memory");
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
                float* d = p.colstats + (size_t)(rec0 + q) * 2 * p.N + col0 + j * 32;
                d[0] = cs[0][j] + cs[1][j];
                d[p.N] = cq[0][j] + cq[1][j];
            }
        }
    }
}

template <int NQ, int JP, int RA = 1>
__device__ __forceinline__ void sw_epilogue_any(const ConvH2Args& p, f32x16 (&acc)[2 * NQ][4], int row0, int colw, int rec0, int lr, int lk,
                                                int HW) {
    if (p.ofmt) sw_epilogue<true, NQ, JP, RA>(p, acc, row0, colw, rec0, lr, lk, HW);
    else sw_epilogue<false, NQ, JP, 1>(p, acc, row0, colw, rec0, lr, lk, HW);
}

// Measured on this epilogue and NOT kept (tests/probes/pp_ablate.py, B=64, bit-identical results):
//   * 16-byte stores and residual loads after a 4 x 4 transpose inside each lane quad (two DPP exchange stages): a quarter of the
//     memory instructions, +0.7 % without a residual, -3 % with one, -2.5 % with fp16 output - the tile's fixed cost (~18 us of
//     an 84 us K = 2304 tile) is not the number of store instructions.
}  // namespace
