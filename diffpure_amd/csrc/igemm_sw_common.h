// Epilogue of the fp16 x fp16 convolution kernels: igemm_h2_sw.hip (four waves, one per SIMD, 128 x 128 wave tiles = 4 x 4
// MFMA tiles of 32 x 32, NQ = 2 column records of 64 rows per wave) and igemm_h2_dw.hip (eight waves, 64 x 128 wave
// tiles = 2 x 4 MFMA tiles, NQ = 1).
#pragma once
#include "igemm_h2.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

#define SW_BARRIER() asm volatile("s_barrier" ::: "memory")

// value of the neighbouring lane (lane ^ 1) through the DPP crossbar (quad_perm [1, 0, 3, 2]): no LDS traffic - sw_swap1u below
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned sw_swap1u(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true); }

// v + (float)h for the low / high half of a packed fp16 pair: ONE instruction (v_fma_mix_f32: h * 1.0 + v, rounded once - the fp32 sum of
// two fp32 values, i.e. the bits of the convert-then-add the other tile kernels do; fp16 subnormals are honoured by the FMA_MIX form)
__device__ __forceinline__ float sw_add_half_lo(float v, unsigned h2) {
    float o;
    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel_hi:[1,0,0]" : "=v"(o) : "v"(h2), "v"(v));
    return o;
}
__device__ __forceinline__ float sw_add_half_hi(float v, unsigned h2) {
    float o;
    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(o) : "v"(h2), "v"(v));
    return o;
}
// load / store at (wave-uniform 64-bit base) + (wave-uniform 32-bit offset) + (32-bit lane offset): raw buffer accesses - the base sits in
// a descriptor in SGPRs, the uniform row advance in the instruction's scalar offset, the lane offset is ONE register per tensor.  (Spelled
// with pointers, the compiler keeps a 64-bit per-lane pointer pair and advances it with a vector add per access.)
typedef __amdgpu_buffer_rsrc_t sw_rsrc;
__device__ __forceinline__ sw_rsrc sw_make_rsrc(const void* base, bool present = true) {
    // raw, stride 0, dword3 of gfx90a / gfx94x / gfx950; an absent tensor gets 0 records: every load from it returns 0 - no branch
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, present ? 0x7fffffff : 0, 0x00020000);
}
// two fp32 -> one dword of two fp16 (round to nearest even each, of the fp32 values as they stand: the operands are pinned so that no
// preceding multiply is folded into the conversion - dp_to_half in dp_common.h)
__device__ __forceinline__ unsigned sw_pack_half2(float lo, float hi) {
    unsigned o;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(o) : "v"(lo), "v"(hi));
    return o;
}

// ---- epilogue of a wave tile of 2 NQ x 4 MFMA tiles: the arithmetic of pp_epilogue (igemm_pp_common.h) and of the generic tiles -
// v = ((acc + bias) + temb + residual) * scale in this order, every operation one IEEE fp32 operation - and their column-record order:
// per 32-row MFMA tile and column, the lane's 16 values are summed as TWO chains (even r, odd r: rows 4 lk + (r & 3) + 8 (r >> 2)),
// the chains are added, then the partner half-wave (lk), then the two MFMA tiles of a 64-row record.
// The whole epilogue is VALU-bound (round 4: its cost did not move with the memory system - 16 tiles on an empty chip, look-ahead of the
// residual loads, de-phased workgroups - and the first version spent ~20 vector instructions per value), so it is written over ROW PAIRS
// (acc[2k], acc[2k + 1]: adjacent rows of one column, an aligned register pair) in packed fp32 arithmetic: v_pk_add_f32 / v_pk_mul_f32
// are two independent IEEE operations per lane - same bits as the scalar form of the other kernels at half the instructions (v_pk_fma
// is NOT used: it would skip the rounding of v * v).  Per pair: bias 1, temb 1, scale 1, records 3 (add, square, add), and
//   OUT16 (p.ofmt 1, the tensor is stored as plain fp16): 1 conversion of the pair (v_cvt_pk_f16_f32), 1 DPP exchange with the lane of
//     the neighbouring column, 1 byte permute - the even lane keeps row 2k and stores columns (lr, lr + 1) as one dword, the odd lane
//     row 2k + 1, columns (lr - 1, lr): 8 dword stores per 32 x 32 tile;
//   RK 2 (p.rfmt 1: fp16 residual, the fp16 residual stream): the wave tile's residual - 64 NQ rows x 256 bytes - is brought into LDS by
//     16 NQ LDS-DMA instructions issued TOGETHER at the top (the operand rings are dead after the k-loop's last barrier; 1 KB = four rows
//     per instruction, blocks SW_EPI_PITCH = 1088 bytes apart so that the two half-waves read different banks): the whole 128 KB of a
//     workgroup is in flight at once - one memory round trip.  (Read through registers, 8-16 dwords per lane in flight, a CU moved its
//     128 KB at ~20 GB/s: 12 000 of the 21 000 cycles of this epilogue at K = 2304.)  A pair is then two ds_read_u16_d16(_hi) at constant
//     offsets - the lane's own column, rows 2k and 2k + 1, as one packed register - and 2 v_fma_mix_f32;
//   RK 1: fp32 residual, 2 loads and 1 packed add per pair.
// row0: first output row of the wave tile; colw: its first column; rec0: index of its first 64-row column record.
// Addressing: every global access is (wave-uniform 64-bit base) + (32-bit per-lane byte offset of the lane's first row and column).
constexpr int SW_EPI_PITCH = 1088;                 // LDS bytes per 4-row block of the residual landing zone (1024 + 16 banks)
constexpr int SW_EPI_LDS = 128 * SW_EPI_PITCH;      // per workgroup: 512 rows x 128 columns (dw: 8 waves x 64 rows, sw: 4 x 128) = 136 KB

// lds_wave: this wave's residual landing zone (RK 2), 16 NQ blocks of SW_EPI_PITCH bytes inside the (dead) operand rings
template <bool OUT16, int NQ, int RK>
__device__ __forceinline__ void sw_epilogue(const ConvH2Args& p, f32x16 (&acc)[2 * NQ][4], int row0, int colw, int rec0, int lr, int lk,
                                            int HW, char* lds_wave) {
#pragma clang fp contract(off)
    static_assert(RK != 2 || OUT16, "an fp16 residual under an fp32 output goes to the generic tiles");
    const float* __restrict__ tembp = p.temb;
    // time-embedding rows: the callers (dp_conv_sw_applies / dp_conv_dw_applies) admit a temb only when H * W % 32 == 0,
    // i.e. when the 32 rows of an MFMA tile belong to ONE sample
    const int col0 = colw + lr;
    const int odd = lr & 1;
    const unsigned sel = odd ? 0x03020706u : 0x05040100u;   // v_perm_b32 {partner, own}: even lane (own.lo, partner.lo), odd (partner.hi, own.hi)
    // bias and every time-embedding value of the wave tile are loaded HERE, together and without a branch (an absent tensor reads as 0):
    // loaded where they are used - one conditional load per MFMA tile - each costs the tile a full memory round trip (and the branch makes
    // the compiler wait for vmcnt(0), i.e. for the residual look-ahead too): 8 round trips per wave tile, most of the first version's time
    float bv[4], tvv[2 * NQ][4];
    {
        const sw_rsrc bb = sw_make_rsrc(p.bias, p.bias != nullptr), tb = sw_make_rsrc(tembp, tembp != nullptr);
#pragma unroll
        for (int j = 0; j < 4; ++j) bv[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(bb, (unsigned)(col0 + j * 32) * 4u, 0, 0));
#pragma unroll
        for (int i = 0; i < 2 * NQ; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                tvv[i][j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(tb, (unsigned)(col0 + j * 32) * 4u, ((row0 + i * 32) / HW) * p.temb_stride * 4, 0));
    }
    const f32x2 sc = {p.scale, p.scale};
    // lane offsets (bytes) inside a 32-row tile; the row of pair k, {0, 2}[k & 1] + 8 (k >> 1) (+ 1 for its second value), is a
    // wave-uniform addend of the base
    const unsigned vr0 = RK == 2 ? ((unsigned)(4 * lk + odd) * (unsigned)p.ldr + (unsigned)(lr - odd)) * 2u      // pair load
                                 : ((unsigned)(4 * lk) * (unsigned)p.ldr + (unsigned)lr) * 4u;
    const unsigned vo0 = OUT16 ? ((unsigned)(4 * lk + odd) * (unsigned)p.ldo + (unsigned)(lr - odd)) * 2u       // pair store
                               : ((unsigned)(4 * lk) * (unsigned)p.ldo + (unsigned)lr) * 4u;
    const int ldr_b = p.ldr * (RK == 2 ? 2 : 4), ldo_b = p.ldo * (OUT16 ? 2 : 4);       // row pitches in bytes
    auto row_of_pair = [](int k) { return 2 * (k & 1) + 8 * (k >> 1); };
    auto res_base = [&](int i, int j) -> sw_rsrc {
        const size_t e = (size_t)(row0 + i * 32) * p.ldr + colw + j * 32;
        return sw_make_rsrc(RK == 2 ? (const void*)(reinterpret_cast<const _Float16*>(p.res) + e) : (const void*)(p.res + e));
    };
    if constexpr (RK == 2) {
        __builtin_amdgcn_sched_barrier(0);          // nothing that waits on the loads above is scheduled in between the DMA issues
        const int lane = lk * 32 + lr;
        const char* src = reinterpret_cast<const char*>(reinterpret_cast<const _Float16*>(p.res) + (size_t)(row0 + (lane >> 4)) * p.ldr + colw + (lane & 15) * 8);
#pragma unroll
        for (int b = 0; b < 16 * NQ; ++b)           // block b = rows 4b .. 4b + 3 of the wave tile; lane l lands at block + 16 l
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)(4 * b) * ldr_b),
                                             (__attribute__((address_space(3))) void*)(lds_wave + b * SW_EPI_PITCH), 16, 0, 0);
        // the wave reads only what it brought in itself: its own vmcnt, no barrier (the compiler does not order LDS reads behind LDS-DMA).
        // First the blocks of the first 32-row MFMA tile row (loads return in issue order: 8 (NQ - 1) + 8 younger DMAs may still fly);
        // the rest is waited for before the second tile row, under whose arithmetic it lands (a CU's L2 -> LDS path moves the 128 KB
        // in ~4 000 cycles)
        asm volatile("s_waitcnt vmcnt(%0)" : : "n"(16 * NQ - 8) : "memory");
        __builtin_amdgcn_sched_barrier(0);
    }
    // the lane's own column in the landing zone: row 32 i + 4 lk + 2 (k & 1) + 8 (k >> 1) + e, column 32 j + lr
    const _Float16* lres = reinterpret_cast<const _Float16*>(lds_wave + lk * SW_EPI_PITCH + lr * 2);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {                  // one 64-row column record = two 32-row MFMA tiles
        float cs[2][4], cq[2][4];
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
            const int i = 2 * q + ii;
            const int rowt = row0 + i * 32;         // wave-uniform
            if constexpr (RK == 2) {
                if (i == 1) {                       // everything has landed (this also waits for the first tile row's stores: loads and stores share the counter)
                    __builtin_amdgcn_sched_barrier(0);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x2 tv = {tvv[i][j], tvv[i][j]};
                float rf[RK == 1 ? 16 : 1];
                if constexpr (RK == 1) {
                    const sw_rsrc rb = res_base(i, j);
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        rf[2 * k] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rb, vr0, row_of_pair(k) * ldr_b, 0));
                        rf[2 * k + 1] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rb, vr0, (row_of_pair(k) + 1) * ldr_b, 0));
                    }
                }
                const sw_rsrc ob = sw_make_rsrc(OUT16 ? (void*)(reinterpret_cast<_Float16*>(p.out) + (size_t)rowt * p.ldo + colw + j * 32)
                                                      : (void*)(p.out + (size_t)rowt * p.ldo + colw + j * 32));
                f32x2 s2 = {0.f, 0.f}, q2 = {0.f, 0.f};
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    f32x2 v = {acc[i][j][2 * k], acc[i][j][2 * k + 1]};
                    v = v + f32x2{bv[j], bv[j]};
                    v = v + tv;
                    if constexpr (RK == 1) v = v + f32x2{rf[2 * k], rf[2 * k + 1]};
                    if constexpr (RK == 2) {
                        const int o = ((8 * i + 2 * (k >> 1)) * SW_EPI_PITCH + 2 * (k & 1) * 256 + j * 64) / 2;
                        const dp_half2 h = {lres[o], lres[o + 128]};                                // (res[2k][lr], res[2k + 1][lr])
                        const unsigned mine = __builtin_bit_cast(unsigned, h);
                        v = f32x2{sw_add_half_lo(v[0], mine), sw_add_half_hi(v[1], mine)};
                    }
                    v = v * sc;
                    s2 = s2 + v;
                    const f32x2 vsq = v * v;
                    q2 = q2 + vsq;
                    if constexpr (OUT16) {
                        const unsigned own = sw_pack_half2(v[0], v[1]);
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_amdgcn_perm(sw_swap1u(own), own, sel), ob, vo0, row_of_pair(k) * ldo_b, 0);
                    } else {
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[0]), ob, vo0, row_of_pair(k) * ldo_b, 0);
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[1]), ob, vo0, (row_of_pair(k) + 1) * ldo_b, 0);
                    }
                }
                cs[ii][j] = s2[0] + s2[1];
                cq[ii][j] = q2[0] + q2[1];
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

template <int NQ>
__device__ __forceinline__ void sw_epilogue_any(const ConvH2Args& p, f32x16 (&acc)[2 * NQ][4], int row0, int colw, int rec0, int lr, int lk,
                                                int HW, char* lds_wave) {
    if (p.ofmt) {
        if (!p.res) sw_epilogue<true, NQ, 0>(p, acc, row0, colw, rec0, lr, lk, HW, lds_wave);
        else if (p.rfmt) sw_epilogue<true, NQ, 2>(p, acc, row0, colw, rec0, lr, lk, HW, lds_wave);
        else sw_epilogue<true, NQ, 1>(p, acc, row0, colw, rec0, lr, lk, HW, lds_wave);
    } else {
        if (!p.res) sw_epilogue<false, NQ, 0>(p, acc, row0, colw, rec0, lr, lk, HW, lds_wave);
        else sw_epilogue<false, NQ, 1>(p, acc, row0, colw, rec0, lr, lk, HW, lds_wave);
    }
}

// Measured on this epilogue and NOT kept (tests/probes/pp_ablate.py, B=64, bit-identical results):
//   * 16-byte stores and residual loads after a 4 x 4 transpose inside each lane quad (two DPP exchange stages): a quarter of the
//     memory instructions, +0.7 % without a residual, -3 % with one, -2.5 % with fp16 output - the tile's fixed cost (~18 us of
//     an 84 us K = 2304 tile) is not the number of store instructions.
}  // namespace
