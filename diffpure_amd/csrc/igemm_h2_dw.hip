// fp16 x fp16 implicit-GEMM convolution, 256 (pixels) x 256 (channels) tile, ONE 8-WAVE WORKGROUP PER CU with two FREE-RUNNING
// waves per SIMD ("dw8"; the dominant kernel of the purification step since round 3).
//
// What bounds the one-wave-per-SIMD kernel (igemm_h2_sw.hip): a lone in-order wave exposes every cycle an
// LDS-DMA issue or a ds_read_b128 costs beyond the 32-cycle shadow of one MFMA.  Here two waves share every SIMD on the SAME
// 256 x 256 tile (wave tile 64 x 128 = 2 x 4 MFMA tiles of 32x32, 128 accumulator registers of the 256 a wave may use): the
// operand traffic of the one-wave-per-SIMD kernel (32 KB per k-tile), and whenever one wave of a SIMD sits in an LDS-DMA issue,
// a ds_read, its vmcnt or the barrier, the other's MFMAs take the pipe.  Unlike the ping-pong kernel (same geometry) the waves
// are not assigned roles: every wave runs the same interleaved MFMA / read / DMA stream and meets the others at ONE barrier per
// k-tile.  (Round 3 also built the two-workgroups-per-CU form on 128 x 256 tiles, with a per-CU ticket to stagger the two
// epilogues: bit-identical, 811-870 vs 1 022-1 130 TFLOP/s because of 1.5x the operand traffic - removed in round 4, see git history
// and DESIGN.md section 6.)
//
// Per k-tile (32 channels of one tap) a wave issues 16 MFMAs, 12 ds_read_b128 and 4 LDS-DMA pieces.  The two operands have
// separate LDS rings of three stages each (prefetch distance 2).
//
//   iteration t:  issue DMA: weights of k-tile t+2, then activations of k-tile t+2          | 8 MFMA (t, s=0), ds_read (t, s=1)
//                 4 MFMA (t, s=1, row 0) ; s_waitcnt vmcnt(in-order count: k-tile t+1 landed) ; s_barrier
//                 ds_read fragments (t+1, s=0)                                          | 4 MFMA (t, s=1, row 1)
//   RAW: every wave waits for its own share of k-tile t+1 before the barrier of iteration t; the reads follow it.
//   WAR: the stages written in iteration t held k-tile t-1, whose last reads precede the barrier of iteration t-1.
// Same operand formats, reduction order and epilogue arithmetic as every other variant: bit-identical output.
// Needs: fp16 activations and weights (a_fmt 1, w_fmt 1, passes 1), M % 256 == 0, N % 256 == 0, C % 32 == 0, >= 4 k-tiles.
//
// 1x1 K-SEGMENTS (round 4; the rolled kernel only): after the KS*KS*C/32 k-tiles over the zero-bordered operand the loop runs on
// over the channels of up to two plain fp16 NHWC tensors (p.seg1, p.seg2; no border) - the raw input(s) of a ResBlock whose
// 1x1 skip_connection is thereby folded into its second 3x3 convolution (igemm_h2.h).  Only the address of an activation
// piece changes (a different base pointer and pixel stride per segment); the weight stream, the rings and the waits do not.
#include <stdlib.h>

#include "dp_tune.h"
#include "igemm_h2.h"
#include "igemm_sw_common.h"

namespace {

constexpr int NXCD = 8;
constexpr int BTILE = 256 * 64;                 // weight tile of one k-tile: 256 rows x 64 bytes (32 fp16)
constexpr int BDEPTH = 3;

template <int N>
__device__ __forceinline__ void dw_wait_vm() {
    static_assert(N == 4 || N == 0, "add the immediate");
    if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// (Round 4 measured a start-up de-phasing of the CUs - the first workgroup of CU slot c starting c / 32 of a tile late, so that the
//  256 epilogues do not hit HBM together: 0 % on 256^2 256->256 at B=64, -2...-4 % on the shorter launches, -1...-2 % on the
//  purification (profiles/r04/dephase_ab.log).  The epilogue's cost is not a lockstep burst.  Removed.)
// MODE (timing ablations, DP_ABLATE builds only; WRONG RESULTS): 1 = no DMA in the steady state, 2 = no barrier / vmcnt wait,
// 4 = no ds_reads, 8 = no epilogue stores, 16 = no activation DMA, 32 = no weight DMA
template <int MODE>
__global__ __launch_bounds__(512, 1) void conv_igemm_dw(ConvH2Args p) {
    constexpr int ADEPTH = 3, DA = ADEPTH - 1;  // ring stages / prefetch distance of both operands
    constexpr int BMT = 256;                    // tile rows: every wave stages 32 of them (2 pieces) and 32 weight rows (2 pieces)
    constexpr int ATILE = BMT * 64;             // activation tile of one k-tile
    constexpr int NPB = 2;                      // weight pieces per wave and k-tile
    constexpr int BBASE = ADEPTH * ATILE;
    __shared__ __attribute__((aligned(1024))) char smem[ADEPTH * ATILE + BDEPTH * BTILE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    int tile;
    {   // XCD-aware bijective remap (speed only)
        const int b = blockIdx.x, x = b % NXCD, q = p.tiles / NXCD, r = p.tiles % NXCD;
        tile = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + b / NXCD;
    }
    const int tile_n = tile % p.tiles_n, tile_m = tile / p.tiles_n;
    const int m0 = tile_m * BMT, n0 = tile_n * 256;
    const int HW = p.H * p.W, Wp = p.W + 2, taps = p.KS * p.KS;
    const int nt = p.K / 32;

    // ---- staging: wave w fills rows [32 w, 32 w + 32) of the A tile and of the B tile, 16 rows per DMA instruction; lane -> row
    // (lane >> 2) of the piece, physical slot lane & 3, logical slot XOR-ed with the row key
    const int lrow = lane >> 2;
    const int ls = (lane & 3) ^ ((lrow >> 2) & 3);
    const char* actr[2];                        // centre pixel of the lane's A row (segments: the lane's pixel), + slot
    const char* bptr[NPB];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int m = m0 + wave * 32 + it * 16 + lrow;
        const int b = m / HW, rem = m - b * HW;
        const int oy = rem / p.W, ox = rem - oy * p.W;
        actr[it] = p.x + ((size_t)(b * (p.H + 2) + oy + 1) * Wp + ox + 1) * p.C * 2 + ls * 16;
    }
#pragma unroll
    for (int it = 0; it < NPB; ++it) {
        const int n = n0 + wave * 32 + it * 16 + lrow;              // block layout of the fp16 panels (ops.order_conv_weight_w16)
        bptr[it] = p.w + (size_t)(n >> 5) * p.K * 64 + (n & 31) * 16 + ls * 512;
    }
    // (tap, slice) of the next activation k-tile to stage, inside the current K-segment: segment 0 = the KS x KS convolution over
    // p.x (C / 32 slices of `taps` k-tiles), then the 1x1 segments over p.seg1 / p.seg2 (segC / 32 slices of one k-tile)
    int cur_tap = 0, cur_c = 0, cur_seg = 0, seg_slices = p.C / 32;
    long long a_off = 0;
    auto pieceA = [&](int aoff, int it) {       // aoff: byte offset of the ring stage
        if (it == 0) {
            if (cur_c == seg_slices) {          // (wave-uniform) this segment is staged: on to the next tensor
                ++cur_seg;
                const char* sb = cur_seg == 1 ? p.seg1 : p.seg2;
                const int sc = cur_seg == 1 ? p.segC1 : p.segC2;
#pragma unroll
                for (int j = 0; j < 2; ++j) actr[j] = sb + (size_t)(m0 + wave * 32 + j * 16 + lrow) * sc * 2 + ls * 16;
                seg_slices = sc / 32;
                cur_c = 0;
            }
            if (cur_seg == 0) {
                const int ky = p.KS == 3 ? (cur_tap * 11) >> 5 : 0, kx = cur_tap - ky * p.KS;     // tap / 3 for tap < 9, no division
                a_off = ((long long)(ky - p.pad) * Wp + (kx - p.pad)) * p.C * 2 + (long long)cur_c * 64;
            } else {
                a_off = (long long)cur_c * 64;
            }
        }
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(actr[it] + a_off),
                                         (__attribute__((address_space(3))) void*)(smem + aoff + (wave * 32 + it * 16) * 64), 16, 0, 0);
        if (it == 1 && (cur_seg != 0 || ++cur_tap == taps)) { cur_tap = 0; ++cur_c; }
    };
    auto pieceB = [&](int boff, int it) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)bptr[it],
                                         (__attribute__((address_space(3))) void*)(smem + boff + (wave * 32 + it * 16) * 64), 16, 0, 0);
        bptr[it] += 2048;
    };
    auto issueA = [&](int aoff) { pieceA(aoff, 0); pieceA(aoff, 1); };
    auto issueB = [&](int boff) {
#pragma unroll
        for (int it = 0; it < NPB; ++it) pieceB(boff, it);
    };

    // ---- fragments: lane -> row lr of a 32-row MFMA tile, k-half lk; 64-byte rows, slot (s*2 + lk) ^ key, key = (row >> 2) & 3
    const int lr = lane & 31, lk = lane >> 5;
    const int arow = (wr * 64 + lr) * 64;                   // + i * 32 * 64
    const int brow = (wc * 128 + lr) * 64;                  // + j * 32 * 64
    int soff[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) soff[s] = ((s * 2 + lk) ^ ((lr >> 2) & 3)) << 4;
    half8 fa[2][2], fb[2][4];                               // [register set = k16 step][tile]
    auto readA = [&](int set, int aoff, int i) { fa[set][i] = *reinterpret_cast<const half8*>(smem + aoff + arow + i * 32 * 64 + soff[set]); };
    auto readB = [&](int set, int boff, int j) { fb[set][j] = *reinterpret_cast<const half8*>(smem + boff + brow + j * 32 * 64 + soff[set]); };
    auto read_frags = [&](int set, int aoff, int boff) {
        readA(set, aoff, 0);
        readA(set, aoff, 1);
#pragma unroll
        for (int j = 0; j < 4; ++j) readB(set, boff, j);
    };

    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    auto mfma_rows = [&](int set, int i0, int i1) {           // MFMA tile rows [i0, i1) of k16 step `set`
#pragma unroll
        for (int i = i0; i < i1; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[set][i], fb[set][j], acc[i][j], 0, 0, 0);
    };

    // ring stage offsets, rotated once per k-tile: ar[0] / br[0] hold k-tile t, ar[1] / br[1] k-tile t+1,
    // ar[2] / br[2] are the stages the iteration writes
    int ar[ADEPTH], br[BDEPTH];
#pragma unroll
    for (int i = 0; i < ADEPTH; ++i) ar[i] = i * ATILE;
#pragma unroll
    for (int i = 0; i < BDEPTH; ++i) br[i] = BBASE + i * BTILE;
    auto rotate = [&]() {
        const int a0 = ar[0], b0 = br[0];
        ar[0] = ar[1];
        ar[1] = ar[2];
        ar[2] = a0;
        br[0] = br[1];
        br[1] = br[2];
        br[2] = b0;
    };

    // ---- prologue (nt >= 4): B(0), A(0), A(1), B(1) in flight - in THAT order, because vmcnt counts in issue order
    // and the steady-state wait "everything up to the weights of k-tile t+1" must leave only younger pieces outstanding
    issueB(br[0]);
    issueA(ar[0]);
    issueA(ar[1]);
    issueB(br[1]);
    dw_wait_vm<NPB + 2>();                      // k-tile 0 landed; A(1), B(1) may fly
    SW_BARRIER();
    read_frags(0, ar[0], br[0]);

    // steady state: k-tile t+2 exists
    int t = 0;
    for (; t + DA < nt; ++t) {
        // first half: 8 MFMAs on fragment set 0 | the 6 reads of set 1 and the NPB + 2 DMA pieces, one (read, piece) pair per MFMA shadow
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            if constexpr (!(MODE & 4)) {
                if (k < 2) readA(1, ar[0], k);
                else readB(1, br[0], k - 2);
            }
            if (k < NPB) {
                if constexpr (!(MODE & 33)) pieceB(br[2], k);
            } else if (k < NPB + 2) {
                if constexpr (!(MODE & 17)) pieceA(ar[DA], k - NPB);
            }
        }
        mfma_rows(0, 0, 2);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            if (k < NPB + 2) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        mfma_rows(1, 0, 1);
        __builtin_amdgcn_sched_barrier(0);
        // outstanding in issue order: [.., B(t+1), A(t+1)] from iteration t-1, [B(t+2), A(t+2)] from this one
        if constexpr (!(MODE & 3)) dw_wait_vm<NPB + 2>();
        if constexpr (!(MODE & 2)) SW_BARRIER();
        // second half: 4 MFMAs | the 6 reads of set 0 of k-tile t+1, two per MFMA shadow
        if constexpr (!(MODE & 4)) read_frags(0, ar[1], br[1]);
        mfma_rows(1, 1, 2);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 1);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
        __builtin_amdgcn_sched_barrier(0);
        rotate();
    }
    // tail: the last two k-tiles, nothing left to stage
    for (; t < nt; ++t) {
        mfma_rows(0, 0, 1);
        __builtin_amdgcn_sched_barrier(0);
        read_frags(1, ar[0], br[0]);
        __builtin_amdgcn_sched_barrier(0);
        mfma_rows(0, 1, 2);
        __builtin_amdgcn_sched_barrier(0);
        mfma_rows(1, 0, 1);
        __builtin_amdgcn_sched_barrier(0);
        dw_wait_vm<0>();
        SW_BARRIER();
        if (t + 1 < nt) read_frags(0, ar[1], br[1]);
        __builtin_amdgcn_sched_barrier(0);
        mfma_rows(1, 1, 2);
        __builtin_amdgcn_sched_barrier(0);
        rotate();
    }

    if constexpr (!(MODE & 8)) sw_epilogue_any<1, 1, 4>(p, acc, m0 + wr * 64, n0 + wc * 128, tile_m * (BMT / 64) + wr, lr, lk, HW);
}


// ---- the 8-wave kernel for 3x3 convolutions with the NINE TAPS OF A CHANNEL SLICE UNROLLED ("dw8u", round 3) -------------------
// Same tile, staging, rings, waits, instruction order and arithmetic as conv_igemm_dw<0> - bit-identical - but the loop
// body is one channel slice = nine k-tiles, so that everything the rolled loop recomputed per k-tile in scalar code is a
// compile-time constant: the ring stages (9 = 0 mod 3: stage = q mod 3, so the LDS addresses of the fragment reads are
// immediates and the M0 values of the DMA pieces are one s_add from a wave constant), the tap of the activation piece (nine
// 64-bit offsets from the centre pixel, computed once) and the loop control.  The rolled loop spends ~50 scalar and ~10 vector
// instructions per k-tile beside its 16 MFMAs, 12 ds_reads and 4 DMA pieces (tap -> (ky, kx) -> 64-bit byte offset by
// multiplies, three-register ring rotations, M0 arithmetic); an in-order wave pays ~5 cycles of issue for each, in the gaps
// where its partner on the SIMD would want the matrix pipe back.
template <int Q>
struct dw_const { static constexpr int value = Q; };

__global__ __launch_bounds__(512, 1) void conv_igemm_dw8u(ConvH2Args p) {
    constexpr int NW = 8, ADEPTH = 3, BMT = 256, ATILE = BMT * 64, NPB = 2, BBASE = ADEPTH * ATILE;
    __shared__ __attribute__((aligned(1024))) char smem[ADEPTH * ATILE + BDEPTH * BTILE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    int tile;
    {   // XCD-aware bijective remap (speed only)
        const int b = blockIdx.x, x = b % NXCD, q = p.tiles / NXCD, r = p.tiles % NXCD;
        tile = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + b / NXCD;
    }
    const int tile_n = tile % p.tiles_n, tile_m = tile / p.tiles_n;
    const int m0 = tile_m * BMT, n0 = tile_n * 256;
    const int HW = p.H * p.W, Wp = p.W + 2;
    const int nsl = p.C / 32;                   // channel slices; k-tile t = 9 * slice + tap

    // ---- staging (as conv_igemm_dw): wave w fills rows [32 w, 32 w + 32) of both tiles, 16 rows per DMA instruction
    const int lrow = lane >> 2;
    const int ls = (lane & 3) ^ ((lrow >> 2) & 3);
    const char* actr[2];                        // centre pixel of the lane's A row, + slot, + 64 bytes per finished slice
    const char* bptr[NPB];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int m = m0 + wave * 32 + it * 16 + lrow;
        const int b = m / HW, rem = m - b * HW;
        const int oy = rem / p.W, ox = rem - oy * p.W;
        actr[it] = p.x + ((size_t)(b * (p.H + 2) + oy + 1) * Wp + ox + 1) * p.C * 2 + ls * 16;
    }
#pragma unroll
    for (int it = 0; it < NPB; ++it) {
        const int n = n0 + wave * 32 + it * 16 + lrow;
        bptr[it] = p.w + (size_t)(n >> 5) * p.K * 64 + (n & 31) * 16 + ls * 512;
    }
    // byte offset of the activation k-tile staged in position q of a slice (= k-tile q + 2: tap (q + 2) mod 9, of the NEXT slice
    // for q >= 7) from the centre pixel of the current slice
    long long toffx[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) {
        const int tap = (q + 2) % 9, ky = tap / 3, kx = tap - ky * 3;
        toffx[q] = ((long long)(ky - p.pad) * Wp + (kx - p.pad)) * p.C * 2 + (q + 2 >= 9 ? 64 : 0);
    }
    const int wdst = wave * 32 * 64;            // this wave's rows inside an A or B stage
    auto pieceA = [&](long long off, int stage, int it) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(actr[it] + off),
                                         (__attribute__((address_space(3))) void*)(smem + stage * ATILE + wdst + it * 1024), 16, 0, 0);
    };
    auto pieceB = [&](int stage, int it) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)bptr[it],
                                         (__attribute__((address_space(3))) void*)(smem + BBASE + stage * BTILE + wdst + it * 1024), 16, 0, 0);
        bptr[it] += 2048;
    };

    // ---- fragments (as conv_igemm_dw)
    const int lr = lane & 31, lk = lane >> 5;
    const char* afr[2];                         // [k16 step]: the lane's row of the wave's first MFMA tile in stage 0, + slot
    const char* bfr[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int so = ((s * 2 + lk) ^ ((lr >> 2) & 3)) << 4;
        afr[s] = smem + (wr * 64 + lr) * 64 + so;
        bfr[s] = smem + BBASE + (wc * 128 + lr) * 64 + so;
    }
    half8 fa[2][2], fb[2][4];
    auto readA = [&](int set, int stage, int i) { fa[set][i] = *reinterpret_cast<const half8*>(afr[set] + stage * ATILE + i * 2048); };
    auto readB = [&](int set, int stage, int j) { fb[set][j] = *reinterpret_cast<const half8*>(bfr[set] + stage * BTILE + j * 2048); };
    auto read_frags = [&](int set, int stage) {
        readA(set, stage, 0);
        readA(set, stage, 1);
#pragma unroll
        for (int j = 0; j < 4; ++j) readB(set, stage, j);
    };

    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    auto mfma_rows = [&](int set, int i0, int i1) {
#pragma unroll
        for (int i = i0; i < i1; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[set][i], fb[set][j], acc[i][j], 0, 0, 0);
    };

    // ---- prologue: B(0), A(0), A(1), B(1) in flight, in that order (see conv_igemm_dw)
    {
        const long long t0 = ((long long)(0 - p.pad) * Wp + (0 - p.pad)) * p.C * 2, t1 = t0 + (long long)p.C * 2;
        pieceB(0, 0); pieceB(0, 1);
        pieceA(t0, 0, 0); pieceA(t0, 0, 1);
        pieceA(t1, 1, 0); pieceA(t1, 1, 1);
        pieceB(1, 0); pieceB(1, 1);
    }
    dw_wait_vm<NPB + 2>();
    SW_BARRIER();
    read_frags(0, 0);
    // static priority for the second-dispatched half of the workgroup (MI355X_MICROARCH.md, "Two waves per SIMD", item 4): waves
    // 4-7 lose the issue arbitration to the older wave of their SIMD on every segment; p.stagger carries DP_H2_DW_PRIO here
    if (p.stagger != 0 && wave >= 4) __builtin_amdgcn_s_setprio(1);

    // steady-state k-tile in position q of a slice: k-tiles t + 2 (both operands) are staged
    auto iter = [&](auto Q) __attribute__((always_inline)) {
        constexpr int q = decltype(Q)::value, s0 = q % 3, s1 = (q + 1) % 3, s2 = (q + 2) % 3;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            if (k < 2) readA(1, s0, k);
            else readB(1, s0, k - 2);
            if (k < NPB) pieceB(s2, k);
            else if (k < NPB + 2) pieceA(toffx[q], s2, k - NPB);
        }
        mfma_rows(0, 0, 2);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            if (k < NPB + 2) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        mfma_rows(1, 0, 1);
        __builtin_amdgcn_sched_barrier(0);
        dw_wait_vm<NPB + 2>();                  // [B(t+1), A(t+1)] of the previous k-tile landed; [B(t+2), A(t+2)] may fly
        SW_BARRIER();
        read_frags(0, s1);
        mfma_rows(1, 1, 2);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 1);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
        __builtin_amdgcn_sched_barrier(0);
    };
    // the last two k-tiles of the convolution (positions 7 and 8 of the last slice): nothing left to stage
    auto tail = [&](auto Q) __attribute__((always_inline)) {
        constexpr int q = decltype(Q)::value, s0 = q % 3, s1 = (q + 1) % 3;
        mfma_rows(0, 0, 1);
        __builtin_amdgcn_sched_barrier(0);
        read_frags(1, s0);
        __builtin_amdgcn_sched_barrier(0);
        mfma_rows(0, 1, 2);
        __builtin_amdgcn_sched_barrier(0);
        mfma_rows(1, 0, 1);
        __builtin_amdgcn_sched_barrier(0);
        dw_wait_vm<0>();
        SW_BARRIER();
        if (q < 8) read_frags(0, s1);
        __builtin_amdgcn_sched_barrier(0);
        mfma_rows(1, 1, 2);
        __builtin_amdgcn_sched_barrier(0);
    };

    for (int s = 0; s + 1 < nsl; ++s) {
        iter(dw_const<0>{}); iter(dw_const<1>{}); iter(dw_const<2>{});
        iter(dw_const<3>{}); iter(dw_const<4>{}); iter(dw_const<5>{});
        iter(dw_const<6>{}); iter(dw_const<7>{}); iter(dw_const<8>{});
        actr[0] += 64;
        actr[1] += 64;
    }
    iter(dw_const<0>{}); iter(dw_const<1>{}); iter(dw_const<2>{});
    iter(dw_const<3>{}); iter(dw_const<4>{}); iter(dw_const<5>{});
    iter(dw_const<6>{});
    tail(dw_const<7>{});
    tail(dw_const<8>{});

    sw_epilogue_any<1, 1, 4>(p, acc, m0 + wr * 64, n0 + wc * 128, tile_m * (BMT / 64) + wr, lr, lk, HW);
}

}  // namespace

bool dp_conv_dw_applies(const ConvH2Args& p) {
    const bool seg_ok = (!p.seg1 || (p.segC1 > 0 && p.segC1 % 32 == 0)) && (!p.seg2 || (p.seg1 && p.segC2 > 0 && p.segC2 % 32 == 0));
    return p.wfmt == 1 && p.afmt == 1 && p.passes == 1 && p.ksplit == 1 && p.M % 256 == 0 && p.N % 256 == 0 && p.C % 32 == 0 &&
           p.K >= 4 * 32 && (!p.temb || (p.H * p.W) % 32 == 0) && seg_ok && (p.rfmt == 0 || p.ofmt == 1);
}

void dp_launch_conv_dw(ConvH2Args& p, hipStream_t s) {
    p.tiles_n = p.N / 256;
    p.tiles = (p.M / 256) * p.tiles_n;
    p.stagger = 0;
    const dim3 g((unsigned)p.tiles), b(512u);
    // 3x3 without K-segments: the slice-unrolled form (DP_H2_DW_UNROLL=0: the rolled loop)
    const bool unrolled = p.KS == 3 && !p.seg1 && dp_tune(DP_T_H2_DW_UNROLL) != 0;
#ifdef DP_ABLATE   // timing ablations (WRONG RESULTS): only in libdiffpure_hip_ablate.so (tests/probes/build_ablate.py)
    {
        const char* e = getenv("DP_H2_DW_MODE");
        switch (e ? atoi(e) : 0) {
#define DW_CASE(M_) case M_: hipLaunchKernelGGL((conv_igemm_dw<M_>), g, b, 0, s, p); return
            DW_CASE(1); DW_CASE(2); DW_CASE(3); DW_CASE(6); DW_CASE(4); DW_CASE(7); DW_CASE(8); DW_CASE(16); DW_CASE(32);
#undef DW_CASE
            default: break;
        }
    }
#endif
    if (unrolled) {
        p.stagger = dp_tune(DP_T_H2_DW_PRIO);
        hipLaunchKernelGGL(conv_igemm_dw8u, g, b, 0, s, p);
    } else {
        hipLaunchKernelGGL((conv_igemm_dw<0>), g, b, 0, s, p);
    }
}
