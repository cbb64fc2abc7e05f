// fp16 x fp16 implicit-GEMM convolution, 128 (pixels) x 256 (channels) tile, TWO WORKGROUPS PER CU ("dw").
//
// What bounds the one-wave-per-SIMD kernel (igemm_h2_sw.hip): (i) a lone in-order wave exposes every
// cycle an LDS-DMA issue or a ds_read_b128 costs beyond the 32-cycle shadow of one MFMA (its k-loop runs at 61 % of the
// matrix rate), and (ii) a tile ends in ~18 us during which the CU's matrix pipes idle: all 256 CUs store their 256 KB
// tiles (and read their residual tiles) at the same moment - a 64 + 64 MB burst that HBM takes 13-16 us to absorb - then
// start the next tile's prologue together.  Both are the SAME defect: nothing else is resident on the CU to use the pipe.
//
// This kernel halves the tile so that two workgroups fit a CU (4 waves each, one per SIMD -> two waves per SIMD; 128 of the
// 256 registers a wave may now use are accumulators; 72 or 80 KB of LDS each) and lets them run FREE of each other: no
// shared barrier, no role assignment (unlike the 8-wave ping-pong kernel, whose load segment outlasts its MFMA segment).
// Whenever one wave of a SIMD waits - for an LDS-DMA issue slot, its vmcnt, its workgroup's barrier, or through its whole
// epilogue - the other wave's MFMAs take the pipe.  The second workgroup to arrive on a CU (a ticket per CU, read from
// HW_ID) starts half a tile late, so the two epilogues of a CU alternate instead of coinciding, and the chip's store burst is
// halved and spread.  Price: (128 + 256) instead of (256 + 256) operand rows per 128 x 256 x 32 products - 1.5x the LDS-DMA
// and ds_read traffic per MFMA.
//
// Wave tile 64 x 128 = 2 x 4 MFMA tiles of 32x32; per k-tile (32 channels of one tap) a wave issues 16 MFMAs, 12
// ds_read_b128 and 6 LDS-DMA pieces.  The two operands have separate LDS rings: weights (L2-resident) three stages
// = prefetch distance 2, activations (the operand that misses to HBM once per nine taps) ADEPTH stages = distance ADEPTH - 1.
//
//   iteration t:  issue DMA: weights of k-tile t+2, then activations of k-tile t+DA     | 8 MFMA (t, s=0), ds_read (t, s=1)
//                 4 MFMA (t, s=1, row 0) ; s_waitcnt vmcnt(in-order count: k-tile t+1 landed) ; s_barrier
//                 ds_read fragments (t+1, s=0)                                          | 4 MFMA (t, s=1, row 1)
//   RAW: every wave waits for its own share of k-tile t+1 before the barrier of iteration t; the reads follow it.
//   WAR: the stages written in iteration t held k-tiles t-1 (weights) / t-1 (activations, ring of DA+1) whose last reads
//        precede the barrier of iteration t-1.
// Same operand formats, reduction order and epilogue arithmetic as every other variant: bit-identical output.
// Needs: fp16 activations and weights (a_fmt 1, w_fmt 1, passes 1), M % 128 == 0, N % 256 == 0, C % 32 == 0, >= 4 k-tiles.
#include <stdlib.h>

#include "dp_tune.h"
#include "igemm_h2.h"
#include "igemm_sw_common.h"

namespace {

constexpr int NXCD = 8;
constexpr int BTILE = 256 * 64;                 // weight tile of one k-tile: 256 rows x 64 bytes (32 fp16)
constexpr int BDEPTH = 3;

// one arrival counter per CU (index: XCC id, then bits 15:8 of HW_ID = CU / SH / SE ids); parity decides who starts late
__device__ unsigned dw_cu_ticket[8 * 256];

template <int N>
__device__ __forceinline__ void dw_wait_vm() {
    static_assert(N == 8 || N == 6 || N == 4 || N == 0, "add the immediate");
    if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// MODE (timing ablations, DP_ABLATE builds only; WRONG RESULTS): 1 = no DMA in the steady state, 2 = no barrier / vmcnt wait,
// 4 = no ds_reads, 8 = no epilogue stores, 16 = no activation DMA, 32 = no weight DMA
// NW = 4: the kernel described above (128 x 256 tile, two workgroups per CU).
// NW = 8 (round 3, "one workgroup, two free-running waves per SIMD"): ONE workgroup of eight waves per CU on a 256 x 256 tile - the
// operand traffic of the one-wave-per-SIMD kernel (32 KB per k-tile: the two waves of a SIMD share the tile in LDS) with two
// waves per SIMD to fill each other's issue stalls, at 1.5x that kernel's ds_reads per MFMA (wave tile 64 x 128).  Unlike the
// ping-pong kernel (same geometry) the waves are not assigned roles: every wave runs the interleaved MFMA / read / DMA stream
// and meets the others at ONE barrier per k-tile.  The epilogues of its waves still coincide (one tile per CU at a time).
template <int ADEPTH, int MODE, int NW>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void conv_igemm_dw(ConvH2Args p) {
    constexpr int DA = ADEPTH - 1;              // prefetch distance of the activation ring (weights: 2)
    constexpr int BMT = NW * 32;                // tile rows: every wave stages 32 of them (2 pieces) and 256 / NW weight rows
    constexpr int ATILE = BMT * 64;             // activation tile of one k-tile
    constexpr int NPB = 256 / NW / 16;          // weight pieces per wave and k-tile (4 | 2)
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

    // ---- de-phase the two workgroups of a CU: of the launch's first residents, the second arrival on a CU sleeps for about
    // half a tile (p.stagger cycles per k-tile); every later workgroup inherits the phase of the one it replaces
    if (NW == 4 && p.stagger > 0 && blockIdx.x < 512) {
        unsigned* flag = reinterpret_cast<unsigned*>(smem + BBASE + 2 * BTILE);    // a stage nothing writes before iteration 0
        if (tid == 0) {
            const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);        // HW_REG_HW_ID
            const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 7;  // HW_REG_XCC_ID
            *flag = atomicAdd(&dw_cu_ticket[xcc * 256 + ((hw >> 8) & 0xFF)], 1u);
        }
        __syncthreads();
        const unsigned ticket = __builtin_amdgcn_readfirstlane(*flag);
        if (ticket & 1) {
            const long long delay = (long long)nt * p.stagger;
            const long long t0 = __builtin_readcyclecounter();
            while ((long long)__builtin_readcyclecounter() - t0 < delay) __builtin_amdgcn_s_sleep(32);
        }
    }

    // ---- staging: wave w fills rows [32 w, 32 w + 32) of the A tile and [64 w, 64 w + 64) of the B tile, 16 rows per DMA
    // instruction; lane -> row (lane >> 2) of the piece, physical slot lane & 3, logical slot XOR-ed with the row key
    const int lrow = lane >> 2;
    const int ls = (lane & 3) ^ ((lrow >> 2) & 3);
    const char* actr[2];                        // centre pixel of the lane's A row, + slot
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
        const int n = n0 + wave * (256 / NW) + it * 16 + lrow;      // block layout of the fp16 panels (ops.order_conv_weight_w16)
        bptr[it] = p.w + (size_t)(n >> 5) * p.K * 64 + (n & 31) * 16 + ls * 512;
    }
    int cur_tap = 0, cur_c = 0;                 // (tap, slice) of the next activation k-tile to stage
    long long a_off = 0;
    auto pieceA = [&](int aoff, int it) {       // aoff: byte offset of the ring stage
        if (it == 0) {
            const int ky = p.KS == 3 ? (cur_tap * 11) >> 5 : 0, kx = cur_tap - ky * p.KS;     // tap / 3 for tap < 9, no division
            a_off = ((long long)(ky - p.pad) * Wp + (kx - p.pad)) * p.C * 2 + (long long)cur_c * 64;
        }
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(actr[it] + a_off),
                                         (__attribute__((address_space(3))) void*)(smem + aoff + (wave * 32 + it * 16) * 64), 16, 0, 0);
        if (it == 1 && ++cur_tap == taps) { cur_tap = 0; ++cur_c; }
    };
    auto pieceB = [&](int boff, int it) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)bptr[it],
                                         (__attribute__((address_space(3))) void*)(smem + boff + (wave * (256 / NW) + it * 16) * 64), 16, 0, 0);
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
    // ar[DA] / br[2] are the stages the iteration writes
    int ar[ADEPTH], br[BDEPTH];
#pragma unroll
    for (int i = 0; i < ADEPTH; ++i) ar[i] = i * ATILE;
#pragma unroll
    for (int i = 0; i < BDEPTH; ++i) br[i] = BBASE + i * BTILE;
    auto rotate = [&]() {
        const int a0 = ar[0], b0 = br[0];
#pragma unroll
        for (int i = 0; i + 1 < ADEPTH; ++i) ar[i] = ar[i + 1];
        ar[ADEPTH - 1] = a0;
        br[0] = br[1];
        br[1] = br[2];
        br[2] = b0;
    };

    // ---- prologue (nt >= 4): B(0), A(0), A(1), B(1), [A(2)] in flight - in THAT order, because vmcnt counts in issue order
    // and the steady-state wait "everything up to the weights of k-tile t+1" must leave only younger pieces outstanding
    issueB(br[0]);
    issueA(ar[0]);
    issueA(ar[1]);
    issueB(br[1]);
    if constexpr (DA == 3) issueA(ar[2]);
    dw_wait_vm<NPB + 2 + (DA == 3 ? 2 : 0)>();  // k-tile 0 landed; A(1), B(1), [A(2)] may fly
    SW_BARRIER();
    read_frags(0, ar[0], br[0]);

    // steady state: k-tiles t+2 (weights) and t+DA (activations) exist
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
        // outstanding in issue order: [.., B(t+1), A(t+DA-1)] from iteration t-1, [B(t+2), A(t+DA)] from this one
        if constexpr (!(MODE & 3)) dw_wait_vm<NPB + 2 + (DA == 3 ? 2 : 0)>();
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
    // tail: the last DA k-tiles; with DA == 3 the first of them still stages the weights of the last k-tile
    for (; t < nt; ++t) {
        if (DA == 3 && t + 2 < nt) issueB(br[2]);
        mfma_rows(0, 0, 1);
        __builtin_amdgcn_sched_barrier(0);
        read_frags(1, ar[0], br[0]);
        __builtin_amdgcn_sched_barrier(0);
        mfma_rows(0, 1, 2);
        __builtin_amdgcn_sched_barrier(0);
        mfma_rows(1, 0, 1);
        __builtin_amdgcn_sched_barrier(0);
        // DA == 3, t == nt-3: outstanding [B(nt-2), A(nt-1)], [B(nt-1)] -> the 2 + NPB pieces behind B(nt-2) may fly
        if (DA == 3 && t + 2 < nt) dw_wait_vm<NPB + 2>();
        else dw_wait_vm<0>();
        SW_BARRIER();
        if (t + 1 < nt) read_frags(0, ar[1], br[1]);
        __builtin_amdgcn_sched_barrier(0);
        mfma_rows(1, 1, 2);
        __builtin_amdgcn_sched_barrier(0);
        rotate();
    }

    if constexpr (!(MODE & 8)) sw_epilogue_any<1, 1>(p, acc, m0 + wr * 64, n0 + wc * 128, tile_m * (BMT / 64) + wr, lr, lk, HW);
}


// ---- the 8-wave kernel for 3x3 convolutions with the NINE TAPS OF A CHANNEL SLICE UNROLLED ("dw8u", round 3) -------------------
// Same tile, staging, rings, waits, instruction order and arithmetic as conv_igemm_dw<3, 0, 8> - bit-identical - but the loop
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

    sw_epilogue_any<1, 1>(p, acc, m0 + wr * 64, n0 + wc * 128, tile_m * (BMT / 64) + wr, lr, lk, HW);
}

}  // namespace

bool dp_conv_dw_applies(const ConvH2Args& p, int waves) {
    return (waves == 4 || waves == 8) && p.wfmt == 1 && p.afmt == 1 && p.passes == 1 && p.ksplit == 1 && p.M % (waves * 32) == 0 &&
           p.N % 256 == 0 && p.C % 32 == 0 && p.K >= 4 * 32 && (!p.temb || (p.H * p.W) % 32 == 0);
}

void dp_launch_conv_dw(ConvH2Args& p, hipStream_t s, int waves) {
    p.tiles_n = p.N / 256;
    p.tiles = (p.M / (waves * 32)) * p.tiles_n;
    // start-up stagger (two workgroups per CU only): only where the launch runs long enough to earn it back (rounds of 512)
    p.stagger = waves == 4 && p.tiles >= 512 * dp_tune(DP_T_H2_DW_MINROUNDS) ? dp_tune(DP_T_H2_DW_STAGGER) : 0;
    const int adepth = dp_tune(DP_T_H2_DW_ADEPTH);
    const dim3 g((unsigned)p.tiles), b((unsigned)(waves * 64));
    // 3x3, 8 waves, ring depth 3: the slice-unrolled form (DP_H2_DW_UNROLL=0: the rolled loop)
    const bool unrolled = waves == 8 && adepth != 4 && p.KS == 3 && dp_tune(DP_T_H2_DW_UNROLL) != 0;
#define DW_LAUNCH(M_)                                                                              \
    do {                                                                                           \
        if (waves == 8 && adepth == 4) hipLaunchKernelGGL((conv_igemm_dw<4, M_, 8>), g, b, 0, s, p);  \
        else if (waves == 8) hipLaunchKernelGGL((conv_igemm_dw<3, M_, 8>), g, b, 0, s, p);            \
        else if (adepth == 4) hipLaunchKernelGGL((conv_igemm_dw<4, M_, 4>), g, b, 0, s, p);           \
        else hipLaunchKernelGGL((conv_igemm_dw<3, M_, 4>), g, b, 0, s, p);                            \
    } while (0)
#ifdef DP_ABLATE   // timing ablations (WRONG RESULTS): only in libdiffpure_hip_ablate.so (tests/probes/build_ablate.py)
    {
        const char* e = getenv("DP_H2_DW_MODE");
        switch (e ? atoi(e) : 0) {
            case 1: DW_LAUNCH(1); return;
            case 2: DW_LAUNCH(2); return;
            case 3: DW_LAUNCH(3); return;
            case 6: DW_LAUNCH(6); return;
            case 4: DW_LAUNCH(4); return;
            case 7: DW_LAUNCH(7); return;
            case 8: DW_LAUNCH(8); return;
            case 16: DW_LAUNCH(16); return;
            case 32: DW_LAUNCH(32); return;
            default: break;
        }
    }
#endif
    if (unrolled) {
        p.stagger = dp_tune(DP_T_H2_DW_PRIO);       // (the start-up stagger belongs to the two-workgroups-per-CU form only)
        hipLaunchKernelGGL(conv_igemm_dw8u, g, b, 0, s, p);
    }
    else DW_LAUNCH(0);
#undef DW_LAUNCH
}
