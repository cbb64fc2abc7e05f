// fp16 x fp16 implicit-GEMM convolution, 256x256 tile, ONE WAVE PER SIMD, software-pipelined ("sw").
//
// The 8-wave ping-pong kernel (igemm_h2_pp.hip) is bound, with one MFMA pass per product, by the load
// segment of a phase: per k-tile a wave has 16 MFMAs (512 cycles) against 12 ds_read_b128, 2-4 LDS-DMA issues and two
// barrier pairs, and the partner wave's load segment outlasts its own MFMA segment.  This kernel
// changes the ratio instead of the schedule: 4 waves per workgroup, one per SIMD, each owning a 128 x 128 wave tile
// (4 x 4 MFMA tiles of 32x32 = 256 accumulator registers of the 512 a lone wave may use).  Per k-tile (32 channels of one
// tap) a wave issues 32 MFMAs (1024 cycles) and only 16 ds_read_b128 + 8 LDS-DMA, which the compiler interleaves into the
// MFMA shadows (an MFMA hides up to ~5 single-issue instructions, MI355X_MICROARCH.md); fragments are double-buffered in
// registers (the reads of k16 step s+1 fly under the MFMAs of step s), operands are staged three k-tiles ahead into a
// four-stage LDS ring (128 KB), and there is ONE barrier per k-tile:
//
//   iteration t:  ds_read fragments (t, s=1) ; then  issue DMA of k-tile t+3 -> stage (t+3) % 4  | 16 MFMA (t, s=0)
//                 s_waitcnt vmcnt(16)  [k-tile t+1 of THIS wave has landed; t+2, t+3 may fly] ; s_barrier
//                 ds_read fragments (t+1, s=0)          | 16 MFMA (t, s=1)
//   RAW: every wave waits for its own share of k-tile t+1 before the barrier of iteration t; the reads follow it.
//   WAR: stage (t+3) % 4 held k-tile t-1, whose last reads (s=1) precede the barrier of iteration t-1, which the issuing
//        wave has passed and every other wave has reached.
// Same operand formats, reduction order and epilogue arithmetic as the other variants: bit-identical output.
// Needs: fp16 activations and weights (a_fmt 1, w_fmt 1, passes 1), M % 256 == 0, N % 256 == 0, C % 32 == 0.
#include <stdlib.h>

#include "dp_tune.h"
#include "igemm_h2.h"
#include "igemm_sw_common.h"

namespace {

constexpr int NT = 256;
constexpr int NXCD = 8;
constexpr int NB = 4, DIST = 3;                 // LDS ring stages, prefetch distance in k-tiles

template <int N>
__device__ __forceinline__ void sw_wait_vm() {
    if constexpr (N == 20) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
    else if constexpr (N == 18) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
    else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else {
        static_assert(N == 0, "add the immediate");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}

// MODE (timing ablations, DP_H2_SW_MODE, DP_ABLATE builds only; WRONG RESULTS): 1 = no DMA in the steady state, 2 = no barrier / vmcnt wait, 4 = no ds_reads,
// 8 = no activation DMA, 16 = no weight DMA
// BN = 256: 256 x 256 tile, waves 2 (M) x 2 (N).  BN = 128 (round 3; layers with 128 output channels - the 32 x 32 level of the
// CIFAR-10 NCSN++, until then on the 8-wave ping-pong kernel at 566 TFLOP/s): 512 x 128 tile, waves 4 (M) x 1 (N), 40 KB per
// LDS stage = all 160 KB.  The wave tile is 128 x 128 in both; a wave stages BM / 64 activation pieces and BN / 64 weight pieces
// per k-tile.
// The DMA issues of a k-tile are SPREAD - behind every second fragment read, the activation pieces in the first half of the
// k-tile and the weight pieces in the second - instead of back to back in consecutive MFMA shadows (an LDS-DMA issue costs
// more than one 32-cycle shadow; consecutive ones queue up in front of the next MFMA: +1...2.5 % measured in round 2).
// (Round 3 also built a persistent form - one workgroup per CU walking its tiles, the next tile's first three k-tiles put in
//  flight before the epilogue: bit-identical, measured 2-3 % SLOWER (the hardware's own dispatch re-balances and de-phases the
//  CUs), removed in round 4: git history, profiles/r03/sw_persistent_ab.log.)
template <int MODE, int BN>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv_igemm_sw(ConvH2Args p) {
    constexpr int BM = BN == 256 ? 256 : 512;
    constexpr int TILE_A = BM * 64, TILE_B = BN * 64;       // one operand tile of a k-tile: rows x 64 bytes (32 fp16)
    constexpr int STAGE = TILE_A + TILE_B;                  // A tile, then B tile
    constexpr int NPA = BM / 64, NPB = BN / 64;             // DMA pieces (16 rows) per wave and k-tile
    // the operand ring; after the k-loop the same memory is the landing zone of the fp16 residual (igemm_sw_common.h, 136 KB)
    __shared__ __attribute__((aligned(1024))) char smem[NB * STAGE > SW_EPI_LDS ? NB * STAGE : SW_EPI_LDS];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = BN == 256 ? wave >> 1 : wave, wc = BN == 256 ? wave & 1 : 0;
    const int HW = p.H * p.W, Wp = p.W + 2, taps = p.KS * p.KS;
    const int nt = p.K / 32;
    // XCD-aware bijective remap of the workgroup id to a tile (speed only)
    auto tile_of = [&](int v) {
        const int x = v % NXCD, q = p.tiles / NXCD, r = p.tiles % NXCD;
        return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + v / NXCD;
    };
    int tile_m = 0, tile_n = 0, m0 = 0, n0 = 0;

    // ---- staging: wave w fills rows [w BM/4, (w+1) BM/4) of the A tile and [w BN/4, (w+1) BN/4) of the B tile, 16 rows per DMA
    // instruction; lane -> row (lane >> 2) of the piece, physical slot lane & 3, logical slot XOR-ed with the row key (row >> 2) & 3
    const int lrow = lane >> 2;
    const int ls = (lane & 3) ^ ((lrow >> 2) & 3);
    const char* actr[NPA];                      // centre pixel of the lane's A row, + slot
    const char* bptr[NPB];
    // (tap, slice) of the next k-tile to stage inside the current K-segment: segment 0 = the KS x KS convolution over p.x, then the
    // 1x1 segments over the plain fp16 tensors p.seg1 / p.seg2 (igemm_h2.h; one k-tile per 32-channel slice)
    int cur_tap = 0, cur_c = 0, cur_seg = 0, seg_slices = p.C / 32;
    int nm0 = 0, nn0 = 0, ntile_m = 0, ntile_n = 0;     // the tile the staging pointers belong to
    auto stage_setup = [&](int tile) {
        ntile_n = tile % p.tiles_n;
        ntile_m = tile / p.tiles_n;
        nm0 = ntile_m * BM;
        nn0 = ntile_n * BN;
#pragma unroll
        for (int it = 0; it < NPA; ++it) {
            const int m = nm0 + wave * (BM / 4) + it * 16 + lrow;
            const int b = m / HW, rem = m - b * HW;
            const int oy = rem / p.W, ox = rem - oy * p.W;
            actr[it] = p.x + ((size_t)(b * (p.H + 2) + oy + 1) * Wp + ox + 1) * p.C * 2 + ls * 16;
        }
#pragma unroll
        for (int it = 0; it < NPB; ++it) {
            const int n = nn0 + wave * (BN / 4) + it * 16 + lrow;   // block layout of the fp16 panels (ops.order_conv_weight_w16)
            bptr[it] = p.w + (size_t)(n >> 5) * p.K * 64 + (n & 31) * 16 + ls * 512;
        }
        cur_tap = 0;
        cur_c = 0;
        cur_seg = 0;
        seg_slices = p.C / 32;
    };
    auto adopt = [&]() { tile_m = ntile_m; tile_n = ntile_n; m0 = nm0; n0 = nn0; };
    // one piece at a time, in PROGRAM order between the fragment reads (an LDS-DMA write and a ds_read may alias as far
    // as the scheduler knows: it never moves one across the other, so the interleave has to be written)
    long long a_off = 0;
    auto pieceA = [&](int stage, int it) {
        if (it == 0) {
            if (cur_c == seg_slices) {          // (wave-uniform) this segment is staged: on to the next tensor
                ++cur_seg;
                const char* sb = cur_seg == 1 ? p.seg1 : p.seg2;
                const int sc = cur_seg == 1 ? p.segC1 : p.segC2;
#pragma unroll
                for (int j = 0; j < NPA; ++j) actr[j] = sb + (size_t)(nm0 + wave * (BM / 4) + j * 16 + lrow) * sc * 2 + ls * 16;
                seg_slices = sc / 32;
                cur_c = 0;
            }
            if (cur_seg == 0) {
                const int ky = p.KS == 3 ? (cur_tap * 11) >> 5 : 0, kx = cur_tap - ky * p.KS;       // tap / 3 for tap < 9, no division
                a_off = ((long long)(ky - p.pad) * Wp + (kx - p.pad)) * p.C * 2 + (long long)cur_c * 64;
            } else {
                a_off = (long long)cur_c * 64;
            }
        }
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(actr[it] + a_off),
                                         (__attribute__((address_space(3))) void*)(smem + stage * STAGE + (wave * (BM / 4) + it * 16) * 64), 16, 0, 0);
        if (it == NPA - 1 && (cur_seg != 0 || ++cur_tap == taps)) { cur_tap = 0; ++cur_c; }
    };
    auto pieceB = [&](int stage, int it) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)bptr[it],
                                         (__attribute__((address_space(3))) void*)(smem + stage * STAGE + TILE_A + (wave * (BN / 4) + it * 16) * 64), 16, 0, 0);
        bptr[it] += 2048;
    };
    auto issue = [&](int stage) {
#pragma unroll
        for (int it = 0; it < NPA; ++it) pieceA(stage, it);
#pragma unroll
        for (int it = 0; it < NPB; ++it) pieceB(stage, it);
    };
    // (tried in round 3: scalar 64-bit bases advanced once per k-tile + fixed 32-bit lane offsets, to take the 64-bit vector
    //  adds out of the loop - the instruction selector only forms the SADDR addressing mode when the zero-extension of the lane
    //  offset sits in the loop's own basic block, the optimiser hoists it, and the variants that pin it there pushed the
    //  loop-carried state into scratch.  Not pursued: the adds sit in MFMA shadows.)

    // ---- fragments: lane -> row lr of a 32-row MFMA tile, k-half lk; 64-byte rows, slot (s*2 + lk) ^ key, key = (row >> 2) & 3
    const int lr = lane & 31, lk = lane >> 5;
    const int arow = (wr * 128 + lr) * 64;                  // + i * 32 * 64
    const int brow = TILE_A + (wc * 128 + lr) * 64;         // + j * 32 * 64
    int soff[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) soff[s] = ((s * 2 + lk) ^ ((lr >> 2) & 3)) << 4;
    half8 fa[2][4], fb[2][4];                               // [register set = k16 step][tile]
    auto read_frags = [&](int set, const char* st) {
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[set][i] = *reinterpret_cast<const half8*>(st + arow + i * 32 * 64 + soff[set]);
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[set][j] = *reinterpret_cast<const half8*>(st + brow + j * 32 * 64 + soff[set]);
    };
    // fragment reads of one k16 step, two at a time: pair q = 0, 1 -> A tiles (2q, 2q+1); q = 2, 3 -> B tiles (2q-4, 2q-3)
    auto read_pair = [&](int set, const char* st, int q) {
        if (q < 2) {
            fa[set][2 * q] = *reinterpret_cast<const half8*>(st + arow + (2 * q) * 32 * 64 + soff[set]);
            fa[set][2 * q + 1] = *reinterpret_cast<const half8*>(st + arow + (2 * q + 1) * 32 * 64 + soff[set]);
        } else {
            fb[set][2 * q - 4] = *reinterpret_cast<const half8*>(st + brow + (2 * q - 4) * 32 * 64 + soff[set]);
            fb[set][2 * q - 3] = *reinterpret_cast<const half8*>(st + brow + (2 * q - 3) * 32 * 64 + soff[set]);
        }
    };

    f32x16 acc[4][4];
    auto mfma_rows = [&](int set, int i0, int i1) {           // MFMA tile rows [i0, i1) of k16 step `set`
#pragma unroll
        for (int i = i0; i < i1; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[set][i], fb[set][j], acc[i][j], 0, 0, 0);
    };

    // ---- prologue of a tile: k-tiles 0 .. DIST-1 in flight
    auto prologue_issue = [&]() {
#pragma unroll
        for (int d = 0; d < DIST; ++d)
            if (d < nt) issue(d);
    };
    stage_setup(tile_of(blockIdx.x));
    prologue_issue();
    adopt();
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // k-tile 0 landed, fragments (0, s=0) read
    if (nt > 2) sw_wait_vm<2 * (NPA + NPB)>();
    else if (nt > 1) sw_wait_vm<NPA + NPB>();
    else sw_wait_vm<0>();
    SW_BARRIER();
    read_frags(0, smem);

    // steady state (one basic block: the compiler interleaves the loads with the MFMAs): k-tile t+DIST exists
    int t = 0;
    for (; t + DIST < nt; ++t) {
        const char* st = smem + (t & (NB - 1)) * STAGE;
        // Instruction order, pinned with sched_group_barrier (left alone the compiler sinks the fragment reads towards their
        // use, and the wave then waits for LDS with an idle matrix pipe).  Each half of a k-tile runs 16 MFMAs on one fragment
        // set and, BEHIND its first MFMA, issues the 8 reads of the other set and the DMA pieces of k-tile t+3 in MFMA shadows:
        // the lgkmcnt wait before a half's first MFMA finds reads issued >= 4 MFMAs earlier.
        // first half: 16 MFMAs | 8 reads and the NPA activation pieces of k-tile t+3: (read, read, DMA [, DMA]) x 4
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if constexpr (!(MODE & 4)) read_pair(1, st, q);
#pragma unroll
            for (int e = 0; e < NPA / 4; ++e)
                if constexpr (!(MODE & 9)) pieceA((t + DIST) & (NB - 1), q * (NPA / 4) + e);
        }
        mfma_rows(0, 0, 4);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            if constexpr (NPA == 4) {       // read | MFMA | read | MFMA | DMA | MFMA
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            } else {                        // read | MFMA | read, DMA | MFMA | DMA | MFMA
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            }
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
        __builtin_amdgcn_sched_barrier(0);
        mfma_rows(1, 0, 1);
        __builtin_amdgcn_sched_barrier(0);
        // in flight at most: the activation pieces of t+3 (just issued), the weight pieces of t+2, the activation pieces of t+2
        if constexpr (!(MODE & 3)) sw_wait_vm<2 * NPA + NPB>();
        if constexpr (!(MODE & 2)) SW_BARRIER();
        // second half after the barrier: 12 MFMAs | 8 reads and the NPB weight pieces of k-tile t+3 in the same pattern.
        // (Tried in round 3: all eight reads first, two per shadow, then the pieces - no difference.  The compiler can only
        //  emit lgkmcnt(0) in front of a half: it books every LDS-DMA as a pending FLAT access to LDS, which forbids counted
        //  lgkmcnt waits, so ordering the reads by first use buys nothing either.)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if constexpr (!(MODE & 4)) read_pair(0, smem + ((t + 1) & (NB - 1)) * STAGE, q);
            if constexpr (!(MODE & 17)) {
                if (q < NPB) pieceB((t + DIST) & (NB - 1), q);
            }
        }
        mfma_rows(1, 1, 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 1);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 1);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
            if (k < NPB) {
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 1);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    // tail: the last DIST k-tiles, nothing left to stage
    for (; t < nt; ++t) {
        const char* st = smem + (t & (NB - 1)) * STAGE;
        mfma_rows(0, 0, 1);
        __builtin_amdgcn_sched_barrier(0);
        read_frags(1, st);
        __builtin_amdgcn_sched_barrier(0);
        mfma_rows(0, 1, 4);
        __builtin_amdgcn_sched_barrier(0);
        mfma_rows(1, 0, 1);
        __builtin_amdgcn_sched_barrier(0);
        if (t + 2 < nt) sw_wait_vm<NPA + NPB>();
        else sw_wait_vm<0>();
        SW_BARRIER();
        if (t + 1 < nt) read_frags(0, smem + ((t + 1) & (NB - 1)) * STAGE);
        __builtin_amdgcn_sched_barrier(0);
        mfma_rows(1, 1, 4);
        __builtin_amdgcn_sched_barrier(0);
    }

    sw_epilogue_any<2>(p, acc, m0 + wr * 128, n0 + wc * 128, tile_m * (BM / 64) + wr * 2, lr, lk, HW, smem + wave * (32 * SW_EPI_PITCH));
}


// Variants of this kernel that were built, verified bit-identical and measured SLOWER (tests/probes/pp_ablate.py, B=64; removed):
//   * weight fragments fetched straight from global memory into registers (the block layout of the fp16 panels makes a B
//     fragment 1 KB of contiguous memory): no LDS write / read / barrier dependence for B, but the two waves that share a column
//     block fetch the same bytes - 48 KB instead of 32 KB per k-tile through the vector-memory path: 891-972 vs 918-1003 TFLOP/s;
//   * activation operand as a 2-D halo tile (as round 2's halo-tile ping-pong kernel; 21.5 KB instead of 32 KB per k-tile): 885-1022 vs 946-1131
//     TFLOP/s - the per-tap address arithmetic, the data-dependent vmcnt variants and the 160 KB of LDS cost more than the
//     bytes save (its no-loads-at-all ablation is already slower than this kernel's: 1277-1599 vs 1498-1880);
//   * the x-halo form (one activation run of R x (W + 2) pixels per (channel slice, ky) serves the three kx taps: 65 DMA pieces
//     per three k-tiles instead of 96; super-tiles of three k-tiles, two 68 KB LDS stages, ONE barrier per 96 MFMAs, the 24
//     shifted fragment offsets precomputed): 876-1041 vs 907-1038 TFLOP/s - a third fewer operand bytes buy nothing, the L2 ->
//     LDS path is not the bound either, and the one-super-tile prefetch distance exposes DMA latency (its no-wait ablation
//     gains 9 %, this kernel's 3 %).
}  // namespace

bool dp_conv_sw_applies(const ConvH2Args& p, int bn) {
    return (bn == 256 || bn == 128) && p.wfmt == 1 && p.afmt == 1 && p.passes == 1 && p.ksplit == 1 && p.M % (bn == 256 ? 256 : 512) == 0 &&
           p.N % bn == 0 && p.C % 32 == 0 && (!p.temb || (p.H * p.W) % 32 == 0) && (p.rfmt == 0 || p.ofmt == 1) &&
           (p.rfmt == 0 || (dp_aligned16(p.res) && p.ldr % 8 == 0));   // 16-byte LDS-DMA pieces of the fp16 residual (igemm_sw_common.h)
}

void dp_launch_conv_sw(ConvH2Args& p, hipStream_t s, int bn) {
    p.tiles_n = p.N / bn;
    p.tiles = (p.M / (bn == 256 ? 256 : 512)) * p.tiles_n;
    const dim3 g((unsigned)p.tiles), b(NT);
#define SW_LAUNCH(M_)                                                                       \
    do {                                                                                    \
        if (bn == 256) hipLaunchKernelGGL((conv_igemm_sw<M_, 256>), g, b, 0, s, p);          \
        else hipLaunchKernelGGL((conv_igemm_sw<M_, 128>), g, b, 0, s, p);                    \
    } while (0)
#ifdef DP_ABLATE   // timing ablations (WRONG RESULTS): only in libdiffpure_hip_ablate.so (tests/probes/build_ablate.py)
    {
        const char* e = getenv("DP_H2_SW_MODE");
        switch (e ? atoi(e) : 0) {
            case 1: SW_LAUNCH(1); return;
            case 4: SW_LAUNCH(4); return;
            case 7: SW_LAUNCH(7); return;
            case 8: SW_LAUNCH(8); return;
            case 16: SW_LAUNCH(16); return;
            default: break;
        }
    }
#endif
    SW_LAUNCH(0);
#undef SW_LAUNCH
}

// Also tried on this kernel and measured SLOWER (tests/probes/pp_ablate.py, B=64): prefetching the residual tile into L2 during the
// last eight steady k-tiles (one 64-line touch per k-tile through a scratch LDS-DMA): 788-1049 vs 843-1061 TFLOP/s on the
// residual-carrying launches - the extra requests queue in front of the operand DMAs.
