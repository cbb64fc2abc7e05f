// f16x3 implicit-GEMM convolution, 256x256 tile, 8 waves in two "ping-pong" groups.
//
// Same arithmetic, operand formats and per-element accumulation order as conv_igemm_h2 (igemm_h2.hip):
// results are bit-identical to the other tile variants.  What changes is the schedule.  The 128x128
// kernel drains its LDS-DMA queue (vmcnt(0)) and meets a barrier once per k-tile, and each wave stalls
// four times per k-tile on a full ds_read round trip; PMC showed the matrix pipe 55 % busy, and the
// ablations put a quarter of the time on operand staging.  Here
//
//   * one workgroup = 8 waves = 2 groups (wr = 0, 1) x 4 (wc); wave tile 128 x 64 (4 x 2 MFMA tiles of
//     32x32); LDS = 2 k-tile buffers x (A 256 rows + B 256 rows) x 128 B = 128 KB -> one workgroup per CU,
//     two waves per SIMD, one from each group (waves 0-3 / 4-7 land on distinct SIMDs);
//   * a k-tile (one 32-channel slice of one tap) is computed in FOUR phases, one 64x32 quadrant of the
//     wave tile each (Q00, Q01, Q11, Q10: only the operand that changes is re-read from LDS);
//     a phase is  { ds_read fragments | issue 2 LDS-DMA loads | s_waitcnt vmcnt(6) } barrier
//     { 12 MFMA } barrier;
//   * group 1 runs ONE BARRIER LATE: while one wave of a SIMD issues its 12 MFMAs (384 cycles) the
//     other one does its ds_reads and DMA issue, then they swap - the matrix pipe always has a wave in
//     its MFMA phase and the loads are spread evenly over time instead of arriving in bursts;
//   * the DMA queue is never drained in the steady state: every phase stages one 16 KB "unit" (2 loads
//     per thread) that is read FOUR phases later, and waits only until at most three units are in flight.
//
// Hazard bookkeeping (phase numbers j = 4 t + ph; interval = time between two consecutive barriers;
// group 0 does the load part of phase j in interval 2j and its MFMA part in 2j+1, group 1 one later):
//   units of k-tile t, buffer t & 1:   A0 = A rows {0..63, 128..191}   read in phase (t,0)
//                                      B0 = B rows {64 wc + 0..31}      read in phase (t,0), KEPT IN REGISTERS for (t,3)
//                                      B1 = B rows {64 wc + 32..63}     read in phase (t,1)
//                                      A1 = A rows {64..127, 192..255}  read in phase (t,2)
//   staged:  A0(t+1) in (t,0), B1(t+1) in (t,1), A1(t+1) in (t,2)  -> other buffer, last read 4 phases ago
//            B0(t+2) in (t,3)  -> the buffer being computed, whose B0 was last read 3 phases ago
//   RAW: a unit staged in phase s is retired by every wave's vmcnt(6) of phase s+3 (in-order return; 2 loads
//        per phase), which sits BEFORE that phase's first barrier; the earliest reader (group 0, phase s+4,
//        interval 2s+8) has passed the barrier that closes interval 2s+7, in which group 1 did that wait.
//   WAR: the last reads of a unit are complete (lgkmcnt(0) after the first barrier of their phase) by
//        interval 2d+2; the earliest restaging DMA is issued in interval 2(d+3).
//   The last three k-tiles use vmcnt(0) (fewer than three units follow them).
// The barriers are inline asm with a memory clobber so that no LDS read is hoisted across them;
// sched_barrier(0) keeps each MFMA cluster inside its interval.
#include <stdlib.h>

#include "dp_tune.h"
#include "igemm_h2.h"
#include "igemm_pp_common.h"

namespace {


constexpr int NT = 512;
constexpr int NXCD = 8;
constexpr int ROWB = 128;               // bytes per LDS row: 32 channels as (hi, lo) fp16 octets

// MODE (timing experiments, DP_H2_PP_MODE): bit 0 = no s_setprio; bit 1 = no operand traffic after k-tile 0 (WRONG
// RESULTS); bit 2 = no barriers in the k-loop (WRONG RESULTS); bit 3 = no ds_reads / bit 4 = no DMA after k-tile 0 / bit 5 = no vmcnt waits (WRONG RESULTS)
// Tile shapes: <256,256> (waves 2 (M) x 4 (N), 128 KB LDS) and <512,128> for layers with 128 output channels
// (waves 4 (M) x 2 (N), 160 KB LDS = all of it).  The wave tile is 128 x 64 and the phase schedule identical in
// both; only the wave -> tile mapping and the number of 64-row DMA pieces per unit differ (NPA, NPB).
// A16: the activation operand is plain fp16 ("h1", 2 bytes per element): A rows are 64 bytes in LDS (4 slots, XOR key
// (row >> 2) & 3), an A unit is half as many DMA instructions (one instruction stages 128 rows), only a_hi fragments
// exist and PASSES is 2 (a_hi*w_lo + a_hi*w_hi) or 1.  B (weights, hi|lo) and the whole phase schedule are unchanged;
// LDS drops to 96 KB.
// SCHED 1 (A16 only): TWO phases per k-tile instead of four - half the barriers and role swaps, 16 (PASSES 2) MFMAs
// per phase.  Phase 0 reads {A0, A1, B0} (256x256: quadrants Q00, Q10) or {A0, B0, B1} (512x128: Q00, Q01), phase 1
// the remaining unit and computes the other two quadrants from fragments that are still in registers.  Every
// phase stages, into the OTHER buffer, exactly the units the same phase of the next k-tile reads:
//   staged in phase j  ->  retired by every wave's counted vmcnt in phase j+1 (before its first barrier; only that
//   phase's own loads may still be in flight)  ->  read in phase j+2.
//   RAW: the earliest reader (group 0, phase j+2, interval 2j+4) has passed the barrier closing interval 2j+3, in
//        which group 1 did its phase-(j+1) wait.   WAR: a unit is re-staged two phases after its last read.
// W16 (A16, PASSES 1, SCHED 1): plain fp16 weights as well - B rows are 64 bytes, a B unit of the 256-wide tile is ONE DMA
// instruction per thread (128 rows), the whole B tile of the 128-wide tile is one (staged in phase 0); LDS 64 KB.
template <int BM, int BN, int MODE, int PASSES, bool A16, int SCHED, bool W16>
__global__ __launch_bounds__(NT) void conv_igemm_h2_pp(ConvH2Args p) {
    static_assert(SCHED == 0 || A16, "the two-phase schedule keeps both A halves in registers: fp16 operand only");
    static_assert(!W16 || (A16 && PASSES == 1 && SCHED == 1), "fp16 weights: one pass, fp16 activations, two-phase schedule");
    static_assert((BM == 256 && BN == 256) || (BM == 512 && BN == 128), "8 waves of 128 x 64");
    static_assert(A16 ? (PASSES == 2 || PASSES == 1) : (PASSES == 3 || PASSES == 12), "operand format / passes");
    constexpr int ESZ = A16 ? 2 : 4;                       // bytes per activation element
    constexpr int AROWB = 32 * ESZ;                        // bytes per LDS row of the A tile
    constexpr int NPA = A16 ? BM / 256 : BM / 128;         // DMA pieces (one instruction per thread) per A unit
    constexpr int BROWB = W16 ? 64 : ROWB;                 // bytes per LDS row of the B tile
    constexpr bool BWHOLE = W16 && BN == 128;              // the one DMA piece of a unit IS the whole B tile (both units)
    constexpr int NPB = W16 ? 1 : BN / 128;                // DMA pieces per B unit (h2: 64-row pieces; fp16: one 128-row piece)
    constexpr int WSZ = W16 ? 2 : 4;                       // bytes per weight element
    constexpr int TILE_A = BM * AROWB, TILE_B = BN * BROWB; // operand tiles of one k-tile
    constexpr int BUF = TILE_A + TILE_B;                   // A tile, then B tile
    constexpr int CNT_A = 2 * NPA + NPB, CNT_B = 2 * NPB + NPA;   // loads of three consecutive phases ending in an A / a B phase
    __shared__ __attribute__((aligned(1024))) char smem[2 * BUF];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;                                   // waves 0-3 / 4-7 sit on distinct SIMDs
    const int wr = BM == 256 ? wave >> 2 : wave & 3;             // 128-row block of the tile
    const int wc = BM == 256 ? wave & 3 : wave >> 2;             // 64-column block of the tile
    int tile;
    {   // XCD-aware bijective remap (speed only)
        const int b = blockIdx.x, x = b % NXCD, q = p.tiles / NXCD, r = p.tiles % NXCD;
        tile = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + b / NXCD;
    }
    const int tile_n = tile % p.tiles_n, tile_m = tile / p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int HW = p.H * p.W, Wp = p.W + 2, taps = p.KS * p.KS;
    const int nt = p.K / 32;
    // (A start-up stagger that de-phased the CUs' epilogues - the first tile of every CU starting (b / 8) % 8 eighths of a tile late -
    //  measured +1.5 ... +9 % on isolated back-to-back launches of one shape and +0.5 % = noise on the whole purification; it was off by
    //  default since round 2 and is removed in round 4.)
    // ---- staging geometry: a B unit is 128 rows = 2 pieces of 64 rows; lane -> row u of the piece, physical slot tid & 7.
    // An A unit is the rows {blk*128 + unit*64 + 0..63} of every 128-row block blk of the tile.  h2 operand: one piece per
    // block (64 rows x 128 B, lane -> row u, slot tid & 7); h1 operand: one piece per TWO blocks (128 rows x 64 B, lane ->
    // row ua = tid >> 2 of the piece, slot tid & 3).
    const int u = tid >> 3;
    const int ls = (tid & 7) ^ ((u >> 1) & 7);     // logical slot fetched (XOR swizzle applied on the source side)
    const int ua = A16 ? tid >> 2 : u;
    const int lsa = A16 ? ((tid & 3) ^ ((ua >> 2) & 3)) : ls;
    // Sources are addressed as (scalar 64-bit base) + (32-bit lane offset): the lane offsets are loop
    // invariant and the per-k-tile part (tap shift, channel slice, weight column block) lives in the
    // scalar base, so a DMA issue costs no vector ALU work and reads one address VGPR per lane.
    // Activation lane offsets are relative to the centre pixel of the tile's first row (always < 2^31:
    // a tile spans 256 consecutive output pixels).
    unsigned aoff[2][NPA];                         // [unit][piece]
    long long aorg;                                // centre pixel of row m0, in bytes from p.x
    {
        const int b = m0 / HW, rem = m0 - b * HW;
        const int oy = rem / p.W, ox = rem - oy * p.W;
        aorg = ((long long)(b * (p.H + 2) + oy + 1) * Wp + ox + 1) * p.C * ESZ;
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int i = 0; i < NPA; ++i) {
            const int m = A16 ? m0 + (2 * i + (ua >> 6)) * 128 + a * 64 + (ua & 63) : m0 + i * 128 + a * 64 + u;
            const int b = m / HW, rem = m - b * HW;
            const int oy = rem / p.W, ox = rem - oy * p.W;
            aoff[a][i] = (unsigned)(((long long)(b * (p.H + 2) + oy + 1) * Wp + ox + 1) * p.C * ESZ - aorg) + lsa * 16;
        }
    unsigned boff[2][NPB];                         // [unit][piece]: weight row n0 + piece*128 + (u>>5)*64 + unit*32 + (u&31)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int i = 0; i < NPB; ++i) {
            if constexpr (W16) {   // 128-row piece, lane -> row ua of the piece (tid >> 2), 4 slots per 64-byte row
                const int row = BWHOLE ? ua : (ua >> 5) * 64 + b * 32 + (ua & 31);
                // block layout of the fp16 panels (ops.order_conv_weight_w16); slot key (row >> 2) & 3 == (ua >> 2) & 3
                boff[b][i] = (unsigned)(row >> 5) * (unsigned)(p.K * 64) + (row & 31) * 16 + lsa * 512;
            } else {
                boff[b][i] = (unsigned)(i * 128 + (u >> 5) * 64 + b * 32 + (u & 31)) * (unsigned)(p.K * 4) + ls * 16;
            }
        }
    const char* const abase = p.x + pp_uniform(aorg);              // uniform
    const char* const bbase = p.w + pp_uniform(W16 ? (long long)(n0 >> 5) * p.K * 64 : (long long)n0 * p.K * WSZ);   // uniform
    const int u0 = wave * 8;                       // first row of this wave inside a 64-row piece (wave-uniform)
    const int ua0 = wave * 16;                     // ... inside a 128-row h1 A piece
    // LDS rows of the A tile are the tile's rows; a wave's DMA instruction fills 1 KB = 8 (h2) / 16 (h1) consecutive rows
    const int adst = A16 ? ((ua0 >> 6) * 128 + (ua0 & 63)) * AROWB : u0 * AROWB;   // + (piece*{256|128} + unit*64) * AROWB
    const int bdst = W16 ? TILE_A + (BWHOLE ? ua0 : (ua0 >> 5) * 64 + (ua0 & 31)) * BROWB   // + unit*32 * BROWB (256-wide tile)
                         : TILE_A + ((u0 >> 5) * 64 + (u0 & 31)) * ROWB;  // + (piece*128 + unit*32) * ROWB

    auto tap_off = [&](int c32, int tap) -> long long {
        const int ky = tap / p.KS, kx = tap - ky * p.KS;
        return ((long long)(ky - p.pad) * Wp + (kx - p.pad)) * p.C * ESZ + (long long)c32 * (32 * ESZ);
    };
    auto stage_a = [&](char* buf, int a, long long off) {
        const char* sb = abase + pp_uniform(off);
#pragma unroll
        for (int i = 0; i < NPA; ++i) pp_glds(aoff[a][i], sb, buf + adst + (i * (A16 ? 256 : 128) + a * 64) * AROWB);
    };
    auto stage_b = [&](char* buf, int b, long long off) {
        const char* sb = bbase + pp_uniform(off);
#pragma unroll
        for (int i = 0; i < NPB; ++i) pp_glds(boff[b][i], sb, buf + bdst + (BWHOLE ? 0 : (i * 128 + b * 32) * BROWB));
    };

    // ---- fragment addressing: lane -> row lr of a 32-row MFMA tile, k-half lk; slot (s*4 + lk*2 + h) ^ key
    const int lr = lane & 31, lk = lane >> 5, key = (lr >> 1) & 7;
    const int arow = (wr * 128 + lr) * AROWB;                // + (sub*64 + i*32) * AROWB
    const int brow = TILE_A + (wc * 64 + lr) * BROWB;        // + (sub*32) * BROWB
    int soff[2][2];                                          // [s][h] byte offset of the fragment inside its (128-byte) row
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int h = 0; h < 2; ++h) soff[s][h] = ((s * 4 + lk * 2 + h) ^ key) << 4;
    int soffa[2];                                            // h1 A rows (64 bytes): slot s*2 + lk, key (row >> 2) & 3
#pragma unroll
    for (int s = 0; s < 2; ++s) soffa[s] = ((s * 2 + lk) ^ ((lr >> 2) & 3)) << 4;

    half8 fah[2][2], fal[2][2];     // A fragments of the current 64-row half: [m-tile i][k16 step s]
    half8 fb0h[2], fb0l[2];         // B fragments, column sub-block 0 (kept from phase 0 to phase 3)
    half8 fb1h[2], fb1l[2];         // B fragments, column sub-block 1
    auto read_a = [&](const char* buf, int sub) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const char* r = buf + arow + (sub * 64 + i * 32) * AROWB;
                if constexpr (A16) {
                    fah[i][s] = *reinterpret_cast<const half8*>(r + soffa[s]);
                } else {
                    fah[i][s] = *reinterpret_cast<const half8*>(r + soff[s][0]);
                    fal[i][s] = *reinterpret_cast<const half8*>(r + soff[s][1]);
                }
            }
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // quadrant (asub, bsub): 2 m-tiles x 1 n-tile x 2 k16 steps x 3 passes, per-element order
    // (a_lo w_hi, a_hi w_lo, a_hi w_hi) per k16 step - identical to conv_igemm_h2
#define PP_MFMA(ASUB, BSUB, BH, BL)                                                                              \
    do {                                                                                                         \
        _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                                          \
            if constexpr (PASSES == 3 || PASSES == 12) {                                                         \
                _Pragma("unroll") for (int i = 0; i < 2; ++i) acc[ASUB * 2 + i][BSUB] =                          \
                    __builtin_amdgcn_mfma_f32_32x32x16_f16(fal[i][s], BH[s], acc[ASUB * 2 + i][BSUB], 0, 0, 0); \
            }                                                                                                    \
            if constexpr (PASSES == 3 || PASSES == 2) {                                                          \
                _Pragma("unroll") for (int i = 0; i < 2; ++i) acc[ASUB * 2 + i][BSUB] =                          \
                    __builtin_amdgcn_mfma_f32_32x32x16_f16(fah[i][s], BL[s], acc[ASUB * 2 + i][BSUB], 0, 0, 0); \
            }                                                                                                    \
            _Pragma("unroll") for (int i = 0; i < 2; ++i) acc[ASUB * 2 + i][BSUB] =                              \
                __builtin_amdgcn_mfma_f32_32x32x16_f16(fah[i][s], BH[s], acc[ASUB * 2 + i][BSUB], 0, 0, 0);     \
        }                                                                                                        \
    } while (0)
    // end of a phase's load part .. MFMA part .. end of phase
#define PP_SYNC_THEN_MFMA(COUNTED, CNT, ASUB, BSUB, BH, BL)                             \
    do {                                                                           \
        if constexpr (!(MODE & 32)) {                                              \
            if (COUNTED) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CNT) : "memory"); \
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  \
        }                                                                          \
        if constexpr (!(MODE & 4)) PP_BARRIER();                                   \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                         \
        __builtin_amdgcn_sched_barrier(0);                                         \
        if constexpr (!(MODE & 1)) __builtin_amdgcn_s_setprio(1);                  \
        PP_MFMA(ASUB, BSUB, BH, BL);                                               \
        if constexpr (!(MODE & 1)) __builtin_amdgcn_s_setprio(0);                  \
        __builtin_amdgcn_sched_barrier(0);                                         \
        if constexpr (!(MODE & 4)) PP_BARRIER();                                   \
    } while (0)

    if constexpr (SCHED == 1) {
        constexpr bool RF = BM == 256;                 // rows first: phase 0 = {A0, A1, B0}; else {A0, B0, B1}
        constexpr int N0 = RF ? 2 * NPA + NPB : NPA + (BWHOLE ? 1 : 2 * NPB), N1 = RF ? NPB : NPA;    // DMA loads per thread per phase
        half8 fa2[2][2][2];                            // [half][m-tile i][k16 step s]
        auto read_a2 = [&](const char* buf, int sub) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int s = 0; s < 2; ++s)
                    fa2[sub][i][s] = *reinterpret_cast<const half8*>(buf + arow + (sub * 64 + i * 32) * AROWB + soffa[s]);
        };
        auto read_b0 = [&](const char* buf) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                if constexpr (W16) {
                    fb0h[s] = *reinterpret_cast<const half8*>(buf + brow + soffa[s]);
                } else {
                    fb0h[s] = *reinterpret_cast<const half8*>(buf + brow + soff[s][0]);
                    fb0l[s] = *reinterpret_cast<const half8*>(buf + brow + soff[s][1]);
                }
            }
        };
        auto read_b1 = [&](const char* buf) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                if constexpr (W16) {
                    fb1h[s] = *reinterpret_cast<const half8*>(buf + brow + 32 * BROWB + soffa[s]);
                } else {
                    fb1h[s] = *reinterpret_cast<const half8*>(buf + brow + 32 * BROWB + soff[s][0]);
                    fb1l[s] = *reinterpret_cast<const half8*>(buf + brow + 32 * BROWB + soff[s][1]);
                }
            }
        };
#define PP_MFMA2(ASUB, BSUB, BH, BL)                                                                                   \
    do {                                                                                                               \
        _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                                                \
            if constexpr (PASSES == 2) {                                                                               \
                _Pragma("unroll") for (int i = 0; i < 2; ++i) acc[ASUB * 2 + i][BSUB] =                                \
                    __builtin_amdgcn_mfma_f32_32x32x16_f16(fa2[ASUB][i][s], BL[s], acc[ASUB * 2 + i][BSUB], 0, 0, 0);  \
            }                                                                                                          \
            _Pragma("unroll") for (int i = 0; i < 2; ++i) acc[ASUB * 2 + i][BSUB] =                                    \
                __builtin_amdgcn_mfma_f32_32x32x16_f16(fa2[ASUB][i][s], BH[s], acc[ASUB * 2 + i][BSUB], 0, 0, 0);      \
        }                                                                                                              \
    } while (0)
#define PP_SYNC2(COUNTED, CNT)                                                      \
    do {                                                                            \
        if constexpr (!(MODE & 32)) {                                               \
            if (COUNTED) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CNT) : "memory"); \
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   \
        }                                                                           \
        PP_BARRIER();                                                               \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                          \
        __builtin_amdgcn_sched_barrier(0);                                          \
    } while (0)
#define PP_END2()                                   \
    do {                                            \
        __builtin_amdgcn_sched_barrier(0);          \
        PP_BARRIER();                               \
    } while (0)
        // prologue: all of k-tile 0, drained
        stage_a(smem, 0, tap_off(0, 0));
        stage_b(smem, 0, 0);
        if constexpr (!BWHOLE) stage_b(smem, 1, 0);
        stage_a(smem, 1, tap_off(0, 0));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PP_BARRIER();
        if (grp == 1) PP_BARRIER();         // group 1 runs one interval behind group 0
        int c1 = 0, tap1 = 1;               // (slice, tap) of k-tile t+1
        if (tap1 == taps) { tap1 = 0; c1 = 1; }
        for (int t = 0; t < nt; ++t) {
            const char* cur = smem + (t & 1) * BUF;
            char* nxt = smem + ((t + 1) & 1) * BUF;
            const bool traffic = !(MODE & (2 | 8)) || t == 0;
            const bool more1 = (!(MODE & (2 | 16)) || t == 0) && t + 1 < nt;
            const long long offa = tap_off(c1, tap1);
            const long long offb1 = (long long)(t + 1) * (W16 ? 2048 : 32 * WSZ);      // fp16 panels: 4 groups of 512 bytes per k-tile
            // phase 0
            if (traffic) {
                read_a2(cur, 0);
                read_b0(cur);
                if constexpr (RF) read_a2(cur, 1);
                else read_b1(cur);
            }
            if (more1) {
                stage_a(nxt, 0, offa);
                stage_b(nxt, 0, offb1);
                if constexpr (RF) stage_a(nxt, 1, offa);
                else if constexpr (!BWHOLE) stage_b(nxt, 1, offb1);
            }
            PP_SYNC2(more1, N0);
            PP_MFMA2(0, 0, fb0h, fb0l);
            if constexpr (RF) PP_MFMA2(1, 0, fb0h, fb0l);
            else PP_MFMA2(0, 1, fb1h, fb1l);
            PP_END2();
            // phase 1
            if (traffic) {
                if constexpr (RF) read_b1(cur);
                else read_a2(cur, 1);
            }
            if (more1) {
                if constexpr (RF) stage_b(nxt, 1, offb1);
                else stage_a(nxt, 1, offa);
            }
            PP_SYNC2(more1, N1);
            PP_MFMA2(1, 1, fb1h, fb1l);
            if constexpr (RF) PP_MFMA2(0, 1, fb1h, fb1l);
            else PP_MFMA2(1, 0, fb0h, fb0l);
            PP_END2();
            if (++tap1 == taps) { tap1 = 0; ++c1; }
        }
        if (grp == 0) PP_BARRIER();
#undef PP_MFMA2
#undef PP_SYNC2
#undef PP_END2
    } else {
    // ---- prologue: all of k-tile 0 and B0 of k-tile 1, drained
    stage_a(smem, 0, tap_off(0, 0));
    stage_b(smem, 0, 0);
    stage_b(smem, 1, 0);
    stage_a(smem, 1, tap_off(0, 0));
    if (nt > 1) stage_b(smem + BUF, 0, 128);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PP_BARRIER();
    if (grp == 1) PP_BARRIER();         // group 1 runs one interval behind group 0

    int c1 = 0, tap1 = 1;               // (slice, tap) of k-tile t+1
    if (tap1 == taps) { tap1 = 0; c1 = 1; }
    for (int t = 0; t < nt; ++t) {
        const char* cur = smem + (t & 1) * BUF;
        char* nxt = smem + ((t + 1) & 1) * BUF;
        const bool traffic = !(MODE & (2 | 8)) || t == 0;                   // ds_reads
        const bool dma = !(MODE & (2 | 16)) || t == 0;
        const bool more1 = dma && t + 1 < nt, more2 = dma && t + 2 < nt;
        const long long offa = tap_off(c1, tap1);
        const long long offb1 = (long long)(t + 1) * 128;

        // phase 0: Q00
        if (traffic) read_a(cur, 0);
#pragma unroll
        for (int s = 0; s < 2 && traffic; ++s) {
            fb0h[s] = *reinterpret_cast<const half8*>(cur + brow + soff[s][0]);
            fb0l[s] = *reinterpret_cast<const half8*>(cur + brow + soff[s][1]);
        }
        if (more1 && !(MODE & 64)) stage_a(nxt, 0, offa);
        PP_SYNC_THEN_MFMA(more1, CNT_A, 0, 0, fb0h, fb0l);

        // phase 1: Q01
#pragma unroll
        for (int s = 0; s < 2 && traffic; ++s) {
            fb1h[s] = *reinterpret_cast<const half8*>(cur + brow + 32 * ROWB + soff[s][0]);
            fb1l[s] = *reinterpret_cast<const half8*>(cur + brow + 32 * ROWB + soff[s][1]);
        }
        if (more1 && !(MODE & 128)) stage_b(nxt, 1, offb1);
        PP_SYNC_THEN_MFMA(more1, CNT_B, 0, 1, fb1h, fb1l);

        // phase 2: Q11
        if (traffic) read_a(cur, 1);
        if (more1 && !(MODE & 64)) stage_a(nxt, 1, offa);
        PP_SYNC_THEN_MFMA(more1, CNT_A, 1, 1, fb1h, fb1l);

        // phase 3: Q10 (B sub-block 0 still in registers); B0 of k-tile t+2 goes into the buffer being computed
        if (more2 && !(MODE & 128)) stage_b(const_cast<char*>(cur), 0, offb1 + 128);
        PP_SYNC_THEN_MFMA(more2, CNT_B, 1, 0, fb0h, fb0l);

        if (++tap1 == taps) { tap1 = 0; ++c1; }
    }
    if (grp == 0) PP_BARRIER();         // re-align the two groups (barrier counts must match)
    }
#undef PP_SYNC_THEN_MFMA
#undef PP_MFMA

    // ---- epilogue: bias / temb / residual / scale, store, per-column (sum, sumsq) of every 64 output rows.
    // 32-row sub-sums (16 values per lane in r order, then the partner half-wave) paired even+odd: the
    // tile-shape-independent order of igemm.hip.  Both sub-sums of a record live in this wave: no LDS.
    // The residual reads of one column block (4 m-tiles x 16 rows) are all issued before the first use: 64
    // loads in flight per lane (the fragment registers are free now) instead of one round trip per element.
    if constexpr (MODE & 256) {      // timing ablation: no epilogue at all (accumulators kept alive)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) asm volatile("" ::"v"(acc[i][j]));
        return;
    }
    pp_epilogue<BM>(p, acc, m0, n0, tile_m, wr, wc, lr, lk, HW);
}

}  // namespace

void dp_launch_conv_h2_pp(ConvH2Args& p, hipStream_t s, int bn) {
    const int bm = bn == 256 ? 256 : 512;
    p.tiles_n = p.N / bn;
    p.tiles = (p.M / bm) * p.tiles_n;
#ifdef DP_ABLATE   // timing ablations (WRONG RESULTS): only in libdiffpure_hip_ablate.so (tests/probes/build_ablate.py)
    const char* e = getenv("DP_H2_PP_MODE");
    const int mode = e ? atoi(e) : 0;
#endif
#define PP_LAUNCH1(BM_, BN_, M_, P_, A_, S_, W_) \
    hipLaunchKernelGGL((conv_igemm_h2_pp<BM_, BN_, M_, P_, A_, S_, W_>), dim3((unsigned)p.tiles), dim3(NT), 0, s, p)
    // fp16-operand kernels run the two-phase schedule (SCHED 1; the four-phase form measured 495-635 vs 621-661 TFLOP/s and is no
    // longer instantiated); (a_fmt 1, passes 1) exists with fp16 weights only (dp_conv2d_nhwc_h2 checks)
#define PP_LAUNCH(BM_, BN_, M_)                                          \
    do {                                                                 \
        if (p.wfmt == 1) PP_LAUNCH1(BM_, BN_, M_, 1, true, 1, true);     \
        else if (p.afmt == 1) PP_LAUNCH1(BM_, BN_, M_, 2, true, 1, false);   \
        else if (M_ == 0 && p.passes == 12) PP_LAUNCH1(BM_, BN_, 0, 12, false, 0, false); \
        else PP_LAUNCH1(BM_, BN_, M_, 3, false, 0, false);                      \
    } while (0)

    if (bn == 128) {
        PP_LAUNCH(512, 128, 0);
        return;
    }
#ifdef DP_ABLATE
    switch (mode) {     // timing experiments (see MODE)
        case 1: PP_LAUNCH(256, 256, 1); return;
        case 2: PP_LAUNCH(256, 256, 2); return;
        case 6: PP_LAUNCH(256, 256, 6); return;
        case 8: PP_LAUNCH(256, 256, 8); return;
        case 16: PP_LAUNCH(256, 256, 16); return;
        case 32: PP_LAUNCH(256, 256, 32); return;
        case 256: PP_LAUNCH(256, 256, 256); return;
        default: break;
    }
#endif
    PP_LAUNCH(256, 256, 0);
#undef PP_LAUNCH
#undef PP_LAUNCH1
}
