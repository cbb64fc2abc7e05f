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
// SHAPE 1 (round 6): the same kernel on 512 (pixels) x 128 (channels) tiles for layers with 128 output channels (N % 256 != 0: the 32x32
// level of NCSN++, 36 % of the convolution time of a CIFAR-10 UNet call, until then on the one-wave-per-SIMD kernel's 512 x 128 tiles at
// 659 TFLOP/s).  The eight waves are stacked along the pixels - wave w owns rows [64 w, 64 w + 64) and ALL 128 columns - so the wave
// tile is the 64 x 128 of the square form: the same fragment reads (2 A + 4 B per eight MFMAs), the same accumulators, the same epilogue
// (igemm_sw_common.h, NQ = 1), the same bits.  What changes is the staging: an activation k-tile is 512 rows (32 KB, four pieces per wave),
// a weight k-tile 128 rows (8 KB, one piece per wave); the older wave of a SIMD issues its own 1 + 4 pieces inside segment A and its partner's
// 1 + 4 behind its vmcnt wait.  40 KB instead of 32 KB of operands per k-tile for the same 128 MFMAs.
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
constexpr int BDEPTH = 3;

template <int N>
__device__ __forceinline__ void dw_wait_vm() {
    static_assert(N == 10 || N == 8 || N == 5 || N == 4 || N == 0, "add the immediate");
    if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// (Round 4 measured a start-up de-phasing of the CUs - the first workgroup of CU slot c starting c / 32 of a tile late, so that the
//  256 epilogues do not hit HBM together: 0 % on 256^2 256->256 at B=64, -2...-4 % on the shorter launches, -1...-2 % on the
//  purification (profiles/r04/dephase_ab.log).  The epilogue's cost is not a lockstep burst.  Removed.)
template <int Q>
struct dw_const { static constexpr int value = Q; };

// MODE (timing ablations, DP_ABLATE builds only; WRONG RESULTS): 1 = no DMA in the steady state, 2 = no barrier / vmcnt wait,
// 4 = no ds_reads, 8 = no epilogue stores, 16 = no activation DMA, 32 = no weight DMA
// 64 (correct results, ~10 % slower): SEGMENT TIMELINE - every wave stamps s_memtime at the five segment boundaries of a steady
// k-tile (A: 8 MFMA | 6 ds_read | 4 DMA pieces; B: 4 MFMA; W: the counted vmcnt wait; S: the barrier; C: 6 ds_read | 4 MFMA), sums
// the segment durations over its k-loop and writes [A, B, W, S, C, epilogue, k-tiles, total] (shader cycles) to p.ws per
// (workgroup, wave): tests/probes/dw8_timeline.py.
// 256 (correct results, full speed): TILE LIFETIME - four stamps per wave (entry, k-loop start, k-loop end, exit) and the CU the workgroup
// ran on (HW_ID / XCC_ID): prologue, k-loop and epilogue cycles per tile and the gap between consecutive workgroups of one CU
// (tests/probes/dw8_lifetime.py).
// 512 (correct results): BUFFER-FORM STAGING - every LDS-DMA piece is a `buffer_load_dwordx4 ... lds` with the tensor base in a descriptor,
// the per-k-tile advance (tap / slice offset, weight k-tile) in the instruction's SCALAR offset and ONE constant 32-bit lane offset per piece,
// instead of a 64-bit per-lane pointer advanced by a vector add per piece (8 v_lshl_add_u64 per k-tile in the staging wave's issue stream,
// which the round-4 timeline names as what bounds a SIMD's pair of waves).  A/B against mode 0 in tests/probes/dw8_timeline.py.
// one LDS-DMA piece in buffer form (MODE 512): 16 bytes per lane from (descriptor base + scalar offset + lane offset) to lds + 16 lane
__device__ __forceinline__ void dw_buf_lds16(sw_rsrc r, char* lds, unsigned voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 0);
}

template <int MODE, int SHAPE = 0>
__global__ __launch_bounds__(512, 1) void conv_igemm_dw(ConvH2Args p) {
    constexpr int ADEPTH = 3, DA = ADEPTH - 1;  // ring stages / prefetch distance of both operands
    constexpr int BMT = SHAPE ? 512 : 256;      // tile rows: every wave owns BMT / 8 of them (2 | 4 pieces of 16 rows) ...
    constexpr int BNT = SHAPE ? 128 : 256;      // ... and BNT / 8 weight rows (2 | 1 pieces)
    constexpr int ATILE = BMT * 64;             // activation tile of one k-tile: 64 bytes (32 fp16) per row
    constexpr int BTILE = BNT * 64;             // weight tile of one k-tile
    constexpr int NAO = BMT / 128, NBO = BNT / 128;   // a wave's OWN activation / weight pieces per k-tile
    constexpr int NPB = NBO;
    constexpr int BBASE = ADEPTH * ATILE;
    static_assert(SHAPE == 0 || (MODE & (128 | 512)) == 0, "the 512 x 128 form exists in the asymmetric pointer-form staging only");
    // the operand rings (96 KB); after the k-loop the same memory is the landing zone of the fp16 residual (igemm_sw_common.h, 136 KB)
    constexpr int RINGS = ADEPTH * ATILE + BDEPTH * BTILE;
    __shared__ __attribute__((aligned(1024))) char smem[RINGS > SW_EPI_LDS ? RINGS : SW_EPI_LDS];

    unsigned life0 = 0;
    if constexpr (MODE & 256) life0 = (unsigned)__builtin_amdgcn_s_memtime();
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = SHAPE ? wave : wave >> 1, wc = SHAPE ? 0 : wave & 1;      // SHAPE 1: the waves are stacked along the pixels
    int tile;
    {   // XCD-aware bijective remap (speed only)
        const int b = blockIdx.x, x = b % NXCD, q = p.tiles / NXCD, r = p.tiles % NXCD;
        tile = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + b / NXCD;
    }
    const int tile_n = tile % p.tiles_n, tile_m = tile / p.tiles_n;
    const int m0 = tile_m * BMT, n0 = tile_n * BNT;
    const int HW = p.H * p.W, Wp = p.W + 2, taps = p.KS * p.KS;
    const int nt = p.K / 32;

    // ---- staging: wave w fills rows [32 w, 32 w + 32) of the A tile and of the B tile, 16 rows per DMA instruction; lane -> row
    // (lane >> 2) of the piece, physical slot lane & 3, logical slot XOR-ed with the row key
    const int lrow = lane >> 2;
    const int ls = (lane & 3) ^ ((lrow >> 2) & 3);
    // ASYMMETRIC STAGING (round 4; MODE & 128 = the symmetric form of round 3, for A/B runs).  The segment timeline
    // (tests/probes/dw8_timeline.py, profiles/r04/dw8_timeline*.log) showed that with every wave staging its own rows the OLDER wave
    // of a SIMD (waves 0-3: the issue arbitration favours it) finishes a k-tile ~440 cycles before its partner and idles at the
    // barrier, while every LDS-DMA piece the YOUNGER wave issues costs it ~145 cycles of its segment A (870 cycles for 8 MFMAs
    // against 290 without any).  So the older wave stages the rows of BOTH waves of its SIMD - its own four pieces interleaved into
    // segment A as before, the partner's four after its vmcnt wait, in what was barrier idle time - and the younger wave issues
    // no LDS-DMA at all: 1 320 -> 1 270 cycles per k-tile, +2.1 ... +3.7 % TFLOP/s on every shape measured, identical bits.
    // (Measured and not kept: the younger wave keeping 1 or 2 of its pieces: -2.5 / -0.5 %; every wave issuing its own pieces late:
    // -0.7 ... -1.8 %, the DMA latency is no longer covered; s_setprio 1 on either wave on top: -2.2 / +0.0 %.)
    // Piece index it: [0, NAO) / [0, NBO) = own rows, the rest = the partner's (wave + 4).
    constexpr bool ASYM = (MODE & 128) == 0;
    constexpr bool BUF = (MODE & 512) != 0;
    constexpr int NPIECE = ASYM ? 2 * NAO : NAO;     // activation pieces a staging wave issues per k-tile (4 | 8; symmetric form 2)
    constexpr int NPIECEB = ASYM ? 2 * NBO : NBO;    // weight pieces (4 | 2; symmetric form 2)
    const char* actr[NPIECE];                   // centre pixel of the lane's A row (segments: the lane's pixel), + slot
    const char* bptr[NPIECEB];
    auto rows_of_piece = [&](int it) { return (wave + (it / NAO) * 4) * (BMT / 8) + (it % NAO) * 16; };
    auto rows_of_pieceB = [&](int it) { return (wave + (it / NBO) * 4) * (BNT / 8) + (it % NBO) * 16; };
#pragma unroll
    for (int it = 0; it < NPIECE; ++it) {
        const int m = m0 + rows_of_piece(it) + lrow;
        const int b = m / HW, rem = m - b * HW;
        const int oy = rem / p.W, ox = rem - oy * p.W;
        actr[it] = p.x + ((size_t)(b * (p.H + 2) + oy + 1) * Wp + ox + 1) * p.C * 2 + ls * 16;
    }
#pragma unroll
    for (int it = 0; it < NPIECEB; ++it) {
        const int n = n0 + rows_of_pieceB(it) + lrow;               // block layout of the fp16 panels (ops.order_conv_weight_w16)
        bptr[it] = p.w + (size_t)(n >> 5) * p.K * 64 + (n & 31) * 16 + ls * 512;
    }
    // buffer-form staging (MODE 512): descriptors + constant lane offsets + scalar offsets
    unsigned voA[NPIECE], voB[NPIECE];
    sw_rsrc rsA, rsB;
    int soA_bias = 0, a_so = 0, soB = 0;
    if constexpr (BUF) {
        auto pix = [&](int m) {                 // pixel index of output row m inside the zero-bordered operand
            const int b = m / HW, rem = m - b * HW;
            const int oy = rem / p.W, ox = rem - oy * p.W;
            return (long long)(b * (p.H + 2) + oy + 1) * Wp + ox + 1;
        };
        const long long P0 = pix(m0);           // (wave-uniform) the tile's first pixel; pixel indices grow with m
        soA_bias = (Wp + 1) * p.C * 2;          // the most negative tap offset: the scalar offset stays >= 0
        rsA = sw_make_rsrc(p.x + (P0 - (Wp + 1)) * p.C * 2);
        rsB = sw_make_rsrc(p.w + (size_t)(n0 >> 5) * p.K * 64);
#pragma unroll
        for (int it = 0; it < NPIECE; ++it) {
            voA[it] = (unsigned)((pix(m0 + rows_of_piece(it) + lrow) - P0) * p.C * 2 + ls * 16);
            const int n = n0 + rows_of_pieceB(it % NPIECEB) + lrow;     // (buffer form: SHAPE 0 only, NPIECEB == NPIECE)
            voB[it] = (unsigned)(((n >> 5) - (n0 >> 5)) * p.K * 64 + (n & 31) * 16 + ls * 512);
        }
    }
    const bool older = !ASYM || wave < 4;       // (wave-uniform) the wave that stages
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
                for (int j = 0; j < NPIECE; ++j) actr[j] = sb + (size_t)(m0 + rows_of_piece(j) + lrow) * sc * 2 + ls * 16;
                if constexpr (BUF) {
                    rsA = sw_make_rsrc(sb + (size_t)m0 * sc * 2);
#pragma unroll
                    for (int j = 0; j < NPIECE; ++j) voA[j] = (unsigned)((rows_of_piece(j) + lrow) * sc * 2 + ls * 16);
                }
                seg_slices = sc / 32;
                cur_c = 0;
            }
            if (cur_seg == 0) {
                const int ky = p.KS == 3 ? (cur_tap * 11) >> 5 : 0, kx = cur_tap - ky * p.KS;     // tap / 3 for tap < 9, no division
                a_off = ((long long)(ky - p.pad) * Wp + (kx - p.pad)) * p.C * 2 + (long long)cur_c * 64;
                if constexpr (BUF) a_so = __builtin_amdgcn_readfirstlane(((ky - p.pad) * Wp + (kx - p.pad)) * p.C * 2 + cur_c * 64 + soA_bias);
            } else {
                a_off = (long long)cur_c * 64;
                if constexpr (BUF) a_so = __builtin_amdgcn_readfirstlane(cur_c * 64);
            }
        }
        if constexpr (BUF)
            dw_buf_lds16(rsA, smem + aoff + rows_of_piece(it) * 64, voA[it], a_so);
        else
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(actr[it] + a_off),
                                         (__attribute__((address_space(3))) void*)(smem + aoff + rows_of_piece(it) * 64), 16, 0, 0);
        if (it == NPIECE - 1 && (cur_seg != 0 || ++cur_tap == taps)) { cur_tap = 0; ++cur_c; }
    };
    auto pieceB = [&](int boff, int it) {
        if constexpr (BUF) {
            dw_buf_lds16(rsB, smem + boff + rows_of_pieceB(it) * 64, voB[it], soB);
            if (it == NPIECEB - 1) soB += 2048; // the pieces of a k-tile are issued in order 0 .. NPIECEB - 1
        } else {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)bptr[it],
                                         (__attribute__((address_space(3))) void*)(smem + boff + rows_of_pieceB(it) * 64), 16, 0, 0);
        bptr[it] += 2048;
        }
    };
    auto issueA = [&](int aoff) {
#pragma unroll
        for (int it = 0; it < NPIECE; ++it) pieceA(aoff, it);
    };
    auto issueB = [&](int boff) {
#pragma unroll
        for (int it = 0; it < NPIECEB; ++it) pieceB(boff, it);
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
    if (older) {
        issueB(br[0]);
        issueA(ar[0]);
        issueA(ar[1]);
        issueB(br[1]);
        dw_wait_vm<NPIECE + NPIECEB>();         // k-tile 0 landed; A(1), B(1) may fly
    }
    SW_BARRIER();
    read_frags(0, ar[0], br[0]);

    // steady state: k-tile t+2 exists
    unsigned tl[5] = {0u, 0u, 0u, 0u, 0u};      // MODE & 64: summed segment durations
    unsigned tl_n = 0, tl_s0 = 0, tl_begin = 0;
    auto stamp = []() { return (unsigned)__builtin_amdgcn_s_memtime(); };
    if constexpr (MODE & (64 | 256)) tl_begin = stamp();
    int t = 0;
    // ROLE (compile time: one copy of the loop per role behind a wave-uniform branch, so that every copy is straight-line code the
    // scheduler can pin): 0 = symmetric (four pieces in segment A), 1 = older wave (four own pieces in A, the partner's four after
    // the vmcnt wait), 2 = younger wave (no LDS-DMA)
    auto steady = [&](auto role_tag) __attribute__((always_inline)) {
        constexpr int ROLE = decltype(role_tag)::value;
        constexpr int NDMA = ROLE == 2 ? 0 : NBO + NAO;      // pieces interleaved into segment A: the wave's own (4 | 5 of the 6 slots)
    for (; t + DA < nt; ++t) {
        if constexpr (MODE & 64) {
            __builtin_amdgcn_sched_barrier(0);
            tl_s0 = stamp();
            __builtin_amdgcn_sched_barrier(0);
        }
        // first half: 8 MFMAs on fragment set 0 | the 6 reads of set 1 and the DMA pieces, one (read, piece) pair per MFMA shadow
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            if constexpr (!(MODE & 4)) {
                if (k < 2) readA(1, ar[0], k);
                else readB(1, br[0], k - 2);
            }
            if constexpr (NDMA != 0) {
                if (k < NBO) {
                    if constexpr (!(MODE & 33)) pieceB(br[2], k);
                } else if (k < NBO + NAO) {
                    if constexpr (!(MODE & 17)) pieceA(ar[DA], k - NBO);
                }
            }
        }
        mfma_rows(0, 0, 2);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            if (k < NDMA) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        unsigned tl_s1 = 0, tl_s2 = 0, tl_s3 = 0, tl_s4 = 0;
        if constexpr (MODE & 64) {
            tl_s1 = stamp();
            __builtin_amdgcn_sched_barrier(0);
        }
        mfma_rows(1, 0, 1);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (MODE & 64) {
            tl_s2 = stamp();
            __builtin_amdgcn_sched_barrier(0);
        }
        // outstanding in issue order: [.., B(t+1), A(t+1)] from iteration t-1, [B(t+2), A(t+2)] from this one
        // (older wave: its late pieces of k-tile t+1 sit in front of these four in issue order, so the same count covers them)
        if constexpr (!(MODE & 3) && ROLE != 2) dw_wait_vm<NBO + NAO>();
        if constexpr (ROLE == 1 && !(MODE & 1)) {   // the partner wave's pieces of k-tile t+2 (2 + 2 | 1 + 4), in what was this wave's barrier idle time
            if constexpr (!(MODE & 32)) {
#pragma unroll
                for (int it = NBO; it < 2 * NBO; ++it) pieceB(br[2], it);
            }
            if constexpr (!(MODE & 16)) {
#pragma unroll
                for (int it = NAO; it < 2 * NAO; ++it) pieceA(ar[DA], it);
            }
        }
        if constexpr (MODE & 64) {
            tl_s3 = stamp();
            asm volatile("" ::: "memory");
        }
        if constexpr (!(MODE & 2)) SW_BARRIER();
        if constexpr (MODE & 64) {
            tl_s4 = stamp();
            __builtin_amdgcn_sched_barrier(0);
        }
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
        if constexpr (MODE & 64) {
            const unsigned tl_s5 = stamp();
            tl[0] += tl_s1 - tl_s0;
            tl[1] += tl_s2 - tl_s1;
            tl[2] += tl_s3 - tl_s2;
            tl[3] += tl_s4 - tl_s3;
            tl[4] += tl_s5 - tl_s4;
            ++tl_n;
            __builtin_amdgcn_sched_barrier(0);
        }
        rotate();
    }
    };
    if constexpr (!ASYM) steady(dw_const<0>{});
    else if (older) steady(dw_const<1>{});
    else steady(dw_const<2>{});
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

    unsigned tl_e0 = 0;
    if constexpr (MODE & (64 | 256)) tl_e0 = stamp();
    if constexpr (!(MODE & 8)) sw_epilogue_any<1>(p, acc, m0 + wr * 64, n0 + wc * 128, tile_m * (BMT / 64) + wr, lr, lk, HW, smem + wave * (16 * SW_EPI_PITCH));
    static_assert(RINGS <= 160 * 1024 && SW_EPI_LDS <= 160 * 1024, "LDS");
    if constexpr (MODE & 64) {
        const unsigned tl_e1 = stamp();
        if (p.ws && lane == 0) {
            float* o = p.ws + ((size_t)blockIdx.x * 8 + wave) * 8;
#pragma unroll
            for (int i = 0; i < 5; ++i) o[i] = (float)tl[i];
            o[5] = (float)(tl_e1 - tl_e0);
            o[6] = (float)tl_n;
            o[7] = (float)(tl_e1 - tl_begin);
        }
    }
    if constexpr (MODE & 256) {
        __builtin_amdgcn_s_waitcnt(0);              // the wave's stores have left: exit = the moment the CU may take the next workgroup's wave
        const unsigned life3 = stamp();
        if (p.ws && lane == 0) {
            float* o = p.ws + ((size_t)blockIdx.x * 8 + wave) * 8;
            o[0] = (float)(life0 & 0xffffu);
            o[1] = (float)(life0 >> 16);
            o[2] = (float)(tl_begin - life0);
            o[3] = (float)(tl_e0 - tl_begin);
            o[4] = (float)(life3 - tl_e0);
            o[5] = (float)(__builtin_amdgcn_s_getreg((15 << 11) | 4) & 0xffffu);          // HW_ID[15:0]: wave, simd, pipe, cu, sh, se
            o[6] = (float)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xfu);             // XCC_ID
            o[7] = (float)tile;
        }
    }
}


// (Round 3 also shipped a form of this kernel with the nine taps of a channel slice unrolled - ring stages, M0 values and tap offsets as
//  compile-time constants, 47 instead of 98 instructions per k-tile: +0.35 %.  The asymmetric staging of round 4 (+2.1 ... +3.7 %)
//  needs the registers that form kept its constants in - with four more staging pointers it spilled inside the k-loop - so the
//  rolled loop above serves every launch now: git history, profiles/r03/dw8_slice_unrolled_ab.log.)

}  // namespace

bool dp_conv_dw_applies(const ConvH2Args& p, int bn) {
    const bool seg_ok = (!p.seg1 || (p.segC1 > 0 && p.segC1 % 32 == 0)) && (!p.seg2 || (p.seg1 && p.segC2 > 0 && p.segC2 % 32 == 0));
    if (bn != 256 && bn != 128) return false;
    return p.wfmt == 1 && p.afmt == 1 && p.passes == 1 && p.ksplit == 1 && p.M % (bn == 256 ? 256 : 512) == 0 && p.N % bn == 0 && p.C % 32 == 0 &&
           p.K >= 4 * 32 && (!p.temb || (p.H * p.W) % 32 == 0) && seg_ok && (p.rfmt == 0 || p.ofmt == 1) &&
           // the fp16 residual lands through 16-byte LDS-DMA pieces: rows and base 16-byte aligned, or the generic tiles take the launch
           (p.rfmt == 0 || (dp_aligned16(p.res) && p.ldr % 8 == 0));
}

void dp_launch_conv_dw(ConvH2Args& p, hipStream_t s, int bn) {
    if (bn == 128) {        // 512 x 128 tiles (SHAPE 1)
        p.tiles_n = p.N / 128;
        p.tiles = (p.M / 512) * p.tiles_n;
        hipLaunchKernelGGL((conv_igemm_dw<0, 1>), dim3((unsigned)p.tiles), dim3(512u), 0, s, p);
        return;
    }
    p.tiles_n = p.N / 256;
    p.tiles = (p.M / 256) * p.tiles_n;

    const dim3 g((unsigned)p.tiles), b(512u);
#ifdef DP_ABLATE   // timing ablations (WRONG RESULTS): only in libdiffpure_hip_ablate.so (tests/probes/build_ablate.py)
    {
        const char* e = getenv("DP_H2_DW_MODE");
        switch (e ? atoi(e) : 0) {
#define DW_CASE(M_) case M_: hipLaunchKernelGGL((conv_igemm_dw<M_>), g, b, 0, s, p); return
            DW_CASE(1); DW_CASE(2); DW_CASE(3); DW_CASE(6); DW_CASE(4); DW_CASE(5); DW_CASE(7); DW_CASE(8); DW_CASE(16); DW_CASE(32); DW_CASE(64); DW_CASE(128); DW_CASE(192); DW_CASE(256); DW_CASE(512);
#undef DW_CASE
            default: break;
        }
    }
#endif
    hipLaunchKernelGGL((conv_igemm_dw<0>), g, b, 0, s, p);
}
