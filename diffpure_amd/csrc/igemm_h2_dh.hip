// fp16 x fp16 implicit-GEMM convolution, 128 (pixels) x 256 (channels) tile, FOUR waves per workgroup, up to TWO workgroups per CU
// ("dh": the half-height twin of the 8-wave kernel of igemm_h2_dw.hip), for launches that do NOT fill the chip with 256 x 256 tiles.
//
// Why it exists (round 5).  Every 256-wide tile kernel of this library owns a whole CU per workgroup, so a launch of fewer than 256
// tiles leaves CUs idle: the 16 x 16 level of the CIFAR-10 NCSN++ at the adjoint benchmark's batch (B = 128: M = 32 768 rows x 256
// channels = 128 tiles of 256 x 256) ran on HALF the chip - forward, taped forward and input-gradient pass alike - and so do the middle
// levels of the guided UNet at the reference's own per-GPU batch of 4 (run_scripts/imagenet/run_in_rand_inf.sh:16).  Halving the tile
// height doubles the workgroups: 128 tiles become 256, one per CU, four waves each (one per SIMD).  The wave tile stays the 8-wave
// kernel's 64 x 128 (2 x 4 MFMA tiles of 32 x 32, 128 accumulator registers), so the epilogue (igemm_sw_common.h, NQ = 1), the
// operand formats, the reduction order and hence the BITS are those of every other variant - which tile kernel runs is the
// dispatcher's choice by (shape, batch) and never changes a result.
// Price: a 128 x 256 tile stages (128 + 256) operand rows per 128 x 256 x 32 products instead of (256 + 256) per 256 x 256 x 32 -
// 1.5x the LDS-DMA traffic per MFMA - and a lone wave per SIMD hides less: on launches that DO fill the chip the 8-wave kernel is
// faster (round 3 measured a two-workgroups-per-CU form of this tile at 811-870 vs 1 022-1 130 TFLOP/s), so the dispatcher takes this
// kernel only below 256 tiles of 256 x 256.  72 KB of LDS and <= 256 registers: two workgroups fit a CU, so 129 .. 255 such tiles
// (257 .. 510 half tiles) still run as one resident wave of workgroups.
//
// Per k-tile (32 channels of one tap) a wave issues 16 MFMAs, 12 ds_read_b128 and 6 LDS-DMA pieces (2 activation, 4 weight: wave w
// stages rows [32 w, 32 w + 32) of the A tile and [64 w, 64 w + 64) of the B tile).  Separate three-stage LDS rings for the two
// operands (prefetch distance 2), counted vmcnt, ONE barrier per k-tile:
//
//   iteration t:  issue DMA: weights of k-tile t+2, then activations of k-tile t+2          | 8 MFMA (t, s=0), ds_read (t, s=1)
//                 4 MFMA (t, s=1, row 0) ; s_waitcnt vmcnt(6) [k-tile t+1 landed] ; s_barrier
//                 ds_read fragments (t+1, s=0)                                          | 4 MFMA (t, s=1, row 1)
//   RAW: every wave waits for its own share of k-tile t+1 before the barrier of iteration t; the reads follow it.
//   WAR: the stages written in iteration t held k-tile t-1, whose last reads precede the barrier of iteration t-1.
// 1x1 K-segments as in the 8-wave kernel (a ResBlock's skip convolution folded into its second 3x3: igemm_h2.h).
// SPLIT-K: the <= 64-pixel levels (split factor fixed by the layer shape, igemm_h2.hip) take this kernel too - grid.y parts, each the
// same floor partition of the k-tiles as the generic tiles' and therefore the same partial sums; raw partials go to the workspace.
// Needs: fp16 activations and weights (a_fmt 1, w_fmt 1, passes 1), M % 128 == 0, N % 256 == 0, C % 32 == 0, >= 4 k-tiles per part.
#include <stdlib.h>

#include "dp_tune.h"
#include "igemm_h2.h"
#include "igemm_sw_common.h"

namespace {

constexpr int NT = 256;
constexpr int NXCD = 8;
constexpr int ADEPTH = 3, BDEPTH = 3, DA = ADEPTH - 1;
constexpr int RINGS = (ADEPTH + BDEPTH) * 0 + 3 * (128 + 256) * 64;     // 72 KB: three stages of (128 + 256) operand rows x 64 bytes (32 fp16), either tile shape
constexpr int EPI_LDS = 4 * 16 * SW_EPI_PITCH;                  // residual landing zone of four wave tiles: 68 KB (inside the dead rings)

template <int N>
__device__ __forceinline__ void dh_wait_vm() {
    static_assert(N == 6 || N == 0, "add the immediate");
    if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// BMT x BNT = 128 x 256 (waves 2 (M) x 2 (N)) or - layers with 128 output channels - 256 x 128 (waves 4 (M) x 1): the same 64 x 128 wave
// tile, the same 384 operand rows per k-tile and 6 DMA pieces per wave (NPA activation + NPB weight pieces of 16 rows), the same LDS.
template <int BMT>
__global__ __launch_bounds__(NT, 2) void conv_igemm_dh(ConvH2Args p) {
    static_assert(BMT == 128 || BMT == 256, "tile shapes");
    constexpr int BNT = 384 - BMT;
    constexpr int ATILE = BMT * 64, BTILE = BNT * 64;           // operand tiles of one k-tile
    constexpr int BBASE = ADEPTH * ATILE;
    constexpr int NPA = BMT / 64, NPB = BNT / 64;               // 16-row DMA pieces per wave and k-tile: 2 + 4 or 4 + 2
    __shared__ __attribute__((aligned(1024))) char smem[RINGS > EPI_LDS ? RINGS : EPI_LDS];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = BMT == 128 ? wave >> 1 : wave, wc = BMT == 128 ? wave & 1 : 0;
    int tile;
    {   // XCD-aware bijective remap (speed only)
        const int b = blockIdx.x, x = b % NXCD, q = p.tiles / NXCD, r = p.tiles % NXCD;
        tile = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + b / NXCD;
    }
    const int tile_n = tile % p.tiles_n, tile_m = tile / p.tiles_n;
    const int m0 = tile_m * BMT, n0 = tile_n * BNT;
    const int HW = p.H * p.W, Wp = p.W + 2, taps = p.KS * p.KS;
    // split-K (the <= 64-pixel levels, igemm_h2.hip::h2_ksplit - a function of the layer shape only): this workgroup reduces k-tiles
    // [t0, t0 + nt) of the layer's K / 32 - the same floor partition as the generic tiles, so the partial sums are the same bits -
    // and stores RAW partial sums; dp_conv2d_nhwc_h2 then runs splitk_epilogue_kernel
    const int ntot = p.K / 32;
    const int t0 = (int)(((long long)blockIdx.y * ntot) / p.ksplit);
    const int nt = (int)(((long long)(blockIdx.y + 1) * ntot) / p.ksplit) - t0;

    // ---- staging: 16 rows per DMA instruction; lane -> row (lane >> 2) of the piece, physical slot lane & 3, logical slot XOR-ed with
    // the row key (the fragment reads below undo it)
    const int lrow = lane >> 2;
    const int ls = (lane & 3) ^ ((lrow >> 2) & 3);
    const char* actr[NPA];                      // centre pixel of the lane's A row (segments: the lane's pixel), + slot
    const char* bptr[NPB];
#pragma unroll
    for (int it = 0; it < NPA; ++it) {
        const int m = m0 + wave * (BMT / 4) + it * 16 + lrow;
        const int b = m / HW, rem = m - b * HW;
        const int oy = rem / p.W, ox = rem - oy * p.W;
        actr[it] = p.x + ((size_t)(b * (p.H + 2) + oy + 1) * Wp + ox + 1) * p.C * 2 + ls * 16;
    }
#pragma unroll
    for (int it = 0; it < NPB; ++it) {
        const int n = n0 + wave * (BNT / 4) + it * 16 + lrow;       // block layout of the fp16 panels (ops.order_conv_weight_w16)
        bptr[it] = p.w + (size_t)(n >> 5) * p.K * 64 + (n & 31) * 16 + ls * 512;
    }
    // (tap, slice) of the next activation k-tile to stage, inside the current K-segment: segment 0 = the KS x KS convolution over
    // p.x (C / 32 slices of `taps` k-tiles), then the 1x1 segments over p.seg1 / p.seg2 (segC / 32 slices of one k-tile)
    int cur_tap = 0, cur_c = 0, cur_seg = 0, seg_slices = p.C / 32;
    auto enter_segment = [&](int seg, int slice) {
        cur_seg = seg;
        const char* sb = seg == 1 ? p.seg1 : p.seg2;
        const int sc = seg == 1 ? p.segC1 : p.segC2;
#pragma unroll
        for (int j = 0; j < NPA; ++j) actr[j] = sb + (size_t)(m0 + wave * (BMT / 4) + j * 16 + lrow) * sc * 2 + ls * 16;
        seg_slices = sc / 32;
        cur_c = slice;
        cur_tap = 0;
    };
    if (t0 != 0) {                              // a split-K part: start the cursor at k-tile t0 (tap fastest, then the channel slice)
        const int conv_tiles = taps * (p.C / 32);
        if (t0 < conv_tiles) {
            cur_c = t0 / taps;
            cur_tap = t0 - cur_c * taps;
        } else {                                // ... inside a 1x1 segment
            const int rem = t0 - conv_tiles;
            if (rem < p.segC1 / 32) enter_segment(1, rem);
            else enter_segment(2, rem - p.segC1 / 32);
        }
#pragma unroll
        for (int it = 0; it < NPB; ++it) bptr[it] += (size_t)t0 * 2048;
    }
    long long a_off = 0;
    auto pieceA = [&](int aoff, int it) {       // aoff: byte offset of the ring stage
        if (it == 0) {
            if (cur_c == seg_slices) enter_segment(cur_seg + 1, 0);     // (wave-uniform) this segment is staged: on to the next tensor
            if (cur_seg == 0) {
                const int ky = p.KS == 3 ? (cur_tap * 11) >> 5 : 0, kx = cur_tap - ky * p.KS;     // tap / 3 for tap < 9, no division
                a_off = ((long long)(ky - p.pad) * Wp + (kx - p.pad)) * p.C * 2 + (long long)cur_c * 64;
            } else {
                a_off = (long long)cur_c * 64;
            }
        }
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(actr[it] + a_off),
                                         (__attribute__((address_space(3))) void*)(smem + aoff + (wave * (BMT / 4) + it * 16) * 64), 16, 0, 0);
        if (it == NPA - 1 && (cur_seg != 0 || ++cur_tap == taps)) { cur_tap = 0; ++cur_c; }
    };
    auto pieceB = [&](int boff, int it) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)bptr[it],
                                         (__attribute__((address_space(3))) void*)(smem + boff + (wave * (BNT / 4) + it * 16) * 64), 16, 0, 0);
        bptr[it] += 2048;
    };
    auto issueA = [&](int aoff) {
#pragma unroll
        for (int it = 0; it < NPA; ++it) pieceA(aoff, it);
    };
    auto issueB = [&](int boff) {
#pragma unroll
        for (int it = 0; it < NPB; ++it) pieceB(boff, it);
    };
    // DMA piece d = 0 .. 5 of a k-tile in issue order: the NPB weight pieces, then the NPA activation pieces
    auto piece = [&](int aoff, int boff, int d) {
        if (d < NPB) pieceB(boff, d);
        else pieceA(aoff, d - NPB);
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

    // ring stage offsets, rotated once per k-tile: ar[0] / br[0] hold k-tile t, ar[1] / br[1] k-tile t+1, ar[2] / br[2] are written
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

    // ---- prologue (nt >= 4): B(0), A(0), A(1), B(1) in flight - in THAT order, because vmcnt counts in issue order and the
    // steady-state wait "everything up to the activations of k-tile t+1" must leave only younger pieces outstanding
    issueB(br[0]);
    issueA(ar[0]);
    issueA(ar[1]);
    issueB(br[1]);
    dh_wait_vm<6>();                            // k-tile 0 landed; A(1), B(1) may fly
    SW_BARRIER();
    read_frags(0, ar[0], br[0]);

    // steady state: k-tile t+2 exists
    int t = 0;
    for (; t + DA < nt; ++t) {
        // first half: 8 MFMAs on fragment set 0 | the 6 reads of set 1 and the 6 DMA pieces, one (read, piece) pair per MFMA shadow
        readA(1, ar[0], 0);
        piece(ar[DA], br[2], 0);
        readA(1, ar[0], 1);
        piece(ar[DA], br[2], 1);
        readB(1, br[0], 0);
        piece(ar[DA], br[2], 2);
        readB(1, br[0], 1);
        piece(ar[DA], br[2], 3);
        readB(1, br[0], 2);
        piece(ar[DA], br[2], 4);
        readB(1, br[0], 3);
        piece(ar[DA], br[2], 5);
        mfma_rows(0, 0, 2);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        mfma_rows(1, 0, 1);
        __builtin_amdgcn_sched_barrier(0);
        // outstanding in issue order: [.., B(t+1), A(t+1)] from iteration t-1, [B(t+2), A(t+2)] from this one
        dh_wait_vm<6>();
        SW_BARRIER();
        // second half: 4 MFMAs | the 6 reads of set 0 of k-tile t+1, two per MFMA shadow
        read_frags(0, ar[1], br[1]);
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
        dh_wait_vm<0>();
        SW_BARRIER();
        if (t + 1 < nt) read_frags(0, ar[1], br[1]);
        __builtin_amdgcn_sched_barrier(0);
        mfma_rows(1, 1, 2);
        __builtin_amdgcn_sched_barrier(0);
        rotate();
    }

    if (p.ksplit > 1) {     // raw partial sums of this part; bias / temb / residual / scale / column records happen in splitk_epilogue_kernel
        float* wsp = p.ws + (size_t)blockIdx.y * p.M * p.N;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float* d = wsp + (size_t)(m0 + wr * 64 + i * 32 + 4 * lk) * p.N + n0 + wc * 128 + j * 32 + lr;
#pragma unroll
                for (int r = 0; r < 16; ++r) d[(size_t)((r & 3) + 8 * (r >> 2)) * p.N] = acc[i][j][r];
            }
        return;
    }
    // after the last barrier nothing reads the rings any more: each wave lands its fp16 residual tile in its own 17 KB of them
    sw_epilogue_any<1>(p, acc, m0 + wr * 64, n0 + wc * 128, tile_m * (BMT / 64) + wr, lr, lk, HW, smem + wave * (16 * SW_EPI_PITCH));
}

}  // namespace

// bn = 256: 128 x 256 tiles (M % 128 == 0, N % 256 == 0); bn = 128: 256 x 128 tiles (M % 256 == 0, N % 128 == 0)
bool dp_conv_dh_applies(const ConvH2Args& p, int bn) {
    const int bm = 384 - bn;
    if (bn != 256 && bn != 128) return false;
    const bool seg_ok = (!p.seg1 || (p.segC1 > 0 && p.segC1 % 32 == 0)) && (!p.seg2 || (p.seg1 && p.segC2 > 0 && p.segC2 % 32 == 0));
    if (!(p.wfmt == 1 && p.afmt == 1 && p.passes == 1 && p.M % bm == 0 && p.N % bn == 0 && p.C % 32 == 0 && seg_ok)) return false;
    if (p.ksplit > 1)       // a split-K part needs its prologue's two k-tiles and two more; the epilogue is splitk_epilogue_kernel's
        return (p.K / 32) / p.ksplit >= 4 && p.ws != nullptr;
    return p.K >= 4 * 32 && (!p.temb || (p.H * p.W) % 32 == 0) && (p.rfmt == 0 || p.ofmt == 1) &&
           (p.rfmt == 0 || (dp_aligned16(p.res) && p.ldr % 8 == 0));      // 16-byte LDS-DMA pieces of the fp16 residual
}

void dp_launch_conv_dh(ConvH2Args& p, hipStream_t s, int bn) {
    p.tiles_n = p.N / bn;
    p.tiles = (p.M / (384 - bn)) * p.tiles_n;
    const dim3 g((unsigned)p.tiles, (unsigned)p.ksplit), b(NT);
    if (bn == 256) hipLaunchKernelGGL((conv_igemm_dh<128>), g, b, 0, s, p);
    else hipLaunchKernelGGL((conv_igemm_dh<256>), g, b, 0, s, p);
}
