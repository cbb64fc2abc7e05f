// 3x3 implicit-GEMM convolution, 256x256 tile, fp16 activations x fp16 weights, one MFMA pass ("f16" / "f16sr"), with the
// activation operand staged as a 2-D HALO TILE.
//
// conv_igemm_h2_pp (igemm_h2_pp.hip) stages, for every k-tile (= one tap of one 32-channel slice), a fresh 256-row A tile by
// LDS-DMA - the nine taps of a slice fetch nine shifted copies of (almost) the same pixels.  With one MFMA pass per product
// that kernel is LDS-DMA-ISSUE bound (tests/probes/pp_ablate.py --w16: 860-970 TFLOP/s as shipped, 1270-1430 without the DMA):
// a wave has 16 MFMAs (512 cycles) per k-tile against 4 DMA instructions of 60-185 issue cycles each.  Here the 256 output
// pixels of a tile are R = 256 / W whole image rows, and ONE halo block of (R + 2) x (W + 2) pixels x 32 channels - contiguous
// pixels of the zero-bordered operand - is staged per channel slice and serves all nine taps: a tap is a row shift
// ky * (W + 2) + kx inside the block.  A DMA per slice: 7 instructions per thread at W = 256 (3 at W <= 32) instead of 18; per
// k-tile a wave issues 2 (weights) + <= 1 (halo piece of the NEXT slice) instead of 4.
//
// Everything else is the two-phase ping-pong schedule of conv_igemm_h2_pp<256, 256, ., 1, true, 1, true>: 8 waves = 2 groups
// (wr) x 4 (wc) one barrier apart, wave tile 128 x 64; phase 0 reads A (both 64-row halves, from the halo with the tap's
// shift) and B0 and computes quadrants Q00, Q10, phase 1 reads B1 and computes Q11, Q01; every phase stages into the other
// buffer what the same phase of the next k-tile reads (B0 / B1), plus - in phase 0 of tap t < pieces - halo piece t of the
// next slice into the other halo buffer:
//   staged in phase j -> retired by every wave's counted vmcnt in phase j+1 (only that phase's own loads may be in flight)
//   -> read in phase j+2 or later.   WAR: a B unit is re-staged two phases after its last read; the other halo buffer was
//   last read in phase 0 of the previous slice's last tap, two phases before the first piece is staged into it.
// Per-element accumulation order (channel slice outermost, taps, k16 steps) is the one of every other variant: results are
// bit-identical (tests/test_gpu_ops.py).
// LDS: 2 halo buffers x 56 KB (896 pixels x 64 B: the 774 pixels of W = 256 padded to whole 128-pixel DMA pieces) + 2 B tiles
// x 16 KB = 144 KB; rows of 64 bytes, 4 slots, XOR key (row >> 2) & 3 applied on the DMA source side and by the readers.
// Needs: KS = 3, W in {16, 32, 64, 128, 256}, H*W % 256 == 0 (a tile never straddles two images), C % 32 == 0, N % 256 == 0.
#include <stdlib.h>

#include "igemm_h2.h"
#include "igemm_pp_common.h"

namespace {

constexpr int NT = 512;
constexpr int NXCD = 8;
constexpr int HP_MAX = 896;                 // halo pixels incl. padding to whole DMA pieces
constexpr int HALO_B = HP_MAX * 64;
constexpr int TILE_B = 256 * 64;

__global__ __launch_bounds__(NT) void conv_igemm_halo(ConvH2Args p) {
    __shared__ __attribute__((aligned(1024))) char smem[2 * HALO_B + 2 * TILE_B];
    char* const hal = smem;                              // hal + (c & 1) * HALO_B
    char* const bt = smem + 2 * HALO_B;                  // bt + (t & 1) * TILE_B

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wr = wave >> 2, wc = wave & 3;
    int tile;
    {   // XCD-aware bijective remap (speed only)
        const int b = blockIdx.x, x = b % NXCD, q = p.tiles / NXCD, r = p.tiles % NXCD;
        tile = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + b / NXCD;
    }
    const int tile_n = tile % p.tiles_n, tile_m = tile / p.tiles_n;
    const int m0 = tile_m * 256, n0 = tile_n * 256;
    const int HW = p.H * p.W, Wp = p.W + 2;
    const int nsl = p.C / 32, nt = 9 * nsl;
    const int wsh = __builtin_ctz(p.W);                  // W is a power of two
    const int R = 256 >> wsh;
    const int hvalid = (R + 2) * Wp;                     // pixels of the halo block
    const int np = (hvalid + 127) >> 7;                  // 128-pixel DMA pieces (<= 7)

    // ---- halo staging: piece = 128 consecutive pixels of the bordered operand, lane -> pixel ua of the piece, slot tid & 3
    const int ua = tid >> 2;
    const int lsa = (tid & 3) ^ ((ua >> 2) & 3);         // logical slot fetched; key of LDS row piece*128 + ua is (ua >> 2) & 3
    long long horg;                                      // bordered pixel (b, row oy0 [= input row oy0 - 1], column 0), bytes from p.x
    {
        const int b = m0 / HW, oy0 = (m0 - b * HW) >> wsh;
        horg = ((long long)(b * (p.H + 2) + oy0) * Wp) * p.C * 2;
    }
    const unsigned hoff = (unsigned)ua * (unsigned)(p.C * 2) + lsa * 16;
    // the padding pixels of the last piece must not be read beyond the tensor: they re-read the block's last pixel
    const unsigned hoff_last = (unsigned)(min((np - 1) * 128 + ua, hvalid - 1) - (np - 1) * 128) * (unsigned)(p.C * 2) + lsa * 16;
    const char* const hbase = p.x + pp_uniform(horg);
    const int ua0 = wave * 16;                           // first pixel of this wave inside a piece
    auto stage_halo = [&](char* hbuf, int piece, int c32) {
        const char* sb = hbase + pp_uniform((long long)piece * 128 * p.C * 2 + (long long)c32 * 64);
        pp_glds(piece == np - 1 ? hoff_last : hoff, sb, hbuf + (piece * 128 + ua0) * 64);
    };
    // ---- weight staging: B unit b = rows {64 wc' + 32 b + 0..31}: one 128-row piece per unit
    unsigned boff[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {   // row (ua >> 5) * 64 + b * 32 + (ua & 31) of the tile, block layout of the fp16 panels
        const int row = (ua >> 5) * 64 + b * 32 + (ua & 31);
        boff[b] = (unsigned)(row >> 5) * (unsigned)(p.K * 64) + (row & 31) * 16 + lsa * 512;
    }
    const char* const bbase = p.w + pp_uniform((long long)(n0 >> 5) * p.K * 64);
    const int bdst = ((ua0 >> 5) * 64 + (ua0 & 31)) * 64;
    auto stage_b = [&](char* buf, int b, long long off) {
        pp_glds(boff[b], bbase + pp_uniform(off), buf + bdst + b * 32 * 64);
    };

    // ---- fragment addressing
    const int lr = lane & 31, lk = lane >> 5;
    int arow[2][2];                                      // halo row of output pixel (wr*128 + sub*64 + i*32 + lr) at tap (0, 0)
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int ml = wr * 128 + sub * 64 + i * 32 + lr;
            arow[sub][i] = (ml >> wsh) * Wp + (ml & (p.W - 1));
        }
    const int brow = (wc * 64 + lr) * 64;
    int soffb[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) soffb[s] = ((s * 2 + lk) ^ ((lr >> 2) & 3)) << 4;

    half8 fa[2][2][2];                                   // [half][m-tile i][k16 step s]
    half8 fb0[2], fb1[2];
    auto read_a = [&](const char* hbuf, int tsh) {
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = arow[sub][i] + tsh;
                const int key = (row >> 2) & 3;
                const char* r = hbuf + row * 64;
#pragma unroll
                for (int s = 0; s < 2; ++s) fa[sub][i][s] = *reinterpret_cast<const half8*>(r + (((s * 2 + lk) ^ key) << 4));
            }
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#define HL_MFMA(ASUB, BSUB, BH)                                                                                   \
    do {                                                                                                          \
        _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                                           \
            _Pragma("unroll") for (int i = 0; i < 2; ++i) acc[ASUB * 2 + i][BSUB] =                               \
                __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[ASUB][i][s], BH[s], acc[ASUB * 2 + i][BSUB], 0, 0, 0);  \
        }                                                                                                         \
    } while (0)
#define HL_SYNC(N)                                                                  \
    do {                                                                            \
        if ((N) >= 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");              \
        else if ((N) == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");         \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       \
        PP_BARRIER();                                                               \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                          \
        __builtin_amdgcn_sched_barrier(0);                                          \
    } while (0)
#define HL_END()                                    \
    do {                                            \
        __builtin_amdgcn_sched_barrier(0);          \
        PP_BARRIER();                               \
    } while (0)

    // ---- prologue: the whole halo of slice 0 and both B units of k-tile 0, drained
    for (int piece = 0; piece < np; ++piece) stage_halo(hal, piece, 0);
    stage_b(bt, 0, 0);
    stage_b(bt, 1, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PP_BARRIER();
    if (grp == 1) PP_BARRIER();         // group 1 runs one interval behind group 0

    int c = 0, tap = 0;
    for (int t = 0; t < nt; ++t) {
        const char* hcur = hal + (c & 1) * HALO_B;
        char* hnxt = hal + ((c + 1) & 1) * HALO_B;
        const char* bcur = bt + (t & 1) * TILE_B;
        char* bnxt = bt + ((t + 1) & 1) * TILE_B;
        const bool more1 = t + 1 < nt;
        const bool piece_now = tap < np && c + 1 < nsl;          // halo piece `tap` of the next slice rides with this k-tile
        const long long offb1 = (long long)(t + 1) * 2048;
        const int ky = tap / 3, kx = tap - ky * 3;
        const int tsh = ky * Wp + kx;

        // phase 0: A (both halves, shifted by the tap) and B0 -> Q00, Q10
        read_a(hcur, tsh);
#pragma unroll
        for (int s = 0; s < 2; ++s) fb0[s] = *reinterpret_cast<const half8*>(bcur + brow + soffb[s]);
        if (more1) stage_b(bnxt, 0, offb1);
        if (piece_now) stage_halo(hnxt, tap, c + 1);
        HL_SYNC((more1 ? 1 : 0) + (piece_now ? 1 : 0));
        HL_MFMA(0, 0, fb0);
        HL_MFMA(1, 0, fb0);
        HL_END();

        // phase 1: B1 -> Q11, Q01
#pragma unroll
        for (int s = 0; s < 2; ++s) fb1[s] = *reinterpret_cast<const half8*>(bcur + brow + 32 * 64 + soffb[s]);
        if (more1) stage_b(bnxt, 1, offb1);
        HL_SYNC(more1 ? 1 : 0);
        HL_MFMA(1, 1, fb1);
        HL_MFMA(0, 1, fb1);
        HL_END();

        if (++tap == 9) { tap = 0; ++c; }
    }
    if (grp == 0) PP_BARRIER();         // re-align the two groups (barrier counts must match)
#undef HL_MFMA
#undef HL_SYNC
#undef HL_END

    pp_epilogue<256>(p, acc, m0, n0, tile_m, wr, wc, lr, lk, HW);
}

}  // namespace

// min_w: the dispatcher's speed threshold (measured, tests/probes/pp_ablate.py --w16, B=64, TFLOP/s per-tap -> halo: W=256
// 865 -> 896 and 975 -> 975, W=128 862 -> 912, W=64 928 -> 1000, W=32 916 -> 935, W=16 959 -> 917); correctness holds from 16.
bool dp_conv_halo_applies(const ConvH2Args& p, int min_w) {
    const int W = p.W;
    return p.wfmt == 1 && p.afmt == 1 && p.passes == 1 && p.KS == 3 && p.ksplit == 1 && W >= min_w && W >= 16 && W <= 256 && (W & (W - 1)) == 0 &&
           (p.H * p.W) % 256 == 0 && p.M % 256 == 0 && p.N % 256 == 0 && p.C % 32 == 0;
}

void dp_launch_conv_halo(ConvH2Args& p, hipStream_t s) {
    p.tiles_n = p.N / 256;
    p.tiles = (p.M / 256) * p.tiles_n;
    hipLaunchKernelGGL(conv_igemm_halo, dim3((unsigned)p.tiles), dim3(NT), 0, s, p);
}
