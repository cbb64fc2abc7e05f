// fp16 x fp16 3x3 implicit-GEMM convolution, 256x256 tile, one wave per SIMD, with the activation operand staged as an
// X-HALO RUN: the three kx taps of a (32-channel slice, ky) pair read ONE run of pixels in LDS ("sx").
//
// Why: igemm_h2_sw.hip moves 32 KB of operands (16 KB activations + 16 KB weights) into LDS per k-tile of 128 MFMAs, 32
// LDS-DMA instructions per k-tile and CU.  Its timing ablations put the kernel at 950-1130 TFLOP/s against 1500-1880 with
// the loads removed: the L2 -> LDS path (32 B / clock / CU of the ~56 the L2 delivers, one 1 KB piece per 32 MFMA cycles)
// is the bound, not the matrix pipe.  Bytes, then: the nine taps of a 3x3 convolution read the SAME pixels shifted.  A tile of
// this kernel is 256 consecutive output pixels = R = 256 / W whole image rows (or one 256-pixel segment of a wider row), and
// for a fixed (slice c, ky) the pixels the three kx taps need are the contiguous run
//     bordered pixels  [(oy0 + ky) * (W + 2) + ox0,  ... + NR),    NR = R * (W + 2)   (258 for W >= 256)
// of the zero-bordered operand - output pixel (r, xl) of the tile reads LDS row r * (W + 2) + xl + kx for tap kx.  One run
// (17-18 DMA pieces of 16 rows x 64 B) replaces three A k-tiles (48 pieces): 65 pieces per three k-tiles instead of 96.
//
// Structure: a SUPER-TILE = (c, ky) = three k-tiles (kx = 0, 1, 2; reduction order c * 9 + ky * 3 + kx as in every other
// variant: bit-identical accumulators).  Two LDS stages of {A run 20 KB | B k-tiles 3 x 16 KB} = 136 KB; the DMA of
// super-tile u + 1 is issued during the first half of super-tile u (one instruction per MFMA shadow) and has the second
// half (48 MFMAs = 1536 cycles) to land; ONE barrier per super-tile (per 96 MFMAs of a wave) instead of one per k-tile.
//   RAW: every wave waits for its own DMAs (vmcnt(0)) before the barrier that ends super-tile u; all reads of u + 1 follow it.
//   WAR: stage (u + 1) & 1 was last read in super-tile u - 1, whose reads precede the barrier that ended it.
// Fragments are double-buffered in registers exactly as in igemm_h2_sw.hip (reads of k16 step h + 1 under the MFMAs of h).
// The XOR swizzle key of an LDS row is a function of the row itself ((row >> 2) & 3), so a shifted read (row + kx) uses the
// key of the row it lands on: the 24 fragment offsets (4 MFMA tile rows x 3 taps x 2 k16 steps) are precomputed registers,
// there is no address arithmetic per tap in the loop, and every ds_read_b128 lane group still touches 16 distinct bank
// slots (16 consecutive rows in any alignment cover the 16 positions).
// Needs: KS = 3, fp16 activations and weights, one pass, M % 256 == 0, N % 256 == 0, C % 32 == 0, H * W % 256 == 0,
// 32 <= W with 256 % W == 0 or W % 256 == 0 (a 32-row MFMA tile never straddles two image rows).
#include <stdlib.h>

#include "igemm_h2.h"
#include "igemm_sw_common.h"

namespace {

constexpr int NT = 256;
constexpr int NXCD = 8;
constexpr int APW = 5;                          // A pieces per wave and super-tile (pieces beyond the run repeat its last row)
constexpr int AREG = 4 * APW * 1024;            // A region of a stage: 20 pieces of 16 rows x 64 bytes
constexpr int BT = 256 * 64;                    // one B k-tile: 256 rows x 64 bytes (32 fp16)
constexpr int STAGE = AREG + 3 * BT;

#define SX_GLDS(src, dst)                                                                      \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src),     \
                                     (__attribute__((address_space(3))) void*)(dst), 16, 0, 0)

// instruction order of one 16-MFMA half: behind its first MFMA the 8 fragment reads of the next half, one per MFMA shadow,
// then NDMA LDS-DMA issues, one per MFMA shadow, then the rest of the MFMAs (see igemm_h2_sw.hip)
template <int NDMA>
__device__ __forceinline__ void sx_sched_half() {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    }
    if constexpr (NDMA == 6) {
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    } else if constexpr (NDMA == 5) {
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
    } else {
        __builtin_amdgcn_sched_group_barrier(0x008, 7, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
}

// MODE (timing ablations, DP_H2_SX_MODE; WRONG RESULTS): 1 = no DMA in the steady state, 2 = no barrier / vmcnt wait, 4 = no ds_reads
template <int MODE>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv_igemm_sx(ConvH2Args p) {
    __shared__ __attribute__((aligned(1024))) char smem[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    int tile;
    {   // XCD-aware bijective remap (speed only)
        const int b = blockIdx.x, x = b % NXCD, q = p.tiles / NXCD, r = p.tiles % NXCD;
        tile = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + b / NXCD;
    }
    const int tile_n = tile % p.tiles_n, tile_m = tile / p.tiles_n;
    const int m0 = tile_m * 256, n0 = tile_n * 256;
    const int HW = p.H * p.W, Wp = p.W + 2;
    const int NU = (p.C / 32) * 3;                      // super-tiles
    const int L = p.W < 256 ? p.W : 256;                // pixels of one image row inside the tile
    const int lsh = 31 - __builtin_clz(L);              // L is a power of two
    const int Lp = L + 2;                               // LDS rows per image-row segment (= W + 2 whenever the tile has several)
    const int NR = (256 / L) * Lp;                      // rows of the halo run

    // ---- the run of (ky = 0, slice 0): first bordered pixel of the tile's first image row
    const char* xrun;
    {
        const int b = m0 / HW, rem = m0 - b * HW;
        const int oy0 = rem / p.W, ox0 = rem - oy0 * p.W;
        xrun = p.x + ((size_t)(b * (p.H + 2) + oy0) * Wp + ox0) * p.C * 2;
    }
    const long long ky_step = (long long)Wp * p.C * 2;  // bytes from the run of ky to the run of ky + 1

    // ---- staging: a DMA instruction fills 16 LDS rows x 64 bytes; lane -> row (lane >> 2) of the piece, physical slot lane & 3,
    // logical slot XOR-ed with the row key (row >> 2) & 3 (pieces start at multiples of 16 rows: the key is the lane's own)
    const int lrow = lane >> 2;
    const int ls = (lane & 3) ^ ((lrow >> 2) & 3);
    unsigned voffA[APW];                        // wave w stages pieces w, w + 4, ..: byte offset of the lane's 16 bytes inside a run
#pragma unroll
    for (int j = 0; j < APW; ++j) {
        int prow = (wave + 4 * j) * 16 + lrow;
        prow = prow < NR ? prow : NR - 1;       // beyond the run: its last row again (never read; keeps the fetch inside the tensor)
        voffA[j] = (unsigned)prow * (unsigned)(p.C * 2) + ls * 16;
    }
    unsigned voffB[4];                          // wave w stages rows [64 w, 64 w + 64) of a B k-tile (block layout of the fp16 panels)
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int n = n0 + wave * 64 + it * 16 + lrow;
        voffB[it] = (unsigned)(n >> 5) * (unsigned)(p.K * 64) + (n & 31) * 16 + ls * 512;
    }
    int nky = 0, nc = 0, nkt = 0;               // (ky, slice) and first B k-tile of the next super-tile to stage
    auto issueA = [&](int stage, int j) {
        const char* src = xrun + (nky * ky_step + (long long)nc * 64);
        SX_GLDS(src + voffA[j], smem + stage * STAGE + (wave + 4 * j) * 1024);
    };
    auto issueB = [&](int stage, int kx, int it) {
        const char* src = p.w + (size_t)(nkt + kx) * 2048;
        SX_GLDS(src + voffB[it], smem + stage * STAGE + AREG + kx * BT + wave * 4096 + it * 1024);
    };
    auto advance = [&]() {
        nkt += 3;
        if (++nky == 3) { nky = 0; ++nc; }
    };

    // ---- fragments: lane -> row lr of a 32-row MFMA tile, k-half lk; 64-byte rows, slot (s*2 + lk) ^ key(row)
    const int lr = lane & 31, lk = lane >> 5;
    int aoff[4][3][2];                          // [MFMA tile row i][kx][k16 step s]: byte offset inside the A region
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ml = wr * 128 + i * 32 + lr;                  // pixel of the tile
        const int row0 = (ml >> lsh) * Lp + (ml & (L - 1));     // its LDS row for kx = 0
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int row = row0 + kx;
#pragma unroll
            for (int s = 0; s < 2; ++s) aoff[i][kx][s] = row * 64 + (((s * 2 + lk) ^ ((row >> 2) & 3)) << 4);
        }
    }
    int boff[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) boff[s] = AREG + (wc * 128 + lr) * 64 + (((s * 2 + lk) ^ ((lr >> 2) & 3)) << 4);
    half8 fa[2][4], fb[2][4];                   // [register set = k16 step][tile]

    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#define SX_READ(set, st, kx, s)                                                                                       \
    do {                                                                                                              \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) fa[set][i_] = *reinterpret_cast<const half8*>((st) + aoff[i_][kx][s]); \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_)                                                            \
            fb[set][j_] = *reinterpret_cast<const half8*>((st) + boff[s] + (kx) * BT + j_ * 2048);                    \
    } while (0)
#define SX_MFMA(set, i0, i1)                                                                                          \
    do {                                                                                                              \
        _Pragma("unroll") for (int i_ = i0; i_ < i1; ++i_)                                                          \
            _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_)                                                        \
                acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[set][i_], fb[set][j_], acc[i_][j_], 0, 0, 0); \
    } while (0)

    // ---- prologue: super-tile 0 staged and landed, fragments (kx 0, s 0) read
#pragma unroll
    for (int j = 0; j < APW; ++j) issueA(0, j);
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int it = 0; it < 4; ++it) issueB(0, kx, it);
    advance();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    SW_BARRIER();
    SX_READ(0, smem, 0, 0);

    int u = 0;
    for (; u + 1 < NU; ++u) {
        const int cur = u & 1, nxt = cur ^ 1;
        const char* st = smem + cur * STAGE;
        // half 0: (kx 0, s 0) | reads (kx 0, s 1) | A run of u + 1 and the first B piece
        if constexpr (!(MODE & 4)) SX_READ(1, st, 0, 1);
        if constexpr (!(MODE & 1)) {
#pragma unroll
            for (int j = 0; j < APW; ++j) issueA(nxt, j);
            issueB(nxt, 0, 0);
        }
        SX_MFMA(0, 0, 4);
        sx_sched_half<6>();
        // half 1: (kx 0, s 1) | reads (kx 1, s 0) | six B pieces
        if constexpr (!(MODE & 4)) SX_READ(0, st, 1, 0);
        if constexpr (!(MODE & 1)) {
            issueB(nxt, 0, 1);
            issueB(nxt, 0, 2);
            issueB(nxt, 0, 3);
            issueB(nxt, 1, 0);
            issueB(nxt, 1, 1);
            issueB(nxt, 1, 2);
        }
        SX_MFMA(1, 0, 4);
        sx_sched_half<6>();
        // half 2: (kx 1, s 0) | reads (kx 1, s 1) | the last five B pieces
        if constexpr (!(MODE & 4)) SX_READ(1, st, 1, 1);
        if constexpr (!(MODE & 1)) {
            issueB(nxt, 1, 3);
            issueB(nxt, 2, 0);
            issueB(nxt, 2, 1);
            issueB(nxt, 2, 2);
            issueB(nxt, 2, 3);
        }
        SX_MFMA(0, 0, 4);
        sx_sched_half<5>();
        // half 3: (kx 1, s 1) | reads (kx 2, s 0)
        if constexpr (!(MODE & 4)) SX_READ(0, st, 2, 0);
        SX_MFMA(1, 0, 4);
        sx_sched_half<0>();
        // half 4: (kx 2, s 0) | reads (kx 2, s 1)
        if constexpr (!(MODE & 4)) SX_READ(1, st, 2, 1);
        SX_MFMA(0, 0, 4);
        sx_sched_half<0>();
        // half 5: (kx 2, s 1); the staged super-tile has landed (this wave's share), barrier, first fragments of u + 1
        SX_MFMA(1, 0, 1);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!(MODE & 3)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr (!(MODE & 2)) SW_BARRIER();
        advance();
        if constexpr (!(MODE & 4)) SX_READ(0, smem + nxt * STAGE, 0, 0);
        SX_MFMA(1, 1, 4);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 1);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 1);
        __builtin_amdgcn_sched_barrier(0);
    }
    {   // the last super-tile: nothing left to stage
        const char* st = smem + (u & 1) * STAGE;
        SX_READ(1, st, 0, 1);
        SX_MFMA(0, 0, 4);
        sx_sched_half<0>();
        SX_READ(0, st, 1, 0);
        SX_MFMA(1, 0, 4);
        sx_sched_half<0>();
        SX_READ(1, st, 1, 1);
        SX_MFMA(0, 0, 4);
        sx_sched_half<0>();
        SX_READ(0, st, 2, 0);
        SX_MFMA(1, 0, 4);
        sx_sched_half<0>();
        SX_READ(1, st, 2, 1);
        SX_MFMA(0, 0, 4);
        sx_sched_half<0>();
        SX_MFMA(1, 0, 4);
    }
#undef SX_READ
#undef SX_MFMA

    sw_epilogue_any(p, acc, m0, n0, tile_m, wr, wc, lr, lk, HW);
}

}  // namespace

bool dp_conv_sx_applies(const ConvH2Args& p) {
    if (!(p.KS == 3 && p.wfmt == 1 && p.afmt == 1 && p.passes == 1 && p.ksplit == 1 && p.M % 256 == 0 && p.N % 256 == 0 && p.C % 32 == 0))
        return false;
    if (p.W < 32 || (p.H * p.W) % 256 != 0) return false;
    return p.W <= 256 ? 256 % p.W == 0 : p.W % 256 == 0;
}

void dp_launch_conv_sx(ConvH2Args& p, hipStream_t s) {
    p.tiles_n = p.N / 256;
    p.tiles = (p.M / 256) * p.tiles_n;
    const char* e = getenv("DP_H2_SX_MODE");
    switch (e ? atoi(e) : 0) {
        case 1: hipLaunchKernelGGL(conv_igemm_sx<1>, dim3((unsigned)p.tiles), dim3(NT), 0, s, p); break;
        case 2: hipLaunchKernelGGL(conv_igemm_sx<2>, dim3((unsigned)p.tiles), dim3(NT), 0, s, p); break;
        case 4: hipLaunchKernelGGL(conv_igemm_sx<4>, dim3((unsigned)p.tiles), dim3(NT), 0, s, p); break;
        case 7: hipLaunchKernelGGL(conv_igemm_sx<7>, dim3((unsigned)p.tiles), dim3(NT), 0, s, p); break;
        default: hipLaunchKernelGGL(conv_igemm_sx<0>, dim3((unsigned)p.tiles), dim3(NT), 0, s, p); break;
    }
}
