// fp16 x fp16 3x3 implicit-GEMM convolution with FEW output channels (N <= 32): the 6-channel head of the guided UNet
// (`out.2`, /root/reference/guided_diffusion/unet.py:623) - M = B * 65 536 pixels, K = 2304, N = 6.
//
// On the generic 64 x 64 tile kernel (igemm_h2.hip) this layer took 2.1 ms per UNet call at B=64: 58 of its 64 tile columns are
// padding, and - what actually bounds it - every one of the nine taps re-reads its activation k-tile from L2: 9 x 2.2 GB through
// the L2 -> LDS path for 116 GFLOP of useful work.  Here the tile is 256 pixels x 32 columns (the fp16 weight panels are padded
// to 32 rows anyway) and the activation operand is staged as an X-HALO RUN: for a (32-channel slice c, ky) pair the pixels the
// three kx taps need are ONE contiguous run of R * (W + 2) pixels of the zero-bordered operand (R = 256 / W image rows per tile;
// 258 pixels of one row when W >= 256) - output pixel (r, xl) reads LDS row r * (W + 2) + xl + kx for tap kx - so the operand
// crosses the L2 -> LDS path three times instead of nine.  The XOR swizzle key of an LDS row is a function of the row itself, so
// a shifted read uses the key of the row it lands on; the fragment offsets of the three taps are precomputed registers.
//
// 4 waves, each 64 pixels x 32 columns (2 MFMA tiles of 32 x 32, 32 accumulator registers); two LDS stages of {run 20 KB | three
// weight k-tiles 6 KB} = 52 KB -> three workgroups per CU, whose load / compute / store phases overlap each other: the k-loop is
// the plain double-buffered form (wait, barrier, issue the next super-tile, compute this one).  Same reduction order and the same
// MFMA operand roles as every other variant: bit-identical output.
// Needs: KS = 3, fp16 activations and weights, one pass, no split-K, N <= 32, M % 256 == 0, C % 32 == 0, H * W % 256 == 0,
// 32 <= W with 256 % W == 0 or W % 256 == 0; epilogue: bias and scale only, fp32 output, no column statistics.
#include <stdlib.h>

#include "igemm_h2.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

constexpr int NT = 256;
constexpr int NXCD = 8;
constexpr int APW = 5;                          // run pieces per wave and super-tile (pieces beyond the run repeat its last row)
constexpr int AREG = 4 * APW * 1024;            // run region of a stage: 20 pieces of 16 rows x 64 bytes
constexpr int BT = 32 * 64;                     // one weight k-tile: 32 rows x 64 bytes (32 fp16)
constexpr int STAGE = AREG + 3 * BT;

#define NN_GLDS(src, dst)                                                                      \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src),     \
                                     (__attribute__((address_space(3))) void*)(dst), 16, 0, 0)

__global__ __launch_bounds__(NT) void conv_igemm_nn(ConvH2Args p) {
    __shared__ __attribute__((aligned(1024))) char smem[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int tile;
    {   // XCD-aware bijective remap (speed only): consecutive image rows share an L2
        const int b = blockIdx.x, x = b % NXCD, q = p.tiles / NXCD, r = p.tiles % NXCD;
        tile = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + b / NXCD;
    }
    const int m0 = tile * 256;
    const int HW = p.H * p.W, Wp = p.W + 2;
    const int NU = (p.C / 32) * 3;                      // super-tiles (slice, ky)
    const int L = p.W < 256 ? p.W : 256;                // pixels of one image row inside the tile (a power of two)
    const int lsh = 31 - __builtin_clz(L);
    const int Lp = L + 2;                               // LDS rows per image-row segment
    const int NR = (256 / L) * Lp;                      // rows of the run

    const char* xrun;                                   // run of (ky = 0, slice 0): first bordered pixel of the tile's first image row
    {
        const int b = m0 / HW, rem = m0 - b * HW;
        const int oy0 = rem / p.W, ox0 = rem - oy0 * p.W;
        xrun = p.x + ((size_t)(b * (p.H + 2) + oy0) * Wp + ox0) * p.C * 2;
    }
    const long long ky_step = (long long)Wp * p.C * 2;

    // ---- staging: one DMA instruction fills 16 LDS rows x 64 bytes; lane -> row (lane >> 2), physical slot lane & 3, logical slot
    // XOR-ed with the row key (row >> 2) & 3
    const int lrow = lane >> 2;
    const int ls = (lane & 3) ^ ((lrow >> 2) & 3);
    unsigned voffA[APW];
#pragma unroll
    for (int j = 0; j < APW; ++j) {
        int prow = (wave + 4 * j) * 16 + lrow;
        prow = prow < NR ? prow : NR - 1;               // beyond the run: its last row again (never read; keeps the fetch inside the tensor)
        voffA[j] = (unsigned)prow * (unsigned)(p.C * 2) + ls * 16;
    }
    // weights: wave w < 3 stages k-tile kx = w of the super-tile, two pieces of 16 rows (block layout of the fp16 panels, one 32-row block)
    unsigned voffB[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) voffB[it] = (unsigned)(it * 16 + lrow) * 16 + ls * 512;
    int nky = 0, nc = 0, nkt = 0;                       // (ky, slice) and first weight k-tile of the next super-tile to stage
    auto issue = [&](int stage) {
        const char* src = xrun + (nky * ky_step + (long long)nc * 64);
#pragma unroll
        for (int j = 0; j < APW; ++j) NN_GLDS(src + voffA[j], smem + stage * STAGE + (wave + 4 * j) * 1024);
        if (wave < 3) {
            const char* wsrc = p.w + (size_t)(nkt + wave) * 2048;
#pragma unroll
            for (int it = 0; it < 2; ++it) NN_GLDS(wsrc + voffB[it], smem + stage * STAGE + AREG + wave * BT + it * 1024);
        }
        nkt += 3;
        if (++nky == 3) { nky = 0; ++nc; }
    };

    // ---- fragments: lane -> row lr of a 32-row MFMA tile, k-half lk
    const int lr = lane & 31, lk = lane >> 5;
    int aoff[2][3][2];                                  // [MFMA tile i][kx][k16 step s]
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int ml = wave * 64 + i * 32 + lr;
        const int row0 = (ml >> lsh) * Lp + (ml & (L - 1));
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int row = row0 + kx;
#pragma unroll
            for (int s = 0; s < 2; ++s) aoff[i][kx][s] = row * 64 + (((s * 2 + lk) ^ ((row >> 2) & 3)) << 4);
        }
    }
    int boff[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) boff[s] = AREG + lr * 64 + (((s * 2 + lk) ^ ((lr >> 2) & 3)) << 4);

    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    issue(0);
    for (int u = 0; u < NU; ++u) {
        const int cur = u & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's share of super-tile u has landed
        asm volatile("s_barrier" ::: "memory");                 // ... everybody's; and everybody has finished reading stage cur ^ 1
        if (u + 1 < NU) issue(cur ^ 1);
        const char* st = smem + cur * STAGE;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const half8 a0 = *reinterpret_cast<const half8*>(st + aoff[0][kx][s]);
                const half8 a1 = *reinterpret_cast<const half8*>(st + aoff[1][kx][s]);
                const half8 b = *reinterpret_cast<const half8*>(st + boff[s] + kx * BT);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b, acc[1], 0, 0, 0);
            }
    }

    // ---- epilogue: column lr (< N), rows (r & 3) + 8 (r >> 2) + 4 lk of each 32-row tile; bias and scale as in every variant
    if (lr < p.N) {
        const float bv = p.bias ? p.bias[lr] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wave * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                float v = acc[i][r] + bv;
                v *= p.scale;
                p.out[(size_t)row * p.ldo + lr] = v;
            }
    }
}

}  // namespace

bool dp_conv_nn_applies(const ConvH2Args& p) {
    if (!(p.KS == 3 && p.wfmt == 1 && p.afmt == 1 && p.passes == 1 && p.ksplit == 1 && p.N <= 32 && p.M % 256 == 0 && p.C % 32 == 0))
        return false;
    if (p.temb || p.res || p.colstats || p.ofmt) return false;
    if (p.W < 32 || (p.H * p.W) % 256 != 0) return false;
    return p.W <= 256 ? 256 % p.W == 0 : p.W % 256 == 0;
}

void dp_launch_conv_nn(ConvH2Args& p, hipStream_t s) {
    p.tiles_n = 1;
    p.tiles = p.M / 256;
    hipLaunchKernelGGL(conv_igemm_nn, dim3((unsigned)p.tiles), dim3(NT), 0, s, p);
}
