// Implicit-GEMM convolution on the fp16 matrix cores with fp32-class accuracy ("f16x3").
//
// Every fp32 operand v is carried as TWO fp16 numbers, hi = fp16(v) and lo = fp16(v - hi)
// (22 significant bits together; gfx950's MFMA honours fp16 subnormals, probed in
// tests/probes/mfma_f16_probe.hip), and every product is evaluated with three
// v_mfma_f32_32x32x16_f16 passes into ONE fp32 accumulator:
//        a*w  ~=  a_lo*w_hi + a_hi*w_lo + a_hi*w_hi          (a_lo*w_lo ~ 2^-22 |a w| dropped)
// Measured effect on purified pixels (oracle/precision_study.py, full NCSN++, 20-step loop):
// 2.4e-6 max-abs vs fp32 - against 8.8e-4 for single-pass fp16 and 7.1e-3 for bf16 - at a matrix
// ceiling of 2.5 PFLOP/s / 3 = 833 TFLOP/s instead of the 157 TFLOP/s of fp32-input MFMA.
//
// "h2" tensor format (written by gn_apply, csrc/norm.hip, and by dp_pack_h2):
//   channels in blocks of 8:  [ 8 x fp16 hi | 8 x fp16 lo ]  = 32 bytes per 8 channels,
//   i.e. 4 bytes per element like fp32, and a 32-channel slice of one pixel is 128 contiguous bytes.
//
// Reduction order: k' = (c32 * KS*KS + tap) * 32 + (ci % 32)  - channel-slice-major, the KS*KS
// taps INNERMOST.  Consecutive k-tiles of a workgroup therefore re-read the same 3 x 130-pixel x
// 128-byte halo strip nine times (reuse distance ~50 KB per workgroup, ~3 MB per XCD: the kx
// shifts hit L1, the ky rows hit the XCD's 4 MB L2) instead of streaming a fresh 128 KB per tap
// (reuse distance 8 MB per XCD with the tap-major order, which sent the whole A stream to the
// fabric).  Weights are packed [N][K'] in the same order by the host (ops.pack_conv_weight_h2).
//
// Tile: 128x128x32 per 256-thread workgroup (4 waves as 2x2, each 2x2 MFMA tiles of 32x32).
// Operands arrive by LDS-DMA (global_load_lds_dwordx4): no staging VGPRs, no ds_write pass.  The
// DMA writes LDS linearly (wave-uniform base + lane*16), so the XOR swizzle of the LDS image is
// applied to the SOURCE address: lane l fills physical slot (l&7) of row (l>>3) with the LOGICAL
// slot (l&7) ^ ((row>>1)&7); readers apply the same involution -> every ds_read_b128 lane group
// touches 16 distinct bank slots.
// The activation operand carries a ONE-PIXEL ZERO BORDER ([B][H+2][W+2][C] h2, written by
// gn_apply): every tap of every output pixel is an in-range read, so the loader has no bounds
// tests, no predication and no per-tap pointer selection - one 64-bit add per DMA piece (PMC on
// the masked version: 28 % of wave cycles went to issuing ~210 non-MFMA instructions per k-tile).
// Workgroups are mapped to tiles XCD-aware: the hardware places workgroup b on XCD b % 8, so tile
// ids are dealt in contiguous chunks per XCD and the n-tiles / halo neighbours of an m-tile share
// one L2.
#include <mutex>
#include <stdlib.h>
#include <type_traits>

#include "dp_tune.h"
#include "igemm_h2.h"

// Every tile variant must produce the SAME bits, column records included (a batch's sharding picks the variant): products and sums stay
// separate IEEE operations in this file - left to itself the compiler contracts `cq += v * v` (and `v *= scale; cs += v`) into FMAs in one
// code shape and not in another.  Explicit fmaf() calls are unaffected.
#pragma clang fp contract(off)

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

constexpr int NT = 256;
constexpr int NXCD = 8;


// Tile variants of THIS file (256 threads = 2x2 waves, wave tile (BM/2) x (BN/2) of 32x32 MFMA tiles, two LDS stages, two
// workgroups per CU) - used where the 8-wave ping-pong kernel of igemm_h2_pp.hip does not apply (few tiles, ragged
// shapes, split-K levels):
//   <128,128,32>  TM=TN=2   64 KB LDS   8 DMA + 16 ds_read per 24 MFMA per wave   (general)
//   <64,64,32>    TM=TN=1   32 KB LDS                                              (few tiles, N <= 64)
// The template also instantiates with BKH = 16 and 4-wave wide tiles (<128,256,16>, <256,128,16>: 342 vs 344 TFLOP/s,
// not faster - 64-byte DMA rows are half cache lines) and with ABL = 1..3 (s_setprio / timing ablations: as is 337
// TFLOP/s on 256^2 x 256->256 at B=8, without the per-tile wait+barrier 338, without any operand DMA 445); none of
// those is built any more.
// BKH = k elements per LDS stage (32 or 16); an LDS row holds BKH (hi,lo) pairs = BKH*4 bytes = SPR 16-byte
// slots.  XOR swizzle of the slot index with row bits chosen so that 16 consecutive rows at one logical slot
// cover 16 distinct 16-byte bank positions: (row>>1)&7 for 128-byte rows, (row>>2)&3 for 64-byte rows.
template <int ROWBYTES>
__device__ __forceinline__ int swz(int row, int slot) {
    if constexpr (ROWBYTES == 128) return row * 128 + ((slot ^ ((row >> 1) & 7)) << 4);
    else if constexpr (ROWBYTES == 64) return row * 64 + ((slot ^ ((row >> 2) & 3)) << 4);
    else return row * 32 + ((slot ^ ((row >> 3) & 1)) << 4);
}

// A16: the activation operand is PLAIN fp16 ("h1": [B][H+2][W+2][C] fp16, zero border) - 2 bytes per element, LDS rows of
// BKH * 2 bytes - and only a_hi exists: PASSES is 2 (a_hi*w_lo + a_hi*w_hi, "f16x2") or 1 (a_hi*w_hi, "f16").  The
// per-element accumulation order is the one of the h2-operand kernel with the a_lo pass left out.
// W16 (with A16, PASSES 1): the weight panel is plain fp16 as well ([N][K'] fp16, same k' order; rounded once at load
// ("f16") or re-rounded stochastically from the fp32 masters before every network call ("f16sr", dp_round_weights)):
// B rows are BKH * 2 bytes and only w_hi fragments exist.
template <int BM, int BN, int BKH, int ABL, int PASSES, bool A16, bool W16>
__global__ __launch_bounds__(NT, 2) void conv_igemm_h2(ConvH2Args p) {
    static_assert(A16 ? (PASSES == 2 || PASSES == 1) : (PASSES == 3 || PASSES == 12), "operand format / passes");
    static_assert(!W16 || (A16 && PASSES == 1), "fp16 weights: one pass, fp16 activations");
    constexpr int TM = BM / 64, TN = BN / 64;           // 2x2 waves, 32x32 MFMA tiles
    constexpr int ESZ = A16 ? 2 : 4;                    // bytes per activation element
    constexpr int AROWB = BKH * ESZ, ASPR = AROWB / 16; // bytes / 16-byte slots per LDS row of the A tile
    constexpr int WSZ = W16 ? 2 : 4;                    // bytes per weight element (plain fp16, or hi|lo)
    constexpr int ROWB = BKH * WSZ, SPR = ROWB / 16;    // ... of the B (weight) tile
    constexpr int ARPP = NT / ASPR, ARPI = 64 / ASPR;   // A rows staged per pass of the workgroup / per DMA instruction
    constexpr int RPP = NT / SPR, RPI = 64 / SPR;       // B rows ...
    constexpr int A_IT = BM / ARPP, B_IT = BN / RPP;
    constexpr int NS = BKH / 16, NSUB = 32 / BKH;       // k16 sub-steps per stage; stages per 32-channel slice
    constexpr int STAGE = BM * AROWB + BN * ROWB;
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave >> 1) * (TM * 32), wn0 = (wave & 1) * (TN * 32);
    // XCD-aware bijective remap of the workgroup id to a tile id (speed only, any placement is correct)
    int tile;
    {
        const int b = blockIdx.x, x = b % NXCD, q = p.tiles / NXCD, r = p.tiles % NXCD;
        tile = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + b / NXCD;
    }
    const int tile_n = tile % p.tiles_n, tile_m = tile / p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int HW = p.H * p.W;
    const int r0 = tid / SPR, r0a = tid / ASPR;
    // logical slot this lane fetches: physical slot (tid % SPR) un-swizzled with the row key, which is the
    // same for every pass because RPP / ARPP are multiples of 16
    const int ls = (ROWB == 128) ? ((tid & 7) ^ ((r0 >> 1) & 7)) : (ROWB == 64 ? ((tid & 3) ^ ((r0 >> 2) & 3)) : ((tid & 1) ^ ((r0 >> 3) & 1)));
    const int lsa = (AROWB == 128) ? ((tid & 7) ^ ((r0a >> 1) & 7)) : (AROWB == 64 ? ((tid & 3) ^ ((r0a >> 2) & 3)) : ((tid & 1) ^ ((r0a >> 3) & 1)));
    const int taps = p.KS * p.KS;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);      // provably uniform: LDS-DMA bases stay scalar
    const int Wp = p.W + 2;

    // per staged A row: pointer to the CENTRE pixel inside the zero-bordered tensor
    const char* ctr[A_IT];
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
        const int m = min(m0 + r0a + it * ARPP, p.M - 1);   // tail rows re-read the last pixel; never stored
        const int b = m / HW, rem = m - b * HW;
        const int oy = rem / p.W, ox = rem - oy * p.W;
        ctr[it] = p.x + ((size_t)(b * (p.H + 2) + oy + 1) * Wp + ox + 1) * p.C * ESZ + lsa * 16;
    }
    const char* bptr[B_IT];
    int bstep[B_IT];
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
        const int n = n0 + r0 + it * RPP;
        const bool ok = n < p.N;
        if constexpr (W16) {    // block layout of the fp16 panels (ops.order_conv_weight_w16): 16-byte piece (n, k'/8 = g) at
                                // ((n/32 * K/8 + g) * 32 + n%32) * 16; a 32-channel k-tile = 4 groups = 2048 bytes further on
            static_assert(BKH == 32, "fp16 weight panels are staged one 32-channel k-tile at a time");
            bptr[it] = ok ? p.w + (size_t)(n >> 5) * p.K * 64 + (n & 31) * 16 + ls * 512 : p.zero + ls * 16;
            bstep[it] = ok ? 2048 : 0;
        } else {
            bptr[it] = ok ? p.w + (size_t)n * p.K * WSZ + ls * 16 : p.zero + ls * 16;
            bstep[it] = ok ? ROWB : 0;
        }
    }

    // workgroup-uniform cursor of the k-stage being staged, in weight order: 32-channel slice c32 outermost,
    // then the tap, then (BKH == 16) the half of the slice
    // split-K: this workgroup reduces k-stages [t0, t0 + nt) of the layer's K / BKH
    // (with 1x1 K-segments the total is not always a multiple of the split factor: floor boundaries - equal parts whenever it is)
    const int ntot = p.K / BKH;
    const int t0 = (int)(((long long)blockIdx.y * ntot) / p.ksplit);
    const int nt = (int)(((long long)(blockIdx.y + 1) * ntot) / p.ksplit) - t0;
    int cur_h = t0 % NSUB, cur_tap = (t0 / NSUB) % taps, cur_c = (t0 / NSUB) / taps;
    // K-segments (fp16 x fp16 only; igemm_h2.h): after the C / 32 slices of the KS x KS convolution the cursor runs over the slices of
    // the plain fp16 NHWC tensors p.seg1 / p.seg2, one k-stage per slice, no tap offset; the row pointers are re-based per segment
    int cur_seg = 0, seg_slices = p.C / 32;
    auto enter_segment = [&](int seg, int slice) {
        cur_seg = seg;
        const char* sb = seg == 1 ? p.seg1 : p.seg2;
        const int sc = seg == 1 ? p.segC1 : p.segC2;
#pragma unroll
        for (int it = 0; it < A_IT; ++it) ctr[it] = sb + (size_t)min(m0 + r0a + it * ARPP, p.M - 1) * sc * ESZ + lsa * 16;
        seg_slices = sc / 32;
        cur_c = slice;
        cur_tap = 0;
    };
    if constexpr (W16) {
        if (p.seg1 && cur_c >= seg_slices) {        // a split-K part that starts inside a segment
            int rem = cur_c * taps + cur_tap - seg_slices * taps;       // stages behind the KS x KS part (NSUB == 1)
            if (rem < p.segC1 / 32) enter_segment(1, rem);
            else enter_segment(2, rem - p.segC1 / 32);
        }
    }
#pragma unroll
    for (int it = 0; it < B_IT; ++it) bptr[it] += (size_t)t0 * bstep[it];
    auto issue = [&](int stage) {
        if constexpr (W16) {
            if (cur_c == seg_slices && p.seg1) enter_segment(cur_seg + 1, 0);      // (workgroup-uniform) on to the next tensor
        }
        const int ky = cur_tap / p.KS, kx = cur_tap - ky * p.KS;
        const long long off = cur_seg ? (long long)cur_c * (32 * ESZ)
                                      : ((long long)(ky - p.pad) * Wp + (kx - p.pad)) * p.C * ESZ + (long long)cur_c * (32 * ESZ) + cur_h * AROWB;
        char* As = smem + stage * STAGE + wave_u * ARPI * AROWB;
        char* Bs = smem + stage * STAGE + BM * AROWB + wave_u * RPI * ROWB;
#pragma unroll
        for (int it = 0; it < A_IT; ++it)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ctr[it] + off),
                                             (__attribute__((address_space(3))) void*)(As + it * ARPP * AROWB), 16, 0, 0);
#pragma unroll
        for (int it = 0; it < B_IT; ++it) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)bptr[it],
                                             (__attribute__((address_space(3))) void*)(Bs + it * RPP * ROWB), 16, 0, 0);
            bptr[it] += bstep[it];
        }
        if (NSUB == 1 || ++cur_h == NSUB) {
            cur_h = 0;
            if (cur_seg != 0 || ++cur_tap == taps) { cur_tap = 0; ++cur_c; }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int lr = lane & 31, lk = lane >> 5;
    auto compute = [&](int stage) {
        const char* As = smem + stage * STAGE;
        const char* Bs = As + BM * AROWB;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            half8 ah[TM], al[TM], bh[TN], bl[TN];
            const int sl = s * 4 + lk * 2;  // hi slot; lo slot = sl + 1
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = wm0 + i * 32 + lr;
                if constexpr (A16) {
                    ah[i] = *reinterpret_cast<const half8*>(As + swz<AROWB>(row, s * 2 + lk));
                } else {
                    ah[i] = *reinterpret_cast<const half8*>(As + swz<AROWB>(row, sl));
                    al[i] = *reinterpret_cast<const half8*>(As + swz<AROWB>(row, sl + 1));
                }
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int row = wn0 + j * 32 + lr;
                if constexpr (W16) {
                    bh[j] = *reinterpret_cast<const half8*>(Bs + swz<ROWB>(row, s * 2 + lk));
                } else {
                    bh[j] = *reinterpret_cast<const half8*>(Bs + swz<ROWB>(row, sl));
                    bl[j] = *reinterpret_cast<const half8*>(Bs + swz<ROWB>(row, sl + 1));
                }
            }
            if constexpr (ABL == 1) __builtin_amdgcn_s_setprio(1);
            if constexpr (PASSES == 3 || PASSES == 12) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
            }
            if constexpr (PASSES == 3 || PASSES == 2) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
            if constexpr (ABL == 1) __builtin_amdgcn_s_setprio(0);
        }
    };

    issue(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int t = 0; t < nt; ++t) {
        const int cur = t & 1;
        if constexpr (ABL == 2) {            // TIMING ABLATION ONLY (wrong results): no DMA after the prologue
            compute(cur);
            __syncthreads();
            continue;
        }
        if (t + 1 < nt) issue(cur ^ 1);      // DMA of stage t+1 flies under the MFMAs of stage t
        compute(cur);
        if constexpr (ABL == 3) continue;    // TIMING ABLATION ONLY (wrong results): no wait, no barrier
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    if (p.ksplit > 1) {      // raw partial sums; bias / temb / residual / scale / column sums happen in splitk_epilogue_kernel
        float* wsp = p.ws + (size_t)blockIdx.y * p.M * p.N;
        const int lr_ = lane & 31, lk_ = lane >> 5;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn0 + j * 32 + lr_;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk_;
                    if (row < p.M && col < p.N) wsp[(size_t)row * p.N + col] = acc[i][j][r];
                }
        }
        return;
    }
    // column partials in the tile-shape-independent order of igemm.hip (32-row sub-sums -> 64-row records)
    float* cs_lds = reinterpret_cast<float*>(smem);     // [BM/32][BN][2] floats, tiles released by the last barrier
    // The residual reads of one column block are all issued before their first use (TM x 16 loads in flight per
    // lane) instead of one dependent round trip per element; temb is one value per 32-row block when a block
    // cannot straddle two samples.  Value order (bias, temb, residual, scale) is the same in every variant.
    const float* __restrict__ resp = p.res;
    const float* __restrict__ tembp = p.temb;
    float* __restrict__ outp = p.out;
    _Float16* __restrict__ outh = reinterpret_cast<_Float16*>(p.out);      // p.ofmt 1: fp16 output
    const bool hw32 = HW % 32 == 0;
    // the output format is a compile-time flag of the store loop (tested per store it costs a branch per element)
    auto store_tile = [&](auto out16) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + wn0 + j * 32 + lr;
        const bool cok = col < p.N;
        const float bv = (cok && p.bias) ? p.bias[col] : 0.f;
        float rv[TM][16];
        float tv[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int rowb = m0 + wm0 + i * 32 + 4 * lk;
            if (resp) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = rowb + (r & 3) + 8 * (r >> 2);
                    rv[i][r] = (cok && row < p.M) ? dp_conv_res(p, (size_t)row, col) : 0.f;
                }
            }
            tv[i] = (tembp && hw32 && cok && rowb < p.M) ? tembp[(size_t)(rowb / HW) * p.temb_stride + col] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            // column records: the lane's 16 values as TWO chains (even r, odd r), then their sum - the order of igemm_sw_common.h,
            // whose packed fp32 arithmetic works on row pairs
            float cs2[2] = {0.f, 0.f}, cq2[2] = {0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (row >= p.M || !cok) continue;
                float v = acc[i][j][r] + bv;
                if (tembp) v += hw32 ? tv[i] : tembp[(size_t)(row / HW) * p.temb_stride + col];
                if (resp) v += rv[i][r];
                v *= p.scale;
                if constexpr (decltype(out16)::value) outh[(size_t)row * p.ldo + col] = dp_to_half(v);
                else outp[(size_t)row * p.ldo + col] = v;
                cs2[r & 1] += v;
                cq2[r & 1] += v * v;
            }
            float cs = cs2[0] + cs2[1], cq = cq2[0] + cq2[1];
            if (p.colstats) {
                cs += __shfl_xor(cs, 32, 64);
                cq += __shfl_xor(cq, 32, 64);
                if (lk == 0) {
                    float* d = cs_lds + ((((wave >> 1) * TM + i) * BN) + wn0 + j * 32 + lr) * 2;
                    d[0] = cs;
                    d[1] = cq;
                }
            }
        }
    }
    };
    if (p.ofmt) store_tile(std::true_type{});
    else store_tile(std::false_type{});
    if (p.colstats) {
        __syncthreads();
        constexpr int REC = BM / 64;
        for (int c = tid; c < BN * REC; c += NT) {
            const int rec = c / BN, cc = c - rec * BN;
            if (n0 + cc >= p.N) continue;
            const float s0 = cs_lds[((2 * rec) * BN + cc) * 2] + cs_lds[((2 * rec + 1) * BN + cc) * 2];
            const float q0 = cs_lds[((2 * rec) * BN + cc) * 2 + 1] + cs_lds[((2 * rec + 1) * BN + cc) * 2 + 1];
            float* d = p.colstats + (size_t)(tile_m * REC + rec) * 2 * p.N + n0 + cc;
            d[0] = s0;
            d[p.N] = q0;
        }
    }
}

// Split-K second pass: out = scale * (res + temb + bias + sum_s ws[s]) in the fixed order s = 0 .. S-1, plus the
// per-column (sum, sumsq) of every 64 output rows.  One workgroup = 64 rows x 64 columns: thread (rg = tid / 16,
// c4 = tid % 16) owns rows rg*4 .. rg*4+3 of columns c4*4 .. c4*4+3; column sums go through LDS in a fixed order.
// The split factor depends on the layer shape only (never on the batch), so results stay identical for any
// sharding of a batch.
__global__ __launch_bounds__(256) void splitk_epilogue_kernel(ConvH2Args p) {
    __shared__ float red[16][64][2];
    const int tid = threadIdx.x, rg = tid >> 4, c4 = tid & 15;
    const int row0 = blockIdx.y * 64 + rg * 4, col0 = blockIdx.x * 64 + c4 * 4;
    const int HW = p.H * p.W;
    const bool cok = col0 < p.N;              // N % 4 == 0 is required by the caller
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};        // bias / temb rows may be unaligned slices of a wider table: scalar loads
    if (cok && p.bias) bv = f32x4{p.bias[col0], p.bias[col0 + 1], p.bias[col0 + 2], p.bias[col0 + 3]};
    f32x4 cs = {0.f, 0.f, 0.f, 0.f}, cq = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = row0 + i;
        if (!cok || row >= p.M) continue;
        f32x4 v = *reinterpret_cast<const f32x4*>(p.ws + (size_t)row * p.N + col0);
        for (int s = 1; s < p.ksplit; ++s) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(p.ws + ((size_t)s * p.M + row) * p.N + col0);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] += w[j];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += bv[j];
        if (p.temb) {
            const float* t = p.temb + (size_t)(row / HW) * p.temb_stride + col0;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] += t[j];
        }
        if (p.res) {
            if (p.rfmt) {
                const dp_half4 r = *reinterpret_cast<const dp_half4*>(reinterpret_cast<const _Float16*>(p.res) + (size_t)row * p.ldr + col0);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] += (float)r[j];
            } else {
                const f32x4 r = *reinterpret_cast<const f32x4*>(p.res + (size_t)row * p.ldr + col0);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] += r[j];
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[j] *= p.scale;
            cs[j] += v[j];
            cq[j] += v[j] * v[j];
        }
        if (p.ofmt) {
            const dp_half4 h = {dp_to_half(v[0]), dp_to_half(v[1]), dp_to_half(v[2]), dp_to_half(v[3])};
            *reinterpret_cast<dp_half4*>(reinterpret_cast<_Float16*>(p.out) + (size_t)row * p.ldo + col0) = h;
        } else {
            *reinterpret_cast<f32x4*>(p.out + (size_t)row * p.ldo + col0) = v;
        }
    }
    if (!p.colstats) return;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        red[rg][c4 * 4 + j][0] = cs[j];
        red[rg][c4 * 4 + j][1] = cq[j];
    }
    __syncthreads();
    if (tid < 128) {
        const int c = tid & 63, which = tid >> 6;
        float a = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) a += red[g][c][which];
        const int col = blockIdx.x * 64 + c;
        if (col < p.N) p.colstats[((size_t)blockIdx.y * 2 + which) * p.N + col] = a;
    }
}

// fp32 [rows][cols] (row-major, ld) -> h2 [rows][cols/8][2][8]
__global__ void pack_h2_kernel(const float* src, long long rows, int cols, int ld, _Float16* dst) {
    const long long nblk = rows * (cols / 8);
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < nblk; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / (cols / 8);
        const int cb = (int)(i - r * (cols / 8));
        const float* s = src + r * ld + cb * 8;
        half8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = s[j];
            hi[j] = (_Float16)v;
            lo[j] = (_Float16)(v - (float)hi[j]);
        }
        half8* d = reinterpret_cast<half8*>(dst + i * 16);
        d[0] = hi;
        d[1] = lo;
    }
}

// 256 zero bytes on the CURRENT device (one page per device: a process may drive several GPUs from several
// threads); allocated on first use, which must therefore happen outside any stream capture.
const char* zero_page() {
    constexpr int MAXDEV = 64;
    static void* z[MAXDEV] = {};
    static std::mutex mu;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXDEV) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    if (!z[dev]) {
        if (hipMalloc(&z[dev], 256) != hipSuccess) return nullptr;
        if (hipMemset(z[dev], 0, 256) != hipSuccess) return nullptr;
    }
    return static_cast<const char*>(z[dev]);
}

}  // namespace

// Split-K factor of a layer: a function of the layer's own shape ONLY (pixels per sample, reduction length) - never of
// the batch - so that any sharding of a batch takes the same arithmetic path.  Low-resolution levels (<= 64 pixels per
// sample) have few output tiles and long reductions (K = 2304 .. 18432): without the split a 4x4 level at B=128 runs
// 128 workgroups of 144 sequential k-tiles on 256 CUs.
static int h2_ksplit_shape(int H, int W, int KS, int C, int N) {
    const int nt = KS * KS * C / 32;
    if (H * W > 64 || N % 4 != 0) return 1;
    if (H * W <= 16 && nt >= 32 && nt % 4 == 0) return 4;
    if (nt >= 16 && nt % 2 == 0) return 2;
    return 1;
}

// Round 6: the same starvation at SMALL BATCHES of the big levels - the reference's own ImageNet scripts run 4 images per GPU
// (run_scripts/imagenet/run_in_rand_inf.sh:16): the 32 x 32 / 16 x 16 / 8 x 8 levels of the guided UNet are then 64 / 32 / 8 tiles of
// 128 x 256 with 144 - 576 sequential k-tiles each, and a third of the step ran at 0.04 of the matrix peak (profiles/r05/
// closing_batch_table.md).  Such launches are split along K by a power of two chosen from the number of 128 x 256 tiles the launch has -
// i.e. per (layer shape, batch BUCKET) - until ~192 workgroups exist, keeping >= 36 k-tiles per part.  A sample's low-order bits then
// depend on the bucket its batch falls in (1e-3 parity against the reference holds in every bucket: tests/test_gpu_loops.py);
// DIFFPURE_BATCH_INVARIANT=1 restores the shape-only rule of rounds 1-5 (bit-identity across batch sizes and shardings).
static int h2_ksplit(int B, int H, int W, int KS, int C, int N) {
    const int s0 = h2_ksplit_shape(H, W, KS, C, N);
    if (dp_tune(DP_T_BATCH_INVARIANT) != 0 || N % 256 != 0 || C % 32 != 0) return s0;
    const long long M = (long long)B * H * W;
    if (M % 128 != 0) return s0;
    const long long wg = (M / 128) * (N / 256);
    const int nt = KS * KS * C / 32;
    // (the two constants were swept on the B = 4 / 8 / 16 purification - 192 / 256 / 384 workgroups x 36 / 18 k-tiles per part: within 0.3 % of each
    //  other up to 256, -1.5 % at 384: profiles/r06/ksplit_constants_sweep.log)
    int s = 1;
    while (wg * s < 192 && s < 8 && nt % (2 * s) == 0 && nt / (2 * s) >= 36) s *= 2;
    return s > s0 ? s : s0;
}

extern "C" long long dp_conv2d_nhwc_h2_workspace(int B, int H, int W, int KS, int C, int N) {
    const int s = h2_ksplit(B, H, W, KS, C, N);
    return s > 1 ? (long long)s * B * H * W * N * 4 : 0;
}

// the split factor of the shape-only rule (<= 64-pixel levels): what the fused block boundary is offered for
extern "C" int dp_conv2d_nhwc_h2_splits_by_shape(int H, int W, int KS, int C, int N) { return h2_ksplit_shape(H, W, KS, C, N) > 1; }

// May an fp16 x fp16 launch (a_fmt 1, w_fmt 1, passes 1) of this LAYER shape carry 1x1 K-segments?  A function of the layer shape
// only - never of the batch - so that any sharding of a batch takes the same arithmetic (fused or not); every fp16 x fp16 tile variant
// but the ping-pong kernel has a segment loader (the dispatcher routes around that one), and all of them give identical bits.
extern "C" int dp_conv2d_nhwc_h2_takes_segments(int H, int W, int KS, int C, int N, int segC1, int segC2) {
    if (H <= 0 || W <= 0 || (KS != 1 && KS != 3) || N <= 0 || N % 4 != 0) return 0;
    if (C <= 0 || C % 32 != 0 || segC1 <= 0 || segC1 % 32 != 0 || segC2 < 0 || segC2 % 32 != 0) return 0;
    return 1;
}

// partials_only (dp_conv2d_nhwc_h2_partials): a split-K layer stops after its partial-sum kernel - the reduction and the epilogue
// belong to the fused block boundary (boundary.hip, dp_splitk_gn)
static int h2_conv(const void* x, int C, int B, int H, int W, int KS, const void* w, int N,
                   const float* bias, const float* temb, int temb_stride, const void* res, int ldr,
                   float scale, void* out, int ldo, float* colstats, int* tile_rows, void* work,
                   long long work_bytes, int passes, int a_fmt, int w_fmt, int out_fmt, int res_fmt,
                   const void* seg1, int segC1, const void* seg2, int segC2, void* stream, bool partials_only) {
    DP_REQUIRE(x && w && (out || partials_only), "dp_conv2d_nhwc_h2: null pointer");
    DP_REQUIRE(res_fmt == 0 || (res_fmt == 1 && ldr % 2 == 0 && ((size_t)res & 3) == 0),
               "dp_conv2d_nhwc_h2: res_fmt must be 0 (fp32) or 1 (fp16; even row stride, 4-byte aligned), got %d", res_fmt);
    DP_REQUIRE((!seg1 && !seg2 && segC1 == 0 && segC2 == 0) ||
                   (seg1 && segC1 > 0 && segC1 % 32 == 0 && dp_aligned16(seg1) && ((seg2 != nullptr) == (segC2 > 0)) && segC2 % 32 == 0 &&
                    (!seg2 || dp_aligned16(seg2)) && w_fmt == 1),
               "dp_conv2d_nhwc_h2: 1x1 K-segments are plain fp16 NHWC tensors of a multiple of 32 channels (seg2 only after seg1) on "
               "the fp16 x fp16 path");
    DP_REQUIRE(KS == 1 || KS == 3, "dp_conv2d_nhwc_h2: kernel size %d unsupported", KS);
    DP_REQUIRE((a_fmt == 0 && (passes == 3 || passes == 12)) || (a_fmt == 1 && (passes == 2 || passes == 1)),
               "dp_conv2d_nhwc_h2: (a_fmt, passes) must be (0, 3 | 12) for h2 activations or (1, 2 | 1) for plain fp16 ones (got %d, %d)",
               a_fmt, passes);
    DP_REQUIRE((w_fmt == 0 && !(a_fmt == 1 && passes == 1)) || (w_fmt == 1 && a_fmt == 1 && passes == 1),
               "dp_conv2d_nhwc_h2: one pass means plain fp16 weights (w_fmt 1) on plain fp16 activations, and vice versa (got a_fmt %d, "
               "w_fmt %d, passes %d)", a_fmt, w_fmt, passes);
    DP_REQUIRE(out_fmt == 0 || (out_fmt == 1 && ldo % 2 == 0 && N % 4 == 0), "dp_conv2d_nhwc_h2: out_fmt must be 0 (fp32) or 1 (fp16; even row stride, N %% 4 == 0), got %d", out_fmt);
    DP_REQUIRE(C > 0 && C % 32 == 0, "dp_conv2d_nhwc_h2: channel count must be a multiple of 32 (got %d)", C);
    DP_REQUIRE(dp_aligned16(x) && dp_aligned16(w), "dp_conv2d_nhwc_h2: misaligned operand");
    DP_REQUIRE(B > 0 && H > 0 && W > 0 && N > 0 && (long long)B * H * W < (1ll << 31), "dp_conv2d_nhwc_h2: bad shape");
    ConvH2Args p;
    p.x = (const char*)x; p.C = C;
    p.B = B; p.H = H; p.W = W; p.KS = KS; p.pad = KS / 2;
    p.w = (const char*)w; p.bias = bias; p.temb = temb; p.temb_stride = temb_stride;
    p.res = static_cast<const float*>(res); p.ldr = ldr; p.out = static_cast<float*>(out); p.ldo = ldo;
    p.M = B * H * W; p.N = N; p.K = KS * KS * C + segC1 + segC2; p.scale = scale;
    p.rfmt = res_fmt;
    p.seg1 = static_cast<const char*>(seg1); p.seg2 = static_cast<const char*>(seg2); p.segC1 = segC1; p.segC2 = segC2;
    p.zero = zero_page();
    DP_REQUIRE(p.zero, "dp_conv2d_nhwc_h2: could not allocate the zero page");
    p.colstats = colstats;
    p.passes = passes;
    p.afmt = a_fmt;
    p.wfmt = w_fmt;
    p.ofmt = out_fmt;

    p.ksplit = h2_ksplit(B, H, W, KS, C, N);
    p.ws = static_cast<float*>(work);
    DP_REQUIRE(!partials_only || p.ksplit > 1, "dp_conv2d_nhwc_h2_partials: this layer is not reduced with split-K (dp_conv2d_nhwc_h2_workspace() == 0)");
    DP_REQUIRE(p.ksplit == 1 || (work && work_bytes >= dp_conv2d_nhwc_h2_workspace(B, H, W, KS, C, N) && dp_aligned16(work) &&
                                 ldo % 4 == 0 && (!res || (ldr % 4 == 0 && ((size_t)res & (res_fmt ? 7 : 15)) == 0))),
               "dp_conv2d_nhwc_h2: this layer is reduced with split-K and needs dp_conv2d_nhwc_h2_workspace() bytes of scratch, row "
               "strides that are multiples of 4 and a residual aligned to four of its elements (the split-K epilogue reads quads)");
    DP_REQUIRE(!colstats || tile_rows, "dp_conv2d_nhwc_h2: colstats needs tile_rows");
    hipStream_t s = static_cast<hipStream_t>(stream);
    void* rec = nullptr;
    // algorithmic HBM bytes: the activation operand once (4 or 2 bytes per element), the h2 weights once, the residual and
    // the fp32 output once
    dp_prof_begin(KS == 3 ? DP_PROF_3X3_OTHER : DP_PROF_1X1, 2.0 * p.M * (double)p.N * p.K,
                  (double)p.M * (C * (a_fmt ? 2 : 4) + 2.0 * (segC1 + segC2)) + (w_fmt ? 2.0 : 4.0) * p.K * N +
                      (double)p.M * N * ((res ? (res_fmt ? 2 : 4) : 0) + (out_fmt ? 2 : 4)), s, &rec);
    auto tiles = [&](int bm, int bn) { return (long long)((p.M + bm - 1) / bm) * ((N + bn - 1) / bn); };
#define DP_H2_LAUNCH(BM_, BN_, BK_, ABL_)                                                                  \
    do {                                                                                                   \
        p.tiles_n = (N + BN_ - 1) / BN_;                                                                   \
        p.tiles = (int)tiles(BM_, BN_);                                                                    \
        const dim3 g_((unsigned)p.tiles, (unsigned)p.ksplit);                                              \
        if (p.wfmt == 1) hipLaunchKernelGGL((conv_igemm_h2<BM_, BN_, BK_, ABL_, 1, true, true>), g_, dim3(NT), 0, s, p);                   \
        else if (p.afmt == 1) hipLaunchKernelGGL((conv_igemm_h2<BM_, BN_, BK_, ABL_, 2, true, false>), g_, dim3(NT), 0, s, p); \
        else if (p.passes == 12) hipLaunchKernelGGL((conv_igemm_h2<BM_, BN_, BK_, ABL_, 12, false, false>), g_, dim3(NT), 0, s, p);        \
        else hipLaunchKernelGGL((conv_igemm_h2<BM_, BN_, BK_, ABL_, 3, false, false>), g_, dim3(NT), 0, s, p);                             \
    } while (0)
    {   // fp16 x fp16: the 8-wave 256x256 kernel (igemm_h2_dw.hip) on launches of at least one tile per CU; DP_H2_DW = 0 never.
        // Bit-identical to the other variants.
        bool force_dw = false;
#ifdef DP_ABLATE    // probes: the 8-wave kernel on launches of ANY number of tiles (an epilogue without 255 other CUs in theirs)
        force_dw = getenv("DP_H2_DW_FORCE") != nullptr;
#endif
        if (dp_tune(DP_T_H2_DW) != 0 && dp_tune(DP_T_H2_PP) != 0 && dp_conv_dw_applies(p) && (tiles(256, 256) >= 256 || force_dw)) {
            dp_launch_conv_dw(p, s);
            dp_prof_set_kind(rec, KS == 3 ? DP_PROF_3X3_PP : DP_PROF_1X1_PP);
            if (tile_rows) *tile_rows = 64;
            dp_prof_end(rec, s);
            DP_LAUNCH_CHECK("conv_igemm_dw");
            return 0;
        }
        // round 6: layers with 128 output channels (N % 256 != 0: the 32x32 level of NCSN++) on the same kernel's 512x128 tiles, eight
        // waves stacked along the pixels, where they fill the chip (>= 256 tiles; fewer: the 4-wave kernel's 256x128 form below).
        // DP_H2_DW = 1 keeps these launches on the one-wave-per-SIMD kernel (rounds 3-5).  3x3 only, like that kernel's 512x128 form.
        if (dp_tune(DP_T_H2_DW) >= 2 && dp_tune(DP_T_H2_PP) != 0 && N % 256 != 0 && KS == 3 && tiles(512, 128) >= 256 && dp_conv_dw_applies(p, 128)) {
            dp_launch_conv_dw(p, s, 128);
            dp_prof_set_kind(rec, DP_PROF_3X3_PP);
            if (tile_rows) *tile_rows = 64;
            dp_prof_end(rec, s);
            DP_LAUNCH_CHECK("conv_igemm_dw<512x128>");
            return 0;
        }
    }
    {   // fp16 x fp16 launches that leave CUs idle on 256x256 tiles (fewer than 256 of them): the 4-wave kernel on 128x256 tiles
        // (igemm_h2_dh.hip) - twice the workgroups, up to two per CU.  From DP_H2_DH_MIN half tiles up (default 32: measured on the guided
        // UNet's low levels at B = 4 / 16 - 128: 9.62 / 17.29, 64: 9.72, 32: 9.91 / 17.37 images/s at t = 20, profiles/r05/dhmin_ab.log -
        // these launches are short of parallelism whatever the tile, but 32 workgroups of this kernel beat 512 of the 64x64 tiles);
        // smaller launches stay on the generic tiles.  DP_H2_DH = 0 never.  Bit-identical to the other variants.
        const long long t256 = tiles(256, 256);
        // below 128 workgroups the kernel is taken only for long reductions (>= 72 k-tiles per workgroup: the guided UNet's low levels at
        // small batches, 144 - 432): with 18 - 36 k-tiles per part (NCSN++ 4x4 / 8x8 levels at B <= 128) a few dozen of its workgroups lose
        // to a few hundred of the 64x64 tiles (28.6 vs 26.6 us at 4x4 B = 128; adjoint bench 142.4 vs 143.0: profiles/r05/abfinal.log)
        auto dh_enough = [&](long long wg) { return wg >= 128 || (wg >= dp_tune(DP_T_H2_DH_MIN) && (p.K / 32) / p.ksplit >= 72); };
        if (dp_tune(DP_T_H2_DH) != 0 && dp_tune(DP_T_H2_PP) != 0 && p.ksplit == 1 && dh_enough(tiles(128, 256)) && t256 < 256 && dp_conv_dh_applies(p, 256)) {
            dp_launch_conv_dh(p, s, 256);
            dp_prof_set_kind(rec, KS == 3 ? DP_PROF_3X3_DH : DP_PROF_1X1_DH);
            if (tile_rows) *tile_rows = 64;
            dp_prof_end(rec, s);
            DP_LAUNCH_CHECK("conv_igemm_dh");
            return 0;
        }
        // layers with 128 output channels (NCSN++ 32x32 level) on the kernel's 256x128 form where the 512x128 one-wave-per-SIMD tiles do
        // not fill the chip (fewer than 256 of them: B < 128 at 32x32; measured at B = 64: 429 -> 625 TFLOP/s).  On launches that do fill
        // it the two forms are within +8 / -3 % of each other per shape and indistinguishable on the purification (720.6 vs 720.0
        // images/s at t = 20, profiles/r05/dh128_ab.log), so the one-wave-per-SIMD tiles keep those; DP_H2_DH = 3 forces this form (probes)
        if (dp_tune(DP_T_H2_DH) >= 2 && dp_tune(DP_T_H2_PP) != 0 && p.ksplit == 1 && N % 256 != 0 && KS == 3 && dh_enough(tiles(256, 128)) &&
            (tiles(512, 128) < 256 || dp_tune(DP_T_H2_DH) >= 3) && dp_conv_dh_applies(p, 128)) {
            dp_launch_conv_dh(p, s, 128);
            dp_prof_set_kind(rec, DP_PROF_3X3_DH);
            if (tile_rows) *tile_rows = 64;
            dp_prof_end(rec, s);
            DP_LAUNCH_CHECK("conv_igemm_dh<256>");
            return 0;
        }
        // split-K levels (<= 64 pixels per sample): the same kernel, one part per grid.y (DP_H2_DH = 1: never; 2, the default: yes); the
        // reduction + epilogue kernel is the one the generic tiles use
        if (dp_tune(DP_T_H2_DH) >= 2 && dp_tune(DP_T_H2_PP) != 0 && p.ksplit > 1 && dh_enough(tiles(128, 256) * p.ksplit) && dp_conv_dh_applies(p, 256)) {
            dp_launch_conv_dh(p, s, 256);
            DP_LAUNCH_CHECK("conv_igemm_dh (split-K)");
            if (!partials_only) hipLaunchKernelGGL(splitk_epilogue_kernel, dim3((unsigned)((N + 63) / 64), (unsigned)((p.M + 63) / 64)), dim3(256), 0, s, p);
            if (tile_rows) *tile_rows = 64;
            dp_prof_end(rec, s);
            DP_LAUNCH_CHECK("splitk_epilogue");
            return 0;
        }
    }
    // 256x256 ping-pong variant (igemm_h2_pp.hip): DP_H2_PP = 0 never, 1 whenever the shape allows, 2 when it
    // also fills the chip (>= one tile per CU); default 2 (dp_tune.h: read once; probes flip it with dp_set_tuning).
    {
        const int pp = dp_tune(DP_T_H2_PP);
        // one workgroup per CU: the grid runs in rounds of 256 tiles; take the variant when the last round is not
        // mostly empty (measured at B=16: 407-450 TFLOP/s vs 326-375 on full rounds, 262 vs 350 on half a round)
        // (round 4: a single round on at least half the CUs is taken too - the 16x16 level of NCSN++ at B = 128, 128 tiles: +1.7 % on the
        // adjoint bench against the 128x128 tiles, profiles/r04/adjoint_pp_fill_rule_ab.log)
        auto fills = [&](int bm, int bn) {
            const long long t = tiles(bm, bn), rounds = (t + 255) / 256;
            return (t >= 256 && t * 5 >= rounds * 256 * 4) || (t >= 128 && t < 256);
        };
        int bn = 0;
        if (p.ksplit > 1) bn = 0;            // split-K layers never take the ping-pong variants (shape-only rule)
        else if (pp != 0 && p.M % 256 == 0 && N % 256 == 0 && (pp == 1 || fills(256, 256))) bn = 256;
        else if (pp != 0 && p.M % 512 == 0 && N % 128 == 0 && (pp == 1 || fills(512, 128))) bn = 128;
        const bool use_sw = bn && dp_tune(DP_T_H2_SW) != 0 && (bn == 256 || (dp_tune(DP_T_H2_SW) >= 2 && KS == 3)) && dp_conv_sw_applies(p, bn);
        if (seg1 && !use_sw) bn = 0;         // 1x1 K-segments: the ping-pong kernel has no segment loader - the generic tiles take the launch
        if (bn) {
            // fp16 x fp16, N % 256 == 0: the one-wave-per-SIMD software-pipelined kernel (igemm_h2_sw.hip) - measured
            // fastest on every shape of both networks (tests/probes/pp_ablate.py --w16); DP_H2_SW=0 falls back
            // (its 512x128 form only for 3x3 layers: on 1x1 layers with 128 output channels it measured 5 % slower than the
            //  ping-pong kernel; tests/probes/n128_probe.py)
            if (use_sw) dp_launch_conv_sw(p, s, bn);
            else dp_launch_conv_h2_pp(p, s, bn);
            dp_prof_set_kind(rec, KS == 3 ? DP_PROF_3X3_PP : DP_PROF_1X1_PP);
            if (tile_rows) *tile_rows = 64;
            dp_prof_end(rec, s);
            DP_LAUNCH_CHECK("conv_igemm_h2_pp");
            return 0;
        }
    }
    {   // few output channels (the 6-channel head): 256 x 32 tiles over x-halo runs; DP_H2_NN=0 falls back to the generic tiles
        if (dp_tune(DP_T_H2_NN) != 0 && !seg1 && dp_conv_nn_applies(p)) {
            dp_launch_conv_nn(p, s);
            dp_prof_end(rec, s);
            DP_LAUNCH_CHECK("conv_igemm_nn");
            return 0;
        }
    }
    // (thresholds 256 / 512 / 1024, a <128,64,32> middle variant and wide 4-wave tiles <128,256,16> / <256,128,16> were
    //  tried on these shapes: all within run-to-run noise or slower; the wide ones are superseded by igemm_h2_pp.hip)
    if (N <= 64 || tiles(128, 128) * p.ksplit < 256) DP_H2_LAUNCH(64, 64, 32, 0);
    else DP_H2_LAUNCH(128, 128, 32, 0);
#undef DP_H2_LAUNCH
    if (p.ksplit > 1 && !partials_only)
        hipLaunchKernelGGL(splitk_epilogue_kernel, dim3((unsigned)((N + 63) / 64), (unsigned)((p.M + 63) / 64)), dim3(256), 0, s, p);
    if (tile_rows) *tile_rows = 64;   // column-sum records are per 64 output rows in every variant
    dp_prof_end(rec, s);
    DP_LAUNCH_CHECK("conv_igemm_h2");
    return 0;
}

extern "C" int dp_conv2d_nhwc_h2(const void* x, int C, int B, int H, int W, int KS, const void* w, int N,
                                 const float* bias, const float* temb, int temb_stride, const void* res, int ldr,
                                 float scale, void* out, int ldo, float* colstats, int* tile_rows, void* work,
                                 long long work_bytes, int passes, int a_fmt, int w_fmt, int out_fmt, int res_fmt,
                                 const void* seg1, int segC1, const void* seg2, int segC2, void* stream) {
    return h2_conv(x, C, B, H, W, KS, w, N, bias, temb, temb_stride, res, ldr, scale, out, ldo, colstats, tile_rows, work, work_bytes, passes, a_fmt,
                   w_fmt, out_fmt, res_fmt, seg1, segC1, seg2, segC2, stream, false);
}

// The split-K partial sums of a convolution and nothing else: work[s][B*H*W][N] fp32, s = 0 .. S-1 (*n_parts = S), for dp_splitk_gn to
// reduce.  Only for layers dp_conv2d_nhwc_h2_workspace() reports scratch for (<= 64 pixels per sample); same kernels, same partial
// sums as dp_conv2d_nhwc_h2 forms before its own reduction pass.
extern "C" int dp_conv2d_nhwc_h2_partials(const void* x, int C, int B, int H, int W, int KS, const void* w, int N, void* work,
                                          long long work_bytes, int passes, int a_fmt, int w_fmt, const void* seg1, int segC1,
                                          const void* seg2, int segC2, int* n_parts, void* stream) {
    DP_REQUIRE(n_parts, "dp_conv2d_nhwc_h2_partials: n_parts is null");
    *n_parts = h2_ksplit(B, H, W, KS, C, N);
    // (the epilogue arguments are unused; ldo = N satisfies the split-K path's row-stride check)
    return h2_conv(x, C, B, H, W, KS, w, N, nullptr, nullptr, 0, nullptr, 0, 1.0f, nullptr, N, nullptr, nullptr, work, work_bytes, passes, a_fmt, w_fmt,
                   0, 0, seg1, segC1, seg2, segC2, stream, true);
}

extern "C" int dp_splitk_epilogue(const float* work, int n_parts, int B, int H, int W, int N, const float* bias, const float* temb,
                                  int temb_stride, const void* res, int res_fmt, float scale, void* out, int out_fmt, float* colstats,
                                  int* tile_rows, void* stream) {
    DP_REQUIRE(work && out && n_parts >= 1 && B > 0 && H > 0 && W > 0 && N > 0 && N % 4 == 0, "dp_splitk_epilogue: bad arguments");
    DP_REQUIRE((long long)B * H * W < (1ll << 31), "dp_splitk_epilogue: M overflows int32");
    DP_REQUIRE(out_fmt == 0 || out_fmt == 1, "dp_splitk_epilogue: out_fmt %d", out_fmt);
    DP_REQUIRE(res_fmt == 0 || res_fmt == 1, "dp_splitk_epilogue: res_fmt %d", res_fmt);
    DP_REQUIRE(dp_aligned16(work) && (!res || ((size_t)res & (res_fmt ? 7 : 15)) == 0) && ((size_t)out & (out_fmt ? 7 : 15)) == 0,
               "dp_splitk_epilogue: misaligned tensor (the kernel reads and writes quads)");
    DP_REQUIRE(!colstats || tile_rows, "dp_splitk_epilogue: colstats needs tile_rows");
    ConvH2Args p = {};
    p.B = B; p.H = H; p.W = W; p.M = B * H * W; p.N = N;
    p.bias = bias; p.temb = temb; p.temb_stride = temb_stride;
    p.res = static_cast<const float*>(res); p.ldr = N; p.rfmt = res_fmt;
    p.out = static_cast<float*>(out); p.ldo = N; p.ofmt = out_fmt;
    p.scale = scale; p.ksplit = n_parts; p.ws = const_cast<float*>(work); p.colstats = colstats;
    hipLaunchKernelGGL(splitk_epilogue_kernel, dim3((unsigned)((N + 63) / 64), (unsigned)((p.M + 63) / 64)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), p);
    if (tile_rows) *tile_rows = 64;
    DP_LAUNCH_CHECK("splitk_epilogue");
    return 0;
}

extern "C" int dp_pack_h2(const float* src, long long rows, int cols, int ld, void* dst, void* stream) {
    DP_REQUIRE(src && dst && rows > 0 && cols > 0 && cols % 8 == 0 && ld >= cols, "dp_pack_h2: cols must be a positive multiple of 8");
    long long g = (rows * (cols / 8) + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(pack_h2_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, src, rows, cols, ld,
                       (_Float16*)dst);
    DP_LAUNCH_CHECK("pack_h2");
    return 0;
}
