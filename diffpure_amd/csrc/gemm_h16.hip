// Strided batched GEMM on the fp16 matrix cores: C[z] = alpha * op(A[z]) op(B[z]) with fp32 operands in memory that are ROUNDED TO
// fp16 (nearest) on their way into LDS, one v_mfma_f32_32x32x16_f16 pass per product, fp32 accumulation, fp32 result.
//
// Round 5: the attention BACKWARD of the fp16 x fp16 precision modes.  The input-gradient convolutions of those modes already run
// "in the arithmetic of the forward they differentiate" (one fp16 MFMA pass on fp16 operands); the five
// attention products of the adjoint step - the recomputed scores q k^T and dV = P^T dO, dP = dO V^T, dQ = dS K, dK = dS^T Q
// (score_sde/models/layerspp.py:75-91, guided_diffusion/unet.py:345-362 under autograd) - still ran on the fp32-input matrix pipe
// (157 TFLOP/s peak): 5.4 % of the CIFAR-10 adjoint benchmark (profiles/r05/cifar_adjoint_b128_10step_kernel_stats_start_of_round.csv).
// Same contract as dp_gemm_strided (igemm.hip) for the shapes this kernel serves: M % 128 == 0, N % 64 == 0, K % 32 == 0 (round 6: the
// 128 x 64 tile - dV / dQ / dK of the guided UNet's 64-wide heads, unet.py:345-362, which stayed on the fp32 pipe in round 5).
// Either operand may also be stored as PLAIN fp16 already (a_fmt / b_fmt 1: q, k, v read in place inside the fp16 qkv tensor the taped
// forward keeps since round 5 - no up-conversion pass, half the operand bytes); leading dimensions and batch strides count ELEMENTS of
// the operand's own type.
//
// Tile 128 x 128, four waves of 64 x 64 (2 x 2 MFMA tiles), or 128 x 64 with waves of 64 x 32 (N % 128 != 0), k-tile 32.  Both operand tiles live in LDS as rows of 64 bytes (32 fp16 of
// one row of op(A) / one column of op(B)) with the 16-byte slot XOR-swizzled by the row key (row >> 2) & 3 - the operand image of the
// convolution kernels (igemm_h2_sw.hip), so a fragment is one conflict-free ds_read_b128.  An operand whose reduction index is
// contiguous in memory is converted eight values at a time (one 16-byte LDS store); the transposed ones are scattered two bytes at a
// time (sixteen ds_write_b16 per thread and k-tile: small next to the 4 x 16 MFMAs of the tile at these sizes).  Registers double-buffer
// the global loads under the MFMAs of the current k-tile.
#include <hip/hip_runtime.h>

#include "dp_common.h"
#include "dp_tune.h"

namespace {

constexpr int NT = 256;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

struct GemmHArgs {
    const float* A;
    const float* B;
    float* C;
    int lda, ldb, ldc;
    long long sAb, sAh, sBb, sBh, sCb, sCh;
    int M, N, K, ZH;
    float alpha;
    int tiles_n, tiles_mn;
    int afmt, bfmt;             // 0: the operand is fp32 in memory, 1: plain fp16
    int xcd_map;                // round 6: the tiles of one batch entry on ONE XCD (they share its operand rows); speed only
};

// KCONTIG: the operand is stored [row][k] (k contiguous); else [k][row] (row contiguous).  `row` = m for A, n for B.  ROWS: rows of the tile.
template <bool KCONTIG, int ROWS>
struct Stager {
    static constexpr int ITK = ROWS / 64, ITR = ROWS / 32, RQ = ROWS / 4;
    f32x4 r[4];                 // fp32 source: the values; fp16 source: raw bits (KCONTIG: r[it] = eight halves; else r[it][0..1] = four halves)
    // `base` points at the operand's first element (float or _Float16 according to f16), ld in elements of that type
    __device__ __forceinline__ void gload(const void* base, bool f16, int ld, int row0, int k0, int tid) {
        if constexpr (KCONTIG) {
#pragma unroll
            for (int it = 0; it < ITK; ++it) {          // unit = (row, 8 consecutive k): ROWS rows x 4 slots, ROWS / 64 units per thread
                const int row = (tid >> 2) + it * 64, q = tid & 3;
                const size_t e = (size_t)(row0 + row) * ld + k0 + q * 8;
                if (f16) {
                    r[2 * it] = *reinterpret_cast<const f32x4*>(static_cast<const _Float16*>(base) + e);       // 16 bytes = 8 halves
                } else {
                    const float* s = static_cast<const float*>(base) + e;
                    r[2 * it] = *reinterpret_cast<const f32x4*>(s);
                    r[2 * it + 1] = *reinterpret_cast<const f32x4*>(s + 4);
                }
            }
        } else {
#pragma unroll
            for (int it = 0; it < ITR; ++it) {          // four values along the rows at one k: 32 k x ROWS / 4 row quads, ROWS / 32 per thread
                const int idx = tid + it * NT, kr = idx / RQ, rq = idx - kr * RQ;
                const size_t e = (size_t)(k0 + kr) * ld + row0 + rq * 4;
                if (f16) {
                    const half4 h = *reinterpret_cast<const half4*>(static_cast<const _Float16*>(base) + e);   // 8 bytes
                    r[it] = f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};                          // exact: re-rounded to the same halves below
                } else {
                    r[it] = *reinterpret_cast<const f32x4*>(static_cast<const float*>(base) + e);
                }
            }
        }
    }
    __device__ __forceinline__ void sstore(char* tile, bool f16, int tid) const {
        if constexpr (KCONTIG) {
#pragma unroll
            for (int it = 0; it < ITK; ++it) {
                const int row = (tid >> 2) + it * 64, q = tid & 3;
                char* d = tile + row * 64 + ((q ^ ((row >> 2) & 3)) << 4);
                if (f16) {
                    *reinterpret_cast<f32x4*>(d) = r[2 * it];       // already eight halves
                } else {
                    half8 h;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        h[j] = (_Float16)r[2 * it][j];
                        h[4 + j] = (_Float16)r[2 * it + 1][j];
                    }
                    *reinterpret_cast<half8*>(d) = h;
                }
            }
        } else {
#pragma unroll
            for (int it = 0; it < ITR; ++it) {
                const int idx = tid + it * NT, kr = idx / RQ, rq = idx - kr * RQ;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int row = rq * 4 + j;
                    *reinterpret_cast<_Float16*>(tile + row * 64 + (((kr >> 3) ^ ((row >> 2) & 3)) << 4) + (kr & 7) * 2) = (_Float16)r[it][j];
                }
            }
        }
    }
};

template <int TRANSA, int TRANSB, int BN>
__global__ __launch_bounds__(NT) void gemm_strided_h16(GemmHArgs p) {
    constexpr int TN = BN / 64;             // MFMA tile columns per wave: wave tile 64 x (BN / 2)
    __shared__ __attribute__((aligned(1024))) char smem[2 * 2 * 128 * 64];      // [buffer][A | B][<= 128 rows x 64 bytes]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    int bid = blockIdx.x;
    if (p.xcd_map) {            // XCD x (= blockIdx % 8) owns a contiguous range of logical tiles (the convolution kernels' map)
        const int x = bid & 7, per = gridDim.x >> 3, rem = gridDim.x & 7;
        bid = (x < rem ? x * (per + 1) : rem * (per + 1) + (x - rem) * per) + (bid >> 3);
    }
    const int z = bid / p.tiles_mn, tz = bid - z * p.tiles_mn;
    const int tile_n = tz % p.tiles_n, tile_m = tz / p.tiles_n;
    const int m0 = tile_m * 128, n0 = tile_n * BN;
    const int zb = z / p.ZH, zh = z - zb * p.ZH;
    const bool af = p.afmt != 0, bf = p.bfmt != 0;           // (workgroup-uniform) operand formats
    const void* A = af ? (const void*)(reinterpret_cast<const _Float16*>(p.A) + zb * p.sAb + zh * p.sAh) : (const void*)(p.A + zb * p.sAb + zh * p.sAh);
    const void* Bm = bf ? (const void*)(reinterpret_cast<const _Float16*>(p.B) + zb * p.sBb + zh * p.sBh) : (const void*)(p.B + zb * p.sBb + zh * p.sBh);
    float* C = p.C + zb * p.sCb + zh * p.sCh;

    // op(A)[m][k]: stored [M][lda] (k contiguous) unless TRANSA ([K][lda]);  op(B)[k][n]: stored [K][ldb] (n contiguous: NOT
    // k-contiguous) unless TRANSB ([N][ldb], k contiguous)
    Stager<!TRANSA, 128> sa;
    Stager<(TRANSB != 0), BN> sb;
    const int lr = lane & 31, lk = lane >> 5;
    int soff[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) soff[s] = ((s * 2 + lk) ^ ((lr >> 2) & 3)) << 4;
    f32x16 acc[2][TN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nt = p.K / 32;
    sa.gload(A, af, p.lda, m0, 0, tid);
    sb.gload(Bm, bf, p.ldb, n0, 0, tid);
    sa.sstore(smem, af, tid);
    sb.sstore(smem + 128 * 64, bf, tid);
    __syncthreads();
    for (int t = 0; t < nt; ++t) {
        const char* at = smem + (t & 1) * (2 * 128 * 64);
        const char* bt = at + 128 * 64;
        if (t + 1 < nt) {
            sa.gload(A, af, p.lda, m0, (t + 1) * 32, tid);
            sb.gload(Bm, bf, p.ldb, n0, (t + 1) * 32, tid);
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            half8 fa[2], fb[TN];
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const half8*>(at + (wr * 64 + i * 32 + lr) * 64 + soff[s]);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const half8*>(bt + (wc * (BN / 2) + j * 32 + lr) * 64 + soff[s]);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        if (t + 1 < nt) {
            char* nx = smem + ((t + 1) & 1) * (2 * 128 * 64);
            sa.sstore(nx, af, tid);
            sb.sstore(nx + 128 * 64, bf, tid);
        }
        __syncthreads();
    }
    // C/D layout of the 32x32 MFMA: column = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float* d = C + (size_t)(m0 + wr * 64 + i * 32 + 4 * lk) * p.ldc + n0 + wc * (BN / 2) + j * 32 + lr;
#pragma unroll
            for (int r = 0; r < 16; ++r) d[(size_t)((r & 3) + 8 * (r >> 2)) * p.ldc] = acc[i][j][r] * p.alpha;
        }
}

}  // namespace

extern "C" int dp_gemm_strided_h16_ok(int M, int N, int K) { return M > 0 && N > 0 && K > 0 && M % 128 == 0 && N % 64 == 0 && K % 32 == 0; }

extern "C" int dp_gemm_strided_h16(const void* A, int a_fmt, int lda, long long sAb, long long sAh, int transA, const void* B, int b_fmt, int ldb,
                                   long long sBb, long long sBh, int transB, float* C, int ldc, long long sCb, long long sCh, int M, int N, int K, int ZB,
                                   int ZH, float alpha, void* stream) {
    DP_REQUIRE(A && B && C, "dp_gemm_strided_h16: null pointer");
    DP_REQUIRE((a_fmt == 0 || a_fmt == 1) && (b_fmt == 0 || b_fmt == 1), "dp_gemm_strided_h16: operand formats are 0 (fp32) or 1 (plain fp16)");
    // 16-byte loads of eight halves / four floats (8-byte loads of four halves in the transposed fp16 form): strides in multiples of 8 fp16 elements
    DP_REQUIRE((!a_fmt || (lda % 8 == 0 && sAb % 8 == 0 && sAh % 8 == 0)) && (!b_fmt || (ldb % 8 == 0 && sBb % 8 == 0 && sBh % 8 == 0)),
               "dp_gemm_strided_h16: an fp16 operand needs row and batch strides that are multiples of 8 elements");
    DP_REQUIRE(dp_gemm_strided_h16_ok(M, N, K), "dp_gemm_strided_h16: needs M %% 128 == 0, N %% 64 == 0, K %% 32 == 0 (got %d, %d, %d); other shapes: dp_gemm_strided", M, N, K);
    DP_REQUIRE(lda % 4 == 0 && ldb % 4 == 0 && sAb % 4 == 0 && sAh % 4 == 0 && sBb % 4 == 0 && sBh % 4 == 0 && dp_aligned16(A) && dp_aligned16(B),
               "dp_gemm_strided_h16: operands 16-byte aligned, row and batch strides multiples of 4");
    DP_REQUIRE(ZB > 0 && ZH > 0, "dp_gemm_strided_h16: empty batch");
    GemmHArgs p;
    p.A = static_cast<const float*>(A); p.B = static_cast<const float*>(B); p.C = C; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.afmt = a_fmt; p.bfmt = b_fmt;
    p.sAb = sAb; p.sAh = sAh; p.sBb = sBb; p.sBh = sBh; p.sCb = sCb; p.sCh = sCh;
    p.M = M; p.N = N; p.K = K; p.ZH = ZH; p.alpha = alpha;
    const int bn = N % 128 == 0 ? 128 : 64;
    p.tiles_n = N / bn;
    p.tiles_mn = p.tiles_n * (M / 128);
    p.xcd_map = dp_tune(DP_T_XCD_MAP) != 0;
    const long long grid = (long long)ZB * ZH * p.tiles_mn;
    DP_REQUIRE(grid < (1ll << 31), "dp_gemm_strided_h16: grid too large");
    const dim3 g((unsigned)grid), b(NT);
    hipStream_t s = static_cast<hipStream_t>(stream);
#define DP_GEMMH(TA_, TB_)                                                                          \
    do {                                                                                            \
        if (bn == 128) hipLaunchKernelGGL((gemm_strided_h16<TA_, TB_, 128>), g, b, 0, s, p);        \
        else hipLaunchKernelGGL((gemm_strided_h16<TA_, TB_, 64>), g, b, 0, s, p);                   \
    } while (0)
    if (transA) {
        if (transB) DP_GEMMH(1, 1);
        else DP_GEMMH(1, 0);
    } else {
        if (transB) DP_GEMMH(0, 1);
        else DP_GEMMH(0, 0);
    }
#undef DP_GEMMH
    DP_LAUNCH_CHECK("gemm_strided_h16");
    return 0;
}
