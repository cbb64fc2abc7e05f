// Implicit-GEMM convolution / linear and strided batched GEMM on the gfx950 matrix cores, fp32.
//
// Math instruction: v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate, bit-exact fmaf chain; 64 cycles
// per SIMD; chip peak 157 TFLOP/s).  A wave64 owns TM x TN tiles of 32x32 outputs; per k-pair it
// needs ONE f32 VGPR of A (lane l: A[row = l&31][k = l>>5]) and one of B (B[k = l>>5][col = l&31]).
//
// Data flow per workgroup (256 threads = 4 waves, one per SIMD):
//   HBM --coalesced float4--> registers --> LDS tiles As[k][m], Bs[k][n] (k-major so that the
//   per-lane MFMA operand reads are conflict-free ds_read_b32 over consecutive m / n)
//   --> MFMA accumulators --> fused epilogue (bias, per-sample temb row, residual, scale) --> HBM.
// Two LDS stages; the global loads of k-tile t+1 are issued before the MFMAs of k-tile t and land
// in LDS after them: one barrier per k-tile.
//
// The A operand is gathered on the fly from NHWC activations (im2col is never materialised):
// GEMM row m = output pixel (b, oy, ox); GEMM column k = ((ky*KW + kx) * Cin + ci); zero padding
// and the channel concatenation of two source tensors are resolved in the loader.
#include "dp_common.h"

// Every tile variant must produce the SAME bits, column records included (a batch's sharding picks the variant): products and sums stay
// separate IEEE operations in this file - left to itself the compiler contracts `cq += v * v` (and `v *= scale; cs += v`) into FMAs in one
// code shape and not in another.  Explicit fmaf() calls are unaffected.
#pragma clang fp contract(off)

namespace {

constexpr int BK = 16;
constexpr int NT = 256;

struct ConvArgs {
    const float* x1;
    const float* x2;
    int C1, C2;
    int B, H, W, KH, KW, pad;
    const float* w;
    int ldw;
    const float* bias;
    const float* temb;
    int temb_stride;
    const float* res;
    int ldr;
    float* out;
    int ldo;
    int M, N, K;
    float scale;
    int tiles_n;
    float* colstats;   // optional [tiles_m][2][N]: per output column, sum and sum of squares over the tile's rows
    int ofmt;          // 0: fp32 output; 1: plain fp16 output (`out` then points at fp16 elements; the column records are those of
                       // the unrounded values) - the stem of a network whose residual stream is fp16
};

template <int TM, int TN, int LDA, int LDB>
__device__ __forceinline__ void mma_ktile(const float* __restrict__ As, const float* __restrict__ Bs,
                                          int wm0, int wn0, int lane, f32x16 (&acc)[TM][TN]) {
    const int lr = lane & 31, lk = lane >> 5;
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
        float a[TM], b[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = As[(kk * 2 + lk) * LDA + wm0 + i * 32 + lr];
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = Bs[(kk * 2 + lk) * LDB + wn0 + j * 32 + lr];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
}

// ------------------------------------------------------------------------------------------------
// convolution / linear
// ------------------------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN, int VEC>
__global__ __launch_bounds__(NT) void conv_igemm_f32(ConvArgs p) {
    static_assert(WM * WN == 4, "4 waves");
    constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
    static_assert(TM >= 1 && TN >= 1, "tile");
    constexpr int LDA = BM + 4, LDB = BN + 4;
    __shared__ __attribute__((aligned(16))) float smem[2 * BK * LDA + 2 * BK * LDB];
    float* As = smem;
    float* Bs = smem + 2 * BK * LDA;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave / WN) * (TM * 32), wn0 = (wave % WN) * (TN * 32);
    const int tile_n = blockIdx.x % p.tiles_n, tile_m = blockIdx.x / p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int Cin = p.C1 + p.C2;
    const int HW = p.H * p.W;

    // ---- A loader state ----
    constexpr int A_IT = (VEC == 4) ? (BM * BK / 4 / NT) : (BM * BK / NT);
    static_assert(A_IT >= 1, "A_IT");
    int a_oy[A_IT], a_ox[A_IT], a_bH[A_IT];
    bool a_ok[A_IT];
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
        const int r = (VEC == 4) ? ((tid >> 2) + it * (NT / 4)) : ((tid >> 4) + it * (NT / 16));
        const int m = m0 + r;
        a_ok[it] = m < p.M;
        const int mm = a_ok[it] ? m : 0;
        const int b = mm / HW, rem = mm - b * HW;
        a_oy[it] = rem / p.W;
        a_ox[it] = rem - a_oy[it] * p.W;
        a_bH[it] = b * p.H;
    }
    // flattened-k cursor of this thread's column group (VEC==4 path): k = t*BK + (tid&3)*4
    int kc_ci = 0, kc_ky = 0, kc_kx = 0, kc_k = 0;
    if constexpr (VEC == 4) {
        kc_k = (tid & 3) * 4;
        int tap = kc_k / Cin;
        kc_ci = kc_k - tap * Cin;
        kc_ky = tap / p.KW;
        kc_kx = tap - kc_ky * p.KW;
    }

    f32x4 ra[(VEC == 4) ? A_IT : 1];
    float ras[(VEC == 1) ? A_IT : 1];

    auto gload_A = [&](int t) {
        if constexpr (VEC == 4) {
#pragma unroll
            for (int it = 0; it < A_IT; ++it) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                const int iy = a_oy[it] + kc_ky - p.pad, ix = a_ox[it] + kc_kx - p.pad;
                if (a_ok[it] && kc_k < p.K && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) {
                    const size_t pix = (size_t)(a_bH[it] + iy) * p.W + ix;
                    const float* src = (kc_ci < p.C1) ? (p.x1 + pix * p.C1 + kc_ci)
                                                      : (p.x2 + pix * p.C2 + (kc_ci - p.C1));
                    v = *reinterpret_cast<const f32x4*>(src);
                }
                ra[it] = v;
            }
            // advance cursor by BK
            kc_k += BK;
            kc_ci += BK;
            while (kc_ci >= Cin) {
                kc_ci -= Cin;
                if (++kc_kx == p.KW) { kc_kx = 0; ++kc_ky; }
            }
        } else {
            const int k = t * BK + (tid & 15);
            const int tap = k / Cin, ci = k - tap * Cin;
            const int ky = tap / p.KW, kx = tap - ky * p.KW;
#pragma unroll
            for (int it = 0; it < A_IT; ++it) {
                float v = 0.f;
                const int iy = a_oy[it] + ky - p.pad, ix = a_ox[it] + kx - p.pad;
                if (a_ok[it] && k < p.K && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) {
                    const size_t pix = (size_t)(a_bH[it] + iy) * p.W + ix;
                    v = (ci < p.C1) ? p.x1[pix * p.C1 + ci] : p.x2[pix * p.C2 + (ci - p.C1)];
                }
                ras[it] = v;
            }
        }
    };
    auto sstore_A = [&](int buf) {
        float* dst = As + buf * BK * LDA;
        if constexpr (VEC == 4) {
            const int kq = tid & 3;
#pragma unroll
            for (int it = 0; it < A_IT; ++it) {
                const int r = (tid >> 2) + it * (NT / 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) dst[(kq * 4 + j) * LDA + r] = ra[it][j];
            }
        } else {
            const int kc = tid & 15;
#pragma unroll
            for (int it = 0; it < A_IT; ++it) dst[kc * LDA + (tid >> 4) + it * (NT / 16)] = ras[it];
        }
    };

    // ---- B loader state: tile is BK x BN floats, float4 along n ----
    constexpr int BQ = BN / 4;                        // float4 per k-row
    constexpr int B_IT = (BK * BQ + NT - 1) / NT;     // 2 (BN=128), 1 (BN=64), 1 (BN=32, half the threads)
    f32x4 rb[B_IT];
    auto gload_B = [&](int t) {
#pragma unroll
        for (int it = 0; it < B_IT; ++it) {
            const int idx = tid + it * NT;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (idx < BK * BQ) {
                const int kr = idx / BQ, nq = idx - kr * BQ;
                const int k = t * BK + kr, n = n0 + nq * 4;
                if (k < p.K && n < p.ldw) v = *reinterpret_cast<const f32x4*>(p.w + (size_t)k * p.ldw + n);
            }
            rb[it] = v;
        }
    };
    auto sstore_B = [&](int buf) {
        float* dst = Bs + buf * BK * LDB;
#pragma unroll
        for (int it = 0; it < B_IT; ++it) {
            const int idx = tid + it * NT;
            if (idx < BK * BQ) {
                const int kr = idx / BQ, nq = idx - kr * BQ;
                *reinterpret_cast<f32x4*>(dst + kr * LDB + nq * 4) = rb[it];
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nt = (p.K + BK - 1) / BK;
    gload_A(0);
    gload_B(0);
    sstore_A(0);
    sstore_B(0);
    __syncthreads();
    for (int t = 0; t < nt; ++t) {
        const int cur = t & 1;
        if (t + 1 < nt) {
            gload_A(t + 1);
            gload_B(t + 1);
        }
        mma_ktile<TM, TN, LDA, LDB>(As + cur * BK * LDA, Bs + cur * BK * LDB, wm0, wn0, lane, acc);
        if (t + 1 < nt) {
            sstore_A(cur ^ 1);
            sstore_B(cur ^ 1);
        }
        __syncthreads();
    }

    // ---- epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    // Optionally also reduces, per output column, sum and sum-of-squares of the FINAL values over the
    // tile's rows (fixed order: registers -> half-wave swap -> wave rows through LDS): the statistics
    // pass of the GroupNorm that consumes this tensor then never re-reads it.
    const int lr = lane & 31, lk = lane >> 5;
    // Column partials are formed in a tile-shape-independent order - 16 register values per lane, the two
    // half-waves, then the two 32-row sub-tiles of a 64-row record - so that every kernel variant (and any
    // sharding of the batch, which changes the variant chosen) produces bit-identical GroupNorm statistics.
    float* cs_lds = smem;     // [BM/32][BN][2] floats; the k-loop's last barrier has released the tiles
    // residual reads of a column block issued together (TM x 16 loads in flight per lane), temb one value per
    // 32-row block when a block cannot straddle two samples - same value order as every other variant
    const float* __restrict__ resp = p.res;
    const float* __restrict__ tembp = p.temb;
    float* __restrict__ outp = p.out;
    const bool hw32 = HW % 32 == 0;
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
                    rv[i][r] = (cok && row < p.M) ? resp[(size_t)row * p.ldr + col] : 0.f;
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
                if (p.ofmt) reinterpret_cast<_Float16*>(outp)[(size_t)row * p.ldo + col] = dp_to_half(v);
                else outp[(size_t)row * p.ldo + col] = v;
                cs2[r & 1] += v;
                cq2[r & 1] += v * v;
            }
            float cs = cs2[0] + cs2[1], cq = cq2[0] + cq2[1];
            if (p.colstats) {
                cs += __shfl_xor(cs, 32, 64);
                cq += __shfl_xor(cq, 32, 64);
                if (lk == 0) {
                    float* d = cs_lds + ((((wave / WN) * TM + i) * BN) + wn0 + j * 32 + lr) * 2;
                    d[0] = cs;
                    d[1] = cq;
                }
            }
        }
    }
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

// ------------------------------------------------------------------------------------------------
// strided batched GEMM (attention cores)
// ------------------------------------------------------------------------------------------------
struct GemmArgs {
    const float* A;
    const float* B;
    float* C;
    int lda, ldb, ldc;
    long long sAb, sAh, sBb, sBh, sCb, sCh;
    int M, N, K, ZH;
    float alpha;
    int tiles_n, tiles_mn;
};

template <int BM, int BN, int WM, int WN, int TRANSB, int TRANSA>
__global__ __launch_bounds__(NT) void gemm_strided_f32(GemmArgs p) {
    constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
    constexpr int LDA = BM + 4, LDB = BN + 4;
    __shared__ __attribute__((aligned(16))) float smem[2 * BK * LDA + 2 * BK * LDB];
    float* As = smem;
    float* Bs = smem + 2 * BK * LDA;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave / WN) * (TM * 32), wn0 = (wave % WN) * (TN * 32);
    const int z = blockIdx.x / p.tiles_mn, tz = blockIdx.x - z * p.tiles_mn;
    const int tile_n = tz % p.tiles_n, tile_m = tz / p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int zb = z / p.ZH, zh = z - zb * p.ZH;
    const float* A = p.A + zb * p.sAb + zh * p.sAh;
    const float* Bm = p.B + zb * p.sBb + zh * p.sBh;
    float* C = p.C + zb * p.sCb + zh * p.sCh;

    // A tile: row-major [M][lda] (k contiguous, transposed into the k-major LDS image) or, with
    // TRANSA, stored [K][lda] (m contiguous: already k-major, copied with float4 along m)
    constexpr int AQ = BM / 4;
    constexpr int A_IT = TRANSA ? ((BK * AQ + NT - 1) / NT) : (BM * BK / 4 / NT);
    f32x4 ra[A_IT];
    auto gload_A = [&](int t) {
        if constexpr (TRANSA) {
#pragma unroll
            for (int it = 0; it < A_IT; ++it) {
                const int idx = tid + it * NT;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (idx < BK * AQ) {
                    const int kr = idx / AQ, mq = idx - kr * AQ;
                    const int k = t * BK + kr, m = m0 + mq * 4;
                    if (k < p.K && m < p.M) v = *reinterpret_cast<const f32x4*>(A + (size_t)k * p.lda + m);
                }
                ra[it] = v;
            }
        } else {
            const int k = t * BK + (tid & 3) * 4;
#pragma unroll
            for (int it = 0; it < A_IT; ++it) {
                const int m = m0 + (tid >> 2) + it * (NT / 4);
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (m < p.M && k < p.K) v = *reinterpret_cast<const f32x4*>(A + (size_t)m * p.lda + k);
                ra[it] = v;
            }
        }
    };
    auto sstore_A = [&](int buf) {
        float* dst = As + buf * BK * LDA;
        if constexpr (TRANSA) {
#pragma unroll
            for (int it = 0; it < A_IT; ++it) {
                const int idx = tid + it * NT;
                if (idx < BK * AQ) {
                    const int kr = idx / AQ, mq = idx - kr * AQ;
                    *reinterpret_cast<f32x4*>(dst + kr * LDA + mq * 4) = ra[it];
                }
            }
        } else {
            const int kq = tid & 3;
#pragma unroll
            for (int it = 0; it < A_IT; ++it) {
                const int r = (tid >> 2) + it * (NT / 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) dst[(kq * 4 + j) * LDA + r] = ra[it][j];
            }
        }
    };

    constexpr int BQ = BN / 4;
    constexpr int B_IT = TRANSB ? (BN * BK / 4 / NT) : ((BK * BQ + NT - 1) / NT);
    f32x4 rb[B_IT];
    auto gload_B = [&](int t) {
        if (TRANSB) {
            const int k = t * BK + (tid & 3) * 4;
#pragma unroll
            for (int it = 0; it < B_IT; ++it) {
                const int n = n0 + (tid >> 2) + it * (NT / 4);
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (n < p.N && k < p.K) v = *reinterpret_cast<const f32x4*>(Bm + (size_t)n * p.ldb + k);
                rb[it] = v;
            }
        } else {
#pragma unroll
            for (int it = 0; it < B_IT; ++it) {
                const int idx = tid + it * NT;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (idx < BK * BQ) {
                    const int kr = idx / BQ, nq = idx - kr * BQ;
                    const int k = t * BK + kr, n = n0 + nq * 4;
                    if (k < p.K && n < p.N) v = *reinterpret_cast<const f32x4*>(Bm + (size_t)k * p.ldb + n);
                }
                rb[it] = v;
            }
        }
    };
    auto sstore_B = [&](int buf) {
        float* dst = Bs + buf * BK * LDB;
        if (TRANSB) {
            const int kq = tid & 3;
#pragma unroll
            for (int it = 0; it < B_IT; ++it) {
                const int r = (tid >> 2) + it * (NT / 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) dst[(kq * 4 + j) * LDB + r] = rb[it][j];
            }
        } else {
#pragma unroll
            for (int it = 0; it < B_IT; ++it) {
                const int idx = tid + it * NT;
                if (idx < BK * BQ) {
                    const int kr = idx / BQ, nq = idx - kr * BQ;
                    *reinterpret_cast<f32x4*>(dst + kr * LDB + nq * 4) = rb[it];
                }
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nt = (p.K + BK - 1) / BK;
    gload_A(0);
    gload_B(0);
    sstore_A(0);
    sstore_B(0);
    __syncthreads();
    for (int t = 0; t < nt; ++t) {
        const int cur = t & 1;
        if (t + 1 < nt) {
            gload_A(t + 1);
            gload_B(t + 1);
        }
        mma_ktile<TM, TN, LDA, LDB>(As + cur * BK * LDA, Bs + cur * BK * LDB, wm0, wn0, lane, acc);
        if (t + 1 < nt) {
            sstore_A(cur ^ 1);
            sstore_B(cur ^ 1);
        }
        __syncthreads();
    }

    const int lr = lane & 31, lk = lane >> 5;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + wn0 + j * 32 + lr;
        if (col >= p.N) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (row < p.M) C[(size_t)row * p.ldc + col] = acc[i][j][r] * p.alpha;
            }
    }
}

// ------------------------------------------------------------------------------------------------
// per-launch profiling with hipEvents on the launch stream
// ------------------------------------------------------------------------------------------------
constexpr int PROF_MAX = 1 << 16;
struct ProfRec {
    hipEvent_t e0, e1;
    double flop, bytes;
    int kind;  // DP_PROF_* (include/diffpure_hip.h)
};
bool g_prof_on = false;
int g_prof_n = 0;
long long g_prof_dropped = 0;
ProfRec* g_prof = nullptr;

}  // namespace

void dp_prof_begin(int kind, double flop, double bytes, hipStream_t s, void** rec_out) {
    *rec_out = nullptr;
    if (!g_prof_on) return;
    if (g_prof_n >= PROF_MAX) {      // never silently: dp_prof_collect reports the count
        ++g_prof_dropped;
        return;
    }
    ProfRec* rec = &g_prof[g_prof_n++];
    rec->flop = flop;
    rec->bytes = bytes;
    rec->kind = kind;
    (void)hipEventRecord(rec->e0, s);
    *rec_out = rec;
}
void dp_prof_end(void* rec, hipStream_t s) {
    if (rec) (void)hipEventRecord(static_cast<ProfRec*>(rec)->e1, s);
}
void dp_prof_set_kind(void* rec, int kind) {
    if (rec) static_cast<ProfRec*>(rec)->kind = kind;
}

extern "C" int dp_prof_enable(int on) {
    if (on && !g_prof) {
        g_prof = new ProfRec[PROF_MAX];
        for (int i = 0; i < PROF_MAX; ++i) {
            if (hipEventCreate(&g_prof[i].e0) != hipSuccess || hipEventCreate(&g_prof[i].e1) != hipSuccess) {
                dp_set_error("hipEventCreate failed");
                return 1;
            }
        }
    }
    if (on) {                 // a new recording window; dp_prof_enable(0) only stops recording (records are kept for collect)
        g_prof_n = 0;
        g_prof_dropped = 0;
    }
    g_prof_on = on != 0;
    return 0;
}

extern "C" int dp_prof_collect(double* ms, long long* n, double* flop, double* bytes, long long* dropped) {
    for (int k = 0; k < DP_PROF_KINDS; ++k) {
        ms[k] = 0.0;
        flop[k] = 0.0;
        bytes[k] = 0.0;
        n[k] = 0;
    }
    for (int i = 0; i < g_prof_n; ++i) {
        if (hipEventSynchronize(g_prof[i].e1) != hipSuccess) {
            dp_set_error("hipEventSynchronize failed");
            return 1;
        }
        float t = 0.f;
        if (hipEventElapsedTime(&t, g_prof[i].e0, g_prof[i].e1) != hipSuccess) {
            dp_set_error("hipEventElapsedTime failed");
            return 1;
        }
        const int k = g_prof[i].kind;
        if (k < 0 || k >= DP_PROF_KINDS) continue;
        ms[k] += t;
        n[k] += 1;
        flop[k] += g_prof[i].flop;
        bytes[k] += g_prof[i].bytes;
    }
    if (dropped) *dropped = g_prof_dropped;
    g_prof_n = 0;
    g_prof_dropped = 0;
    return 0;
}

extern "C" int dp_conv2d_nhwc(const float* x1, int C1, const float* x2, int C2, int B, int H, int W, int KH,
                              int KW, const float* w, int ldw, int N, const float* bias, const float* temb,
                              int temb_stride, const float* res, int ldr, float scale, void* out, int ldo,
                              int out_fmt, float* colstats, int* tile_rows, void* stream) {
    DP_REQUIRE(out_fmt == 0 || out_fmt == 1, "dp_conv2d_nhwc: out_fmt %d (0 = fp32, 1 = plain fp16)", out_fmt);
    DP_REQUIRE(x1 && w && out, "dp_conv2d_nhwc: null pointer");
    DP_REQUIRE(KH == KW && (KH == 1 || KH == 3), "dp_conv2d_nhwc: kernel %dx%d unsupported", KH, KW);
    DP_REQUIRE(C1 > 0 && C2 >= 0 && (C2 == 0 || x2), "dp_conv2d_nhwc: bad channel split %d+%d", C1, C2);
    DP_REQUIRE(ldw % 4 == 0 && ldw >= N && dp_aligned16(w), "dp_conv2d_nhwc: weight panel must be 16B aligned, ldw%%4==0");
    DP_REQUIRE(B > 0 && H > 0 && W > 0 && N > 0, "dp_conv2d_nhwc: bad shape");
    DP_REQUIRE((long long)B * H * W < (1ll << 31), "dp_conv2d_nhwc: M overflows int32");
    ConvArgs p;
    p.x1 = x1; p.x2 = x2; p.C1 = C1; p.C2 = C2;
    p.B = B; p.H = H; p.W = W; p.KH = KH; p.KW = KW; p.pad = KH / 2;
    p.w = w; p.ldw = ldw; p.bias = bias; p.temb = temb; p.temb_stride = temb_stride;
    p.res = res; p.ldr = ldr; p.out = static_cast<float*>(out); p.ldo = ldo; p.ofmt = out_fmt;
    p.M = B * H * W; p.N = N; p.K = KH * KW * (C1 + C2);
    p.scale = scale;
    p.colstats = colstats;
    DP_REQUIRE(!colstats || tile_rows, "dp_conv2d_nhwc: colstats needs tile_rows");
    const bool vec = (C1 % 4 == 0) && (C2 % 4 == 0) && dp_aligned16(x1) && (C2 == 0 || dp_aligned16(x2));
    hipStream_t s = static_cast<hipStream_t>(stream);

    void* rec = nullptr;
    // algorithmic HBM bytes: every operand once (activations, weights, residual) + the output once
    dp_prof_begin(KH == 3 ? DP_PROF_3X3_OTHER : DP_PROF_1X1, 2.0 * p.M * (double)p.N * p.K,
                  4.0 * ((double)p.M * (C1 + C2) + (double)p.K * N + (double)p.M * N * (res ? 2 : 1)), s, &rec);
    auto tiles = [&](int bm, int bn) { return (long long)((p.M + bm - 1) / bm) * ((N + bn - 1) / bn); };
    int bm = 128;
    if (!vec) {
        p.tiles_n = (N + 127) / 128;
        hipLaunchKernelGGL((conv_igemm_f32<128, 128, 2, 2, 1>), dim3((unsigned)tiles(128, 128)), dim3(NT), 0, s, p);
    } else if (N <= 32) {
        p.tiles_n = 1;
        hipLaunchKernelGGL((conv_igemm_f32<128, 32, 4, 1, 4>), dim3((unsigned)tiles(128, 32)), dim3(NT), 0, s, p);
    } else if (N <= 64 || tiles(128, 128) < 384) {
        p.tiles_n = (N + 63) / 64;
        bm = 64;
        hipLaunchKernelGGL((conv_igemm_f32<64, 64, 2, 2, 4>), dim3((unsigned)tiles(64, 64)), dim3(NT), 0, s, p);
    } else {
        p.tiles_n = (N + 127) / 128;
        hipLaunchKernelGGL((conv_igemm_f32<128, 128, 2, 2, 4>), dim3((unsigned)tiles(128, 128)), dim3(NT), 0, s, p);
    }
    (void)bm;
    if (tile_rows) *tile_rows = 64;   // records are always per 64 output rows, whatever the tile
    dp_prof_end(rec, s);
    DP_LAUNCH_CHECK("conv_igemm_f32");
    return 0;
}

extern "C" int dp_gemm_strided(const float* A, int lda, long long sAb, long long sAh, int transA, const float* B,
                               int ldb, long long sBb, long long sBh, int transB, float* C, int ldc, long long sCb,
                               long long sCh, int M, int N, int K, int ZB, int ZH, float alpha, void* stream) {
    DP_REQUIRE(A && B && C, "dp_gemm_strided: null pointer");
    DP_REQUIRE(lda % 4 == 0 && ldb % 4 == 0, "dp_gemm_strided: lda, ldb must be multiples of 4");
    DP_REQUIRE(transA ? M % 4 == 0 : K % 4 == 0, "dp_gemm_strided: the contiguous extent of A must be a multiple of 4");
    DP_REQUIRE(transB ? K % 4 == 0 : N % 4 == 0, "dp_gemm_strided: the contiguous extent of B must be a multiple of 4");
    DP_REQUIRE(sAb % 4 == 0 && sAh % 4 == 0 && sBb % 4 == 0 && sBh % 4 == 0, "dp_gemm_strided: batch strides must be multiples of 4");
    DP_REQUIRE(dp_aligned16(A) && dp_aligned16(B), "dp_gemm_strided: operands must be 16B aligned");
    GemmArgs p;
    p.A = A; p.B = B; p.C = C; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.sAb = sAb; p.sAh = sAh; p.sBb = sBb; p.sBh = sBh; p.sCb = sCb; p.sCh = sCh;
    p.M = M; p.N = N; p.K = K; p.ZH = ZH; p.alpha = alpha;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long long Z = (long long)ZB * ZH;
    const bool big = M >= 128 && N >= 128;
    const int bm = big ? 128 : 64;
    p.tiles_n = (N + bm - 1) / bm;
    p.tiles_mn = p.tiles_n * ((M + bm - 1) / bm);
    const long long grid = Z * p.tiles_mn;
    DP_REQUIRE(grid < (1ll << 31), "dp_gemm_strided: grid too large");
    const dim3 g((unsigned)grid), b(NT);
#define DP_GEMM(BM_, TB_, TA_) hipLaunchKernelGGL((gemm_strided_f32<BM_, BM_, 2, 2, TB_, TA_>), g, b, 0, s, p)
    if (big) {
        if (transA) { if (transB) DP_GEMM(128, 1, 1); else DP_GEMM(128, 0, 1); }
        else { if (transB) DP_GEMM(128, 1, 0); else DP_GEMM(128, 0, 0); }
    } else {
        if (transA) { if (transB) DP_GEMM(64, 1, 1); else DP_GEMM(64, 0, 1); }
        else { if (transB) DP_GEMM(64, 1, 0); else DP_GEMM(64, 0, 0); }
    }
#undef DP_GEMM
    DP_LAUNCH_CHECK("gemm_strided_f32");
    return 0;
}
