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
// "h2" tensor format (written by gn_apply, csrc/norm.hip, and by the host weight packer):
//   channels in blocks of 8:  [ 8 x fp16 hi | 8 x fp16 lo ]  = 32 bytes per 8 channels,
//   i.e. 4 bytes per element like fp32, and a 32-channel k-tile of one pixel is 128 contiguous
//   bytes.  Weights are stored [N][K] with the same blocking along k = (ky*KW+kx)*Cin + ci.
//
// Tile: 128x128x32 per 256-thread workgroup (4 waves as 2x2, each 2x2 MFMA tiles of 32x32).
// LDS image per operand and stage: 128 rows x 128 B; the 16-byte slot s of row r lives at slot
// s ^ ((r>>1)&7)  -> every ds_read_b128 lane group touches 16 distinct bank slots (conflict-free),
// and every ds_write_b128 of a staged 16-byte piece likewise.  Per k16 sub-step a wave issues
// 8 ds_read_b128 for 12 MFMAs (384 matrix-pipe cycles): the loop is matrix-bound, not LDS-bound.
#include "dp_common.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int NT = 256;
constexpr int BKH = 32;          // k elements per tile
constexpr int ROWB = 128;        // bytes per LDS row (32 k x (hi,lo) fp16)

struct ConvH2Args {
    const char* x1;
    const char* x2;
    int C1, C2;
    int B, H, W, KS, pad;
    const char* w;
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
};

__device__ __forceinline__ int swz(int row, int slot) { return row * ROWB + ((slot ^ ((row >> 1) & 7)) << 4); }

template <int BM, int BN>
__global__ __launch_bounds__(NT) void conv_igemm_h2(ConvH2Args p) {
    constexpr int TM = BM / 64, TN = BN / 64;           // 2x2 waves, 32x32 MFMA tiles
    constexpr int A_IT = BM * 8 / NT, B_IT = BN * 8 / NT;
    constexpr int STAGE = (BM + BN) * ROWB;
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave >> 1) * (TM * 32), wn0 = (wave & 1) * (TN * 32);
    const int tile_n = blockIdx.x % p.tiles_n, tile_m = blockIdx.x / p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int Cin = p.C1 + p.C2, HW = p.H * p.W;
    const int slot = tid & 7, r0 = tid >> 3;            // this thread stages slot `slot` of rows r0 + 32*it

    int a_oy[A_IT], a_ox[A_IT], a_bH[A_IT];
    bool a_ok[A_IT];
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
        const int m = m0 + r0 + it * 32;
        a_ok[it] = m < p.M;
        const int mm = a_ok[it] ? m : 0;
        const int b = mm / HW, rem = mm - b * HW;
        a_oy[it] = rem / p.W;
        a_ox[it] = rem - a_oy[it] * p.W;
        a_bH[it] = b * p.H;
    }
    // workgroup-uniform k cursor: tap (ky,kx) and first channel of the current 32-channel tile
    int ci0 = 0, ky = 0, kx = 0;

    u32x4 ra[A_IT], rb[B_IT];
    auto gload = [&](int t) {
        const bool first = ci0 < p.C1;
        const char* base = first ? p.x1 : p.x2;
        const int Cs = first ? p.C1 : p.C2;
        const int cs = first ? ci0 : ci0 - p.C1;
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            u32x4 v = {0u, 0u, 0u, 0u};
            const int iy = a_oy[it] + ky - p.pad, ix = a_ox[it] + kx - p.pad;
            if (a_ok[it] && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) {
                const size_t pix = (size_t)(a_bH[it] + iy) * p.W + ix;
                v = *reinterpret_cast<const u32x4*>(base + (pix * Cs + cs) * 4 + slot * 16);
            }
            ra[it] = v;
        }
#pragma unroll
        for (int it = 0; it < B_IT; ++it) {
            u32x4 v = {0u, 0u, 0u, 0u};
            const int n = n0 + r0 + it * 32;
            if (n < p.N) v = *reinterpret_cast<const u32x4*>(p.w + ((size_t)n * p.K + (size_t)t * BKH) * 4 + slot * 16);
            rb[it] = v;
        }
        ci0 += BKH;
        if (ci0 == Cin) {
            ci0 = 0;
            if (++kx == p.KS) { kx = 0; ++ky; }
        }
    };
    auto sstore = [&](int stage) {
        char* As = smem + stage * STAGE;
        char* Bs = As + BM * ROWB;
#pragma unroll
        for (int it = 0; it < A_IT; ++it) *reinterpret_cast<u32x4*>(As + swz(r0 + it * 32, slot)) = ra[it];
#pragma unroll
        for (int it = 0; it < B_IT; ++it) *reinterpret_cast<u32x4*>(Bs + swz(r0 + it * 32, slot)) = rb[it];
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
        const char* Bs = As + BM * ROWB;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            half8 ah[TM], al[TM], bh[TN], bl[TN];
            const int sl = s * 4 + lk * 2;  // hi slot; lo slot = sl + 1
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = wm0 + i * 32 + lr;
                ah[i] = *reinterpret_cast<const half8*>(As + swz(row, sl));
                al[i] = *reinterpret_cast<const half8*>(As + swz(row, sl + 1));
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int row = wn0 + j * 32 + lr;
                bh[j] = *reinterpret_cast<const half8*>(Bs + swz(row, sl));
                bl[j] = *reinterpret_cast<const half8*>(Bs + swz(row, sl + 1));
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    };

    const int nt = p.K / BKH;
    gload(0);
    sstore(0);
    __syncthreads();
    for (int t = 0; t < nt; ++t) {
        const int cur = t & 1;
        if (t + 1 < nt) gload(t + 1);
        compute(cur);
        if (t + 1 < nt) sstore(cur ^ 1);
        __syncthreads();
    }

#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + wn0 + j * 32 + lr;
        if (col >= p.N) continue;
        const float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (row >= p.M) continue;
                float v = acc[i][j][r] + bv;
                if (p.temb) v += p.temb[(size_t)(row / HW) * p.temb_stride + col];
                if (p.res) v += p.res[(size_t)row * p.ldr + col];
                p.out[(size_t)row * p.ldo + col] = v * p.scale;
            }
        }
    }
}

// fp32 [rows][cols] (row-major, ld) -> h2 [rows][cols/8][2][8]; host-side weight / tensor packer
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

}  // namespace

extern "C" int dp_conv2d_nhwc_h2(const void* x1, int C1, const void* x2, int C2, int B, int H, int W, int KS,
                                 const void* w, int N, const float* bias, const float* temb, int temb_stride,
                                 const float* res, int ldr, float scale, float* out, int ldo, void* stream) {
    DP_REQUIRE(x1 && w && out, "dp_conv2d_nhwc_h2: null pointer");
    DP_REQUIRE(KS == 1 || KS == 3, "dp_conv2d_nhwc_h2: kernel size %d unsupported", KS);
    DP_REQUIRE(C1 > 0 && C2 >= 0 && (C2 == 0 || x2), "dp_conv2d_nhwc_h2: bad channel split %d+%d", C1, C2);
    DP_REQUIRE(C1 % 32 == 0 && C2 % 32 == 0, "dp_conv2d_nhwc_h2: channel counts must be multiples of 32 (got %d+%d)", C1, C2);
    DP_REQUIRE(dp_aligned16(x1) && dp_aligned16(w) && (C2 == 0 || dp_aligned16(x2)), "dp_conv2d_nhwc_h2: misaligned operand");
    DP_REQUIRE(B > 0 && H > 0 && W > 0 && N > 0 && (long long)B * H * W < (1ll << 31), "dp_conv2d_nhwc_h2: bad shape");
    ConvH2Args p;
    p.x1 = (const char*)x1; p.x2 = (const char*)x2; p.C1 = C1; p.C2 = C2;
    p.B = B; p.H = H; p.W = W; p.KS = KS; p.pad = KS / 2;
    p.w = (const char*)w; p.bias = bias; p.temb = temb; p.temb_stride = temb_stride;
    p.res = res; p.ldr = ldr; p.out = out; p.ldo = ldo;
    p.M = B * H * W; p.N = N; p.K = KS * KS * (C1 + C2); p.scale = scale;
    hipStream_t s = static_cast<hipStream_t>(stream);
    void* rec = nullptr;
    dp_prof_begin(KS == 3 ? 0 : 1, 2.0 * p.M * (double)p.N * p.K, s, &rec);
    auto tiles = [&](int bm, int bn) { return (long long)((p.M + bm - 1) / bm) * ((N + bn - 1) / bn); };
    if (N <= 64 || tiles(128, 128) < 256) {
        p.tiles_n = (N + 63) / 64;
        hipLaunchKernelGGL((conv_igemm_h2<64, 64>), dim3((unsigned)tiles(64, 64)), dim3(NT), 0, s, p);
    } else {
        p.tiles_n = (N + 127) / 128;
        hipLaunchKernelGGL((conv_igemm_h2<128, 128>), dim3((unsigned)tiles(128, 128)), dim3(NT), 0, s, p);
    }
    dp_prof_end(rec, s);
    DP_LAUNCH_CHECK("conv_igemm_h2");
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
