// The stem of a score network: a 3x3 convolution of the few-channel SDE state into the residual stream
// (guided_diffusion/unet.py:478-484: 3 -> 256 at 256^2; score_sde/models/ncsnpp.py:232-236: 3 -> 128 at 32^2).
//
// K = 9 * Cin = 27 is ONE 32-wide k-tile of the fp16 matrix cores, so the layer is bound by its OUTPUT: 2.15 GB of fp16 at
// B = 64 against 50 MB of input - a write-bound kernel (the generic fp32-MFMA tiles it replaces ran the K = 27 reduction as
// 14 k-pairs of v_mfma_f32_32x32x2_f32 behind scalar gathers: 2.3 ms per launch = 0.95 TB/s, 1.5 % of the headline step).
//
// Arithmetic: the state is fp32 and stays fp32-class - x = hi + lo and w = hi + lo as fp16 pairs (22 significant bits each), three
// v_mfma_f32_32x32x16_f16 passes (x_lo w_hi + x_hi w_lo + x_hi w_hi) into one fp32 accumulator, the arithmetic of "f16x3"
// (igemm_h2.hip); the matrix work of all three passes is 0.1 ms of a 0.5 ms launch.
//
// One wave owns 64 consecutive output pixels (one 64-row column record).  Lane = pixel: it gathers its 3 x 3 x Cin patch from the
// fp32 NHWC state (L2-resident; zero padding resolved per tap - no bordered copy), splits every value into its (hi, lo) pair and
// lays the two 64-byte rows into the wave's PRIVATE patch of LDS; the A fragments are then ds_read_b128 like the weights'.  The
// wave runs over the output channels in halves of 128 columns (2 x 4 MFMA tiles = the wave tile of the 256-wide convolution
// kernels), so the epilogue - bias, column records, paired fp16 stores - IS theirs (igemm_sw_common.h::sw_epilogue): same
// record order, same rounding.  The (hi | lo) weight panel, 2 x N x 64 bytes, sits in LDS in the XOR-swizzled 64-byte rows of
// the other kernels; a workgroup (4 waves) keeps it for STEM_TPB tiles of 256 pixels.
#include "igemm_sw_common.h"

#pragma clang fp contract(off)

namespace {

constexpr int STEM_NT = 256;
constexpr int STEM_TPB = 4;        // 256-pixel tiles per workgroup (the 32 KB weight image is loaded once per workgroup)
constexpr int STEM_NMAX = 256;

__device__ __forceinline__ int stem_swz64(int row, int slot) { return row * 64 + ((slot ^ ((row >> 2) & 3)) << 4); }

template <int CIN, bool OUT16>
__global__ __launch_bounds__(STEM_NT, 2) void conv_stem_h16(ConvH2Args p) {
    static_assert(9 * CIN <= 32, "one k-tile");
    __shared__ __attribute__((aligned(16))) char wl[2 * STEM_NMAX * 64];     // weights: hi rows, then lo rows
    __shared__ __attribute__((aligned(16))) char al_[4 * 2 * 64 * 64];       // per wave: 64 patch rows hi, then lo
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 31, lk = lane >> 5;
    // weight image: panel [2][N][32] fp16 (hi rows, then lo rows), 16-byte slot (n, s) -> swizzled LDS row n of its half
    for (int idx = tid; idx < 2 * p.N * 4; idx += STEM_NT) {
        const int half = idx / (p.N * 4), rem = idx - half * p.N * 4;
        const int n = rem >> 2, s = rem & 3;
        *reinterpret_cast<f32x4*>(wl + half * p.N * 64 + stem_swz64(n, s)) = *reinterpret_cast<const f32x4*>(p.w + (size_t)idx * 16);
    }
    __syncthreads();
    const float* __restrict__ x = reinterpret_cast<const float*>(p.x);
    const int HW = p.H * p.W;
    char* aw = al_ + wave * (2 * 64 * 64);
    for (int tt = 0; tt < STEM_TPB; ++tt) {
        const int m0 = (blockIdx.x * STEM_TPB + tt) * 256 + wave * 64;
        if (m0 >= p.M) break;                            // (wave-uniform; M % 64 == 0)
        {   // ---- the patch of pixel m0 + lane ----
            const int m = m0 + lane;
            const int b = m / HW, rem = m - b * HW;
            const int oy = rem / p.W, ox = rem - oy * p.W;
            const float* xc = x + (size_t)m * CIN;       // channel 0 of the centre pixel
            const bool top = oy > 0, bot = oy < p.H - 1, lft = ox > 0, rgt = ox < p.W - 1;
            float v[32];
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int ky = t / 3, kx = t % 3;
                const bool ok = (ky != 0 || top) && (ky != 2 || bot) && (kx != 0 || lft) && (kx != 2 || rgt);
                const float* src = xc + (ok ? ((ky - 1) * p.W + (kx - 1)) * CIN : 0);
#pragma unroll
                for (int c = 0; c < CIN; ++c) {
                    const float u = src[c];
                    v[t * CIN + c] = ok ? u : 0.f;
                }
            }
#pragma unroll
            for (int e = 9 * CIN; e < 32; ++e) v[e] = 0.f;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                half8 h, l;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    h[j] = (_Float16)v[s * 8 + j];
                    l[j] = (_Float16)(v[s * 8 + j] - (float)h[j]);
                }
                const int o = stem_swz64(lane, s);
                *reinterpret_cast<half8*>(aw + o) = h;
                *reinterpret_cast<half8*>(aw + 4096 + o) = l;
            }
        }
        for (int n0 = 0; n0 < p.N; n0 += 128) {
            f32x16 acc[2][4];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                half8 ah[2], al[2], bh[4], bl[4];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int o = stem_swz64(i * 32 + lr, s * 2 + lk);
                    ah[i] = *reinterpret_cast<const half8*>(aw + o);
                    al[i] = *reinterpret_cast<const half8*>(aw + 4096 + o);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int o = stem_swz64(n0 + j * 32 + lr, s * 2 + lk);
                    bh[j] = *reinterpret_cast<const half8*>(wl + o);
                    bl[j] = *reinterpret_cast<const half8*>(wl + p.N * 64 + o);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
            }
            sw_epilogue<OUT16, 1, 0>(p, acc, m0, n0, m0 >> 6, lr, lk, HW, nullptr);
        }
    }
}

}  // namespace

extern "C" int dp_conv2d_stem_ok(int Cin, int B, int H, int W, int N) {
    return Cin == 3 && N >= 128 && N % 128 == 0 && N <= STEM_NMAX && B > 0 && H > 0 && W > 0 &&
           ((long long)B * H * W) % 64 == 0 && (long long)B * H * W * (N > Cin ? N : Cin) < (1ll << 31);
}

extern "C" int dp_conv2d_stem(const float* x, int Cin, int B, int H, int W, const void* w, int N, const float* bias, void* out,
                              int out_fmt, float* colstats, int* tile_rows, void* stream) {
    DP_REQUIRE(x && w && out, "dp_conv2d_stem: null pointer");
    DP_REQUIRE(out_fmt == 0 || out_fmt == 1, "dp_conv2d_stem: out_fmt %d (0 = fp32, 1 = plain fp16)", out_fmt);
    DP_REQUIRE(dp_conv2d_stem_ok(Cin, B, H, W, N), "dp_conv2d_stem: shape Cin=%d B=%d H=%d W=%d N=%d is not served (dp_conv2d_stem_ok)", Cin, B, H,
               W, N);
    DP_REQUIRE(dp_aligned16(w), "dp_conv2d_stem: the weight panel must be 16-byte aligned");
    DP_REQUIRE(!colstats || tile_rows, "dp_conv2d_stem: colstats needs tile_rows");
    ConvH2Args p = {};
    p.x = reinterpret_cast<const char*>(x);
    p.C = Cin; p.B = B; p.H = H; p.W = W; p.KS = 3; p.pad = 1;
    p.w = static_cast<const char*>(w);
    p.bias = bias; p.temb = nullptr; p.temb_stride = 0; p.res = nullptr; p.ldr = 0;
    p.out = static_cast<float*>(out); p.ldo = N;
    p.M = B * H * W; p.N = N; p.K = 32;
    p.scale = 1.0f;
    p.colstats = colstats;
    p.ofmt = out_fmt; p.rfmt = 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    void* rec = nullptr;
    dp_prof_begin(DP_PROF_3X3_OTHER, 2.0 * p.M * (double)N * 9 * Cin,
                  4.0 * (double)p.M * Cin + 4.0 * 32 * N + (double)p.M * N * (out_fmt ? 2 : 4), s, &rec);
    const int tiles = (p.M + 255) / 256;
    const dim3 grid((unsigned)((tiles + STEM_TPB - 1) / STEM_TPB)), block(STEM_NT);
    if (out_fmt) hipLaunchKernelGGL((conv_stem_h16<3, true>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((conv_stem_h16<3, false>), grid, block, 0, s, p);
    if (tile_rows) *tile_rows = 64;
    dp_prof_end(rec, s);
    DP_LAUNCH_CHECK("conv_stem_h16");
    return 0;
}
