// Argument block shared by the f16x3 implicit-GEMM convolution kernels (igemm_h2.hip, igemm_h2_pp.hip).
#pragma once
#include "dp_common.h"

struct ConvH2Args {
    const char* x;      // [B][H+2][W+2][C] h2 (afmt 0) or plain fp16 (afmt 1), zero border
    int C;
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
    int tiles_n, tiles;
    const char* zero;   // >= 128 zero bytes in device memory (weight rows n >= N)
    int ksplit;         // > 1: split-K - blockIdx.y handles k-tiles [y, y+1) * nt / ksplit and stores RAW partial sums
    float* ws;          //      into ws[y][M][N]; dp_conv2d_nhwc_h2 then runs the reduction + epilogue kernel
    float* colstats;    // optional [M/64][2][N] per-column (sum, sumsq) of the final values (see igemm.hip)
    int passes;         // MFMA passes per product: 3 = a_lo*w_hi + a_hi*w_lo + a_hi*w_hi ("f16x3"); 2 = a_hi*w_lo + a_hi*w_hi
                        // (activations rounded to fp16, weights to 22 bits); 12 = a_lo*w_hi + a_hi*w_hi (weights rounded);
                        // 1 = a_hi*w_hi (plain fp16 operands, fp32 accumulation)
    int wfmt;           // weight panel: 0 = h2 (hi|lo), 1 = plain fp16 (afmt 1, passes 1 only)
    int afmt;           // activation operand: 0 = h2 ([..][C/8][hi 8|lo 8] fp16, passes 3 | 12), 1 = h1 (plain fp16, passes 2 | 1)
    int ofmt;           // output: 0 = fp32 [M][ldo]; 1 = plain fp16 [M][ldo] (the final fp32 value rounded to nearest; `out` then
                        // points at fp16 elements).  Column statistics are those of the UNROUNDED values in both cases.
    int rfmt;           // residual: 0 = fp32 [M][ldr]; 1 = plain fp16 [M][ldr] (`res` then points at fp16 elements) - the fp16
                        // residual stream of the fp16 x fp16 modes (the reference's own `use_fp16` torso keeps h in fp16,
                        // guided_diffusion/unet.py:626-632, fp16_util.py:23-40)
    // 1x1 "skip" K-segments (fp16 x fp16 kernels: igemm_h2_dw.hip, igemm_h2_sw.hip, the generic tiles incl. split-K): after the KS*KS*C reduction over `x` the k-loop runs on
    // over the channels of up to two PLAIN fp16 NHWC tensors [B][H][W][Cs] (no border: a 1x1 tap never leaves the image) whose
    // weight columns follow in the same panel: out += [seg1 | seg2] . W[:, KS*KS*C :].  K counts all of it.  This is the 1x1
    // skip_connection of a ResBlock (unet.py:223-230, 262-264; layerspp.py:268-272) folded into its second 3x3 convolution.
    const char* seg1;
    const char* seg2;
    int segC1, segC2;
};

// residual value of output element (row, col) in the format p.rfmt names (generic per-element path)
__device__ __forceinline__ float dp_conv_res(const ConvH2Args& p, size_t row, int col) {
    return p.rfmt ? (float)reinterpret_cast<const _Float16*>(p.res)[row * p.ldr + col] : p.res[row * p.ldr + col];
}

typedef _Float16 dp_half2 __attribute__((ext_vector_type(2)));
typedef _Float16 dp_half4 __attribute__((ext_vector_type(4)));

// one output element in the format p.ofmt names (the generic, one-element-per-lane path of the tile variants)
__device__ __forceinline__ void dp_conv_store(const ConvH2Args& p, size_t row, int col, float v) {
    if (p.ofmt) reinterpret_cast<_Float16*>(p.out)[row * p.ldo + col] = dp_to_half(v);
    else p.out[row * p.ldo + col] = v;
}

// 8-wave "ping-pong" variants (igemm_h2_pp.hip): bn = 256 -> 256x256 tiles (needs M % 256 == 0, N % 256 == 0),
// bn = 128 -> 512x128 tiles (M % 512 == 0, N % 128 == 0); C % 32 == 0.  Fills p.tiles / p.tiles_n itself.
void dp_launch_conv_h2_pp(ConvH2Args& p, hipStream_t s, int bn);

// One-wave-per-SIMD software-pipelined variant (igemm_h2_sw.hip): fp16 x fp16, 256x256 tile, 4 waves of 128x128.
// bn = 256: 256x256 tiles (M % 256 == 0, N % 256 == 0); bn = 128: 512x128 tiles (M % 512 == 0, N % 128 == 0).
bool dp_conv_sw_applies(const ConvH2Args& p, int bn);
void dp_launch_conv_sw(ConvH2Args& p, hipStream_t s, int bn);

// One 8-wave workgroup per CU, two free-running waves per SIMD sharing the tile (igemm_h2_dw.hip): fp16 x fp16; bn = 256: 256x256 tiles
// (M % 256 == 0, N % 256 == 0), bn = 128: 512x128 tiles (M % 512 == 0, N % 128 == 0 - layers with 128 output channels); the launcher
// fills p.tiles.
bool dp_conv_dw_applies(const ConvH2Args& p, int bn = 256);
void dp_launch_conv_dw(ConvH2Args& p, hipStream_t s, int bn = 256);

// The same wave tiles on 128x256 tiles, four waves per workgroup, two workgroups per CU (igemm_h2_dh.hip): launches that do not fill
// the chip with 256x256 tiles; the launcher fills p.tiles.
bool dp_conv_dh_applies(const ConvH2Args& p, int bn);     // bn = 256: 128 x 256 tiles; bn = 128: 256 x 128 tiles (N % 256 != 0 layers)
void dp_launch_conv_dh(ConvH2Args& p, hipStream_t s, int bn);

// Few output channels (N <= 32: the 6-channel head), 3x3, fp16 x fp16: 256 x 32 tiles over x-halo activation runs (igemm_h2_nn.hip).
bool dp_conv_nn_applies(const ConvH2Args& p);
void dp_launch_conv_nn(ConvH2Args& p, hipStream_t s);

