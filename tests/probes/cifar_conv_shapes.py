"""Per-shape timing of the fp16 x fp16 convolution launches of the CIFAR-10 NCSN++ as the f16sr engine issues them (fp16 output,
fp16 residual, column records): which tile kernel the dispatcher picks is a function of (shape, batch), so the table is printed at
B = 256 (configs[1]) and B = 128 (configs[4]); the adjoint's taped forward / dgrad launches (fp32 output, fp32 residual) in a second
column.   python tests/probes/cifar_conv_shapes.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from diffpure_amd import ops  # noqa: E402

DEV = "cuda:0"
ALT = int(os.environ.get("PROBE_ALT_DH", "0"))     # the DP_H2_DH value of the comparison column (3: the 256x128 form on the 128-channel layers)
# (H, Cin, Cout, count per forward)
SHAPES = [(32, 128, 128, 34), (32, 256, 128, 9), (32, 384, 128, 1), (16, 256, 256, 33), (16, 128, 256, 1), (16, 512, 256, 8), (16, 384, 256, 1),
          (8, 256, 256, 34), (8, 512, 256, 9), (4, 256, 256, 38), (4, 512, 256, 9)]


def timeit(fn, iters):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    for B in (256, 128, 64):
        tot16 = tot32 = totf = old16 = old32 = 0.0
        print(f"== B={B}: H Cin->Cout (x count) | tiles of 256x256 | stream form (fp16 out + fp16 res): us, TFLOP/s | taped form (fp32 out + fp32 res): us, TFLOP/s")
        for (H, ci, co, cnt) in SHAPES:
            x = torch.randn(B, H, H, ci)
            w = torch.randn(co, ci, 3, 3) * (1.0 / (9 * ci)) ** 0.5
            wh = ops.order_conv_weight_w16(w).half().to(DEV)
            xh = torch.nn.functional.pad(x, (0, 0, 1, 1, 1, 1)).half().contiguous().to(DEV)
            bias = torch.randn(co, device=DEV)
            r32 = torch.randn(B, H, H, co, device=DEV)
            r16 = r32.half()
            flop = 2.0 * B * H * H * co * 9 * ci
            o16 = (H * H) % 64 == 0
            f16 = lambda: ops.conv2d_h2(xh, wh, co, 3, bias=bias, res=r16 if o16 else r32, scale=0.7071, colstats=True, w_fmt=1, out_f16=o16)
            f32 = lambda: ops.conv2d_h2(xh, wh, co, 3, bias=bias, res=r32, scale=0.7071, colstats=True, w_fmt=1)
            t16, t32 = timeit(f16, 30), timeit(f32, 30)
            tot16 += t16 * cnt
            tot32 += t32 * cnt
            totf += flop * cnt
            with ops.tuning(DP_H2_DH=ALT):      # round 5: the same launches without the half-height tile kernel (ALT = 0: what took them in round 4)
                o16, o32 = timeit(f16, 30), timeit(f32, 30)
            old16 += o16 * cnt
            old32 += o32 * cnt
            print(f"{H:3d} {ci:4d}->{co:3d} (x{cnt:2d}) | {B * H * H // 256 * (co // 128) // 2:5d} | {t16 * 1e3:7.1f} {flop / t16 / 1e9:6.0f} | {t32 * 1e3:7.1f} {flop / t32 / 1e9:6.0f}"
                  f" | DP_H2_DH={ALT}: {o16 * 1e3:7.1f} {flop / o16 / 1e9:6.0f} | {o32 * 1e3:7.1f} {flop / o32 / 1e9:6.0f}", flush=True)
        print(f"-- B={B} weighted over one forward: stream form {tot16:.2f} ms ({totf / tot16 / 1e9:.0f} TFLOP/s), taped form {tot32:.2f} ms ({totf / tot32 / 1e9:.0f} TFLOP/s); with DP_H2_DH={ALT}: {old16:.2f} / {old32:.2f} ms")


if __name__ == "__main__":
    main()
