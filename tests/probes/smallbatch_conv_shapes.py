"""The guided UNet's 3x3 shapes at SMALL batches (B = 4: the reference's own per-GPU batch, run_scripts/imagenet/run_in_rand_inf.sh:16; 8, 16):
the launches with fewer than 128 half tiles of 128x256 ran on the generic 64x64 / 128x128 tiles (98-143 TFLOP/s in the batch table).
Per shape: the default dispatch against conv_igemm_dh taken from 64 / 32 / 16 half tiles up (DP_H2_DH_MIN).
    python tests/probes/smallbatch_conv_shapes.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from diffpure_amd import ops  # noqa: E402

DEV = "cuda:0"
# (H, Cin, Cout, count per forward) of the levels below 128^2
SHAPES = [(64, 512, 512, 8), (64, 1024, 512, 2), (32, 512, 512, 9), (32, 1024, 512, 2), (32, 1536, 512, 1), (16, 1024, 1024, 8), (16, 2048, 1024, 2),
          (8, 1024, 1024, 13), (8, 2048, 1024, 3)]


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
    mins = (128, 64, 32, 16)
    for B in (4, 8, 16):
        tot = {m: 0.0 for m in mins}
        print(f"== B={B}: H Cin->Cout (x count) | half tiles of 128x256 (x split-K parts) | us, TFLOP/s with DP_H2_DH_MIN = " + " / ".join(str(m) for m in mins))
        for (H, ci, co, cnt) in SHAPES:
            x = torch.randn(B, H, H, ci)
            w = torch.randn(co, ci, 3, 3) * (1.0 / (9 * ci)) ** 0.5
            wh = ops.order_conv_weight_w16(w).half().to(DEV)
            xh = torch.nn.functional.pad(x, (0, 0, 1, 1, 1, 1)).half().contiguous().to(DEV)
            bias = torch.randn(co, device=DEV)
            r16 = torch.randn(B, H, H, co, device=DEV).half()
            flop = 2.0 * B * H * H * co * 9 * ci
            fn = lambda: ops.conv2d_h2(xh, wh, co, 3, bias=bias, res=r16, colstats=True, w_fmt=1, out_f16=True)
            base = None
            line = f"{H:3d} {ci:4d}->{co:4d} (x{cnt:2d}) | {B * H * H // 128 * (co // 256):4d} |"
            for m in mins:
                with ops.tuning(DP_H2_DH_MIN=m):
                    y = fn()
                    if base is None:
                        base = y
                    ok = torch.equal(y.t, base.t) and torch.equal(y.cols.buf, base.cols.buf)
                    t = timeit(fn, 20)
                tot[m] += t * cnt
                line += f" {t * 1e3:7.1f} {flop / t / 1e9:5.0f}{'' if ok else ' [DIFF]'} |"
            print(line, flush=True)
        print(f"-- B={B} weighted over one forward (these levels): " + " / ".join(f"{tot[m]:.2f} ms" for m in mins))


if __name__ == "__main__":
    main()
