"""Layers with 128 output channels (the 32x32 level of the CIFAR-10 NCSN++, B=256): the 8-wave ping-pong kernel on 512x128 tiles
(DP_H2_SW=1) against the 512x128 form of the one-wave-per-SIMD kernel (DP_H2_SW=2); bit-identity and TFLOP/s (algorithmic).
    python tests/probes/n128_probe.py [--batch B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from diffpure_amd import ops  # noqa: E402

DEV = "cuda:0"


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
    B = int(sys.argv[sys.argv.index("--batch") + 1]) if "--batch" in sys.argv else 256
    for (H, ci, co, ks) in [(32, 128, 128, 3), (32, 256, 128, 3), (32, 128, 128, 1), (64, 128, 128, 3)]:
        x = torch.randn(B, H, H, ci)
        w = torch.randn(co, ci, ks, ks) * (1.0 / (ks * ks * ci)) ** 0.5
        wh = ops.order_conv_weight_w16(w).half().to(DEV)
        xh = torch.nn.functional.pad(x, (0, 0, 1, 1, 1, 1)).half().contiguous().to(DEV)
        bias = torch.randn(co, device=DEV)
        rs = torch.randn(B, H, H, co, device=DEV)
        flop = 2.0 * B * H * H * co * ks * ks * ci
        fns = dict(plain=lambda: ops.conv2d_h2(xh, wh, co, ks, bias=bias, colstats=True, w_fmt=1),
                   res=lambda: ops.conv2d_h2(xh, wh, co, ks, bias=bias, res=rs, scale=0.7071, colstats=True, w_fmt=1),
                   f16=lambda: ops.conv2d_h2(xh, wh, co, ks, bias=bias, colstats=True, w_fmt=1, out_f16=True))
        line = f"{H:3d}^2 {ci:4d}->{co:3d} k{ks} B={B} |"
        res = {}
        for sw in (1, 2, 1, 2):
            ops.set_tuning("DP_H2_SW", sw)
            outs = {k: fn() for k, fn in fns.items()}
            if sw == 1 and not res:
                res = outs
            ok = all(torch.equal(outs[k].t, res[k].t) and torch.equal(outs[k].cols.buf, res[k].cols.buf) for k in fns)
            line += f" SW={sw} " + " ".join(f"{flop / timeit(fn, 20) / 1e9:5.0f}" for fn in fns.values()) + (" [ok] |" if ok else " [DIFF] |")
        print(line, flush=True)
    ops.set_tuning("DP_H2_SW", 1)


if __name__ == "__main__":
    main()
