"""Probe (GPU): the dominant convolution shapes of the 256x256 guided UNet at B=64 on the 8-wave kernel, by epilogue traffic:
fp32 residual + fp32 output (round 3's stream) against fp16 residual + fp16 output (the fp16 residual stream of round 4), and a
decoder ResBlock's second convolution with its 1x1 skip as K-segments against the un-fused pair of launches.
    python tests/probes/conv_epi_probe.py [--batch 64]"""
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
    B = int(sys.argv[sys.argv.index("--batch") + 1]) if "--batch" in sys.argv else 64
    for (H, ci, co) in [(256, 256, 256), (128, 512, 512), (64, 512, 512), (32, 1024, 1024)]:
        x = torch.randn(B, H, H, ci)
        w = torch.randn(co, ci, 3, 3) * (1.0 / (9 * ci)) ** 0.5
        wh = ops.order_conv_weight_w16(w).half().to(DEV)
        xh = torch.nn.functional.pad(x, (0, 0, 1, 1, 1, 1)).half().contiguous().to(DEV)
        del x
        bias = torch.randn(co, device=DEV)
        r32 = torch.randn(B, H, H, co, device=DEV)
        r16 = r32.half()
        flop = 2.0 * B * H * H * co * 9 * ci
        it = max(3, min(20, int(1.5e12 / flop)))
        f = lambda **kw: (lambda: ops.conv2d_h2(xh, wh, co, 3, bias=bias, colstats=True, w_fmt=1, **kw))
        t = {k: timeit(fn, it) for k, fn in (("plain", f()), ("out16", f(out_f16=True)), ("res32", f(res=r32)), ("res16+out16", f(res=r16, out_f16=True)))}
        print(f"{H:4d} {ci:5d}->{co:4d} B={B} | " + " | ".join(f"{k} {flop / v / 1e9:5.0f} TF ({v:.3f} ms)" for k, v in t.items()), flush=True)
        del r32, r16
    # decoder ResBlock (cin = 2 co): 3x3 (co -> co) + 1x1 skip over cat(x [co], x2 [co])
    for (H, co) in [(256, 256), (128, 256), (64, 512)]:
        c1 = c2 = co
        h = torch.nn.functional.pad(torch.randn(B, H, H, co), (0, 0, 1, 1, 1, 1)).half().contiguous().to(DEV)
        s1, s2 = torch.randn(B, H, H, c1).half().to(DEV), torch.randn(B, H, H, c2).half().to(DEV)
        w3 = torch.randn(co, co, 3, 3) * (1.0 / (9 * co)) ** 0.5
        ws = torch.randn(co, c1 + c2, 1, 1) * (1.0 / (c1 + c2)) ** 0.5
        wf = ops.order_conv_weight_w16(ops.fuse_skip_weight(w3, ws)).half().to(DEV)
        w3p, wsp = ops.order_conv_weight_w16(w3).half().to(DEV), ops.order_conv_weight_w16(ws).half().to(DEV)
        bias = torch.randn(co, device=DEV)
        if not ops.takes_segments(H, H, 3, co, co, c1, c2):
            print(f"{H} {co}: segments not taken at B={B}")
            continue
        raw = torch.nn.functional.pad(torch.cat([s1, s2], dim=3), (0, 0, 1, 1, 1, 1)).contiguous()

        def fused():
            return ops.conv2d_h2(h, wf, co, 3, bias=bias, colstats=True, w_fmt=1, out_f16=True, segs=(s1, s2))

        def pair():
            skip = ops.conv2d_h2(raw, wsp, co, 1, bias=bias, w_fmt=1, out_f16=True)
            return ops.conv2d_h2(h, w3p, co, 3, bias=bias, res=skip, colstats=True, w_fmt=1, out_f16=True)

        flop = 2.0 * B * H * H * co * (9 * co + c1 + c2)
        it = max(3, min(20, int(1.5e12 / flop)))
        tf, tp = timeit(fused, it), timeit(pair, it)
        print(f"skip fusion {H:4d} {2 * co}->{co} B={B} | fused {tf:.3f} ms ({flop / tf / 1e9:5.0f} TF) | 1x1 + 3x3 {tp:.3f} ms (+ the raw-operand write of GroupNorm-apply, not timed)", flush=True)


if __name__ == "__main__":
    main()
