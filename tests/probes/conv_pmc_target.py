"""Target of tools/pmc_conv.sh: a few launches of ONE convolution shape through the product library (default: the most common
shape of the 256x256 guided UNet, 3x3 256->256 at 256^2, B=64, fp16 x fp16, column statistics on).
    python tests/probes/conv_pmc_target.py [--dw 0|1] [--res | --res16] [--f16out] [--shape H CI CO] [--batch B] [--iters N]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from diffpure_amd import ops  # noqa: E402


def arg(name, default, n=1):
    if name not in sys.argv:
        return default
    i = sys.argv.index(name)
    v = [int(x) for x in sys.argv[i + 1:i + 1 + n]]
    return v[0] if n == 1 else v


def main():
    B, iters = arg("--batch", 64), arg("--iters", 4)
    H, ci, co = arg("--shape", [256, 256, 256], 3)
    ops.set_tuning("DP_H2_DW", arg("--dw", 1))
    dev = "cuda:0"
    x = torch.randn(B, H, H, ci)
    w = torch.randn(co, ci, 3, 3) * (1.0 / (9 * ci)) ** 0.5
    wh = ops.order_conv_weight_w16(w).half().to(dev)
    xh = torch.nn.functional.pad(x, (0, 0, 1, 1, 1, 1)).half().contiguous().to(dev)
    bias = torch.randn(co, device=dev)
    rs = torch.randn(B, H, H, co, device=dev) if ("--res" in sys.argv or "--res16" in sys.argv) else None
    if rs is not None and "--res16" in sys.argv:
        rs = rs.half()                                  # the fp16 residual stream
    for _ in range(iters):
        ops.conv2d_h2(xh, wh, co, 3, bias=bias, res=rs, colstats=True, w_fmt=1, out_f16="--f16out" in sys.argv)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
