"""The 6-channel head convolution (256 -> 6, 3x3, 256^2) on the generic tiles vs the few-output-channels kernel (tuning aid):
    python tests/probes/head_conv.py [--batch B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from diffpure_amd import ops  # noqa: E402

DEV = "cuda:0"


def timeit(fn, iters=10):
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
    for (H, C, N) in ((256, 256, 6), (64, 256, 6), (32, 128, 3)):
        xh = torch.nn.functional.pad(torch.randn(B, H, H, C), (0, 0, 1, 1, 1, 1)).half().contiguous().to(DEV)
        w = torch.randn(N, C, 3, 3) * (1.0 / (9 * C)) ** 0.5
        wh = ops.order_conv_weight_w16(w).half().to(DEV)
        bias = torch.randn(N, device=DEV)
        fn = lambda: ops.conv2d_h2(xh, wh, N, 3, bias=bias, w_fmt=1)
        os.environ["DP_H2_NN"] = "0"
        base, t0 = fn(), timeit(fn)
        os.environ["DP_H2_NN"] = "1"
        got, t1 = fn(), timeit(fn)
        gb = xh.numel() * 2 / 1e9
        print(f"{H:4d}^2 {C:4d}->{N:2d} B={B} | generic {t0:7.3f} ms | few-channel kernel {t1:7.3f} ms ({gb / t1:5.2f} TB/s of operand) "
              f"[{'ok' if torch.equal(got, base) else 'DIFF'}]", flush=True)
    os.environ.pop("DP_H2_NN", None)


if __name__ == "__main__":
    main()
