"""The stem convolution (3 -> N, 3x3) on the generic fp32-MFMA tiles vs the write-bound stem kernel of round 6 (csrc/stem.hip):
    python tests/probes/stem_bench.py [--batch B]
prints ms per launch and the output write rate of both, and their largest difference."""
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
    for (H, N, b) in ((256, 256, B), (256, 256, max(1, B // 16)), (32, 128, 4 * B)):
        x = (torch.randn(b, H, H, 3) * 1.5).to(DEV)
        w = torch.randn(N, 3, 3, 3) * 0.2
        wp, w16 = ops.pack_conv_weight(w).to(DEV), ops.pack_stem_weight(w).to(DEV)
        bias = torch.randn(N, device=DEV)
        for f16 in (True, False):
            old = lambda: ops.conv2d(x, wp, N, 3, bias=bias, colstats=True, out_f16=f16)
            new = lambda: ops.conv2d_stem(x, w16, N, bias=bias, colstats=True, out_f16=f16)
            a, t0 = old(), timeit(old)
            c, t1 = new(), timeit(new)
            gb = a.t.numel() * a.t.element_size() / 1e9
            err = (a.t.float() - c.t.float()).abs().max().item()
            print(f"{H:4d}^2 3->{N:3d} B={b:4d} out={'fp16' if f16 else 'fp32'} | generic fp32 tiles {t0:7.3f} ms ({gb / t0:5.2f} TB/s written) | "
                  f"stem kernel {t1:7.3f} ms ({gb / t1:5.2f} TB/s) | max diff {err:.2e}", flush=True)


if __name__ == "__main__":
    main()
