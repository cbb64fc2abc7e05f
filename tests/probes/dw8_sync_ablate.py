"""What the per-k-tile synchronisation of the 8-wave convolution kernel costs (timing ablations, WRONG RESULTS; needs
tests/probes/build_ablate.py):   python tests/probes/dw8_sync_ablate.py
DP_H2_DW_MODE: 0 as shipped, 2 no vmcnt wait / no barrier, 1 no DMA, 3 neither, 4 no ds_reads, 6 no reads and no waits, 7 none of them."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from diffpure_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.join(ROOT, "diffpure_amd", "csrc", "libdiffpure_hip_ablate.so")
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
    B = 64
    ops.set_tuning("DP_H2_DW", 1)
    for (H, ci, co) in [(256, 256, 256), (128, 512, 512), (64, 512, 512)]:
        x = torch.randn(B, H, H, ci)
        w = torch.randn(co, ci, 3, 3) * (1.0 / (9 * ci)) ** 0.5
        wh = ops.order_conv_weight_w16(w).half().to(DEV)
        xh = torch.nn.functional.pad(x, (0, 0, 1, 1, 1, 1)).half().contiguous().to(DEV)
        bias = torch.randn(co, device=DEV)
        flop = 2.0 * B * H * H * co * 9 * ci
        fn = lambda: ops.conv2d_h2(xh, wh, co, 3, bias=bias, colstats=True, w_fmt=1)
        line = f"{H:4d} {ci:5d}->{co:4d} B={B} |"
        for adepth in (3, 4):
            line += f" a{adepth}:"
            for m in (0, 2, 1, 3, 4, 6, 7, 0):
                os.environ["DP_H2_DW_MODE"] = str(m)
                line += f" m{m} {flop / timeit(fn, 6) / 1e9:5.0f}"
            line += " |"
        os.environ["DP_H2_DW_MODE"] = "0"
        print(line, flush=True)


if __name__ == "__main__":
    main()
