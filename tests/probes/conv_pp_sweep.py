"""Sweep an environment knob of the ping-pong convolution on a few shapes (tuning aid).
    python tests/probes/conv_pp_sweep.py VAR v1,v2,... [B]"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from diffpure_amd import ops  # noqa: E402
from conv_bench import timeit  # noqa: E402

DEV = "cuda:0"
var, vals = sys.argv[1], sys.argv[2].split(",")
B = int(sys.argv[3]) if len(sys.argv) > 3 else 16
os.environ["DP_H2_PP"] = "1"
for (H, ci, co) in [(256, 256, 256), (256, 512, 256), (128, 256, 256), (64, 512, 512)]:
    x = torch.randn(B, H, H, ci, device=DEV)
    w = torch.randn(co, ci, 3, 3) * (1.0 / (9 * ci)) ** 0.5
    wh = ops.pack_conv_weight_h2(w, DEV)
    xh = ops.pack_h2(torch.nn.functional.pad(x, (0, 0, 1, 1, 1, 1)).reshape(-1, ci)).reshape(B, H + 2, H + 2, 2 * ci)
    bias = torch.randn(co, device=DEV)
    res = torch.randn(B, H, H, co, device=DEV)
    flop = 2.0 * B * H * H * co * 9 * ci
    iters = max(3, min(30, int(1e12 / flop)))
    for with_res in (False, True):
        fn = lambda: ops.conv2d_h2(xh, wh, co, 3, bias=bias, res=res if with_res else None, colstats=True)
        t = {v: [] for v in vals}
        for _ in range(3):
            for v in vals:
                os.environ[var] = v
                t[v].append(timeit(fn, iters))
        print(f"{H:4d} {ci:5d} {co:5d} res={int(with_res)} | " +
              " | ".join(f"{v}: {flop / statistics.median(t[v]) / 1e9:6.1f}" for v in vals), flush=True)
