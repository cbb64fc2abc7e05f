"""One convolution shape, a few launches: target for rocprofv3 --pmc runs.
    python tests/probes/conv_one.py H Cin Cout B [f32|h2] [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from diffpure_amd import ops  # noqa: E402

H, ci, co, B = (int(v) for v in sys.argv[1:5])
mode = sys.argv[5] if len(sys.argv) > 5 else "h2"
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 5
dev = "cuda:0"
x = torch.randn(B, H, H, ci, device=dev)
w = torch.randn(co, ci, 3, 3) * (1.0 / (9 * ci)) ** 0.5
bias = torch.zeros(co, device=dev)
if mode == "h2":
    wh = ops.pack_conv_weight_h2(w, dev)
    xh = ops.pack_h2(torch.nn.functional.pad(x, (0, 0, 1, 1, 1, 1)).reshape(-1, ci)).reshape(B, H + 2, H + 2, 2 * ci)
    fn = lambda: ops.conv2d_h2(xh, wh, co, 3, bias=bias)
else:
    wp = ops.pack_conv_weight(w).to(dev)
    fn = lambda: ops.conv2d(x, wp, co, 3, bias=bias)
for _ in range(iters):
    fn()
torch.cuda.synchronize()
print("done", H, ci, co, B, mode)
