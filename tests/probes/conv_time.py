"""Time one h2 conv shape: python tests/probes/conv_time.py H Cin Cout B [iters]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from diffpure_amd import ops
H, ci, co, B = (int(v) for v in sys.argv[1:5])
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 20
dev = "cuda:0"
x = torch.randn(B, H, H, ci, device=dev)
w = torch.randn(co, ci, 3, 3) * (1.0 / (9 * ci)) ** 0.5
wh = ops.pack_conv_weight_h2(w, dev)
xh = ops.pack_h2(torch.nn.functional.pad(x, (0, 0, 1, 1, 1, 1)).reshape(-1, ci)).reshape(B, H + 2, H + 2, 2 * ci)
bias = torch.zeros(co, device=dev)
fn = lambda: ops.conv2d_h2(xh, wh, co, 3, bias=bias)
fn(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters): fn()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
print(f"H={H} Cin={ci} Cout={co} B={B}: {ms:.3f} ms  {2.0*B*H*H*co*9*ci/ms/1e9:.1f} TF  (DP_H2_PRIO={os.environ.get('DP_H2_PRIO','0')})")
