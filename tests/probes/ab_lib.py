"""A/B of two builds of the library on the same box (tuning aid): python tests/probes/ab_lib.py <other.so>"""
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 2 and sys.argv[2] == "child":
    sys.path.insert(0, ROOT)
    import torch
    from diffpure_amd import _lib
    if sys.argv[1] != "-":
        _lib.LIB_PATH = sys.argv[1]
    from diffpure_amd import ops
    sys.path.insert(0, os.path.join(ROOT, "tests", "probes"))
    from conv_bench import timeit
    B, DEV = 16, "cuda:0"
    for (H, ci, co) in [(256, 256, 256), (256, 512, 256), (64, 512, 512)]:
        x = torch.randn(B, H, H, ci, device=DEV)
        w = torch.randn(co, ci, 3, 3) * (1.0 / (9 * ci)) ** 0.5
        wh = ops.pack_conv_weight_h2(w, DEV)
        xh = ops.pack_h2(torch.nn.functional.pad(x, (0, 0, 1, 1, 1, 1)).reshape(-1, ci)).reshape(B, H + 2, H + 2, 2 * ci)
        bias = torch.randn(co, device=DEV)
        flop = 2.0 * B * H * H * co * 9 * ci
        fn = lambda: ops.conv2d_h2(xh, wh, co, 3, bias=bias, colstats=True)
        ts = [timeit(fn, 10) for _ in range(3)]
        print(f"{H} {ci} {co}: {flop / statistics.median(ts) / 1e9:.1f} TF", flush=True)
else:
    for rnd in range(2):
        for lib in ("-", sys.argv[1]):
            out = subprocess.run([sys.executable, __file__, lib, "child"], capture_output=True, text=True).stdout.strip().replace("\n", " | ")
            print("HEAD " if lib == "-" else "OTHER", out, flush=True)
