"""GroupNorm-apply bandwidth probe (tuning aid, not a test):  python tests/probes/gn_bench.py [--batch B]
For the ResBlock shapes of the 256x256 guided UNet: the fp32-in -> fp16-operand pass (6 HBM bytes per element) against the
fp16-in one (dp_gn_apply_f16in, 4 bytes per element); prints ms and the HBM rate each reaches."""
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
    for (H, C) in ((256, 256), (128, 256), (64, 512), (32, 512), (16, 1024)):
        x = torch.randn(B, H, H, C, device=DEV)
        x16 = x.half()
        gamma, beta = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
        table = torch.randn(1, 2 * C, device=DEV) * 0.1
        film = (table[:, :C], table[:, C:])
        stats = ops.group_norm_stats(x, 32, 1e-5)
        n = x.numel()
        t32 = timeit(lambda: ops.group_norm(x, 32, 1e-5, gamma, beta, film=film, act=True, split="h1", stats=stats))
        t16 = timeit(lambda: ops.group_norm_f16in(x16, 32, gamma, beta, stats, film=film, act=True))
        print(f"{H:4d}^2 x {C:5d} B={B} | fp32 in {t32:7.3f} ms {6 * n / t32 / 1e9:6.2f} TB/s | fp16 in {t16:7.3f} ms {4 * n / t16 / 1e9:6.2f} TB/s", end="", flush=True)
        # round 6 probe: non-temporal hints on the streaming loads (bit 0) / stores (bit 1) of the fp16 pass
        for nt in (1, 2, 3):
            with ops.tuning(DP_GN_NT=nt):
                tn = timeit(lambda: ops.group_norm_f16in(x16, 32, gamma, beta, stats, film=film, act=True))
            print(f" | NT={nt}: {4 * n / tn / 1e9:5.2f}", end="", flush=True)
        print(flush=True)


if __name__ == "__main__":
    main()
