"""Times the one-pass GroupNorm backward against the three-launch form (+ add) on the adjoint's tensor shapes.
    python tests/probes/gn_bwd_one_pass_probe.py [--batch 128]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from diffpure_amd import ops  # noqa: E402

SHAPES = [  # H, W, C1, C2, G, resample, addends
    (32, 32, 128, 0, 32, 0, 1), (32, 32, 128, 0, 32, 2, 1), (32, 32, 256, 0, 32, 0, 1), (16, 16, 256, 0, 32, 0, 1), (16, 16, 256, 256, 32, 0, 2),
    (16, 16, 256, 128, 32, 1, 2), (8, 8, 256, 0, 32, 0, 1), (8, 8, 256, 256, 32, 0, 2), (4, 4, 256, 256, 32, 0, 2),
    (8, 8, 1024, 1024, 32, 0, 2), (16, 16, 512, 0, 32, 0, 1), (16, 16, 1024, 512, 32, 0, 2),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    rn = lambda *s: torch.randn(*s, device=dev, generator=g)
    for (H, W, C1, C2, G, rs, nadd) in SHAPES:
        B = a.batch
        C = C1 + C2
        ok = ops.gn_bwd_fused_ok(H, W, C1, C2, G, rs)
        x, x2 = rn(B, H, W, C1), (rn(B, H, W, C2) if C2 else None)
        ho, wo = (2 * H, 2 * W) if rs == 1 else ((H // 2, W // 2) if rs == 2 else (H, W))
        dy = rn(B, ho, wo, C)
        gamma, beta = rn(C), rn(C)
        st = ops.group_norm_stats(x, G, 1e-5, x2)
        kw = dict(x2=x2, act=True, resample=rs, addend=rn(B, H, W, C1) if nadd else None, addend2=rn(B, H, W, C2) if nadd > 1 and C2 else None)
        res = {}
        for name, op in (("three", False), ("one", None)):
            if op is None and not ok:
                continue
            for _ in range(3):
                ops.group_norm_bwd(x, G, gamma, beta, st, dy, one_pass=op, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ops.group_norm_bwd(x, G, gamma, beta, st, dy, one_pass=op, **kw)
            e1.record()
            torch.cuda.synchronize()
            res[name] = e0.elapsed_time(e1) / 20 * 1e3
        byt = (B * H * W * C * 4 * (2 + (1 if nadd else 0)) + B * ho * wo * C * 4)
        line = f"{H:3d}x{W:<3d} {C1:4d}+{C2:<4d} rs={rs} B={B}: three-launch+add {res['three']:7.1f} us"
        if "one" in res:
            line += f"  one-pass {res['one']:7.1f} us ({byt / res['one'] / 1e6:.2f} TB/s algorithmic)  x{res['three'] / res['one']:.2f}"
        else:
            line += "  one-pass: shape not served"
        print(line, flush=True)


if __name__ == "__main__":
    main()
