"""Does the last-level cache (256 MB Infinity Cache) serve the second pass of the three-launch GroupNorm backward when the two passes
run SAMPLE BY SAMPLE?  The statistics pass and the apply pass both read x (fp16) and dy (fp32): over a whole batch of 256^2 maps that
is 6.4 GB between the two reads of a byte, over one sample 100-200 MB.  Times ops.group_norm_bwd(one_pass=False) on the whole batch
against a loop over chunks of the batch (same kernels, B = chunk, offset pointers), for the shapes of the guided UNet's adjoint.
    python tests/probes/gn_bwd_chunk_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from diffpure_amd import ops  # noqa: E402

DEV = "cuda:0"


def timeit(fn, iters=5):
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
    G = 32
    for (B, H, C, split, addend) in [(32, 256, 256, "h1", False), (32, 256, 256, False, True), (32, 128, 512, "h1", False), (32, 128, 512, False, True),
                                     (32, 64, 512, "h1", False), (32, 64, 1024, False, True)]:
        x = (torch.randn(B, H, H, C, device=DEV) * 2 + 0.5).half()
        dy = torch.randn(B, H, H, C, device=DEV)
        ad = torch.randn(B, H, H, C, device=DEV) if addend else None
        gamma, beta = torch.rand(C, device=DEV) + 0.5, torch.randn(C, device=DEV) * 0.1
        st = torch.stack([torch.zeros(B, G, device=DEV) + 0.5, torch.ones(B, G, device=DEV) * 0.5], dim=2).contiguous()
        tab = torch.randn(B, 2 * C, device=DEV) * 0.3
        elems = B * H * H * C
        byts = elems * (2 * (4 + 2) + (4 + 4 if addend else 2))          # both passes read x + dy; apply also addend + writes dx

        def whole():
            return ops.group_norm_bwd(x, G, gamma, beta, st, dy, film=(tab[:, :C], tab[:, C:]), act=True, split=split, addend=ad, one_pass=False)

        def chunked(n):
            def f():
                for b0 in range(0, B, n):
                    sl = slice(b0, b0 + n)
                    ops.group_norm_bwd(x[sl], G, gamma, beta, st[sl], dy[sl], film=(tab[sl, :C], tab[sl, C:]), act=True, split=split,
                                       addend=None if ad is None else ad[sl], one_pass=False)
            return f

        row = [f"{H:3d}^2 x {C:4d} B={B} {'h1 out' if split else 'fp32 out + addend':18s} | whole {timeit(whole):7.3f} ms"]
        for nt in (0,):
            with ops.tuning(DP_GNB_NT=nt):
                row.append(f"NT=0 whole {timeit(whole):7.3f}")
                for n in (() if os.environ.get("PROBE_WHOLE_ONLY") else (1, 2, 4, 8)):
                    t = timeit(chunked(n))
                    row.append(f"chunks of {n}: {t:7.3f} ({byts / t / 1e9:5.2f} TB/s)")
        print(" | ".join(row), flush=True)
        del x, dy, ad
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
