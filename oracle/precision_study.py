"""TEST INFRASTRUCTURE / design study: how much purified-pixel error does a reduced-precision
matrix-core path cost?  Emulates, on the CPU oracle, MFMA variants with fp32 accumulation by
rounding the OPERANDS of every convolution / linear / NIN / attention matmul:

  fp32      reference (no rounding)
  fp16      a, w -> fp16                         (v_mfma_f32_32x32x16_f16, 1 pass)
  bf16      a, w -> bf16                         (v_mfma_f32_32x32x16_bf16, 1 pass)
  bf16x3    a = ah + al, w = wh + wl (bf16 each), ah*wh + ah*wl + al*wh   (3 passes)
  fp16x3    same split in fp16 (lo parts scaled by 2^11 to stay normal)    (3 passes)

Usage: python -m oracle.precision_study [--steps 20] [--batch 2]
"""
import argparse
import contextlib

import torch
import torch.nn.functional as F

from diffpure_amd.synth import synth_state_dict  # shapes only (plumbing)


def _split(x, dt):
    hi = x.to(dt).float()
    lo = (x - hi).to(dt).float()
    return hi, lo


class Mode:
    name = "fp32"


def _q(x):
    m = Mode.name
    if m == "fp16":
        return x.half().float()
    if m == "bf16":
        return x.bfloat16().float()
    return x


def _mm2(fn, a, w):
    """fn(a, w) bilinear in (a, w); apply the emulated precision."""
    m = Mode.name
    if m in ("fp32",):
        return fn(a, w)
    if m in ("fp16", "bf16"):
        return fn(_q(a), _q(w))
    dt = torch.bfloat16 if m == "bf16x3" else torch.float16
    ah, al = _split(a, dt)
    wh, wl = _split(w, dt)
    return fn(ah, wh) + (fn(ah, wl) + fn(al, wh))


@contextlib.contextmanager
def emulate(mode):
    Mode.name = mode
    oc, ol, oe = F.conv2d, F.linear, torch.einsum

    def conv2d(x, w, b=None, *a, **k):
        y = _mm2(lambda p, q: oc(p, q, None, *a, **k), x, w)
        return y if b is None else y + b.view(1, -1, 1, 1)

    def linear(x, w, b=None):
        y = _mm2(lambda p, q: ol(p, q), x, w)
        return y if b is None else y + b

    def einsum(eq, p, q):
        return _mm2(lambda u, v: oe(eq, u, v), p, q)

    F.conv2d, F.linear, torch.einsum = conv2d, linear, einsum
    try:
        yield
    finally:
        F.conv2d, F.linear, torch.einsum = oc, ol, oe
        Mode.name = "fp32"


def main():
    import yaml
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--modes", default="fp16,bf16,bf16x3,fp16x3")
    a = ap.parse_args()
    from oracle import ncsnpp as on, solvers as osol
    import diffpure_amd.ncsnpp as pn
    import bench
    cfg = on.parse_ncsnpp_config(bench.CIFAR_CFG)
    sd = synth_state_dict(pn.param_shapes(pn.parse_config(bench.CIFAR_CFG)), 1234)
    score = osol.make_score_fn("ncsnpp", sd, cfg)
    gen = torch.Generator().manual_seed(1234)
    x0 = torch.rand(a.batch, 3, 32, 32, generator=gen) * 2 - 1
    dt = 0.1 / a.steps
    n = len(osol.sde_time_grid(100, dt)) - 1
    e = torch.randn(x0.shape, generator=gen)
    zs = [torch.randn(x0.shape, generator=gen) for _ in range(n)]
    with torch.no_grad():
        ref = osol.sde_purify(score, x0, e, zs, 100, dt)
        ref1 = on.ncsnpp_forward(sd, cfg, x0, torch.full((a.batch,), 99.9))
        print(f"steps={n} |x|max={ref.abs().max():.3f} |eps|mean={ref1.abs().mean():.3f}")
        for m in a.modes.split(","):
            with emulate(m):
                out = osol.sde_purify(score, x0, e, zs, 100, dt)
                out1 = on.ncsnpp_forward(sd, cfg, x0, torch.full((a.batch,), 99.9))
            print(f"{m:8s} single-forward max-abs {float((out1 - ref1).abs().max()):.3e}   "
                  f"purified max-abs {float((out - ref).abs().max()):.3e}  mean-abs {float((out - ref).abs().mean()):.3e}")


if __name__ == "__main__":
    main()
