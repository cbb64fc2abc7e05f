"""-m gpu: backward operators and the adjoint-ODE path (BASELINE.json configs[4]) on the HIP engine
against torch.autograd (ops) and the CPU oracle (network VJP, adjoint integration)."""
import argparse

import math

import pytest
import torch

import refops
from conftest import load_golden
from diffpure_amd.synth import synth_state_dict

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

# Tolerances: north_star's bar - 1e-3 max-abs on purified pixels - and 5e-3 relative on gradients, for EVERY arithmetic incl.
# "f16sr", the one every runner ships with (fp16 activations x fp16 weights re-rounded stochastically per UNet call).  The loops in
# this file are 8-10 steps at dt = 1e-2 on small networks, ten times the product's step, so the per-step fp16 perturbation is ten
# times larger than at dt = 1e-3 and has nothing to average over: f16sr measures 5.2e-4 / 6.4e-4 at worst here (round 3), 22-bit
# "f16x3" ~1e-5.  At the product's grid (100-150 steps, dt = 1e-3) see tests/test_gpu_loops.py.
PIX_TOL = {"f32": 1e-3, "f16x3": 1e-3, "f16sr": 1e-3}
GRAD_TOL = {"f32": 5e-3, "f16x3": 5e-3, "f16sr": 5e-3}
SHIPPED = ["f16x3", "f16sr"]



def rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


def relerr(a, b):
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-20)).item()


GNB_CASES = [
    # B, H, W, C1, C2, G, film_rows, act, resample
    (2, 8, 8, 128, 0, 32, 0, True, 0),
    (2, 8, 8, 256, 128, 32, 0, True, 0),     # split input -> two gradient tensors
    (2, 4, 4, 1024, 512, 32, 0, True, 0),    # group of 48 straddles the split
    (3, 8, 8, 256, 0, 32, 2, True, 0),       # per-sample FiLM
    (1, 8, 8, 256, 0, 32, 1, True, 0),       # broadcast FiLM
    (2, 8, 8, 128, 0, 32, 0, True, 1),       # forward nearest x2
    (2, 8, 8, 128, 0, 32, 0, True, 2),       # forward mean 2x2
    (2, 16, 16, 32, 0, 8, 0, False, 0),      # attention pre-norm (no activation)
    (2, 8, 8, 128, 0, 32, 0, True, 3),       # forward FIR x2 up   (fir: True networks; asymmetric taps expose a flipped order)
    (2, 8, 8, 128, 0, 32, 0, True, 4),       # forward FIR x2 down
    (2, 6, 10, 64, 64, 32, 2, True, 3),      # FIR up, split input, FiLM, non-square
    (1, 2, 2, 128, 0, 32, 0, True, 4),       # FIR down to a single pixel: every tap but one falls outside
    (3, 4, 6, 256, 128, 32, 0, False, 4),
]
FIR_TAPS = (0.1, 0.2, 0.3, 0.4)


@pytest.mark.parametrize("case", GNB_CASES, ids=[str(c) for c in GNB_CASES])
def test_group_norm_bwd(case):
    from diffpure_amd import ops
    B, H, W, C1, C2, G, film_rows, act, rs = case
    C = C1 + C2
    x = rnd(B, H, W, C1, seed=1) * 2 + 0.5
    x2 = (rnd(B, H, W, C2, seed=2) - 1.0) if C2 else None
    gamma, beta = 1 + 0.1 * rnd(C, seed=3), 0.1 * rnd(C, seed=4)
    film = None
    if film_rows:
        tab = 0.3 * rnd(B if film_rows == 2 else 1, 2 * C, seed=5)
        film = (tab[:, :C], tab[:, C:])
    eps = 1e-5
    ho, wo = (2 * H, 2 * W) if rs in (1, 3) else ((H // 2, W // 2) if rs in (2, 4) else (H, W))
    fir = FIR_TAPS if rs >= 3 else None
    dy = rnd(B, ho, wo, C, seed=6)
    d64 = lambda t: None if t is None else t.double()
    st64 = refops.group_norm_stats(x.double(), G, eps, d64(x2)).double()
    ref1, ref2 = refops.group_norm_bwd(x.double(), G, gamma.double(), beta.double(), st64, dy.double(), d64(x2),
                                       None if film is None else (film[0].double(), film[1].double()), act, rs, fir=fir)
    d = lambda t: None if t is None else t.to(DEV)
    film_d = None
    if film is not None:
        tab_d = tab.to(DEV)
        film_d = (tab_d[:, :C], tab_d[:, C:])
    st = ops.group_norm_stats(d(x), G, eps, d(x2))
    g1, g2 = ops.group_norm_bwd(d(x), G, d(gamma), d(beta), st, d(dy), x2=d(x2), film=film_d, act=act, resample=rs, fir=fir)
    assert relerr(g1.cpu(), ref1.float()) < 2e-4
    if C2:
        assert relerr(g2.cpu(), ref2.float()) < 2e-4
    else:
        gh, _ = ops.group_norm_bwd(d(x), G, d(gamma), d(beta), st, d(dy), film=film_d, act=act, resample=rs, split=True, fir=fir)
        assert torch.equal(gh.cpu(), refops.to_h2(g1.cpu()))     # h2 form of the same gradient
        g16, _ = ops.group_norm_bwd(d(x), G, d(gamma), d(beta), st, d(dy), film=film_d, act=act, resample=rs, split="h1", fir=fir)
        assert torch.equal(g16.cpu(), torch.nn.functional.pad(g1.cpu(), (0, 0, 1, 1, 1, 1)).half())     # and its plain-fp16 form


ONE_PASS_CASES = [
    # B, H, W, C1, C2, G, act, resample, addends, scale       (the adjoint's shapes: NCSN++ 32..4, guided 8 / 16)
    (3, 32, 32, 128, 0, 32, True, 0, 1, 0.7071067811865476),
    (3, 32, 32, 128, 0, 32, True, 2, 1, 1.0),
    (2, 16, 16, 256, 0, 32, True, 0, 1, 0.7071067811865476),
    (2, 16, 16, 256, 256, 32, True, 0, 2, 1.0),
    (2, 16, 16, 128, 128, 32, True, 1, 2, 1.0),
    (2, 8, 8, 256, 256, 32, True, 0, 2, 1.0),
    (2, 8, 8, 256, 0, 32, False, 0, 1, 0.7071067811865476),      # attention pre-norm
    (5, 4, 4, 256, 0, 32, True, 1, 0, 1.0),
    (2, 8, 8, 1024, 1024, 32, True, 0, 2, 1.0),                  # guided, 8 x 8
    (2, 16, 16, 512, 0, 32, True, 0, 1, 1.0),
    (2, 6, 10, 64, 64, 32, True, 0, 2, 1.0),                     # ragged pixel count
]


@pytest.mark.parametrize("case", ONE_PASS_CASES, ids=[str(c) for c in ONE_PASS_CASES])
def test_group_norm_bwd_one_pass_form_with_folded_skip_gradient(case):
    """The one-pass GroupNorm backward (csrc/norm_bwd.hip gn_bwd_fused_kernel: group sums and dx from registers, skip-branch
    gradient added in the same pass) against the fp64 autograd reference and against the three-launch form + add."""
    from diffpure_amd import ops
    B, H, W, C1, C2, G, act, rs, nadd, scale = case
    C = C1 + C2
    assert ops.gn_bwd_fused_ok(H, W, C1, C2, G, rs)
    x = rnd(B, H, W, C1, seed=1) * 2 + 0.5
    x2 = (rnd(B, H, W, C2, seed=2) - 1.0) if C2 else None
    gamma, beta = 1 + 0.1 * rnd(C, seed=3), 0.1 * rnd(C, seed=4)
    tab = 0.3 * rnd(B, 2 * C, seed=5)
    ho, wo = (2 * H, 2 * W) if rs == 1 else ((H // 2, W // 2) if rs == 2 else (H, W))
    dy = rnd(B, ho, wo, C, seed=6)
    ad = rnd(B, H, W, C1, seed=7) if nadd >= 1 else None
    ad2 = rnd(B, H, W, C2, seed=8) if nadd >= 2 and C2 else None
    d64 = lambda t: None if t is None else t.double()
    st64 = refops.group_norm_stats(x.double(), G, 1e-5, d64(x2)).double()
    ref1, ref2 = refops.group_norm_bwd(x.double(), G, gamma.double(), beta.double(), st64, dy.double(), d64(x2),
                                       (tab[:, :C].double(), tab[:, C:].double()), act, rs, addend=d64(ad), addend2=d64(ad2),
                                       addend_scale=scale)
    d = lambda t: None if t is None else t.to(DEV)
    tab_d = tab.to(DEV)
    film_d = (tab_d[:, :C], tab_d[:, C:])
    st = ops.group_norm_stats(d(x), G, 1e-5, d(x2))
    kw = dict(x2=d(x2), film=film_d, act=act, resample=rs, addend=d(ad), addend2=d(ad2), addend_scale=scale)
    g1, g2 = ops.group_norm_bwd(d(x), G, d(gamma), d(beta), st, d(dy), **kw)
    t1, t2 = ops.group_norm_bwd(d(x), G, d(gamma), d(beta), st, d(dy), one_pass=False, **kw)
    assert relerr(g1.cpu(), ref1.float()) < 2e-4 and relerr(g1, t1) < 2e-6
    if C2:
        assert relerr(g2.cpu(), ref2.float()) < 2e-4 and relerr(g2, t2) < 2e-6
    else:
        for split in (True, "h1"):
            kw2 = dict(film=film_d, act=act, resample=rs, split=split)
            gh, _ = ops.group_norm_bwd(d(x), G, d(gamma), d(beta), st, d(dy), **kw2)
            g0, _ = ops.group_norm_bwd(d(x), G, d(gamma), d(beta), st, d(dy), film=film_d, act=act, resample=rs)
            want = refops.to_h2(g0.cpu()) if split is True else torch.nn.functional.pad(g0.cpu(), (0, 0, 1, 1, 1, 1)).half()
            assert torch.equal(gh.cpu(), want)
    # a sample's gradient does not depend on the batch it sits in
    s1, _ = ops.group_norm_bwd(d(x)[1:2].contiguous(), G, d(gamma), d(beta), st[1:2].contiguous(), d(dy)[1:2].contiguous(),
                               x2=None if x2 is None else d(x2)[1:2].contiguous(), film=(tab_d[1:2, :C], tab_d[1:2, C:]), act=act, resample=rs,
                               addend=None if ad is None else d(ad)[1:2].contiguous(), addend2=None if ad2 is None else d(ad2)[1:2].contiguous(),
                               addend_scale=scale)
    assert torch.equal(s1[0], g1[1])


LEAN_CASES = [
    # B, H, W, C1, C2, G, act, film rows (0 none / 1 shared / B per sample), addends (0 / 1 / 2)
    (2, 32, 32, 128, 128, 32, True, 2, 2),        # concatenated skip, both addends, per-sample FiLM (an up-path ResBlock)
    (3, 16, 16, 256, 0, 32, True, 3, 1),          # single source, one addend
    (2, 6, 10, 64, 0, 16, True, 1, 0),            # ragged pixel count, groups of 4 channels (an octet spans two groups), shared FiLM row
    (5, 8, 8, 512, 256, 32, False, 0, 2),         # C = 768: 96 channel octets (not a power of two), no activation (attention pre-norm)
    (1, 64, 64, 64, 64, 32, True, 1, 0),          # one sample, groups of 4 channels
    (4, 4, 4, 1024, 0, 32, True, 4, 1),           # more octets than a workgroup has threads / 2
    (2, 16, 16, 128, 128, 32, True, 2, 1),        # ONE addend on a concatenated source: served by the generic kernel (must still agree)
]


@pytest.mark.parametrize("case", LEAN_CASES, ids=str)
def test_group_norm_bwd_lean_apply_pass_same_bits(case):
    """Round 6: the lean apply pass of the three-launch GroupNorm backward (gn_bwd_apply_lean_kernel: a thread keeps one channel octet,
    per-sample values re-loaded only at sample boundaries, an incremental pixel walk without divisions, output format and addends as
    template parameters) against the generic apply kernel it replaces on the un-resampled fp16-tape launches (DP_GNB_LEAN=0): identical
    bytes for the fp32 output (with and without addends, two sources) and for the zero-bordered fp16 operand (border included)."""
    from diffpure_amd import ops
    B, H, W, C1, C2, G, act, frows, nadd = case
    C = C1 + C2
    d = lambda t: None if t is None else t.to(DEV)
    x = d(rnd(B, H, W, C1, seed=1) * 2 + 0.5)
    x2 = d(rnd(B, H, W, C2, seed=2) - 1.0) if C2 else None
    gamma, beta = d(1 + 0.1 * rnd(C, seed=3)), d(0.1 * rnd(C, seed=4))
    tab = d(0.3 * rnd(frows, 2 * C, seed=5)) if frows else None
    film = None if tab is None else (tab[:, :C], tab[:, C:])
    dy = d(rnd(B, H, W, C, seed=6))
    ad = d(rnd(B, H, W, C1, seed=7)) if nadd >= 1 else None
    ad2 = d(rnd(B, H, W, C2, seed=8)) if nadd >= 2 and C2 else None
    x16, x2_16 = x.half(), None if x2 is None else x2.half()
    st = ops.group_norm_stats(x16.float(), G, 1e-5, None if x2_16 is None else x2_16.float())     # statistics of the values the tape holds

    def run():
        outs = list(ops.group_norm_bwd(x16, G, gamma, beta, st, dy, x2=x2_16, film=film, act=act, addend=ad, addend2=ad2, addend_scale=0.7071,
                                       one_pass=False))
        if C2 == 0:
            outs.append(ops.group_norm_bwd(x16, G, gamma, beta, st, dy, film=film, act=act, split="h1", one_pass=False)[0])
        return [o for o in outs if o is not None]

    with ops.tuning(DP_GNB_LEAN=0):
        base = run()
    with ops.tuning(DP_GNB_LEAN=1):
        got = run()
        with ops.tuning(DP_GNB_NT=3):
            got_nt = run()
    assert len(got) == len(base) and all(torch.equal(g, b) for g, b in zip(got, base)), case
    assert all(torch.equal(g, b) for g, b in zip(got_nt, base)), case
    # against the fp64 reference as well (the fp16-rounded tape values are the inputs)
    d64 = lambda t: None if t is None else t.cpu().double()
    ref1, ref2 = refops.group_norm_bwd(x16.cpu().double(), G, d64(gamma), d64(beta), d64(st), d64(dy), d64(x2_16),
                                       None if tab is None else (d64(tab[:, :C]).expand(B, C), d64(tab[:, C:]).expand(B, C)), act, 0,
                                       addend=d64(ad), addend2=d64(ad2), addend_scale=0.7071)
    assert relerr(got[0].cpu(), ref1.float()) < 2e-4
    if C2:
        assert relerr(got[1].cpu(), ref2.float()) < 2e-4


@pytest.mark.parametrize("x16", [False, True], ids=["fp32 tape", "fp16 tape"])
def test_group_norm_bwd_three_launch_form_non_temporal_hints_same_bits(x16):
    """Round 6: the statistics and apply passes of the three-launch GroupNorm backward take non-temporal hints on their streaming
    accesses for gradient maps far beyond the last-level cache (DP_GNB_NT: -1 by size, 0 never, 3 forced).  Pure addressing hints:
    the same bytes, for the fp32 output with both addends and for the fp16 operand output."""
    from diffpure_amd import ops
    B, H, W, C1, C2, G = 2, 32, 32, 128, 128, 32
    C = C1 + C2
    d = lambda t: None if t is None else t.to(DEV)
    x, x2 = d(rnd(B, H, W, C1, seed=1) * 2 + 0.5), d(rnd(B, H, W, C2, seed=2) - 1.0)
    gamma, beta = d(1 + 0.1 * rnd(C, seed=3)), d(0.1 * rnd(C, seed=4))
    tab = d(0.3 * rnd(B, 2 * C, seed=5))
    film = (tab[:, :C], tab[:, C:])
    dy, ad, ad2 = d(rnd(B, H, W, C, seed=6)), d(rnd(B, H, W, C1, seed=7)), d(rnd(B, H, W, C2, seed=8))
    st = ops.group_norm_stats(x, G, 1e-5, x2)
    if x16:
        x, x2 = x.half(), x2.half()
    film1 = (tab[:, :C1].contiguous(), tab[:, C:C + C1].contiguous())
    st1 = ops.group_norm_stats(x.float(), G, 1e-5)

    def run():
        a = ops.group_norm_bwd(x, G, gamma, beta, st, dy, x2=x2, film=film, act=True, addend=ad, addend2=ad2, addend_scale=0.7, one_pass=False)
        b = ops.group_norm_bwd(x, G, gamma[:C1].contiguous(), beta[:C1].contiguous(), st1, dy[..., :C1].contiguous(), film=film1, act=True,
                               split="h1", one_pass=False)
        return [a[0], a[1], b[0]]

    with ops.tuning(DP_GNB_NT=0):
        base = run()
    for nt in (1, 2, 3, -1):
        with ops.tuning(DP_GNB_NT=nt):
            got = run()
        assert all(torch.equal(g, b) for g, b in zip(got, base)), nt


@pytest.mark.parametrize("case", [(2, 16, 256, 1, "split"), (2, 64, 128, 2, "legacy"), (1, 256, 256, 4, "legacy"), (2, 64, 128, 2, "split")],
                         ids=str)
def test_attention_bwd(case):
    from diffpure_amd import ops
    B, T, C, heads, layout = case
    qkv, dout = rnd(B, T, 3 * C, seed=11), rnd(B, T, C, seed=12)
    ref = refops.attention_bwd(qkv.double(), None, dout.double(), heads, layout).float()
    out, probs = ops.attention(qkv.to(DEV), heads, layout, return_probs=True)
    got = ops.attention_bwd(qkv.to(DEV), probs, dout.to(DEV), heads, layout)
    assert relerr(got.cpu(), ref) < 1e-4


def test_small_backward_pieces():
    from diffpure_amd import ops
    for mode in (1, 2):
        dy = rnd(2, 8, 8, 64, seed=20)
        torch.testing.assert_close(ops.resample_bwd(dy.to(DEV), mode).cpu(), refops.resample_bwd(dy, mode), rtol=1e-6, atol=1e-6)
    # the FIR resamplers' adjoints: against autograd through the upfirdn2d statement (asymmetric taps, odd sizes) and against
    # autograd through the reference's own upsample_2d / downsample_2d (golden fir_ops.pt, make_golden_fir.py)
    for mode, shape in ((3, (2, 8, 8, 64)), (4, (2, 8, 8, 64)), (3, (1, 6, 10, 8)), (4, (1, 3, 5, 8)), (4, (1, 1, 1, 4)), (3, (1, 2, 2, 4))):
        dy = rnd(*shape, seed=27)
        torch.testing.assert_close(ops.resample_bwd(dy.to(DEV), mode, fir=FIR_TAPS).cpu(), refops.resample_bwd(dy.double(), mode, FIR_TAPS).float(),
                                   rtol=1e-5, atol=1e-6)
    gf = load_golden("fir_ops.pt")
    taps = ops.fir_taps(gf["k"])
    for mode, name in ((ops.RESAMPLE_FIR_UP, "up"), (ops.RESAMPLE_FIR_DOWN, "down")):
        got = nchw(ops.resample_bwd(nhwc(gf["dy_" + name]).to(DEV), mode, fir=taps).cpu())
        torch.testing.assert_close(got, gf["dx_" + name], rtol=1e-5, atol=1e-6)
    a, b = rnd(3, 5, 8, seed=21), rnd(3, 5, 8, seed=22)
    assert torch.equal(ops.add(a.to(DEV), b.to(DEV)).cpu(), a + b)
    x = rnd(2, 4, 4, 64, seed=23)
    assert torch.equal(ops.to_h2(x.to(DEV)).cpu(), refops.to_h2(x))
    # dgrad weight identity: <conv(x, W), dy> == <x, conv(dy, Wd)>
    w = rnd(24, 16, 3, 3, seed=24)
    xx, dy = rnd(2, 6, 6, 16, seed=25), rnd(2, 6, 6, 24, seed=26)
    y = ops.conv2d(xx.to(DEV), ops.pack_conv_weight(w).to(DEV), 24, 3).cpu()
    dx = ops.conv2d(dy.to(DEV), ops.pack_conv_weight(ops.dgrad_weight(w)).to(DEV), 16, 3).cpu()
    assert abs((y * dy).sum() - (xx * dx).sum()) < 1e-3 * (y * dy).sum().abs()


@pytest.mark.parametrize("precision", ["f32", "f16x3", "f16sr"])
def test_ncsnpp_fir_vjp_vs_reference_autograd(precision):
    """SURVEY.md 8f-4 / 8a adjoint: the input gradient of a `fir: True` NCSN++ (upfirdn2d resamplers in the BigGAN blocks) on the
    HIP engine against torch.autograd THROUGH THE REFERENCE MODULE for a seeded cotangent (tests/golden/make_golden_fir.py)."""
    from diffpure_amd import ncsnpp as pn
    g = load_golden("ncsnpp_fir_small.pt")
    cfg = pn.parse_config(g["cfg"])
    sd = synth_state_dict(pn.param_shapes(cfg), g["seed"])
    net = pn.NCSNpp(cfg, DEV, precision).load_state_dict(sd)
    tape = []
    out = nchw(net.forward(nhwc(g["x"]).to(DEV), g["labels"].to(DEV), tape=tape)).cpu()
    got = nchw(net.vjp(tape, nhwc(g["cot"]).to(DEV))).cpu()
    err = relerr(got, g["vjp"])
    print(f"fir NCSN++ vjp [{precision}] vs reference autograd: rel {err:.3e} (forward max-abs {(out - g['out']).abs().max():.3e})")
    assert err < (2e-3 if precision != "f16sr" else 5e-3), err


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_ncsnpp_full_vjp_vs_autograd(precision):
    from diffpure_amd import ncsnpp as pn
    from oracle import ncsnpp as on
    g = load_golden("ncsnpp_full.pt")
    cfg = pn.parse_config(g["cfg"])
    sd = synth_state_dict(pn.param_shapes(cfg), g["seed"])
    net = pn.NCSNpp(cfg, DEV, precision).load_state_dict(sd)
    x, lab = g["x"], g["labels"]
    u = rnd(*x.shape, seed=1)
    xr = x.clone().requires_grad_(True)
    (ref,) = torch.autograd.grad(on.ncsnpp_forward(sd, on.parse_ncsnpp_config(g["cfg"]), xr, lab), xr, u)
    tape = []
    net.forward(nhwc(x).to(DEV), lab.to(DEV), tape=tape)
    got = nchw(net.vjp(tape, nhwc(u).to(DEV))).cpu()
    assert relerr(got, ref) < 2e-3, relerr(got, ref)


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_guided_small_vjp_vs_autograd(precision):
    from diffpure_amd import guided_unet as pg
    from oracle import guided_unet as og
    g = load_golden("guided_small.pt")
    cfg = pg.parse_config(g["cfg"])
    sd = synth_state_dict(pg.param_shapes(cfg), g["seed"])
    net = pg.GuidedUNet(cfg, DEV, precision).load_state_dict(sd)
    x, t = g["x"], g["t"]
    u = rnd(*x.shape, seed=2)
    xr = x.clone().requires_grad_(True)
    (ref,) = torch.autograd.grad(og.guided_unet_forward(sd, og.parse_guided_config(g["cfg"]), xr, t)[:, :3], xr, u)
    tape = []
    net.forward(nhwc(x).to(DEV), t.float().to(DEV), tape=tape)
    got = nchw(net.vjp(tape, nhwc(u).to(DEV))).cpu()
    assert relerr(got, ref) < 2e-3, relerr(got, ref)


def test_config5_adjoint_ode_vs_oracle():
    """BASELINE.json configs[4] at test size: CIFAR-10 NCSN++ (full), adjoint-ODE dL/dx, B=2, 10 steps,
    against the oracle's restated continuous adjoint on the same grid."""
    from diffpure_amd import ncsnpp as pn
    from diffpure_amd.sde import Purifier
    from oracle import ncsnpp as on, solvers as osol
    g = load_golden("ncsnpp_full.pt")
    cfg = pn.parse_config(g["cfg"])
    sd = synth_state_dict(pn.param_shapes(cfg), g["seed"])
    score = osol.make_score_fn("ncsnpp", sd, on.parse_ncsnpp_config(g["cfg"]))
    gen = torch.Generator().manual_seed(9)
    x0 = g["x"]
    e = torch.randn(x0.shape, generator=gen)
    cot = torch.randn(x0.shape, generator=gen)
    step = 1e-2
    for precision in ("f32", "f16x3"):
        net = pn.NCSNpp(cfg, DEV, precision).load_state_dict(sd)
        pur = Purifier(net, "ncsnpp", DEV)
        xf = pur.ode(x0, 100, step, noise=dict(e=e, z=[]))
        with torch.no_grad():
            xf_ref = osol.ode_purify(score, x0, e, 100, step)
        assert (xf.cpu() - xf_ref).abs().max() < 1e-3
        ref = osol.ode_diffuse_grad(osol.ode_adjoint_grad(score, xf_ref, cot, 100, step), 100)
        got = (pur.ode_vjp(xf, cot, 100, step) * pur.diffuse_scale(100)).cpu()
        assert relerr(got, ref) < 5e-3, (precision, relerr(got, ref))


@pytest.mark.parametrize("precision", SHIPPED)
def test_ode_runner_autograd_on_gpu(tmp_path, precision):
    from runners.diffpure_ode import OdeGuidedDiffusion
    g = load_golden("ncsnpp_small.pt")

    def ns(d):
        n = argparse.Namespace()
        for k, v in d.items():
            setattr(n, k, ns(v) if isinstance(v, dict) else v)
        return n

    config = ns(g["cfg"])
    config.device = torch.device(DEV)
    args = argparse.Namespace(t=100, rand_t=False, t_delta=15, use_bm=False, sample_step=1, log_dir=str(tmp_path),
                              score_type="score_sde", seed=1234, synthetic_weights=True, step_size=1e-2, precision=precision)
    runner = OdeGuidedDiffusion(args, config, device=config.device)
    x = (torch.rand(3, 3, 16, 16) * 2 - 1).to(DEV).requires_grad_(True)
    out = runner.image_editing_sample(x, bs_id=9)
    loss = (out ** 2).sum()
    (gx,) = torch.autograd.grad(loss, x)
    assert gx.shape == x.shape and torch.isfinite(gx).all() and gx.abs().max() > 0


@pytest.mark.parametrize("precision", SHIPPED)
def test_sde_stochastic_adjoint_vs_oracle(precision):
    """dL/dx through the reverse-SDE solve with injected noise, full NCSN++, 10 steps."""
    from diffpure_amd import ncsnpp as pn
    from diffpure_amd.sde import Purifier
    from oracle import ncsnpp as on, solvers as osol
    g = load_golden("ncsnpp_full.pt")
    cfg = pn.parse_config(g["cfg"])
    sd = synth_state_dict(pn.param_shapes(cfg), g["seed"])
    score = osol.make_score_fn("ncsnpp", sd, on.parse_ncsnpp_config(g["cfg"]))
    gen = torch.Generator().manual_seed(9)
    x0 = g["x"]
    dt = 1e-2
    e = torch.randn(x0.shape, generator=gen)
    zs = [torch.randn(x0.shape, generator=gen) for _ in range(10)]
    cot = torch.randn(x0.shape, generator=gen)
    net = pn.NCSNpp(cfg, DEV, precision).load_state_dict(sd)
    pur = Purifier(net, "ncsnpp", DEV)
    noise = dict(e=e, z=zs)
    xf = pur.sde(x0, 100, dt, noise=noise)
    with torch.no_grad():
        xf_ref = osol.sde_purify(score, x0, e, zs, 100, dt)
    err_x = (xf.cpu() - xf_ref).abs().max().item()
    ref = osol.ode_diffuse_grad(osol.sde_adjoint_grad(score, xf_ref, cot, zs, 100, dt), 100)
    got = (pur.sde_vjp(xf, cot, 100, dt, noise=noise) * pur.diffuse_scale(100)).cpu()
    print(f"stochastic adjoint, full NCSN++, 10 steps [{precision}]: x max-abs {err_x:.3e}, dL/dx rel. error {relerr(got, ref):.3e}")
    assert err_x < PIX_TOL[precision], err_x
    assert relerr(got, ref) < GRAD_TOL[precision], relerr(got, ref)
    # with Philox noise the backward pass regenerates the forward path: deterministic
    xf2 = pur.sde(x0, 100, dt, seed=3, sample0=0)
    g1 = pur.sde_vjp(xf2, cot, 100, dt, seed=3, sample0=0)
    g2 = pur.sde_vjp(xf2, cot, 100, dt, seed=3, sample0=0)
    assert torch.equal(g1, g2) and torch.isfinite(g1).all()


# ---- the steps either side of the purifier --------------------------------------------
RESIZE_CASES = [
    # B, C, Hi, Wi, Ho, Wo, in_nhwc, out_nhwc, shift, scale
    (2, 3, 224, 224, 256, 256, False, True, -0.5, 2.0),     # eval_sde_adv.py:74-75,78
    (2, 3, 256, 256, 224, 224, True, False, 1.0, 0.5),      # eval_sde_adv.py:81-82,89
    (3, 3, 32, 32, 32, 32, False, True, -0.5, 2.0),         # CIFAR-10: identity resize, affine + repack only
    (1, 5, 17, 23, 40, 9, False, False, 0.25, -1.5),        # odd sizes, up and down at once, NCHW -> NCHW
    (2, 4, 9, 9, 31, 31, True, True, 0.0, 1.0),             # large up-scale factor
]


@pytest.mark.parametrize("case", RESIZE_CASES, ids=[str(c) for c in RESIZE_CASES])
def test_resize_affine_and_its_adjoint_vs_torch(case):
    """y = (F.interpolate(x, bilinear, align_corners=False) + shift) * scale and dL/dx, against torch on the CPU."""
    from diffpure_amd import ops
    B, C, Hi, Wi, Ho, Wo, in_nhwc, out_nhwc, shift, scale = case
    x = rnd(B, C, Hi, Wi, seed=1).requires_grad_(True)
    ref = (torch.nn.functional.interpolate(x, size=(Ho, Wo), mode="bilinear", align_corners=False) + shift) * scale
    cot = rnd(B, C, Ho, Wo, seed=2)
    (gref,) = torch.autograd.grad((ref * cot).sum(), x)
    xin = (nhwc(x.detach()) if in_nhwc else x.detach()).to(DEV)
    y = ops.resize_affine(xin, (Ho, Wo), shift, scale, in_nhwc, out_nhwc)
    y_nchw = nchw(y.cpu()) if out_nhwc else y.cpu()
    assert (y_nchw - ref.detach()).abs().max() < 2e-6 * max(1.0, ref.abs().max().item())
    dy = (nhwc(cot) if out_nhwc else cot).to(DEV)
    dx = ops.resize_affine_bwd(dy, (Hi, Wi), scale, in_nhwc, out_nhwc)
    dx_nchw = nchw(dx.cpu()) if in_nhwc else dx.cpu()
    assert relerr(dx_nchw, gref) < 1e-5
    # deterministic (gather form, no atomics)
    assert torch.equal(dx, ops.resize_affine_bwd(dy, (Hi, Wi), scale, in_nhwc, out_nhwc))


class _TinyClassifier(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(rnd(7, 3, seed=11))

    def forward(self, x):
        return x.mean(dim=(2, 3)) @ self.w.t()


@pytest.mark.parametrize("precision", SHIPPED)
@pytest.mark.parametrize("diffusion_type", ["sde", "ode"])
def test_adv_model_equals_the_torch_composition_and_is_differentiable(tmp_path, diffusion_type, precision):
    """diffpure_amd.adv_model.SDE_Adv_Model (fused resize/affine/repack kernels, NHWC in and out of the runner)
    against the upstream composition of eval_sde_adv.py:73-89 spelled with torch ops around the same runner:
    logits, and dL/dx of the whole defence (adjoint resize o adjoint purifier o adjoint resize)."""
    from diffpure_amd.adv_model import SDE_Adv_Model
    g = load_golden("ncsnpp_small.pt")

    def ns(d):
        n = argparse.Namespace()
        for k, v in d.items():
            setattr(n, k, ns(v) if isinstance(v, dict) else v)
        return n

    config = ns(g["cfg"])
    config.device = torch.device(DEV)
    args = argparse.Namespace(t=100, rand_t=False, t_delta=15, use_bm=False, sample_step=1, log_dir=str(tmp_path),
                              score_type="score_sde", seed=1234, synthetic_weights=True, step_size=2e-2, dt=2e-2, precision=precision,
                              diffusion_type=diffusion_type, domain="cifar10", classifier_name="none", diffusion_size=(16, 16))
    model = SDE_Adv_Model(args, config, classifier=_TinyClassifier())
    runner = model.runner
    F = torch.nn.functional
    x = torch.rand(3, 3, 14, 14, generator=torch.Generator().manual_seed(5)).to(DEV)
    cot = rnd(3, 7, seed=6).to(DEV)

    def composition(xin):
        up = F.interpolate(xin, size=(16, 16), mode="bilinear", align_corners=False)
        pur = runner.image_editing_sample((up - 0.5) * 2, bs_id=9)
        return model.classifier((F.interpolate(pur, size=(14, 14), mode="bilinear", align_corners=False) + 1) * 0.5)

    x1 = x.clone().requires_grad_(True)
    runner._calls = 0
    model.counter.fill_(9)
    out = model(x1)
    (g1,) = torch.autograd.grad((out * cot).sum(), x1)
    x2 = x.clone().requires_grad_(True)
    runner._calls = 0
    ref = composition(x2)
    (g2,) = torch.autograd.grad((ref * cot).sum(), x2)
    assert out.shape == (3, 7)
    assert (out - ref).abs().max() < 1e-4 * max(1.0, ref.abs().max().item())
    # the two spellings feed the purifier inputs that differ in the last bits (fused vs torch bilinear resize): under f16sr a
    # last-bit change of an activation can flip its fp16 rounding, so the gradients agree to the arithmetic's own noise level
    print(f"adv model vs torch composition [{diffusion_type}, {precision}]: dL/dx rel. difference {relerr(g1.cpu(), g2.cpu()):.3e}")
    assert relerr(g1.cpu(), g2.cpu()) < (1e-3 if precision == "f16x3" else 5e-3)
    # EOT replicas as one batch == separate purifications with the matching global sample indices
    with torch.no_grad():
        runner._calls = 0
        eot = model.forward_eot(x, 2)
        runner._calls = 0
        both = model(x.repeat(2, 1, 1, 1))
    assert eot.shape == (2, 3, 7) and torch.equal(eot.reshape(6, 7), both)
    assert (eot[0] - eot[1]).abs().max() > 0 if diffusion_type == "sde" else True


def test_bpda_purify_mode_with_150_replicas_per_image_on_gpu(tmp_path):
    """The BPDA+EOT driver's model (eval_sde_adv_bpda.py:83-106) on the HIP engine at the shipped arithmetic: `mode='purify'`
    arrives with `eot_defense_reps` = 150 replicas of every image in ONE batch (bpda_eot_attack.py:98-110) - a large-effective-batch
    forward.  Checked: shapes and ranges of the three modes, the call counter, that the 150 replicas of an image are 150 DIFFERENT
    purifications (noise keyed by the global sample index r * B + b), that any replica equals a separate purification of that image
    under the same sample index, and 'purify_and_classify' == classify(purify)."""
    from diffpure_amd.adv_model import SDE_Adv_Model
    g = load_golden("ncsnpp_small.pt")

    def ns(d):
        n = argparse.Namespace()
        for k, v in d.items():
            setattr(n, k, ns(v) if isinstance(v, dict) else v)
        return n

    config = ns(g["cfg"])
    config.device = torch.device(DEV)
    args = argparse.Namespace(t=100, rand_t=False, t_delta=15, use_bm=False, sample_step=1, log_dir=str(tmp_path),
                              score_type="score_sde", seed=1234, synthetic_weights=True, dt=2e-2, precision="f16sr",
                              diffusion_type="sde", domain="cifar10", classifier_name="none")
    model = SDE_Adv_Model(args, config, classifier=_TinyClassifier())
    runner = model.runner
    reps, b = 150, 2
    x = torch.rand(b, 3, 16, 16, generator=torch.Generator().manual_seed(8)).to(DEV)
    with torch.no_grad():
        runner._calls = 0
        model.counter.fill_(7)
        pur = model(x.repeat(reps, 1, 1, 1), mode="purify")            # [300, 3, 16, 16] in [0, 1]-ish
        assert pur.shape == (reps * b, 3, 16, 16) and torch.isfinite(pur).all() and int(model.counter.item()) == 8
        per_img = pur.reshape(reps, b, 3, 16, 16)
        spread = (per_img - per_img[:1]).abs().amax(dim=(1, 2, 3, 4))
        assert (spread[1:] > 1e-3).all()                               # every replica walked its own Brownian path
        # replica r of image i is sample r * b + i of the call: a separate purification with that sample index reproduces it
        r, i = 77, 1
        runner._calls = 0
        one = runner.purifier.sde((x[i:i + 1] - 0.5) * 2, args.t, args.dt, seed=args.seed, sample0=r * b + i)
        # (tolerance, not equality: the fused resize/affine kernel and the torch spelling may differ in the last bit of the input,
        #  which under f16sr can flip an fp16 rounding; a WRONG sample index would differ by ~1e-1)
        assert ((one + 1) * 0.5 - per_img[r, i:i + 1]).abs().max() < 1e-4
        logits = model(pur, mode="classify")
        assert logits.shape == (reps * b, 7) and torch.equal(logits, model.resnet(pur))
        runner._calls = 0
        assert torch.equal(model(x.repeat(reps, 1, 1, 1), mode="purify_and_classify"), logits)
    with pytest.raises(NotImplementedError):
        model(x, mode="nonsense")


@pytest.mark.parametrize("precision", SHIPPED)
def test_ldsde_runner_autograd_on_gpu_vs_oracle_adjoint(tmp_path, precision):
    """LDGuidedDiffusion.image_editing_sample is differentiable w.r.t. the input on the HIP engine; dL/dx equals the oracle's
    restated stochastic adjoint (through the initial state) on the same injected noise."""
    from oracle import ncsnpp as on
    from oracle import solvers as osol
    from diffpure_amd import ncsnpp as pn
    from runners.diffpure_ldsde import LDGuidedDiffusion
    g = load_golden("ncsnpp_small.pt")

    def ns(d):
        n = argparse.Namespace()
        for k, v in d.items():
            setattr(n, k, ns(v) if isinstance(v, dict) else v)
        return n

    config = ns(g["cfg"])
    config.device = torch.device(DEV)
    args = argparse.Namespace(t=100, rand_t=False, t_delta=15, use_bm=False, sample_step=1, log_dir=str(tmp_path), score_type="score_sde",
                              seed=1234, synthetic_weights=True, sigma2=0.001, lambda_ld=0.01, eta=5, precision=precision)
    runner = LDGuidedDiffusion(args, config, device=config.device)
    sd = synth_state_dict(pn.param_shapes(pn.parse_config(g["cfg"])), 1234)
    score = osol.make_score_fn("ncsnpp", sd, on.parse_ncsnpp_config(g["cfg"]))
    gen = torch.Generator().manual_seed(4)
    x0 = g["x"]
    zs = [torch.randn(x0.shape, generator=gen) for _ in range(10)]
    cot = torch.randn(x0.shape, generator=gen)
    x = x0.clone().to(DEV).requires_grad_(True)
    out = runner.image_editing_sample(x, bs_id=9, noise=dict(z=zs))
    (gx,) = torch.autograd.grad((out * cot.to(DEV)).sum(), x)
    with torch.no_grad():
        xf = osol.ldsde_purify(score, x0, zs, 100, 0.001, 0.01, 5)
    ref = osol.ldsde_adjoint_grad(score, xf, cot, x0, zs, 100, 0.001, 0.01, 5)
    err_x = (out.detach().cpu() - xf).abs().max().item()
    print(f"ldsde runner + adjoint, small NCSN++ [{precision}]: x max-abs {err_x:.3e}, dL/dx rel. error {relerr(gx.cpu(), ref):.3e}")
    assert err_x < PIX_TOL[precision], err_x
    assert relerr(gx.cpu(), ref) < GRAD_TOL[precision], relerr(gx.cpu(), ref)


# ---- round 5: the adjoint under the SHIPPED arithmetic against the forward solve it differentiates ------------------------------
def _ncsnpp_full_f16sr():
    from diffpure_amd import ncsnpp as pn
    g = load_golden("ncsnpp_full.pt")
    cfg = pn.parse_config(g["cfg"])
    sd = synth_state_dict(pn.param_shapes(cfg), g["seed"])
    return pn.NCSNpp(cfg, DEV, "f16sr").load_state_dict(sd), g


def test_taped_and_untaped_forward_agree_under_f16sr():
    """The adjoint solves re-run the network TAPED.  Round 4 taped an fp32-stream variant (fp32 residual stream, three-pass attention,
    separate w2 / skip panels with their own stochastic-rounding draws) while the forward solve ran UNTAPED on the fp16 stream with
    fused [w2 | skip] panels (advisor finding): the two agreed up to rounding noise only.  Round 5 tapes the fp16 stream itself: same
    re-rounding key, same input -> the two evaluations of eps are EQUAL bit for bit, i.e. the adjoint differentiates exactly the network
    the forward solve evaluated."""
    net, g = _ncsnpp_full_f16sr()
    x, lab = nhwc(g["x"]).to(DEV), g["labels"].to(DEV)
    net.reround(7)
    a = net.forward(x, lab).float().cpu()
    net.reround(7)
    tape = []
    b = net.forward(x, lab, tape=tape).float().cpu()
    del tape
    ref = nhwc(g["out"])
    scale = ref.abs().max().item()
    d_ab, d_a, d_b = (a - b).abs().max().item(), (a - ref).abs().max().item(), (b - ref).abs().max().item()
    print(f"f16sr eps: untaped vs taped max-abs {d_ab:.3e} (mean {(a - b).abs().mean():.3e}); vs the reference module: untaped {d_a:.3e}, "
          f"taped {d_b:.3e}; largest entry {scale:.3f}")
    assert torch.equal(a, b), d_ab
    assert d_a < 1e-2 * scale, (d_a, scale)


def test_ode_vjp_f16sr_directional_derivative_vs_finite_differences_of_the_f16sr_forward_solve():
    """`ode_vjp` re-rounds the weights of adjoint step k with the key of forward step N-1-k (sde.py: the interval it re-crosses).  What
    that must deliver is the derivative of the forward solve AS SHIPPED (f16sr: fp16 stream, stochastic weight rounding keyed by the
    step).  Checked directly: L(x0) = <cot, ode(x0)> with the forward noise fixed; the central difference of L along the unit
    direction d = grad / |grad| (|eps d|_2 = 0.5: ~6e-3 per pixel, far above the fp16 rounding noise of L, far below the curvature
    scale) against <grad, d> = |grad| from the adjoint.  20 Euler steps of the product's dt = 1e-3 on the full NCSN++, B=2.  The
    continuous adjoint (optimise-then-discretise, as torchdiffeq's) differs from the exact gradient of the discrete solve by
    O(dt): the bar is 2 %."""
    from diffpure_amd.sde import Purifier
    net, g = _ncsnpp_full_f16sr()
    pur = Purifier(net, "ncsnpp", DEV)
    gen = torch.Generator().manual_seed(11)
    x0 = g["x"]
    e = torch.randn(x0.shape, generator=gen)
    cot = torch.randn(x0.shape, generator=gen)
    t, step = 20, 1e-3
    solve = lambda xx: pur.ode(xx, t, step, noise=dict(e=e, z=[])).cpu()
    xf = pur.ode(x0, t, step, noise=dict(e=e, z=[]))
    grad = (pur.ode_vjp(xf, cot, t, step) * pur.diffuse_scale(t)).cpu().double()
    gn = grad.norm().item()
    d = (grad / gn).float()
    h = 0.5
    lp = (cot.double() * solve(x0 + h * d).double()).sum().item()
    lm = (cot.double() * solve(x0 - h * d).double()).sum().item()
    fd = (lp - lm) / (2 * h)
    h2 = 0.25
    fd2 = ((cot.double() * solve(x0 + h2 * d).double()).sum().item() - (cot.double() * solve(x0 - h2 * d).double()).sum().item()) / (2 * h2)
    print(f"f16sr ode_vjp: <grad, d> = |grad| = {gn:.5f}; central difference of the f16sr forward solve: {fd:.5f} (h=0.5), {fd2:.5f} (h=0.25); "
          f"relative gaps {abs(fd - gn) / gn:.3e} / {abs(fd2 - gn) / gn:.3e}")
    assert abs(fd - gn) < 2e-2 * gn and abs(fd2 - gn) < 2e-2 * gn, (fd, fd2, gn)
    # round 6 (advisor): the direction grad / |grad| alone cannot see a gradient that is an orthogonal PROJECTION of the true one (a
    # missing branch): two random unit directions with a component along the gradient of 1/2 each - <grad, r> is then half |grad| plus
    # whatever the orthogonal part contributes - against the central difference along them
    for sd_ in (21, 22):
        rdir = torch.randn(x0.shape, generator=torch.Generator().manual_seed(sd_)).double()
        rdir = rdir - (rdir * d.double()).sum() * d.double()
        rdir = (0.5 * d.double() + math.sqrt(0.75) * rdir / rdir.norm()).float()
        want = (grad * rdir.double()).sum().item()
        fdr = ((cot.double() * solve(x0 + h * rdir).double()).sum().item() - (cot.double() * solve(x0 - h * rdir).double()).sum().item()) / (2 * h)
        print(f"   probe direction (seed {sd_}): <grad, r> = {want:.5f}, central difference {fdr:.5f}, gap {abs(fdr - want) / gn:.3e} of |grad|")
        assert abs(fdr - want) < 2e-2 * gn, (fdr, want, gn)


def test_adjoint_of_a_tiny_cotangent_keeps_its_relative_accuracy_under_f16sr():
    """Round 6 (advisor, round 5): in the fp16 x fp16 modes the network VJP rounds gradient operands to plain fp16 (dgrad convolutions,
    dO / dP / dS of the attention backward).  A realistic attack cotangent (cross-entropy dL/dx: 1e-3 ... 1e-6 per pixel) would push dS -
    another factor 1/T smaller - below fp16's normal range and flush the dQ / dK terms with no error.  The adjoint solves are linear in the
    cotangent: Purifier.*_vjp normalise it by a power of two once per solve (exact in fp32) and scale the result back.  Checked: the
    gradient for 1e-6 x cot is 1e-6 x the gradient for cot to fp16-rounding accuracy (without the normalisation the attention terms
    underflow: the same comparison on the bare network VJP, printed, is the control)."""
    from diffpure_amd.sde import Purifier
    net, g = _ncsnpp_full_f16sr()
    pur = Purifier(net, "ncsnpp", DEV)
    gen = torch.Generator().manual_seed(5)
    x0 = g["x"]
    e = torch.randn(x0.shape, generator=gen)
    cot = torch.randn(x0.shape, generator=gen)
    t, step = 5, 1e-3
    xf = pur.ode(x0, t, step, noise=dict(e=e, z=[]))
    g1 = pur.ode_vjp(xf, cot, t, step).cpu().double()
    for tiny in (1e-6, 3e-9):
        g2 = pur.ode_vjp(xf, cot * tiny, t, step).cpu().double() / tiny
        rel = (g2 - g1).abs().max().item() / g1.abs().max().item()
        # control: one bare network VJP at the same scale, no normalisation
        tape = []
        net.reround(0)
        net.forward(nhwc(x0).to(DEV), g["labels"].to(DEV), tape=tape)
        v1 = net.vjp(tape, nhwc(cot).to(DEV)).cpu().double()
        v2 = net.vjp(tape, nhwc(cot * tiny).to(DEV)).cpu().double() / tiny
        del tape
        relc = (v2 - v1).abs().max().item() / v1.abs().max().item()
        print(f"cotangent x {tiny:g}: ode_vjp (normalised) deviates {rel:.3e} of the largest entry from the unit-scale gradient; bare network VJP {relc:.3e}")
        assert rel < 2e-3, (tiny, rel)
    # the stochastic adjoint goes through the same normalisation
    zs = [torch.randn(x0.shape, generator=gen) for _ in range(t)]
    xs = pur.sde(x0, t, step, noise=dict(e=e, z=zs))
    s1 = pur.sde_vjp(xs, cot, t, step, noise=dict(e=e, z=zs)).cpu().double()
    s2 = pur.sde_vjp(xs, cot * 1e-6, t, step, noise=dict(e=e, z=zs)).cpu().double() / 1e-6
    rel = (s2 - s1).abs().max().item() / s1.abs().max().item()
    print(f"sde_vjp, cotangent x 1e-6: {rel:.3e}")
    assert rel < 2e-3, rel


def test_guided_taped_and_untaped_forward_agree_under_f16sr():
    """the same equality for the guided UNet (FiLM ResBlocks, multi-head attention, fused [w2 | skip] panels, resampled identity skips)"""
    from diffpure_amd import guided_unet as pg
    g = load_golden("guided_small.pt")
    cfg = pg.parse_config(g["cfg"])
    net = pg.GuidedUNet(cfg, DEV, "f16sr").load_state_dict(synth_state_dict(pg.param_shapes(cfg), g["seed"]))
    x, t = nhwc(g["x"]).to(DEV), g["t"].float().to(DEV)
    net.reround(3)
    a = net.forward(x, t).float().cpu()
    net.reround(3)
    tape = []
    b = net.forward(x, t, tape=tape).float().cpu()
    del tape
    assert torch.equal(a, b), (a - b).abs().max().item()
