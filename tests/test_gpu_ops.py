"""-m gpu: every HIP operator (through the C ABI) against its plain-PyTorch fp32 statement
(tests/refops.py, evaluated on CPU in fp32/fp64)."""
import math

import pytest
import torch

import refops

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    from diffpure_amd import _lib
    _lib.load()  # fail loudly if the HIP library is missing
    return torch.device("cuda:0")


def rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def close(got, ref, rtol, atol):
    torch.testing.assert_close(got.cpu(), ref, rtol=rtol, atol=atol)


CONV_CASES = [
    # B, H, W, C1, C2, N, k, bias, temb(rows), res, scale
    (2, 16, 16, 128, 0, 256, 3, True, 0, False, 1.0),       # main 128x128 tile path
    (2, 16, 16, 128, 64, 128, 3, True, 2, True, 0.70710678),  # channel-split input + temb per sample + residual
    (1, 5, 7, 132, 0, 72, 3, True, 1, False, 1.0),          # ragged M, Cin % 16 != 0 (k-cursor crosses taps mid-tile)
    (3, 8, 8, 20, 12, 40, 3, False, 0, True, 1.0),           # small Cin, C1 % 16 != 0 split
    (2, 32, 32, 3, 0, 128, 3, True, 0, False, 1.0),          # stem: Cin = 3 scalar loader
    (2, 16, 16, 128, 0, 6, 3, True, 0, False, 1.0),          # head: N = 6 (ldw 8), 128x32 tile
    (2, 16, 16, 128, 0, 3, 3, True, 0, False, 1.0),          # head: N = 3 (ldw 4)
    (4, 4, 4, 256, 256, 256, 3, True, 1, True, 0.70710678),  # low resolution, few tiles -> 64x64 tiles
    (2, 16, 16, 256, 0, 768, 1, True, 0, False, 1.0),        # 1x1 (qkv)
    (2, 8, 8, 384, 128, 256, 1, True, 0, False, 1.0),        # 1x1 skip over a split input
    (1, 1, 1, 512, 0, 1000, 1, True, 0, False, 1.0),         # linear with M = 1 (time table)
    (2, 64, 64, 64, 0, 64, 3, False, 0, False, 1.0),         # larger M
]


@pytest.mark.parametrize("case", CONV_CASES, ids=[str(c) for c in CONV_CASES])
def test_conv2d(dev, case):
    from diffpure_amd import ops
    B, H, W, C1, C2, N, k, has_bias, temb_rows, has_res, scale = case
    x = rnd(B, H, W, C1, seed=1)
    x2 = rnd(B, H, W, C2, seed=2) if C2 else None
    w = rnd(N, C1 + C2, k, k, seed=3, scale=1.0 / math.sqrt((C1 + C2) * k * k))
    wp = ops.pack_conv_weight(w)
    bias = rnd(N, seed=4) if has_bias else None
    temb = None
    if temb_rows:
        table = rnd(B if temb_rows == 2 else 1, N + 8, seed=5)
        temb = table[:, 4:4 + N]  # column view of a wider table, as the engines pass it
    res = rnd(B, H, W, N, seed=6) if has_res else None
    ref = refops.conv2d(x.double(), wp.double(), N, k, None if bias is None else bias.double(),
                        None if x2 is None else x2.double(), None if temb is None else temb.double(),
                        None if res is None else res.double(), scale).float()
    d = lambda t: None if t is None else t.to(dev)
    table_d = None if temb is None else table.to(dev)
    temb_d = None if temb is None else table_d[:, 4:4 + N]
    got = ops.conv2d(d(x), d(wp), N, k, bias=d(bias), x2=d(x2), temb=temb_d, res=d(res), scale=scale)
    close(got, ref, rtol=1e-4, atol=2e-5)


def test_conv_is_an_exact_fp32_fma_chain(dev):
    """A = I check with an asymmetric weight panel: catches any transposed MFMA fragment mapping."""
    from diffpure_amd import ops
    C = 64
    x = torch.zeros(1, 8, 8, C)
    for p in range(64):
        x[0, p // 8, p % 8, p] = 1.0          # pixel p carries unit vector e_p
    w = torch.arange(C * C, dtype=torch.float32).reshape(C, C) / 7.0   # w[out, in], asymmetric
    got = ops.conv2d(x.to(dev), ops.pack_conv_weight(w).to(dev), C, 1).cpu()
    assert torch.equal(got.reshape(64, C), w.t().contiguous())  # out[p, n] = w[n, p] exactly


GN_CASES = [
    # B, H, W, C1, C2, G, film_rows, act, resample
    (2, 16, 16, 128, 0, 32, 0, True, 0),
    (2, 8, 8, 256, 128, 32, 0, True, 0),      # split input, groups straddle nothing (384/32 = 12)
    (2, 8, 8, 1024, 512, 32, 0, True, 0),     # C4 = 384 threads, group of 48 straddles the split
    (1, 4, 4, 2048, 0, 32, 1, True, 0),       # 512 threads, broadcast FiLM
    (3, 8, 8, 256, 0, 32, 2, True, 0),        # per-sample FiLM
    (2, 8, 8, 128, 0, 32, 0, True, 1),        # + nearest x2
    (2, 8, 8, 128, 0, 32, 0, True, 2),        # + mean 2x2
    (2, 16, 16, 32, 0, 8, 0, False, 0),       # tiny C (NCSN++ small), no activation
    (1, 64, 64, 256, 0, 32, 0, True, 0),      # many pixel slabs
]


@pytest.mark.parametrize("case", GN_CASES, ids=[str(c) for c in GN_CASES])
def test_group_norm(dev, case):
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
    ref = refops.group_norm(x.double(), G, eps, gamma.double(), beta.double(), None if x2 is None else x2.double(),
                            None if film is None else (film[0].double(), film[1].double()), act, rs).float()
    d = lambda t: None if t is None else t.to(dev)
    film_d = None
    if film is not None:
        tab_d = tab.to(dev)
        film_d = (tab_d[:, :C], tab_d[:, C:])
    got = ops.group_norm(d(x), G, eps, d(gamma), d(beta), x2=d(x2), film=film_d, act=act, resample=rs)
    close(got, ref, rtol=2e-5, atol=2e-5)
    st = ops.group_norm_stats(d(x), G, eps, d(x2))
    close(st, refops.group_norm_stats(x, G, eps, x2), rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("mode", [1, 2])
def test_resample(dev, mode):
    from diffpure_amd import ops
    x = rnd(2, 8, 8, 64, seed=9)
    close(ops.resample(x.to(dev), mode), refops.resample(x, mode), rtol=0, atol=1e-7)


ATT_CASES = [(2, 16, 256, 1, "split"), (2, 256, 256, 1, "split"), (2, 64, 128, 2, "legacy"), (1, 1024, 128, 2, "legacy"),
             (2, 64, 128, 2, "split"), (3, 256, 256, 4, "legacy")]


@pytest.mark.parametrize("case", ATT_CASES, ids=[str(c) for c in ATT_CASES])
def test_attention(dev, case):
    from diffpure_amd import ops
    B, T, C, heads, layout = case
    qkv = rnd(B, T, 3 * C, seed=11)
    ref = refops.attention(qkv.double(), heads, layout).float()
    close(ops.attention(qkv.to(dev), heads, layout), ref, rtol=1e-4, atol=1e-5)
    # the three-kernel path (what the taped forward of the VJP uses) on every shape
    out3, probs = ops.attention(qkv.to(dev), heads, layout, return_probs=True)
    close(out3, ref, rtol=1e-4, atol=1e-5)
    assert probs.shape == (B * heads, T, T)


FUSED_ATT_CASES = [(2, 64, 128, 2, "legacy", 1.0), (3, 256, 256, 4, "legacy", 1.0), (1, 1024, 512, 8, "legacy", 1.0),
                   (2, 256, 128, 2, "split", 1.0), (2, 1024, 128, 2, "legacy", 8.0), (5, 64, 1024, 16, "legacy", 3.0),
                   # head dimension 256: NCSN++'s AttnBlockpp at 16x16 (one head of C = 256; layerspp.py:75-91), and 2 x 256 / long rows
                   (3, 256, 256, 1, "split", 1.0), (2, 256, 256, 1, "split", 6.0), (1, 512, 512, 2, "legacy", 1.0), (2, 128, 512, 2, "split", 2.0)]


@pytest.mark.parametrize("case", FUSED_ATT_CASES, ids=[str(c) for c in FUSED_ATT_CASES])
def test_attention_fused_flash_kernel(dev, case, monkeypatch):
    """csrc/attention.hip (scores never materialised, split-fp16 MFMA) against fp64 attention and against the
    three-kernel fp32 path; `gain` > 1 makes the logits large so that the running-max rescale is exercised."""
    from diffpure_amd import ops
    B, T, C, heads, layout, gain = case
    qkv = rnd(B, T, 3 * C, seed=11) * gain
    ref = refops.attention(qkv.double(), heads, layout).float()
    assert ops.attention_fused_ok(T, C // heads)
    got = ops.attention_fused(qkv.to(dev), heads, layout)
    # logits of magnitude ~gain^2 * sqrt(d): their 22-bit products (and the fp32 ones of the other path) carry an
    # absolute error ~gain^2 * 1e-5, which softmax turns into a relative error of the same size on outputs ~gain
    tol = 2e-5 * gain ** 3
    close(got, ref, rtol=1e-4, atol=tol)
    monkeypatch.setenv("DIFFPURE_ATTN_FUSED", "0")
    close(got, ops.attention(qkv.to(dev), heads, layout).cpu(), rtol=1e-4, atol=tol)
    assert torch.equal(got, ops.attention_fused(qkv.to(dev), heads, layout))      # deterministic


@pytest.mark.parametrize("case", FUSED_ATT_CASES, ids=[str(c) for c in FUSED_ATT_CASES])
def test_attention_fused_one_pass_fp16(dev, case):
    """The one-pass form of csrc/attention.hip (round 4; the fp16 x fp16 precision modes): qkv as the qkv convolution stores it - plain
    fp16 -, Q and K read in place, V^T packed, ONE fp16 MFMA pass per product, 1/sqrt(d) on the scores.  Against fp64 attention of the SAME
    fp16-rounded qkv (the error that is left is the fp16 rounding of the probabilities, 2^-11 relative, and fp32 accumulation), against
    the three-pass kernel on the same values, and the bordered-operand output against the plain one."""
    from diffpure_amd import ops
    B, T, C, heads, layout, gain = case
    q16 = (rnd(B, T, 3 * C, seed=11) * gain).half()
    ref = refops.attention(q16.double(), heads, layout).float()
    got = ops.attention_fused(q16.to(dev), heads, layout)
    assert got.dtype == torch.float32
    tol = 1.5e-3 * max(1.0, gain)           # |out| <~ max |v| ~ 4 gain; P rounded to fp16: relative 2^-11 per term, averaging over T keys
    close(got, ref, rtol=2e-3, atol=tol)
    three = ops.attention_fused(q16.float().to(dev), heads, layout)
    close(got, three.cpu(), rtol=2e-3, atol=tol)
    assert torch.equal(got, ops.attention_fused(q16.to(dev), heads, layout))      # deterministic
    err = (got.cpu() - ref).abs().max().item()
    print(f"one-pass fp16 attention {case}: max-abs error {err:.2e} (outputs up to {ref.abs().max().item():.2f})")
    d = C // heads
    ww = 16 if T % 16 == 0 else 8
    op = ops.attention_fused(q16.to(dev), heads, layout, operand_hw=(T // ww, ww))
    want = torch.nn.functional.pad(got.view(B, T // ww, ww, C), (0, 0, 1, 1, 1, 1)).half()
    diff = (op.float() - want.float()).abs()
    assert (diff <= want.float().abs() * 2.0 ** -10 + 1e-7).all() and (op != want).float().mean().item() < 1e-3


@pytest.mark.parametrize("cols", [64, 256, 1024, 200, 2048])
def test_softmax_rows_and_backward_register_forms(dev, cols):
    """Round 6: rows of 64 / 256 / 1 024 columns (the attention sizes of both networks) run the softmax and its backward with the row
    held in registers - one read, one write - on the lane -> column map and summation order of the three-sweep kernels (other widths
    still take those): against fp64, forward and dS = P (dP - sum dP P), incl. a row of large logits and a ragged row count."""
    from diffpure_amd import _lib
    rows = 37
    x = rnd(rows, cols, seed=3) * 3.0
    x[5] *= 40.0
    xd = x.to(dev).clone()
    s_ = torch.cuda.current_stream().cuda_stream
    _lib.call("dp_softmax_rows", xd.data_ptr(), rows, cols, s_)
    ref = torch.softmax(x.double(), -1)
    close(xd, ref.float(), rtol=1e-5, atol=1e-7)
    dp = rnd(rows, cols, seed=4)
    dpd = dp.to(dev).clone()
    _lib.call("dp_softmax_bwd_rows", xd.data_ptr(), dpd.data_ptr(), rows, cols, s_)
    p64 = xd.cpu().double()
    want = p64 * (dp.double() - (dp.double() * p64).sum(-1, keepdim=True))
    close(dpd, want.float(), rtol=1e-4, atol=1e-6)


def test_softmax_forced_large_logits(dev):
    from diffpure_amd import _lib
    x = rnd(37, 200, seed=12) * 30
    x[5, 17] = 500.0  # one dominating logit
    xd = x.to(dev).contiguous()
    _lib.call("dp_softmax_rows", xd.data_ptr(), 37, 200, torch.cuda.current_stream().cuda_stream)
    close(xd, torch.softmax(x.double(), -1).float(), rtol=1e-5, atol=1e-7)


def test_small_elementwise(dev):
    from diffpure_amd import ops
    x, y = rnd(5, 333, seed=13) * 4, rnd(5, 333, seed=14)
    close(ops.silu(x.to(dev)), refops.silu(x.double()).float(), rtol=1e-6, atol=1e-7)
    close(ops.axpby(x.to(dev), 0.9, y.to(dev), -0.3), x * 0.9 + y * -0.3, rtol=1e-6, atol=1e-6)
    freqs = torch.exp(-math.log(10000) * torch.arange(128, dtype=torch.float32) / 128)
    t = torch.tensor([0.0, 1.0, 37.0, 99.0, 999.0])
    for cf in (True, False):
        close(ops.timestep_embedding(t.to(dev), freqs.to(dev), cf), refops.timestep_embedding(t, freqs, cf), rtol=0, atol=2e-5)


def test_philox_matches_numpy_restatement_and_is_shard_invariant(dev):
    from diffpure_amd import ops
    shape = (3, 8, 8, 3)
    got = ops.philox_normal(shape, seed=0x1234ABCD5678, sample0=5, step=17, device=dev).cpu()
    ref = refops.philox_normal(shape, 0x1234ABCD5678, 5, 17)
    torch.testing.assert_close(got, ref, rtol=0, atol=3e-5)
    part = ops.philox_normal((1, 8, 8, 3), seed=0x1234ABCD5678, sample0=7, step=17, device=dev).cpu()
    assert torch.equal(part[0], got[2])
    big = ops.philox_normal((64, 32, 32, 3), seed=1, sample0=0, step=0, device=dev)
    assert abs(big.mean().item()) < 5e-3 and abs(big.std().item() - 1) < 5e-3


def test_em_step_injected_and_philox(dev):
    from diffpure_amd import ops
    x, eps6, z = rnd(2, 8, 8, 3, seed=20), rnd(2, 8, 8, 6, seed=21), rnd(2, 8, 8, 3, seed=22)
    args = dict(neg_half_beta=-1.045, gg=2.09, score_coef=-3.1, score_div=0, h=1e-3, g=1.4457, sqrt_h=0.0316228)
    ref = refops.em_step(x, eps6, noise=z, **args)
    close(ops.em_step(x.to(dev), eps6.to(dev), noise=z.to(dev), **args), ref, rtol=1e-6, atol=1e-6)
    args["score_div"], args["score_coef"] = 1, 0.322
    ref = refops.em_step(x, eps6, seed=99, sample0=4, step=3, **args)
    close(ops.em_step(x.to(dev), eps6.to(dev), seed=99, sample0=4, step=3, **args), ref, rtol=1e-5, atol=2e-6)
    args["g"] = 0.0
    close(ops.em_step(x.to(dev), eps6.to(dev), **args), refops.em_step(x, eps6, **args), rtol=1e-6, atol=1e-6)


def test_ddpm_step(dev):
    from diffpure_amd import ops
    x, o6, z = rnd(2, 8, 8, 3, seed=30), rnd(2, 8, 8, 6, seed=31), rnd(2, 8, 8, 3, seed=32)
    a = dict(sr=1.02, srm1=0.2, c1=0.3, c2=0.69, min_log=-9.0, max_log=-6.0)
    for nz in (True, False):
        ref = refops.ddpm_step(x, o6, nonzero=nz, noise=z, **a)
        close(ops.ddpm_step(x.to(dev), o6.to(dev), nonzero=nz, noise=z.to(dev), **a), ref, rtol=1e-5, atol=1e-6)


def test_ops_refuse_cpu_tensors(dev):
    from diffpure_amd import _lib, ops
    with pytest.raises(_lib.DiffpureHipError):
        ops.silu(torch.zeros(4))


H2_CASES = [
    # B, H, W, C, N, k, temb_rows, res, scale
    (2, 16, 16, 128, 256, 3, 0, False, 1.0),           # 128x128 tiles
    (2, 16, 16, 384, 128, 3, 2, True, 0.70710678),     # temb rows, residual, scale
    (1, 5, 7, 64, 96, 3, 1, False, 1.0),               # ragged M and N -> 64x64 tiles, masked rows/cols
    (4, 4, 4, 512, 256, 3, 0, True, 1.0),              # low resolution
    (2, 16, 16, 128, 6, 3, 0, False, 1.0),             # head: N = 6
    (2, 16, 16, 256, 768, 1, 0, False, 1.0),           # 1x1 qkv
    (2, 32, 32, 32, 32, 3, 0, False, 1.0),             # Cin = 32: one channel slice
    (1, 64, 64, 64, 192, 3, 0, False, 1.0),            # larger M, 128x128 with N not a multiple of 128
    (3, 9, 9, 96, 160, 3, 0, False, 1.0),              # tile tail rows re-read the last pixel (M % 128 != 0)
    (2, 160, 160, 64, 256, 3, 2, True, 0.70710678),    # M = 51200 (% 256 == 0), N = 256: 200 tiles of 256 x 256 -> 128x128 tiles
    (2, 160, 160, 32, 128, 3, 0, False, 1.0),          # M = 51200 (% 512 == 0), N = 128: 100 tiles of 512 x 128 -> 128x128 tiles
    (3, 130, 130, 32, 96, 3, 1, True, 1.0),            # ragged M (50700) and N < BN
    (2, 160, 160, 64, 512, 1, 0, False, 1.0),          # 1x1, four n-tiles
]


def _h2_bordered(x, dev):
    """fp32 [B,H,W,C] -> zero-bordered h2 [B,H+2,W+2,2C] on the device via the device packer."""
    from diffpure_amd import ops
    B, H, W, C = x.shape
    xp = torch.nn.functional.pad(x, (0, 0, 1, 1, 1, 1)).contiguous()
    return ops.pack_h2(xp.reshape(-1, C).to(dev)).reshape(B, H + 2, W + 2, 2 * C)


# the last four shapes are large but do not fill the chip with ping-pong tiles: they run on <128,128,32>
@pytest.mark.parametrize("case", H2_CASES, ids=[str(c) for c in H2_CASES])
def test_conv2d_h2_split_fp16(dev, case):
    from diffpure_amd import ops
    B, H, W, C, N, k, temb_rows, has_res, scale = case
    x = rnd(B, H, W, C, seed=1)
    w = rnd(N, C, k, k, seed=3, scale=1.0 / math.sqrt(C * k * k))
    bias = rnd(N, seed=4)
    table = rnd(B if temb_rows == 2 else 1, N + 8, seed=5) if temb_rows else None
    res = rnd(B, H, W, N, seed=6) if has_res else None
    # exact fp64 convolution of the fp32 operands: the f16x3 path must be fp32-class accurate
    ref = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w.double(), bias.double(), padding=k // 2).permute(0, 2, 3, 1)
    if table is not None:
        ref = ref + table[:, 4:4 + N].double().reshape(-1, 1, 1, N)
    if res is not None:
        ref = ref + res.double()
    ref = (ref * scale).float()
    xh = _h2_bordered(x, dev)
    wh = ops.pack_conv_weight_h2(w, dev)
    assert torch.equal(xh.cpu(), refops.h2_encode(torch.nn.functional.pad(x, (0, 0, 1, 1, 1, 1))))   # packer == format statement
    table_d = None if table is None else table.to(dev)
    got = ops.conv2d_h2(xh, wh, N, k, bias=bias.to(dev), temb=None if table is None else table_d[:, 4:4 + N],
                        res=None if res is None else res.to(dev), scale=scale)
    close(got, ref, rtol=2e-5, atol=2e-5)
    # and the torch statement of the same contract agrees with it
    close(got, refops.conv2d_h2(xh.cpu(), wh.cpu(), N, k, bias, None if table is None else table[:, 4:4 + N], res, scale),
          rtol=2e-5, atol=2e-5)


PP_CASES = [
    # B, H, W, C, N, k, temb_rows, res, scale      (M % 256 == 0 and N % 256 == 0: the 256x256 ping-pong variant applies)
    (2, 16, 16, 128, 256, 3, 2, True, 0.70710678),     # 2 tiles, 36 k-tiles, every epilogue term
    (1, 16, 16, 32, 256, 3, 0, False, 1.0),            # 9 k-tiles (one channel slice)
    (1, 16, 16, 64, 512, 1, 0, False, 1.0),            # 1x1: 2 k-tiles, two n-tiles
    (1, 16, 16, 32, 256, 1, 1, False, 1.0),            # a single k-tile (prologue only)
    (1, 16, 16, 96, 256, 1, 0, True, 1.0),             # 3 k-tiles (all of them in the drained tail)
    (4, 32, 32, 256, 512, 3, 2, True, 1.0),            # 32 tiles, several tiles per sample, 72 k-tiles
    # N = 128 -> the 512x128 tiling of the same schedule (M % 512 == 0)
    (2, 16, 16, 128, 128, 3, 2, True, 0.70710678),     # one tile, 36 k-tiles, every epilogue term
    (1, 32, 16, 32, 128, 1, 1, False, 1.0),            # a single k-tile
    (8, 32, 32, 128, 384, 3, 2, True, 1.0),            # 16 x 3 tiles (NCSN++ 32x32 level shape, N = 3 x 128)
    (2, 32, 32, 96, 128, 3, 0, False, 1.0),            # 27 k-tiles
]


@pytest.mark.parametrize("case", PP_CASES, ids=[str(c) for c in PP_CASES])
def test_conv2d_h2_pingpong_variant_is_bit_identical(dev, case, tune):
    """The 8-wave 256x256 variant (igemm_h2_pp.hip) against the fp64 convolution AND bit-for-bit against the
    128x128 / 64x64 variants, column-sum records included; repeated launches screen for LDS-DMA races."""
    from diffpure_amd import ops
    B, H, W, C, N, k, temb_rows, has_res, scale = case
    x = rnd(B, H, W, C, seed=1)
    w = rnd(N, C, k, k, seed=3, scale=1.0 / math.sqrt(C * k * k))
    bias = rnd(N, seed=4).to(dev)
    table = rnd(B if temb_rows == 2 else 1, N + 8, seed=5).to(dev) if temb_rows else None
    res = rnd(B, H, W, N, seed=6).to(dev) if has_res else None
    ref = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w.double(), bias.cpu().double(), padding=k // 2).permute(0, 2, 3, 1)
    if table is not None:
        ref = ref + table.cpu()[:, 4:4 + N].double().reshape(-1, 1, 1, N)
    if res is not None:
        ref = ref + res.cpu().double()
    ref = (ref * scale).float()
    xh, wh = _h2_bordered(x, dev), ops.pack_conv_weight_h2(w, dev)

    def run():
        y = ops.conv2d_h2(xh, wh, N, k, bias=bias, temb=None if table is None else table[:, 4:4 + N], res=res, scale=scale,
                          colstats=True)
        return y.t, y.cols.buf.clone()

    tune.setenv("DP_H2_PP", "0")
    base, base_cs = run()
    tune.setenv("DP_H2_PP", "1")
    for _ in range(6):
        got, got_cs = run()
        assert torch.equal(got, base)
        assert torch.equal(got_cs, base_cs)
    close(got, ref, rtol=2e-5, atol=2e-5)


def _h1_bordered(x, dev):
    """fp32 [B,H,W,C] -> zero-bordered plain-fp16 operand [B,H+2,W+2,C] ("h1")."""
    return torch.nn.functional.pad(x, (0, 0, 1, 1, 1, 1)).half().contiguous().to(dev)


H1_CASES = H2_CASES + PP_CASES + [(2, 32, 32, 256, 256, 3, 2, True, 1.0), (8, 64, 64, 128, 256, 3, 0, False, 1.0),
                                  (8, 64, 64, 64, 128, 3, 1, True, 1.0),       # these two fill the chip: ping-pong by default
                                  # halo-tile variant of the fp16 x fp16 kernel (igemm_h2_halo.hip): W = 128 / 256 / 64 / 16, one to
                                  # three channel slices (halo buffer swaps), several tile rows per image, images per tile boundary
                                  (1, 128, 128, 64, 256, 3, 1, True, 0.5), (1, 256, 256, 32, 256, 3, 0, False, 1.0),
                                  (2, 256, 256, 64, 256, 3, 2, True, 1.0), (3, 64, 64, 96, 512, 3, 0, True, 1.0),
                                  (5, 16, 16, 160, 256, 3, 2, False, 1.0),
                                  # one / several image rows per 256-pixel tile, 1 .. 3 channel slices, two column tiles, a wide image
                                  (1, 256, 256, 96, 512, 3, 1, True, 1.0), (4, 32, 32, 32, 256, 3, 2, True, 0.5), (1, 8, 512, 64, 256, 3, 0, False, 1.0),
                                  # > 256 tiles, a ragged last round (600 tiles of 256x256; 300 of 512x128)
                                  (3, 160, 320, 64, 256, 3, 2, True, 0.5), (3, 160, 320, 32, 128, 3, 1, False, 1.0)]


@pytest.mark.parametrize("passes", [2])
@pytest.mark.parametrize("case", H1_CASES, ids=[str(c) for c in H1_CASES])
def test_conv2d_h1_fp16_activations(dev, case, passes, tune):
    """Plain-fp16 activation operand ("h1") x split-fp16 weights, passes=2 ("f16x2"): must equal the exact convolution of
    the fp16-ROUNDED activations with the full weights to fp32-class accuracy; every tile variant gives the same bits
    (column-sum records included).  (One pass on hi|lo panels - round 2's "f16" - is gone since ABI 6: one pass means plain
    fp16 panels, test_conv2d_fp16_weights_single_pass.)"""
    from diffpure_amd import ops
    B, H, W, C, N, k, temb_rows, has_res, scale = case
    x = rnd(B, H, W, C, seed=1)
    w = rnd(N, C, k, k, seed=3, scale=1.0 / math.sqrt(C * k * k))
    bias = rnd(N, seed=4).to(dev)
    table = rnd(B if temb_rows == 2 else 1, N + 8, seed=5).to(dev) if temb_rows else None
    res = rnd(B, H, W, N, seed=6).to(dev) if has_res else None
    wr = w.half().double() if passes == 1 else w.double()
    ref = torch.nn.functional.conv2d(x.half().double().permute(0, 3, 1, 2), wr, bias.cpu().double(), padding=k // 2).permute(0, 2, 3, 1)
    if table is not None:
        ref = ref + table.cpu()[:, 4:4 + N].double().reshape(-1, 1, 1, N)
    if res is not None:
        ref = ref + res.cpu().double()
    ref = (ref * scale).float()
    xh, wh = _h1_bordered(x, dev), ops.pack_conv_weight_h2(w, dev)

    def run():
        y = ops.conv2d_h2(xh, wh, N, k, bias=bias, temb=None if table is None else table[:, 4:4 + N], res=res, scale=scale,
                          colstats=True, passes=passes)
        return y.t, y.cols.buf.clone()

    tune.setenv("DP_H2_PP", "0")
    base, base_cs = run()
    close(base, ref, rtol=2e-5, atol=2e-5)
    if B * H * W % 256 == 0 and N % 128 == 0 and not (H * W <= 64):
        tune.setenv("DP_H2_PP", "1")
        for _ in range(4):
            got, got_cs = run()
            assert torch.equal(got, base)
            assert torch.equal(got_cs, base_cs)
    tune.delenv("DP_H2_PP")
    got, got_cs = run()                      # the dispatcher's own choice
    assert torch.equal(got, base) and torch.equal(got_cs, base_cs)


@pytest.mark.parametrize("case", H1_CASES, ids=[str(c) for c in H1_CASES])
def test_conv2d_fp16_weights_single_pass(dev, case, tune):
    """Plain fp16 activations x plain fp16 weights (w_fmt 1, one MFMA pass - "f16" / "f16sr"): equals the exact
    convolution of the two fp16-rounded operands to fp32-class accuracy; every tile variant gives the same bits."""
    from diffpure_amd import ops
    B, H, W, C, N, k, temb_rows, has_res, scale = case
    x = rnd(B, H, W, C, seed=1)
    w = rnd(N, C, k, k, seed=3, scale=1.0 / math.sqrt(C * k * k))
    bias = rnd(N, seed=4).to(dev)
    table = rnd(B if temb_rows == 2 else 1, N + 8, seed=5).to(dev) if temb_rows else None
    res = rnd(B, H, W, N, seed=6).to(dev) if has_res else None
    ref = torch.nn.functional.conv2d(x.half().double().permute(0, 3, 1, 2), w.half().double(), bias.cpu().double(), padding=k // 2).permute(0, 2, 3, 1)
    if table is not None:
        ref = ref + table.cpu()[:, 4:4 + N].double().reshape(-1, 1, 1, N)
    if res is not None:
        ref = ref + res.cpu().double()
    ref = (ref * scale).float()
    xh = _h1_bordered(x, dev)
    w16 = ops.order_conv_weight_w16(w).half().to(dev)           # plain fp16 panel in the kernels' block layout
    assert torch.equal(ops.unorder_conv_weight_w16(w16.cpu(), N), ops.order_conv_weight_h2(w).half())

    def run():
        y = ops.conv2d_h2(xh, w16, N, k, bias=bias, temb=None if table is None else table[:, 4:4 + N], res=res, scale=scale,
                          colstats=True, w_fmt=1)
        return y.t, y.cols.buf.clone()

    tune.setenv("DP_H2_PP", "0")
    base, base_cs = run()
    close(base, ref, rtol=2e-5, atol=2e-5)
    # every 256x256 variant: ping-pong (DP_H2_SW=0), one-wave-per-SIMD software-pipelined (DP_H2_SW=1), both with the 8-wave kernel off
    if B * H * W % 256 == 0 and N % 256 == 0 and not (H * W <= 64):
        tune.setenv("DP_H2_PP", "1")
        tune.setenv("DP_H2_DW", "0")
        for sw in ("0", "1"):
            tune.setenv("DP_H2_SW", sw)
            for _ in range(3):
                got, got_cs = run()
                assert torch.equal(got, base), sw
                assert torch.equal(got_cs, base_cs), sw
        for name in ("DP_H2_SW", "DP_H2_DW"):
            tune.delenv(name)
        tune.setenv("DP_H2_PP", "0")
    # the 512x128 form of the one-wave-per-SIMD kernel (layers with 128 output channels; DP_H2_SW=2)
    if B * H * W % 512 == 0 and N % 128 == 0 and not (H * W <= 64):
        tune.setenv("DP_H2_PP", "1")
        tune.setenv("DP_H2_DW", "0")
        tune.setenv("DP_H2_SW", "2")
        for _ in range(3):
            got, got_cs = run()
            assert torch.equal(got, base), "sw 512x128"
            assert torch.equal(got_cs, base_cs), "sw 512x128"
        for name in ("DP_H2_SW", "DP_H2_DW"):
            tune.delenv(name)
        tune.setenv("DP_H2_PP", "0")
    # the 8-wave kernel (igemm_h2_dw.hip: one workgroup per CU on 256x256 tiles, two free-running waves per SIMD, the older wave of every
    # SIMD staging the rows of both); taken only where the launch has >= 256 tiles, elsewhere the other variants run
    if B * H * W % 256 == 0 and N % 256 == 0 and not (H * W <= 64) and C * k * k >= 128:
        tune.setenv("DP_H2_DW", "1")
        tune.setenv("DP_H2_PP", "1")
        for _ in range(3):
            got, got_cs = run()
            assert torch.equal(got, base), "dw8"
            assert torch.equal(got_cs, base_cs), "dw8"
        tune.delenv("DP_H2_DW")
        tune.setenv("DP_H2_PP", "0")
    if B * H * W % 256 == 0 and N % 128 == 0 and not (H * W <= 64):
        tune.setenv("DP_H2_PP", "1")
        for _ in range(4):
            got, got_cs = run()
            assert torch.equal(got, base)
            assert torch.equal(got_cs, base_cs)
    tune.delenv("DP_H2_PP")
    got, got_cs = run()
    assert torch.equal(got, base) and torch.equal(got_cs, base_cs)


def test_weight_pool_side_stream_prefetch_gives_the_in_stream_bits(dev, monkeypatch):
    """Round 6 (opt-in, DIFFPURE_ROUND_PREFETCH=1; measured slower and off by default): the next key's stochastic rounding on a side
    stream into a second buffer, the bound parameter-dict entries re-pointed at every flip - ascending keys, a descending run (the
    adjoints), a repeated key and a jump all give the panels of the one-buffer pool, and a convolution queued right after round() reads
    the new panels."""
    from diffpure_amd import ops
    w1, w2 = rnd(64, 32, 3, 3, seed=1, scale=0.05), rnd(96, 64, 1, 1, seed=2, scale=0.1)

    def build(prefetch):
        monkeypatch.setenv("DIFFPURE_ROUND_PREFETCH", "1" if prefetch else "0")
        pool = ops.WeightPool(dev, stochastic=True, seed=7)
        pool.add("a", w1)
        pool.add("b", w2)
        pool.finalize()
        table = {}
        pool.bind(table, "wa", "a")
        pool.bind(table, "wb", "b")
        return pool, table

    ref, rt = build(False)
    pre, pt = build(True)
    assert pre._other is not None and ref._other is None
    x = torch.nn.functional.pad(rnd(2, 8, 8, 32, seed=3), (0, 0, 1, 1, 1, 1)).half().to(dev)
    for key in [0, 1, 2, 3, 4, 4, 9, 8, 7, 6, 2, 3]:
        ref.round(key)
        pre.round(key)
        assert torch.equal(pt["wa"], rt["wa"]) and torch.equal(pt["wb"], rt["wb"]), key
        assert torch.equal(ops.conv2d_h2(x, pt["wa"], 64, 3, w_fmt=1), ops.conv2d_h2(x, rt["wa"], 64, 3, w_fmt=1)), key
    torch.cuda.synchronize()


def test_round_weights_nearest_and_stochastic(dev):
    """dp_round_weights: round-to-nearest equals torch's .half(); stochastic rounding returns one of the two fp16
    neighbours, is unbiased (the mean over many keys converges to the fp32 value), reproducible per (seed, key), and
    independent between keys - through the fp16 subnormal range and at exactly representable values as well."""
    from diffpure_amd import ops
    g = torch.Generator().manual_seed(5)
    w = torch.cat([torch.randn(4096, generator=g) * 0.02, torch.randn(1024, generator=g) * 3e-5, torch.randn(1024, generator=g) * 2e-7,
                   torch.tensor([0.0, -0.0, 1.0, -2.5, 6.1035e-05, 65504.0, -65504.0, 1e6]), torch.randn(2040, generator=g) * 30.0])
    pool = ops.WeightPool(dev, stochastic=False)
    pool.master, pool.work = w.to(dev), torch.empty(w.numel(), dtype=torch.float16, device=dev)
    pool.round(0)
    wc = w.clamp(-65504, 65504)
    assert torch.equal(pool.work.cpu()[:-2040 - 1], w.half()[:-2040 - 1])          # (1e6 -> inf under RTN, as torch)
    sr = ops.WeightPool(dev, stochastic=True, seed=99)
    sr.master, sr.work = w.to(dev), torch.empty(w.numel(), dtype=torch.float16, device=dev)
    lo = torch.where(w.half().float().abs() > wc.abs(), torch.nextafter(w.half(), torch.zeros(()).half()), w.half())   # towards zero
    lo = torch.where(lo.float().abs() > 65504, torch.sign(w).half() * 65504, lo)
    hi = torch.nextafter(lo, (torch.sign(w) * float("inf")).half())
    acc = torch.zeros_like(w, dtype=torch.float64)
    n_keys = 400
    first = None
    for key in range(n_keys):
        sr.round(key)
        got = sr.work.cpu()
        if key == 0:
            first = got.clone()
        ok = (got == lo) | (got == hi)
        assert ok.all(), (w[~ok][:4], got[~ok][:4], lo[~ok][:4], hi[~ok][:4])
        acc += got.double()
    mean = (acc / n_keys).float()
    ulp = (hi.float() - lo.float()).abs().clamp(max=64.0)
    exact = lo.float() == wc
    assert torch.equal(mean[exact], wc[exact])                                      # representable values never move
    assert ((mean - wc).abs() <= 0.12 * ulp + 1e-12)[~exact & (w.abs() < 65504)].all()   # |bias| well under an ulp / sqrt(n)
    sr.round(0)
    assert torch.equal(sr.work.cpu(), first)                                        # keyed: reproducible
    sr.round(1)
    assert 0.2 < (sr.work.cpu() != first)[~exact].float().mean() < 0.8              # and independent between keys


@pytest.mark.parametrize("mode", [3, 4])
def test_fir_resampling_upfirdn2d(dev, mode):
    """`fir: True` resampling (upfirdn2d with the separable [1,3,3,1] filter) inside GroupNorm-apply: against the
    reference's own upsample_2d / downsample_2d (golden fir_ops.pt), and - fused behind GroupNorm + SiLU, every output
    format, channel-split input - against the torch statement of the same operator."""
    from conftest import load_golden
    from diffpure_amd import ops
    g = load_golden("fir_ops.pt")
    taps = ops.fir_taps(g["k"])
    x = g["x"].permute(0, 2, 3, 1).contiguous()
    got = ops.resample(x.to(dev), mode, fir=taps).cpu().permute(0, 3, 1, 2)
    close(got, g["up"] if mode == 3 else g["down"], rtol=1e-5, atol=1e-6)
    x1, x2 = (rnd(2, 8, 12, 96, seed=1) * 2 + 0.5).to(dev), rnd(2, 8, 12, 32, seed=2).to(dev)
    gamma, beta = (1 + 0.1 * rnd(128, seed=3)).to(dev), (0.1 * rnd(128, seed=4)).to(dev)
    ref = refops.group_norm(x1.cpu(), 32, 1e-6, gamma.cpu(), beta.cpu(), x2=x2.cpu(), act=True, resample=mode, fir=taps)
    y32 = ops.group_norm(x1, 32, 1e-6, gamma, beta, x2=x2, act=True, resample=mode, fir=taps)
    close(y32, ref, rtol=2e-5, atol=2e-5)
    pad = torch.nn.functional.pad(y32.cpu(), (0, 0, 1, 1, 1, 1))
    assert torch.equal(ops.group_norm(x1, 32, 1e-6, gamma, beta, x2=x2, act=True, resample=mode, fir=taps, split="h1").cpu(), pad.half())
    assert torch.equal(ops.group_norm(x1, 32, 1e-6, gamma, beta, x2=x2, act=True, resample=mode, fir=taps, split="h2").cpu(), refops.h2_encode(pad))
    assert torch.equal(ops.to_h2(x1, mode, fmt="h1", fir=taps).cpu(), torch.nn.functional.pad(ops.resample(x1, mode, fir=taps).cpu(), (0, 0, 1, 1, 1, 1)).half())


def test_group_norm_h1_output_is_the_fp16_rounding_of_the_fp32_output(dev):
    from diffpure_amd import ops
    x = (rnd(2, 8, 8, 256, seed=1) * 2 + 0.5).to(dev)
    x2 = rnd(2, 8, 8, 128, seed=2).to(dev)
    gamma, beta = (1 + 0.1 * rnd(384, seed=3)).to(dev), (0.1 * rnd(384, seed=4)).to(dev)
    for rs in (0, 1, 2):
        y32 = ops.group_norm(x, 32, 1e-5, gamma, beta, x2=x2, act=True, resample=rs)
        yh = ops.group_norm(x, 32, 1e-5, gamma, beta, x2=x2, act=True, resample=rs, split="h1")
        assert yh.dtype == torch.float16 and yh.shape == (2, y32.shape[1] + 2, y32.shape[2] + 2, 384)
        assert torch.equal(yh.cpu(), torch.nn.functional.pad(y32.cpu(), (0, 0, 1, 1, 1, 1)).half())
        assert torch.equal(ops.to_h2(x, rs, fmt="h1").cpu(), torch.nn.functional.pad(ops.resample(x, rs).cpu() if rs else x.cpu(), (0, 0, 1, 1, 1, 1)).half())
    y, yr = ops.group_norm(x, 32, 1e-5, gamma, beta, x2=x2, act=True, split="h1", raw=True)
    assert torch.equal(y, ops.group_norm(x, 32, 1e-5, gamma, beta, x2=x2, act=True, split="h1"))
    assert torch.equal(yr.cpu(), torch.nn.functional.pad(torch.cat([x, x2], dim=3).cpu(), (0, 0, 1, 1, 1, 1)).half())


def test_group_norm_split_output_is_bordered_h2_of_fp32_output(dev):
    from diffpure_amd import ops
    x = (rnd(2, 8, 8, 256, seed=1) * 2 + 0.5).to(dev)
    x2 = rnd(2, 8, 8, 128, seed=2).to(dev)
    gamma, beta = (1 + 0.1 * rnd(384, seed=3)).to(dev), (0.1 * rnd(384, seed=4)).to(dev)
    for rs in (0, 1, 2):
        y32 = ops.group_norm(x, 32, 1e-5, gamma, beta, x2=x2, act=True, resample=rs)
        yh = ops.group_norm(x, 32, 1e-5, gamma, beta, x2=x2, act=True, resample=rs, split=True)
        assert yh.dtype == torch.float16 and yh.shape == (2, y32.shape[1] + 2, y32.shape[2] + 2, 768)
        assert torch.equal(yh.cpu(), refops.h2_encode(torch.nn.functional.pad(y32.cpu(), (0, 0, 1, 1, 1, 1))))


def test_group_norm_raw_second_output(dev):
    """GroupNorm-apply also emits the un-normalised cat(x, x2) in operand form (input of the 1x1 skip)."""
    from diffpure_amd import ops
    x = (rnd(2, 8, 8, 256, seed=1) * 2 + 0.5).to(dev)
    x2 = rnd(2, 8, 8, 128, seed=2).to(dev)
    gamma, beta = (1 + 0.1 * rnd(384, seed=3)).to(dev), (0.1 * rnd(384, seed=4)).to(dev)
    y, yr = ops.group_norm(x, 32, 1e-5, gamma, beta, x2=x2, act=True, split=True, raw=True)
    y0 = ops.group_norm(x, 32, 1e-5, gamma, beta, x2=x2, act=True, split=True)
    assert torch.equal(y, y0)
    assert torch.equal(yr.cpu(), refops.to_h2(torch.cat([x, x2], dim=3).cpu()))
    for mode in (1, 2):
        assert torch.equal(ops.to_h2(x, mode).cpu(), refops.to_h2(x.cpu(), mode))


@pytest.mark.parametrize("case", [(2, 16, 16, 128, 256, 3, "f32"), (2, 16, 16, 128, 256, 3, "h2"), (4, 8, 8, 256, 128, 1, "f32"),
                                  (8, 8, 8, 256, 256, 3, "h2"), (2, 32, 32, 64, 384, 3, "h2"), (3, 4, 4, 128, 128, 3, "f32")], ids=str)
def test_conv_epilogue_column_sums_give_groupnorm_stats(dev, case):
    """The statistics GroupNorm derives from the convolution epilogue's per-column partials equal the
    ones its own reduction pass computes from the tensor (also for a channel-split pair of tensors,
    and with the fallback when a 64/128-row tile would straddle two samples)."""
    from diffpure_amd import ops
    B, H, W, C, N, k, kind = case
    x = rnd(B, H, W, C, seed=1)
    w = rnd(N, C, k, k, seed=3, scale=1.0 / math.sqrt(C * k * k))
    bias, res = rnd(N, seed=4).to(dev), rnd(B, H, W, N, seed=6).to(dev)
    if kind == "h2":
        y = ops.conv2d_h2(_h2_bordered(x, dev), ops.pack_conv_weight_h2(w, dev), N, k, bias=bias, res=res, scale=0.7, colstats=True)
    else:
        y = ops.conv2d(x.to(dev), ops.pack_conv_weight(w).to(dev), N, k, bias=bias, res=res, scale=0.7, colstats=True)
    assert isinstance(y, ops.Act) and y.cols is not None
    G = 32
    fused = ops.group_norm_stats(y, G, 1e-5)
    plain = ops.group_norm_stats(y.t, G, 1e-5)               # the bare tensor -> reduction kernel
    close(fused, plain.cpu(), rtol=2e-5, atol=2e-6)
    # channel-split pair: both sources carry partials
    y2 = ops.conv2d(x.to(dev), ops.pack_conv_weight(rnd(128, C, 1, 1, seed=8)).to(dev), 128, 1, colstats=True)
    fused2 = ops.group_norm_stats(y, G, 1e-5, y2)
    plain2 = ops.group_norm_stats(y.t, G, 1e-5, y2.t)
    close(fused2, plain2.cpu(), rtol=2e-5, atol=2e-6)


def test_fast_silu_accuracy_and_extremes(dev):
    """dp_silu_f (hardware exp2 + reciprocal with first-order corrections, csrc/dp_common.h) against fp64 SiLU over
    the whole range, including the arguments where exp overflows / underflows."""
    from diffpure_amd import ops
    x = torch.cat([torch.linspace(-120, 120, 200001), torch.tensor([0.0, -0.0, 88.7, -88.7, 89.0, -89.0, 104.0, -104.0, 1e4, -1e4,
                                                                    1e-30, -1e-30, 3.0e38, -3.0e38])])
    pad = (-x.numel()) % 4
    x = torch.cat([x, torch.zeros(pad)])
    got = ops.silu(x.to(dev)).cpu()
    ref = (x.double() * torch.sigmoid(x.double())).float()
    assert torch.isfinite(got).all()
    err = (got - ref).abs()
    # |x| > 88.7: exp overflows in fp32 (as it does in torch's own fp32 SiLU) and results of size ~1e-37 flush to 0
    assert (err <= 4e-7 * ref.abs() + 2e-36).all(), (err / ref.abs().clamp_min(1e-30)).max()


@pytest.mark.batch_invariant
@pytest.mark.parametrize("case", [(6, 4, 4, 256, 256, 3), (6, 8, 8, 512, 256, 3), (4, 8, 8, 1024, 768, 1), (5, 2, 2, 128, 128, 3)], ids=str)
def test_conv2d_h2_split_k_levels_are_batch_shard_invariant(dev, case):
    """Low-resolution levels (H*W <= 64) are reduced with split-K; the split factor is a function of the layer shape
    only, so a batch and any slice of it give bit-identical rows (results AND the GroupNorm column sums when the
    64-row records line up with the samples), and they match the fp64 convolution."""
    from diffpure_amd import _lib, ops
    B, H, W, C, N, k = case
    assert _lib.load().dp_conv2d_nhwc_h2_workspace(B, H, W, k, C, N) > 0
    assert _lib.load().dp_conv2d_nhwc_h2_workspace(B, 16, 16, k, C, N) == 0
    x = rnd(B, H, W, C, seed=1)
    w = rnd(N, C, k, k, seed=3, scale=1.0 / math.sqrt(C * k * k))
    bias, res = rnd(N, seed=4).to(dev), rnd(B, H, W, N, seed=6).to(dev)
    wh = ops.pack_conv_weight_h2(w, dev)
    full = ops.conv2d_h2(_h2_bordered(x, dev), wh, N, k, bias=bias, res=res, scale=0.5, colstats=True)
    ref = ((torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w.double(), bias.cpu().double(), padding=k // 2)
            .permute(0, 2, 3, 1) + res.cpu().double()) * 0.5).float()
    close(full.t, ref, rtol=2e-5, atol=2e-5)
    for lo, hi in ((0, 1), (1, B), (2, 4)):
        part = ops.conv2d_h2(_h2_bordered(x[lo:hi], dev), wh, N, k, bias=bias, res=res[lo:hi].contiguous(), scale=0.5, colstats=True)
        assert torch.equal(part.t, full.t[lo:hi])
        if H * W == 64:     # one 64-row record per sample
            assert torch.equal(part.cols.buf[:hi - lo], full.cols.buf[lo:hi])      # (buffers are padded to whole 512-row tiles)


def test_torch_ops_namespace_runs_the_hip_kernels(dev):
    """torch.ops.diffpure_hip.* - registered from C++ (csrc/torch_binding.cpp, TORCH_LIBRARY) - run the same kernels as
    diffpure_amd.ops, bit for bit, with the column statistics as an explicit second return."""
    from diffpure_amd import ops, torch_ops  # noqa: F401
    T = torch.ops.diffpure_hip
    x = rnd(2, 16, 16, 128, seed=1).to(dev)
    w = rnd(96, 128, 3, 3, seed=2, scale=0.05)
    bias = rnd(96, seed=3).to(dev)
    wp = ops.pack_conv_weight(w).to(dev)
    assert torch.equal(T.conv2d_nhwc(x, wp, bias, 96, 3), ops.conv2d(x, wp, 96, 3, bias=bias))
    y, cols = T.conv2d_nhwc_stats(x, wp, bias, 96, 3)
    ref = ops.conv2d(x, wp, 96, 3, bias=bias, colstats=True)
    assert torch.equal(y, ref.t) and torch.equal(cols, ref.cols.buf)
    assert torch.equal(T.group_norm_stats_from_cols(cols, 2, 256, 32, 1e-5), ops.group_norm_stats(ref, 32, 1e-5))
    gamma, beta = (1 + 0.1 * rnd(128, seed=4)).to(dev), (0.1 * rnd(128, seed=5)).to(dev)
    for fmt, split in ((0, False), (1, "h2"), (2, "h1")):
        assert torch.equal(T.group_norm_silu(x, gamma, beta, 32, 1e-5, True, fmt), ops.group_norm(x, 32, 1e-5, gamma, beta, act=True, split=split))
    st = ops.group_norm_stats(x, 32, 1e-5)
    assert torch.equal(T.group_norm_silu(x, gamma, beta, 32, 1e-5, True, 2, st), ops.group_norm(x, 32, 1e-5, gamma, beta, act=True, split="h1", stats=st))
    wh = ops.pack_conv_weight_h2(w, dev)
    for split, passes in (("h2", 3), ("h1", 2)):
        xh = ops.group_norm(x, 32, 1e-5, gamma, beta, act=True, split=split)
        assert torch.equal(T.conv2d_h2(xh, wh, None, 96, 3, passes), ops.conv2d_h2(xh, wh, 96, 3, passes=passes))
        y2, c2 = T.conv2d_h2_stats(xh, wh, bias, 96, 3)          # passes = 0: the full arithmetic of the operand format
        r2 = ops.conv2d_h2(xh, wh, 96, 3, bias=bias, colstats=True)
        assert torch.equal(y2, r2.t) and torch.equal(c2, r2.cols.buf)
    for heads, c in ((2, 128), (3, 96)):                 # head dimension 64: flash kernel; 32: GEMM + softmax path
        qkv = rnd(2, 64, 3 * c, seed=6).to(dev)
        assert torch.equal(T.attention(qkv, heads, True), ops.attention(qkv, heads, "legacy"))
        assert torch.equal(T.attention(qkv, heads, False), ops.attention(qkv, heads, "split"))
    img = torch.rand(2, 3, 28, 28).to(dev)
    assert torch.equal(T.resize_affine(img, 32, 32, -0.5, 2.0, False, True), ops.resize_affine(img, (32, 32), -0.5, 2.0, False, True))
    xs, eps = rnd(2, 8, 8, 3, seed=7).to(dev), rnd(2, 8, 8, 6, seed=8).to(dev)
    assert torch.equal(T.em_step(xs, eps, -0.5, 1.1, -2.0, False, 1e-3, 1.05, 0.0316, 77, 5, 3),
                       ops.em_step(xs, eps, -0.5, 1.1, -2.0, False, 1e-3, 1.05, 0.0316, seed=77, sample0=5, step=3))
    with pytest.raises(RuntimeError):
        T.conv2d_nhwc(x, wp[:, :8].contiguous(), bias, 96, 3)     # TORCH_CHECK -> RuntimeError with the library's message


def test_torch_ops_abi6_operators_and_autograd_formulas(dev):
    """Round 5: torch.ops.diffpure_hip at ABI 6 - the fp16-stream operators run the same kernels as diffpure_amd.ops bit for bit, and
    torch.autograd works THROUGH conv2d_nhwc / group_norm_silu / attention / resize_affine (dL/dx against torch's own autograd of the
    equivalent torch expressions)."""
    import torch.nn.functional as F
    from diffpure_amd import ops, torch_ops  # noqa: F401
    T = torch.ops.diffpure_hip
    B, H, W, C, N = 2, 16, 16, 64, 64
    x = rnd(B, H, W, C, seed=1)
    xh = _h1_bordered(x, dev)
    w3 = rnd(N, C, 3, 3, seed=2, scale=1.0 / math.sqrt(9 * C))
    ws = rnd(N, 32, 1, 1, seed=3, scale=1.0 / math.sqrt(32))
    bias = rnd(N, seed=4).to(dev)
    table = rnd(B, N + 8, seed=5).to(dev)
    res16 = rnd(B, H, W, N, seed=6).half().to(dev)
    seg = rnd(B, H, W, 32, seed=7).half().to(dev)
    w16 = ops.order_conv_weight_w16(w3).half().to(dev)
    wf = ops.order_conv_weight_w16(ops.fuse_skip_weight(w3, ws)).half().to(dev)
    # conv2d_h2_ex: temb rows + fp16 residual + fp16 output + records; then with a 1x1 K-segment
    y, cols = T.conv2d_h2_ex(xh, w16, bias, table[:, 4:4 + N], res16, None, None, N, 3, 0, 1, 0.5, True, True)
    r = ops.conv2d_h2(xh, w16, N, 3, bias=bias, temb=table[:, 4:4 + N], res=res16, scale=0.5, colstats=True, w_fmt=1, out_f16=True)
    assert y.dtype == torch.float16 and torch.equal(y, r.t) and torch.equal(cols, r.cols.buf)
    y, cols = T.conv2d_h2_ex(xh, wf, bias, None, None, seg, None, N, 3, 0, 1, 1.0, True, True)
    r = ops.conv2d_h2(xh, wf, N, 3, bias=bias, colstats=True, w_fmt=1, out_f16=True, segs=(seg,))
    assert torch.equal(y, r.t) and torch.equal(cols, r.cols.buf)
    # gn_apply_h16 on that fp16 tensor with its records' statistics: operand form, FiLM + SiLU; the resampled plain tensor
    st = ops.group_norm_stats(r, 16, 1e-5)           # 64 channels: 16 groups of 4
    gamma, beta = (1 + 0.1 * rnd(N, seed=8)).to(dev), (0.1 * rnd(N, seed=9)).to(dev)
    film = rnd(B, 2 * N, seed=10, scale=0.3).to(dev)
    a, _ = T.gn_apply_h16(r.t, None, st, gamma, beta, film[:, :N], film[:, N:], 16, True, 0, 2, False)
    assert torch.equal(a, ops.group_norm(r.t, 16, 1e-5, gamma, beta, film=(film[:, :N], film[:, N:]), act=True, split="h1", stats=st))
    a, _ = T.gn_apply_h16(r.t, None, None, None, None, None, None, 1, False, 2, 3, False)
    assert torch.equal(a, ops.resample(r.t, ops.RESAMPLE_DOWN))
    # attention_fused: fp16 qkv, one pass, bordered operand out; fp32 qkv
    qkv = rnd(2, 256, 3 * 128, seed=11).to(dev)
    assert torch.equal(T.attention_fused(qkv, 2, True, 0), ops.attention_fused(qkv, 2, "legacy"))
    q16 = qkv.half()
    assert torch.equal(T.attention_fused(q16, 2, True, 16), ops.attention_fused(q16, 2, "legacy", operand_hw=(16, 16)))
    # round_weights == WeightPool.round (stochastic, keyed)
    pool = ops.WeightPool(torch.device(dev), stochastic=True)
    pool.add("w", w3)
    pool.finalize()
    pool.round(5)
    work = torch.empty_like(pool.work)
    T.round_weights(pool.master, work, True, pool.seed, 5)
    assert torch.equal(work, pool.work)
    # ---- autograd through the operators --------------------------------------------------------------------------------------
    wp = ops.pack_conv_weight(w3).to(dev)
    xr = x.to(dev).requires_grad_(True)
    cot = rnd(B, H, W, N, seed=12).to(dev)
    (g,) = torch.autograd.grad(T.conv2d_nhwc(xr, wp, bias, N, 3), xr, cot)
    xt = x.to(dev).requires_grad_(True)
    (gt,) = torch.autograd.grad(F.conv2d(xt.permute(0, 3, 1, 2), w3.to(dev), bias, padding=1).permute(0, 2, 3, 1), xt, cot)
    assert (g - gt).abs().max() < 1e-4 * gt.abs().max(), (g - gt).abs().max()
    gam, bet = (1 + 0.1 * rnd(C, seed=13)).to(dev), (0.1 * rnd(C, seed=14)).to(dev)
    xr = (x * 2 + 0.5).to(dev).requires_grad_(True)
    cot = rnd(B, H, W, C, seed=15).to(dev)
    (g,) = torch.autograd.grad(T.group_norm_silu(xr, gam, bet, 16, 1e-5, True, 0), xr, cot)
    xt = (x * 2 + 0.5).to(dev).requires_grad_(True)
    (gt,) = torch.autograd.grad(F.silu(F.group_norm(xt.permute(0, 3, 1, 2), 16, gam, bet, 1e-5)).permute(0, 2, 3, 1), xt, cot)
    assert (g - gt).abs().max() < 1e-4 * gt.abs().max(), (g - gt).abs().max()
    for heads, c, legacy in ((2, 128, True), (3, 96, False)):       # flash forward (d = 64) / GEMM + softmax forward (d = 32); both backward by GEMMs
        qr = rnd(2, 64, 3 * c, seed=16).to(dev).requires_grad_(True)
        cot = rnd(2, 64, c, seed=17).to(dev)
        (g,) = torch.autograd.grad(T.attention(qr, heads, legacy), qr, cot)
        qt = qr.detach().clone().requires_grad_(True)
        d = c // heads
        if legacy:
            q_, k_, v_ = qt.view(2, 64, heads, 3, d).unbind(3)
        else:
            q_, k_, v_ = qt.view(2, 64, 3, heads, d).unbind(2)
        att = torch.softmax(torch.einsum("bthd,bshd->bhts", q_, k_) / math.sqrt(d), dim=-1)
        (gt,) = torch.autograd.grad(torch.einsum("bhts,bshd->bthd", att, v_).reshape(2, 64, c), qt, cot)
        assert (g - gt).abs().max() < 2e-4 * gt.abs().max(), (heads, (g - gt).abs().max())
    img = torch.rand(2, 3, 28, 28).to(dev).requires_grad_(True)
    cot = rnd(2, 32, 32, 3, seed=18).to(dev)
    (g,) = torch.autograd.grad(T.resize_affine(img, 32, 32, -0.5, 2.0, False, True), img, cot)
    it = img.detach().clone().requires_grad_(True)
    (gt,) = torch.autograd.grad(((F.interpolate(it, size=(32, 32), mode="bilinear", align_corners=False) - 0.5) * 2.0).permute(0, 2, 3, 1), it, cot)
    assert (g - gt).abs().max() < 1e-5 * max(1.0, gt.abs().max().item())


FP16_OUT_CASES = [(2, 32, 32, 256, 256, 3, 2, True, 1.0), (1, 128, 128, 64, 256, 3, 1, False, 0.5), (4, 16, 16, 128, 128, 3, 0, False, 1.0),
                  (8, 8, 8, 256, 256, 3, 1, True, 1.0), (3, 16, 16, 96, 72, 3, 0, False, 1.0), (2, 64, 64, 64, 512, 1, 0, True, 1.0)]


@pytest.mark.parametrize("case", FP16_OUT_CASES, ids=[str(c) for c in FP16_OUT_CASES])
def test_conv2d_fp16_output_is_the_rounded_fp32_output(dev, case, tune):
    """dp_conv2d_nhwc_h2 out_fmt 1: the tensor is stored as plain fp16 = the fp32 result rounded to nearest, bit for bit, in
    every tile variant (generic 128 / 64 tiles, split-K level, ping-pong, halo, one-wave-per-SIMD with the paired-lane
    packed stores); the column records stay those of the unrounded values."""
    from diffpure_amd import ops
    B, H, W, C, N, k, temb_rows, has_res, scale = case
    x = rnd(B, H, W, C, seed=1)
    w = rnd(N, C, k, k, seed=3, scale=1.0 / math.sqrt(C * k * k))
    bias = rnd(N, seed=4).to(dev)
    table = rnd(B if temb_rows == 2 else 1, N + 8, seed=5).to(dev) if temb_rows else None
    res = rnd(B, H, W, N, seed=6).to(dev) if has_res else None
    xh = _h1_bordered(x, dev)
    w16 = ops.order_conv_weight_w16(w).half().to(dev)

    def run(f16):
        y = ops.conv2d_h2(xh, w16, N, k, bias=bias, temb=None if table is None else table[:, 4:4 + N], res=res, scale=scale,
                          colstats=True, w_fmt=1, out_f16=f16)
        return y.t, y.cols.buf.clone()

    res16 = None if res is None else res.half()
    res16f = None if res is None else res16.float()

    def run_res16(fp16_res):
        # the same residual VALUES (fp16-representable) as an fp16 tensor (res_fmt 1) and as an fp32 tensor: identical bits
        y = ops.conv2d_h2(xh, w16, N, k, bias=bias, temb=None if table is None else table[:, 4:4 + N], res=res16 if fp16_res else res16f,
                          scale=scale, colstats=True, w_fmt=1, out_f16=True)
        return y.t, y.cols.buf.clone()

    base32 = basecs = base_r = None
    combos = [("0", None, "0")]                                     # generic tiles only
    if B * H * W % 256 == 0 and N % 128 == 0 and not (H * W <= 64):
        combos += [("1", "0", "0"), ("1", "1", "0"), ("1", "2", "0")]      # ping-pong / one-wave-per-SIMD 256x256 / + 512x128
    if B * H * W % 256 == 0 and N % 256 == 0 and not (H * W <= 64) and C * k * k >= 128:
        combos += [("1", "2", "1")]                                 # the 8-wave kernel (where the launch has >= 256 tiles)
    for pp, sw, dw in combos:
        tune.setenv("DP_H2_PP", pp)
        if sw is None:
            tune.delenv("DP_H2_SW", raising=False)
        else:
            tune.setenv("DP_H2_SW", sw)
        tune.setenv("DP_H2_DW", dw)
        y32, cs32 = run(False)
        y16, cs16 = run(True)
        assert y16.dtype == torch.float16 and y16.shape == y32.shape
        assert torch.equal(y16, y32.half()), (pp, sw, dw)
        assert torch.equal(cs16, cs32), (pp, sw, dw)
        if base32 is None:
            base32, basecs = y32, cs32
        assert torch.equal(y32, base32) and torch.equal(cs32, basecs), (pp, sw, dw)
        if res is not None:             # fp16 residual stream: res_fmt 1 == the same values as fp32, in every variant
            ya, csa = run_res16(True)
            yb, csb = run_res16(False)
            assert torch.equal(ya, yb) and torch.equal(csa, csb), (pp, sw, dw)
            if base_r is None:
                base_r = (ya, csa)
            assert torch.equal(ya, base_r[0]) and torch.equal(csa, base_r[1]), (pp, sw, dw)


SEG_CASES = [
    # B, H, W, C (3x3 input = output channels of the block), N, C1, C2, scale
    (64, 32, 32, 256, 256, 256, 256, 1.0),          # decoder ResBlock (512 -> 256): two skip sources; 256 tiles -> the 8-wave kernel
    (32, 32, 32, 512, 512, 512, 256, 1.0),          # 768 -> 512
    (64, 32, 32, 256, 256, 128, 0, 0.70710678),     # encoder channel change (128 -> 256), NCSN++ skip_rescale
    (256, 16, 16, 256, 256, 256, 128, 0.70710678),  # NCSN++ 16x16 up path (384 -> 256) at the benchmarked batch
    (8, 32, 32, 128, 128, 128, 128, 0.70710678),    # 128 output channels: the 512x128 one-wave-per-SIMD tiles (NCSN++ 32x32 level)
    (2, 16, 16, 256, 256, 256, 128, 1.0),           # two tiles: one-wave-per-SIMD 256x256 / generic tiles
    (3, 8, 8, 512, 256, 512, 256, 1.0),             # 64 pixels per sample: split-K (2 parts of 84 k-tiles; 144 + 24 in all)
    (4, 4, 4, 64, 128, 512, 512, 1.0),              # split-K (2 parts of 25): the second part STARTS inside the first segment
    (5, 4, 4, 256, 256, 256, 256, 0.70710678),      # 16 pixels per sample: split-K (4 parts), ragged M (80 rows)
    (1, 5, 7, 64, 72, 32, 32, 1.0),                 # ragged everything: 64x64 generic tiles, N not a multiple of 32
]


@pytest.mark.parametrize("case", SEG_CASES, ids=[str(c) for c in SEG_CASES])
def test_conv2d_with_1x1_skip_k_segments(dev, case, tune):
    """A ResBlock's 1x1 skip_connection over its raw input folded into its second 3x3 convolution (ABI 6, K-segments): against the
    exact fp64 sum of the two convolutions of the fp16-rounded operands, against the un-fused pair of launches (3x3 with the 1x1's
    fp16 output as residual: they differ only by the rounding of that intermediate tensor), and IDENTICAL BITS from every tile variant
    that has a segment loader - the 8-wave kernel, both forms of the one-wave-per-SIMD kernel, the generic tiles incl. split-K -
    which is what makes "fused or not" a property of the layer and keeps results independent of the batch sharding."""
    from diffpure_amd import ops
    B, H, W, C, N, C1, C2, scale = case
    h = rnd(B, H, W, C, seed=1)
    s1 = rnd(B, H, W, C1, seed=2).half()
    s2 = rnd(B, H, W, C2, seed=3).half() if C2 else None
    w3 = rnd(N, C, 3, 3, seed=4, scale=1.0 / math.sqrt(9 * C))
    ws = rnd(N, C1 + C2, 1, 1, seed=5, scale=1.0 / math.sqrt(C1 + C2))
    b3, bs = rnd(N, seed=6), rnd(N, seed=7)
    assert ops.takes_segments(H, W, 3, C, N, C1, C2)
    hh = _h1_bordered(h, dev)
    wf = ops.order_conv_weight_w16(ops.fuse_skip_weight(w3, ws)).half().to(dev)
    segs = (s1.to(dev),) if s2 is None else (s1.to(dev), s2.to(dev))
    bias = (b3 + bs).to(dev)
    out16 = (H * W) % 64 == 0

    def run():
        y = ops.conv2d_h2(hh, wf, N, 3, bias=bias, scale=scale, colstats=True, w_fmt=1, out_f16=out16, segs=segs)
        return y.t, y.cols.buf.clone()

    tune.setenv("DP_H2_PP", "0")                    # generic tiles (and split-K) only
    base, base_cs = run()
    M = B * H * W
    combos = []
    if M % 256 == 0 and N % 256 == 0 and H * W > 64:
        combos += [("1", "1", "0"), ("1", "0", "0")]                    # one-wave-per-SIMD 256x256; ping-pong off-limits -> generic
        if (M // 256) * (N // 256) >= 256:
            combos += [("1", "2", "1")]                                   # the 8-wave kernel
    if M % 512 == 0 and N % 128 == 0 and H * W > 64:
        combos += [("1", "2", "0")]                                       # 512x128 one-wave-per-SIMD tiles
    for pp, sw, dw in combos:
        tune.setenv("DP_H2_PP", pp)
        tune.setenv("DP_H2_SW", sw)
        tune.setenv("DP_H2_DW", dw)
        for _ in range(2):
            got, got_cs = run()
            assert torch.equal(got, base) and torch.equal(got_cs, base_cs), (pp, sw, dw)
    for name in ("DP_H2_PP", "DP_H2_SW", "DP_H2_DW"):
        tune.delenv(name, raising=False)
    got, got_cs = run()                             # the dispatcher's own choice
    assert torch.equal(got, base) and torch.equal(got_cs, base_cs)
    raw = torch.cat([s1] + ([] if s2 is None else [s2]), dim=3)
    sub = slice(0, 2)                   # fp64 reference on two samples
    ref = torch.nn.functional.conv2d(h[sub].half().double().permute(0, 3, 1, 2), w3.half().double(), None, padding=1).permute(0, 2, 3, 1)
    ref = ref + raw[sub].double() @ ws[:, :, 0, 0].half().double().t() + (b3 + bs).double()
    ref = (ref * scale).float()
    err = (got[sub].float().cpu() - ref).abs().max().item()
    assert err < (2e-3 if out16 else 2e-5) * max(1.0, ref.abs().max().item()), err          # one fp16 rounding of the output
    # the un-fused pair: skip = 1x1(raw) stored in the stream's format, then the 3x3 with it as residual
    sraw = torch.nn.functional.pad(raw, (0, 0, 1, 1, 1, 1)).to(dev).contiguous()
    skip = ops.conv2d_h2(sraw, ops.order_conv_weight_w16(ws).half().to(dev), N, 1, bias=bs.to(dev), w_fmt=1, out_f16=out16)
    two = ops.conv2d_h2(hh, ops.order_conv_weight_w16(w3).half().to(dev), N, 3, bias=b3.to(dev), res=skip, scale=scale, colstats=True,
                        w_fmt=1, out_f16=out16)
    d = (got.float() - two.t.float()).abs().max().item()
    print(f"K-segment fusion {case}: max-abs vs fp64 {err:.2e}, vs the un-fused pair {d:.2e}")
    assert d < (4e-3 if out16 else 4e-5) * max(1.0, ref.abs().max().item()), d
    if out16:       # the column records describe the stored tensor's unrounded values: GroupNorm statistics of the two agree
        close(ops.group_norm_stats(ops.Act(got, ops.ColStats(got_cs, 64, N)), 4, 1e-5),
              ops.group_norm_stats(two, 4, 1e-5).cpu(), rtol=2e-3, atol=2e-3)


def test_stem_convolution_fp16_output(dev):
    """dp_conv2d_nhwc out_fmt 1: the fp32-MFMA stem writes the first tensor of the fp16 residual stream; records as the fp32 run's."""
    from diffpure_amd import ops
    x = rnd(3, 32, 32, 3, seed=1).to(dev)
    wp = ops.pack_conv_weight(rnd(128, 3, 3, 3, seed=2, scale=0.2)).to(dev)
    b = rnd(128, seed=3).to(dev)
    y32 = ops.conv2d(x, wp, 128, 3, bias=b, colstats=True)
    y16 = ops.conv2d(x, wp, 128, 3, bias=b, colstats=True, out_f16=True)
    assert y16.t.dtype == torch.float16 and torch.equal(y16.t, y32.t.half()) and torch.equal(y16.cols.buf, y32.cols.buf)


def ulp_close(got, ref, what, rate=0.05):
    """fp16 tensors equal up to one fp16 ulp (2^-10 relative, one subnormal step absolute) in at most `rate` of the elements"""
    g, r = got.float(), ref.float()
    diff = (g - r).abs()
    assert (diff <= r.abs() * 2.0 ** -10 + 6.0e-8).all(), (what, diff.max().item())
    assert (got != ref).float().mean().item() < rate, (what, (got != ref).float().mean().item())


H16_CASES = [(2, 8, 8, 256, 0, 32), (3, 6, 10, 64, 0, 16), (1, 16, 16, 1024, 0, 32), (2, 8, 8, 256, 128, 32), (2, 16, 16, 128, 128, 32),
             (1, 4, 260, 128, 0, 32), (2, 8, 8, 512, 1024, 32)]


@pytest.mark.parametrize("case", H16_CASES, ids=[str(c) for c in H16_CASES])
def test_group_norm_fp16_input_equals_group_norm_of_the_upconverted_tensors(dev, case):
    """dp_gn_apply_h16 (plain fp16 NHWC in - a first convolution's fp16 output, the fp16 residual stream, the two sources of a skip
    concatenation - -> bordered fp16 operand, or a plain fp16 tensor) gives the bytes of dp_gn_apply(out_fmt 2) on the same values held
    as fp32: FiLM rows per sample / broadcast / absent, with and without SiLU, 2x nearest-up / 2x2 mean-down, the raw second output, no
    normalisation at all (resampled identity skip, raw operand of a 1x1 convolution).  With SiLU the fp16 kernel evaluates the sigmoid on the
    hardware exp2 / reciprocal units without the fp32-accuracy corrections of the fp32 kernels (dp_silu_fast_f): the stored fp16 values
    differ from the fp32 kernel's by at most one fp16 ulp, in a small fraction of the elements."""
    from diffpure_amd import ops
    B, H, W, C1, C2, G = case
    C = C1 + C2
    x16 = (rnd(B, H, W, C1, seed=1) * 2 + 0.5).half().to(dev)
    x2_16 = (rnd(B, H, W, C2, seed=2) - 0.3).half().to(dev) if C2 else None
    xf, x2f = x16.float(), None if x2_16 is None else x2_16.float()
    gamma, beta = (1 + 0.1 * rnd(C, seed=3)).to(dev), (0.1 * rnd(C, seed=4)).to(dev)
    stats = ops.group_norm_stats(xf, G, 1e-5, x2f)
    for film_rows in (0, 1, B):
        table = rnd(film_rows, 2 * C + 8, seed=7).to(dev) if film_rows else None
        film = None if table is None else (table[:, 4:4 + C], table[:, 4 + C:4 + 2 * C])
        for act in (True, False):
            for rs in (0, 1, 2) if H % 2 == 0 and W % 2 == 0 else (0, 1):
                ref = ops.group_norm(xf, G, 1e-5, gamma, beta, x2=x2f, film=film, act=act, resample=rs, split="h1", stats=stats)
                got = ops.group_norm(x16, G, 1e-5, gamma, beta, x2=x2_16, film=film, act=act, resample=rs, split="h1", stats=stats)
                assert got.dtype == torch.float16 and got.shape == ref.shape
                if act:
                    ulp_close(got, ref, (case, film_rows, act, rs))
                else:
                    assert torch.equal(got, ref), (case, film_rows, act, rs)
    y, yr = ops.group_norm(x16, G, 1e-5, gamma, beta, x2=x2_16, act=True, split="h1", stats=stats, raw=True)
    y0, yr0 = ops.group_norm(xf, G, 1e-5, gamma, beta, x2=x2f, act=True, split="h1", stats=stats, raw=True)
    ulp_close(y, y0, "raw")
    assert torch.equal(yr, yr0)
    # round 6: with 2x resampling the raw second output is the RESAMPLED input as a plain fp16 tensor (the identity skip of an up / down
    # ResBlock) - the bytes of the stand-alone resampler - and the operand is the one of the call without it
    for rs in (1, 2) if H % 2 == 0 and W % 2 == 0 else (1,):
        for act in (True, False):
            y1, s1 = ops.group_norm(x16, G, 1e-5, gamma, beta, x2=x2_16, act=act, resample=rs, split="h1", stats=stats, raw=True)
            y2 = ops.group_norm(x16, G, 1e-5, gamma, beta, x2=x2_16, act=act, resample=rs, split="h1", stats=stats)
            cat = x16 if x2_16 is None else torch.cat([x16, x2_16], dim=3)
            assert torch.equal(y1, y2) and s1.dtype == torch.float16 and torch.equal(s1, ops.resample(cat, rs)), (case, rs, act)
    if C2 == 0:
        assert torch.equal(ops.group_norm_f16in(x16, G, gamma, beta, stats, act=True), y)
        for rs in (0, 1, 2) if H % 2 == 0 and W % 2 == 0 else (0, 1):
            assert torch.equal(ops.to_h2(x16, rs, fmt="h1"), ops.to_h2(xf, rs, fmt="h1")), rs
            if rs:
                r16 = ops.resample(x16, rs)
                assert r16.dtype == torch.float16 and torch.equal(r16, ops.resample(xf, rs).half()), rs
    with pytest.raises(Exception):
        ops.group_norm(x16, G, 1e-5, gamma, beta, x2=x2_16, act=True, split="h2", stats=stats)      # fp16 in -> "h1" only


@pytest.mark.parametrize("case", H16_CASES + [(4, 64, 64, 256, 0, 32), (2, 32, 32, 512, 512, 32)], ids=str)
def test_group_norm_fp16_rows_cut_across_workgroups_same_bits(dev, case, tune):
    """Round 6: dp_gn_apply_h16 cuts an output row across several workgroups (grid.y) while the launch would have fewer than DP_GN_WG
    (default 2048) of them - small batches, low levels.  Elementwise work: the same bytes as one workgroup per row (DP_GN_WG=0) and as
    the most cuts a row admits (DP_GN_WG=10^6), for every resampling mode, with and without the raw second output, borders included."""
    from diffpure_amd import ops
    B, H, W, C1, C2, G = case
    C = C1 + C2
    x16 = (rnd(B, H, W, C1, seed=1) * 2 + 0.5).half().to(dev)
    x2_16 = (rnd(B, H, W, C2, seed=2) - 0.3).half().to(dev) if C2 else None
    gamma, beta = (1 + 0.1 * rnd(C, seed=3)).to(dev), (0.1 * rnd(C, seed=4)).to(dev)
    stats = ops.group_norm_stats(x16.float(), G, 1e-5, None if x2_16 is None else x2_16.float())
    table = rnd(B, 2 * C + 8, seed=7).to(dev)
    film = (table[:, 4:4 + C], table[:, 4 + C:4 + 2 * C])

    def run():
        outs = []
        for rs in (0, 1, 2) if H % 2 == 0 and W % 2 == 0 else (0, 1):
            for raw in (False, True):
                r = ops.group_norm(x16, G, 1e-5, gamma, beta, x2=x2_16, film=film, act=True, resample=rs, split="h1", stats=stats, raw=raw)
                outs += list(r) if raw else [r]
            outs.append(ops.to_h2(x16, rs, fmt="h1"))
            if rs and C2 == 0:
                outs.append(ops.resample(x16, rs))           # plain fp16 output (out_fmt 3)
        return outs

    tune.setenv("DP_GN_WG", "0")
    base = run()
    for wg in ("2048", "1000000"):
        tune.setenv("DP_GN_WG", wg)
        got = run()
        assert len(got) == len(base) and all(torch.equal(a, b) for a, b in zip(got, base)), (case, wg)
    tune.delenv("DP_GN_WG")


def test_attention_fused_operand_output(dev):
    """dp_attention_fused out_fmt 1: the attention output lands as the zero-bordered fp16 operand of proj_out - the fp32
    result rounded to nearest in the interior pixels, zeros on the border."""
    from diffpure_amd import ops
    for (B, hh, ww, heads, layout, d) in ((2, 8, 8, 4, "legacy", 64), (1, 16, 16, 8, "legacy", 64), (2, 8, 16, 2, "split", 64),
                                          (3, 16, 16, 1, "split", 256)):
        c = heads * d
        qkv = rnd(B, hh * ww, 3 * c, seed=11).to(dev)
        ref = ops.attention_fused(qkv, heads, layout)
        got = ops.attention_fused(qkv, heads, layout, operand_hw=(hh, ww))
        want = torch.nn.functional.pad(ref.view(B, hh, ww, c), (0, 0, 1, 1, 1, 1)).half()
        assert got.dtype == torch.float16 and got.shape == want.shape
        # the kernel converts o * (1 / l) to fp16 in ONE rounding (v_fma_mixlo_f16); rounding the fp32 product again may differ by
        # one fp16 ulp in a few elements per 10^5 (measured: 6 of 165 888): equal up to that, exact zeros on the border
        diff = (got.float() - want.float()).abs()
        assert (diff <= want.float().abs() * 2.0 ** -10 + 1e-7).all(), ((B, hh, ww, heads, layout), diff.max().item())
        assert (got != want).float().mean().item() < 1e-3
        border = torch.ones_like(got, dtype=torch.bool)
        border[:, 1:-1, 1:-1, :] = False
        assert (got[border] == 0).all()


NN_CASES = [(2, 32, 32, 64, 6), (1, 256, 256, 32, 6), (3, 64, 64, 96, 3), (1, 128, 128, 64, 32), (1, 8, 512, 32, 16), (4, 16, 64, 64, 6)]


@pytest.mark.parametrize("case", NN_CASES, ids=[str(c) for c in NN_CASES])
def test_conv2d_few_output_channels_kernel(dev, case, tune):
    """igemm_h2_nn.hip (3x3, fp16 x fp16, N <= 32: the 6-channel head): identical bits to the generic tile kernel it replaces, and the
    exact convolution of the fp16-rounded operands to fp32-class accuracy; one to three channel slices, one / several image rows
    per 256-pixel tile, a wide image, the last tile of the tensor."""
    from diffpure_amd import ops
    B, H, W, C, N = case
    x = rnd(B, H, W, C, seed=1)
    w = rnd(N, C, 3, 3, seed=3, scale=1.0 / math.sqrt(C * 9))
    bias = rnd(N, seed=4).to(dev)
    ref = torch.nn.functional.conv2d(x.half().double().permute(0, 3, 1, 2), w.half().double(), bias.cpu().double(), padding=1).permute(0, 2, 3, 1)
    ref = (ref * 0.5).float()
    xh = _h1_bordered(x, dev)
    w16 = ops.order_conv_weight_w16(w).half().to(dev)
    tune.setenv("DP_H2_NN", "0")
    base = ops.conv2d_h2(xh, w16, N, 3, bias=bias, scale=0.5, w_fmt=1)
    close(base, ref, rtol=2e-5, atol=2e-5)
    tune.setenv("DP_H2_NN", "1")
    for _ in range(3):
        got = ops.conv2d_h2(xh, w16, N, 3, bias=bias, scale=0.5, w_fmt=1)
        assert torch.equal(got, base)
    # the epilogues this kernel does not carry go to the generic tiles, whatever the switch says
    y = ops.conv2d_h2(xh, w16, N, 3, bias=bias, scale=0.5, w_fmt=1, colstats=True)
    assert torch.equal(y.t, base)



def test_conv2d_fp16_residual_through_the_c_abi_with_an_offset_strided_residual(dev):
    """Round 5 (advisor): the C ABI accepts any 4-byte-aligned fp16 residual with an even row stride; the 256-wide tile kernels fetch
    the residual in 16-byte LDS-DMA pieces and therefore take only 16-byte-aligned residuals whose row stride is a multiple of 8
    elements - any other goes to the generic tiles.  A launch that fills the chip (256 tiles: the 8-wave kernel's case) with the
    residual at a 4-byte offset and a row stride of N + 2 elements must give the bits of the same launch on a contiguous copy."""
    import ctypes
    from diffpure_amd import _lib, ops
    B, H, W, C, N = 64, 32, 32, 64, 256
    x = _h1_bordered(rnd(B, H, W, C, seed=1), dev)
    wf = ops.order_conv_weight_w16(rnd(N, C, 3, 3, seed=2, scale=1.0 / math.sqrt(9 * C))).half().to(dev)
    bias = rnd(N, seed=3).to(dev)
    M = B * H * W
    res = rnd(M, N, seed=4).half().to(dev)
    want = ops.conv2d_h2(x, wf, N, 3, bias=bias, res=res.view(B, H, W, N), scale=0.5, colstats=True, w_fmt=1, out_f16=True)
    ldr = N + 2
    store = torch.zeros(M * ldr + 2, dtype=torch.float16, device=dev)
    store[2:].view(M, ldr)[:, :N] = res                     # residual rows start 4 bytes into the allocation, stride N + 2
    out = torch.empty((B, H, W, N), device=dev, dtype=torch.float16)
    cs, tr = ops._colstats_alloc(M, N, x.device)
    _lib.call("dp_conv2d_nhwc_h2", x.data_ptr(), C, B, H, W, 3, wf.data_ptr(), N, bias.data_ptr(), None, 0, store.data_ptr() + 4, ldr, 0.5,
              out.data_ptr(), N, cs.data_ptr(), ctypes.addressof(tr), None, 0, 1, 1, 1, 1, 1, None, 0, None, 0, ops._stream())
    torch.cuda.synchronize()
    assert tr.value == want.cols.tile_rows
    assert torch.equal(out, want.t) and torch.equal(cs, want.cols.buf)


DH_CASES = [
    # B, H, W, C, N, temb rows (0 none / 1 broadcast / 2 per sample), residual (0 none / 16 / 32), fp16 output, K-segment channels (C1, C2), scale
    (128, 16, 16, 256, 256, 2, 16, True, (0, 0), 0.70710678),     # NCSN++ 16x16 level at the adjoint benchmark's batch: 128 tiles of 256x256
    (128, 16, 16, 256, 256, 0, 32, False, (0, 0), 0.70710678),    # the same launch in the taped forward (fp32 stream)
    (128, 16, 16, 256, 256, 0, 0, False, (0, 0), 1.0),            # ... and as an input-gradient convolution
    (128, 16, 16, 256, 256, 0, 0, True, (256, 128), 0.70710678),  # up path 384 -> 256: the 1x1 skip as two K-segments
    (4, 64, 64, 512, 512, 1, 16, True, (0, 0), 1.0),              # guided UNet 64x64 level at the reference's per-GPU batch of 4
    (3, 64, 64, 256, 512, 2, 0, True, (256, 0), 1.0),             # 96 tiles of 256x256 -> 192 half tiles (two per CU on some), one segment
    (65, 16, 16, 64, 256, 2, 16, True, (0, 0), 1.0),              # M = 16 640 = 130 x 128: an odd number of half tiles (M % 256 != 0)
    # split-K levels (<= 64 pixels per sample; the split factor is fixed by the layer shape): one part per grid.y, raw partial sums
    (256, 8, 8, 256, 256, 2, 16, True, (0, 0), 0.70710678),       # NCSN++ 8x8 level at B = 256: 128 half tiles x 2 parts of 36 k-tiles
    (256, 4, 4, 256, 256, 2, 32, False, (0, 0), 0.70710678),      # 4x4 level: 32 half tiles x 4 parts of 18 k-tiles, fp32 stream
    (128, 8, 8, 256, 256, 0, 0, True, (256, 128), 0.70710678),    # 8x8 up path with two K-segments: 2 parts of 42 k-tiles, the cut inside segment 1
    (64, 8, 8, 1024, 1024, 1, 16, True, (0, 0), 1.0),             # guided UNet 8x8 level at B = 64: 32 x 4 half tiles x 2 parts of 144
    # 128 output channels (NCSN++ 32x32 level): the 256x128 form of the kernel, taken with DP_H2_DH = 3
    (64, 32, 32, 128, 128, 2, 16, True, (0, 0), 0.70710678),
    (64, 32, 32, 128, 128, 0, 32, False, (128, 128), 0.70710678),
]


DW128_CASES = [
    # B, H, W, C, N, temb rows (0 none / 1 shared / 2 per sample), residual (0 / 16 / 32), fp16 output, (C1, C2) K-segments, scale
    (128, 32, 32, 128, 128, 2, 16, True, (0, 0), 0.70710678),     # NCSN++ 32x32 level, stream form: 256 tiles of 512x128 - one round
    (128, 32, 32, 128, 128, 0, 32, False, (128, 128), 0.70710678),  # fp32 stream + the 1x1 skip as two K-segments
    (128, 32, 32, 256, 128, 1, 16, True, (0, 0), 1.0),            # 256 -> 128 (up path), shared time-embedding row
    (160, 32, 32, 128, 128, 0, 0, True, (256, 0), 1.0),           # 320 tiles: a second, partial round; one segment
    (128, 32, 32, 128, 128, 0, 0, False, (0, 0), 1.0),            # as an input-gradient convolution (fp32 out, nothing else)
    (32, 64, 64, 384, 128, 2, 16, True, (0, 0), 1.0),             # 64x64 maps, 108 k-tiles
]


@pytest.mark.parametrize("case", DW128_CASES, ids=[str(c) for c in DW128_CASES])
def test_conv2d_eight_wave_kernel_on_512x128_tiles_is_bit_identical(dev, case, tune):
    """Round 6: the 8-wave kernel's 512x128 form (conv_igemm_dw<0, 1>: eight waves stacked along the pixels, the wave tile, fragment reads
    and epilogue of the square form) takes the 3x3 layers with 128 output channels on launches of >= 256 tiles.  Identical bits to the
    generic tiles and to the one-wave-per-SIMD kernel's 512x128 tiles it replaces (DP_H2_DW = 1) - output (fp16 or fp32) and column
    records - with time-embedding rows, fp16 / fp32 residual and 1x1 K-segments; fp64 reference on two samples."""
    from diffpure_amd import ops
    B, H, W, C, N, temb_rows, res_kind, out16, (C1, C2), scale = case
    assert (B * H * W) % 512 == 0 and (B * H * W // 512) * (N // 128) >= 256
    h = rnd(B, H, W, C, seed=1)
    w3 = rnd(N, C, 3, 3, seed=2, scale=1.0 / math.sqrt(9 * C))
    hh = _h1_bordered(h, dev)
    bias = rnd(N, seed=3).to(dev)
    table = rnd(B if temb_rows == 2 else 1, N + 8, seed=4).to(dev) if temb_rows else None
    res = None
    if res_kind:
        res = rnd(B, H, W, N, seed=5).to(dev)
        res = res.half() if res_kind == 16 else res
    segs = None
    if C1:
        ws = rnd(N, C1 + C2, 1, 1, seed=6, scale=1.0 / math.sqrt(C1 + C2))
        wf = ops.order_conv_weight_w16(ops.fuse_skip_weight(w3, ws)).half().to(dev)
        segs = (rnd(B, H, W, C1, seed=7).half().to(dev),) + ((rnd(B, H, W, C2, seed=8).half().to(dev),) if C2 else ())
    else:
        wf = ops.order_conv_weight_w16(w3).half().to(dev)

    def run():
        y = ops.conv2d_h2(hh, wf, N, 3, bias=bias, temb=None if table is None else table[:, 4:4 + N], res=res, scale=scale, colstats=True,
                          w_fmt=1, out_f16=out16, segs=segs)
        return y.t, y.cols.buf.clone()

    tune.setenv("DP_H2_PP", "0")                    # generic tiles
    base, base_cs = run()
    tune.delenv("DP_H2_PP")
    tune.setenv("DP_H2_DW", "1")                    # what took these launches in rounds 3-5: the one-wave-per-SIMD kernel's 512x128 tiles
    old, old_cs = run()
    assert torch.equal(old, base) and torch.equal(old_cs, base_cs)
    tune.setenv("DP_H2_DW", "2")
    for _ in range(3):
        got, got_cs = run()
        assert torch.equal(got, base) and torch.equal(got_cs, base_cs)
    tune.delenv("DP_H2_DW")
    got, got_cs = run()                             # the dispatcher's own choice (= 2)
    assert torch.equal(got, base) and torch.equal(got_cs, base_cs)
    sub = slice(0, 2)                               # fp64 reference on two samples
    ref = torch.nn.functional.conv2d(h[sub].half().double().permute(0, 3, 1, 2), w3.half().double(), None, padding=1).permute(0, 2, 3, 1) + bias.cpu().double()
    if segs:
        ref = ref + torch.cat([s_[sub].cpu() for s_ in segs], dim=3).double() @ ws[:, :, 0, 0].half().double().t()
    if table is not None:
        ref = ref + table[:, 4:4 + N].cpu().double()[:2 if temb_rows == 2 else 1].view(-1, 1, 1, N)
    if res is not None:
        ref = ref + res[sub].cpu().double()
    ref = (ref * scale).float()
    err = (got[sub].float().cpu() - ref).abs().max().item()
    assert err < (2e-3 if out16 else 2e-5) * max(1.0, ref.abs().max().item()), err


@pytest.mark.parametrize("case", DH_CASES, ids=[str(c) for c in DH_CASES])
def test_conv2d_half_height_tile_kernel_is_bit_identical(dev, case, tune):
    """Round 5: conv_igemm_dh (128x256 tiles, four waves, two workgroups per CU) takes the fp16 x fp16 launches that leave CUs idle on
    256x256 tiles.  Identical bits to the generic tiles and to the one-wave-per-SIMD kernel it replaces there - output (fp16 or fp32),
    column records - with time-embedding rows, fp16 / fp32 residual and 1x1 K-segments; and the dispatcher's own choice is this kernel."""
    from diffpure_amd import ops
    B, H, W, C, N, temb_rows, res_kind, out16, (C1, C2), scale = case
    h = rnd(B, H, W, C, seed=1)
    w3 = rnd(N, C, 3, 3, seed=2, scale=1.0 / math.sqrt(9 * C))
    hh = _h1_bordered(h, dev)
    bias = rnd(N, seed=3).to(dev)
    table = rnd(B if temb_rows == 2 else 1, N + 8, seed=4).to(dev) if temb_rows else None
    res = None
    if res_kind:
        res = rnd(B, H, W, N, seed=5).to(dev)
        res = res.half() if res_kind == 16 else res
    segs = None
    if C1:
        ws = rnd(N, C1 + C2, 1, 1, seed=6, scale=1.0 / math.sqrt(C1 + C2))
        wf = ops.order_conv_weight_w16(ops.fuse_skip_weight(w3, ws)).half().to(dev)
        segs = (rnd(B, H, W, C1, seed=7).half().to(dev),) + ((rnd(B, H, W, C2, seed=8).half().to(dev),) if C2 else ())
    else:
        wf = ops.order_conv_weight_w16(w3).half().to(dev)

    def run():
        y = ops.conv2d_h2(hh, wf, N, 3, bias=bias, temb=None if table is None else table[:, 4:4 + N], res=res, scale=scale, colstats=True,
                          w_fmt=1, out_f16=out16, segs=segs)
        return y.t, y.cols.buf.clone()

    tune.setenv("DP_H2_PP", "0")                    # generic tiles
    base, base_cs = run()
    tune.delenv("DP_H2_PP")
    tune.setenv("DP_H2_DH", "0")                    # what took these launches before: the one-wave-per-SIMD kernel / generic tiles
    old, old_cs = run()
    assert torch.equal(old, base) and torch.equal(old_cs, base_cs)
    tune.delenv("DP_H2_DH")
    ops.prof_enable(True)
    for _ in range(2):
        got, got_cs = run()                         # default dispatch: conv_igemm_dh
        assert torch.equal(got, base) and torch.equal(got_cs, base_cs)
    ops.prof_enable(False)
    prof = ops.prof_collect()
    if H * W > 64:      # (un-split launches are booked under the kernel's own kind since ABI 8, i.e. they did not run on the generic tiles)
        assert prof["dh3x3"]["n"] == 2 and prof["other3x3"]["n"] == 0 and prof["pp3x3"]["n"] == 0, prof
    if N % 256 != 0:    # the 256x128 form (DP_H2_DH = 3): same bits as the 512x128 one-wave-per-SIMD tiles the default takes
        tune.setenv("DP_H2_DH", "3")
        got3, got3_cs = run()
        assert torch.equal(got3, base) and torch.equal(got3_cs, base_cs)
        tune.delenv("DP_H2_DH")
    sub = slice(0, 2)                               # fp64 reference on two samples
    ref = torch.nn.functional.conv2d(h[sub].half().double().permute(0, 3, 1, 2), w3.half().double(), None, padding=1).permute(0, 2, 3, 1) + bias.cpu().double()
    if segs:
        ref = ref + torch.cat([s_[sub].cpu() for s_ in segs], dim=3).double() @ ws[:, :, 0, 0].half().double().t()
    if table is not None:
        ref = ref + table[:, 4:4 + N].cpu().double()[:2 if temb_rows == 2 else 1].view(-1, 1, 1, N)
    if res is not None:
        ref = ref + res[sub].cpu().double()
    ref = (ref * scale).float()
    err = (got[sub].float().cpu() - ref).abs().max().item()
    assert err < (2e-3 if out16 else 3e-5) * max(1.0, ref.abs().max().item()), err


@pytest.mark.parametrize("N", [128, 64, 192])
@pytest.mark.parametrize("trans", [(0, 0), (0, 1), (1, 0), (1, 1)], ids=str)
def test_gemm_strided_on_the_fp16_matrix_cores(dev, trans, N):
    """Round 5: dp_gemm_strided_h16 - the strided batched GEMM with operands rounded to fp16 on their way into LDS, one fp16 MFMA pass,
    fp32 accumulation - against the exact fp64 product of the fp16-rounded operands, in all four storage forms, with batch and head
    strides that differ from the dense ones (the attention backward reads q, k, v in place inside qkv).  Round 6: N = 64 / 192 take the
    128 x 64 tile (dV / dQ / dK of the guided UNet's 64-wide heads)."""
    from diffpure_amd import _lib
    ta, tb = trans
    ZB, ZH, M, K = 2, 3, 256, 96
    lda, ldb, ldc = (M if ta else K) + 8, (K if tb else N) + 16, N + 12
    ra, rb = (K if ta else M), (N if tb else K)            # stored rows
    A = rnd(ZB, ZH, ra, lda, seed=1).to(dev)
    B = rnd(ZB, ZH, rb, ldb, seed=2).to(dev)
    Cm = torch.zeros(ZB, ZH, M, ldc, device=dev)
    assert _lib.load().dp_gemm_strided_h16_ok(M, N, K) == 1 and _lib.load().dp_gemm_strided_h16_ok(M, 32, K) == 0 and _lib.load().dp_gemm_strided_h16_ok(64, N, K) == 0
    s_ = torch.cuda.current_stream().cuda_stream
    _lib.call("dp_gemm_strided_h16", A.data_ptr(), 0, lda, ZH * ra * lda, ra * lda, ta, B.data_ptr(), 0, ldb, ZH * rb * ldb, rb * ldb, tb,
              Cm.data_ptr(), ldc, ZH * M * ldc, M * ldc, M, N, K, ZB, ZH, 0.25, s_)
    a = A.cpu().half().double()
    b = B.cpu().half().double()
    a = (a[..., :M].transpose(-1, -2) if ta else a[..., :K])          # [.., M, K]
    b = (b[..., :K].transpose(-1, -2) if tb else b[..., :N])          # [.., K, N]
    ref = (0.25 * (a @ b)).float()
    got = Cm.cpu()[..., :N]
    assert (got - ref).abs().max() < 2e-5 * max(1.0, ref.abs().max().item()), (got - ref).abs().max()
    assert (Cm.cpu()[..., N:] == 0).all()                             # nothing written beyond the N columns
    # operands that ARE fp16 in memory (a_fmt / b_fmt 1: q, k, v of the taped fp16 qkv, read in place): the same bits, in every mix
    A16, B16 = A.half(), B.half()
    for af, bf in ((1, 0), (0, 1), (1, 1)):
        C2 = torch.zeros_like(Cm)
        _lib.call("dp_gemm_strided_h16", (A16 if af else A).data_ptr(), af, lda, ZH * ra * lda, ra * lda, ta, (B16 if bf else B).data_ptr(), bf, ldb,
                  ZH * rb * ldb, rb * ldb, tb, C2.data_ptr(), ldc, ZH * M * ldc, M * ldc, M, N, K, ZB, ZH, 0.25, s_)
        assert torch.equal(C2, Cm), (af, bf)


@pytest.mark.parametrize("case", [(2, 256, 256, 1, "split"), (2, 256, 128, 2, "legacy")], ids=str)
def test_attention_backward_on_the_fp16_matrix_cores(dev, case):
    """The attention backward of the fp16 x fp16 precision modes (h16=True): recomputed probabilities (no P V product) and the four
    gradient products on dp_gemm_strided_h16 where the shape allows (NCSN++: T = 256, d = 256 - all of them; guided heads of d = 64:
    q k^T and dP), against fp64 autograd: within 3e-3 of the largest entry (the operands carry 11 bits)."""
    from diffpure_amd import ops
    B, T, C, heads, layout = case
    qkv, dout = rnd(B, T, 3 * C, seed=11), rnd(B, T, C, seed=12)
    ref = refops.attention_bwd(qkv.double(), None, dout.double(), heads, layout).float()
    none, probs = ops.attention(qkv.to(dev), heads, layout, probs_only=True, h16=True)
    assert none is None
    got = ops.attention_bwd(qkv.to(dev), probs, dout.to(dev), heads, layout, h16=True)
    err = ((got.cpu() - ref).abs().max() / ref.abs().max()).item()
    _, p32 = ops.attention(qkv.to(dev), heads, layout, return_probs=True)
    g32 = ops.attention_bwd(qkv.to(dev), p32, dout.to(dev), heads, layout)
    err32 = ((g32.cpu() - ref).abs().max() / ref.abs().max()).item()
    print(f"attention backward {case}: fp16 matrix cores {err:.2e} of the largest entry, fp32-input MFMA {err32:.2e}")
    assert err < 3e-3 and err32 < 1e-4, (err, err32)
    if ops.attention_h16_serves(T, C // heads):     # the taped fp16 qkv read in place: the bits of the same values handed over as fp32
        q16 = qkv.half().to(dev)
        _, p_a = ops.attention(q16, heads, layout, probs_only=True, h16=True)
        g_a = ops.attention_bwd(q16, p_a, dout.to(dev), heads, layout, h16=True)
        _, p_b = ops.attention(q16.float(), heads, layout, probs_only=True, h16=True)
        g_b = ops.attention_bwd(q16.float(), p_b, dout.to(dev), heads, layout, h16=True)
        assert g_a.dtype == torch.float32 and torch.equal(p_a, p_b) and torch.equal(g_a, g_b)
    else:
        assert heads == 2


STEM_CASES = [(2, 32, 32, 128), (1, 256, 256, 256), (3, 16, 24, 256), (5, 8, 8, 128)]


@pytest.mark.parametrize("case", STEM_CASES, ids=str)
def test_stem_kernel_against_fp64_and_its_column_records(dev, case):
    """Round 6: dp_conv2d_stem (csrc/stem.hip) - the 3 -> N stem as a write-bound kernel on 22-bit (hi, lo) operands - against the fp64
    convolution (fp32-class: 1e-5), zero padding at all four image edges included; fp16 output = the fp32 output rounded, same records;
    the records give the GroupNorm statistics of the unrounded tensor; a batch's leading samples are bit-identical to the small batch."""
    from diffpure_amd import ops
    B, H, W, N = case
    assert ops.conv2d_stem_ok(3, B, H, W, N)
    x = rnd(B, H, W, 3, seed=1) * 1.5
    w = rnd(N, 3, 3, 3, seed=2, scale=0.2)
    bias = rnd(N, seed=3)
    ref = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w.double(), bias.double(), padding=1).permute(0, 2, 3, 1).float()
    w16 = ops.pack_stem_weight(w).to(dev)
    y32 = ops.conv2d_stem(x.to(dev), w16, N, bias=bias.to(dev), colstats=True)
    y16 = ops.conv2d_stem(x.to(dev), w16, N, bias=bias.to(dev), colstats=True, out_f16=True)
    close(y32.t, ref, rtol=1e-5, atol=1e-5)
    assert y16.t.dtype == torch.float16 and torch.equal(y16.t, y32.t.half())
    rec = (B * H * W) // 64
    assert torch.equal(y16.cols.buf[:rec], y32.cols.buf[:rec]) and y32.cols.tile_rows == 64
    st = ops.group_norm_stats(y16, 32, 1e-5).cpu()
    v = ref.double().reshape(B, H * W, 32, N // 32)
    mean, var = v.mean(dim=(1, 3)), v.var(dim=(1, 3), unbiased=False)
    close(st[:, :, 0], mean.float(), rtol=1e-4, atol=1e-5)
    close(st[:, :, 1], (var + 1e-5).rsqrt().float(), rtol=1e-4, atol=1e-5)
    if B > 1:       # shard invariance: the first sample alone
        one = ops.conv2d_stem(x[:1].contiguous().to(dev), w16, N, bias=bias.to(dev), colstats=True, out_f16=True)
        assert torch.equal(one.t, y16.t[:1]) and torch.equal(one.cols.buf[:(H * W) // 64], y16.cols.buf[:(H * W) // 64])
    # the dispatcher binding runs the same kernel
    import diffpure_amd.torch_ops  # noqa: F401
    o2, c2 = torch.ops.diffpure_hip.conv2d_stem(x.to(dev), w16, bias.to(dev), True, True)
    assert torch.equal(o2, y16.t) and torch.equal(c2[:rec], y16.cols.buf[:rec])


def test_stem_kernel_rejects_shapes_it_does_not_serve(dev):
    from diffpure_amd import ops, _lib
    assert not ops.conv2d_stem_ok(4, 2, 32, 32, 128) and not ops.conv2d_stem_ok(3, 1, 5, 5, 128) and not ops.conv2d_stem_ok(3, 2, 32, 32, 96)
    with pytest.raises(_lib.DiffpureHipError):
        ops.conv2d_stem(torch.zeros(1, 5, 5, 3, device=dev), ops.pack_stem_weight(torch.zeros(128, 3, 3, 3)).to(dev), 128)


# ---- round 6: the fused block boundary of the <= 64-pixel levels (csrc/boundary.hip) -----------------------------------------------
BOUNDARY_CASES = [
    # B, H, C, N, C2, G, out16, temb, res, film, act, raw
    (3, 8, 256, 256, 0, 32, True, True, False, False, True, False),      # NCSN++ Conv_0 -> GroupNorm_1 at 8 x 8 (additive temb)
    (3, 8, 256, 256, 0, 32, True, False, True, False, True, False),      # Conv_1 (+ residual, 1/sqrt 2) -> next block's GroupNorm_0
    (2, 8, 256, 256, 256, 32, True, False, True, False, True, False),    # ... over a skip concatenation (up path)
    (5, 4, 256, 256, 0, 32, False, True, False, False, True, False),     # 4 x 4: fp32 stream, four split-K parts
    (4, 4, 512, 256, 256, 32, False, False, True, False, True, True),    # 4 x 4 up path: concatenation + the raw operand of the 1x1 shortcut
    (2, 8, 1024, 1024, 0, 32, True, False, True, True, True, False),     # guided 8 x 8: FiLM, 1024 channels = four channel blocks
    (2, 8, 1024, 1024, 1024, 32, True, False, True, False, True, False), # guided 8 x 8 output block: 2048-channel concatenation
    (2, 8, 1024, 1024, 0, 32, True, False, True, False, False, False),   # attention block's GroupNorm (no SiLU)
    (2, 8, 384, 128, 0, 32, True, True, False, False, True, False),      # 128-channel blocks
]


@pytest.mark.batch_invariant        # (its last paragraph compares a sample alone with the same sample inside the batch)
@pytest.mark.parametrize("case", BOUNDARY_CASES, ids=str)
def test_fused_block_boundary_equals_the_four_launch_chain(dev, case):
    """dp_conv2d_nhwc_h2_partials + dp_splitk_gn against conv2d_h2 (its own reduction + epilogue) -> group_norm_stats -> group_norm: the stream
    tensor is IDENTICAL (same partial sums, same reduction order, same epilogue arithmetic), the statistics agree to fp32 summation noise,
    the operand up to one fp16 ulp in a few elements (the statistics differ in the last place; SiLU on the hardware units in both), and its
    border is zero; Deferred.resolve() - the fallback for consumers the fused form does not serve - gives the convolution's own result."""
    from diffpure_amd import ops
    B, H, C, N, C2, G, out16, has_temb, has_res, has_film, act, raw = case
    W = H
    assert ops.conv_defers(H, W, 3, C, N) and ops.splitk_gn_ok(H, W, N, C2, G)
    x = _h1_bordered(rnd(B, H, W, C, seed=1), dev)
    w16 = ops.order_conv_weight_w16(rnd(N, C, 3, 3, seed=2, scale=1.0 / math.sqrt(9 * C))).half().to(dev)
    bias = rnd(N, seed=3).to(dev)
    table = rnd(B, N + 8, seed=4).to(dev) if has_temb else None
    temb = None if table is None else table[:, 4:4 + N]
    res = None
    if has_res:
        res = rnd(B, H, W, N, seed=5).to(dev)
        res = res.half() if out16 else res
    x2 = None
    if C2:
        x2 = (rnd(B, H, W, C2, seed=6) * 1.5 + 0.2).to(dev)
        x2 = x2.half() if out16 else x2
    Ct = N + C2
    gamma, beta = (1 + 0.1 * rnd(Ct, seed=7)).to(dev), (0.1 * rnd(Ct, seed=8)).to(dev)
    ftab = (0.3 * rnd(B, 2 * Ct, seed=9)).to(dev) if has_film else None
    film = None if ftab is None else (ftab[:, :Ct], ftab[:, Ct:])
    scale = 0.70710678
    kw = dict(bias=bias, temb=temb, res=res, scale=scale, w_fmt=1, out_f16=out16)
    # the four-launch chain
    ya = ops.conv2d_h2(x, w16, N, 3, colstats=True, **kw)
    if out16:
        x2a = None if x2 is None else ops.Act(x2, _records_of(x2, dev))
        st = ops.group_norm_stats(ya, G, 1e-6, x2a)
        ref = ops.group_norm(ya.t, G, 1e-6, gamma, beta, x2=x2, film=film, act=act, split="h1", stats=st, raw=raw)
    else:
        st = ops.group_norm_stats(ya.t, G, 1e-6, x2)
        ref = ops.group_norm(ya.t, G, 1e-6, gamma, beta, x2=x2, film=film, act=act, split="h1", stats=st, raw=raw)
    ref, ref_raw = ref if raw else (ref, None)
    # the fused boundary
    d = ops.conv2d_h2(x, w16, N, 3, defer=True, **kw)
    assert isinstance(d, ops.Deferred) and not d.resolved and d.shape == ya.t.shape and d.dtype == ya.t.dtype
    assert ops.deferred_fusable(d, x2, G)
    y, st2, yr = ops.group_norm_deferred(d, G, 1e-6, gamma, beta, x2=x2, film=film, act=act, raw=raw, want_out=True, want_stats=True)
    assert d.resolved and torch.equal(d.t, ya.t)
    close(st2[:, :, 0], st[:, :, 0].cpu(), rtol=1e-5, atol=1e-6)
    close(st2[:, :, 1], st[:, :, 1].cpu(), rtol=1e-5, atol=0)
    assert y.shape == ref.shape and y.dtype == torch.float16
    diff = (y.float() - ref.float()).abs()
    assert (diff <= ref.float().abs() * 2.0 ** -9 + 2e-3 * (0 if out16 else 1) + 1e-6).all(), diff.max().item()
    assert (y != ref).float().mean().item() < (0.02 if out16 else 0.5), (y != ref).float().mean().item()
    border = torch.ones_like(y, dtype=torch.bool)
    border[:, 1:-1, 1:-1, :] = False
    assert (y[border] == 0).all()
    if raw:
        assert torch.equal(yr, ref_raw)
    if H * W == 64:     # the sample's one 64-row record: what group_norm_stats of a later consumer reads
        rec = ya.cols.buf[:B]
        close(d.cols.buf[:B], rec.cpu(), rtol=1e-5, atol=1e-4)
        st3 = ops.group_norm_stats(d, G, 1e-6, None) if C2 == 0 else None
        if st3 is not None:
            close(st3, st.cpu(), rtol=1e-5, atol=1e-6)
    else:
        assert d.cols is None
    # the fallback: the plain reduction + epilogue on the partial sums
    d2 = ops.conv2d_h2(x, w16, N, 3, defer=True, **kw)
    assert torch.equal(ops.tensor_of(d2), ya.t) and torch.equal(d2.cols.buf[:(B * H * W + 63) // 64], ya.cols.buf[:(B * H * W + 63) // 64])
    # without the stream tensor (the tensor between a block's two convolutions, no tape): same operand
    d3 = ops.conv2d_h2(x, w16, N, 3, defer=True, **kw)
    y3, s3, _ = ops.group_norm_deferred(d3, G, 1e-6, gamma, beta, x2=x2, film=film, act=act, want_out=False)
    assert torch.equal(y3, y) and s3 is None and d3.t is None
    # batch invariance: the first sample alone
    d4 = ops.conv2d_h2(x[:1].contiguous(), w16, N, 3, defer=True, bias=bias, temb=None if temb is None else table[:1, 4:4 + N],
                       res=None if res is None else res[:1].contiguous(), scale=scale, w_fmt=1, out_f16=out16)
    y4, _, _ = ops.group_norm_deferred(d4, G, 1e-6, gamma, beta, x2=None if x2 is None else x2[:1].contiguous(),
                                       film=None if film is None else (ftab[:1, :Ct], ftab[:1, Ct:]), act=act)
    assert torch.equal(y4, y[:1]) and torch.equal(d4.t, d.t[:1])


def _records_of(t16, dev):
    """64-row column records of an fp16 NHWC tensor (stand-in for what its producing convolution would have left)"""
    from diffpure_amd import ops
    b, h, w, c = t16.shape
    m = b * h * w
    v = t16.float().reshape(m // 64, 64, c)
    buf, _ = ops._colstats_alloc(m, c, dev)
    buf[:m // 64, 0] = v.sum(dim=1)
    buf[:m // 64, 1] = (v * v).sum(dim=1)
    return ops.ColStats(buf, 64, c)


BUCKET_CASES = [(4, 32, 512, 512, 0), (4, 32, 1024, 512, 512), (4, 16, 1024, 1024, 0), (8, 16, 2048, 1024, 0), (4, 8, 1024, 1024, 0), (2, 64, 512, 512, 0)]


@pytest.mark.parametrize("case", BUCKET_CASES, ids=str)
def test_batch_bucketed_split_k_of_few_tile_launches(dev, case, tune):
    """Round 6: at small per-GPU batches (the reference's own 4 images per GPU) the middle levels of the guided UNet are a few dozen tiles with
    144 - 576 sequential k-tiles each; by default such launches are split along K per (layer shape, batch bucket).  Against the fp64
    convolution of the fp16-rounded operands (fp32-accumulation accuracy), against the shape-only rule (DIFFPURE_BATCH_INVARIANT=1: the same
    sums in another order - fp32 noise), with a residual, fp16 output, column records that give the same GroupNorm statistics, and 1x1
    K-segments running through the split; reproducible run to run."""
    from diffpure_amd import ops, _lib
    B, H, C, N, CS = case
    lib = _lib.load()
    parts = lib.dp_conv2d_nhwc_h2_workspace(B, H, H, 3, C, N) // (B * H * H * N * 4)
    tune.setenv("DIFFPURE_BATCH_INVARIANT", 1)
    parts_inv = lib.dp_conv2d_nhwc_h2_workspace(B, H, H, 3, C, N) // (B * H * H * N * 4)
    tune.delenv("DIFFPURE_BATCH_INVARIANT")
    assert parts in (2, 4, 8) and parts > parts_inv, (parts, parts_inv)
    x = rnd(B, H, H, C, seed=1)
    w3 = rnd(N, C, 3, 3, seed=2, scale=1.0 / math.sqrt(9 * C))
    bias = rnd(N, seed=3).to(dev)
    res = rnd(B, H, H, N, seed=4).half().to(dev)
    segs, ws = None, None
    if CS:
        ws = rnd(N, CS, 1, 1, seed=5, scale=1.0 / math.sqrt(CS))
        segs = (rnd(B, H, H, CS, seed=6).half().to(dev),)
        wf = ops.order_conv_weight_w16(ops.fuse_skip_weight(w3, ws)).half().to(dev)
    else:
        wf = ops.order_conv_weight_w16(w3).half().to(dev)
    xh = _h1_bordered(x, dev)
    run = lambda f16: ops.conv2d_h2(xh, wf, N, 3, bias=bias, res=res, scale=0.5, colstats=True, w_fmt=1, out_f16=f16, segs=segs)
    y32, y16 = run(False), run(True)
    again = run(True)
    assert torch.equal(again.t, y16.t) and torch.equal(again.cols.buf, y16.cols.buf)
    assert torch.equal(y16.t, y32.t.half()) and torch.equal(y16.cols.buf, y32.cols.buf)
    tune.setenv("DIFFPURE_BATCH_INVARIANT", 1)
    inv = run(False)
    tune.delenv("DIFFPURE_BATCH_INVARIANT")
    close(y32.t, inv.t.cpu(), rtol=2e-5, atol=2e-5)
    sub = slice(0, 1)
    ref = torch.nn.functional.conv2d(x[sub].half().double().permute(0, 3, 1, 2), w3.half().double(), None, padding=1).permute(0, 2, 3, 1) + bias.cpu().double()
    if CS:
        ref = ref + segs[0][sub].cpu().double() @ ws[:, :, 0, 0].half().double().t()
    ref = ((ref + res[sub].cpu().double()) * 0.5).float()
    close(y32.t[sub], ref, rtol=1e-4, atol=1e-4)
    close(ops.group_norm_stats(y16, 32, 1e-5), ops.group_norm_stats(inv, 32, 1e-5).cpu(), rtol=1e-4, atol=1e-5)
