"""TEST INFRASTRUCTURE: plain-PyTorch fp32 statements of every `diffpure_amd.ops` contract.

Two uses:
  * `-m gpu` tests compare each HIP operator against the function of the same name here;
  * `-m "not gpu"` tests monkeypatch `diffpure_amd.ops` with these (see `patch_ops`) so that the
    HOST logic of the engine - block wiring, weight packing, channel-split loaders, time tables,
    schedules - is checked on CPU against the oracle / the reference's golden vectors.
The product never imports this module.
"""
import math

import numpy as np
import os

import torch
import torch.nn.functional as F

RESAMPLE_NONE, RESAMPLE_UP, RESAMPLE_DOWN = 0, 1, 2


def _cat(x, x2):
    return x if x2 is None else torch.cat([x, x2], dim=3)


def _up(t):
    """plain fp16 tensors (the fp16 residual stream) are evaluated in fp32; fp32 / fp64 inputs keep their type"""
    return t.float() if t.dtype == torch.float16 else t


def h2_encode(t):
    """fp32 [..., C] -> h2 [..., 2C] fp16: per 8 channels, 8 hi then 8 lo."""
    shp = t.shape
    v = t.reshape(-1, shp[-1] // 8, 8).float()
    hi = v.half()
    lo = (v - hi.float()).half()
    return torch.stack([hi, lo], dim=2).reshape(*shp[:-1], 2 * shp[-1])


def h2_decode(t):
    shp = t.shape
    v = t.reshape(-1, shp[-1] // 16, 2, 8).double()
    return (v[:, :, 0] + v[:, :, 1]).reshape(*shp[:-1], shp[-1] // 2)


def pack_h2(t):
    return h2_encode(t)


def pack_conv_weight_h2(w, device=None):
    """k' = (c32 * KH*KW + tap) * 32 + ci % 32 (see csrc/igemm_h2.hip)."""
    if w.dim() == 2:
        w = w[:, :, None, None]
    elif w.dim() == 3:
        w = w[:, :, :, None]
    o, i, kh, kw = w.shape
    return h2_encode(w.float().reshape(o, i // 32, 32, kh, kw).permute(0, 1, 3, 4, 2).reshape(o, kh * kw * i))


def _h2_hi(t):
    """h2 [..., 2C] -> the hi halves only ([..., C], double)."""
    shp = t.shape
    return t.reshape(-1, shp[-1] // 16, 2, 8).double()[:, :, 0].reshape(*shp[:-1], shp[-1] // 2)


def _h2_ksplit(h, w, ksize, c, n_out):
    """igemm_h2.hip::h2_ksplit - the split-K factor of a layer (a function of its shape only)"""
    nt = ksize * ksize * c // 32
    if h * w > 64 or n_out % 4 != 0:
        return 1
    if h * w <= 16 and nt >= 32 and nt % 4 == 0:
        return 4
    if nt >= 16 and nt % 2 == 0:
        return 2
    return 1


def conv_defers(h, w, ksize, c, n_out):
    """dp_conv2d_nhwc_h2_splits_by_shape (the shape-only rule: <= 64-pixel levels)"""
    return os.environ.get("DIFFPURE_BOUNDARY", "1") != "0" and _h2_ksplit(h, w, ksize, c, n_out) > 1


def splitk_gn_ok(h, w, n, c2, groups):
    """dp_splitk_gn_ok"""
    hw, c = h * w, n + c2
    if hw not in (64, 16) or c % groups != 0:
        return False
    cpg, cb = c // groups, (256 if n % 256 == 0 and c2 % 256 == 0 else 128)
    return cpg % 4 == 0 and n % cb == 0 and c2 % cb == 0 and cb % cpg == 0 and cb // cpg <= 64


def _deferred_value(d):
    """the finished value of a stand-in Deferred: its `ws` holds the plain convolution sum (one part)"""
    b, h, w, n = d.dims
    y = d.ws.double()
    if d.bias is not None:
        y = y + d.bias[:n]
    if d.temb is not None:
        y = y + d.temb[:, :n].reshape(-1, 1, 1, n)
    if d.res is not None:
        y = y + d.res.double()
    return (y * d.scale).float().contiguous()


def deferred_resolve(d):
    from diffpure_amd import ops
    if d.t is None:
        y = _deferred_value(d)
        d.t, d.cols, d.ws = (y.half() if d.out_f16 else y), None, None
    return d


def group_norm_deferred(d, groups, eps, gamma, beta, x2=None, film=None, act=False, raw=False, want_out=True, want_stats=False):
    """dp_splitk_gn: split-K reduction + epilogue + GroupNorm (+FiLM) (+SiLU) of cat(d, x2) -> (operand, stats | None, raw operand | None)"""
    from diffpure_amd import ops
    v = _deferred_value(d)
    x2 = ops.tensor_of(x2)
    stored = v.half().float() if d.out_f16 else v
    full = stored if x2 is None else torch.cat([stored, x2.float()], dim=3)
    unr = v if x2 is None else torch.cat([v, x2.float()], dim=3)
    stats = group_norm_stats(unr, groups, eps)
    b, h, w, c = full.shape
    mean, rstd = stats[:, :, 0], stats[:, :, 1]
    y = (full.reshape(b, h * w, groups, c // groups) - mean[:, None, :, None]) * rstd[:, None, :, None]
    y = y.reshape(b, h, w, c) * gamma + beta
    if film is not None:
        fs, fh = film
        y = y * (1 + fs.reshape(-1, 1, 1, fs.shape[-1])) + fh.reshape(-1, 1, 1, fh.shape[-1])
    if act:
        y = F.silu(y)
    if want_out:
        d.t, d.cols = (v.half() if d.out_f16 else v), None
    d.ws = None
    pad = lambda t: F.pad(t, (0, 0, 1, 1, 1, 1)).half().contiguous()
    return pad(y), (stats if want_stats else None), (pad(full) if raw else None)


def conv2d_h2(x, wh, n_out, ksize, bias=None, temb=None, res=None, scale=1.0, colstats=False, passes=None, w_fmt=0, out_f16=False,
              segs=None, defer=False):
    """Statement of the fp16-matrix-core contract (include/diffpure_hip.h, dp_conv2d_nhwc_h2): exact products of the
    operands each mode keeps - f16x3 (h2 activations, passes 3): (hi+lo) x (hi+lo) (the dropped lo*lo term is ~2^-22
    relative, below the test tolerance); passes 12: (hi+lo) x w_hi; h1 activations (plain fp16): passes 2: a x (w_hi+w_lo),
    passes 1: a x w_hi.  x carries a one-pixel zero border.  out_f16: the result rounded to fp16 (out_fmt 1).  res: fp32 or fp16.
    segs: 1x1 K-segments - plain fp16 NHWC tensors whose weight columns follow the KS x KS part in the panel."""
    wseg = None
    if w_fmt:
        from diffpure_amd import ops
        wh = ops.unorder_conv_weight_w16(wh, n_out)          # block layout of the fp16 panels -> [N, K'] reduction order
    if segs:
        assert w_fmt == 1
        cs = sum(sg.shape[3] for sg in segs)
        wh, wseg = wh[:, :wh.shape[1] - cs], wh[:, wh.shape[1] - cs:]
    cin = wh.shape[1] // ((1 if w_fmt else 2) * ksize * ksize)
    h1 = x.shape[3] == cin
    if passes is None:
        passes = 1 if w_fmt else (2 if h1 else 3)
    assert (h1 and passes in (1, 2)) or (not h1 and passes in (3, 12)), (x.shape, wh.shape, passes)
    assert not w_fmt or (h1 and passes == 1)
    xin = (x.double() if h1 else h2_decode(x))[:, 1:-1, 1:-1, :]
    wdec = wh.double() if w_fmt else (_h2_hi(wh) if passes in (1, 12) else h2_decode(wh))
    wf = wdec.reshape(n_out, cin // 32, ksize, ksize, 32)       # [N, c32, ky, kx, 32]
    wt = wf.permute(0, 1, 4, 2, 3).reshape(n_out, cin, ksize, ksize).contiguous()
    y = F.conv2d(xin.permute(0, 3, 1, 2), wt, None, padding=ksize // 2).permute(0, 2, 3, 1)
    if wseg is not None:
        y = y + torch.cat([sg.double() for sg in segs], dim=3) @ wseg.double().t()
    if defer:       # the split-K partial sums only (one part here); the epilogue belongs to the consumer (ops.Deferred)
        from diffpure_amd import ops
        assert w_fmt == 1 and h1 and conv_defers(xin.shape[1], xin.shape[2], ksize, cin, n_out)
        return ops.Deferred(y.float().contiguous(), 1, bias, temb, 0, res, scale, bool(out_f16), tuple(y.shape))
    if bias is not None:
        y = y + bias[:n_out]
    if temb is not None:
        y = y + temb[:, :n_out].reshape(-1, 1, 1, n_out)
    if res is not None:
        y = y + res.double()
    y = (y * scale).float().contiguous()
    if out_f16:
        y = y.half()
    return _act(y) if colstats else y


def takes_segments(h, w, ksize, c, n_out, c1, c2=0):
    """dp_conv2d_nhwc_h2_takes_segments: whole 32-channel slices everywhere; a function of the layer shape only"""
    return c % 32 == 0 and c1 > 0 and c1 % 32 == 0 and c2 % 32 == 0 and n_out % 4 == 0


def group_norm_f16in(x16, groups, gamma, beta, stats, film=None, act=False):
    """dp_gn_apply_f16in: GroupNorm-apply of the fp16 tensor (statistics as given) -> bordered "h1" operand"""
    b, h, w, c = x16.shape
    mean, rstd = stats[:, :, 0], stats[:, :, 1]
    y = (x16.float().reshape(b, h * w, groups, c // groups) - mean[:, None, :, None]) * rstd[:, None, :, None]
    y = y.reshape(b, h, w, c) * gamma + beta
    if film is not None:
        fs, fh = film
        y = y * (1 + fs.reshape(-1, 1, 1, fs.shape[-1])) + fh.reshape(-1, 1, 1, fh.shape[-1])
    if act:
        y = F.silu(y)
    return F.pad(y, (0, 0, 1, 1, 1, 1)).half().contiguous()


def attention_fused_ok(t, d):
    return (d == 64 and t % 64 == 0) or (d == 256 and t % 128 == 0)


def attention_fused(qkv, n_heads, layout, operand_hw=None):
    out = attention(_up(qkv), n_heads, layout)
    if operand_hw is None:
        return out
    hh, ww = operand_hw
    return F.pad(out.reshape(out.shape[0], hh, ww, out.shape[2]), (0, 0, 1, 1, 1, 1)).half().contiguous()


def round_weights(master, work, stochastic, seed, key):
    """torch statement of dp_round_weights: round to nearest, or unbiased stochastic rounding (a DIFFERENT random stream
    than the kernel's Philox bits - the host-logic tests need the contract, not the bits)."""
    if not stochastic:
        work.copy_(master.half())
        return
    g = torch.Generator().manual_seed((int(seed) * 1000003 + int(key)) & 0x7FFFFFFF)
    aw = master.abs()
    hn = aw.half()
    bits = hn.view(torch.int16).clone()
    bits -= (hn.float() > aw).to(torch.int16)                     # towards zero
    h0 = bits.view(torch.float16)
    h1 = (bits + 1).view(torch.float16)
    f0, f1 = h0.float(), h1.float()
    p = torch.where(aw > f0, (aw - f0) / (f1 - f0), torch.zeros_like(aw))
    up = torch.rand(aw.shape, generator=g) < p
    work.copy_(torch.where(up, h1, h0) * torch.sign(master).half())


def _act(y):
    from diffpure_amd import ops
    return ops.Act(y, None)


def conv2d(x, wp, n_out, ksize, bias=None, x2=None, temb=None, res=None, scale=1.0, out=None, colstats=False, out_f16=False):
    xin = _cat(x, x2)
    b, h, w, cin = xin.shape
    wt = wp[:, :n_out].reshape(ksize, ksize, cin, n_out).permute(3, 2, 0, 1).contiguous()
    y = F.conv2d(xin.permute(0, 3, 1, 2), wt, None, padding=ksize // 2).permute(0, 2, 3, 1)
    if bias is not None:
        y = y + bias[:n_out]
    if temb is not None:
        y = y + temb[:, :n_out].reshape(-1, 1, 1, n_out)
    if res is not None:
        y = y + res
    y = (y * scale).contiguous()
    if out_f16:
        y = y.half()
    if out is not None:
        out.copy_(y)
        y = out
    return _act(y) if colstats else y


def conv2d_stem_ok(cin, b, h, w, n_out):
    return cin == 3 and n_out in (128, 256) and (b * h * w) % 64 == 0


def conv2d_stem(x, wpanel, n_out, bias=None, colstats=False, out_f16=False):
    """the (hi | lo) panel of ops.pack_stem_weight back to OIHW, then the plain convolution"""
    cin = x.shape[3]
    wk = (wpanel[0].float() + wpanel[1].float())[:, :9 * cin].reshape(n_out, 3, 3, cin).permute(0, 3, 1, 2).contiguous()
    y = F.conv2d(x.permute(0, 3, 1, 2), wk, bias, padding=1).permute(0, 2, 3, 1).contiguous()
    if out_f16:
        y = y.half()
    return _act(y) if colstats else y


def linear(x, wp, n_out, bias=None):
    m, k = x.shape
    return conv2d(x.view(m, 1, 1, k), wp, n_out, 1, bias=bias).view(m, n_out)


def group_norm_stats(x, groups, eps, x2=None):
    from diffpure_amd import ops
    xin = _up(_cat(ops.tensor_of(x), ops.tensor_of(x2)))         # (fp16 stream tensors: statistics of the rounded values here)
    b, h, w, c = xin.shape
    v = xin.reshape(b, h * w, groups, c // groups).double()
    mean = v.mean(dim=(1, 3))
    var = v.var(dim=(1, 3), unbiased=False)
    return torch.stack([mean, 1.0 / torch.sqrt(var + eps)], dim=-1).float()


def _upfirdn2d(x_nchw, k2d, up, down, pad0, pad1):
    """torch statement of upfirdn2d (/root/reference/score_sde/op/upfirdn2d.py:167-211, upfirdn2d_native): zero insertion,
    zero padding, correlation with the flipped kernel, decimation."""
    n, c, h, w = x_nchw.shape
    u = x_nchw.new_zeros(n, c, h * up, w * up)
    u[:, :, ::up, ::up] = x_nchw
    u = F.pad(u, (pad0, pad1, pad0, pad1))
    wk = torch.flip(k2d, [0, 1])[None, None].to(u.dtype)
    out = F.conv2d(u.reshape(n * c, 1, u.shape[2], u.shape[3]), wk)
    return out[:, :, ::down, ::down].reshape(n, c, out.shape[2] // down + (out.shape[2] % down > 0), -1)


def _fir_resample(y, mode, fir):
    from diffpure_amd import ops
    k = torch.tensor(fir, dtype=torch.float64)
    k2 = torch.outer(k, k)
    x = y.permute(0, 3, 1, 2).double()
    if mode == ops.RESAMPLE_FIR_UP:        # upsample_2d: k * factor^2, pad ((p+1)//2 + factor-1, p//2), p = 4 - 2
        out = _upfirdn2d(x, k2 * 4, 2, 1, 2, 1)
    else:                                  # downsample_2d: pad ((p+1)//2, p//2)
        out = _upfirdn2d(x, k2, 1, 2, 1, 1)
    return out.permute(0, 2, 3, 1).to(y.dtype)


def _resample(y, mode, fir=None):
    if mode in (3, 4):
        return _fir_resample(y, mode, fir)
    if mode == RESAMPLE_UP:
        return y.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2)
    if mode == RESAMPLE_DOWN:
        b, h, w, c = y.shape
        return y.reshape(b, h // 2, 2, w // 2, 2, c).mean(dim=(2, 4))
    return y


def group_norm(x, groups, eps, gamma, beta, x2=None, film=None, act=False, resample=RESAMPLE_NONE, stats=None,
               split=False, raw=False, fir=None):
    xin = _up(_cat(x, x2))                                       # fp32 / fp64, or plain fp16 on the fp16 residual stream
    y = F.group_norm(xin.permute(0, 3, 1, 2), groups, gamma, beta, eps).permute(0, 2, 3, 1)
    if film is not None:
        fs, fh = film
        y = y * (1 + fs.reshape(-1, 1, 1, fs.shape[-1])) + fh.reshape(-1, 1, 1, fh.shape[-1])
    if act:
        y = F.silu(y)
    y = _resample(y, resample, fir).contiguous()
    enc = _operand_encoder(split)
    if raw and resample:       # fp16 stream: the resampled raw input as a plain fp16 tensor (dp_gn_apply_h16, ABI 8)
        return enc(F.pad(y, (0, 0, 1, 1, 1, 1))), _resample(xin.float(), resample, fir).half().contiguous()
    if raw:
        return enc(F.pad(y, (0, 0, 1, 1, 1, 1))), enc(F.pad(xin, (0, 0, 1, 1, 1, 1)))
    return enc(F.pad(y, (0, 0, 1, 1, 1, 1))) if enc else y


def _operand_encoder(split):
    """split argument of ops.group_norm / fmt of ops.to_h2 -> encoder of the bordered operand (None = fp32 output)."""
    from diffpure_amd import ops
    fmt = ops._fmt_of(split)
    return {ops.FMT_F32: None, ops.FMT_H2: h2_encode, ops.FMT_H1: lambda t: t.half()}[fmt]


def resample(x, mode, fir=None):
    if x.dtype == torch.float16:                                 # the fp16 residual stream stays fp16 (dp_gn_apply_h16 out_fmt 3)
        return _resample(x.float(), mode, fir).half().contiguous()
    return _resample(x, mode, fir).contiguous()


def attention(qkv, n_heads, layout, return_probs=False, probs_only=False, h16=False):
    """(h16 - fp16 matrix cores for the gradient path - has no CPU counterpart: the stand-in computes in fp32 either way)"""
    if probs_only:
        return None, attention(qkv, n_heads, layout, return_probs=True)[1]
    qkv = qkv.float()            # (the taped fp16 qkv is read in place by the GPU kernels: same values)
    b, t, c3 = qkv.shape
    c = c3 // 3
    d = c // n_heads
    if layout == "legacy":
        v = qkv.reshape(b, t, n_heads, 3, d)
        q, k, vv = v[:, :, :, 0], v[:, :, :, 1], v[:, :, :, 2]
    else:
        v = qkv.reshape(b, t, 3, n_heads, d)
        q, k, vv = v[:, :, 0], v[:, :, 1], v[:, :, 2]
    w = torch.einsum("bthd,bshd->bhts", q, k) / math.sqrt(d)
    w = torch.softmax(w, dim=-1)
    out = torch.einsum("bhts,bshd->bthd", w, vv).reshape(b, t, c).contiguous()
    return (out, w.reshape(b * n_heads, t, t).contiguous()) if return_probs else out


def attention_h16_serves(t, d):
    return t % 128 == 0 and t % 32 == 0 and d % 64 == 0 and d % 32 == 0


def attention_bwd(qkv, probs, dout, n_heads, layout, h16=False):
    with torch.enable_grad():
        q = qkv.detach().float().clone().requires_grad_(True)
        out = attention(q, n_heads, layout)
        (g,) = torch.autograd.grad(out, q, dout)
    return g


def gn_bwd_fused_ok(h, w, c1, c2, groups, resample):
    return False


def group_norm_bwd(x, groups, gamma, beta, stats, dy, x2=None, film=None, act=False, resample=RESAMPLE_NONE, split=False,
                   fir=None, addend=None, addend2=None, addend_scale=1.0, one_pass=None):
    """autograd through the torch statement of the forward (eps folded back out of `stats`); dx += addend_scale * addend."""
    with torch.enable_grad():
        # (x / x2 may be the fp16 tensors of a tape that ran on the fp16 residual stream: same values, gradients are fp32)
        a = x.detach().float().clone().requires_grad_(True)
        b2 = None if x2 is None else x2.detach().float().clone().requires_grad_(True)
        xin = _cat(a, b2)
        bb, h, w, c = xin.shape
        v = xin.reshape(bb, h * w, groups, c // groups)
        mean = stats[..., 0].reshape(bb, 1, groups, 1)
        rstd = stats[..., 1].reshape(bb, 1, groups, 1)
        # mean/rstd are functions of x: rebuild them differentiably, using the forward's eps
        mu = v.mean(dim=(1, 3), keepdim=True)
        var = v.var(dim=(1, 3), unbiased=False, keepdim=True)
        eps = (1.0 / rstd ** 2 - var.detach()).clamp_min(0)
        y = ((v - mu) / torch.sqrt(var + eps)).reshape(bb, h, w, c) * gamma + beta
        if film is not None:
            fs, fh = film
            y = y * (1 + fs.reshape(-1, 1, 1, c)) + fh.reshape(-1, 1, 1, c)
        if act:
            y = F.silu(y)
        y = _resample(y, resample, fir)
        grads = torch.autograd.grad(y, [a] + ([] if b2 is None else [b2]), dy)
    dx = grads[0]
    dx2 = grads[1] if b2 is not None else None
    if addend is not None:
        dx = dx + addend_scale * addend
    if addend2 is not None:
        dx2 = dx2 + addend_scale * addend2
    if split == "h1":
        return F.pad(dx, (0, 0, 1, 1, 1, 1)).half(), None
    if split:
        return h2_encode(F.pad(dx, (0, 0, 1, 1, 1, 1))), None
    return dx.contiguous(), None if dx2 is None else dx2.contiguous()


def resample_bwd(dy, mode, fir=None):
    b, ho, wo, c = dy.shape
    h, w = (ho // 2, wo // 2) if mode in (RESAMPLE_UP, 3) else (ho * 2, wo * 2)
    with torch.enable_grad():
        x = torch.zeros(b, h, w, c, dtype=dy.dtype, requires_grad=True)
        (g,) = torch.autograd.grad(_resample(x, mode, fir), x, dy)
    return g.contiguous()


def fir_adjoint_stencil(dy, mode, fir):
    """The transposed FIR stencils as csrc/norm_bwd.hip fir_adjoint evaluates them (pure indexing, no autograd):
    up   (3): dx[i]  = 2 (k0 dy[2i-1] + k1 dy[2i] + k2 dy[2i+1] + k3 dy[2i+2])   per axis
    down (4): dx[2n] = k2 dy[n] + k0 dy[n-1],  dx[2n+1] = k1 dy[n] + k3 dy[n+1]  per axis;  dy zero outside."""
    k = [float(v) for v in fir]

    def axis(t, dim):
        t = t.movedim(dim, 0)
        n = t.shape[0]
        z = torch.zeros_like(t[:1])
        if mode == 3:
            pad = torch.cat([z, t, z, z], 0)                   # pad[j] = dy[j - 1]
            out = sum(2 * k[a] * pad[a:a + n:2][: n // 2] for a in range(4))   # dy[2i - 1 + a]
        else:
            prev = torch.cat([z, t[:-1]], 0)                   # dy[n-1]
            nxt = torch.cat([t[1:], z], 0)                     # dy[n+1]
            out = torch.stack([k[2] * t + k[0] * prev, k[1] * t + k[3] * nxt], 1).reshape(2 * n, *t.shape[1:])
        return out.movedim(0, dim)

    return axis(axis(dy, 1), 2).contiguous()


def add(a, b):
    return a + b


def to_h2(x, mode=RESAMPLE_NONE, fmt="h2", fir=None):
    return _operand_encoder(fmt)(F.pad(_resample(_up(x), mode, fir), (0, 0, 1, 1, 1, 1)))


def silu(x):
    return F.silu(x)


def axpby(x, a, y, b):
    return x * a + y * b


def timestep_embedding(t, freqs, cos_first):
    a = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(a), torch.sin(a)] if cos_first else [torch.sin(a), torch.cos(a)], dim=-1)


# ---- Philox4x32-10 + Box-Muller, numpy restatement of csrc/elementwise.hip ----------------------
_M0, _M1, _W0, _W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    c0, c1, c2, c3 = (np.asarray(v, dtype=np.uint64) for v in (c0, c1, c2, c3))
    k0, k1 = np.uint64(k0), np.uint64(k1)
    mask = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = np.uint64(_M0) * c0
        p1 = np.uint64(_M1) * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & mask
        hi1, lo1 = p1 >> np.uint64(32), p1 & mask
        c0, c1, c2, c3 = (hi1 ^ c1 ^ k0) & mask, lo1, (hi0 ^ c3 ^ k1) & mask, lo0
        k0 = (k0 + np.uint64(_W0)) & mask
        k1 = (k1 + np.uint64(_W1)) & mask
    return c0.astype(np.uint32), c1.astype(np.uint32), c2.astype(np.uint32), c3.astype(np.uint32)


def philox_normal(shape, seed, sample0, step, device=None):
    b = shape[0]
    per = int(np.prod(shape[1:]))
    assert per % 4 == 0
    nq = per // 4
    out = np.empty((b, nq, 4), dtype=np.float32)
    q = np.arange(nq, dtype=np.uint64)
    for i in range(b):
        smp = (sample0 + i) & 0xFFFFFFFFFFFFFFFF
        r = philox4x32_10(q, np.full(nq, (step + 1) & 0xFFFFFFFF, dtype=np.uint64),
                          np.full(nq, smp & 0xFFFFFFFF, dtype=np.uint64), np.full(nq, smp >> 32, dtype=np.uint64),
                          seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
        u = [((v >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 16777216.0) for v in r]
        r0 = np.sqrt(np.float32(-2.0) * np.log(u[0]))
        r1 = np.sqrt(np.float32(-2.0) * np.log(u[2]))
        a0 = np.float32(6.283185307179586) * u[1]
        a1 = np.float32(6.283185307179586) * u[3]
        out[i, :, 0], out[i, :, 1] = r0 * np.cos(a0), r0 * np.sin(a0)
        out[i, :, 2], out[i, :, 3] = r1 * np.cos(a1), r1 * np.sin(a1)
    t = torch.from_numpy(out.reshape(shape))
    return t if device is None else t.to(device)


def em_step(x, eps, neg_half_beta, gg, score_coef, score_div, h, g, sqrt_h, noise=None, seed=0, sample0=0, step=0,
            out=None):
    f32 = lambda v: torch.tensor(v, dtype=torch.float32)
    c = x.shape[3]
    e = eps[..., :c]
    score = (-e) / f32(score_coef) if score_div else f32(score_coef) * e
    drift = f32(neg_half_beta) * x - f32(gg) * score
    if g != 0.0:
        z = noise if noise is not None else philox_normal(tuple(x.shape), seed, sample0, step)
        y = x + (-drift) * f32(h) + f32(g) * (z * f32(sqrt_h))
    else:
        y = x + (-drift) * f32(h)
    if out is not None:
        out.copy_(y)
        return out
    return y


def ddpm_step(x, out6, sr, srm1, c1, c2, min_log, max_log, nonzero, noise=None, seed=0, sample0=0, step=0, out=None):
    c = x.shape[3]
    eps, v = out6[..., :c], out6[..., c:]
    frac = (v + 1) / 2
    logvar = frac * max_log + (1 - frac) * min_log
    x0 = (sr * x - srm1 * eps).clamp(-1, 1)
    mean = c1 * x0 + c2 * x
    if nonzero:
        z = noise if noise is not None else philox_normal(tuple(x.shape), seed, sample0, step)
        y = mean + torch.exp(0.5 * logvar) * z
    else:
        y = mean
    if out is not None:
        out.copy_(y)
        return out
    return y


def _dims(shape, nhwc):
    return (shape[0], shape[3], shape[1], shape[2]) if nhwc else tuple(shape)


def resize_affine(x, size, shift, scale, in_nhwc=False, out_nhwc=False):
    """(F.interpolate(x, size, bilinear, align_corners=False) + shift) * scale with layout choice."""
    xc = x.permute(0, 3, 1, 2) if in_nhwc else x
    y = (torch.nn.functional.interpolate(xc, size=tuple(size), mode="bilinear", align_corners=False) + shift) * scale
    return y.permute(0, 2, 3, 1).contiguous() if out_nhwc else y.contiguous()


def resize_affine_bwd(dy, in_size, scale, in_nhwc=False, out_nhwc=False):
    b, c, ho, wo = _dims(dy.shape, out_nhwc)
    x = torch.zeros((b, c) + tuple(in_size), requires_grad=True)
    with torch.enable_grad():
        y = torch.nn.functional.interpolate(x, size=(ho, wo), mode="bilinear", align_corners=False) * scale
        (dx,) = torch.autograd.grad(y, x, dy.permute(0, 3, 1, 2) if out_nhwc else dy)
    return dx.permute(0, 2, 3, 1).contiguous() if in_nhwc else dx.contiguous()


PATCHED = ["resize_affine", "resize_affine_bwd", "conv2d", "conv2d_stem", "conv2d_stem_ok", "conv2d_h2", "pack_h2", "pack_conv_weight_h2", "linear", "attention_bwd", "group_norm_bwd", "gn_bwd_fused_ok",
           "resample_bwd", "add", "to_h2", "group_norm_stats", "group_norm", "group_norm_f16in", "resample", "attention", "attention_fused",
           "attention_fused_ok", "attention_h16_serves", "silu", "axpby", "takes_segments", "conv_defers", "splitk_gn_ok", "group_norm_deferred",
           "timestep_embedding", "philox_normal", "em_step", "ddpm_step"]


def patch_ops(monkeypatch):
    """Route diffpure_amd.ops through the torch-CPU statements above (host-logic tests only)."""
    import sys

    from diffpure_amd import ops

    me = sys.modules[__name__]
    for name in PATCHED:
        monkeypatch.setattr(ops, name, getattr(me, name))

    def pool_round(self, key):
        if not self.stochastic and self._last_key is not None:
            return
        round_weights(self.master, self.work, self.stochastic, self.seed, key)
        self._last_key = key

    monkeypatch.setattr(ops.WeightPool, "round", pool_round)
    monkeypatch.setattr(ops.Deferred, "resolve", deferred_resolve)
