"""Oracle (TEST INFRASTRUCTURE): functional CPU fp32 restatement of score_sde's NCSN++.

Follows /root/reference/score_sde/models/ncsnpp.py (topology :139-230, forward :232-381) for the
configuration family DiffPure ships in configs/cifar10.yml (resblock_type 'biggan', fir False - and, since round 3, fir True
with a separable fir_kernel: upfirdn2d resampling in the BigGAN blocks -, progressive 'none', progressive_input 'none',
embedding_type 'positional', conditional True,
scale_by_sigma False, centered data), score_sde/models/layerspp.py (ResnetBlockBigGANpp :242-274,
AttnBlockpp :75-91), score_sde/models/layers.py (get_timestep_embedding :515-529, NIN :546-555) and
score_sde/models/up_or_down_sampling.py (naive_upsample_2d / naive_downsample_2d :67-77; _setup_kernel :189-200, upsample_2d
:203-233, downsample_2d :236-265) over score_sde/op/upfirdn2d.py (upfirdn2d_native :167-211, the reference's own CPU branch).
Pinned: forward and torch.autograd input gradient against tests/golden/ncsnpp_small.pt, ncsnpp_full.pt and (fir) fir_ops.pt,
ncsnpp_fir_small.pt, all produced by the reference's modules (tests/golden/make_golden*.py).
Consumes the reference's state_dict key names (`all_modules.N.*`).
"""
import math

import torch
import torch.nn.functional as F


def parse_ncsnpp_config(cfg):
    """cfg: the yaml dict of configs/cifar10.yml (keys 'data', 'model')."""
    m, d = cfg["model"], cfg["data"]
    assert m["name"] == "ncsnpp" and m["resblock_type"].lower() == "biggan"
    assert m["progressive"].lower() == "none" and m["progressive_input"].lower() == "none"
    assert m["embedding_type"].lower() == "positional" and m["conditional"]
    assert m["nonlinearity"].lower() == "swish" and not m["scale_by_sigma"] and d["centered"]
    return dict(
        nf=int(m["nf"]),
        ch_mult=tuple(m["ch_mult"]),
        num_res_blocks=int(m["num_res_blocks"]),
        attn_resolutions=tuple(m["attn_resolutions"]),
        image_size=int(d["image_size"]),
        channels=int(d["num_channels"]),
        skip_rescale=bool(m["skip_rescale"]),
        fir=bool(m.get("fir", False)),
        fir_kernel=tuple(m.get("fir_kernel", (1, 3, 3, 1))),
    )


def get_timestep_embedding(timesteps, dim, max_positions=10000):
    # layers.py:515-529 : sin first, then cos; divisor is (half - 1).
    half = dim // 2
    e = math.log(max_positions) / (half - 1)
    e = torch.exp(torch.arange(half, dtype=torch.float32) * -e)
    e = timesteps.float()[:, None] * e[None, :]
    return torch.cat([torch.sin(e), torch.cos(e)], dim=1)


def _gn(sd, p, x):
    c = x.shape[1]
    return F.group_norm(x, min(c // 4, 32), sd[p + ".weight"], sd[p + ".bias"], eps=1e-6)


def _nin(sd, p, x):
    # layers.py:552-555 : y[b,h,w,u] = sum_c x[b,h,w,c] W[c,u] + b[u]
    y = torch.einsum("bchw,cu->buhw", x, sd[p + ".W"])
    return y + sd[p + ".b"][None, :, None, None]


def upfirdn2d(x, k2d, up=1, down=1, pad=(0, 0)):
    """op/upfirdn2d.py:167-211 (upfirdn2d_native with equal factors and paddings on both axes, non-negative pads): insert
    up-1 zeros AFTER every sample, pad (pad[0] before, pad[1] after) with zeros, correlate every channel with the FLIPPED
    kernel ('valid'), keep every down-th output."""
    n, c, h, w = x.shape
    z = x.new_zeros(n, c, h, up, w, up)
    z[:, :, :, 0, :, 0] = x
    z = F.pad(z.reshape(n, c, h * up, w * up), (pad[0], pad[1], pad[0], pad[1]))
    kh, kw = k2d.shape
    out = F.conv2d(z.reshape(n * c, 1, z.shape[2], z.shape[3]), torch.flip(k2d, [0, 1]).reshape(1, 1, kh, kw).to(x.dtype))
    return out.reshape(n, c, out.shape[2], out.shape[3])[:, :, ::down, ::down]


def _fir_kernel2d(k):
    # up_or_down_sampling.py:189-200 (_setup_kernel): a 1-D filter becomes its outer product, normalised to sum 1
    k = torch.tensor(k, dtype=torch.float32)
    k = torch.outer(k, k) if k.dim() == 1 else k
    return k / k.sum()


def upsample_2d(x, k, factor=2):
    # up_or_down_sampling.py:203-233: gain 1 -> kernel * factor^2; p = kH - factor; pad ((p+1)//2 + factor-1, p//2)
    k2 = _fir_kernel2d(k) * factor ** 2
    p = k2.shape[0] - factor
    return upfirdn2d(x, k2, up=factor, pad=((p + 1) // 2 + factor - 1, p // 2))


def downsample_2d(x, k, factor=2):
    # up_or_down_sampling.py:236-265: p = kH - factor; pad ((p+1)//2, p//2)
    k2 = _fir_kernel2d(k)
    p = k2.shape[0] - factor
    return upfirdn2d(x, k2, down=factor, pad=((p + 1) // 2, p // 2))


def _resblock(sd, p, x, temb, cfg, up=False, down=False):
    # layerspp.py:242-274
    h = F.silu(_gn(sd, p + ".GroupNorm_0", x))
    if up and cfg.get("fir"):          # layerspp.py:245-248
        h, x = upsample_2d(h, cfg["fir_kernel"]), upsample_2d(x, cfg["fir_kernel"])
    elif down and cfg.get("fir"):      # layerspp.py:252-255
        h, x = downsample_2d(h, cfg["fir_kernel"]), downsample_2d(x, cfg["fir_kernel"])
    elif up:
        h = h.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)
        x = x.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)
    elif down:
        n, c, hh, ww = h.shape
        h = h.reshape(n, c, hh // 2, 2, ww // 2, 2).mean(dim=(3, 5))
        x = x.reshape(n, x.shape[1], hh // 2, 2, ww // 2, 2).mean(dim=(3, 5))
    h = F.conv2d(h, sd[p + ".Conv_0.weight"], sd[p + ".Conv_0.bias"], padding=1)
    h = h + F.linear(F.silu(temb), sd[p + ".Dense_0.weight"], sd[p + ".Dense_0.bias"])[:, :, None, None]
    h = F.silu(_gn(sd, p + ".GroupNorm_1", h))
    h = F.conv2d(h, sd[p + ".Conv_1.weight"], sd[p + ".Conv_1.bias"], padding=1)
    if (p + ".Conv_2.weight") in sd:
        x = F.conv2d(x, sd[p + ".Conv_2.weight"], sd[p + ".Conv_2.bias"])
    return (x + h) / math.sqrt(2.0) if cfg["skip_rescale"] else x + h


def _attn(sd, p, x, cfg):
    # layerspp.py:75-91
    b, c, hh, ww = x.shape
    h = _gn(sd, p + ".GroupNorm_0", x)
    q, k, v = _nin(sd, p + ".NIN_0", h), _nin(sd, p + ".NIN_1", h), _nin(sd, p + ".NIN_2", h)
    w = torch.einsum("bchw,bcij->bhwij", q, k) * (int(c) ** (-0.5))
    w = F.softmax(w.reshape(b, hh, ww, hh * ww), dim=-1).reshape(b, hh, ww, hh, ww)
    h = torch.einsum("bhwij,bcij->bchw", w, v)
    h = _nin(sd, p + ".NIN_3", h)
    return (x + h) / math.sqrt(2.0) if cfg["skip_rescale"] else x + h


def ncsnpp_forward(sd, cfg, x, time_cond):
    """NCSNpp.forward (ncsnpp.py:232-381). `time_cond` is the float label 999*s."""
    nf, nrb, mults = cfg["nf"], cfg["num_res_blocks"], cfg["ch_mult"]
    nres = len(mults)
    res = [cfg["image_size"] // (2 ** i) for i in range(nres)]
    M = "all_modules."
    temb = get_timestep_embedding(time_cond, nf)
    temb = F.linear(temb, sd[M + "0.weight"], sd[M + "0.bias"])
    temb = F.linear(F.silu(temb), sd[M + "1.weight"], sd[M + "1.bias"])
    i = 2
    hs = [F.conv2d(x, sd[M + f"{i}.weight"], sd[M + f"{i}.bias"], padding=1)]
    i += 1
    for lvl in range(nres):
        for _ in range(nrb):
            h = _resblock(sd, M + str(i), hs[-1], temb, cfg)
            i += 1
            if h.shape[-1] in cfg["attn_resolutions"]:
                h = _attn(sd, M + str(i), h, cfg)
                i += 1
            hs.append(h)
        if lvl != nres - 1:
            hs.append(_resblock(sd, M + str(i), hs[-1], temb, cfg, down=True))
            i += 1
    h = hs[-1]
    h = _resblock(sd, M + str(i), h, temb, cfg); i += 1
    h = _attn(sd, M + str(i), h, cfg); i += 1
    h = _resblock(sd, M + str(i), h, temb, cfg); i += 1
    for lvl in reversed(range(nres)):
        for _ in range(nrb + 1):
            h = _resblock(sd, M + str(i), torch.cat([h, hs.pop()], dim=1), temb, cfg)
            i += 1
        if h.shape[-1] in cfg["attn_resolutions"]:
            h = _attn(sd, M + str(i), h, cfg)
            i += 1
        if lvl != 0:
            h = _resblock(sd, M + str(i), h, temb, cfg, up=True)
            i += 1
    assert not hs
    h = F.silu(_gn(sd, M + str(i), h)); i += 1
    h = F.conv2d(h, sd[M + f"{i}.weight"], sd[M + f"{i}.bias"], padding=1)
    return h
