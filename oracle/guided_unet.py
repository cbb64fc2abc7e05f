"""Oracle (TEST INFRASTRUCTURE): functional CPU fp32 restatement of guided_diffusion's UNetModel.

Follows /root/reference/guided_diffusion/unet.py (UNetModel.__init__ :436-626 for the block
topology, UNetModel.forward :642-671, ResBlock._forward :244-264, AttentionBlock._forward :307-313,
QKVAttentionLegacy.forward :345-362, QKVAttention.forward :377-397, Upsample :89-118,
Downsample :121-148) and guided_diffusion/nn.py (GroupNorm32 :25-27, timestep_embedding :111-129).
Consumes a state_dict with the reference's own key names so that the reference module, this oracle
and the HIP engine can all be fed the same weights.
"""
import math

import torch
import torch.nn.functional as F


def parse_guided_config(model_cfg):
    """Resolve the yaml/`model_and_diffusion_defaults` dict the way script_util.create_model does
    (guided_diffusion/script_util.py:138-192)."""
    image_size = int(model_cfg["image_size"])
    cm = model_cfg.get("channel_mult", "")
    if cm == "" or cm is None:
        cm = {512: (0.5, 1, 1, 2, 2, 4, 4), 256: (1, 1, 2, 2, 4, 4), 128: (1, 1, 2, 3, 4), 64: (1, 2, 3, 4)}[image_size]
    elif isinstance(cm, str):
        cm = tuple(int(c) for c in cm.split(","))
    att = model_cfg.get("attention_resolutions", "16,8")
    if isinstance(att, str):
        att = tuple(image_size // int(r) for r in att.split(","))
    nhu = model_cfg.get("num_heads_upsample", -1)
    nh = model_cfg.get("num_heads", 4)
    return dict(
        image_size=image_size,
        model_channels=int(model_cfg["num_channels"]),
        channel_mult=tuple(cm),
        num_res_blocks=int(model_cfg["num_res_blocks"]),
        attention_ds=tuple(att),
        num_heads=nh,
        num_heads_upsample=nh if nhu == -1 else nhu,
        num_head_channels=model_cfg.get("num_head_channels", -1),
        out_channels=6 if model_cfg.get("learn_sigma", False) else 3,
        resblock_updown=bool(model_cfg.get("resblock_updown", False)),
        use_scale_shift_norm=bool(model_cfg.get("use_scale_shift_norm", True)),
        use_new_attention_order=bool(model_cfg.get("use_new_attention_order", False)),
    )


def timestep_embedding(timesteps, dim, max_period=10000):
    # nn.py:111-129 : cos first, then sin; frequency divisor is `half`.
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32) / half)
    args = timesteps[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def _gn32(sd, p, x):
    # nn.py:25-27 + normalization() :101-108: GroupNorm(32, C), eps 1e-5, fp32.
    return F.group_norm(x.float(), 32, sd[p + ".weight"], sd[p + ".bias"], eps=1e-5)


def _conv(sd, p, x, padding):
    return F.conv2d(x, sd[p + ".weight"], sd[p + ".bias"], padding=padding)


def _resblock(sd, p, x, emb, cfg, up=False, down=False):
    # unet.py:244-264
    h = F.silu(_gn32(sd, p + ".in_layers.0", x))
    if up:
        h = F.interpolate(h, scale_factor=2, mode="nearest")
        x = F.interpolate(x, scale_factor=2, mode="nearest")
    elif down:
        h = F.avg_pool2d(h, 2, 2)
        x = F.avg_pool2d(x, 2, 2)
    h = _conv(sd, p + ".in_layers.2", h, 1)
    emb_out = F.linear(F.silu(emb), sd[p + ".emb_layers.1.weight"], sd[p + ".emb_layers.1.bias"])[:, :, None, None]
    if cfg["use_scale_shift_norm"]:
        scale, shift = torch.chunk(emb_out, 2, dim=1)
        h = _gn32(sd, p + ".out_layers.0", h) * (1 + scale) + shift
        h = F.silu(h)
    else:
        h = h + emb_out
        h = F.silu(_gn32(sd, p + ".out_layers.0", h))
    h = _conv(sd, p + ".out_layers.3", h, 1)
    if (p + ".skip_connection.weight") in sd:
        w = sd[p + ".skip_connection.weight"]
        x = F.conv2d(x, w, sd[p + ".skip_connection.bias"], padding=w.shape[-1] // 2)
    return x + h


def _attention(sd, p, x, n_heads, new_order):
    # unet.py:307-313 and :345-362 / :377-397
    b, c, hh, ww = x.shape
    xf = x.reshape(b, c, -1)
    qkv = F.conv1d(_gn32(sd, p + ".norm", xf), sd[p + ".qkv.weight"], sd[p + ".qkv.bias"])
    length = qkv.shape[-1]
    ch = qkv.shape[1] // (3 * n_heads)
    scale = 1 / math.sqrt(math.sqrt(ch))
    if new_order:
        q, k, v = qkv.chunk(3, dim=1)
        q = q.reshape(b * n_heads, ch, length)
        k = k.reshape(b * n_heads, ch, length)
        v = v.reshape(b * n_heads, ch, length)
    else:
        q, k, v = qkv.reshape(b * n_heads, ch * 3, length).split(ch, dim=1)
    w = torch.einsum("bct,bcs->bts", q * scale, k * scale)
    w = torch.softmax(w.float(), dim=-1)
    a = torch.einsum("bts,bcs->bct", w, v).reshape(b, -1, length)
    hproj = F.conv1d(a, sd[p + ".proj_out.weight"], sd[p + ".proj_out.bias"])
    return (xf + hproj).reshape(b, c, hh, ww)


def _heads(cfg, ch, upsample):
    if cfg["num_head_channels"] == -1:
        return cfg["num_heads_upsample"] if upsample else cfg["num_heads"]
    return ch // cfg["num_head_channels"]


def guided_unet_forward(sd, cfg, x, timesteps):
    """UNetModel.forward (unet.py:642-671) for the topology built at unet.py:484-620.
    `cfg` = parse_guided_config(...). Returns [N, out_channels, H, W]."""
    mc = cfg["model_channels"]
    nrb = cfg["num_res_blocks"]
    mults = cfg["channel_mult"]
    emb = timestep_embedding(timesteps, mc)
    emb = F.linear(emb, sd["time_embed.0.weight"], sd["time_embed.0.bias"])
    emb = F.linear(F.silu(emb), sd["time_embed.2.weight"], sd["time_embed.2.bias"])

    hs = []
    h = _conv(sd, "input_blocks.0.0", x.float(), 1)
    hs.append(h)
    idx, ds = 1, 1
    for level, mult in enumerate(mults):
        for _ in range(nrb):
            p = f"input_blocks.{idx}"
            h = _resblock(sd, p + ".0", h, emb, cfg)
            ch = int(mult * mc)
            if ds in cfg["attention_ds"]:
                h = _attention(sd, p + ".1", h, _heads(cfg, ch, False), cfg["use_new_attention_order"])
            hs.append(h)
            idx += 1
        if level != len(mults) - 1:
            p = f"input_blocks.{idx}.0"
            if cfg["resblock_updown"]:
                h = _resblock(sd, p, h, emb, cfg, down=True)
            elif (p + ".op.weight") in sd:  # Downsample with conv: 3x3 stride 2 (unet.py:140-143)
                h = F.conv2d(h, sd[p + ".op.weight"], sd[p + ".op.bias"], stride=2, padding=1)
            else:
                h = F.avg_pool2d(h, 2, 2)
            hs.append(h)
            idx += 1
            ds *= 2

    h = _resblock(sd, "middle_block.0", h, emb, cfg)
    ch = int(mults[-1] * mc)
    h = _attention(sd, "middle_block.1", h, _heads(cfg, ch, False), cfg["use_new_attention_order"])
    h = _resblock(sd, "middle_block.2", h, emb, cfg)

    oidx = 0
    for level, mult in list(enumerate(mults))[::-1]:
        for i in range(nrb + 1):
            p = f"output_blocks.{oidx}"
            h = torch.cat([h, hs.pop()], dim=1)
            h = _resblock(sd, p + ".0", h, emb, cfg)
            ch = int(mult * mc)
            sub = 1
            if ds in cfg["attention_ds"]:
                h = _attention(sd, f"{p}.{sub}", h, _heads(cfg, ch, True), cfg["use_new_attention_order"])
                sub += 1
            if level and i == nrb:
                q = f"{p}.{sub}"
                if cfg["resblock_updown"]:
                    h = _resblock(sd, q, h, emb, cfg, up=True)
                else:  # Upsample (unet.py:107-118)
                    h = F.interpolate(h, scale_factor=2, mode="nearest")
                    if (q + ".conv.weight") in sd:
                        h = F.conv2d(h, sd[q + ".conv.weight"], sd[q + ".conv.bias"], padding=1)
                ds //= 2
            oidx += 1
    assert not hs
    h = F.silu(_gn32(sd, "out.0", h))
    return _conv(sd, "out.2", h, 1)
