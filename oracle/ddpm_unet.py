"""Oracle (TEST INFRASTRUCTURE): functional CPU fp32 restatement of the CelebA-HQ DDPM UNet and of the
denoising loop that drives it.

Follows /root/reference/ddpm/unet_ddpm.py (get_timestep_embedding :13-33, Normalize :41-42, Upsample :45-63,
Downsample :66-87, ResnetBlock :90-149, AttnBlock :152-205, Model.__init__ :208-302, Model.forward :304-345)
and /root/reference/runners/diffpure_ddpm.py (get_beta_schedule :18-22, extract :25-33,
image_editing_denoising_step_flexible_mask :36-55, Diffusion.__init__ :80-98, image_editing_sample :100-142).
Consumes the reference's state_dict key names.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this package.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


def parse_ddpm_config(cfg):
    """cfg: the yaml dict of configs/celeba.yml (keys 'data', 'model', 'diffusion')."""
    m, d = cfg["model"], cfg["data"]
    assert m["type"] == "simple" and m["resamp_with_conv"]
    return dict(ch=int(m["ch"]), out_ch=int(m["out_ch"]), ch_mult=tuple(m["ch_mult"]), num_res_blocks=int(m["num_res_blocks"]),
                attn_resolutions=tuple(m["attn_resolutions"]), in_channels=int(m["in_channels"]),
                resolution=int(d["image_size"]), var_type=m["var_type"])


def get_timestep_embedding(timesteps, dim):
    # unet_ddpm.py:13-33 : sin first, then cos; divisor is (half - 1)
    half = dim // 2
    e = math.log(10000) / (half - 1)
    e = torch.exp(torch.arange(half, dtype=torch.float32) * -e)
    e = timesteps.float()[:, None] * e[None, :]
    return torch.cat([torch.sin(e), torch.cos(e)], dim=1)


def _gn(sd, p, x):
    return F.group_norm(x, 32, sd[p + ".weight"], sd[p + ".bias"], eps=1e-6)        # :41-42


def _conv(sd, p, x, stride=1, padding=1):
    return F.conv2d(x, sd[p + ".weight"], sd[p + ".bias"], stride=stride, padding=padding)


def _resblock(sd, p, x, temb):
    # :126-149
    h = _conv(sd, p + ".conv1", F.silu(_gn(sd, p + ".norm1", x)))
    h = h + F.linear(F.silu(temb), sd[p + ".temb_proj.weight"], sd[p + ".temb_proj.bias"])[:, :, None, None]
    h = _conv(sd, p + ".conv2", F.silu(_gn(sd, p + ".norm2", h)))
    if (p + ".nin_shortcut.weight") in sd:
        x = _conv(sd, p + ".nin_shortcut", x, padding=0)
    return x + h


def _attn(sd, p, x):
    # :181-205
    h = _gn(sd, p + ".norm", x)
    q, k, v = (_conv(sd, p + "." + n, h, padding=0) for n in ("q", "k", "v"))
    b, c, hh, ww = q.shape
    w = torch.bmm(q.reshape(b, c, hh * ww).permute(0, 2, 1), k.reshape(b, c, hh * ww)) * (int(c) ** (-0.5))
    w = F.softmax(w, dim=2)
    h = torch.bmm(v.reshape(b, c, hh * ww), w.permute(0, 2, 1)).reshape(b, c, hh, ww)
    return x + _conv(sd, p + ".proj_out", h, padding=0)


def unet_forward(sd, cfg, x, t):
    """Model.forward (:304-345). x: [B,C,H,W], t: [B] integer timesteps."""
    nres, nrb = len(cfg["ch_mult"]), cfg["num_res_blocks"]
    temb = get_timestep_embedding(t, cfg["ch"])
    temb = F.linear(temb, sd["temb.dense.0.weight"], sd["temb.dense.0.bias"])
    temb = F.linear(F.silu(temb), sd["temb.dense.1.weight"], sd["temb.dense.1.bias"])
    hs = [_conv(sd, "conv_in", x)]
    for lvl in range(nres):
        for ib in range(nrb):
            h = _resblock(sd, f"down.{lvl}.block.{ib}", hs[-1], temb)
            if f"down.{lvl}.attn.{ib}.norm.weight" in sd:
                h = _attn(sd, f"down.{lvl}.attn.{ib}", h)
            hs.append(h)
        if lvl != nres - 1:
            # Downsample (:78-87): pad right/bottom by one, 3x3 stride-2 convolution without padding
            hs.append(_conv(sd, f"down.{lvl}.downsample.conv", F.pad(hs[-1], (0, 1, 0, 1)), stride=2, padding=0))
    h = _resblock(sd, "mid.block_1", hs[-1], temb)
    h = _attn(sd, "mid.attn_1", h)
    h = _resblock(sd, "mid.block_2", h, temb)
    for lvl in reversed(range(nres)):
        for ib in range(nrb + 1):
            h = _resblock(sd, f"up.{lvl}.block.{ib}", torch.cat([h, hs.pop()], dim=1), temb)
            if f"up.{lvl}.attn.{ib}.norm.weight" in sd:
                h = _attn(sd, f"up.{lvl}.attn.{ib}", h)
        if lvl != 0:
            # Upsample (:57-63): nearest x2, then 3x3 convolution
            h = _conv(sd, f"up.{lvl}.upsample.conv", F.interpolate(h, scale_factor=2.0, mode="nearest"))
    return _conv(sd, "conv_out", F.silu(_gn(sd, "norm_out", h)))


class CelebaSchedule:
    """Diffusion.__init__ (diffpure_ddpm.py:80-98): float64 numpy schedule, coefficients taken as float32."""

    def __init__(self, beta_start=1e-4, beta_end=2e-2, steps=1000, var_type="fixedsmall"):
        betas = np.linspace(beta_start, beta_end, steps, dtype=np.float64)
        self.betas = torch.from_numpy(betas).float()
        alphas = 1.0 - betas
        ac = np.cumprod(alphas, axis=0)
        ac_prev = np.append(1.0, ac[:-1])
        post_var = betas * (1.0 - ac_prev) / (1.0 - ac)
        if var_type == "fixedlarge":
            self.logvar = np.log(np.append(post_var[1], betas[1:]))
        elif var_type == "fixedsmall":
            self.logvar = np.log(np.maximum(post_var, 1e-20))
        else:
            raise ValueError(var_type)


def denoising_step(unet_fn, sched, x, i, z):
    """image_editing_denoising_step_flexible_mask (:36-55) with the noise injected. i: python int (same for the batch)."""
    betas = sched.betas
    alphas = 1.0 - betas
    alphas_cumprod = alphas.cumprod(dim=0)
    t = torch.full((x.shape[0],), i, dtype=torch.long)
    eps = unet_fn(x, t)
    weighted_score = betas / torch.sqrt(1 - alphas_cumprod)
    mean = (1 / torch.sqrt(alphas))[i] * (x - weighted_score[i] * eps)
    logvar = torch.tensor(sched.logvar, dtype=torch.float)[i]
    mask = 0.0 if i == 0 else 1.0
    return (mean + mask * torch.exp(0.5 * logvar) * z).float()


def celeba_ddpm_purify(unet_fn, sched, x0, e, noises, t_int):
    """Diffusion.image_editing_sample (:116-131) for one sample_step. noises[k] belongs to step i = t_int-1-k."""
    a = (1 - sched.betas).cumprod(dim=0)
    x = x0 * a[t_int - 1].sqrt() + e * (1.0 - a[t_int - 1]).sqrt()
    for k, i in enumerate(reversed(range(t_int))):
        x = denoising_step(unet_fn, sched, x, i, noises[k])
    return x
