"""Seeded synthetic weights for the two score networks.

The pretrained checkpoints DiffPure downloads (256x256_diffusion_uncond.pt, checkpoint_8.pth) are
not available offline, and the default initialisation of BOTH reference networks gives an
identically-zero output (zero_module at guided_diffusion/unet.py:218,302,623; init_scale 0 at
configs/cifar10.yml:37), which would make every parity test and benchmark vacuous.  This module
fills every parameter with deterministic, non-trivial values keyed by (seed, parameter name), so
the reference modules, the CPU oracle and the HIP engine can be given byte-identical weights.
"""
import zlib

import torch


def synth_tensor(key, shape, seed):
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(key.encode()) ^ (int(seed) * 0x9E3779B1)) & 0x7FFFFFFF)
    shape = tuple(shape)
    if len(shape) >= 2:
        if key.endswith(".W"):          # score_sde NIN: [in, out]
            fan_in = shape[0]
        else:                           # conv OIHW / OIk, linear [out, in]
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
        return torch.randn(shape, generator=g, dtype=torch.float32) * (1.0 / fan_in) ** 0.5
    if key.endswith("weight"):          # normalisation gains
        return 1.0 + 0.1 * torch.randn(shape, generator=g, dtype=torch.float32)
    return 0.1 * torch.randn(shape, generator=g, dtype=torch.float32)  # biases


def synth_state_dict(shapes, seed=1234):
    """shapes: mapping key -> shape (e.g. `param_shapes(cfg)` or `{k: v.shape for k, v in
    module.state_dict().items()}`).  Non-parameter buffers named 'sigmas' are skipped."""
    sd = {}
    for k, shp in shapes.items():
        if k == "sigmas":
            continue
        sd[k] = synth_tensor(k, shp, seed)
    return sd
