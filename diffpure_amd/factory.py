"""Build the score-network engines from the reference's own (args, config) objects.

Mirrors the model-construction half of the reference runners' constructors
(/root/reference/runners/diffpure_sde.py:151-195, diffpure_ode.py:135-181, diffpure_guided.py:18-39):
same config fields, same checkpoint locations, same EMA handling - but produces
`diffpure_amd.GuidedUNet` / `diffpure_amd.NCSNpp` engines instead of nn.Module trees.
"""
import os

import torch

from . import ddpm_unet, guided_unet, ncsnpp, synth


def _ns_to_dict(ns):
    if isinstance(ns, dict):
        return {k: _ns_to_dict(v) for k, v in ns.items()}
    if hasattr(ns, "__dict__"):
        return {k: _ns_to_dict(v) for k, v in vars(ns).items()}
    return ns


def guided_model_defaults():
    """model_and_diffusion_defaults() of guided_diffusion/script_util.py:51-73 (model part)."""
    return dict(image_size=64, num_channels=128, num_res_blocks=2, num_heads=4, num_heads_upsample=-1,
                num_head_channels=-1, attention_resolutions="16,8", channel_mult="", dropout=0.0, class_cond=False,
                use_checkpoint=False, use_scale_shift_norm=True, resblock_updown=False, use_fp16=False,
                use_new_attention_order=False, learn_sigma=False, diffusion_steps=1000, noise_schedule="linear",
                timestep_respacing="", use_kl=False, predict_xstart=False, rescale_timesteps=False,
                rescale_learned_sigmas=False)


DEFAULT_PRECISION = "f16sr"


def precision_of(args):
    """args.precision / DIFFPURE_PRECISION - the arithmetic of the 3x3 / 1x1 convolutions that follow a GroupNorm (fp32
    accumulation, fp32 GroupNorm, residual stream and solver state in every mode):
      "f16sr" (default) fp16 activations x fp16 weights, one MFMA pass; the fp16 weight panels are re-rounded stochastically
                        from the fp32 masters before every UNet call (purified pixels 2.2e-4 from the reference over 100 steps)
      "f16x2"           fp16 activations x 22-bit split weights, two passes (1.3e-4)
      "f16x3"           22-bit split operands, three passes (4e-6: fp32-class)
      "f16"             fp16 x fp16 with round-to-nearest weights = the reference's use_fp16 torso (1.0e-3)
      "f32"             fp32-input MFMA everywhere (1e-6)"""
    return getattr(args, "precision", None) or os.environ.get("DIFFPURE_PRECISION", DEFAULT_PRECISION)


def want_synthetic(args):
    return bool(getattr(args, "synthetic_weights", False)) or os.environ.get("DIFFPURE_SYNTH_WEIGHTS", "0") == "1"


def ncsnpp_state_from_checkpoint(ckpt, cfg):
    """checkpoint_8.pth holds {'optimizer','model','ema','step'}; the reference loads 'model' and
    then OVERWRITES every parameter, in parameters() order, with ema['shadow_params']
    (diffpure_sde.py:42-47,:182; score_sde/models/ema.py:61-72).  Effective weights = EMA list."""
    keys = [k for k in ncsnpp.param_shapes(cfg) if k != "sigmas"]
    shadow = ckpt["ema"]["shadow_params"]
    if len(shadow) != len(keys):
        raise ValueError(f"EMA list has {len(shadow)} tensors, the NCSN++ engine expects {len(keys)}")
    return {k: v for k, v in zip(keys, shadow)}


def build_guided(args, config, device, model_dir="pretrained/guided_diffusion"):
    """-> (GuidedUNet, model_config dict).  config.model overrides model_and_diffusion_defaults()
    exactly as diffpure_sde.py:163-165 does."""
    mc = guided_model_defaults()
    mc.update(_ns_to_dict(config.model))
    if mc.get("class_cond"):
        raise NotImplementedError("class-conditional guided diffusion is outside the purification path")
    cfg = guided_unet.parse_config(mc)
    net = guided_unet.GuidedUNet(cfg, device, precision_of(args))
    path = f"{model_dir}/256x256_diffusion_uncond.pt"
    if os.path.exists(path):
        sd = torch.load(path, map_location="cpu")
    elif want_synthetic(args):
        sd = synth.synth_state_dict(guided_unet.param_shapes(cfg), getattr(args, "seed", 1234) or 1234)
    else:
        raise FileNotFoundError(f"{path} not found (set args.synthetic_weights=True or DIFFPURE_SYNTH_WEIGHTS=1 "
                                "to run on seeded synthetic weights)")
    net.load_state_dict(sd)
    # The reference converts the torso to fp16 when use_fp16 (diffpure_sde.py:169-170); this
    # engine's fp32-MFMA path is at least as precise, so the flag only selects nothing here.
    return net, mc


def build_ncsnpp(args, config, device, model_dir="pretrained/score_sde"):
    cfg = ncsnpp.parse_config(_ns_to_dict(config))
    net = ncsnpp.NCSNpp(cfg, device, precision_of(args))
    path = f"{model_dir}/checkpoint_8.pth"
    if os.path.exists(path):
        sd = ncsnpp_state_from_checkpoint(torch.load(path, map_location="cpu"), cfg)
    elif want_synthetic(args):
        sd = synth.synth_state_dict(ncsnpp.param_shapes(cfg), getattr(args, "seed", 1234) or 1234)
    else:
        raise FileNotFoundError(f"{path} not found (set args.synthetic_weights=True or DIFFPURE_SYNTH_WEIGHTS=1 "
                                "to run on seeded synthetic weights)")
    net.load_state_dict(sd)
    return net, cfg


def build_celeba(args, config, device, model_dir="pretrained"):
    """CelebA-HQ DDPM UNet of runners/diffpure_ddpm.py:60-75.  The reference downloads `celeba_hq.ckpt` with
    torch.hub; here it is looked up locally (`<model_dir>/celeba_hq.ckpt`, then the torch.hub checkpoint cache)."""
    cfg = ddpm_unet.parse_config(_ns_to_dict(config))
    net = ddpm_unet.DdpmUNet(cfg, device, precision_of(args))
    cands = [f"{model_dir}/celeba_hq.ckpt", os.path.join(torch.hub.get_dir(), "checkpoints", "celeba_hq.ckpt")]
    path = next((c for c in cands if os.path.exists(c)), None)
    if path is not None:
        sd = torch.load(path, map_location="cpu")
    elif want_synthetic(args):
        sd = synth.synth_state_dict(ddpm_unet.param_shapes(cfg), getattr(args, "seed", 1234) or 1234)
    else:
        raise FileNotFoundError(f"celeba_hq.ckpt not found in {cands} (there is no network here; set args.synthetic_weights=True "
                                "or DIFFPURE_SYNTH_WEIGHTS=1 to run on seeded synthetic weights)")
    net.load_state_dict(sd)
    return net, cfg


def build_for_dataset(args, config, device):
    """Dispatch on config.data.dataset as the reference does (diffpure_sde.py:160-185).
    -> (net, kind, img_shape)"""
    ds = config.data.dataset
    if ds == "ImageNet":
        net, _ = build_guided(args, config, device)
        return net, "guided", (3, 256, 256)
    if ds == "CIFAR10":
        net, _ = build_ncsnpp(args, config, device)
        return net, "ncsnpp", (3, 32, 32)
    raise NotImplementedError(f"Unknown dataset {ds}!")


SCORE_TYPE_TO_KIND = {"guided_diffusion": "guided", "score_sde": "ncsnpp"}
