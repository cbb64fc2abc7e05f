"""Golden vectors for the CelebA-HQ DDPM UNet FROM THE REFERENCE ITSELF.

Run in the build container only (needs /root/reference):   python tests/golden/make_golden_ddpm.py
Imports the reference's ddpm/unet_ddpm.py and runners/diffpure_ddpm.py from the scratch copy
(make_golden.import_reference) and stores, for seeded synthetic weights (diffpure_amd.synth, keyed by name):
  ddpm_unet_small.pt   Model(ch=128, ch_mult [1,2,2], 2 res blocks, attention at 8x8, 16x16 images) forward, B=2
  ddpm_unet_full.pt    configs/celeba.yml Model (ch=128, [1,1,2,2,4,4], attention at 16x16, 256x256) forward, B=1, strided crop
  ldsde_fg.pt          LDSDE.f / .g (runners/diffpure_ldsde.py) on the two small score networks of make_golden.py
  celeba_step.pt       image_editing_denoising_step_flexible_mask (fixedsmall variance) on the small model with the
                       noise injected (torch.randn_like patched), steps i = 7 and i = 0
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import torch  # noqa: E402
import yaml  # noqa: E402

import make_golden as mg  # noqa: E402
from diffpure_amd.synth import synth_state_dict  # noqa: E402


def small_cfg():
    cfg = yaml.safe_load(open(os.path.join(mg.REF, "configs/celeba.yml")))
    cfg["model"].update(ch=128, ch_mult=[1, 2, 2], num_res_blocks=2, attn_resolutions=[8])
    cfg["data"]["image_size"] = 16
    return cfg


def main():
    mg.import_reference()
    from ddpm.unet_ddpm import Model
    rd = mg.ref_module("ref_diffpure_ddpm", "runners/diffpure_ddpm.py")
    torch.manual_seed(0)
    out = {}
    for name, cfg, bsz in (("small", small_cfg(), 2), ("full", yaml.safe_load(open(os.path.join(mg.REF, "configs/celeba.yml"))), 1)):
        model = Model(mg.d2n(cfg)).eval()
        sd = synth_state_dict({k: v.shape for k, v in model.state_dict().items()}, 1234)
        model.load_state_dict(sd)
        res = cfg["data"]["image_size"]
        x = torch.rand(bsz, 3, res, res, generator=torch.Generator().manual_seed(4321)) * 2 - 1
        t = torch.tensor([3, 977][:bsz])
        with torch.no_grad():
            y = model(x, t)
        rec = dict(cfg=cfg, seed=1234, x_seed=4321, t=t, keys=list(sd.keys()), shapes=[tuple(v.shape) for v in sd.values()])
        if name == "small":
            rec.update(x=x, y=y)
            small_model, small_sd = model, sd
        else:
            rec.update(y_crop=y[:, :, ::8, ::8].contiguous(), y_abs_mean=y.abs().mean())
        torch.save(rec, os.path.join(HERE, f"ddpm_unet_{name}.pt"))
        print(name, tuple(y.shape), float(y.abs().mean()))
    # one denoising step of the runner, noise injected
    cfg = small_cfg()
    sched_betas = rd.get_beta_schedule(beta_start=cfg["diffusion"]["beta_start"], beta_end=cfg["diffusion"]["beta_end"],
                                       num_diffusion_timesteps=cfg["diffusion"]["num_diffusion_timesteps"])
    import numpy as np
    betas = torch.from_numpy(sched_betas).float()
    alphas = 1.0 - sched_betas
    ac = np.cumprod(alphas, axis=0)
    ac_prev = np.append(1.0, ac[:-1])
    logvar = np.log(np.maximum(sched_betas * (1.0 - ac_prev) / (1.0 - ac), 1e-20))       # 'fixedsmall' (:96-97)
    x = torch.rand(2, 3, 16, 16, generator=torch.Generator().manual_seed(77)) * 2 - 1
    z = torch.randn(2, 3, 16, 16, generator=torch.Generator().manual_seed(78))
    steps = {}
    orig = torch.randn_like
    torch.randn_like = lambda a: z
    try:
        for i in (7, 0):
            with torch.no_grad():
                steps[i] = rd.image_editing_denoising_step_flexible_mask(x, t=torch.tensor([i, i]), model=small_model, logvar=logvar,
                                                                        betas=betas)
    finally:
        torch.randn_like = orig
    torch.save(dict(x=x, z=z, out=steps), os.path.join(HERE, "celeba_step.pt"))
    print("step", {k: float(v.abs().mean()) for k, v in steps.items()})

    # ---- LDSDE.f / .g on the small NCSN++ and guided UNet (same models and inputs as ncsnpp_small.pt / guided_small.pt)
    from guided_diffusion.script_util import create_model
    from score_sde.models import utils as mutils
    ld = mg.ref_module("ref_diffpure_ldsde", "runners/diffpure_ldsde.py")
    torch.set_grad_enabled(False)
    gn, gg = torch.load(os.path.join(HERE, "ncsnpp_small.pt")), torch.load(os.path.join(HERE, "guided_small.pt"))
    nmod = mg.load_synth(mutils.create_model(mg.d2n(gn["cfg"])), 1234)
    kw = {k: gg["cfg"][k] for k in ("image_size", "num_channels", "num_res_blocks", "channel_mult", "learn_sigma", "class_cond",
                                    "attention_resolutions", "num_heads", "num_head_channels", "num_heads_upsample",
                                    "use_scale_shift_norm", "resblock_updown", "use_fp16", "use_new_attention_order")}
    gmod = mg.load_synth(create_model(**kw), 1234)
    rec, sigma2, lam, eta = {}, 0.001, 0.01, 5
    for name, mod, x0, shape in (("score_sde", nmod, gn["x"], (3, 16, 16)), ("guided_diffusion", gmod, gg["x"], (3, 32, 32))):
        xc = x0 + 0.05 * torch.randn(x0.shape, generator=torch.Generator().manual_seed(11))      # current state != anchor
        sde = ld.LDSDE(model=mod, x_init=x0.reshape(2, -1), score_type=name, img_shape=shape, sigma2=sigma2, lambda_ld=lam, eta=eta,
                       model_kwargs=None)
        t = torch.tensor(0.93, dtype=torch.float32)
        rec[(name, "x")] = xc
        rec[(name, "f")] = sde.f(t, xc.reshape(2, -1)).reshape(xc.shape)
        rec[(name, "g")] = sde.g(t, xc.reshape(2, -1))[:, 0].clone()
    torch.save(dict(rec=rec, sigma2=sigma2, lambda_ld=lam, eta=eta), os.path.join(HERE, "ldsde_fg.pt"))
    print("ldsde", {k: float(v.abs().mean()) for k, v in rec.items()})


if __name__ == "__main__":
    main()
