"""Checkpoint golden for the guided-diffusion runners: a `256x256_diffusion_uncond.pt`-shaped file WRITTEN BY THE
REFERENCE'S OWN MODULE (torch.save(model.state_dict())) and read back the way the reference's runners read it
(runners/diffpure_sde.py:163-170, diffpure_guided.py:24-33):
    model_config = model_and_diffusion_defaults(); model_config.update(vars(config.model))
    model, _ = create_model_and_diffusion(**model_config)
    model.load_state_dict(torch.load(f'{model_dir}/256x256_diffusion_uncond.pt', map_location='cpu'))
    if model_config['use_fp16']: model.convert_to_fp16()
on a SMALL configuration of the same family (resblock_updown + scale-shift norm + learn_sigma, 32x32, 128/128 channels - GroupNorm32 needs >= 4 channels per group on the engine -, four
32-channel heads at 16x16), with `use_fp16` both False and True.

Run in the build container only (needs /root/reference):   python tests/golden/make_golden_guided_ckpt.py
  tests/golden/ckpt/guided/256x256_diffusion_uncond.pt   the state_dict file (fp32 tensors, as the real one)
  tests/golden/guided_ckpt.pt                            cfg, x, t, out_fp32 (use_fp16=False), out_fp16 (use_fp16=True: the reference's
                                                         own fp16 torso, convert_to_fp16 of guided_diffusion/unet.py:626-632)
The engine must load the same file through diffpure_amd.factory.build_guided (`args` carry no synthetic_weights flag) and
reproduce out_fp32 (to the precision mode's tolerance); out_fp16 documents how far the reference's OWN fp16 arithmetic is from it."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import torch  # noqa: E402

import make_golden as mg  # noqa: E402


def main():
    mg.import_reference()
    from guided_diffusion.script_util import create_model_and_diffusion, model_and_diffusion_defaults
    model_cfg = dict(image_size=32, num_channels=128, num_res_blocks=1, channel_mult="1,1", attention_resolutions="16",
                     num_head_channels=32, resblock_updown=True, use_scale_shift_norm=True, learn_sigma=True, class_cond=False,
                     use_fp16=False, diffusion_steps=1000, noise_schedule="linear", timestep_respacing="1000", rescale_timesteps=True)
    torch.manual_seed(0)

    def build(use_fp16):
        mc = model_and_diffusion_defaults()
        mc.update(model_cfg)
        mc["use_fp16"] = use_fp16
        model, _ = create_model_and_diffusion(**mc)
        return model, mc

    # --- the "released checkpoint": the reference module's own state_dict with non-trivial weights
    trained, _ = build(False)
    mg.load_synth(trained, 4321)
    os.makedirs(os.path.join(HERE, "ckpt", "guided"), exist_ok=True)
    path = os.path.join(HERE, "ckpt", "guided", "256x256_diffusion_uncond.pt")
    torch.save(trained.state_dict(), path)

    g = torch.Generator().manual_seed(777)
    x = torch.rand(2, 3, 32, 32, generator=g) * 2 - 1
    t = torch.tensor([63, 912])
    outs = {}
    for use_fp16 in (False, True):          # --- the reference's own load path
        model, mc = build(use_fp16)
        model.load_state_dict(torch.load(path, map_location="cpu"))
        if mc["use_fp16"]:
            model.convert_to_fp16()
        model.eval()
        with torch.no_grad():
            outs[use_fp16] = model(x, t).float()
    d = (outs[True] - outs[False]).abs().max().item()
    torch.save(dict(cfg=model_cfg, x=x, t=t, out_fp32=outs[False], out_fp16=outs[True], fp16_vs_fp32_maxabs=d), os.path.join(HERE, "guided_ckpt.pt"))
    print("guided ckpt golden: out abs-mean", float(outs[False].abs().mean()), "reference fp16 torso vs fp32: max-abs", d,
          "file bytes", os.path.getsize(path))


if __name__ == "__main__":
    main()
