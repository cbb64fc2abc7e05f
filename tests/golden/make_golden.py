"""Generate the golden vectors in this directory FROM THE REFERENCE ITSELF.

Run in the build container only (needs /root/reference):   python tests/golden/make_golden.py
The reference is imported from a scratch copy (never from /root/reference directly: importing
score_sde.models there would JIT-write into the read-only tree) with stubs for the packages that
are not installed (torchsde, torchvision, torchdiffeq) and for score_sde.op (fir=False => unused).

What is pinned (all fp32, CPU, seeded weights from diffpure_amd.synth keyed by parameter name):
  ncsnpp_small.pt   NCSNpp (nf=32, 2 levels, attention at 8x8) forward, B=2
  guided_small.pt   UNetModel (128 ch, 2 levels, attention, up/down ResBlocks, FiLM) forward, B=2
  ncsnpp_full.pt    configs/cifar10.yml NCSNpp forward, B=2, full output
  guided_full.pt    configs/imagenet.yml UNetModel (use_fp16=False) forward, B=1, strided crop
  sde_fg.pt         RevVPSDE.f / .g and VPODE.forward on the two small networks
  ddpm_psample.pt   GaussianDiffusion.p_sample (learned-range variance) with injected noise
"""
import argparse
import os
import shutil
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

import torch  # noqa: E402
import yaml  # noqa: E402

from diffpure_amd.synth import synth_state_dict  # noqa: E402

REF = "/root/reference"
SCRATCH = "/tmp/refcopy"


def import_reference():
    if not os.path.isdir(SCRATCH):
        shutil.copytree(REF, SCRATCH)
    sys.path.insert(0, SCRATCH)
    for n in ("torchsde", "torchvision", "torchvision.utils", "torchdiffeq"):
        sys.modules[n] = types.ModuleType(n)
    sys.modules["torchvision"].utils = sys.modules["torchvision.utils"]
    sys.modules["torchdiffeq"].odeint_adjoint = None
    op = types.ModuleType("score_sde.op")
    op.upfirdn2d = None
    sys.modules["score_sde.op"] = op


def ref_module(name, relpath):
    """Import a reference file by path (the reference's `runners/` has no __init__.py, so a plain
    `import runners.x` would resolve to THIS repository's drop-in package of the same name)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location(name, os.path.join(SCRATCH, relpath))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def d2n(c):
    ns = argparse.Namespace()
    for k, v in c.items():
        setattr(ns, k, d2n(v) if isinstance(v, dict) else v)
    return ns


def small_ncsnpp_cfg():
    cfg = yaml.safe_load(open(os.path.join(REF, "configs/cifar10.yml")))
    cfg["model"].update(nf=32, ch_mult=[1, 2], num_res_blocks=2, attn_resolutions=[8])
    cfg["data"]["image_size"] = 16
    return cfg


def small_guided_cfg():
    return dict(image_size=32, num_channels=128, num_res_blocks=1, channel_mult="1,2", attention_resolutions="16",
                num_head_channels=64, num_heads=4, num_heads_upsample=-1, resblock_updown=True,
                use_scale_shift_norm=True, learn_sigma=True, use_new_attention_order=False, class_cond=False,
                use_fp16=False)


def load_synth(module, seed):
    sd = synth_state_dict({k: v.shape for k, v in module.state_dict().items()}, seed)
    missing = module.load_state_dict(sd, strict=False)
    assert set(missing.missing_keys) <= {"sigmas"}, missing
    return module.eval()


def main():
    import_reference()
    from guided_diffusion.script_util import create_model, create_gaussian_diffusion, model_and_diffusion_defaults
    VPODE = ref_module("ref_diffpure_ode", "runners/diffpure_ode.py").VPODE
    RevVPSDE = ref_module("ref_diffpure_sde", "runners/diffpure_sde.py").RevVPSDE
    from score_sde.models import utils as mutils

    torch.manual_seed(0)
    torch.set_grad_enabled(False)
    g = torch.Generator().manual_seed(1234)

    # ---- NCSN++ small ----
    ncfg = small_ncsnpp_cfg()
    nmod = load_synth(mutils.create_model(d2n(ncfg)), 1234)
    xn = torch.rand(2, 3, 16, 16, generator=g) * 2 - 1
    labels = torch.tensor([0.37 * 999, 0.05 * 999])
    torch.save(dict(cfg=ncfg, seed=1234, x=xn, labels=labels, out=nmod(xn, labels)), os.path.join(HERE, "ncsnpp_small.pt"))

    # ---- guided small ----
    gcfg = small_guided_cfg()
    kw = {k: gcfg[k] for k in ("image_size", "num_channels", "num_res_blocks", "channel_mult", "learn_sigma",
                               "class_cond", "attention_resolutions", "num_heads", "num_head_channels",
                               "num_heads_upsample", "use_scale_shift_norm", "resblock_updown", "use_fp16",
                               "use_new_attention_order")}
    gmod = load_synth(create_model(**kw), 1234)
    xg = torch.rand(2, 3, 32, 32, generator=g) * 2 - 1
    tg = torch.tensor([37, 950])
    torch.save(dict(cfg=gcfg, seed=1234, x=xg, t=tg, out=gmod(xg, tg)), os.path.join(HERE, "guided_small.pt"))

    # ---- RevVPSDE.f/.g, VPODE.forward on the small nets ----
    rec = {}
    for name, mod, x, shape in (("score_sde", nmod, xn, (3, 16, 16)), ("guided_diffusion", gmod, xg, (3, 32, 32))):
        rv = RevVPSDE(model=mod, score_type=name, img_shape=shape, model_kwargs=None)
        vo = VPODE(model=mod, score_type=name, img_shape=shape, model_kwargs=None)
        for tp in (0.9, 0.9635, 0.99999):
            t = torch.tensor(tp, dtype=torch.float32)
            rec[(name, "f", tp)] = rv.f(t, x.reshape(2, -1)).reshape(x.shape)
            rec[(name, "g", tp)] = rv.g(t, x.reshape(2, -1))[:, 0].clone()
        for s in (0.1, 0.0365, 1e-5):
            t = torch.tensor(s, dtype=torch.float32)
            rec[(name, "ode", s)] = vo(t, (x.reshape(2, -1),))[0].reshape(x.shape)
    torch.save(dict(rec=rec, xn=xn, xg=xg), os.path.join(HERE, "sde_fg.pt"))

    # ---- GaussianDiffusion.p_sample with injected noise ----
    import guided_diffusion.gaussian_diffusion as gd
    diffusion = create_gaussian_diffusion(steps=1000, learn_sigma=True, noise_schedule="linear", use_kl=False,
                                          predict_xstart=False, rescale_timesteps=True, rescale_learned_sigmas=False,
                                          timestep_respacing="1000")
    z = torch.randn(2, 3, 32, 32, generator=g)
    orig = gd.th.randn_like
    gd.th.randn_like = lambda t_: z
    try:
        outs = {}
        for i in (0, 1, 57, 99):
            outs[i] = diffusion.p_sample(gmod, xg, torch.tensor([i, i]), clip_denoised=True)["sample"]
    finally:
        gd.th.randn_like = orig
    torch.save(dict(z=z, outs=outs), os.path.join(HERE, "ddpm_psample.pt"))

    # ---- full-size NCSN++ ----
    fcfg = yaml.safe_load(open(os.path.join(REF, "configs/cifar10.yml")))
    fmod = load_synth(mutils.create_model(d2n(fcfg)), 1234)
    xf = torch.rand(2, 3, 32, 32, generator=g) * 2 - 1
    lf = torch.tensor([0.1 * 999, 0.02 * 999])
    torch.save(dict(cfg=fcfg, seed=1234, x=xf, labels=lf, out=fmod(xf, lf)), os.path.join(HERE, "ncsnpp_full.pt"))
    del fmod

    # ---- full-size guided UNet (fp32 torso) ----
    mc = model_and_diffusion_defaults()
    mc.update(yaml.safe_load(open(os.path.join(REF, "configs/imagenet.yml")))["model"])
    mc["use_fp16"] = False
    kw = {k: mc[k] for k in kw}
    big = load_synth(create_model(**kw), 1234)
    xb = torch.rand(1, 3, 256, 256, generator=torch.Generator().manual_seed(4321)) * 2 - 1  # regenerated by tests
    tb = torch.tensor([100])
    ob = big(xb, tb)
    torch.save(dict(cfg={k: mc[k] for k in kw}, seed=1234, x_seed=4321, t=tb, out_crop=ob[:, :, ::16, ::16].clone(),
                    out_absmean=ob.abs().mean().item(), out_std=ob.std().item()), os.path.join(HERE, "guided_full.pt"))
    print("golden vectors written to", HERE)
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".pt"):
            print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
