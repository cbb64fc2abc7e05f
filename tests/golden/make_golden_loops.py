"""Loop-level golden vectors FROM THE REFERENCE'S OWN MODULES, at the step counts north_star names.

Run in the build container only (needs /root/reference; ~25 min on 8 cores):
    python tests/golden/make_golden_loops.py [guided_loop] [ncsnpp_loop] [ode_adjoint] [guided_full] [guided_vjp]

What is driven and by what
  * the score network is the reference's own nn.Module (guided_diffusion.unet.UNetModel built from
    configs/imagenet.yml with use_fp16=False, score_sde NCSNpp built from configs/cifar10.yml);
  * drift / diffusion are the reference's own `RevVPSDE.f` / `.g` (runners/diffpure_sde.py:131-147) and
    `VPODE.forward` (runners/diffpure_ode.py:124-131), called with the flat [B, D] state exactly as
    torchsde / torchdiffeq call them;
  * the stepping is the fixed-step Euler(-Maruyama) update on the float32 clock (torchsde / torchdiffeq are
    not installable here: oracle/solvers.py header), the adjoint's vector-Jacobian products are torch.autograd
    through the reference module;
  * the noise is the engine's Philox stream restated in numpy (tests/refops.py::philox_normal), keyed by
    (seed, global sample index, step) over the NHWC state - so the GPU test runs the PRODUCT path (in-kernel
    noise, no injection) against these files.

Files
  guided_loop100.pt        256x256 guided UNet, B=2, t*=0.1, dt=1e-3, 100 EM steps: x0, purified x (full tensors),
                           plus the state after 10 and 25 steps
  ncsnpp_loop100.pt        CIFAR NCSN++, B=4, t*=0.1, dt=1e-3, 100 EM steps
  ncsnpp_ode_adjoint100.pt CIFAR NCSN++, B=2: 100 Euler steps of the probability-flow ODE + 100 steps of the
                           continuous adjoint for a seeded cotangent (BASELINE.json configs[4] at reduced batch)
  guided_full.pt           (rewritten) one forward of the full guided UNet, B=1: the WHOLE [1,6,256,256] output
  guided_full_vjp.pt       dL/dx of that forward for a seeded cotangent on the eps channels (torch.autograd)
Round 3 (each ~10 min on 8 cores):
  guided_ddpm_loop100.pt   256x256 guided UNet, B=2, t=100: the reference's OWN sampling loop of
                           runners/diffpure_guided.py:58-75 - `create_model_and_diffusion` (imagenet.yml) ->
                           SpacedDiffusion.p_sample through _WrappedModel (respace.py:124-136), learned-range variance
                           and x0 clamp of gaussian_diffusion.py:240-334, 403-447 - with `th.randn_like` returning the
                           engine's Philox draw of that step.  Nothing of this loop is restated: it is in-tree upstream.
  guided_loop150.pt        the SDE loop at t*=0.15, dt=1e-3 = 150 EM steps (what run_scripts/imagenet/*.sh run), B=1
  guided_loop100_seeds.pt  the 100-step SDE loop, B=1, for two more noise seeds (x0 = sample 0 of guided_loop100.pt)
Round 4:
  ncsnpp_sde_adjoint100.pt CIFAR NCSN++, B=2, t*=0.1, dt=1e-3: 100 EM steps through the reference's RevVPSDE.f / .g, then the
                           stochastic adjoint over the reversed SAME Brownian path (what torchsde.sdeint_adjoint gives the
                           `--diffusion_type sde` attacks, runners/diffpure_sde.py:236-238), every vector-Jacobian product by
                           torch.autograd THROUGH THE REFERENCE'S RevVPSDE.f (score network included), seeded cotangent
  guided_sde_adjoint10.pt  the same on the full 256x256 guided UNet, B=1, t=10 (10 EM steps of dt=1e-3 + 10 adjoint steps)
Round 5:
  guided_loop150_dt0.0015.pt  BASELINE.json configs[2] AS WRITTEN: t*=0.15 in 100 EM steps, i.e. dt=1.5e-3 (B=1; every rounding enters the
                           state 1.5x larger than on the dt=1e-3 grid of guided_loop150.pt)          [target: guided_loop150_dt]
Round 6:
  ncsnpp_loop20_dt0.005.pt BASELINE.json configs[0] AS WRITTEN: CIFAR NCSN++, B=4, t*=0.1 in 20 EM steps of dt=5e-3      [target: ncsnpp_loop20; ~1 min]
Round 5 (continued):
  guided_sde_adjoint100.pt the stochastic adjoint on the full guided UNet at the PRODUCT grid: t=100, 100 EM steps + 100 adjoint
                           steps, B=1 (what run_scripts/imagenet/run_in_rand_inf.sh differentiates, at t=100)   [target: guided_sde_adjoint100; ~35 min]
"""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402
import yaml  # noqa: E402

import make_golden as mg  # noqa: E402
import refops  # noqa: E402
from oracle import solvers as osol  # noqa: E402

SEED = 1234


def philox_nchw(shape_nchw, seed, sample0, step):
    b, c, h, w = shape_nchw
    return refops.philox_normal((b, h, w, c), seed, sample0, step).permute(0, 3, 1, 2).contiguous()


def build_guided():
    from guided_diffusion.script_util import create_model, model_and_diffusion_defaults
    mc = model_and_diffusion_defaults()
    mc.update(yaml.safe_load(open(os.path.join(mg.REF, "configs/imagenet.yml")))["model"])
    mc["use_fp16"] = False
    keys = ("image_size", "num_channels", "num_res_blocks", "channel_mult", "learn_sigma", "class_cond",
            "attention_resolutions", "num_heads", "num_head_channels", "num_heads_upsample", "use_scale_shift_norm",
            "resblock_updown", "use_fp16", "use_new_attention_order")
    kw = {k: mc[k] for k in keys}
    return mg.load_synth(create_model(**kw), SEED), kw


def build_ncsnpp():
    from score_sde.models import utils as mutils
    cfg = yaml.safe_load(open(os.path.join(mg.REF, "configs/cifar10.yml")))
    return mg.load_synth(mutils.create_model(mg.d2n(cfg)), SEED), cfg


def em_loop(rv, x0, t_int, dt, seed, snaps=()):
    """diffuse (runners/diffpure_sde.py:222-223) + fixed-step Euler-Maruyama on the reference's f / g."""
    b = x0.shape[0]
    e = philox_nchw(x0.shape, seed, 0, -1)
    x = osol.diffuse(x0, e, t_int)
    grid = osol.sde_time_grid(t_int, dt)
    keep = {}
    t0 = time.time()
    for k in range(len(grid) - 1):
        tk, tn = grid[k], grid[k + 1]
        h = tn - tk
        xf = x.reshape(b, -1)
        f = rv.f(tk, xf).reshape(x.shape)
        g = rv.g(tk, xf).reshape(x.shape)
        z = philox_nchw(x0.shape, seed, 0, k)
        x = x + f * h + g * (z * torch.sqrt(h))
        if k + 1 in snaps:
            keep[k + 1] = x.clone()
        if k % 10 == 0:
            print(f"  step {k}: {time.time() - t0:.0f} s", flush=True)
    return x, keep, len(grid) - 1


def guided_loop():
    RevVPSDE = mg.ref_module("ref_diffpure_sde", "runners/diffpure_sde.py").RevVPSDE
    mod, kw = build_guided()
    rv = RevVPSDE(model=mod, score_type="guided_diffusion", img_shape=(3, 256, 256), model_kwargs=None)
    x0 = torch.rand(2, 3, 256, 256, generator=torch.Generator().manual_seed(99)) * 2 - 1
    with torch.no_grad():
        x, keep, n = em_loop(rv, x0, 100, 1e-3, SEED, snaps=(1, 10, 25))
    torch.save(dict(cfg=kw, seed=SEED, noise_seed=SEED, t=100, dt=1e-3, steps=n, x0=x0, out=x, snaps=keep),
               os.path.join(HERE, "guided_loop100.pt"))
    print("guided_loop100", n, float(x.abs().mean()))


def ncsnpp_loop():
    RevVPSDE = mg.ref_module("ref_diffpure_sde", "runners/diffpure_sde.py").RevVPSDE
    mod, cfg = build_ncsnpp()
    rv = RevVPSDE(model=mod, score_type="score_sde", img_shape=(3, 32, 32), model_kwargs=None)
    x0 = torch.rand(4, 3, 32, 32, generator=torch.Generator().manual_seed(98)) * 2 - 1
    with torch.no_grad():
        x, keep, n = em_loop(rv, x0, 100, 1e-3, SEED, snaps=(25,))
    torch.save(dict(cfg=cfg, seed=SEED, noise_seed=SEED, t=100, dt=1e-3, steps=n, x0=x0, out=x, snaps=keep),
               os.path.join(HERE, "ncsnpp_loop100.pt"))
    print("ncsnpp_loop100", n, float(x.abs().mean()))


def ncsnpp_loop20():
    """BASELINE.json configs[0] as written: CIFAR NCSN++, B=4, t*=0.1 in 20 EM steps (dt=5e-3) - the reference's CPU-runnable case."""
    RevVPSDE = mg.ref_module("ref_diffpure_sde", "runners/diffpure_sde.py").RevVPSDE
    mod, cfg = build_ncsnpp()
    rv = RevVPSDE(model=mod, score_type="score_sde", img_shape=(3, 32, 32), model_kwargs=None)
    x0 = torch.rand(4, 3, 32, 32, generator=torch.Generator().manual_seed(97)) * 2 - 1
    with torch.no_grad():
        x, keep, n = em_loop(rv, x0, 100, 5e-3, SEED, snaps=(5,))
    torch.save(dict(cfg=cfg, seed=SEED, noise_seed=SEED, t=100, dt=5e-3, steps=n, x0=x0, out=x, snaps=keep),
               os.path.join(HERE, "ncsnpp_loop20_dt0.005.pt"))
    print("ncsnpp_loop20", n, float(x.abs().mean()))


def ode_adjoint():
    VPODE = mg.ref_module("ref_diffpure_ode", "runners/diffpure_ode.py").VPODE
    mod, cfg = build_ncsnpp()
    vo = VPODE(model=mod, score_type="score_sde", img_shape=(3, 32, 32), model_kwargs=None)
    b, t_int, step = 2, 100, 1e-3
    x0 = torch.rand(b, 3, 32, 32, generator=torch.Generator().manual_seed(97)) * 2 - 1
    cot = torch.randn(b, 3, 32, 32, generator=torch.Generator().manual_seed(96))
    e = philox_nchw(x0.shape, SEED, 0, -1)
    # forward: torchdiffeq fixed-grid Euler on tau = -s (decreasing span), runners/diffpure_ode.py:229-238
    ts = torch.linspace(t_int * 1.0 / 1000, 1e-5, 2)
    tau = osol.ode_grid(-ts, step)
    x = osol.diffuse(x0, e, t_int)
    with torch.no_grad():
        for k in range(len(tau) - 1):
            dtau = tau[k + 1] - tau[k]
            F = vo(-tau[k], (x.reshape(b, -1),))[0].reshape(x.shape)
            x = x + dtau * (-F)
    x_final = x.clone()
    # adjoint: augmented Euler over the flipped span, da/ds = -a^T dF/dy
    grid = osol.ode_grid(ts.flip(0), step)
    y, a = x_final.clone(), cot.clone()
    for k in range(len(grid) - 1):
        ds = grid[k + 1] - grid[k]
        with torch.enable_grad():
            yy = y.detach().requires_grad_(True)
            F = vo(grid[k], (yy.reshape(b, -1),))[0].reshape(y.shape)
            (vjp,) = torch.autograd.grad(F, yy, -a)
        y = y + ds * F.detach()
        a = a + ds * vjp
    grad = osol.ode_diffuse_grad(a, t_int)
    torch.save(dict(cfg=cfg, seed=SEED, noise_seed=SEED, t=t_int, step=step, x0=x0, cot=cot, x_final=x_final, grad=grad,
                    steps=len(tau) - 1), os.path.join(HERE, "ncsnpp_ode_adjoint100.pt"))
    print("ncsnpp_ode_adjoint100", len(tau) - 1, float(x_final.abs().mean()), float(grad.abs().mean()))


def sde_adjoint_loop(rv, x0, cot, t_int, dt, seed, snap_after=10):
    """forward: em_loop (reference f / g); backward: Euler on the forward clock walked in reverse along the regenerated
    Brownian path - y_k = y_{k+1} - f(t_{k+1}, y_{k+1}) h_k - g(t_{k+1}) dW_k, a_k = a_{k+1} + h_k (df/dy)^T a_{k+1} (g is
    state-independent: no noise term and no Ito correction in the adjoint) - with f, g and df/dy^T a all taken from the
    reference's RevVPSDE by torch.autograd.  -> x_final, dL/dx0 (incl. the forward-diffusion scaling)."""
    b = x0.shape[0]
    with torch.no_grad():
        x_final, _, n = em_loop(rv, x0, t_int, dt, seed)
    grid = osol.sde_time_grid(t_int, dt)
    y, a = x_final.clone(), cot.clone()
    t0 = time.time()
    snap = {}
    for k in reversed(range(len(grid) - 1)):
        if k == len(grid) - 2 - snap_after:              # state after `snap_after` adjoint steps (the CPU suite re-walks only those)
            snap = dict(k_stop=k + 1, y=y.clone(), a=a.clone())
        tk, tn = grid[k], grid[k + 1]
        h = tn - tk
        with torch.enable_grad():
            yy = y.detach().requires_grad_(True)
            f = rv.f(tn, yy.reshape(b, -1)).reshape(y.shape)
            (vjp,) = torch.autograd.grad(f, yy, a)
        with torch.no_grad():
            g = rv.g(tn, y.reshape(b, -1)).reshape(y.shape)
            z = philox_nchw(x0.shape, seed, 0, k)
            y = y - f.detach() * h - g * (z * torch.sqrt(h))
            a = a + h * vjp
        if k % 10 == 0:
            print(f"  adjoint step {k}: {time.time() - t0:.0f} s", flush=True)
    return x_final, osol.ode_diffuse_grad(a, t_int), y, n, snap


def ncsnpp_sde_adjoint():
    RevVPSDE = mg.ref_module("ref_diffpure_sde", "runners/diffpure_sde.py").RevVPSDE
    mod, cfg = build_ncsnpp()
    rv = RevVPSDE(model=mod, score_type="score_sde", img_shape=(3, 32, 32), model_kwargs=None)
    b, t_int, dt = 2, 100, 1e-3
    x0 = torch.rand(b, 3, 32, 32, generator=torch.Generator().manual_seed(93)) * 2 - 1
    cot = torch.randn(b, 3, 32, 32, generator=torch.Generator().manual_seed(92))
    x_final, grad, y_back, n, snap = sde_adjoint_loop(rv, x0, cot, t_int, dt, SEED)
    torch.save(dict(cfg=cfg, snap=snap, seed=SEED, noise_seed=SEED, t=t_int, dt=dt, steps=n, x0=x0, cot=cot, x_final=x_final, grad=grad,
                    y_back=y_back), os.path.join(HERE, "ncsnpp_sde_adjoint100.pt"))
    print("ncsnpp_sde_adjoint100", n, float(x_final.abs().mean()), float(grad.abs().mean()), float(grad.abs().max()))


def guided_sde_adjoint(t_int=10):
    RevVPSDE = mg.ref_module("ref_diffpure_sde", "runners/diffpure_sde.py").RevVPSDE
    mod, kw = build_guided()
    rv = RevVPSDE(model=mod, score_type="guided_diffusion", img_shape=(3, 256, 256), model_kwargs=None)
    b, dt = 1, 1e-3
    x0 = torch.rand(b, 3, 256, 256, generator=torch.Generator().manual_seed(91)) * 2 - 1
    cot = torch.randn(b, 3, 256, 256, generator=torch.Generator().manual_seed(90))
    x_final, grad, y_back, n, _ = sde_adjoint_loop(rv, x0, cot, t_int, dt, SEED)
    torch.save(dict(cfg=kw, seed=SEED, noise_seed=SEED, t=t_int, dt=dt, steps=n, x0=x0, cot=cot, x_final=x_final,
                    grad=grad), os.path.join(HERE, f"guided_sde_adjoint{t_int}.pt"))
    print(f"guided_sde_adjoint{t_int}", n, float(x_final.abs().mean()), float(grad.abs().mean()), float(grad.abs().max()))


def guided_ddpm_loop():
    """runners/diffpure_guided.py:58-75 with the reference's own diffusion object; only the two noise sources
    (torch.randn_like at :59, th.randn_like in p_sample, gaussian_diffusion.py:438) are replaced by the Philox draws."""
    import guided_diffusion.gaussian_diffusion as gd
    from guided_diffusion.script_util import create_model_and_diffusion, model_and_diffusion_defaults
    mc = model_and_diffusion_defaults()
    mc.update(yaml.safe_load(open(os.path.join(mg.REF, "configs/imagenet.yml")))["model"])
    mc["use_fp16"] = False
    model, diffusion = create_model_and_diffusion(**mc)
    model = mg.load_synth(model, SEED)
    betas = torch.from_numpy(diffusion.betas).float()                     # :39
    b, t_int = 2, 100
    x0 = torch.rand(b, 3, 256, 256, generator=torch.Generator().manual_seed(95)) * 2 - 1
    e = philox_nchw(x0.shape, SEED, 0, -1)
    a = (1 - betas).cumprod(dim=0)                                        # :61-62
    x = x0 * a[t_int - 1].sqrt() + e * (1.0 - a[t_int - 1]).sqrt()
    keep, step = {}, [0]
    orig = gd.th.randn_like
    gd.th.randn_like = lambda t_: philox_nchw(tuple(t_.shape), SEED, 0, step[0])
    t0 = time.time()
    try:
        with torch.no_grad():
            for k, i in enumerate(reversed(range(t_int))):                # :66-71
                step[0] = k
                t = torch.tensor([i] * b)
                x = diffusion.p_sample(model, x, t, clip_denoised=True, denoised_fn=None, cond_fn=None, model_kwargs=None)["sample"]
                if k + 1 in (1, 10, 50):
                    keep[k + 1] = x.clone()
                if k % 10 == 0:
                    print(f"  ddpm step {k} (i={i}): {time.time() - t0:.0f} s", flush=True)
    finally:
        gd.th.randn_like = orig
    kw = {k: mc[k] for k in ("image_size", "num_channels", "num_res_blocks", "channel_mult", "learn_sigma", "class_cond",
                             "attention_resolutions", "num_heads", "num_head_channels", "num_heads_upsample",
                             "use_scale_shift_norm", "resblock_updown", "use_fp16", "use_new_attention_order")}
    torch.save(dict(cfg=kw, seed=SEED, noise_seed=SEED, t=t_int, steps=t_int, x0=x0, out=x, snaps=keep),
               os.path.join(HERE, "guided_ddpm_loop100.pt"))
    print("guided_ddpm_loop100", float(x.abs().mean()))


def guided_loop150(dt=1e-3):
    RevVPSDE = mg.ref_module("ref_diffpure_sde", "runners/diffpure_sde.py").RevVPSDE
    mod, kw = build_guided()
    rv = RevVPSDE(model=mod, score_type="guided_diffusion", img_shape=(3, 256, 256), model_kwargs=None)
    x0 = torch.rand(1, 3, 256, 256, generator=torch.Generator().manual_seed(94)) * 2 - 1
    with torch.no_grad():
        x, keep, n = em_loop(rv, x0, 150, dt, SEED, snaps=(100,) if dt == 1e-3 else (50,))
    name = "guided_loop150" if dt == 1e-3 else f"guided_loop150_dt{dt:g}"
    torch.save(dict(cfg=kw, seed=SEED, noise_seed=SEED, t=150, dt=dt, steps=n, x0=x0, out=x, snaps=keep),
               os.path.join(HERE, name + ".pt"))
    print(name, n, float(x.abs().mean()))


def guided_loop_seeds():
    RevVPSDE = mg.ref_module("ref_diffpure_sde", "runners/diffpure_sde.py").RevVPSDE
    mod, kw = build_guided()
    rv = RevVPSDE(model=mod, score_type="guided_diffusion", img_shape=(3, 256, 256), model_kwargs=None)
    x0 = (torch.rand(2, 3, 256, 256, generator=torch.Generator().manual_seed(99)) * 2 - 1)[:1]    # sample 0 of guided_loop100.pt
    outs = {}
    for seed in (7, 20240926):
        with torch.no_grad():
            x, _, n = em_loop(rv, x0, 100, 1e-3, seed)
        outs[seed] = x
        print("guided_loop100 seed", seed, n, float(x.abs().mean()), flush=True)
    torch.save(dict(cfg=kw, seed=SEED, t=100, dt=1e-3, steps=n, x0=x0, outs=outs), os.path.join(HERE, "guided_loop100_seeds.pt"))


def guided_full(with_vjp):
    mod, kw = build_guided()
    xb = torch.rand(1, 3, 256, 256, generator=torch.Generator().manual_seed(4321)) * 2 - 1
    tb = torch.tensor([100])
    if with_vjp:
        cot = torch.randn(1, 3, 256, 256, generator=torch.Generator().manual_seed(4322))
        xr = xb.clone().requires_grad_(True)
        with torch.enable_grad():
            ob = mod(xr, tb)
            (dx,) = torch.autograd.grad(ob[:, :3], xr, cot)
        ob = ob.detach()
        torch.save(dict(cfg=kw, seed=SEED, x_seed=4321, cot_seed=4322, t=tb, dx=dx), os.path.join(HERE, "guided_full_vjp.pt"))
        print("guided_full_vjp", float(dx.abs().mean()), float(dx.abs().max()))
    else:
        with torch.no_grad():
            ob = mod(xb, tb)
    torch.save(dict(cfg=kw, seed=SEED, x_seed=4321, t=tb, out=ob.clone(), out_crop=ob[:, :, ::16, ::16].clone(),
                    out_absmean=ob.abs().mean().item(), out_std=ob.std().item()), os.path.join(HERE, "guided_full.pt"))
    print("guided_full", float(ob.abs().mean()))


def main():
    mg.import_reference()
    torch.manual_seed(0)
    what = sys.argv[1:] or ["ncsnpp_loop", "ode_adjoint", "guided_vjp", "guided_loop"]
    for w in what:
        t0 = time.time()
        if w == "guided_loop":
            guided_loop()
        elif w == "ncsnpp_loop":
            ncsnpp_loop()
        elif w == "ncsnpp_loop20":
            ncsnpp_loop20()
        elif w == "ode_adjoint":
            ode_adjoint()
        elif w == "guided_full":
            guided_full(False)
        elif w == "guided_vjp":
            guided_full(True)
        elif w == "guided_ddpm_loop":
            guided_ddpm_loop()
        elif w == "guided_loop150":
            guided_loop150()
        elif w == "guided_loop_seeds":
            guided_loop_seeds()
        elif w == "ncsnpp_sde_adjoint":
            ncsnpp_sde_adjoint()
        elif w == "guided_sde_adjoint":
            guided_sde_adjoint()
        elif w == "guided_sde_adjoint100":
            guided_sde_adjoint(100)
        elif w == "guided_loop150_dt":
            guided_loop150(1.5e-3)
        else:
            raise SystemExit(f"unknown target {w}")
        print(f"{w}: {time.time() - t0:.0f} s", flush=True)


if __name__ == "__main__":
    main()
