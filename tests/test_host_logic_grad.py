"""CPU checks of the input-gradient (VJP / adjoint-ODE) HOST logic with the HIP operators replaced
by their torch statements: tape bookkeeping, dgrad weight panels, skip-gradient routing, adjoint
stepping.  Oracle: torch.autograd through the CPU oracle network, and oracle.solvers.ode_adjoint_grad."""
import argparse

import pytest
import torch

import refops
from conftest import load_golden
from diffpure_amd import guided_unet as pg
from diffpure_amd import ncsnpp as pn
from diffpure_amd import sde as psde
from diffpure_amd.synth import synth_state_dict
from oracle import guided_unet as og
from oracle import ncsnpp as on
from oracle import solvers as osol


@pytest.fixture(autouse=True)
def _cpu_ops(monkeypatch):
    refops.patch_ops(monkeypatch)
    yield


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_ncsnpp_vjp_matches_autograd(precision):
    g = load_golden("ncsnpp_small.pt")
    cfg = pn.parse_config(g["cfg"])
    sd = synth_state_dict(pn.param_shapes(cfg), g["seed"])
    net = pn.NCSNpp(cfg, "cpu", precision).load_state_dict(sd)
    x, lab = g["x"], g["labels"]
    u = torch.randn(x.shape, generator=torch.Generator().manual_seed(1))
    xr = x.clone().requires_grad_(True)
    out = on.ncsnpp_forward(sd, on.parse_ncsnpp_config(g["cfg"]), xr, lab)
    (ref,) = torch.autograd.grad(out, xr, u)
    tape = []
    net.forward(nhwc(x), lab, tape=tape)
    got = nchw(net.vjp(tape, nhwc(u)))
    torch.testing.assert_close(got, ref, rtol=2e-3, atol=2e-4 * ref.abs().max().item())


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_guided_vjp_matches_autograd(precision):
    g = load_golden("guided_small.pt")
    cfg = pg.parse_config(g["cfg"])
    sd = synth_state_dict(pg.param_shapes(cfg), g["seed"])
    net = pg.GuidedUNet(cfg, "cpu", precision).load_state_dict(sd)
    x, t = g["x"], g["t"]
    u = torch.randn(x.shape, generator=torch.Generator().manual_seed(2))      # cotangent on the eps half only
    xr = x.clone().requires_grad_(True)
    out = og.guided_unet_forward(sd, og.parse_guided_config(g["cfg"]), xr, t)
    (ref,) = torch.autograd.grad(out[:, :3], xr, u)
    tape = []
    net.forward(nhwc(x), t.float(), tape=tape)
    got = nchw(net.vjp(tape, nhwc(u)))
    torch.testing.assert_close(got, ref, rtol=2e-3, atol=2e-4 * ref.abs().max().item())


@pytest.mark.parametrize("grad16", ["1", "0"])
@pytest.mark.parametrize("precision", ["f16", "f16sr"])
@pytest.mark.parametrize("kind", ["ncsnpp", "guided"])
def test_fp16_modes_gradient_wiring(kind, precision, grad16, monkeypatch):
    """fp16 x fp16 modes: the dgrad convolutions take plain-fp16 gradient operands and fp16 dgrad panels that are views into a
    SECOND weight pool, re-rounded together with the forward's panels ("Gradients"); DIFFPURE_GRAD16=0 keeps
    the three-pass split-fp16 panels.  Either way the VJP agrees with torch.autograd through the oracle to fp16 accuracy."""
    monkeypatch.setenv("DIFFPURE_GRAD16", grad16)
    if kind == "ncsnpp":
        g = load_golden("ncsnpp_small.pt")
        cfg = pn.parse_config(g["cfg"])
        sd = synth_state_dict(pn.param_shapes(cfg), g["seed"])
        net = pn.NCSNpp(cfg, "cpu", precision).load_state_dict(sd)
        x, lab = g["x"], g["labels"]
        xr = x.clone().requires_grad_(True)
        out = on.ncsnpp_forward(sd, on.parse_ncsnpp_config(g["cfg"]), xr, lab)
        args = (nhwc(x), lab)
    else:
        g = load_golden("guided_small.pt")
        cfg = pg.parse_config(g["cfg"])
        sd = synth_state_dict(pg.param_shapes(cfg), g["seed"])
        net = pg.GuidedUNet(cfg, "cpu", precision).load_state_dict(sd)
        x, t = g["x"], g["t"]
        xr = x.clone().requires_grad_(True)
        out = og.guided_unet_forward(sd, og.parse_guided_config(g["cfg"]), xr, t)[:, :3]
        args = (nhwc(x), t.float())
    u = torch.randn(x.shape, generator=torch.Generator().manual_seed(3))
    (ref,) = torch.autograd.grad(out, xr, u)
    tape = []
    net.reround(7)
    net.forward(*args, tape=tape)
    got = nchw(net.vjp(tape, nhwc(u)))
    err = ((got - ref).abs().max() / ref.abs().max()).item()
    assert err < 2e-2, err
    # the gradient panels are built lazily by the first vjp, after that call's reround: they must carry the call's key all the same
    tape = []
    net.reround(7)
    net.forward(*args, tape=tape)
    assert torch.equal(nchw(net.vjp(tape, nhwc(u))), got)
    panels = {k: v for k, v in net.p.items() if ".dw" in k}
    assert panels
    if grad16 == "1":
        pool = net._gpool
        assert pool is not None and pool.stochastic == (precision == "f16sr")
        base = pool.work.untyped_storage().data_ptr()
        pooled = [k for k, v in panels.items() if v.dtype == torch.float16 and v.untyped_storage().data_ptr() == base]
        assert len(pooled) >= len(panels) - 1, (len(pooled), len(panels))         # everything but the 3-channel head's dgrad
        net.reround(7)                 # (the pool was created inside vjp, after the call's reround: finalize() rounds with key 0)
        before = pool.work.clone()
        net.reround(8)
        if precision == "f16sr":       # re-rounded together with the forward's panels, keyed by the call
            assert 0.2 < (pool.work != before).float().mean().item() < 0.8
            net.reround(7)
        assert torch.equal(pool.work, before)
    else:
        assert net._gpool is None
        assert all(v.dtype == torch.float16 and v.shape[1] % 2 == 0 or v.dtype == torch.float32 for v in panels.values())


def test_ode_adjoint_matches_oracle():
    g = load_golden("ncsnpp_small.pt")
    cfg = pn.parse_config(g["cfg"])
    sd = synth_state_dict(pn.param_shapes(cfg), g["seed"])
    net = pn.NCSNpp(cfg, "cpu").load_state_dict(sd)
    score = osol.make_score_fn("ncsnpp", sd, on.parse_ncsnpp_config(g["cfg"]))
    gen = torch.Generator().manual_seed(9)
    x0 = g["x"]
    e = torch.randn(x0.shape, generator=gen)
    cot = torch.randn(x0.shape, generator=gen)
    step = 2e-2
    with torch.no_grad():
        xf = osol.ode_purify(score, x0, e, 100, step)
    ref = osol.ode_diffuse_grad(osol.ode_adjoint_grad(score, xf, cot, 100, step), 100)
    pur = psde.Purifier(net, "ncsnpp", "cpu")
    got = pur.ode_vjp(xf, cot, 100, step) * pur.diffuse_scale(100)
    torch.testing.assert_close(got, ref, rtol=2e-3, atol=2e-4 * ref.abs().max().item())


def test_ode_runner_is_differentiable():
    """The drop-in OdeGuidedDiffusion returns a tensor autograd can differentiate w.r.t. the input
    (what an adaptive attack does upstream through torchdiffeq.odeint_adjoint)."""
    from runners.diffpure_ode import OdeGuidedDiffusion
    g = load_golden("ncsnpp_small.pt")

    def ns(d):
        n = argparse.Namespace()
        for k, v in d.items():
            setattr(n, k, ns(v) if isinstance(v, dict) else v)
        return n

    config = ns(g["cfg"])
    config.device = torch.device("cpu")
    args = argparse.Namespace(t=100, rand_t=False, t_delta=15, use_bm=False, sample_step=1, log_dir=None, score_type="score_sde",
                              seed=1234, synthetic_weights=True, step_size=2e-2, precision="f16x3")
    runner = OdeGuidedDiffusion(args, config, device="cpu")
    x0 = g["x"].clone().requires_grad_(True)
    e = torch.randn(x0.shape, generator=torch.Generator().manual_seed(9))
    cot = torch.randn(x0.shape, generator=torch.Generator().manual_seed(10))
    out = runner.image_editing_sample(x0, bs_id=7, noise=dict(e=e, z=[]))
    (out * cot).sum().backward()
    sd = synth_state_dict(pn.param_shapes(pn.parse_config(g["cfg"])), 1234)
    score = osol.make_score_fn("ncsnpp", sd, on.parse_ncsnpp_config(g["cfg"]))
    ref = osol.ode_diffuse_grad(osol.ode_adjoint_grad(score, out.detach(), cot, 100, 2e-2), 100)
    torch.testing.assert_close(x0.grad, ref, rtol=2e-3, atol=2e-4 * ref.abs().max().item())


def test_oracle_adjoint_converges_to_unrolled_autograd():
    """Pins the restated continuous adjoint (oracle/solvers.py) to ground truth: as the step shrinks it
    converges to torch.autograd through the unrolled Euler loop (measured cosine 0.80 / 0.977 / 0.996 at
    step 2e-2 / 5e-3 / 2e-3 on this network) - optimise-then-discretise vs discretise-then-optimise."""
    g = load_golden("ncsnpp_small.pt")
    sd = synth_state_dict(pn.param_shapes(pn.parse_config(g["cfg"])), 1234)
    score = osol.make_score_fn("ncsnpp", sd, on.parse_ncsnpp_config(g["cfg"]))
    e = torch.randn(g["x"].shape, generator=torch.Generator().manual_seed(9))
    cot = torch.randn(g["x"].shape, generator=torch.Generator().manual_seed(10))
    cos = []
    for step in (2e-2, 5e-3):
        xr = g["x"].clone().requires_grad_(True)
        with torch.enable_grad():
            o2 = osol.ode_purify(score, xr, e, 100, step)
            (g2,) = torch.autograd.grad(o2, xr, cot)
        ref = osol.ode_diffuse_grad(osol.ode_adjoint_grad(score, o2.detach(), cot, 100, step), 100)
        cos.append(torch.nn.functional.cosine_similarity(ref.flatten(), g2.flatten(), dim=0).item())
    assert cos[1] > 0.95 and cos[1] > cos[0], cos


@pytest.mark.parametrize("kind", ["ncsnpp", "guided"])
def test_sde_stochastic_adjoint_matches_oracle(kind):
    if kind == "ncsnpp":
        g = load_golden("ncsnpp_small.pt")
        cfg = pn.parse_config(g["cfg"])
        sd = synth_state_dict(pn.param_shapes(cfg), g["seed"])
        net = pn.NCSNpp(cfg, "cpu").load_state_dict(sd)
        score = osol.make_score_fn("ncsnpp", sd, on.parse_ncsnpp_config(g["cfg"]))
    else:
        g = load_golden("guided_small.pt")
        cfg = pg.parse_config(g["cfg"])
        sd = synth_state_dict(pg.param_shapes(cfg), g["seed"])
        net = pg.GuidedUNet(cfg, "cpu").load_state_dict(sd)
        score = osol.make_score_fn("guided", sd, og.parse_guided_config(g["cfg"]))
    gen = torch.Generator().manual_seed(4)
    x0 = g["x"]
    dt = 2e-2
    n = len(osol.sde_time_grid(100, dt)) - 1
    e = torch.randn(x0.shape, generator=gen)
    zs = [torch.randn(x0.shape, generator=gen) for _ in range(n)]
    cot = torch.randn(x0.shape, generator=gen)
    with torch.no_grad():
        xf = osol.sde_purify(score, x0, e, zs, 100, dt)
    ref = osol.ode_diffuse_grad(osol.sde_adjoint_grad(score, xf, cot, zs, 100, dt), 100)
    pur = psde.Purifier(net, kind, "cpu")
    got = pur.sde_vjp(xf, cot, 100, dt, noise=dict(e=e, z=zs)) * pur.diffuse_scale(100)
    torch.testing.assert_close(got, ref, rtol=2e-3, atol=2e-4 * ref.abs().max().item())


def test_oracle_sde_adjoint_converges_to_unrolled_autograd():
    """The restated stochastic adjoint approaches torch.autograd through the unrolled Euler-Maruyama loop
    (same injected noise) as dt shrinks."""
    g = load_golden("ncsnpp_small.pt")
    sd = synth_state_dict(pn.param_shapes(pn.parse_config(g["cfg"])), 1234)
    score = osol.make_score_fn("ncsnpp", sd, on.parse_ncsnpp_config(g["cfg"]))
    gen = torch.Generator().manual_seed(9)
    e = torch.randn(g["x"].shape, generator=gen)
    cot = torch.randn(g["x"].shape, generator=gen)
    cos = []
    for dt in (2e-2, 5e-3):
        n = len(osol.sde_time_grid(100, dt)) - 1
        zs = [torch.randn(g["x"].shape, generator=torch.Generator().manual_seed(100 + i)) for i in range(n)]
        xr = g["x"].clone().requires_grad_(True)
        with torch.enable_grad():
            o2 = osol.sde_purify(score, xr, e, zs, 100, dt)
            (g2,) = torch.autograd.grad(o2, xr, cot)
        ref = osol.ode_diffuse_grad(osol.sde_adjoint_grad(score, o2.detach(), cot, zs, 100, dt), 100)
        cos.append(torch.nn.functional.cosine_similarity(ref.flatten(), g2.flatten(), dim=0).item())
    assert cos[1] > 0.93 and cos[1] > cos[0], cos   # measured 0.69 / 0.947 / 0.988 / 0.997 at dt 2e-2 / 5e-3 / 2e-3 / 1e-3


def test_sde_runner_is_differentiable():
    from runners.diffpure_sde import RevGuidedDiffusion
    g = load_golden("ncsnpp_small.pt")

    def ns(d):
        n = argparse.Namespace()
        for k, v in d.items():
            setattr(n, k, ns(v) if isinstance(v, dict) else v)
        return n

    config = ns(g["cfg"])
    config.device = torch.device("cpu")
    args = argparse.Namespace(t=100, rand_t=False, t_delta=15, use_bm=False, sample_step=1, log_dir=None, score_type="score_sde",
                              seed=1234, synthetic_weights=True, dt=2e-2, precision="f32")
    runner = RevGuidedDiffusion(args, config, device="cpu")
    gen = torch.Generator().manual_seed(4)
    x0 = g["x"].clone().requires_grad_(True)
    e = torch.randn(x0.shape, generator=gen)
    zs = [torch.randn(x0.shape, generator=gen) for _ in range(5)]
    cot = torch.randn(x0.shape, generator=gen)
    out = runner.image_editing_sample(x0, bs_id=7, noise=dict(e=e, z=zs))
    (out * cot).sum().backward()
    sd = synth_state_dict(pn.param_shapes(pn.parse_config(g["cfg"])), 1234)
    score = osol.make_score_fn("ncsnpp", sd, on.parse_ncsnpp_config(g["cfg"]))
    ref = osol.ode_diffuse_grad(osol.sde_adjoint_grad(score, out.detach(), cot, zs, 100, 2e-2), 100)
    torch.testing.assert_close(x0.grad, ref, rtol=2e-3, atol=2e-4 * ref.abs().max().item())


def test_adv_model_host_logic_vs_oracle(monkeypatch):
    """SURVEY.md section 8f-2 on the CPU (ops routed through their torch statements): the NHWC-in / NHWC-out
    path of the runners and the fused resize/affine steps of diffpure_amd.adv_model.SDE_Adv_Model reproduce
    oracle/adv.py (= eval_sde_adv.py:73-89) around the same runner, forward and dL/dx."""
    import refops
    from diffpure_amd import adv_model
    from oracle import adv as oadv
    refops.patch_ops(monkeypatch)
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    g = load_golden("ncsnpp_small.pt")

    def ns(d):
        n = argparse.Namespace()
        for k, v in d.items():
            setattr(n, k, ns(v) if isinstance(v, dict) else v)
        return n

    config = ns(g["cfg"])
    config.device = torch.device("cpu")
    args = argparse.Namespace(t=100, rand_t=False, t_delta=15, use_bm=False, sample_step=1, log_dir=None, score_type="score_sde",
                              seed=1234, synthetic_weights=True, step_size=5e-2, dt=5e-2, diffusion_type="ode", domain="cifar10",
                              classifier_name="none", diffusion_size=(16, 16), precision="f32")
    w = torch.randn(5, 3, generator=torch.Generator().manual_seed(3))

    class Clf(torch.nn.Module):
        def forward(self, x):
            return x.mean(dim=(2, 3)) @ w.t()

    model = adv_model.SDE_Adv_Model(args, config, classifier=Clf())
    runner = model.runner
    x = torch.rand(2, 3, 12, 12, generator=torch.Generator().manual_seed(4))
    cot = torch.randn(2, 5, generator=torch.Generator().manual_seed(5))
    x1 = x.clone().requires_grad_(True)
    model.counter.fill_(7)
    runner._calls = 0
    out = model(x1)
    (g1,) = torch.autograd.grad((out * cot).sum(), x1)
    x2 = x.clone().requires_grad_(True)
    runner._calls = 0
    ref = oadv.sde_adv_forward(lambda im: runner.image_editing_sample(im, bs_id=7), Clf(), x2, diffusion_size=(16, 16))
    (g2,) = torch.autograd.grad((ref * cot).sum(), x2)
    torch.testing.assert_close(out, ref, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(g1, g2, rtol=1e-4, atol=1e-6 * g2.abs().max().item())
    # the BPDA+EOT driver's dialect (eval_sde_adv_bpda.py:83-118, bpda_eot_attack.py:98-101): replicas in one batch
    with torch.no_grad():
        runner._calls = 0
        model.counter.fill_(7)
        pur = model(x.repeat(3, 1, 1, 1), mode="purify")
        assert pur.shape == (6, 3, 12, 12) and int(model.counter.item()) == 8
        torch.testing.assert_close(model(pur, mode="classify"), model.resnet(pur))
        runner._calls = 0
        torch.testing.assert_close(model(x.repeat(3, 1, 1, 1), mode="purify_and_classify"), model.resnet(pur))
    with pytest.raises(NotImplementedError):
        model(x, mode="nonsense")


def test_ldsde_stochastic_adjoint_matches_oracle():
    """Purifier.ldsde_vjp against the oracle's restated adjoint (gradient through the initial state), and the runner's
    autograd wrapper returns it."""
    g = load_golden("ncsnpp_small.pt")
    cfg = pn.parse_config(g["cfg"])
    sd = synth_state_dict(pn.param_shapes(cfg), g["seed"])
    net = pn.NCSNpp(cfg, "cpu").load_state_dict(sd)
    score = osol.make_score_fn("ncsnpp", sd, on.parse_ncsnpp_config(g["cfg"]))
    gen = torch.Generator().manual_seed(4)
    x0 = g["x"]
    zs = [torch.randn(x0.shape, generator=gen) for _ in range(10)]
    cot = torch.randn(x0.shape, generator=gen)
    with torch.no_grad():
        xf = osol.ldsde_purify(score, x0, zs, 100, 0.001, 0.01, 5)
    ref = osol.ldsde_adjoint_grad(score, xf, cot, x0, zs, 100, 0.001, 0.01, 5)
    pur = psde.Purifier(net, "ncsnpp", "cpu")
    got = pur.ldsde_vjp(xf, cot, x0, 100, 0.001, 0.01, 5, noise=dict(z=zs))
    torch.testing.assert_close(got, ref, rtol=2e-3, atol=2e-4 * ref.abs().max().item())
