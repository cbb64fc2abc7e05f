"""Pin the CPU oracle against vectors produced by the reference's own modules (tests/golden/)."""
import pytest
import torch

from conftest import load_golden
from diffpure_amd.synth import synth_state_dict
from oracle import guided_unet as og
from oracle import ncsnpp as on
from oracle import solvers as osol

TOL = dict(rtol=2e-4, atol=2e-5)


def _ncsnpp(name):
    g = load_golden(name)
    cfg = on.parse_ncsnpp_config(g["cfg"])
    import diffpure_amd.ncsnpp as pn
    sd = synth_state_dict(pn.param_shapes(pn.parse_config(g["cfg"])), g["seed"])
    return g, cfg, sd


def _guided(name):
    g = load_golden(name)
    cfg = og.parse_guided_config(g["cfg"])
    import diffpure_amd.guided_unet as pg
    sd = synth_state_dict(pg.param_shapes(pg.parse_config(g["cfg"])), g["seed"])
    return g, cfg, sd


def test_ncsnpp_small_matches_reference():
    g, cfg, sd = _ncsnpp("ncsnpp_small.pt")
    out = on.ncsnpp_forward(sd, cfg, g["x"], g["labels"])
    assert g["out"].abs().mean() > 0.05  # non-vacuous weights
    torch.testing.assert_close(out, g["out"], **TOL)


def test_guided_small_matches_reference():
    g, cfg, sd = _guided("guided_small.pt")
    out = og.guided_unet_forward(sd, cfg, g["x"], g["t"])
    assert g["out"].abs().mean() > 0.05
    torch.testing.assert_close(out, g["out"], **TOL)


def test_ncsnpp_full_matches_reference():
    g, cfg, sd = _ncsnpp("ncsnpp_full.pt")
    out = on.ncsnpp_forward(sd, cfg, g["x"], g["labels"])
    torch.testing.assert_close(out, g["out"], rtol=1e-3, atol=1e-4)


def test_fir_resamplers_and_fir_ncsnpp_match_reference_forward_and_gradient():
    """`fir: True`: the oracle's upfirdn2d / upsample_2d / downsample_2d restatement against the reference's own
    functions (golden fir_ops.pt: outputs AND torch.autograd input gradients), and the fir NCSN++ against the reference module's
    forward and input gradient for a seeded cotangent (ncsnpp_fir_small.pt) - tests/golden/make_golden_fir.py."""
    g = load_golden("fir_ops.pt")
    for name, fn in (("up", on.upsample_2d), ("down", on.downsample_2d)):
        x = g["x"].clone().requires_grad_(True)
        y = fn(x, g["k"])
        torch.testing.assert_close(y, g[name], **TOL)
        (dx,) = torch.autograd.grad(y, x, g["dy_" + name])
        torch.testing.assert_close(dx, g["dx_" + name], **TOL)
    g, cfg, sd = _ncsnpp("ncsnpp_fir_small.pt")
    assert cfg["fir"] and cfg["fir_kernel"] == (1, 3, 3, 1)
    x = g["x"].clone().requires_grad_(True)
    out = on.ncsnpp_forward(sd, cfg, x, g["labels"])
    torch.testing.assert_close(out, g["out"], **TOL)
    (vjp,) = torch.autograd.grad(out, x, g["cot"])
    torch.testing.assert_close(vjp, g["vjp"], **TOL)


@pytest.mark.slow
def test_guided_full_matches_reference():
    g, cfg, sd = _guided("guided_full.pt")
    x = torch.rand(1, 3, 256, 256, generator=torch.Generator().manual_seed(g["x_seed"])) * 2 - 1
    out = og.guided_unet_forward(sd, cfg, x, g["t"])
    torch.testing.assert_close(out[:, :, ::16, ::16], g["out_crop"], rtol=1e-3, atol=2e-4)
    assert abs(out.abs().mean().item() - g["out_absmean"]) < 1e-4


def test_sde_drift_diffusion_and_ode_rhs_match_reference():
    g = load_golden("sde_fg.pt")
    gn, cfgn, sdn = _ncsnpp("ncsnpp_small.pt")
    gg, cfgg, sdg = _guided("guided_small.pt")
    cases = (("score_sde", osol.make_score_fn("ncsnpp", sdn, cfgn), g["xn"]),
             ("guided_diffusion", osol.make_score_fn("guided", sdg, cfgg), g["xg"]))
    for name, score, x in cases:
        for tp in (0.9, 0.9635, 0.99999):
            t = torch.tensor(tp, dtype=torch.float32)
            torch.testing.assert_close(osol.rev_sde_f(score, t, x), g["rec"][(name, "f", tp)], rtol=5e-4, atol=5e-4)
            torch.testing.assert_close(osol.rev_sde_g(t, 2), g["rec"][(name, "g", tp)], rtol=1e-6, atol=0)
        for s in (0.1, 0.0365, 1e-5):
            t = torch.tensor(s, dtype=torch.float32)
            torch.testing.assert_close(osol.ode_rhs(score, t, x), g["rec"][(name, "ode", s)], rtol=5e-4, atol=5e-4)


def test_ddpm_p_sample_matches_reference():
    g = load_golden("ddpm_psample.pt")
    gg, cfgg, sdg = _guided("guided_small.pt")
    sched = osol.DdpmSchedule(1000)
    unet = lambda x, ts: og.guided_unet_forward(sdg, cfgg, x, ts)
    for i, ref in g["outs"].items():
        out = osol.ddpm_p_sample(unet, sched, gg["x"], i, g["z"])
        torch.testing.assert_close(out, ref, rtol=2e-4, atol=2e-5)


def test_solver_clocks():
    # exactly 100 steps for t=100 and 150 for t=150 at dt=1e-3, short last step
    for t_int in (100, 150):
        grid = osol.sde_time_grid(t_int)
        assert len(grid) - 1 == t_int
        assert abs(float(grid[-1] - grid[-2]) - 9.9e-4) < 2e-5
    assert len(osol.sde_time_grid(100, 5e-3)) - 1 == 20
    # BASELINE.json configs[2] as written (t* = 0.15 in 100 steps, dt = 1.5e-3): 100 strides, the last one short; the engine's clock
    # (diffpure_amd/sde.py::sde_clock) is the same float32 sequence, value for value
    from diffpure_amd.sde import sde_clock
    g15 = osol.sde_time_grid(150, 1.5e-3)
    assert len(g15) - 1 == 100 and float(g15[-1] - g15[-2]) < 1.5e-3
    assert [float(v) for v in g15] == [float(v) for v in sde_clock(150, 1.5e-3)]
    ts = torch.linspace(0.1, 1e-5, 2)
    tau = osol.ode_grid(-ts, 1e-3)
    assert len(tau) == 101 and tau[0] == -ts[0] and tau[-1] == -ts[1]


# ---- CelebA-HQ DDPM UNet and its denoising step ---------------------------------------
def test_ddpm_unet_small_and_full_vs_reference_golden():
    from oracle import ddpm_unet as od
    for name, rtol, atol in (("ddpm_unet_small.pt", 1e-4, 1e-5), ("ddpm_unet_full.pt", 1e-3, 1e-4)):
        g = load_golden(name)
        cfg = od.parse_ddpm_config(g["cfg"])
        sd = synth_state_dict(dict(zip(g["keys"], g["shapes"])), g["seed"])
        res = cfg["resolution"]
        x = g["x"] if "x" in g else torch.rand(1, 3, res, res, generator=torch.Generator().manual_seed(g["x_seed"])) * 2 - 1
        with torch.no_grad():
            y = od.unet_forward(sd, cfg, x, g["t"])
        if "y" in g:
            torch.testing.assert_close(y, g["y"], rtol=rtol, atol=atol)
        else:
            torch.testing.assert_close(y[:, :, ::8, ::8], g["y_crop"], rtol=rtol, atol=atol)
            torch.testing.assert_close(y.abs().mean(), g["y_abs_mean"], rtol=1e-4, atol=0)


def test_celeba_denoising_step_vs_reference_golden():
    from oracle import ddpm_unet as od
    g, s = load_golden("ddpm_unet_small.pt"), load_golden("celeba_step.pt")
    cfg = od.parse_ddpm_config(g["cfg"])
    sd = synth_state_dict(dict(zip(g["keys"], g["shapes"])), g["seed"])
    d = g["cfg"]["diffusion"]
    sched = od.CelebaSchedule(d["beta_start"], d["beta_end"], d["num_diffusion_timesteps"], "fixedsmall")
    unet = lambda x, t: od.unet_forward(sd, cfg, x, t)
    with torch.no_grad():
        for i, ref in s["out"].items():
            torch.testing.assert_close(od.denoising_step(unet, sched, s["x"], i, s["z"]), ref, rtol=1e-5, atol=1e-6)


def test_ldsde_f_g_vs_reference_golden():
    """oracle/solvers.py ldsde_f / ldsde_g against LDSDE.f / .g of runners/diffpure_ldsde.py on both small networks."""
    r = load_golden("ldsde_fg.pt")
    for kind, name, gname in (("ncsnpp", "score_sde", "ncsnpp_small.pt"), ("guided", "guided_diffusion", "guided_small.pt")):
        g = load_golden(gname)
        if kind == "ncsnpp":
            cfg, shapes = on.parse_ncsnpp_config(g["cfg"]), None
            from diffpure_amd import ncsnpp as pn
            sd = synth_state_dict(pn.param_shapes(pn.parse_config(g["cfg"])), g["seed"])
        else:
            cfg = og.parse_guided_config(g["cfg"])
            from diffpure_amd import guided_unet as pg
            sd = synth_state_dict(pg.param_shapes(pg.parse_config(g["cfg"])), g["seed"])
        score = osol.make_score_fn(kind, sd, cfg)
        with torch.no_grad():
            f = osol.ldsde_f(score, r["rec"][(name, "x")], g["x"], r["sigma2"], r["lambda_ld"])
        torch.testing.assert_close(f, r["rec"][(name, "f")], rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(osol.ldsde_g(2, r["lambda_ld"], r["eta"]), r["rec"][(name, "g")], rtol=1e-6, atol=0)


def test_oracle_stochastic_adjoint_at_the_product_grid_vs_reference_golden():
    """The oracle's restated SDE loop + stochastic adjoint (oracle/solvers.py sde_purify / sde_adjoint_grad) on the product grid
    (t* = 0.1, dt = 1e-3, full-size NCSN++, B=2) against the golden the reference's own RevVPSDE.f / .g and torch.autograd through
    the reference NCSNpp produced on the same Philox path (make_golden_loops.py ncsnpp_sde_adjoint): the whole 100-step forward
    solve, and the first 10 adjoint steps from the golden's x_final (the golden stores the adjoint state there; the remaining 90
    steps repeat the same update and are covered at full length on the GPU, tests/test_gpu_loops.py).  ~40 s of CPU."""
    import refops
    g, cfg, sd = _ncsnpp("ncsnpp_sde_adjoint100.pt")
    score = osol.make_score_fn("ncsnpp", sd, cfg)
    x0, t_int, dt, seed = g["x0"], g["t"], g["dt"], g["noise_seed"]
    b, c, h, w = x0.shape
    ph = lambda step: refops.philox_normal((b, h, w, c), seed, 0, step).permute(0, 3, 1, 2).contiguous()
    zs = [ph(k) for k in range(g["steps"])]
    with torch.no_grad():
        xf = osol.sde_purify(score, x0, ph(-1), zs, t_int, dt)
    assert (xf - g["x_final"]).abs().max() < 2e-5, (xf - g["x_final"]).abs().max()
    snap = g["snap"]
    assert snap["k_stop"] == g["steps"] - 10
    a, y = osol.sde_adjoint_grad(score, g["x_final"], g["cot"], zs, t_int, dt, k_stop=snap["k_stop"], return_state=True)
    assert (y - snap["y"]).abs().max() < 2e-5, (y - snap["y"]).abs().max()
    assert (a - snap["a"]).abs().max() < 1e-4 * snap["a"].abs().max(), ((a - snap["a"]).abs().max(), snap["a"].abs().max())


def test_oracle_config1_loop_vs_reference_golden():
    """BASELINE.json configs[0] as written (CIFAR NCSN++, B=4, t*=0.1 in 20 EM steps of dt=5e-3): the oracle's restated loop against the file
    the reference's own NCSNpp + RevVPSDE.f / .g wrote on the same Philox path (make_golden_loops.py ncsnpp_loop20).  ~10 s of CPU."""
    import refops
    g, cfg, sd = _ncsnpp("ncsnpp_loop20_dt0.005.pt")
    score = osol.make_score_fn("ncsnpp", sd, cfg)
    x0, seed = g["x0"], g["noise_seed"]
    b, c, h, w = x0.shape
    ph = lambda step: refops.philox_normal((b, h, w, c), seed, 0, step).permute(0, 3, 1, 2).contiguous()
    assert g["steps"] == 20
    with torch.no_grad():
        xf = osol.sde_purify(score, x0, ph(-1), [ph(k) for k in range(20)], g["t"], g["dt"])
    assert (xf - g["out"]).abs().max() < 2e-5, (xf - g["out"]).abs().max()


# ---- round 5: the oracle against the reference's OWN modules, live (oracle/_ref travels to the GPU box; skipped where it is absent) -------------
@pytest.mark.parametrize("kind", ["guided", "ncsnpp"])
def test_oracle_equals_the_reference_modules_of_oracle_ref_on_fresh_inputs(kind):
    """oracle/_ref is a byte-exact, sha256-tracked copy of the reference's guided_diffusion/ + score_sde/models/ (oracle/make_ref.py; git-ignored,
    travels with the snapshot).  The golden files pin the oracle at the inputs they were generated on; this test pins it on FRESH seeded inputs,
    forward and input gradient, against the reference's nn.Module itself - wherever the copy is present and verified (build container AND GPU box)."""
    from oracle import ref_loader
    if not ref_loader.available():
        pytest.skip("oracle/_ref absent or not matching oracle/ref_modules.sha256 (python oracle/make_ref.py needs /root/reference)")
    gen = torch.Generator().manual_seed(20260926)
    if kind == "guided":
        g, cfg, sd = _guided("guided_small.pt")
        mod = ref_loader.guided_unet(g["cfg"], sd)
        x = torch.rand(g["x"].shape, generator=gen) * 2 - 1
        t = torch.tensor([3.0, 977.0])[: x.shape[0]]
        ref_fn, ora_fn = (lambda xx: mod(xx, t)), (lambda xx: og.guided_unet_forward(sd, cfg, xx, t))
    else:
        g, cfg, sd = _ncsnpp("ncsnpp_small.pt")
        mod = ref_loader.ncsnpp(g["cfg"], sd)
        x = torch.rand(g["x"].shape, generator=gen) * 2 - 1
        lab = torch.tensor([0.91 * 999, 0.002 * 999])[: x.shape[0]]
        ref_fn, ora_fn = (lambda xx: mod(xx, lab)), (lambda xx: on.ncsnpp_forward(sd, cfg, xx, lab))
    cot = None
    outs = []
    for fn in (ref_fn, ora_fn):
        xr = x.clone().requires_grad_(True)
        y = fn(xr)
        cot = torch.randn(y.shape, generator=torch.Generator().manual_seed(7)) if cot is None else cot
        (gx,) = torch.autograd.grad(y, xr, cot)
        outs.append((y.detach(), gx))
    assert outs[0][0].abs().mean() > 0.05
    torch.testing.assert_close(outs[1][0], outs[0][0], **TOL)
    torch.testing.assert_close(outs[1][1], outs[0][1], rtol=1e-3, atol=1e-4 * outs[0][1].abs().max().item())
