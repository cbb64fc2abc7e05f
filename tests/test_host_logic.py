"""CPU checks of the engine's HOST logic (block wiring, weight packing, channel-split inputs, time
tables, schedules) with the HIP operators replaced by their torch statements (tests/refops.py).
The kernels themselves are checked on the GPU in test_gpu_*.py."""
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
    # the engine moves tensors to its device; on CPU that is a no-op
    yield


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


def make_ncsnpp(name):
    g = load_golden(name)
    cfg = pn.parse_config(g["cfg"])
    sd = synth_state_dict(pn.param_shapes(cfg), g["seed"])
    return g, pn.NCSNpp(cfg, "cpu").load_state_dict(sd), sd


def make_guided(name):
    g = load_golden(name)
    cfg = pg.parse_config(g["cfg"])
    sd = synth_state_dict(pg.param_shapes(cfg), g["seed"])
    return g, pg.GuidedUNet(cfg, "cpu").load_state_dict(sd), sd


def test_ncsnpp_small_engine_wiring():
    g, net, _ = make_ncsnpp("ncsnpp_small.pt")
    out = nchw(net.forward(nhwc(g["x"]), g["labels"]))
    torch.testing.assert_close(out, g["out"], rtol=2e-4, atol=2e-5)


def test_guided_small_engine_wiring():
    g, net, _ = make_guided("guided_small.pt")
    out = nchw(net.forward(nhwc(g["x"]), g["t"].float()))
    torch.testing.assert_close(out, g["out"], rtol=2e-4, atol=2e-5)


def test_ncsnpp_full_engine_wiring():
    g, net, _ = make_ncsnpp("ncsnpp_full.pt")
    out = nchw(net.forward(nhwc(g["x"]), g["labels"]))
    torch.testing.assert_close(out, g["out"], rtol=1e-3, atol=1e-4)


def test_param_shapes_match_reference_keys():
    # golden files were produced by loading synth weights BY KEY into the reference modules with
    # strict checking of missing keys, so equality of outputs above already pins names + shapes;
    # here: the EMA order (all_modules order) and counts from SURVEY Appendix B.
    cfg = pn.parse_config(load_golden("ncsnpp_full.pt")["cfg"])
    shapes = pn.param_shapes(cfg)
    n = sum(int(torch.tensor(s).prod()) for k, s in shapes.items() if k != "sigmas")
    assert n == 106632579
    keys = [k for k in shapes if k != "sigmas"]
    assert keys[:4] == ["all_modules.0.weight", "all_modules.0.bias", "all_modules.1.weight", "all_modules.1.bias"]
    gcfg = pg.parse_config(load_golden("guided_full.pt")["cfg"])
    gs = pg.param_shapes(gcfg)
    assert sum(int(torch.tensor(s).prod()) for s in gs.values()) == 552814086


@pytest.mark.parametrize("kind", ["ncsnpp", "guided"])
def test_sde_loop_matches_oracle(kind):
    """5 coarse EM steps with injected noise: engine loop (patched ops) == oracle loop."""
    if kind == "ncsnpp":
        g, net, sd = make_ncsnpp("ncsnpp_small.pt")
        score = osol.make_score_fn("ncsnpp", sd, on.parse_ncsnpp_config(g["cfg"]))
    else:
        g, net, sd = make_guided("guided_small.pt")
        score = osol.make_score_fn("guided", sd, og.parse_guided_config(g["cfg"]))
    x0 = g["x"]
    gen = torch.Generator().manual_seed(7)
    t_int, dt = 100, 2e-2
    nsteps = len(osol.sde_time_grid(t_int, dt)) - 1
    assert nsteps == 5
    e = torch.randn(x0.shape, generator=gen)
    zs = [torch.randn(x0.shape, generator=gen) for _ in range(nsteps)]
    ref = osol.sde_purify(score, x0, e, zs, t_int, dt)
    pur = psde.Purifier(net, kind, "cpu")
    out = pur.sde(x0, t_int, dt, noise=dict(e=e, z=zs))
    torch.testing.assert_close(out, ref, rtol=1e-4, atol=1e-4)


def test_ode_loop_matches_oracle():
    g, net, sd = make_ncsnpp("ncsnpp_small.pt")
    score = osol.make_score_fn("ncsnpp", sd, on.parse_ncsnpp_config(g["cfg"]))
    x0 = g["x"]
    e = torch.randn(x0.shape, generator=torch.Generator().manual_seed(3))
    ref = osol.ode_purify(score, x0, e, 100, step=2e-2)
    out = psde.Purifier(net, "ncsnpp", "cpu").ode(x0, 100, step=2e-2, noise=dict(e=e, z=[]))
    torch.testing.assert_close(out, ref, rtol=1e-4, atol=1e-4)


def test_ddpm_loop_matches_oracle():
    g, net, sd = make_guided("guided_small.pt")
    cfg = og.parse_guided_config(g["cfg"])
    unet = lambda x, ts: og.guided_unet_forward(sd, cfg, x, ts)
    x0 = g["x"]
    gen = torch.Generator().manual_seed(5)
    e = torch.randn(x0.shape, generator=gen)
    zs = [torch.randn(x0.shape, generator=gen) for _ in range(4)]
    ref = osol.ddpm_purify(unet, x0, e, zs, 4)
    out = psde.Purifier(net, "guided", "cpu").ddpm(x0, 4, noise=dict(e=e, z=zs))
    torch.testing.assert_close(out, ref, rtol=1e-4, atol=1e-4)


def test_schedules_follow_reference_clock():
    sch = psde.sde_schedule("guided", 100)
    assert len(sch) == 100 and len(psde.sde_schedule("guided", 150)) == 150
    # (s*1000).long() must walk 100, 99, ..., 1 without an off-by-one from float rounding
    ref_t = [float((1 - t).mul(1000).long()) for t in osol.sde_time_grid(100)[:-1]]
    assert [s["model_time"] for s in sch] == ref_t
    assert abs(sch[-1]["h"] - 9.9e-4) < 2e-5
    assert len(psde.ode_schedule("ncsnpp", 100)) == 100
    assert len(psde.ode_schedule("ncsnpp", 100, reverse=True)) == 100


def test_philox_reference_stream_is_shard_invariant():
    full = refops.philox_normal((4, 3, 8, 8), seed=1234, sample0=0, step=3)
    part = refops.philox_normal((2, 3, 8, 8), seed=1234, sample0=2, step=3)
    assert torch.equal(full[2:], part)
    assert abs(full.mean()) < 0.15 and abs(full.std() - 1) < 0.15
    other = refops.philox_normal((4, 3, 8, 8), seed=1234, sample0=0, step=4)
    assert not torch.equal(full, other)


@pytest.mark.parametrize("precision", ["f16x3", "f16x2", "f16", "f16sr"])
@pytest.mark.parametrize("kind", ["ncsnpp", "guided"])
def test_f16x3_mode_engine_wiring(kind, precision):
    """precision="f16x3" / "f16x2" / "f16": GroupNorm emits the convolution operand format of the mode (split-fp16 "h2" or
    plain fp16 "h1") and the 3x3 / qkv convolutions consume it with pre-split weights; on CPU the ops are the torch
    statements of the same contract."""
    tol = dict(f16x3=(2e-4, 3e-5), f16x2=(2e-3, 2e-3), f16=(1e-2, 1e-2), f16sr=(1e-2, 1e-2))[precision]
    if kind == "ncsnpp":
        g = load_golden("ncsnpp_small.pt")
        cfg = pn.parse_config(g["cfg"])
        net = pn.NCSNpp(cfg, "cpu", precision=precision).load_state_dict(synth_state_dict(pn.param_shapes(cfg), g["seed"]))
        out = nchw(net.forward(nhwc(g["x"]), g["labels"]))
        assert all(r["h2_0"] and r["h2_1"] for b in net.plan["down"] for r in b if r["kind"] == "res")
    else:
        g = load_golden("guided_small.pt")
        cfg = pg.parse_config(g["cfg"])
        net = pg.GuidedUNet(cfg, "cpu", precision=precision).load_state_dict(synth_state_dict(pg.param_shapes(cfg), g["seed"]))
        out = nchw(net.forward(nhwc(g["x"]), g["t"].float()))
        assert net._out_h2 and net.p["out.w"].dtype == torch.float16
        if precision in ("f16", "f16sr"):       # plain fp16 panels: views into ONE pool buffer
            assert net.p["out.w"].untyped_storage().data_ptr() == net._pool.work.untyped_storage().data_ptr()
    torch.testing.assert_close(out, g["out"], rtol=tol[0], atol=tol[1])
    if precision == "f16sr":                    # a new key re-rounds the weights: same network, different fp16 panels
        before = net._pool.work.clone()
        net.reround(5)
        changed = (net._pool.work != before).float().mean().item()
        assert 0.2 < changed < 0.8, changed
        assert (net._pool.work.float() - net._pool.master).abs().max() <= net._pool.master.abs().max() * 2 ** -10
        net.reround(0)
        assert torch.equal(net._pool.work, before)      # keyed: reproducible
    if precision != "f16x3":
        assert (out - g["out"]).abs().max() > 1e-6       # the mode really rounds its operands


@pytest.mark.parametrize("kind", ["ncsnpp", "guided"])
def test_fp16_residual_stream_and_fused_skip_wiring(kind, monkeypatch):
    """Round 4, fp16 x fp16 modes: the residual stream travels as plain fp16 wherever a level has whole column records per sample
    (H*W % 64 == 0), every channel-changing ResBlock without resampling folds its 1x1 skip into its second convolution as
    K-segments over the raw fp16 input(s); the taped forward (adjoints) runs the same launches (round 5).  Traced on the torch statements of the ops:
    dtypes of every convolution's residual / output, the K-segment calls, and the result against the reference golden with the
    fp16 stream on and off (DIFFPURE_LEAN16=0)."""
    from diffpure_amd import ops
    if kind == "ncsnpp":
        g = load_golden("ncsnpp_small.pt")
        mod, cfg = pn, pn.parse_config(g["cfg"])
        build = lambda: pn.NCSNpp(cfg, "cpu", precision="f16sr").load_state_dict(synth_state_dict(pn.param_shapes(cfg), g["seed"]))
        run = lambda net, **kw: net.forward(nhwc(g["x"]), g["labels"], **kw)
    else:
        g = load_golden("guided_small.pt")
        mod, cfg = pg, pg.parse_config(g["cfg"])
        build = lambda: pg.GuidedUNet(cfg, "cpu", precision="f16sr").load_state_dict(synth_state_dict(pg.param_shapes(cfg), g["seed"]))
        run = lambda net, **kw: net.forward(nhwc(g["x"]), g["t"].float(), **kw)
    calls = []
    real = ops.conv2d_h2

    def spy(x, wh, n_out, ksize, **kw):
        y = real(x, wh, n_out, ksize, **kw)
        t = ops.tensor_of(y)
        calls.append(dict(k=ksize, hw=t.shape[1] * t.shape[2], out=t.dtype, res=None if kw.get("res") is None else kw["res"].dtype,
                          segs=None if not kw.get("segs") else [s.dtype for s in kw["segs"]]))
        return y

    monkeypatch.setattr(ops, "conv2d_h2", spy)
    net = build()
    assert net._lean16
    out = nchw(run(net))
    torch.testing.assert_close(out, g["out"], rtol=1e-2, atol=1e-2)
    stream = [c for c in calls if c["k"] == 3 and (c["res"] is not None or c["segs"])]      # second convolutions of the ResBlocks
    assert stream and all(c["out"] == (torch.float16 if c["hw"] % 64 == 0 else torch.float32) for c in stream)
    assert any(c["out"] == torch.float16 for c in stream)
    fused = [c for c in calls if c["segs"]]
    assert fused and all(all(d == torch.float16 for d in c["segs"]) and c["res"] is None for c in fused)
    assert any(len(c["segs"]) == 2 for c in fused)                     # a decoder block: two skip sources
    n_fusable = sum(1 for r in ([r for b in net.plan["down"] for r in b] + net.plan["mid"] + net.plan["up"] if kind == "ncsnpp" else
                                [r for b in net.plan["inp"] for r in b] + net.plan["mid"] + [r for b in net.plan["out"] for r in b])
                    if r["kind"] == "res" and ((r["cin"] != r["cout"] and not r["mode"]) or (kind == "ncsnpp" and r["mode"])))
    # (round 6: the up / down blocks of NCSN++ fold their Conv_2 too - over the resampled raw input GroupNorm-apply hands out)
    assert len(fused) <= n_fusable and len(fused) >= 1
    if kind == "ncsnpp":
        assert any(len(c["segs"]) == 1 and c["res"] is None for c in fused)
    # round 5: the taped forward (adjoints) issues EXACTLY the launches of the untaped one - same stream format, same fused skips - and
    # gives the same bits; the tape then holds fp16 tensors (round 4 kept an fp32-stream variant under a tape)
    first = list(calls)
    calls.clear()
    tape = []
    out_t = nchw(run(net, tape=tape))
    assert calls == first and torch.equal(out_t, out)
    res_t = [t_ for t_ in tape if "hmid" in t_]
    assert res_t and any(t_["x"].dtype == torch.float16 for t_ in res_t) and any(t_["hmid"].dtype == torch.float16 for t_ in res_t)
    # ... and DIFFPURE_TAPE16=0 restores the fp32-stream tape: fp32 outputs on the stream, no fused skips
    monkeypatch.setenv("DIFFPURE_TAPE16", "0")
    net32 = build()
    calls.clear()
    tape = []
    run(net32, tape=tape)
    assert all(c["out"] == torch.float32 or c["res"] is None for c in calls if c["k"] == 3 and c["res"] is not None)
    assert not any(c["segs"] for c in calls)
    assert all(t_["x"].dtype == torch.float32 for t_ in tape if "hmid" in t_)
    monkeypatch.delenv("DIFFPURE_TAPE16")
    # switch: DIFFPURE_LEAN16=0 -> round 3's fp32 stream, same function to the same tolerance
    monkeypatch.setenv("DIFFPURE_LEAN16", "0")
    net0 = build()
    assert not net0._lean16
    calls.clear()
    out0 = nchw(run(net0))
    assert not any(c["segs"] or c["res"] == torch.float16 for c in calls)
    torch.testing.assert_close(out0, g["out"], rtol=1e-2, atol=1e-2)
    assert (out0 - out).abs().max() < 2e-2


def test_h2_format_roundtrip():
    x = torch.randn(3, 5, 64) * 3
    enc = refops.h2_encode(x)
    assert enc.dtype == torch.float16 and enc.shape == (3, 5, 128)
    dec = refops.h2_decode(enc).float()
    assert (dec - x).abs().max() < 3e-6 * x.abs().max()
    # layout: first 8 halves of each 16-half block are fp16(x)
    assert torch.equal(enc.reshape(3, 5, 8, 2, 8)[:, :, :, 0], x.reshape(3, 5, 8, 8).half())


# ---- CelebA-HQ DDPM UNet engine and runner ---------------------------------------------
def make_ddpm(name, precision="f32"):
    from diffpure_amd import ddpm_unet as pd
    g = load_golden(name)
    cfg = pd.parse_config(g["cfg"])
    shapes = pd.param_shapes(cfg)
    assert list(shapes.keys()) == g["keys"] and [tuple(v) for v in shapes.values()] == [tuple(v) for v in g["shapes"]]
    sd = synth_state_dict(shapes, g["seed"])
    return g, pd.DdpmUNet(cfg, "cpu", precision).load_state_dict(sd), sd


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_ddpm_unet_small_engine_wiring(precision):
    g, net, _ = make_ddpm("ddpm_unet_small.pt", precision)
    out = nchw(net.forward(nhwc(g["x"]), g["t"].float()))
    torch.testing.assert_close(out, g["y"], rtol=2e-4, atol=2e-5)


def test_ddpm_unet_full_engine_param_keys():
    from diffpure_amd import ddpm_unet as pd
    g = load_golden("ddpm_unet_full.pt")
    shapes = pd.param_shapes(pd.parse_config(g["cfg"]))
    assert list(shapes.keys()) == g["keys"] and [tuple(v) for v in shapes.values()] == [tuple(v) for v in g["shapes"]]


def test_celeba_ddpm_runner_matches_oracle_loop(tmp_path):
    import argparse
    from oracle import ddpm_unet as od
    from runners.diffpure_ddpm import Diffusion
    g = load_golden("ddpm_unet_small.pt")

    def ns(d):
        n = argparse.Namespace()
        for k, v in d.items():
            setattr(n, k, ns(v) if isinstance(v, dict) else v)
        return n

    config = ns(g["cfg"])
    args = argparse.Namespace(t=6, sample_step=1, log_dir=None, seed=g["seed"], synthetic_weights=True, precision="f32")
    runner = Diffusion(args, config, device="cpu")
    sd = synth_state_dict(dict(zip(g["keys"], g["shapes"])), g["seed"])
    ocfg = od.parse_ddpm_config(g["cfg"])
    d = g["cfg"]["diffusion"]
    sched = od.CelebaSchedule(d["beta_start"], d["beta_end"], d["num_diffusion_timesteps"], g["cfg"]["model"]["var_type"])
    gen = torch.Generator().manual_seed(5)
    x0 = g["x"]
    e = torch.randn(x0.shape, generator=gen)
    zs = [torch.randn(x0.shape, generator=gen) for _ in range(6)]
    with torch.no_grad():
        ref = od.celeba_ddpm_purify(lambda x, t: od.unet_forward(sd, ocfg, x, t), sched, x0, e, zs, 6)
    out = runner.image_editing_sample(x0, bs_id=5, noise=dict(e=e, z=zs))
    torch.testing.assert_close(out, ref, rtol=1e-4, atol=1e-4)
    # Philox path: keyed by the global sample index, so shards reproduce the full batch (bit-exact on the GPU engine,
    # tests/test_gpu_models.py; torch's CPU kernels used here differ in the last bit between batch sizes)
    runner._calls = 0
    a = runner.image_editing_sample(x0, bs_id=5)
    runner._calls = 0
    b = torch.cat([runner.purifier.celeba_ddpm(x0[:1], 6, runner.sched, seed=g["seed"], sample0=0),
                   runner.purifier.celeba_ddpm(x0[1:], 6, runner.sched, seed=g["seed"], sample0=1)])
    torch.testing.assert_close(a, b, rtol=1e-5, atol=2e-6)
    # an empty batch comes back empty, in the input's shape, without a launch (the kernels refuse B = 0)
    empty = runner.image_editing_sample(x0[:0], bs_id=5)
    assert empty.shape == x0[:0].shape and empty.dtype == x0.dtype


@pytest.mark.parametrize("kind", ["ncsnpp", "guided"])
def test_ldsde_loop_matches_oracle(kind):
    """Purifier.ldsde (Langevin dynamics anchored at the input, runners/diffpure_ldsde.py) against the oracle loop."""
    if kind == "ncsnpp":
        g, net, sd = make_ncsnpp("ncsnpp_small.pt")
        score = osol.make_score_fn("ncsnpp", sd, on.parse_ncsnpp_config(g["cfg"]))
    else:
        g, net, sd = make_guided("guided_small.pt")
        score = osol.make_score_fn("guided", sd, og.parse_guided_config(g["cfg"]))
    x0 = g["x"]
    gen = torch.Generator().manual_seed(7)
    zs = [torch.randn(x0.shape, generator=gen) for _ in range(10)]
    with torch.no_grad():
        ref = osol.ldsde_purify(score, x0, zs, 100, 0.001, 0.01, 5)
    out = psde.Purifier(net, kind, "cpu").ldsde(x0, 100, 0.001, 0.01, 5, noise=dict(z=zs))
    torch.testing.assert_close(out, ref, rtol=1e-4, atol=1e-4)


def test_reference_driver_imports_resolve():
    """eval_sde_adv.py:27-31 imports five runner classes by module path; all of them exist in the drop-in package."""
    from runners.diffpure_ddpm import Diffusion  # noqa: F401
    from runners.diffpure_guided import GuidedDiffusion  # noqa: F401
    from runners.diffpure_ldsde import LDGuidedDiffusion  # noqa: F401
    from runners.diffpure_ode import OdeGuidedDiffusion  # noqa: F401
    from runners.diffpure_sde import RevGuidedDiffusion  # noqa: F401


def _runner_args(**kw):
    import argparse
    base = dict(t=100, rand_t=False, t_delta=15, use_bm=False, sample_step=1, log_dir=None, score_type="score_sde", seed=1234,
                synthetic_weights=True, dt=2e-2, precision="f32")
    base.update(kw)
    return argparse.Namespace(**base)


def _small_config():
    import argparse
    g = load_golden("ncsnpp_small.pt")

    def ns(d):
        n = argparse.Namespace()
        for k, v in d.items():
            setattr(n, k, ns(v) if isinstance(v, dict) else v)
        return n

    config = ns(g["cfg"])
    config.device = torch.device("cpu")
    return g, config


def test_rand_t_randomises_the_diffusion_level_only(monkeypatch):
    """/root/reference/runners/diffpure_sde.py:218-229: `rand_t` changes total_noise_levels of the forward diffusion; the
    reverse SDE is still integrated from t0 = 1 - args.t/1000 with the score schedule of args.t."""
    import numpy as np
    from runners.diffpure_sde import RevGuidedDiffusion
    g, config = _small_config()
    runner = RevGuidedDiffusion(_runner_args(rand_t=True, t_delta=15), config, device="cpu")
    monkeypatch.setattr(np.random, "randint", lambda lo, hi: 9)
    x0 = g["x"]
    n = len(osol.sde_time_grid(100, 2e-2)) - 1
    gen = torch.Generator().manual_seed(3)
    e = torch.randn(x0.shape, generator=gen)
    zs = [torch.randn(x0.shape, generator=gen) for _ in range(n)]
    out = runner.image_editing_sample(x0, bs_id=9, noise=dict(e=e, z=zs))
    sd = synth_state_dict(pn.param_shapes(pn.parse_config(g["cfg"])), 1234)
    score = osol.make_score_fn("ncsnpp", sd, on.parse_ncsnpp_config(g["cfg"]))
    with torch.no_grad():
        x = osol.diffuse(x0, e, 109)                       # diffusion at the randomised level ...
        grid = osol.sde_time_grid(100, 2e-2)               # ... solve over the span of args.t
        for k in range(len(grid) - 1):
            h = grid[k + 1] - grid[k]
            x = x + osol.rev_sde_f(score, grid[k], x) * h + osol.rev_sde_g(grid[k], x.shape[0])[:, None, None, None] * (zs[k] * torch.sqrt(h))
    torch.testing.assert_close(out, x, rtol=1e-4, atol=1e-4)
    wrong = osol.sde_purify(score, x0, e, zs[: len(osol.sde_time_grid(109, 2e-2)) - 1] + zs, 109, 2e-2)   # the round-1 reading
    assert (out - wrong).abs().max() > 1e-3


def test_engine_pool_serves_one_engine_per_device_and_survives_replication():
    """nn.DataParallel shallow-copies the runner for every GPU (eval_sde_adv.py:227-228): replicas share the pool and ask it
    for the engine of the device their input lives on; an engine is built once per device."""
    import copy
    from runners import _common
    built = []

    class FakePur:
        def __init__(self, dev):
            self.device = dev

    pool = _common.EnginePool(lambda dev: built.append(dev) or FakePur(dev), "cpu")
    assert pool.get("cpu") is pool.get(torch.device("cpu")) and len(built) == 1
    replica = copy.copy(pool)                                   # what replicate() does to plain attributes
    other = replica.get(torch.device("meta"))
    assert other.device.type == "meta" and len(built) == 2
    assert pool.get("meta") is other and len(built) == 2        # shared between replicas, built once
    assert pool.for_input(torch.zeros(1)) is pool.get("cpu")    # host tensors go to the home engine
    assert pool.devices() == [("cpu", None), ("meta", None)]


def test_noise_differs_across_calls_and_replicas_under_data_parallel_replication():
    """nn.DataParallel re-replicates the runner on EVERY forward (`replica.__dict__ = module.__dict__.copy()`), so a call
    counter kept as a plain attribute is bumped on a throw-away copy: every call would purify with the same Philox path and
    the randomised defence would be deterministic.  The counter lives on the shared EnginePool instead, per device; replicas
    on other GPUs add a sample offset so that image i of GPU0's slice and image i of GPU1's slice draw different noise."""
    from runners import _common
    from runners.diffpure_sde import RevGuidedDiffusion
    g, config = _small_config()
    runner = RevGuidedDiffusion(_runner_args(), config, device="cpu")
    x0 = g["x"]
    outs = []
    for _ in range(2):                                          # two DataParallel forwards = two fresh replicas
        replica = runner._replicate_for_data_parallel()
        assert replica is not runner and replica._pool is runner._pool
        outs.append(replica.image_editing_sample(x0, bs_id=5))
    assert runner._calls == 2                                   # the ORIGINAL sees both calls
    assert (outs[0] - outs[1]).abs().max() > 1e-3               # different call seeds -> different Brownian paths
    runner._calls = 0
    again = runner._replicate_for_data_parallel().image_editing_sample(x0, bs_id=5)
    torch.testing.assert_close(again, outs[0], rtol=0, atol=0)  # reproducible from the counter

    # replicas on other devices: distinct sample offsets (the home device keeps offset 0 = plain batch positions)
    class FakePur:
        def __init__(self, dev):
            self.device = dev

    pool = _common.EnginePool(lambda dev: FakePur(dev), "cpu")
    pool.get(torch.device("meta"))
    assert pool.replica_offset("cpu") == 0 and pool.replica_offset(torch.device("meta")) == 1 << 32
    assert [pool.next_call("cpu"), pool.next_call("cpu"), pool.next_call(torch.device("meta"))] == [0, 1, 0]
    pool.calls = 0
    assert pool.next_call("cpu") == 0 and pool.next_call(torch.device("meta")) == 0
    # ... and the offset reaches the engine: the same image purified as "GPU1's sample 0" differs from "GPU0's sample 0"
    a = runner.purifier.sde(x0[:1], 100, 2e-2, seed=1234, sample0=0)
    b = runner.purifier.sde(x0[:1], 100, 2e-2, seed=1234, sample0=1 << 32)
    assert (a - b).abs().max() > 1e-3


def test_ldsde_repeats_stay_anchored_at_the_original_input():
    """sample_step > 1: upstream builds LDSDE once with x_init = the ORIGINAL input (diffpure_ldsde.py:212-214); only the
    loop state is chained over the repeats."""
    from runners.diffpure_ldsde import LDGuidedDiffusion
    g, config = _small_config()
    args = _runner_args(sample_step=2, t=100, sigma2=0.001, lambda_ld=0.01, eta=5)
    runner = LDGuidedDiffusion(args, config, device="cpu")
    x0 = g["x"]
    grid = osol.sde_time_grid(100, 1e-2)
    gen = torch.Generator().manual_seed(4)
    zs = [torch.randn(x0.shape, generator=gen) for _ in range(len(grid) - 1)]
    out = runner.image_editing_sample(x0, bs_id=9, noise=dict(e=None, z=zs))
    sd = synth_state_dict(pn.param_shapes(pn.parse_config(g["cfg"])), 1234)
    score = osol.make_score_fn("ncsnpp", sd, on.parse_ncsnpp_config(g["cfg"]))

    def loop(x, anchor):
        for k in range(len(grid) - 1):
            h = grid[k + 1] - grid[k]
            f = osol.ldsde_f(score, x, anchor, 0.001, 0.01)
            x = x + f * h + osol.ldsde_g(x.shape[0], 0.01, 5)[:, None, None, None] * (zs[k] * torch.sqrt(h))
        return x

    with torch.no_grad():
        a = loop(x0, x0)
        b = loop(a, x0)                      # second repeat: state chained, anchor still the original input
    torch.testing.assert_close(out, torch.cat([a, b]), rtol=1e-4, atol=1e-4)


def test_sample_offset_separates_ranks_of_an_unsharded_evaluation(monkeypatch):
    from runners import _common
    import argparse
    assert _common.sample_offset(argparse.Namespace()) == 0
    assert _common.sample_offset(argparse.Namespace(sample_offset=77)) == 77
    monkeypatch.setattr(_common.ddist, "world", lambda: (3, 8))
    assert _common.sample_offset(argparse.Namespace()) == 3 << 40


def test_checkpoint_format_round_trip_through_the_factory():
    """a checkpoint_8.pth-shaped file written and restored by the REFERENCE's own code
    (tests/golden/make_golden_ckpt.py: 'model' entry with module.-prefixed keys that strict=False ignores, effective weights =
    the EMA shadow list in parameters() order) loads through diffpure_amd.factory.build_ncsnpp and gives the reference's output."""
    import argparse
    import os
    from conftest import GOLDEN
    from diffpure_amd import factory
    g = load_golden("ncsnpp_ckpt.pt")

    def ns(d):
        n = argparse.Namespace()
        for k, v in d.items():
            setattr(n, k, ns(v) if isinstance(v, dict) else v)
        return n

    args = argparse.Namespace(precision="f32")          # no synthetic_weights: the file must be found and used
    net, cfg = factory.build_ncsnpp(args, ns(g["cfg"]), "cpu", model_dir=os.path.join(GOLDEN, "ckpt"))
    out = nchw(net.forward(nhwc(g["x"]), g["labels"]))
    torch.testing.assert_close(out, g["out"], rtol=2e-4, atol=2e-5)
    with pytest.raises(FileNotFoundError):
        factory.build_ncsnpp(args, ns(g["cfg"]), "cpu", model_dir=os.path.join(GOLDEN, "no_such_dir"))
    ck = torch.load(os.path.join(GOLDEN, "ckpt", "checkpoint_8.pth"), map_location="cpu")
    bad = dict(ck, ema=dict(ck["ema"], shadow_params=ck["ema"]["shadow_params"][:-1]))
    with pytest.raises(ValueError):
        factory.ncsnpp_state_from_checkpoint(bad, pn.parse_config(g["cfg"]))


def test_guided_checkpoint_file_round_trip_through_the_factory():
    """A 256x256_diffusion_uncond.pt-shaped state_dict file written by the REFERENCE's own module and read back by its own load path
    (tests/golden/make_golden_guided_ckpt.py; runners/diffpure_sde.py:163-170) loads through diffpure_amd.factory.build_guided - found in
    model_dir, no synthetic-weights flag - and gives the reference's output; `use_fp16` in the config selects nothing here (the precision
    mode does), and the reference's own fp16 torso is further from its fp32 self than the engine is."""
    import argparse
    import os
    from conftest import GOLDEN
    from diffpure_amd import factory
    g = load_golden("guided_ckpt.pt")
    for use_fp16 in (False, True):
        config = argparse.Namespace(model=argparse.Namespace(**dict(g["cfg"], use_fp16=use_fp16)))
        net, mc = factory.build_guided(argparse.Namespace(precision="f32"), config, "cpu", model_dir=os.path.join(GOLDEN, "ckpt", "guided"))
        assert mc["use_fp16"] is use_fp16
        out = nchw(net.forward(nhwc(g["x"]), g["t"].float()))
        torch.testing.assert_close(out, g["out_fp32"], rtol=2e-4, atol=2e-5)
    assert (out - g["out_fp16"]).abs().max() > 1e-3 and g["fp16_vs_fp32_maxabs"] > 1e-3
    with pytest.raises(FileNotFoundError):
        factory.build_guided(argparse.Namespace(precision="f32"), config, "cpu", model_dir=os.path.join(GOLDEN, "no_such_dir"))
    bad = dict(torch.load(os.path.join(GOLDEN, "ckpt", "guided", "256x256_diffusion_uncond.pt"), map_location="cpu"))
    bad.pop("out.2.weight")
    with pytest.raises(KeyError):
        pg.GuidedUNet(pg.parse_config(g["cfg"]), "cpu").load_state_dict(bad)


def test_replica_offsets_do_not_depend_on_the_order_engines_are_built_in():
    """Advisor (round 3): the sample offset of a DataParallel replica was the ordinal of its GPU among the engines built SO FAR - built
    lazily from the replica threads, so cuda:2 before cuda:1 gave both 1 << 32.  It is a function of the device now."""
    from runners import _common

    class FakePur:
        def __init__(self, dev):
            self.device = dev

    pool = _common.EnginePool(lambda dev: FakePur(dev), "cpu")
    want = {i: (1 + i) << 32 for i in range(4)}
    assert pool.replica_offset(torch.device("cuda", 2)) == want[2]          # nothing built for it yet
    assert pool.replica_offset(torch.device("cuda", 1)) == want[1]
    pool._by_dev[("cuda", 2)] = FakePur(torch.device("cuda", 2))           # "built" in the wrong order
    pool._by_dev[("cuda", 1)] = FakePur(torch.device("cuda", 1))
    assert [pool.replica_offset(torch.device("cuda", i)) for i in range(4)] == [want[i] for i in range(4)]
    assert len({pool.replica_offset(d) for d in ("cpu", torch.device("cuda", 0), torch.device("cuda", 1), torch.device("cuda", 7))}) == 4
    assert pool.replica_offset("cpu") == 0


def test_fir_resamplers_match_the_reference_upfirdn2d():
    """tests/refops.py's statement of the FIR modes (what the HIP kernels are held to on the GPU) against
    up_or_down_sampling.upsample_2d / downsample_2d of the reference (golden: tests/golden/make_golden_fir.py)."""
    from diffpure_amd import ops
    g = load_golden("fir_ops.pt")
    taps = ops.fir_taps(g["k"])
    x = nhwc(g["x"])
    torch.testing.assert_close(nchw(refops.resample(x, ops.RESAMPLE_FIR_UP, fir=taps)), g["up"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(nchw(refops.resample(x, ops.RESAMPLE_FIR_DOWN, fir=taps)), g["down"], rtol=1e-5, atol=1e-6)


def test_ncsnpp_fir_engine_wiring():
    """`fir: True` NCSN++ (upfirdn2d resampling in the BigGAN blocks) against the reference module's forward."""
    g = load_golden("ncsnpp_fir_small.pt")
    cfg = pn.parse_config(g["cfg"])
    assert cfg["fir"] and cfg["fir_kernel"] == (1, 3, 3, 1)
    net = pn.NCSNpp(cfg, "cpu").load_state_dict(synth_state_dict(pn.param_shapes(cfg), g["seed"]))
    out = nchw(net.forward(nhwc(g["x"]), g["labels"]))
    torch.testing.assert_close(out, g["out"], rtol=2e-4, atol=2e-5)
    plain = load_golden("ncsnpp_small.pt")
    assert (g["out"] - plain["out"]).abs().max() > 1e-2 or not torch.equal(g["x"], plain["x"])    # fir changes the function
    # input gradient through the FIR blocks: the engine's backward wiring (taped forward + vjp, the resampler adjoints taken in
    # the GroupNorm backward of the h-branch and in resample_bwd of the skip branch) against torch.autograd THROUGH THE REFERENCE
    # MODULE for a seeded cotangent
    tape = []
    net.forward(nhwc(g["x"]), g["labels"], tape=tape)
    got = nchw(net.vjp(tape, nhwc(g["cot"])))
    torch.testing.assert_close(got, g["vjp"], rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize("taps", [(1, 3, 3, 1), (1, 2, 3, 4)], ids=str)
def test_fir_adjoint_stencils_match_autograd(taps):
    """The transposed FIR stencils csrc/norm_bwd.hip evaluates (restated index by index in refops.fir_adjoint_stencil) against
    torch.autograd through the upfirdn2d statement, for the symmetric filter of score_sde and an asymmetric one (which would expose
    a flipped tap order), odd and minimal sizes; and against autograd through the reference's own upsample_2d / downsample_2d
    (golden fir_ops.pt)."""
    from diffpure_amd import ops
    k = ops.fir_taps(taps)
    for mode, shape in ((3, (2, 12, 16, 8)), (4, (2, 6, 8, 8)), (3, (1, 2, 2, 4)), (4, (1, 1, 1, 4)), (3, (1, 6, 10, 4)), (4, (1, 3, 5, 4))):
        dy = torch.randn(shape, dtype=torch.float64, generator=torch.Generator().manual_seed(7))
        torch.testing.assert_close(refops.fir_adjoint_stencil(dy, mode, k), refops.resample_bwd(dy, mode, k), rtol=1e-12, atol=1e-12)
    if taps == (1, 3, 3, 1):
        g = load_golden("fir_ops.pt")
        for mode, name in ((ops.RESAMPLE_FIR_UP, "up"), (ops.RESAMPLE_FIR_DOWN, "down")):
            got = nchw(refops.fir_adjoint_stencil(nhwc(g["dy_" + name]), mode, k))
            torch.testing.assert_close(got, g["dx_" + name], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("kind", ["ncsnpp", "guided"])
def test_one_pass_attention_and_folded_skip_gradient_wiring(kind, monkeypatch):
    """Round 4: (1) in the fp16 x fp16 modes the qkv convolution of an attention block the flash kernel covers hands it an fp16 qkv
    (one fp16 pass, Q / K read in place) - since round 5 with a tape as well (the backward pass recomputes the probabilities from the taped
    fp16 qkv) - and DIFFPURE_ATTN16=0 switches it off; (2) the backward pass hands the gradient of every ResBlock's skip branch and of an attention
    block's residual to the GroupNorm backward as `addend` (one pass on the GPU where the shape allows) instead of separate add launches -
    the input gradient still equals torch.autograd through the oracle's restatement of the reference network."""
    from diffpure_amd import ops
    if kind == "ncsnpp":
        g = load_golden("ncsnpp_full.pt")                 # 16x16 attention with one head of 256 channels: the flash kernel's D = 256 form
        cfg = pn.parse_config(g["cfg"])
        sd = synth_state_dict(pn.param_shapes(cfg), g["seed"])
        build = lambda prec: pn.NCSNpp(cfg, "cpu", precision=prec).load_state_dict(sd)
        x, tt = g["x"][:1], g["labels"][:1]
        ocfg = on.parse_ncsnpp_config(g["cfg"])
        ofwd = lambda xx: on.ncsnpp_forward(sd, ocfg, xx, tt)
    else:
        g = load_golden("guided_small.pt")
        cfg = pg.parse_config(g["cfg"])
        sd = synth_state_dict(pg.param_shapes(cfg), g["seed"])
        build = lambda prec: pg.GuidedUNet(cfg, "cpu", precision=prec).load_state_dict(sd)
        x, tt = g["x"][:1], g["t"][:1].float()
        ocfg = og.parse_guided_config(g["cfg"])
        ofwd = lambda xx: og.guided_unet_forward(sd, ocfg, xx, tt)
    seen = []
    real_att = ops.attention_fused
    monkeypatch.setattr(ops, "attention_fused", lambda qkv, *a, **kw: (seen.append(qkv.dtype), real_att(qkv, *a, **kw))[1])
    net = build("f16sr")
    fusable = [r for r in (net.plan["mid"] + [r for b in net.plan.get("down", net.plan.get("inp")) for r in b]) if r["kind"] == "attn" and r.get("proj16")]
    net.forward(nhwc(x), tt)
    if fusable:
        assert seen and all(d == torch.float16 for d in seen), seen
        seen.clear()
        net.forward(nhwc(x), tt, tape=[])
        assert seen and all(d == torch.float16 for d in seen), seen       # round 5: taped = untaped (fp16 qkv, one pass); round 4 kept fp32 here
        seen.clear()
        monkeypatch.setenv("DIFFPURE_ATTN16", "0")
        net.forward(nhwc(x), tt)
        assert seen and all(d == torch.float32 for d in seen), seen
        monkeypatch.delenv("DIFFPURE_ATTN16")
    # (2) backward wiring at fp32-class arithmetic
    adds, bwd = [], []
    real_add, real_bwd = ops.add, ops.group_norm_bwd
    monkeypatch.setattr(ops, "add", lambda a, b: (adds.append(1), real_add(a, b))[1])
    monkeypatch.setattr(ops, "group_norm_bwd", lambda *a, **kw: (bwd.append(kw.get("addend") is not None), real_bwd(*a, **kw))[1])
    net32 = build("f16x3")
    tape = []
    out = net32.forward(nhwc(x), tt, tape=tape)
    cot = torch.randn(out.shape, generator=torch.Generator().manual_seed(3))
    got = nchw(net32.vjp(tape, cot))
    xr = x.clone().requires_grad_(True)
    with torch.enable_grad():
        (want,) = torch.autograd.grad(ofwd(xr), xr, nchw(cot))
    assert (got - want).abs().max() < 2e-3 * want.abs().max(), ((got - want).abs().max().item(), want.abs().max().item())
    assert sum(bwd) >= len([r for r in tape if isinstance(r, dict)]) // 2      # every ResBlock / attention block folds its skip gradient
    assert len(adds) < sum(bwd)                                                   # and hardly any separate add is left
