"""The UNCHANGED reference driver over this repository's runners (build container only: needs /root/reference).

/root/reference/eval_sde_adv.py is loaded from a scratch copy exactly as it is; only the packages that are not
installable here and have nothing to do with the path are stubbed (autoattack, stadv_eot, and the reference's own
utils.py, whose imports pull torchvision / lmdb / robustbench).  `from runners.diffpure_sde import RevGuidedDiffusion`
etc. (eval_sde_adv.py:27-31) then resolve to THIS repository's drop-in package, the reference's own
`SDE_Adv_Model(args, config)` is constructed (:34-60) and its `forward` (:67-93) and the gradient an adaptive attack
takes through it are run.  Device work goes through the torch statements of the HIP operators (tests/refops.py) - the
kernels themselves are covered on the GPU (tests/test_gpu_*.py); what is proven here is the boundary: module paths,
class names, constructor and method signatures, nn.Module ownership, differentiability, buffers / .eval() / .to()."""
import argparse
import importlib.util
import os
import shutil
import sys
import types

import pytest
import torch

import refops
from conftest import ROOT, load_golden

REF_DRIVER = "/root/reference/eval_sde_adv.py"
# the byte-identical copy that travels to the GPU box (tests/golden/make_ref_driver_fixture.py: git-ignored, sha256 tracked)
FIXTURE = os.path.join(ROOT, "tests", "golden", "_ref_driver", "eval_sde_adv.py")
FIXTURE_SHA = os.path.join(ROOT, "tests", "golden", "ref_driver.sha256")
needs_reference = pytest.mark.skipif(not os.path.exists(REF_DRIVER), reason="the reference checkout is not present on this machine")
needs_fixture = pytest.mark.skipif(not os.path.exists(FIXTURE), reason="tests/golden/_ref_driver/eval_sde_adv.py not generated "
                                   "(python tests/golden/make_ref_driver_fixture.py in the build container)")


class _Classifier(torch.nn.Module):
    """stands in for utils.get_image_classifier(...) (the classifier zoo is outside the scope contract)"""

    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(torch.randn(7, 3, generator=torch.Generator().manual_seed(3)))

    def forward(self, x):
        assert x.dim() == 4
        return x.mean(dim=(2, 3)) @ self.w.t()


def _load_driver(path, tmp_path, monkeypatch):
    monkeypatch.setenv("DIFFPURE_SYNTH_WEIGHTS", "1")        # no checkpoint files here; the driver's args stay untouched
    scratch = tmp_path / "eval_sde_adv.py"
    shutil.copyfile(path, scratch)
    stubs = {}
    aa = types.ModuleType("autoattack")
    aa.AutoAttack = object
    stubs["autoattack"] = aa
    st = types.ModuleType("stadv_eot")
    sta = types.ModuleType("stadv_eot.attacks")
    sta.StAdvAttack = object
    st.attacks = sta
    stubs["stadv_eot"], stubs["stadv_eot.attacks"] = st, sta
    ut = types.ModuleType("utils")
    ut.str2bool = lambda v: str(v).lower() in ("1", "true", "yes")
    ut.get_accuracy = lambda *a, **k: 0.0
    ut.load_data = lambda *a, **k: None
    ut.get_image_classifier = lambda name: _Classifier()
    ut.Logger = object
    stubs["utils"] = ut
    for k, v in stubs.items():
        monkeypatch.setitem(sys.modules, k, v)
    monkeypatch.syspath_prepend(ROOT)                         # `runners` = this repository's drop-in package
    for k in [k for k in sys.modules if k == "runners" or k.startswith("runners.")]:
        monkeypatch.delitem(sys.modules, k)
    spec = importlib.util.spec_from_file_location("ref_eval_sde_adv", str(scratch))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.RevGuidedDiffusion.__module__ == "runners.diffpure_sde"
    assert os.path.realpath(sys.modules["runners"].__path__[0]) == os.path.realpath(os.path.join(ROOT, "runners"))
    return mod


@pytest.fixture()
def ref_driver(tmp_path, monkeypatch):
    """CPU: the device work goes through the torch statements of the operators (tests/refops.py), fp32"""
    refops.patch_ops(monkeypatch)
    monkeypatch.setenv("DIFFPURE_PRECISION", "f32")
    return _load_driver(REF_DRIVER, tmp_path, monkeypatch)


@pytest.fixture()
def ref_driver_gpu(tmp_path, monkeypatch):
    """GPU box: the travelling copy of the SAME file (sha256 checked), the HIP engine, the runners' default precision"""
    import hashlib
    want = open(FIXTURE_SHA).read().split()[0]
    assert hashlib.sha256(open(FIXTURE, "rb").read()).hexdigest() == want, "fixture is not the reference's eval_sde_adv.py"
    monkeypatch.delenv("DIFFPURE_PRECISION", raising=False)
    return _load_driver(FIXTURE, tmp_path, monkeypatch)


def _ns(d):
    n = argparse.Namespace()
    for k, v in d.items():
        setattr(n, k, _ns(v) if isinstance(v, dict) else v)
    return n


def _args(diffusion_type, tmp_path, **kw):
    # the fields the reference's argparse defines and the runners read (eval_sde_adv.py:176-213)
    base = dict(diffusion_type=diffusion_type, domain="cifar10", classifier_name="cifar10-wideresnet-28-10", t=100, rand_t=False,
                t_delta=15, use_bm=False, sample_step=1, log_dir=str(tmp_path / "log"), score_type="score_sde", seed=1234,
                step_size=5e-2, sigma2=1e-3, lambda_ld=1e-2, eta=5.0, eot_iter=1)
    base.update(kw)
    return argparse.Namespace(**base)


@needs_reference
@pytest.mark.parametrize("diffusion_type", ["sde", "ode", "ldsde"])
def test_reference_sde_adv_model_runs_and_differentiates_over_the_drop_in_runners(ref_driver, tmp_path, diffusion_type):
    g = load_golden("ncsnpp_small.pt")
    config = _ns(g["cfg"])
    config.device = torch.device("cpu")
    args = _args(diffusion_type, tmp_path, dt=5e-2)
    model = ref_driver.SDE_Adv_Model(args, config)            # the reference's class, unchanged
    assert isinstance(model.runner, torch.nn.Module) and type(model.runner).__module__.startswith("runners.")
    model = model.eval().to(config.device)                    # as the driver does (:228)
    model.set_tag("t0")
    x = torch.rand(2, 3, 16, 16, generator=torch.Generator().manual_seed(1)).requires_grad_(True)
    logits = model(x)                                         # SDE_Adv_Model.forward (:67-93)
    assert logits.shape == (2, 7) and torch.isfinite(logits).all()
    assert model.counter.item() == 1
    assert os.path.isdir(os.path.join(args.log_dir, "bs0.0_t0"))     # bs_id arrives as a Python float (counter.item())
    (gx,) = torch.autograd.grad(logits.sum(), x)              # what AutoAttack's APGD takes
    assert gx.shape == x.shape and torch.isfinite(gx).all() and gx.abs().max() > 0
    again = model(x.detach())                                  # counter = 1: no logging, different noise (call counter)
    assert again.shape == (2, 7) and not torch.equal(again, logits.detach())


@needs_reference
def test_reference_sde_adv_model_ddpm_on_the_guided_runner(ref_driver, tmp_path):
    g = load_golden("guided_small.pt")
    config = _ns(dict(model=g["cfg"], data=dict(dataset="ImageNet", image_size=32)))
    config.device = torch.device("cpu")
    args = _args("ddpm", tmp_path, t=4, score_type="guided_diffusion")
    model = ref_driver.SDE_Adv_Model(args, config).eval().to(config.device)
    x = torch.rand(2, 3, 32, 32, generator=torch.Generator().manual_seed(2))
    with torch.no_grad():
        logits = model(x)
    assert logits.shape == (2, 7) and torch.isfinite(logits).all()


# ---- the same, on the GPU: the reference's unchanged SDE_Adv_Model over the HIP engine at the shipped precision --------------
@needs_fixture
@pytest.mark.gpu
@pytest.mark.parametrize("diffusion_type", ["sde", "ode", "ldsde"])
def test_reference_sde_adv_model_on_the_hip_engine(ref_driver_gpu, tmp_path, diffusion_type):
    """eval_sde_adv.py:34-93 as it is: SDE_Adv_Model(args, config) picks the drop-in runner, .eval().to(device), forward (logs for
    counter < 2), and torch.autograd.grad through it - every kernel launch is the product library's (no patched operator)."""
    from diffpure_amd import factory
    g = load_golden("ncsnpp_small.pt")
    config = _ns(g["cfg"])
    config.device = torch.device("cuda:0")
    args = _args(diffusion_type, tmp_path, dt=5e-2)
    model = ref_driver_gpu.SDE_Adv_Model(args, config)
    assert type(model.runner).__module__.startswith("runners.")
    assert model.runner.model.precision == factory.DEFAULT_PRECISION          # what ships, not a test-only arithmetic
    model = model.eval().to(config.device)
    model.set_tag("t0")
    x = torch.rand(2, 3, 16, 16, generator=torch.Generator().manual_seed(1)).to(config.device).requires_grad_(True)
    logits = model(x)
    assert logits.shape == (2, 7) and logits.is_cuda and torch.isfinite(logits).all()
    assert model.counter.item() == 1 and os.path.isdir(os.path.join(args.log_dir, "bs0.0_t0"))
    (gx,) = torch.autograd.grad(logits.sum(), x)
    assert gx.shape == x.shape and torch.isfinite(gx).all() and gx.abs().max() > 0
    again = model(x.detach())
    assert again.shape == (2, 7) and not torch.equal(again, logits.detach())      # a new call draws a new Brownian path


@needs_fixture
@pytest.mark.gpu
def test_reference_sde_adv_model_ddpm_on_the_hip_engine(ref_driver_gpu, tmp_path):
    g = load_golden("guided_small.pt")
    config = _ns(dict(model=g["cfg"], data=dict(dataset="ImageNet", image_size=32)))
    config.device = torch.device("cuda:0")
    args = _args("ddpm", tmp_path, t=4, score_type="guided_diffusion")
    model = ref_driver_gpu.SDE_Adv_Model(args, config).eval().to(config.device)
    x = torch.rand(2, 3, 32, 32, generator=torch.Generator().manual_seed(2)).to(config.device)
    with torch.no_grad():
        logits = model(x)
    assert logits.shape == (2, 7) and torch.isfinite(logits).all()


@needs_fixture
@pytest.mark.gpu
def test_reference_driver_under_nn_data_parallel_on_the_hip_engine(ref_driver_gpu, tmp_path):
    """eval_sde_adv.py:227-228 wraps the model in nn.DataParallel when several GPUs are visible.  One GPU here, so the two
    replicas alias cuda:0 (device_ids=[0, 0]): scatter, per-forward replication (the runner's __dict__ is copied), one thread
    per replica, gather - all real; the replicas share one engine and take its lock in turn.  Two forwards must draw
    different noise (the call counter lives on the shared pool, not on the throw-away replicas), and with the counter reset
    the call is reproducible."""
    g = load_golden("ncsnpp_small.pt")
    config = _ns(g["cfg"])
    config.device = torch.device("cuda:0")
    args = _args("sde", tmp_path, dt=5e-2)
    model = ref_driver_gpu.SDE_Adv_Model(args, config).eval().to(config.device)
    model.counter.fill_(5)                                      # no logging
    dp = torch.nn.DataParallel(model, device_ids=[0, 0])
    x = torch.rand(4, 3, 16, 16, generator=torch.Generator().manual_seed(1)).to(config.device)
    with torch.no_grad():
        a = dp(x)
        b = dp(x)
    assert a.shape == (4, 7) and torch.isfinite(a).all()
    assert model.runner._calls == 4                             # two replicas x two forwards, counted on the shared pool
    assert not torch.equal(a, b)                                # round 2: identical (the counter was bumped on the copies)
    assert not torch.equal(a[:2], a[2:])                        # the two slices hold different images AND different calls
    # Same-device replicas take the pool's call indices {0, 1} in either thread order.  Both assignments are computed WITHOUT
    # DataParallel (the plain model on each slice with the counter set by hand), and every DataParallel forward from a reset
    # counter must equal one of the two - bit for bit.
    def plain(sl, call):
        model.runner._calls = call
        with torch.no_grad():
            return model(x[sl])
    lo, hi = slice(0, 2), slice(2, 4)
    order_a = torch.cat([plain(lo, 0), plain(hi, 1)])           # replica 0 ran first
    order_b = torch.cat([plain(lo, 1), plain(hi, 0)])           # replica 1 ran first
    assert not torch.equal(order_a, order_b)
    assert torch.equal(a, order_a) or torch.equal(a, order_b)
    for _ in range(2):
        model.runner._calls = 0
        with torch.no_grad():
            c = dp(x)
        assert torch.equal(c, order_a) or torch.equal(c, order_b)


# ---- round 6: the SECOND caller of the boundary - the unchanged eval_sde_adv_bpda.py and the attack class it drives -------
REF_BPDA = "/root/reference/eval_sde_adv_bpda.py"
REF_ATTACK = "/root/reference/bpda_eot/bpda_eot_attack.py"
FIX_DIR = os.path.join(ROOT, "tests", "golden", "_ref_driver")
needs_bpda_fixture = pytest.mark.skipif(not os.path.exists(os.path.join(FIX_DIR, "eval_sde_adv_bpda.py")),
                                        reason="tests/golden/_ref_driver/eval_sde_adv_bpda.py not generated (make_ref_driver_fixture.py)")


def _load_bpda_driver(driver_path, attack_path, tmp_path, monkeypatch):
    """eval_sde_adv_bpda.py as it is (its SDE_Adv_Model: :53-118) with bpda_eot/bpda_eot_attack.py as it is (BPDA_EOT_Attack: the
    class that calls model(x, mode=...)); stubbed: utils (classifier zoo, data loaders) - outside the scope contract."""
    monkeypatch.setenv("DIFFPURE_SYNTH_WEIGHTS", "1")
    (tmp_path / "bpda_eot").mkdir()
    shutil.copyfile(attack_path, tmp_path / "bpda_eot" / "bpda_eot_attack.py")
    (tmp_path / "bpda_eot" / "__init__.py").write_text("")
    scratch = tmp_path / "eval_sde_adv_bpda.py"
    shutil.copyfile(driver_path, scratch)
    ut = types.ModuleType("utils")
    ut.str2bool = lambda v: str(v).lower() in ("1", "true", "yes")
    ut.get_accuracy = lambda *a, **k: 0.0
    ut.load_data = lambda *a, **k: None
    ut.get_image_classifier = lambda name: _Classifier()
    ut.Logger = object
    monkeypatch.setitem(sys.modules, "utils", ut)
    for k in [k for k in sys.modules if k == "runners" or k.startswith("runners.") or k == "bpda_eot" or k.startswith("bpda_eot.")]:
        monkeypatch.delitem(sys.modules, k)
    monkeypatch.syspath_prepend(ROOT)                         # `runners` = this repository's drop-in package
    monkeypatch.syspath_prepend(str(tmp_path))                # `bpda_eot` = the scratch copy of the reference's package
    spec = importlib.util.spec_from_file_location("ref_eval_sde_adv_bpda", str(scratch))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.RevGuidedDiffusion.__module__ == "runners.diffpure_sde" and mod.Diffusion.__module__ == "runners.diffpure_ddpm"
    assert mod.BPDA_EOT_Attack.__module__ == "bpda_eot.bpda_eot_attack"
    return mod


def _bpda_round_trip(mod, args, config, x, reps, device):
    """what eval_bpda does with the model (eval_sde_adv_bpda.py:121-178 -> bpda_eot_attack.py:98-125), minus the data loader"""
    model = mod.SDE_Adv_Model(args, config)                   # the reference's class, unchanged
    assert isinstance(model.runner, torch.nn.Module) and type(model.runner).__module__.startswith("runners.")
    model = model.eval().to(device)
    mod.config = config                                       # reset_counter reads a module-global `config` (:76-77: it exists when the file runs as __main__)
    model.reset_counter()
    model.set_tag("no_adv")
    with torch.no_grad():
        pur = model(x, mode="purify")
        cls = model(x, mode="classify")
        both = model(x, mode="purify_and_classify")
    assert pur.shape == x.shape and torch.isfinite(pur).all() and cls.shape == (x.shape[0], 7) and both.shape == (x.shape[0], 7)
    assert int(model.counter.item()) == 2                      # two purifications so far
    with pytest.raises(NotImplementedError):
        model(x, mode="nonsense")
    y = torch.tensor([1, 4], device=device)
    adv = mod.BPDA_EOT_Attack(model, adv_eps=8.0 / 255, eot_defense_reps=3, eot_attack_reps=reps)
    model.set_tag()
    correct, grad = adv.purify_and_predict(x, y, purify_reps=reps)        # X.repeat(reps) -> model(mode='purify') -> classify with grad
    assert correct.shape == (2,) and grad.shape == x.shape and torch.isfinite(grad).all() and grad.abs().max() > 0
    defended, grad2 = adv.eval_and_bpda_eot_grad(x, y, torch.ones(2, dtype=torch.bool, device=device))
    assert defended.shape == (2,) and grad2.shape == x.shape
    x_adv = adv.pgd_update(x.clone(), grad2, x, "l_inf", 8.0 / 255, 2.0 / 255)
    assert (x_adv - x).abs().max() <= 8.0 / 255 + 1e-6
    return model, pur


@needs_reference
@pytest.mark.parametrize("diffusion_type", ["sde", "ddpm"])
def test_reference_bpda_driver_over_the_drop_in_runners(tmp_path, monkeypatch, diffusion_type):
    refops.patch_ops(monkeypatch)
    monkeypatch.setenv("DIFFPURE_PRECISION", "f32")
    mod = _load_bpda_driver(REF_BPDA, REF_ATTACK, tmp_path, monkeypatch)
    if diffusion_type == "sde":
        g = load_golden("ncsnpp_small.pt")
        config, hw = _ns(g["cfg"]), 16
        args = _args("sde", tmp_path, dt=5e-2)
    else:
        g = load_golden("guided_small.pt")
        config, hw = _ns(dict(model=g["cfg"], data=dict(dataset="ImageNet", image_size=32))), 32
        args = _args("ddpm", tmp_path, t=3, score_type="guided_diffusion")
    config.device = torch.device("cpu")
    x = torch.rand(2, 3, hw, hw, generator=torch.Generator().manual_seed(1))
    _bpda_round_trip(mod, args, config, x, 3, config.device)


@needs_bpda_fixture
@pytest.mark.gpu
@pytest.mark.parametrize("diffusion_type", ["sde", "ddpm"])
def test_reference_bpda_driver_on_the_hip_engine(tmp_path, monkeypatch, diffusion_type):
    """The travelling copies of eval_sde_adv_bpda.py and bpda_eot_attack.py (sha256 checked) over the HIP engine at the shipped precision:
    SDE_Adv_Model(args, config), the three forward modes, and BPDA_EOT_Attack.purify_and_predict with 15 EOT replicas per image."""
    import hashlib
    from diffpure_amd import factory
    want = dict(line.split()[::-1] for line in open(FIXTURE_SHA).read().splitlines())
    for name in ("eval_sde_adv_bpda.py", "bpda_eot_attack.py"):
        assert hashlib.sha256(open(os.path.join(FIX_DIR, name), "rb").read()).hexdigest() == want[name], f"{name} is not the reference's file"
    monkeypatch.delenv("DIFFPURE_PRECISION", raising=False)
    mod = _load_bpda_driver(os.path.join(FIX_DIR, "eval_sde_adv_bpda.py"), os.path.join(FIX_DIR, "bpda_eot_attack.py"), tmp_path, monkeypatch)
    if diffusion_type == "sde":
        g = load_golden("ncsnpp_small.pt")
        config, hw = _ns(g["cfg"]), 16
        args = _args("sde", tmp_path, dt=5e-2)
    else:
        g = load_golden("guided_small.pt")
        config, hw = _ns(dict(model=g["cfg"], data=dict(dataset="ImageNet", image_size=32))), 32
        args = _args("ddpm", tmp_path, t=4, score_type="guided_diffusion")
    config.device = torch.device("cuda:0")
    x = torch.rand(2, 3, hw, hw, generator=torch.Generator().manual_seed(1)).to(config.device)
    model, pur = _bpda_round_trip(mod, args, config, x, 15, config.device)
    assert model.runner.model.precision == factory.DEFAULT_PRECISION and pur.is_cuda
