"""Drop-in for /root/reference/runners/diffpure_ode.py: `OdeGuidedDiffusion(args, config, device)`
- probability-flow ODE purification with fixed-step Euler (reference :134-249) on the MI355X engine.

`args.fix_rand` is read by the reference (:202) but defined by no argparse;
it is treated as False when absent.
"""
import os

import torch

from diffpure_amd import dist as ddist
from diffpure_amd import factory
from diffpure_amd.sde import BETA_MAX, BETA_MIN, N_DISC, Purifier

from . import _common


class _OdePurify(torch.autograd.Function):
    """purified = ODE-solve(diffuse(img)); backward = the continuous adjoint on the HIP engine
    (what torchdiffeq.odeint_adjoint provides upstream, runners/diffpure_ode.py:229-238)."""

    @staticmethod
    def forward(ctx, img, runner, t, step, noise, seed, sample0, nhwc=False):
        with torch.no_grad():
            out = runner.purifier.ode(img, t, step, noise=noise, seed=seed, sample0=sample0, nhwc=nhwc)
        ctx.runner, ctx.t, ctx.step, ctx.nhwc = runner, t, step, nhwc
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (out,) = ctx.saved_tensors
        with torch.no_grad():
            a = ctx.runner.purifier.ode_vjp(out, grad_out, ctx.t, ctx.step, nhwc=ctx.nhwc)
            a = a * ctx.runner.purifier.diffuse_scale(ctx.t)
        return a, None, None, None, None, None, None, None


class OdeGuidedDiffusion(torch.nn.Module):
    def __init__(self, args, config, device=None):
        super().__init__()
        self.args = args
        self.config = config
        if device is None:
            device = torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")
        self.device = torch.device(device)
        net, kind, img_shape = factory.build_for_dataset(args, config, self.device)
        want = factory.SCORE_TYPE_TO_KIND.get(args.score_type)
        if want is None:
            raise NotImplementedError(f"Unknown score type in RevVPSDE: {args.score_type}!")
        if want != kind:
            raise ValueError(f"score_type {args.score_type} does not match dataset {config.data.dataset}")
        self.model = net
        self.img_shape = img_shape
        self.purifier = Purifier(net, kind, self.device)
        self.betas = torch.linspace(BETA_MIN / N_DISC, BETA_MAX / N_DISC, N_DISC).float().to(self.device)
        self.atol, self.rtol = 1e-3, 1e-3
        self.method = "euler"
        self._calls = 0
        print(f"method: {self.method}, atol: {self.atol}, rtol: {self.rtol}, step_size: {self.args.step_size}")

    def image_editing_sample(self, img, bs_id=0, tag=None, noise=None, nhwc=False):
        """nhwc=True (extension): `img` and the result are the NHWC state of the loop (diffpure_amd.adv_model)."""
        assert isinstance(img, torch.Tensor)
        assert img.ndim == 4, img.ndim
        out_dir = _common.out_dir_for(self.args, bs_id, tag)
        log = bs_id < 2 and out_dir is not None
        need_grad = img.requires_grad and torch.is_grad_enabled()
        if need_grad and getattr(self.args, "shard_batch", False):
            raise NotImplementedError("gradients through a batch-sharded purification call: run the attack per rank")
        with torch.set_grad_enabled(need_grad):
            x0 = img.to(self.device)
            if log:
                os.makedirs(out_dir, exist_ok=True)
                _common.save_image(_common.as_nchw(x0, nhwc), os.path.join(out_dir, "original_input.png"))
            seed = int(getattr(self.args, "seed", 0) or 0)
            step = float(self.args.step_size)
            xs = []
            for it in range(self.args.sample_step):
                inj = noise
                if inj is None and getattr(self.args, "fix_rand", False):
                    # one fixed noise image repeated over the batch (reference :202-207)
                    g = torch.Generator().manual_seed(int(self.args.seed))
                    e1 = torch.randn((1,) + tuple(_common.as_nchw_shape(x0.shape, nhwc)[1:]), generator=g)
                    inj = dict(e=e1.repeat(x0.shape[0], 1, 1, 1), z=[])
                call_seed = seed + 1000003 * self._calls
                self._calls += 1

                def run(xl, sample0, inj=inj, call_seed=call_seed):
                    loc = inj if inj is None else dict(e=inj["e"][sample0:sample0 + xl.shape[0]], z=[])
                    if need_grad:
                        return _OdePurify.apply(xl, self, self.args.t, step, loc, call_seed, sample0, nhwc)
                    return self.purifier.ode(xl, self.args.t, step, noise=loc, seed=call_seed, sample0=sample0, nhwc=nhwc)

                x0 = ddist.sharded_purify(run, x0) if getattr(self.args, "shard_batch", False) else run(x0, 0)
                if log:
                    _common.save_image(_common.as_nchw(x0, nhwc), os.path.join(out_dir, f"samples_{it}.png"))
                xs.append(x0)
            return torch.cat(xs, dim=0)
