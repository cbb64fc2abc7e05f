"""Drop-in for /root/reference/runners/diffpure_sde.py: `RevGuidedDiffusion(args, config, device)`
with `.image_editing_sample(img, bs_id=0, tag=None)` (reference :150-247), on the MI355X engine.

Differences that are deliberate:
  * the reverse VP-SDE is integrated by this repo's own fixed-step Euler-Maruyama loop
    (diffpure_amd/sde.py) instead of torchsde.sdeint_adjoint - same float32 clock, same update;
  * noise comes from an in-kernel Philox stream keyed by (args.seed, global sample index, step)
    unless `noise=` is injected (torchsde's BrownianInterval stream is not reproducible anyway);
    `args.use_bm` therefore changes nothing;
  * `args.dt` (default 1e-3 = torchsde's default) is exposed.
"""
import os

import numpy as np
import torch

from diffpure_amd import dist as ddist
from diffpure_amd import factory
from diffpure_amd.sde import BETA_MAX, BETA_MIN, N_DISC, Purifier

from . import _common


class _SdePurify(torch.autograd.Function):
    """purified = EM-solve(diffuse(img)); backward = the stochastic adjoint on the HIP engine with the SAME
    (regenerated) Brownian path - what torchsde.sdeint_adjoint provides upstream (reference :236-238)."""

    @staticmethod
    def forward(ctx, img, runner, t, dt, noise, seed, sample0, nhwc=False):
        with torch.no_grad():
            out = runner.purifier.sde(img, t, dt, noise=noise, seed=seed, sample0=sample0, nhwc=nhwc)
        ctx.runner, ctx.cfg = runner, (t, dt, noise, seed, sample0, nhwc)
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (out,) = ctx.saved_tensors
        t, dt, noise, seed, sample0, nhwc = ctx.cfg
        with torch.no_grad():
            a = ctx.runner.purifier.sde_vjp(out, grad_out, t, dt, noise=noise, seed=seed, sample0=sample0, nhwc=nhwc)
            a = a * ctx.runner.purifier.diffuse_scale(t)
        return a, None, None, None, None, None, None, None


class RevGuidedDiffusion(torch.nn.Module):
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
        self._calls = 0
        print(f"t: {args.t}, rand_t: {args.rand_t}, t_delta: {args.t_delta}")
        print(f"use_bm: {args.use_bm}")

    def image_editing_sample(self, img, bs_id=0, tag=None, noise=None, nhwc=False):
        """nhwc=True (extension): `img` and the result are the NHWC state of the loop - used by
        diffpure_amd.adv_model, whose fused resize kernels read/write that layout directly."""
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
            dt = float(getattr(self.args, "dt", 1e-3) or 1e-3)
            xs = []
            for it in range(self.args.sample_step):
                t = self.args.t
                if self.args.rand_t:
                    t = self.args.t + np.random.randint(-self.args.t_delta, self.args.t_delta)
                    print(f"total_noise_levels: {t}")
                call_seed = seed + 1000003 * self._calls
                self._calls += 1

                def run(xl, sample0, t=t, call_seed=call_seed):
                    if need_grad:
                        return _SdePurify.apply(xl, self, t, dt, noise, call_seed, sample0, nhwc)
                    return self.purifier.sde(xl, t, dt, noise=noise, seed=call_seed, sample0=sample0, nhwc=nhwc)

                x0 = ddist.sharded_purify(run, x0) if getattr(self.args, "shard_batch", False) else run(x0, 0)
                if log:
                    _common.save_image(_common.as_nchw(x0, nhwc), os.path.join(out_dir, f"samples_{it}.png"))
                xs.append(x0)
            return torch.cat(xs, dim=0)
