"""Drop-in for /root/reference/runners/diffpure_sde.py: `RevGuidedDiffusion(args, config, device)`
with `.image_editing_sample(img, bs_id=0, tag=None)` (reference :150-247), on the MI355X engine.

Differences that are deliberate:
  * the reverse VP-SDE is integrated by this repo's own fixed-step Euler-Maruyama loop
    (diffpure_amd/sde.py) instead of torchsde.sdeint_adjoint - same float32 clock, same update;
  * noise comes from an in-kernel Philox stream keyed by (args.seed, global sample index, step)
    unless `noise=` is injected (torchsde's BrownianInterval stream is not reproducible anyway);
    `args.use_bm` therefore changes nothing;
  * `args.dt` (default 1e-3 = torchsde's default) is exposed;
  * under the driver's nn.DataParallel every replica purifies on the GPU its input slice lives on, with an engine
    that is built once per device and stays resident (runners/_common.py::EnginePool);
  * `args.shard_batch` (extension): one process per GPU, the batch sharded over the ranks, differentiable.
`args.rand_t` follows upstream exactly: it randomises the forward-DIFFUSION level only; the reverse SDE is always
integrated over t' in [1 - args.t/1000, 1 - 1e-5] (reference :218-229).
"""
import os

import numpy as np
import torch

from diffpure_amd.sde import BETA_MAX, BETA_MIN, N_DISC

from . import _common


class _SdePurify(torch.autograd.Function):
    """purified = EM-solve(diffuse(img)); backward = the stochastic adjoint on the HIP engine with the SAME
    (regenerated) Brownian path - what torchsde.sdeint_adjoint provides upstream (reference :236-238)."""

    @staticmethod
    def forward(ctx, img, pur, t, dt, noise, seed, sample0, nhwc=False, t_diffuse=None):
        with torch.no_grad():
            out = pur.sde(img, t, dt, noise=noise, seed=seed, sample0=sample0, nhwc=nhwc, t_diffuse=t_diffuse)
        ctx.pur, ctx.cfg = pur, (t, dt, noise, seed, sample0, nhwc, t if t_diffuse is None else t_diffuse)
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (out,) = ctx.saved_tensors
        t, dt, noise, seed, sample0, nhwc, t_diffuse = ctx.cfg
        with torch.no_grad():
            a = ctx.pur.sde_vjp(out, grad_out, t, dt, noise=noise, seed=seed, sample0=sample0, nhwc=nhwc)
            a = a * ctx.pur.diffuse_scale(t_diffuse)
        return a, None, None, None, None, None, None, None, None


class RevGuidedDiffusion(_common.PooledRunner, torch.nn.Module):
    def __init__(self, args, config, device=None):
        super().__init__()
        self.args = args
        self.config = config
        if device is None:
            device = torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")
        self.device = torch.device(device)

        self._pool = _common.EnginePool(lambda dev: _common.build_purifier(args, config, dev), self.device)
        self.purifier = self._pool.get(self.device)
        self.model = self.purifier.net
        self.img_shape = self.purifier.img_shape
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
        pur = self._pool.for_input(img)          # DataParallel replica: the engine of the GPU this slice lives on
        with self._pool.lock(pur.device), torch.set_grad_enabled(need_grad):
            x0 = img.to(pur.device)
            if log:
                os.makedirs(out_dir, exist_ok=True)
                _common.save_image(_common.as_nchw(x0, nhwc), os.path.join(out_dir, "original_input.png"))
            seed = int(getattr(self.args, "seed", 0) or 0)
            dt = float(getattr(self.args, "dt", 1e-3) or 1e-3)
            xs = []
            for it in range(self.args.sample_step):
                call_seed = seed + 1000003 * self._pool.next_call(pur.device)
                t_diffuse = self.args.t
                if self.args.rand_t:
                    # upstream draws from numpy's global RNG (:220); when the call is sharded over ranks every rank must
                    # draw the SAME level, so the draw is then a function of the shared call seed
                    rng = np.random.RandomState(call_seed & 0x7FFFFFFF) if getattr(self.args, "shard_batch", False) else np.random
                    t_diffuse = self.args.t + int(rng.randint(-self.args.t_delta, self.args.t_delta))
                    print(f"total_noise_levels: {t_diffuse}")

                def run(xl, sample0, t_diffuse=t_diffuse, call_seed=call_seed):
                    if need_grad:
                        return _SdePurify.apply(xl, pur, self.args.t, dt, noise, call_seed, sample0, nhwc, t_diffuse)
                    return pur.sde(xl, self.args.t, dt, noise=noise, seed=call_seed, sample0=sample0, nhwc=nhwc,
                                   t_diffuse=t_diffuse)

                x0 = _common.dispatch(self.args, run, x0, self._pool.replica_offset(pur.device))
                if log:
                    _common.save_image(_common.as_nchw(x0, nhwc), os.path.join(out_dir, f"samples_{it}.png"))
                xs.append(x0)
            return torch.cat(xs, dim=0)
