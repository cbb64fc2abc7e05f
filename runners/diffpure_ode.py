"""Drop-in for /root/reference/runners/diffpure_ode.py: `OdeGuidedDiffusion(args, config, device)`
- probability-flow ODE purification with fixed-step Euler (reference :134-249) on the MI355X engine.

`args.fix_rand` is read by the reference (:202) but defined by no argparse;
it is treated as False when absent.
"""
import os

import torch

from diffpure_amd.sde import BETA_MAX, BETA_MIN, N_DISC

from . import _common


class _OdePurify(torch.autograd.Function):
    """purified = ODE-solve(diffuse(img)); backward = the continuous adjoint on the HIP engine
    (what torchdiffeq.odeint_adjoint provides upstream, runners/diffpure_ode.py:229-238)."""

    @staticmethod
    def forward(ctx, img, pur, t, step, noise, seed, sample0, nhwc=False):
        with torch.no_grad():
            out = pur.ode(img, t, step, noise=noise, seed=seed, sample0=sample0, nhwc=nhwc)
        ctx.pur, ctx.t, ctx.step, ctx.nhwc = pur, t, step, nhwc
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (out,) = ctx.saved_tensors
        with torch.no_grad():
            a = ctx.pur.ode_vjp(out, grad_out, ctx.t, ctx.step, nhwc=ctx.nhwc)
            a = a * ctx.pur.diffuse_scale(ctx.t)
        return a, None, None, None, None, None, None, None


class OdeGuidedDiffusion(_common.PooledRunner, torch.nn.Module):
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
        pur = self._pool.for_input(img)          # DataParallel replica: the engine of the GPU this slice lives on
        with self._pool.lock(pur.device), torch.set_grad_enabled(need_grad):
            x0 = img.to(pur.device)
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
                call_seed = seed + 1000003 * self._pool.next_call(pur.device)

                def run(xl, sample0, inj=inj, call_seed=call_seed):
                    lo = sample0 if getattr(self.args, "shard_batch", False) else 0      # this shard's rows of injected noise
                    loc = inj if inj is None else dict(e=inj["e"][lo:lo + xl.shape[0]], z=[])
                    if need_grad:
                        return _OdePurify.apply(xl, pur, self.args.t, step, loc, call_seed, sample0, nhwc)
                    return pur.ode(xl, self.args.t, step, noise=loc, seed=call_seed, sample0=sample0, nhwc=nhwc)

                x0 = _common.dispatch(self.args, run, x0, self._pool.replica_offset(pur.device))
                if log:
                    _common.save_image(_common.as_nchw(x0, nhwc), os.path.join(out_dir, f"samples_{it}.png"))
                xs.append(x0)
            return torch.cat(xs, dim=0)
