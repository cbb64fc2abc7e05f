"""Drop-in for /root/reference/runners/diffpure_ldsde.py: `LDGuidedDiffusion(args, config, device)` - Langevin
dynamics anchored at the input, score frozen at noise level 1e-2 (`--diffusion_type ldsde`, eval_sde_adv.py:50-51) -
with `.image_editing_sample(img, bs_id=0, tag=None)` (reference :198-252) on the MI355X engine.  Reads
args.{t, sigma2, lambda_ld, eta, sample_step, score_type, seed, log_dir}.  torch.no_grad: the attack scripts that use
this runner differentiate it through `sdeint_adjoint` upstream; here the backward is the stochastic adjoint on the HIP engine
(gradient through the initial state, exactly what torchsde returns for a module whose anchor is a plain tensor attribute)."""
import os

import torch

from diffpure_amd.sde import BETA_MAX, BETA_MIN, N_DISC

from . import _common


class _LdPurify(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, anchor, pur, cfg, noise, seed, sample0, nhwc):
        """img: the loop's initial state; anchor: x_init of the Langevin drift (the ORIGINAL input of the call - a plain
        tensor attribute upstream, diffpure_ldsde.py:63,:212-214 - not differentiated, as upstream)."""
        with torch.no_grad():
            out = pur.ldsde(img, *cfg, noise=noise, seed=seed, sample0=sample0, nhwc=nhwc, x_init=anchor)
        ctx.pur, ctx.cfg = pur, (cfg, noise, seed, sample0, nhwc)
        ctx.save_for_backward(out, anchor.detach())
        return out

    @staticmethod
    def backward(ctx, grad_out):
        out, anchor = ctx.saved_tensors
        cfg, noise, seed, sample0, nhwc = ctx.cfg
        t, sigma2, lam, eta, dt = cfg
        with torch.no_grad():
            a = ctx.pur.ldsde_vjp(out, grad_out, anchor, t, sigma2, lam, eta, dt=dt, noise=noise, seed=seed,
                                  sample0=sample0, nhwc=nhwc)
        return a, None, None, None, None, None, None, None


class LDGuidedDiffusion(_common.PooledRunner, torch.nn.Module):
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
        self.args_dict = {"method": "euler", "adaptive": False, "dt": 1e-2}
        self._calls = 0
        print(f"use_bm: {args.use_bm}")
        print(f"args_dict: {self.args_dict}")

    def image_editing_sample(self, img, bs_id=0, tag=None, noise=None, nhwc=False):
        assert isinstance(img, torch.Tensor)
        assert img.ndim == 4, img.ndim
        need_grad = img.requires_grad and torch.is_grad_enabled()
        pur = self._pool.for_input(img)          # DataParallel replica: the engine of the GPU this slice lives on
        with self._pool.lock(pur.device), torch.set_grad_enabled(need_grad):
            out_dir = _common.out_dir_for(self.args, bs_id, tag)
            log = bs_id < 2 and out_dir is not None
            x0 = img.to(pur.device)
            anchor = x0.detach()           # x_init: every repeat is anchored at the ORIGINAL input (reference :212-214)
            if log:
                os.makedirs(out_dir, exist_ok=True)
                _common.save_image(_common.as_nchw(x0, nhwc), os.path.join(out_dir, "original_input.png"))
            print(f"sigma2: {self.args.sigma2}, lambda_ld: {self.args.lambda_ld}, eta: {self.args.eta}")
            seed = int(getattr(self.args, "seed", 0) or 0)
            cfg = (self.args.t, float(self.args.sigma2), float(self.args.lambda_ld), float(self.args.eta), self.args_dict["dt"])
            xs = []
            for it in range(self.args.sample_step):
                call_seed = seed + 1000003 * self._pool.next_call(pur.device)

                def run(xl, sample0, call_seed=call_seed):
                    lo = sample0 if getattr(self.args, "shard_batch", False) else 0      # this shard's rows of the anchor
                    al = anchor[lo:lo + xl.shape[0]]
                    if need_grad:
                        return _LdPurify.apply(xl, al, pur, cfg, noise, call_seed, sample0, nhwc)
                    t, sigma2, lam, eta, dt = cfg
                    return pur.ldsde(xl, t, sigma2, lam, eta, dt=dt, noise=noise, seed=call_seed, sample0=sample0, nhwc=nhwc,
                                     x_init=al)

                x0 = _common.dispatch(self.args, run, x0, self._pool.replica_offset(pur.device))
                if log:
                    _common.save_image(_common.as_nchw(x0, nhwc), os.path.join(out_dir, f"samples_{it}.png"))
                xs.append(x0)
            return torch.cat(xs, dim=0)
