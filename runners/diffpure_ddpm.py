"""Drop-in for /root/reference/runners/diffpure_ddpm.py: `Diffusion(args, config, device)` - the CelebA-HQ
DDPM purifier (`--diffusion_type celebahq-ddpm`, eval_sde_adv.py:52-53) - with
`.image_editing_sample(img, bs_id=0, tag=None)` (reference :100-142) on the MI355X engine.

Same deliberate differences as the other runners: own fixed loop, Philox noise keyed by (args.seed, global
sample index, step) unless `noise=` is injected, optional batch sharding.  The checkpoint is looked up locally
(no torch.hub download)."""
import os

import torch

from diffpure_amd import factory
from diffpure_amd.sde import CelebaSchedule, Purifier

from . import _common


class Diffusion(_common.PooledRunner, torch.nn.Module):
    def __init__(self, args, config, device=None):
        super().__init__()
        self.args = args
        self.config = config
        if device is None:
            device = torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")
        self.device = torch.device(device)
        print("Loading model")
        if self.config.data.dataset != "CelebA_HQ":
            raise ValueError
        cfgs = {}

        def build(dev):
            net, cfg = factory.build_celeba(args, config, dev)
            cfgs["cfg"] = cfg
            return Purifier(net, "ddpm_celeba", dev)

        self._pool = _common.EnginePool(build, self.device)      # one resident engine per GPU (nn.DataParallel replicas)
        self.purifier = self._pool.get(self.device)
        self.model = self.purifier.net
        cfg = cfgs["cfg"]
        self.img_shape = (cfg["in_channels"], cfg["resolution"], cfg["resolution"])
        self.model_var_type = config.model.var_type
        d = config.diffusion
        self.sched = CelebaSchedule(d.beta_start, d.beta_end, d.num_diffusion_timesteps, self.model_var_type)
        self.betas = self.sched.betas
        self.logvar = self.sched.logvar.numpy()
        self.num_timesteps = self.betas.shape[0]
        self._calls = 0

    def image_editing_sample(self, img=None, bs_id=0, tag=None, noise=None, nhwc=False):
        """nhwc=True (extension): `img` and the result are the NHWC state of the loop (diffpure_amd.adv_model)."""
        assert isinstance(img, torch.Tensor)
        with torch.no_grad():
            assert img.ndim == 4, img.ndim
            out_dir = _common.out_dir_for(self.args, bs_id, tag)
            log = bs_id < 2 and out_dir is not None
            pur = self._pool.for_input(img)
            x0 = img.to(pur.device)
            if log:
                os.makedirs(out_dir, exist_ok=True)
                _common.save_image(_common.as_nchw(x0, nhwc), os.path.join(out_dir, "original_input.png"))
            seed = int(getattr(self.args, "seed", 0) or 0)
            xs = []
            for it in range(self.args.sample_step):
                call_seed = seed + 1000003 * self._pool.next_call(pur.device)

                def run(xl, sample0, call_seed=call_seed):
                    return pur.celeba_ddpm(xl, self.args.t, self.sched, noise=noise, seed=call_seed, sample0=sample0,
                                                     nhwc=nhwc)

                with self._pool.lock(pur.device):
                    x0 = _common.dispatch(self.args, run, x0, self._pool.replica_offset(pur.device))
                if log:
                    _common.save_image(_common.as_nchw(x0, nhwc), os.path.join(out_dir, f"samples_{it}.png"))
                xs.append(x0)
            return torch.cat(xs, dim=0)
