"""Drop-in for /root/reference/runners/diffpure_guided.py: `GuidedDiffusion(args, config, device)`
- DDPM ancestral purification (`p_sample` x t, no_grad; reference :17-89) on the MI355X engine."""
import os

import torch

from diffpure_amd import factory
from diffpure_amd.sde import DdpmSchedule, Purifier

from . import _common


class GuidedDiffusion(_common.PooledRunner, torch.nn.Module):
    def __init__(self, args, config, device=None, model_dir="pretrained/guided_diffusion"):
        super().__init__()
        self.args = args
        self.config = config
        if device is None:
            device = torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")
        self.device = torch.device(device)
        mcs = {}

        def build(dev):
            net, mc = factory.build_guided(args, config, dev, model_dir=model_dir)
            mcs["mc"] = mc
            return Purifier(net, "guided", dev)

        self._pool = _common.EnginePool(build, self.device)      # one resident engine per GPU (nn.DataParallel replicas)
        self.purifier = self._pool.get(self.device)
        self.model = self.purifier.net
        mc = mcs["mc"]
        self.diffusion_steps = int(mc.get("diffusion_steps", 1000))
        if str(mc.get("timestep_respacing", "")) not in ("", str(self.diffusion_steps)):
            raise NotImplementedError("timestep_respacing other than the full schedule is not used by DiffPure")
        self.betas = torch.from_numpy(DdpmSchedule(self.diffusion_steps).betas).float().to(self.device)
        self._calls = 0

    def image_editing_sample(self, img, bs_id=0, tag=None, noise=None, nhwc=False):
        """nhwc=True (extension): `img` and the result are the NHWC state of the loop (diffpure_amd.adv_model)."""
        with torch.no_grad():
            assert isinstance(img, torch.Tensor)
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
                    return pur.ddpm(xl, self.args.t, noise=noise, seed=call_seed, sample0=sample0,
                                              diffusion_steps=self.diffusion_steps, nhwc=nhwc)

                with self._pool.lock(pur.device):
                    x0 = _common.dispatch(self.args, run, x0, self._pool.replica_offset(pur.device))
                if log:
                    _common.save_image(_common.as_nchw(x0, nhwc), os.path.join(out_dir, f"samples_{it}.png"))
                xs.append(x0)
            return torch.cat(xs, dim=0)
