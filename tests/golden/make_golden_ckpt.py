"""Checkpoint-FORMAT golden: a `checkpoint_8.pth`-shaped file written and read back BY THE
REFERENCE'S OWN CODE, plus the forward output of the model the reference ends up with.

Run in the build container only (needs /root/reference):   python tests/golden/make_golden_ckpt.py

The real score_sde checkpoint was saved from an nn.DataParallel model: its 'model' entry carries `module.`-prefixed keys,
which the reference's `restore_checkpoint(..., strict=False)` (runners/diffpure_sde.py:42-47) silently ignores, and the
weights that take effect are the EMA shadow parameters copied over `model.parameters()` in registration order
(`ema.copy_to`, :182; score_sde/models/ema.py:61-72).  This script reproduces that situation on the small NCSN++:
  tests/golden/ckpt/checkpoint_8.pth   {'optimizer', 'model' (module.-prefixed, DIFFERENT weights), 'ema', 'step'}
  tests/golden/ncsnpp_ckpt.pt          cfg, x, labels, out = forward of the reference module after ITS restore path
The engine must load the same file through diffpure_amd.factory.build_ncsnpp and reproduce `out`."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import torch  # noqa: E402

import make_golden as mg  # noqa: E402
from diffpure_amd.synth import synth_state_dict  # noqa: E402


def main():
    mg.import_reference()
    ref_sde = mg.ref_module("ref_diffpure_sde", "runners/diffpure_sde.py")
    from score_sde.losses import get_optimizer
    from score_sde.models import utils as mutils
    from score_sde.models.ema import ExponentialMovingAverage
    cfg = mg.small_ncsnpp_cfg()
    config = mg.d2n(cfg)
    torch.manual_seed(0)

    # --- a "training run": raw weights (seed 77) in the model, EMA weights (seed 1234) in the shadow list
    trained = mutils.create_model(config)
    mg.load_synth(trained, 77)
    ema = ExponentialMovingAverage(trained.parameters(), decay=config.model.ema_rate)
    names = [k for k, _ in trained.named_parameters()]
    eff = synth_state_dict({k: v.shape for k, v in trained.state_dict().items()}, 1234)
    ema.shadow_params = [eff[k].clone() for k in names]
    ema.num_updates = 1300001
    opt = get_optimizer(config, trained.parameters())
    os.makedirs(os.path.join(HERE, "ckpt"), exist_ok=True)
    path = os.path.join(HERE, "ckpt", "checkpoint_8.pth")
    torch.save(dict(optimizer=opt.state_dict(), model={"module." + k: v for k, v in trained.state_dict().items()},
                    ema=ema.state_dict(), step=1300001), path)

    # --- the reference's own restore path (runners/diffpure_sde.py:172-182) on a fresh model
    model = mutils.create_model(config)
    optimizer = get_optimizer(config, model.parameters())
    ema2 = ExponentialMovingAverage(model.parameters(), decay=config.model.ema_rate)
    state = dict(step=0, optimizer=optimizer, model=model, ema=ema2)
    ref_sde.restore_checkpoint(path, state, "cpu")
    ema2.copy_to(model.parameters())
    model.eval()
    g = torch.Generator().manual_seed(4242)
    x = torch.rand(2, 3, 16, 16, generator=g) * 2 - 1
    labels = torch.tensor([0.61 * 999, 0.03 * 999])
    with torch.no_grad():
        out = model(x, labels)
    torch.save(dict(cfg=cfg, x=x, labels=labels, out=out, n_params=len(names), step=state["step"]), os.path.join(HERE, "ncsnpp_ckpt.pt"))
    print("checkpoint", os.path.getsize(path), "bytes;", len(names), "parameter tensors; out abs-mean", float(out.abs().mean()))


if __name__ == "__main__":
    main()
