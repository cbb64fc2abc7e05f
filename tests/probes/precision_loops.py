"""GPU probe: purified-pixel error of every arithmetic mode of the h2 convolution path over the FULL loops
north_star names (256x256 guided UNet, 100 EM steps, dt=1e-3; CIFAR NCSN++ 100 steps), against
  (a) the exact fp32-input engine (precision="f32") on the same Philox noise, and
  (b) the reference-module golden (tests/golden/*_loop100.pt) when present.
Writes gpurun_out/precision_loops.json.   python tests/probes/precision_loops.py [guided] [ncsnpp]
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from bench import IMAGENET_CFG, CIFAR_CFG  # noqa: E402
from diffpure_amd import guided_unet, ncsnpp, synth  # noqa: E402
from diffpure_amd.sde import Purifier  # noqa: E402

DEV = "cuda:0"
MODES = ["f32", "f16x3", "f16x2", "f16x2w", "f16"]


def run(kind, out):
    gname = os.path.join(ROOT, "tests", "golden", "guided_loop100.pt" if kind == "guided" else "ncsnpp_loop100.pt")
    gold = torch.load(gname, map_location="cpu", weights_only=False) if os.path.exists(gname) else None
    if kind == "guided":
        cfg = guided_unet.parse_config(IMAGENET_CFG)
        sd = synth.synth_state_dict(guided_unet.param_shapes(cfg), 1234)
        mk = lambda prec: guided_unet.GuidedUNet(cfg, DEV, prec).load_state_dict(sd)
        x0 = gold["x0"] if gold else torch.rand(2, 3, 256, 256, generator=torch.Generator().manual_seed(99)) * 2 - 1
        tb = 16
    else:
        cfg = ncsnpp.parse_config(CIFAR_CFG)
        sd = synth.synth_state_dict(ncsnpp.param_shapes(cfg), 1234)
        mk = lambda prec: ncsnpp.NCSNpp(cfg, DEV, prec).load_state_dict(sd)
        x0 = gold["x0"] if gold else torch.rand(4, 3, 32, 32, generator=torch.Generator().manual_seed(98)) * 2 - 1
        tb = 256
    ref = None
    xt = (torch.rand(tb, 3, x0.shape[2], x0.shape[3]) * 2 - 1).to(DEV)
    for mode in MODES:
        net = mk(mode)
        pur = Purifier(net, kind, DEV)
        snaps = {}
        y = pur.sde(x0, 100, 1e-3, seed=1234, sample0=0).cpu()
        for n in (10, 25):   # partial loops: error growth (the schedule of the first n steps is the same)
            pass
        rec = dict(kind=kind, mode=mode)
        if mode == "f32":
            ref = y
        else:
            rec["max_abs_vs_f32_engine"] = (y - ref).abs().max().item()
            rec["mean_abs_vs_f32_engine"] = (y - ref).abs().mean().item()
        if gold is not None:
            rec["max_abs_vs_reference_golden"] = (y - gold["out"]).abs().max().item()
            rec["mean_abs_vs_reference_golden"] = (y - gold["out"]).abs().mean().item()
        # speed: 5 EM steps at a chip-filling batch
        if mode != "f32" or kind != "guided":
            pur.sde(xt, 5, 1e-3, seed=1, sample0=0)
            torch.cuda.synchronize()
            t0 = time.time()
            pur.sde(xt, 5, 1e-3, seed=1, sample0=0)
            torch.cuda.synchronize()
            rec["ms_per_unet_call_at_batch"] = dict(batch=tb, ms=(time.time() - t0) / 5 * 1e3)
        print(json.dumps(rec), flush=True)
        out.append(rec)
        del net, pur
        torch.cuda.empty_cache()


def main():
    what = sys.argv[1:] or ["ncsnpp", "guided"]
    out = []
    for k in what:
        run(k, out)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "precision_loops.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
