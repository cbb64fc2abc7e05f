"""Probe (GPU): where does a network call stop being independent of the batch?  Runs one NCSN++ (or guided UNet) forward on the
first `small` samples alone and leading a batch of `big`, records a checksum of the first `small` samples of EVERY operator output
(in call order: the sequence is the same for both, all dispatch rules being shape-only) and prints the first operators that differ.
    python tests/probes/batch_invariance_trace.py [ncsnpp|guided] [--small 4] [--big 256] [--precision f16sr]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden  # noqa: E402
from diffpure_amd import ops  # noqa: E402
from diffpure_amd.synth import synth_state_dict  # noqa: E402

DEV = "cuda:0"


def arg(name, default):
    return type(default)(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else "ncsnpp"
    small, big, prec = arg("--small", 4), arg("--big", 256), arg("--precision", "f16sr")
    trace, cur = {}, []
    names = ["conv2d", "conv2d_h2", "group_norm", "group_norm_stats", "group_norm_f16in", "resample", "to_h2", "attention", "attention_fused"]
    for nm in names:
        real = getattr(ops, nm)

        def wrap(*a, _real=real, _nm=nm, **kw):
            y = _real(*a, **kw)
            outs = y if isinstance(y, tuple) else (y,)
            for o in outs:
                t = ops.tensor_of(o)
                if isinstance(t, torch.Tensor):
                    v = t[:small].float()
                    cur.append((_nm, tuple(t.shape[1:]), str(t.dtype), v.double().sum().item(), v.abs().double().sum().item(), v.cpu()))
            return y

        setattr(ops, nm, wrap)
    if kind == "ncsnpp":
        from diffpure_amd import ncsnpp as pn
        g = load_golden("ncsnpp_full.pt")
        cfg = pn.parse_config(g["cfg"])
        net = pn.NCSNpp(cfg, DEV, prec).load_state_dict(synth_state_dict(pn.param_shapes(cfg), g["seed"]))
        hw = 32
    else:
        from diffpure_amd import guided_unet as pg
        g = load_golden("guided_full.pt")
        cfg = pg.parse_config(g["cfg"])
        net = pg.GuidedUNet(cfg, DEV, prec).load_state_dict(synth_state_dict(pg.param_shapes(cfg), g["seed"]))
        hw = 256
    x = (torch.rand(big, hw, hw, 3, generator=torch.Generator().manual_seed(1)) * 2 - 1).to(DEV)
    t = torch.tensor([100.0], device=DEV)
    for label, xb in (("small", x[:small].contiguous()), ("big", x)):
        cur.clear()
        if hasattr(net, "reround"):
            net.reround(0)
        net.forward(xb, t)
        torch.cuda.synchronize()
        trace[label] = list(cur)
    a, b = trace["small"], trace["big"]
    print(f"{kind} [{prec}] {len(a)} operator outputs; batch {small} alone vs leading a batch of {big}")
    assert len(a) == len(b)
    shown = 0
    for i, (ra, rb) in enumerate(zip(a, b)):
        same = torch.equal(ra[5], rb[5])
        if not same:
            d = (ra[5] - rb[5]).abs()
            print(f"  #{i:4d} {ra[0]:18s} {ra[1]} {ra[2]}: DIFFERS max {d.max().item():.3e} in {(d > 0).float().mean().item():.2e} of the elements")
            shown += 1
            if shown >= 12:
                break
    if not shown:
        print("  every operator output of the leading samples is bit-identical")


if __name__ == "__main__":
    main()
