"""Golden vectors for `fir: True` FROM THE REFERENCE ITSELF.

Run in the build container only (needs /root/reference):   python tests/golden/make_golden_fir.py

score_sde/op/upfirdn2d.py JIT-compiles a CUDA extension when imported, so the module is not imported: its pure-PyTorch
`upfirdn2d_native` (the CPU branch of the reference's own `upfirdn2d`, :150-211) is lifted out of the reference source AT
GENERATION TIME (ast, nothing is copied into this repository) and installed as `score_sde.op.upfirdn2d`; everything else
(up_or_down_sampling.upsample_2d / downsample_2d, ResnetBlockBigGANpp, NCSNpp) is the reference's own code.
  fir_ops.pt          upsample_2d / downsample_2d of a random tensor with fir_kernel [1, 3, 3, 1], and torch.autograd's
                      input gradients of both for seeded cotangents (dy_up -> dx_up, dy_down -> dx_down)
  ncsnpp_fir_small.pt the small NCSN++ of make_golden.py with fir: True, forward, B=2; and the input gradient of that
                      forward for a seeded cotangent (cot -> vjp), torch.autograd through the reference module"""
import ast
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import make_golden as mg  # noqa: E402


def reference_upfirdn2d():
    src = open(os.path.join(mg.REF, "score_sde/op/upfirdn2d.py")).read()
    fn = next(n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "upfirdn2d_native")
    ns = {"torch": torch, "F": F}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "upfirdn2d_native", "exec"), ns)
    native = ns["upfirdn2d_native"]

    def upfirdn2d(input, kernel, up=1, down=1, pad=(0, 0)):     # the CPU branch of the reference's wrapper (:150-164)
        return native(input, kernel, up, up, down, down, pad[0], pad[1], pad[0], pad[1])

    return upfirdn2d


def main():
    mg.import_reference()
    op = types.ModuleType("score_sde.op")
    op.upfirdn2d = reference_upfirdn2d()
    sys.modules["score_sde.op"] = op
    from score_sde.models import up_or_down_sampling as uds
    from score_sde.models import utils as mutils
    torch.manual_seed(0)
    g = torch.Generator().manual_seed(31)
    x = torch.randn(2, 8, 10, 6, generator=g)
    ops_g = dict(x=x, k=(1, 3, 3, 1), up=uds.upsample_2d(x, (1, 3, 3, 1), factor=2), down=uds.downsample_2d(x, (1, 3, 3, 1), factor=2))
    cfg = mg.small_ncsnpp_cfg()
    cfg["model"]["fir"] = True
    mod = mg.load_synth(mutils.create_model(mg.d2n(cfg)), 1234)
    xn = torch.rand(2, 3, 16, 16, generator=g) * 2 - 1
    labels = torch.tensor([0.37 * 999, 0.05 * 999])
    with torch.no_grad():
        out = mod(xn, labels)
    # input gradients (drawn AFTER everything above so that the forward fields keep their values)
    for name, fn in (("up", uds.upsample_2d), ("down", uds.downsample_2d)):
        dy = torch.randn(ops_g[name].shape, generator=g)
        xr = x.clone().requires_grad_(True)
        (dx,) = torch.autograd.grad(fn(xr, (1, 3, 3, 1), factor=2), xr, dy)
        ops_g["dy_" + name], ops_g["dx_" + name] = dy, dx
    torch.save(ops_g, os.path.join(HERE, "fir_ops.pt"))
    cot = torch.randn(out.shape, generator=g)
    xr = xn.clone().requires_grad_(True)
    (vjp,) = torch.autograd.grad(mod(xr, labels), xr, cot)
    torch.save(dict(cfg=cfg, seed=1234, x=xn, labels=labels, out=out, cot=cot, vjp=vjp), os.path.join(HERE, "ncsnpp_fir_small.pt"))
    print("fir goldens written; out abs-mean", float(out.abs().mean()), "vjp abs-mean", float(vjp.abs().mean()))


if __name__ == "__main__":
    main()
