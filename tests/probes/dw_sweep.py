"""Two-workgroups-per-CU convolution (igemm_h2_dw.hip) against the one-wave-per-SIMD kernel (igemm_h2_sw.hip), tuning aid:
    python tests/probes/dw_sweep.py [--batch B] [--ablate]
Per shape: sw, then dw for (activation-ring depth, start-up stagger) combinations, each plain / with a residual / with fp16
output, in TFLOP/s (algorithmic); checks dw == sw bit for bit.  --ablate (needs tests/probes/build_ablate.py): the timing
ablations of dw (m1 no DMA, m2 no waits / barriers, m4 no ds_reads, m7 none of the three, m8 no epilogue; WRONG RESULTS)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
ABL = "--ablate" in sys.argv
if ABL:
    from diffpure_amd import _lib
    _lib.LIB_PATH = os.path.join(ROOT, "diffpure_amd", "csrc", "libdiffpure_hip_ablate.so")
from diffpure_amd import ops  # noqa: E402

DEV = "cuda:0"
SHAPES = [(256, 256, 256), (256, 512, 256), (128, 256, 256), (128, 512, 512), (64, 512, 512), (32, 512, 512), (16, 1024, 1024)]


def timeit(fn, iters):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    B = int(sys.argv[sys.argv.index("--batch") + 1]) if "--batch" in sys.argv else 64
    combos = [(4, 0)] if "--dw" in sys.argv or ABL else []
    for (H, ci, co) in SHAPES:
        x = torch.randn(B, H, H, ci)
        w = torch.randn(co, ci, 3, 3) * (1.0 / (9 * ci)) ** 0.5
        wh = ops.order_conv_weight_w16(w).half().to(DEV)
        xh = torch.nn.functional.pad(x, (0, 0, 1, 1, 1, 1)).half().contiguous().to(DEV)
        bias = torch.randn(co, device=DEV)
        rs = torch.randn(B, H, H, co, device=DEV)
        flop = 2.0 * B * H * H * co * 9 * ci
        iters = max(3, min(20, int(1.5e12 / flop)))
        fns = dict(plain=lambda: ops.conv2d_h2(xh, wh, co, 3, bias=bias, colstats=True, w_fmt=1),
                   res=lambda: ops.conv2d_h2(xh, wh, co, 3, bias=bias, res=rs, colstats=True, w_fmt=1),
                   f16=lambda: ops.conv2d_h2(xh, wh, co, 3, bias=bias, colstats=True, w_fmt=1, out_f16=True))
        tf = lambda fn: flop / timeit(fn, iters) / 1e9
        line = f"{H:4d} {ci:5d}->{co:4d} B={B} |"
        ops.set_tuning("DP_H2_DW", 0)
        base = {k: fn() for k, fn in fns.items()}
        line += " sw " + " ".join(f"{k} {tf(fn):5.0f}" for k, fn in fns.items()) + " |"
        line += " sw again " + " ".join(f"{tf(fn):5.0f}" for fn in fns.values()) + " |"
        for adepth in (3, 4):                   # 8-wave form: one workgroup per CU, 256x256 tile, two free-running waves per SIMD
            ops.set_tuning("DP_H2_DW", 8)
            ops.set_tuning("DP_H2_DW_ADEPTH", adepth)
            ok = all(torch.equal(fn().t, base[k].t) and torch.equal(fn().cols.buf, base[k].cols.buf) for k, fn in fns.items())
            line += f" dw8 a{adepth} " + " ".join(f"{tf(fn):5.0f}" for fn in fns.values()) + (" [ok] |" if ok else " [DIFF] |")
        ops.set_tuning("DP_H2_DW", 2)
        ops.set_tuning("DP_H2_DW_MINROUNDS", 0)
        for adepth, stag in combos:
            ops.set_tuning("DP_H2_DW_ADEPTH", adepth)
            ops.set_tuning("DP_H2_DW_STAGGER", stag)
            ok = True
            for k, fn in fns.items():
                y = fn()
                ok = ok and torch.equal(y.t, base[k].t) and torch.equal(y.cols.buf, base[k].cols.buf)
            line += f" dw a{adepth} s{stag} " + " ".join(f"{tf(fn):5.0f}" for fn in fns.values()) + (" [ok] |" if ok else " [DIFF] |")
        if ABL:
            ops.set_tuning("DP_H2_DW", 0)
            line += " sw abl:"
            for m in (1, 8, 16, 4, 7):          # no DMA / no activation DMA / no weight DMA / no ds_reads / none of them
                os.environ["DP_H2_SW_MODE"] = str(m)
                line += f" m{m} {tf(fns['plain']):5.0f}"
            os.environ["DP_H2_SW_MODE"] = "0"
            ops.set_tuning("DP_H2_DW", 8)
            ops.set_tuning("DP_H2_DW_ADEPTH", 4)
            line += " dw8 abl:"
            for m in (1, 16, 32, 4, 7, 8):
                os.environ["DP_H2_DW_MODE"] = str(m)
                line += f" m{m} {tf(fns['plain']):5.0f}"
            os.environ["DP_H2_DW_MODE"] = "0"
            ops.set_tuning("DP_H2_DW", 2)
            ops.set_tuning("DP_H2_DW_ADEPTH", 4)
            for stag in (0,):
                ops.set_tuning("DP_H2_DW_STAGGER", stag)
                line += f" abl s{stag}:"
                for m in (1, 16, 32, 4, 7, 8):
                    os.environ["DP_H2_DW_MODE"] = str(m)
                    line += f" m{m} {tf(fns['plain']):5.0f}"
                os.environ["DP_H2_DW_MODE"] = "0"
            line += " |"
        print(line, flush=True)
        del x, w, wh, xh, rs, base
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
