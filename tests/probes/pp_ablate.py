"""Timing ablations of the fp16-operand ping-pong convolution (tuning aid, not a test):
    python tests/probes/pp_ablate.py [--batch B]
For every shape: schedule 0 (four phases per k-tile) and 1 (two phases), each in modes 0 (as shipped), 2 (no operand
traffic after k-tile 0), 8 (no ds_reads), 16 (no DMA), 256 (no epilogue) - modes other than 0 give WRONG RESULTS and exist
to attribute time.  Also checks schedule 1 == schedule 0 == the 128x128 kernel bit for bit."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from diffpure_amd import ops  # noqa: E402

DEV = "cuda:0"
SHAPES = [(256, 256, 256), (256, 512, 256), (128, 256, 256), (64, 512, 512), (32, 512, 512), (16, 1024, 1024)]


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
    B = int(sys.argv[sys.argv.index("--batch") + 1]) if "--batch" in sys.argv else 16
    passes = int(sys.argv[sys.argv.index("--passes") + 1]) if "--passes" in sys.argv else 2
    w16 = "--w16" in sys.argv          # plain fp16 weights, one pass ("f16" / "f16sr")
    for (H, ci, co) in SHAPES:
        x = torch.randn(B, H, H, ci)
        w = torch.randn(co, ci, 3, 3) * (1.0 / (9 * ci)) ** 0.5
        wh = ops.order_conv_weight_w16(w).half().to(DEV) if w16 else ops.pack_conv_weight_h2(w, DEV)
        xh = torch.nn.functional.pad(x, (0, 0, 1, 1, 1, 1)).half().contiguous().to(DEV)
        bias = torch.randn(co, device=DEV)
        flop = 2.0 * B * H * H * co * 9 * ci
        iters = max(3, min(30, int(2e12 / flop)))
        fn = (lambda: ops.conv2d_h2(xh, wh, co, 3, bias=bias, colstats=True, w_fmt=1)) if w16 else \
             (lambda: ops.conv2d_h2(xh, wh, co, 3, bias=bias, colstats=True, passes=passes))
        os.environ["DP_H2_PP"] = "0"
        base = fn()
        tbase = timeit(fn, iters)
        os.environ["DP_H2_PP"] = "1"
        line = f"{H:4d} {ci:5d}->{co:4d} B={B} | 128x128 {flop / tbase / 1e9:6.0f} TF |"
        if w16:     # halo-tile variant against the per-tap kernel
            os.environ["DP_H2_PP_MODE"] = "0"
            tt = {}
            for halo in ("0", "1"):
                os.environ["DP_H2_HALO"] = halo
                y = fn()
                ok = torch.equal(y.t, base.t) and torch.equal(y.cols.buf, base.cols.buf)
                tt[halo] = (flop / timeit(fn, iters) / 1e9, ok)
            line += f" per-tap {tt['0'][0]:5.0f} halo {tt['1'][0]:5.0f} [{'ok' if tt['1'][1] and tt['0'][1] else 'DIFF'}] |"
            os.environ["DP_H2_HALO"] = "0"
            os.environ["DP_H2_SW"] = "1"          # one wave per SIMD, software-pipelined (igemm_h2_sw.hip)
            os.environ["DP_H2_SW_VAR"] = "0"
            y = fn()
            ok = torch.equal(y.t, base.t) and torch.equal(y.cols.buf, base.cols.buf)
            line += f" sw {flop / timeit(fn, iters) / 1e9:5.0f} [{'ok' if ok else 'DIFF'}]"
            for m in (1, 2, 3, 4, 7):
                os.environ["DP_H2_SW_MODE"] = str(m)
                line += f" m{m}:{flop / timeit(fn, iters) / 1e9:5.0f}"
            os.environ["DP_H2_SW_MODE"] = "0"
            line += " |"
            os.environ["DP_H2_SW_VAR"] = "1"      # DMA issues spread between the fragment reads
            y = fn()
            ok = torch.equal(y.t, base.t) and torch.equal(y.cols.buf, base.cols.buf)
            line += f" sw.var1 {flop / timeit(fn, iters) / 1e9:5.0f} [{'ok' if ok else 'DIFF'}]"
            for m in (1, 4, 7):
                os.environ["DP_H2_SW_MODE"] = str(m)
                line += f" m{m}:{flop / timeit(fn, iters) / 1e9:5.0f}"
            os.environ["DP_H2_SW_MODE"] = "0"
            # with a residual (the second convolution of a ResBlock) and with fp16 output (the first)
            rs = torch.randn(B, H, H, co, device=DEV)
            fnr = lambda: ops.conv2d_h2(xh, wh, co, 3, bias=bias, res=rs, colstats=True, w_fmt=1)
            fn16 = lambda: ops.conv2d_h2(xh, wh, co, 3, bias=bias, colstats=True, w_fmt=1, out_f16=True)
            line += f" +res {flop / timeit(fnr, iters) / 1e9:5.0f} f16out {flop / timeit(fn16, iters) / 1e9:5.0f} |"
            del rs
            os.environ["DP_H2_SW"] = "0"
        for sched in ((1,) if w16 else (0, 1)):
            os.environ["DP_H2_PP_SCHED"] = str(sched)
            os.environ["DP_H2_PP_MODE"] = "0"
            y = fn()
            same = torch.equal(y.t, base.t) and torch.equal(y.cols.buf, base.cols.buf)
            res = []
            for mode in (0,):
                os.environ["DP_H2_PP_MODE"] = str(mode)
                res.append(f"m{mode}:{flop / timeit(fn, iters) / 1e9:5.0f}")
            line += f" sched{sched} [{'ok' if same else 'DIFF'}] " + " ".join(res) + " |"
        print(line, flush=True)
    for k in ("DP_H2_PP", "DP_H2_PP_SCHED", "DP_H2_PP_MODE", "DP_H2_HALO", "DP_H2_SW", "DP_H2_SW_MODE", "DP_H2_SW_VAR"):
        os.environ.pop(k, None)


if __name__ == "__main__":
    main()
