"""Per-shape timing of the convolution kernels on the GPU box (tuning aid, not a test).
    python tests/probes/conv_bench.py [cifar|imagenet|all] [--batch B] [--pp]
--pp: A/B of the 256x256 ping-pong variant (DP_H2_PP=1) against the default tiles on the f16x3 kernel only,
interleaved rounds in one process, outputs compared bit for bit.
Prints algorithmic TFLOP/s of the fp32-MFMA kernel and the f16x3 kernel for the 3x3 shapes of the
two UNets.  Timing: torch.cuda.Event on torch's current stream = the stream the kernels launch on."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from diffpure_amd import ops  # noqa: E402

DEV = "cuda:0"
SHAPES = {
    # (H, Cin, Cout, count per forward)
    "cifar": [(32, 128, 128, 34), (32, 256, 128, 9), (32, 384, 128, 1), (16, 256, 256, 33), (16, 128, 256, 1),
              (16, 512, 256, 8), (16, 384, 256, 1), (8, 256, 256, 34), (8, 512, 256, 9), (4, 256, 256, 38), (4, 512, 256, 9)],
    "imagenet": [(256, 256, 256, 9), (256, 512, 256, 3), (128, 256, 256, 9), (128, 512, 256, 2), (128, 768, 256, 1),
                 (64, 512, 512, 8), (64, 1024, 512, 2), (64, 256, 512, 1), (32, 512, 512, 9), (32, 1024, 512, 2),
                 (32, 1536, 512, 1), (16, 1024, 1024, 8), (16, 2048, 1024, 2), (16, 512, 1024, 1), (8, 1024, 1024, 13),
                 (8, 2048, 1024, 3)],
}


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


def main_pp(which, batch):
    import statistics
    for name in (["cifar", "imagenet"] if which == "all" else [which]):
        B = batch[name]
        tot = {"a": 0.0, "b": 0.0, "flop": 0.0}
        print(f"== {name} B={B}: H Cin Cout | M | base TF | pp TF | ms base / pp (x count) | bit-equal")
        for (H, ci, co, cnt) in SHAPES[name]:
            x = torch.randn(B, H, H, ci, device=DEV)
            w = torch.randn(co, ci, 3, 3) * (1.0 / (9 * ci)) ** 0.5
            wh = ops.pack_conv_weight_h2(w, DEV)
            xh = ops.pack_h2(torch.nn.functional.pad(x, (0, 0, 1, 1, 1, 1)).reshape(-1, ci)).reshape(B, H + 2, H + 2, 2 * ci)
            bias = torch.randn(co, device=DEV)
            flop = 2.0 * B * H * H * co * 9 * ci
            iters = max(2, min(30, int(1e12 / flop)))
            fn = lambda: ops.conv2d_h2(xh, wh, co, 3, bias=bias, colstats=True)
            ta, tb, same = [], [], True
            for rnd_ in range(3):
                os.environ["DP_H2_PP"] = "0"
                ya = fn()
                ta.append(timeit(fn, iters))
                os.environ["DP_H2_PP"] = "1"
                yb = fn()
                tb.append(timeit(fn, iters))
                same = same and torch.equal(ya.t, yb.t) and torch.equal(ya.cols.buf, yb.cols.buf)
            a, b = statistics.median(ta), statistics.median(tb)
            tot["a"] += a * cnt
            tot["b"] += min(a, b) * cnt
            tot["flop"] += flop * cnt
            print(f"{H:4d} {ci:5d} {co:5d} | {B * H * H:8d} | {flop / a / 1e9:7.1f} | {flop / b / 1e9:7.1f} | "
                  f"{a:8.3f} / {b:8.3f} (x{cnt}) | {same}", flush=True)
            del x, xh, wh
        print(f"-- {name} weighted 3x3 total: base {tot['a']:.1f} ms ({tot['flop'] / tot['a'] / 1e9:.1f} TF), "
              f"best-of {tot['b']:.1f} ms ({tot['flop'] / tot['b'] / 1e9:.1f} TF)")
    os.environ.pop("DP_H2_PP", None)


def main():
    which = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "all"
    batch = {"cifar": 256, "imagenet": 8}
    if "--batch" in sys.argv:
        b = int(sys.argv[sys.argv.index("--batch") + 1])
        batch = {"cifar": b, "imagenet": b}
    if "--pp" in sys.argv:
        return main_pp(which, batch)
    for name in (["cifar", "imagenet"] if which == "all" else [which]):
        B = batch[name]
        tot = {"f32": 0.0, "h2": 0.0, "flop": 0.0}
        print(f"== {name} B={B}: H Cin Cout | M | f32 TF | f16x3 TF | ms f32 / f16x3 (x count)")
        for (H, ci, co, cnt) in SHAPES[name]:
            x = torch.randn(B, H, H, ci, device=DEV)
            w = torch.randn(co, ci, 3, 3) * (1.0 / (9 * ci)) ** 0.5
            wp = ops.pack_conv_weight(w).to(DEV)
            wh = ops.pack_conv_weight_h2(w, DEV)
            xh = ops.pack_h2(torch.nn.functional.pad(x, (0, 0, 1, 1, 1, 1)).reshape(-1, ci)).reshape(B, H + 2, H + 2, 2 * ci)
            bias = torch.zeros(co, device=DEV)
            flop = 2.0 * B * H * H * co * 9 * ci
            iters = max(2, min(50, int(2e12 / flop)))
            t32 = timeit(lambda: ops.conv2d(x, wp, co, 3, bias=bias), iters)
            th = timeit(lambda: ops.conv2d_h2(xh, wh, co, 3, bias=bias), iters)
            tot["f32"] += t32 * cnt
            tot["h2"] += th * cnt
            tot["flop"] += flop * cnt
            print(f"{H:4d} {ci:5d} {co:5d} | {B * H * H:8d} | {flop / t32 / 1e9:7.1f} | {flop / th / 1e9:7.1f} | "
                  f"{t32:8.3f} / {th:8.3f} (x{cnt})", flush=True)
            del x, xh, wp, wh
        print(f"-- {name} weighted 3x3 total: f32 {tot['f32']:.1f} ms ({tot['flop'] / tot['f32'] / 1e9:.1f} TF), "
              f"f16x3 {tot['h2']:.1f} ms ({tot['flop'] / tot['h2'] / 1e9:.1f} TF)")


if __name__ == "__main__":
    main()
