"""GPU probe: do an HBM-bound GroupNorm-apply pass and an MFMA-bound convolution overlap when their streams own disjoint halves of
the chip (hipExtStreamCreateWithCUMask)?  Synthetic tensors, no engine (runs in seconds):
    python tests/probes/cumask_overlap.py [B]
Prints, for each mask layout: the pass alone on its half, the convolution alone on its half, both together."""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from diffpure_amd import ops  # noqa: E402

DEV = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
NCU = torch.cuda.get_device_properties(0).multi_processor_count
lib = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))


def masked_stream(bits):
    words = (NCU + 31) // 32
    mask = (ctypes.c_uint32 * words)()
    for i in bits:
        mask[i // 32] |= 1 << (i % 32)
    s = ctypes.c_void_p()
    rc = lib.hipExtStreamCreateWithCUMask(ctypes.byref(s), ctypes.c_uint32(words), mask)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask failed: {rc}")
    return torch.cuda.ExternalStream(s.value)


def main():
    H, C = 256, 256
    x = torch.randn(B, H, H, C, device=DEV)
    gamma, beta = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
    stats = ops.group_norm_stats(x, 32, 1e-5)
    xh = torch.zeros(B, H + 2, H + 2, C, device=DEV, dtype=torch.float16)
    xh[:, 1:-1, 1:-1] = torch.randn(B, H, H, C, device=DEV, dtype=torch.float16)
    w = ops.order_conv_weight_w16(torch.randn(C, C, 3, 3) * (1.0 / (9 * C)) ** 0.5).half().to(DEV)
    bias = torch.zeros(C, device=DEV)
    gn = lambda: ops.group_norm(x, 32, 1e-5, gamma, beta, act=True, split="h1", stats=stats)
    conv = lambda: ops.conv2d_h2(xh, w, C, 3, bias=bias, colstats=True, w_fmt=1)
    NG, NC = 8, 4

    def run(sa, sb, do_a, do_b):
        torch.cuda.synchronize()
        t0 = time.time()
        if do_a:
            with torch.cuda.stream(sa):
                for _ in range(NG):
                    gn()
        if do_b:
            with torch.cuda.stream(sb):
                for _ in range(NC):
                    conv()
        torch.cuda.synchronize()
        return (time.time() - t0) * 1e3

    gn(), conv()
    cur = torch.cuda.current_stream()
    print(f"{NCU} CUs, B={B}: whole chip, one stream: {NG} GroupNorm passes {run(cur, cur, True, False):.2f} ms, {NC} convolutions "
          f"{run(cur, cur, False, True):.2f} ms, back to back {run(cur, cur, True, True):.2f} ms", flush=True)
    layouts = {"halves": (range(0, NCU // 2), range(NCU // 2, NCU)),
               "xcd-halves": ([i for i in range(NCU) if i % 8 < 4], [i for i in range(NCU) if i % 8 >= 4]),
               "even-odd": (range(0, NCU, 2), range(1, NCU, 2))}
    for name, (a, b) in layouts.items():
        try:
            sa, sb = masked_stream(a), masked_stream(b)
        except Exception as e:      # noqa: BLE001
            print(f"{name}: {e}")
            continue
        run(sa, sb, True, True)
        print(f"{name:11s}: GroupNorm on half A {run(sa, sb, True, False):.2f} ms | convolutions on half B {run(sa, sb, False, True):.2f} ms | "
              f"both at once {run(sa, sb, True, True):.2f} ms", flush=True)


if __name__ == "__main__":
    main()
