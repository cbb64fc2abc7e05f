"""GPU probe: which CUs does a hipExtStreamCreateWithCUMask bit select on this part?  Times four 256^2 256->256 convolutions (B=16,
16 rounds of tiles on 256 CUs: time ~ 1 / enabled CUs) on streams with different masks:  python tests/probes/cumask_map.py"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from diffpure_amd import ops  # noqa: E402

DEV = "cuda:0"
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


B, H, C = 16, 256, 256
xh = torch.zeros(B, H + 2, H + 2, C, device=DEV, dtype=torch.float16)
xh[:, 1:-1, 1:-1] = torch.randn(B, H, H, C, device=DEV, dtype=torch.float16)
w = ops.order_conv_weight_w16(torch.randn(C, C, 3, 3) * (1.0 / (9 * C)) ** 0.5).half().to(DEV)
bias = torch.zeros(C, device=DEV)
conv = lambda: ops.conv2d_h2(xh, w, C, 3, bias=bias, colstats=True, w_fmt=1)
conv()
masks = {"all 256": range(NCU), "first 64": range(64), "first 128": range(128), "first 192": range(192), "i % 2 == 0": range(0, NCU, 2),
         "i % 4 == 0": range(0, NCU, 4), "(i // 32) % 2 == 0": [i for i in range(NCU) if (i // 32) % 2 == 0], "i % 8 < 4": [i for i in range(NCU) if i % 8 < 4],
         "i % 16 < 8": [i for i in range(NCU) if i % 16 < 8], "i % 64 < 32": [i for i in range(NCU) if i % 64 < 32]}
for name, bits in masks.items():
    s = masked_stream(bits)
    with torch.cuda.stream(s):
        conv()
    torch.cuda.synchronize()
    t0 = time.time()
    with torch.cuda.stream(s):
        for _ in range(4):
            conv()
    torch.cuda.synchronize()
    print(f"{name:20s} ({len(list(bits)):3d} bits): {(time.time() - t0) * 1e3:6.2f} ms", flush=True)
