"""GPU probe (NOT YET RUN - written at the end of round 2 for the first GPU call of the next round): two half-batches on two HIP
streams that own DISJOINT HALVES OF THE CHIP (hipExtStreamCreateWithCUMask), two host threads, two engines.

Why: two plain streams gave 1.00-1.01x (tests/probes/two_stream.py) because a convolution launch takes every CU (one 128 KB-LDS
workgroup per CU) and the other stream's kernels simply queue behind it.  With complementary CU masks the two purifications really
run side by side: the HBM-bound GroupNorm-apply passes of one half (17.6 % of the kernel time, ~5.2 TB/s whether 256 or 128 CUs
pull) sit under the MFMA-bound convolutions of the other, and the epilogue bursts of the convolutions (all CUs of a launch store
their 256 KB tiles at the same moment: ~10 us of every 84 us tile) are de-phased between the halves.
Upper bound if GroupNorm-apply hides completely: +15 %.

The mapping of mask bits to CUs / XCDs is not documented for gfx950 in this image, so three layouts are timed:
  halves      bits [0, 128) | [128, 256)
  xcd-halves  bit i in A when (i % 8) < 4           (the split by XCD if CU ids are dealt round-robin over the 8 XCDs)
  even-odd    bit i in A when i is even
Results must equal the one-stream run bit for bit (the noise is keyed by the global sample index).
    python tests/probes/two_stream_cumask.py [B per stream] [steps]"""
import ctypes
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from bench import IMAGENET_CFG  # noqa: E402
from diffpure_amd import guided_unet, synth  # noqa: E402
from diffpure_amd.sde import Purifier  # noqa: E402

DEV = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T = int(sys.argv[2]) if len(sys.argv) > 2 else 10
NCU = torch.cuda.get_device_properties(0).multi_processor_count


def hip():
    return ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))


def masked_stream(lib, bits):
    """bits: iterable of CU indices -> torch ExternalStream over a hipExtStreamCreateWithCUMask stream"""
    words = (NCU + 31) // 32
    mask = (ctypes.c_uint32 * words)()
    for i in bits:
        mask[i // 32] |= 1 << (i % 32)
    s = ctypes.c_void_p()
    rc = lib.hipExtStreamCreateWithCUMask(ctypes.byref(s), ctypes.c_uint32(words), mask)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask failed: {rc}")
    return torch.cuda.ExternalStream(s.value)


def timed(fn):
    fn()
    torch.cuda.synchronize()
    t0 = time.time()
    r = fn()
    torch.cuda.synchronize()
    return time.time() - t0, r


def main():
    cfg = guided_unet.parse_config(IMAGENET_CFG)
    sd = synth.synth_state_dict(guided_unet.param_shapes(cfg), 1234)
    nets = [guided_unet.GuidedUNet(cfg, DEV, "f16sr").load_state_dict(sd) for _ in range(2)]
    purs = [Purifier(n, "guided", DEV) for n in nets]
    x = (torch.rand(2 * B, 3, 256, 256) * 2 - 1).to(DEV)
    t_full, y_full = timed(lambda: purs[0].sde(x, T, 1e-3, seed=1, sample0=0))
    print(f"{NCU} CUs; B={2 * B}, {T} steps on one stream: {t_full * 1e3 / T:.1f} ms/step", flush=True)
    lib = hip()
    layouts = {
        "plain streams (no mask)": None,
        "halves": (range(0, NCU // 2), range(NCU // 2, NCU)),
        "xcd-halves": ([i for i in range(NCU) if i % 8 < 4], [i for i in range(NCU) if i % 8 >= 4]),
        "even-odd": (range(0, NCU, 2), range(1, NCU, 2)),
    }
    for name, sets in layouts.items():
        try:
            streams = [torch.cuda.Stream(), torch.cuda.Stream()] if sets is None else [masked_stream(lib, s) for s in sets]
        except Exception as e:      # noqa: BLE001
            print(f"{name}: {e}")
            continue

        def split():
            out = [None, None]

            def work(i):
                torch.cuda.set_device(0)
                with torch.cuda.stream(streams[i]):
                    out[i] = purs[i].sde(x[i * B:(i + 1) * B], T, 1e-3, seed=1, sample0=i * B)

            th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
            for t in th:
                t.start()
            for t in th:
                t.join()
            for st in streams:
                st.synchronize()
            return torch.cat(out)

        t_split, y_split = timed(split)
        print(f"{name:26s}: {t_split * 1e3 / T:.1f} ms/step ({t_full / t_split:.3f}x), identical results: {torch.equal(y_full, y_split)}", flush=True)


if __name__ == "__main__":
    main()
