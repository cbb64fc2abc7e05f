"""Segment timeline of the 8-wave convolution kernel: where do the cycles of a steady k-tile go?
Needs the -DDP_ABLATE library (tests/probes/build_ablate.py).  DP_H2_DW_MODE=64 (shipped staging) / 192 (round 3's symmetric staging) makes every wave stamp s_memtime at the five
segment boundaries of each steady k-tile and sum the durations (igemm_h2_dw.hip): results stay correct, the kernel runs ~10 % slower
(the stamps are scalar-memory instructions whose values are consumed at the end of the k-tile).

  A  8 MFMA (set 0) | 6 ds_read_b128 (set 1) | 4 LDS-DMA pieces (k-tile t+2; the staging wave only)   ideal for the matrix pipe: 8 x 32 = 256 cycles
  B  4 MFMA (set 1, first row)                                                    128
  W  s_waitcnt vmcnt(4): this wave's pieces of k-tile t+1 have landed (+ the older wave's 4 late pieces of k-tile t+2, shipped form)
  S  s_barrier
  C  6 ds_read_b128 (set 0 of k-tile t+1) | 4 MFMA (set 1, second row)            128
A wave's k-tile holds 16 MFMAs = 512 pipe cycles; the two waves of a SIMD together 1024 per k-tile if the pipe never idles.

    python tests/probes/dw8_timeline.py [--batch 64]"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from diffpure_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.join(ROOT, "diffpure_amd", "csrc", "libdiffpure_hip_ablate.so")
from diffpure_amd import ops  # noqa: E402

DEV = "cuda:0"


def main():
    B = int(sys.argv[sys.argv.index("--batch") + 1]) if "--batch" in sys.argv else 64
    shapes = [(B, 256, 256, 256, True), (B, 128, 512, 512, True), (B, 64, 512, 512, False)]
    if "--few-tiles" in sys.argv:       # the epilogue of a launch with 16 / 64 tiles: few other CUs share HBM with it (DP_H2_DW_FORCE, ablate build)
        os.environ["DP_H2_DW_FORCE"] = "1"
        shapes = [(1, 64, 256, 256, True), (4, 64, 256, 256, True), (16, 64, 256, 256, True), (64, 64, 256, 256, True), (64, 256, 256, 256, True),
                  (1, 64, 256, 256, False), (64, 256, 256, 256, False)]
    for (B, H, ci, co, with_res) in shapes:
        x = torch.randn(B, H, H, ci)
        w = torch.randn(co, ci, 3, 3) * (1.0 / (9 * ci)) ** 0.5
        wh = ops.order_conv_weight_w16(w).half().to(DEV)
        xh = torch.nn.functional.pad(x, (0, 0, 1, 1, 1, 1)).half().contiguous().to(DEV)
        del x
        bias = torch.randn(co, device=DEV)
        res = torch.randn(B, H, H, co, device=DEV).half() if with_res else None
        M = B * H * H
        tiles = (M // 256) * (co // 256)
        out = torch.empty((B, H, H, co), device=DEV, dtype=torch.float16)
        cs = torch.zeros(((M + 511) // 512 * 8, 2, co), device=DEV)
        tr = ctypes.c_int(0)
        ws = torch.zeros(tiles * 8 * 8, device=DEV)
        s = torch.cuda.current_stream().cuda_stream

        def call():
            _lib.call("dp_conv2d_nhwc_h2", xh.data_ptr(), ci, B, H, H, 3, wh.data_ptr(), co, bias.data_ptr(), None, 0,
                      None if res is None else res.data_ptr(), 0 if res is None else co, 1.0, out.data_ptr(), co, cs.data_ptr(), ctypes.addressof(tr),
                      ws.data_ptr(), ws.numel() * 4, 1, 1, 1, 1, 1 if res is not None else 0, None, 0, None, 0, s)

        def timed(mode, iters=5):
            os.environ["DP_H2_DW_MODE"] = str(mode)
            call()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                call()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / iters

        flop = 2.0 * M * co * 9 * ci
        timed(0, 20)                                    # warm the clocks
        ref = out.clone()
        modes = {128: "symmetric staging (round 3: every wave its own 4 pieces inside segment A)",
                 0: "asymmetric staging (round 4, shipped: the older wave of every SIMD stages all 8 pieces of the pair - 4 in A, 4 after its vmcnt wait)",
                 512: "shipped staging with BUFFER-form LDS-DMA (descriptor base, scalar tap / k-tile offset, one constant lane offset per piece; not shipped: measured here)"}
        rates = {m: [] for m in modes}
        same = {}
        for _ in range(4):                              # alternate: the chip is power-limited, single measurements drift by several %
            for m in modes:
                rates[m].append(flop / timed(m, 10) / 1e9)
                same[m] = torch.equal(out, ref)
        med = lambda v: sorted(v)[len(v) // 2]
        print(f"{H:4d} {ci}->{co} B={B} res16={with_res}:", flush=True)
        for m, name in modes.items():
            print(f"   mode {m:3d} {name}: {' '.join(f'{v:.0f}' for v in rates[m])} median {med(rates[m]):.0f} TFLOP/s = {med(rates[m]) / med(rates[128]) - 1:+.1%}, "
                  f"bit-identical: {same[m]}", flush=True)
        for tmode in (192, 64):
            t64 = timed(tmode)
            r = ws.view(tiles, 8, 8).cpu().double()
            r = r[256:] if tiles > 512 else r                # skip the first round of tiles (cold start)
            n = r[:, :, 6].mean().item()
            seg = r[:, :, :5].sum(dim=(0, 1)) / r[:, :, 6].sum()          # cycles per k-tile and wave
            epi = r[:, :, 5].mean().item()
            tot = r[:, :, 7].mean().item()
            names = ["A 8MFMA+6rd+4DMA", "B 4MFMA", "W vmcnt", "S barrier", "C 6rd+4MFMA"]
            per = seg.sum().item()
            print(f"  [{ {192: 'symmetric staging', 64: 'asymmetric staging (shipped)'}[tmode] }] {flop / t64 / 1e9:.0f} TFLOP/s with the stamps; "
                  f"{n:.0f} steady k-tiles per tile; cycles per k-tile and wave: " + ", ".join(f"{nm} {v:.0f}" for nm, v in zip(names, seg.tolist())) +
                  f" = {per:.0f} (matrix-pipe floor per SIMD: 2 waves x 512 = 1024 -> pipe busy {1024 / per:.1%} inside the loop); "
                  f"epilogue {epi:.0f} cycles = {epi / tot:.1%} of a tile's {tot:.0f}", flush=True)
            # by wave: the older (0-3) and the younger (4-7, s_setprio 1 in the unrolled kernel) wave of each SIMD
            by_wave = (r[:, :, :5].sum(dim=0) / r[:, :, 6].sum(dim=0, keepdim=True).T)
            for wv in (0, 4):
                print(f"      wave {wv}: " + " ".join(f"{v:6.0f}" for v in by_wave[wv].tolist()) + f"   sum {by_wave[wv].sum().item():6.0f}")
    os.environ["DP_H2_DW_MODE"] = "0"


if __name__ == "__main__":
    main()
