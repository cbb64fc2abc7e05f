"""Lifetime of a tile of the 8-wave convolution kernel, at full speed (DP_H2_DW_MODE=256 of the -DDP_ABLATE library: four s_memtime stamps
per wave and the CU id, results correct): prologue (entry -> first fragments read), k-loop, epilogue (incl. the drain of the wave's
stores), and the GAP between the exit of a workgroup's last wave and the entry of the next workgroup on the same CU.
    python tests/probes/build_ablate.py && python tests/probes/dw8_lifetime.py"""
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
    for (B, H, ci, co, with_res) in [(64, 256, 256, 256, True), (64, 256, 256, 256, False), (64, 128, 512, 512, True), (64, 64, 512, 512, True)]:
        x = torch.randn(B, H, H, ci)
        w = torch.randn(co, ci, 3, 3) * (1.0 / (9 * ci)) ** 0.5
        wh = ops.order_conv_weight_w16(w).half().to(DEV)
        xh = torch.nn.functional.pad(x, (0, 0, 1, 1, 1, 1)).half().contiguous().to(DEV)
        del x
        bias = torch.randn(co, device=DEV)
        temb = torch.randn(B, co, device=DEV)
        res = torch.randn(B, H, H, co, device=DEV).half() if with_res else None
        M = B * H * H
        tiles = (M // 256) * (co // 256)
        out = torch.empty((B, H, H, co), device=DEV, dtype=torch.float16)
        cs = torch.zeros(((M + 511) // 512 * 8, 2, co), device=DEV)
        tr = ctypes.c_int(0)
        ws = torch.zeros(tiles * 8 * 8, device=DEV)
        s = torch.cuda.current_stream().cuda_stream

        def call():
            _lib.call("dp_conv2d_nhwc_h2", xh.data_ptr(), ci, B, H, H, 3, wh.data_ptr(), co, bias.data_ptr(), temb.data_ptr(), co,
                      None if res is None else res.data_ptr(), 0 if res is None else co, 1.0, out.data_ptr(), co, cs.data_ptr(), ctypes.addressof(tr),
                      ws.data_ptr(), ws.numel() * 4, 1, 1, 1, 1, 1 if res is not None else 0, None, 0, None, 0, s)

        os.environ["DP_H2_DW_MODE"] = "0"
        for _ in range(20):
            call()
        torch.cuda.synchronize()
        os.environ["DP_H2_DW_MODE"] = "256"
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            call()
        e0.record()
        call()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        r = ws.view(tiles, 8, 8).cpu().double()
        entry = r[:, :, 0] + 65536.0 * r[:, :, 1]                 # low 32 bits of s_memtime at the wave's entry
        pro, loop, epi = r[:, :, 2], r[:, :, 3], r[:, :, 4]
        exit_ = entry + pro + loop + epi
        cu = (r[:, 0, 5].long() >> 8) & 0x7f                      # HW_ID: cu_id[11:8], sh_id[12], se_id[14:13]
        key = r[:, 0, 6].long() * 128 + cu
        t_in = entry.min(dim=1).values                            # first wave in
        t_out = exit_.max(dim=1).values                           # last wave out
        gaps, per_cu = [], []
        for k in key.unique().tolist():
            idx = (key == k).nonzero().flatten()
            order = idx[torch.argsort(t_in[idx])]
            per_cu.append(len(order))
            a, b = t_out[order][:-1], t_in[order][1:]
            d = (b - a)
            d = d[(d > -1e6) & (d < 1e6)]                         # (the 32-bit counter may wrap once)
            gaps.append(d)
        gaps = torch.cat(gaps)
        med = lambda t: t.flatten().median().item()
        span = (t_out.max() - t_in.min()).item()
        flop = 2.0 * M * co * 9 * ci
        print(f"{H:4d} {ci}->{co} B={B} res16={with_res}: {ms:.3f} ms = {flop / ms / 1e9:.0f} TFLOP/s; {tiles} tiles on {len(per_cu)} CUs "
              f"({min(per_cu)}-{max(per_cu)} per CU); first entry -> last exit {span:.0f} cycles (= {span / ms / 1e3:.0f} MHz); per tile, median over waves: "
              f"prologue {med(pro):.0f}, k-loop {med(loop):.0f} ({ci * 9 // 32} k-tiles: {med(loop) / (ci * 9 // 32):.0f} each), epilogue + store drain {med(epi):.0f}; "
              f"workgroup lifetime (first wave in -> last wave out) {med(t_out - t_in):.0f}; gap to the next workgroup on the same CU: median {med(gaps):.0f}, "
              f"mean {gaps.mean().item():.0f} cycles; sum x tiles per CU = {(med(t_out - t_in) + gaps.mean().item()) * max(per_cu):.0f}", flush=True)
    os.environ["DP_H2_DW_MODE"] = "0"


if __name__ == "__main__":
    main()
