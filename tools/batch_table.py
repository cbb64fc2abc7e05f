#!/usr/bin/env python
"""Batch-sensitivity table of the headline workload: images/s and the convolution kernels' achieved TFLOP/s at
B in {4, 8, 16, 32, 64} per GPU - B = 4 is the reference's own per-GPU batch (run_scripts/imagenet/run_in_rand_inf.sh:16), and the
table predicts the strong-scaling curve of BASELINE.json configs[3] (a fixed global batch split 8 ways).

    python tools/batch_table.py [--batches 4,8,16,32,64] [--t 100] [--workload imagenet256_guided] > gpurun_out/batch_table.json

ONE engine build; per batch: one short warm-up call (t = 2) and one full timed purification through the runner boundary
(runner.image_editing_sample on a pinned host batch), with the library's per-launch hipEvent profile over the whole call:
  pp3x3    = 3x3 launches on the 256-wide tile kernels (conv_igemm_dw from 256 tiles up, conv_igemm_sw / the ping-pong kernel below)
  other3x3 = 3x3 launches on the generic 128x128 / 64x64 tiles (small grids, split-K levels), the stem and the head
Prints one JSON line per batch and a markdown table at the end (stderr)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="4,8,16,32,64")
    ap.add_argument("--t", type=int, default=100)
    ap.add_argument("--dt", type=float, default=1e-3)
    ap.add_argument("--workload", default="imagenet256_guided", choices=["imagenet256_guided", "cifar32_ncsnpp"])
    ap.add_argument("--precision", default="f16sr")
    ap.add_argument("--seed", type=int, default=1234)
    a = ap.parse_args()
    from diffpure_amd import ops
    from diffpure_amd.sde import sde_schedule
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    wl = bench.WORKLOADS[a.workload]
    runner = bench.build_runner(a.workload, dev, a, 0, 0)
    n_steps = len(sde_schedule(wl["kind"], a.t, a.dt))
    hw = wl["hw"]
    rows = []
    for B in [int(v) for v in a.batches.split(",")]:
        xh = (torch.rand(B, 3, hw, hw, generator=torch.Generator().manual_seed(a.seed)) * 2 - 1).pin_memory()
        t_keep = runner.args.t
        runner.args.t = 2
        with torch.no_grad():
            runner.image_editing_sample(xh, bs_id=2)          # warm-up: allocator, time-table cache of this batch shape
        runner.args.t = t_keep
        torch.cuda.synchronize()
        clock = bench.SclkSampler(dev)
        clock.start()
        t0 = time.time()
        with torch.no_grad():                                  # the timed call: no per-launch events
            y = runner.image_editing_sample(xh, bs_id=2)
        torch.cuda.synchronize()
        el_clean = time.time() - t0
        sclk = clock.stop()
        ops.prof_enable(True)
        t0 = time.time()
        with torch.no_grad():                                  # the profiled call: hipEvent pairs around every convolution / GroupNorm-apply launch
            y = runner.image_editing_sample(xh, bs_id=2)
        torch.cuda.synchronize()
        el = time.time() - t0
        ops.prof_enable(False)
        prof = ops.prof_collect()
        assert torch.isfinite(y).all()
        for k in ("ms", "n", "flop", "bytes"):       # the 256-wide-tile kernels as one group, as in rounds 4-5: conv_igemm_dw + conv_igemm_dh
            prof["pp3x3"][k] += prof["dh3x3"][k]
            prof["pp1x1"][k] += prof["dh1x1"][k]
        tf = lambda k: (prof[k]["flop"] / (prof[k]["ms"] * 1e-3) / 1e12) if prof[k]["ms"] > 0 else None
        conv_ms = sum(prof[k]["ms"] for k in ("pp3x3", "other3x3", "pp1x1", "conv1x1"))
        conv_fl = sum(prof[k]["flop"] for k in ("pp3x3", "other3x3", "pp1x1", "conv1x1"))
        row = dict(workload=a.workload, per_gpu_batch=B, t=a.t, steps=n_steps, images_per_s=B / el_clean, s_per_call=el_clean, s_per_call_profiled=el,
                   unet_tflops=B / el_clean * wl["gflop"] * n_steps / 1e3, frac_of_2500=B / el_clean * wl["gflop"] * n_steps / 1e3 / 2500.0,
                   tflops_3x3_wide_tiles=tf("pp3x3"), share_3x3_wide_tiles=prof["pp3x3"]["ms"] / (el * 1e3),
                   tflops_3x3_other_tiles=tf("other3x3"), share_3x3_other_tiles=prof["other3x3"]["ms"] / (el * 1e3),
                   tflops_all_convolutions=conv_fl / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else None, share_all_convolutions=conv_ms / (el * 1e3),
                   gn_apply_gbs=(prof["gn_apply"]["bytes"] / (prof["gn_apply"]["ms"] * 1e-3) / 1e9) if prof["gn_apply"]["ms"] > 0 else None,
                   share_gn_apply=prof["gn_apply"]["ms"] / (el * 1e3), dropped_records=prof["dropped"], sclk_mhz=sclk)
        rows.append(row)
        print(json.dumps(row), flush=True)
        del xh, y
        torch.cuda.empty_cache()
    f = lambda v, d=0: "-" if v is None else f"{v:.{d}f}"
    print("| B per GPU | images/s | s per call | UNet TFLOP/s (frac of 2.5 PF) | 3x3 on 256-wide tiles: TFLOP/s (share) | 3x3 on other tiles: TFLOP/s (share) | GroupNorm-apply GB/s (share) | sclk MHz |", file=sys.stderr)
    print("|---|---|---|---|---|---|---|---|", file=sys.stderr)
    for r in rows:
        print(f"| {r['per_gpu_batch']} | {r['images_per_s']:.3f} | {r['s_per_call']:.2f} | {r['unet_tflops']:.0f} ({r['frac_of_2500']:.3f}) | "
              f"{f(r['tflops_3x3_wide_tiles'])} ({r['share_3x3_wide_tiles']:.2f}) | {f(r['tflops_3x3_other_tiles'])} ({r['share_3x3_other_tiles']:.2f}) | "
              f"{f(r['gn_apply_gbs'])} ({r['share_gn_apply']:.2f}) | {(r['sclk_mhz'] or {}).get('median')} |", file=sys.stderr)


if __name__ == "__main__":
    main()
