#!/usr/bin/env python
"""Headline benchmark: purified images/sec, 256x256 guided-diffusion VP-SDE, t*=0.1, 100
Euler-Maruyama steps (BASELINE.json `metric`), on N GPUs of one node.

A "step" is ONE full purification call (diffuse + 100 reverse-SDE steps, one UNet forward each)
over one per-GPU batch of synthetic images already resident in HBM.  Weak scaling: the per-GPU
batch is fixed, rank r purifies global samples [r*B, (r+1)*B), and the purified shards are
reassembled with one RCCL all_gather inside the timed region.

    python bench.py --gpus 1 --steps 1 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line (rank 0).  Extra objects: `roofline` for the dominant kernel (3x3 implicit-
GEMM convolution; algorithmic FLOPs / hipEvent-timed launch durations over the timed region) and
`cpu_baseline` (the CPU oracle of the same network timed on this box's host cores, rank 0, N=1).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

# MI355X_MICROARCH.md: fp32-input MFMA = 157.3 TF; dense fp16 MFMA ~2.5 PF (the f16x3 path spends
# three fp16 MFMA passes per algorithmic MAC, so its matrix ceiling in ALGORITHMIC flops is 2500/3).
PEAK_TFLOPS = {"f32": 157.3, "f16x3": 2500.0, "f16x2": 2500.0, "f16": 2500.0}
MFMA_PASSES = {"f32": 1, "f16x3": 3, "f16x2": 2, "f16": 1}
WORKLOADS = {
    # name: (kind, config file section, image size, algorithmic GFLOP / image / UNet call, 3x3 share)
    "imagenet256_guided": dict(kind="guided", hw=256, gflop=2239.67, gflop3x3=2115.44),
    "cifar32_ncsnpp": dict(kind="ncsnpp", hw=32, gflop=37.094, gflop3x3=33.629),
    # BASELINE.json configs[4]: probability-flow ODE forward + continuous-adjoint backward (dL/dx only):
    # 100 forward UNet calls + 100 x (forward + input-gradient pass) = 300 F of convolution work
    "cifar32_ncsnpp_adjoint": dict(kind="ncsnpp", hw=32, gflop=3 * 37.094, gflop3x3=3 * 33.629),
}
IMAGENET_CFG = dict(attention_resolutions="32,16,8", class_cond=False, diffusion_steps=1000, rescale_timesteps=True,
                    timestep_respacing="1000", image_size=256, learn_sigma=True, noise_schedule="linear",
                    num_channels=256, num_head_channels=64, num_res_blocks=2, resblock_updown=True, use_fp16=True,
                    use_scale_shift_norm=True)   # /root/reference/configs/imagenet.yml:5-19
CIFAR_CFG = dict(
    data=dict(dataset="CIFAR10", image_size=32, num_channels=3, centered=True),
    model=dict(sigma_min=0.01, sigma_max=50, num_scales=1000, beta_min=0.1, beta_max=20.0, dropout=0.1, name="ncsnpp",
               scale_by_sigma=False, ema_rate=0.9999, normalization="GroupNorm", nonlinearity="swish", nf=128,
               ch_mult=[1, 2, 2, 2], num_res_blocks=8, attn_resolutions=[16], resamp_with_conv=True, conditional=True,
               fir=False, fir_kernel=[1, 3, 3, 1], skip_rescale=True, resblock_type="biggan", progressive="none",
               progressive_input="none", progressive_combine="sum", attention_type="ddpm", init_scale=0.0,
               embedding_type="positional", fourier_scale=16, conv_size=3))   # configs/cifar10.yml


def build_engine(workload, device, seed, precision):
    from diffpure_amd import guided_unet, ncsnpp, synth
    if workload == "imagenet256_guided":
        cfg = guided_unet.parse_config(IMAGENET_CFG)
        sd = synth.synth_state_dict(guided_unet.param_shapes(cfg), seed)
        return guided_unet.GuidedUNet(cfg, device, precision).load_state_dict(sd), sd, cfg
    cfg = ncsnpp.parse_config(CIFAR_CFG)   # cifar32_ncsnpp and cifar32_ncsnpp_adjoint
    sd = synth.synth_state_dict(ncsnpp.param_shapes(cfg), seed)
    return ncsnpp.NCSNpp(cfg, device, precision).load_state_dict(sd), sd, cfg


def cpu_baseline(workload, sd, t_int, n_steps, budget_s=12.0):
    """Oracle (CPU restatement of the reference's PyTorch path) on this box's host cores: time whole
    UNet forwards at a reduced batch until ~budget_s of CPU work is spent, extrapolate to images/s
    for the full n_steps-step purification (the UNet call is >99.9 % of a step)."""
    from oracle import guided_unet as og
    from oracle import ncsnpp as on
    # usable cores = the affinity mask (os.cpu_count() over-reports inside a cgroup / container);
    # oneDNN convolutions at batch 1-4 peak at ~16 threads on the MI355X host (probe: 8t 0.152 s, 16t 0.112 s, 32t 0.199 s, 64t 0.409 s, 256t >100 s).
    cores = min(len(os.sched_getaffinity(0)), int(os.environ.get("DIFFPURE_CPU_THREADS", "16")))
    torch.set_num_threads(cores)
    if workload == "imagenet256_guided":
        cfg = og.parse_guided_config(IMAGENET_CFG)
        b = 1
        x = torch.rand(b, 3, 256, 256) * 2 - 1
        fn = lambda: og.guided_unet_forward(sd, cfg, x, torch.full((b,), float(t_int)))
    else:
        cfg = on.parse_ncsnpp_config(CIFAR_CFG)
        b = 4
        x = torch.rand(b, 3, 32, 32) * 2 - 1
        fn = lambda: on.ncsnpp_forward(sd, cfg, x, torch.full((b,), 99.9))

    def timed(f):
        f()  # warm-up (oneDNN primitive creation)
        calls, t0 = 0, time.time()
        while calls < 1 or (time.time() - t0 < budget_s and calls < 64):
            f()
            calls += 1
        return (time.time() - t0) / (calls * b), calls, time.time() - t0

    with torch.no_grad():
        s_fwd, calls, el = timed(fn)
    if workload.endswith("_adjoint"):
        # the adjoint solve costs one forward + one input-gradient pass per step (autograd on the CPU)
        def fb():
            xr = x.clone().requires_grad_(True)
            out = on.ncsnpp_forward(sd, cfg, xr, torch.full((b,), 99.9))
            torch.autograd.grad(out, xr, torch.ones_like(out))
        s_fb, calls2, el2 = timed(fb)
        per_image = n_steps * s_fwd + n_steps * s_fb
        return dict(value=1.0 / per_image, unit="images/s", cores=cores, kind="port",
                    sample=f"{calls} forward(s) + {calls2} forward+input-gradient pass(es) at batch {b} ({el + el2:.1f} s of CPU "
                           f"work); {n_steps} ODE steps + {n_steps} adjoint steps extrapolated; torch-CPU oracle")
    return dict(value=1.0 / (s_fwd * n_steps), unit="images/s", cores=cores, kind="port",
                sample=f"{calls} UNet forward(s) at batch {b} ({el:.1f} s of CPU work), x{n_steps} steps extrapolated; "
                       "oracle = torch-CPU fp32 restatement of the reference modules (pinned to them by tests/golden)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="imagenet256_guided", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="images per GPU per purification call (0 = workload default)")
    ap.add_argument("--t", type=int, default=100, help="t* in 1/1000 (100 = t*=0.1)")
    ap.add_argument("--dt", type=float, default=1e-3)
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-conv-profile", action="store_true",
                    help="do not time the convolution launches with hipEvents (roofline.achieved = null); needed to see the "
                         "HIP-graph step of small batches, which is never used while that profiler records")
    ap.add_argument("--precision", default="f16x2", choices=["f32", "f16x3", "f16x2", "f16"],
                    help="f32: fp32-input MFMA; f16x3: split-fp16 3-pass MFMA (fp32-class accuracy)")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the purification engine has no CPU path)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)

    from diffpure_amd import ops
    from diffpure_amd.sde import Purifier, sde_schedule

    wl = WORKLOADS[a.workload]
    adjoint = a.workload.endswith("_adjoint")
    # per-GPU batches of BASELINE.json's configs: ImageNet 64 (configs[2], and 512 sharded 8 ways in configs[3]),
    # CIFAR 256 (configs[1]), CIFAR adjoint 128 (configs[4])
    B = a.batch or (64 if a.workload == "imagenet256_guided" else (128 if adjoint else 256))
    net, sd, _ = build_engine(a.workload, dev, a.seed, a.precision)
    pur = Purifier(net, wl["kind"], dev)
    n_steps = len(sde_schedule(wl["kind"], a.t, a.dt))
    hw = wl["hw"]
    gen = torch.Generator().manual_seed(a.seed + rank)
    x = (torch.rand(B, 3, hw, hw, generator=gen) * 2 - 1).to(dev)      # resident in HBM before timing
    gathered = torch.empty((world * B, 3, hw, hw), device=dev) if world > 1 else None

    cot = torch.randn(B, 3, hw, hw, generator=gen).to(dev) if adjoint else None
    if adjoint:
        net.enable_grad()

    def one_call(i):
        if adjoint:   # forward ODE solve, then the adjoint solve for dL/dx with a fixed cotangent
            xf = pur.ode(x, a.t, a.dt, seed=a.seed + 1000003 * i, sample0=rank * B)
            y = pur.ode_vjp(xf, cot, a.t, a.dt) * pur.diffuse_scale(a.t)
        else:
            y = pur.sde(x, a.t, a.dt, seed=a.seed + 1000003 * i, sample0=rank * B)
        if world > 1:
            dist.all_gather_into_tensor(gathered, y)
        return y

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(a.warmup):
        one_call(i)
    fence()
    ops.prof_enable(not a.no_conv_profile)
    t0 = time.time()
    for i in range(a.steps):
        y = one_call(a.warmup + i)
    fence()
    el = time.time() - t0
    prof = ops.prof_collect()
    ops.prof_enable(False)
    assert torch.isfinite(y).all()

    if world > 1:
        t = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = t.item()
    images = world * B * a.steps
    value = images / el

    out = None
    if rank == 0:
        ach = prof["flop3x3"] / (prof["ms3x3"] * 1e-3) / 1e12 if prof["ms3x3"] > 0 else None
        peak = PEAK_TFLOPS[a.precision]
        unet_tflops = value / world * wl["gflop"] * n_steps / 1e3
        out = {
            "metric": "purified images/sec (whole node), 256x256 GuidedDiff VP-SDE t*=0.1 100-step"
            if a.workload == "imagenet256_guided" and a.t == 100 and n_steps == 100
            else (f"images/sec with input gradient (whole node), {a.workload}: ODE purification + adjoint dL/dx, t={a.t} "
                  f"{n_steps}-step" if adjoint else
                  f"purified images/sec (whole node), {a.workload} VP-SDE t={a.t} {n_steps}-step"),
            "value": value,
            "unit": "images/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": el / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": a.precision,
            "data": "synthetic (seeded uniform images in [-1,1]; seeded non-trivial random weights of the named "
                    "architecture; Philox noise)",
            "config": {"workload": (f"{a.workload}: probability-flow ODE purification ({n_steps} Euler steps) + continuous-adjoint "
                                    f"backward for dL/dx ({n_steps} steps, each one UNet forward + one input-gradient pass), "
                                    f"t*={a.t / 1000:g}, step={a.dt:g}") if adjoint else
                                   (f"{a.workload}: reverse VP-SDE purification, t*={a.t / 1000:g}, dt={a.dt:g}, "
                                    f"{n_steps} Euler-Maruyama steps, one UNet call per step"),
                       "per_gpu_batch": B, "global_batch": world * B, "image": f"3x{hw}x{hw}",
                       "parallelism": f"batch-sharded x{world}, one all_gather of outputs"},
            "roofline": {"bound": "mfma",
                         "kernel": "conv_igemm_f32 (3x3 implicit GEMM, v_mfma_f32_32x32x2_f32)" if a.precision == "f32" else
                                   "conv_igemm_h2 (3x3 implicit GEMM, 3 x v_mfma_f32_32x32x16_f16 per product; "
                                   "executed MFMA flops = 3 x achieved)",
                         "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": None if ach is None else ach / peak,
                         # HBM bytes per launch need PMC counters (separate rocprofv3 --pmc passes), which cannot be
                         # collected from inside this process: measured offline for the dominant kernel, see the note
                         "traffic": None,
                         "traffic_note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE on conv_igemm_h2_pp, 256^2 x 256->256, B=16: "
                                         "1.44 GB fetched (gfx950-corrected) + 1.07 GB written per launch vs 2.16 GB algorithmic "
                                         "= 0.84 TB/s of 8 TB/s (profiles/README.md, section 1)",
                         "launches": prof["n3x3"],
                         # f16x3 spends 3 fp16 MFMA passes per algorithmic MAC: the matrix pipe executes 3x `achieved`
                         "mfma_passes": MFMA_PASSES[a.precision],
                         "executed_frac": None if ach is None else ach * MFMA_PASSES[a.precision] / peak,
                         "avg_launch_ms": prof["ms3x3"] / max(1, prof["n3x3"]),
                         "time_share_of_step": prof["ms3x3"] * 1e-3 / el,
                         "end_to_end_unet_tflops_per_gpu": unet_tflops},
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.workload, sd, a.t, n_steps)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
