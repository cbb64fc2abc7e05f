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
`cpu_baseline` (the CPU oracle of the same network timed on ALL of this box's host cores - cores / 16 pinned 16-thread workers at
once - rank 0, N=1).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# the host driver of this pool only supports dmabuf IPC: RCCL's intra-node transport fails with `hipIpcGetMemHandle: invalid
# argument` without it.  Exported on the GPU boxes already; set here too so that a bare torchrun line works.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

# MI355X_MICROARCH.md: fp32-input MFMA = 157.3 TF; dense fp16 MFMA ~2.5 PF (the f16x3 path spends
# three fp16 MFMA passes per algorithmic MAC, so its matrix ceiling in ALGORITHMIC flops is 2500/3).
PEAK_TFLOPS = {"f32": 157.3, "f16x3": 2500.0, "f16x2": 2500.0, "f16": 2500.0, "f16sr": 2500.0}
MFMA_PASSES = {"f32": 1, "f16x3": 3, "f16x2": 2, "f16": 1, "f16sr": 1}
DTYPE_NOTES = {
    "f32": "fp32-input MFMA, exact fp32 products and accumulation",
    "f16x3": "split-fp16 operands (22 significant bits each), 3 fp16 MFMA passes per product, fp32 accumulation; purified "
             "pixels 4e-6 from the reference modules over the 100-step 256^2 loop (tests/test_gpu_loops.py)",
    "f16x2": "fp16 activations x split-fp16 weights (22 bits), 2 fp16 MFMA passes per product, fp32 accumulation, fp32 "
             "GroupNorm / residuals / SDE state; purified pixels 1.3e-4 max-abs from the reference modules over the 100-step "
             "256^2 loop (north_star bar 1e-3; tests/test_gpu_loops.py) - wider than the reference's own use_fp16 torso "
             "(configs/imagenet.yml:18)",
    "f16sr": "fp16 activations x fp16 weights, 1 MFMA pass per product, fp32 accumulation, fp32 GroupNorm statistics / SDE state, fp16 residual "
             "stream and one-pass fp16 attention (the reference's own use_fp16 torso arithmetic, configs/imagenet.yml:18; DIFFPURE_LEAN16=0 / "
             "DIFFPURE_ATTN16=0 keep them fp32 / three-pass); the "
             "fp16 weight panels are re-rounded STOCHASTICALLY (unbiased, Philox-keyed by the step) from the fp32 masters before "
             "every UNet call, so the weight-rounding error averages out over the solver steps instead of accumulating as a fixed "
             "model perturbation: purified pixels 2.3e-4 max-abs from the reference modules over the 100-step 256^2 loop (north_star "
             "bar 1e-3; round-to-nearest fp16 weights - the reference's own use_fp16 arithmetic - give 1.0e-3)",
    "f16": "fp16 activations x fp16 weights, 1 MFMA pass, fp32 accumulation = the arithmetic of the reference's use_fp16 "
           "torso; purified pixels 1.0e-3 from fp32 (at the north_star bar, not under it: not the default)",
}
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


class SclkSampler:
    """Samples the GPU's current shader clock (MHz) from sysfs (pp_dpm_sclk: the level marked '*') every 0.5 s while the
    timed region runs: the roofline fraction is quoted against a peak that assumes 2.4 GHz, the chip clocks to its
    power budget (~1.8-2.0 GHz under MFMA load), so the clock belongs next to the number."""

    def __init__(self, dev):
        import glob
        import threading
        self.samples, self._stop, self._th, self.path = [], threading.Event(), None, None
        try:
            bus = torch.cuda.get_device_properties(dev).pci_bus_id
        except Exception:
            bus = None
        cands = sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"))
        for c in cands:
            link = os.path.realpath(os.path.dirname(c))
            if bus is not None and f":{bus:02x}:" in link:
                self.path = c
        if self.path is None and len(cands) == 1:
            self.path = cands[0]
        if self.path:
            self._th = threading.Thread(target=self._run, daemon=True)

    def _read(self):
        try:
            for line in open(self.path):
                if "*" in line:
                    return float(line.split(":")[1].lower().replace("mhz", "").replace("*", "").strip())
        except Exception:
            return None
        return None

    def _run(self):
        while not self._stop.is_set():
            v = self._read()
            if v:
                self.samples.append(v)
            self._stop.wait(0.5)

    def start(self):
        if self._th:
            self._th.start()

    def stop(self):
        self._stop.set()
        if self._th:
            self._th.join()
        if not self.samples:
            return None
        s_ = sorted(self.samples)
        return dict(median=s_[len(s_) // 2], min=s_[0], max=s_[-1], samples=len(s_), source=self.path)


def pmc_traffic(workload, batch, precision):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc table (separate FETCH_SIZE and
    WRITE_SIZE passes over this very workload; profiles/README.md): PMC counters cannot be read from inside the
    process being timed, so the table is collected offline and looked up here.  None if no pass matches."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        table = json.load(open(path))
    except Exception:
        return None
    for row in table.get("rows", []):
        if row["workload"] == workload and row["per_gpu_batch"] == batch and row["precision"] == precision:
            return row
    return None


def pmc_mfma_busy():
    """MFMA utilisation of the dominant kernel by rocprofv3's own counter, from the committed PMC pass over its most common
    shape (3x3 256->256 at 256^2, B=64; tools/pmc_conv.sh): SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs).
    Collected offline (counters cannot be read from inside the timed process), at the profiler's clock.  None if absent."""
    path = os.path.join(ROOT, "profiles", "r04", "pmc_conv_igemm_dw_256x256_256to256_b64_res16_out16.json")
    try:
        row = json.load(open(path))["conv_igemm_dw"]
        return dict(value=row["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * row["GRBM_GUI_ACTIVE"] / 8.0),
                    source="profiles/r04/pmc_conv_igemm_dw_256x256_256to256_b64_res16_out16.json (SQ_VALU_MFMA_BUSY_CYCLES over the SIMD "
                           "cycles of the launch, 3x3 256->256 at 256^2, B=64, fp16 residual + fp16 output, under rocprofv3 --pmc: "
                           "collected at HEAD by tools/final_measure_r04.sh)")
    except Exception:
        return None


def build_engine(workload, device, seed, precision):
    from diffpure_amd import guided_unet, ncsnpp, synth
    if workload == "imagenet256_guided":
        cfg = guided_unet.parse_config(IMAGENET_CFG)
        sd = synth.synth_state_dict(guided_unet.param_shapes(cfg), seed)
        return guided_unet.GuidedUNet(cfg, device, precision).load_state_dict(sd), sd, cfg
    cfg = ncsnpp.parse_config(CIFAR_CFG)   # cifar32_ncsnpp and cifar32_ncsnpp_adjoint
    sd = synth.synth_state_dict(ncsnpp.param_shapes(cfg), seed)
    return ncsnpp.NCSNpp(cfg, device, precision).load_state_dict(sd), sd, cfg


CPU_WORKER_THREADS = 16     # oneDNN convolutions at batch 1-4 peak at ~16 threads on the MI355X host (probe: 8t 0.152 s, 16t 0.112 s,
                            # 32t 0.199 s, 64t 0.409 s, 256t > 100 s per NCSN++ forward): the host is used as cores / 16 workers of 16 threads


def cpu_worker(workload, t_int, seed, budget_s, cores):
    """One oracle worker (a process of its own: `bench.py --cpu-worker`): pinned to `cores`, `len(cores)` torch threads, times whole
    UNet forwards (and forward + input-gradient passes for the adjoint workload) at a small batch for ~budget_s.  Prints one JSON line."""
    os.sched_setaffinity(0, cores)
    torch.set_num_threads(len(cores))
    from diffpure_amd import guided_unet, ncsnpp, synth
    from oracle import guided_unet as og
    from oracle import ncsnpp as on
    if workload == "imagenet256_guided":
        sd = synth.synth_state_dict(guided_unet.param_shapes(guided_unet.parse_config(IMAGENET_CFG)), seed)
        cfg = og.parse_guided_config(IMAGENET_CFG)
        b = 1
        x = torch.rand(b, 3, 256, 256) * 2 - 1
        fn = lambda: og.guided_unet_forward(sd, cfg, x, torch.full((b,), float(t_int)))
    else:
        sd = synth.synth_state_dict(ncsnpp.param_shapes(ncsnpp.parse_config(CIFAR_CFG)), seed)
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
    rec = dict(s_fwd=s_fwd, calls=calls, cpu_s=el, batch=b, threads=len(cores))
    if workload.endswith("_adjoint"):
        # the adjoint solve costs one forward + one input-gradient pass per step (autograd on the CPU)
        def fb():
            xr = x.clone().requires_grad_(True)
            out = on.ncsnpp_forward(sd, cfg, xr, torch.full((b,), 99.9))
            torch.autograd.grad(out, xr, torch.ones_like(out))
        s_fb, calls2, el2 = timed(fb)
        rec.update(s_fb=s_fb, calls_fb=calls2, cpu_s=el + el2)
    print("CPUWORKER " + json.dumps(rec), flush=True)


def cpu_baseline(workload, t_int, n_steps, seed, budget_s=12.0):
    """Oracle (CPU restatement of the reference's PyTorch path) on THIS BOX'S HOST CORES - all of them: the cores this process may
    use (affinity mask) are split into workers of CPU_WORKER_THREADS threads (the size at which one oracle forward is fastest),
    every worker is a process of its own pinned to its block of cores, all run at once for ~budget_s, and their rates add up
    (independent images).  Extrapolated to images/s of the full n_steps-step purification (the UNet call is > 99.9 % of a step)."""
    import subprocess
    usable = sorted(os.sched_getaffinity(0))
    per = min(CPU_WORKER_THREADS, len(usable))
    nw = max(1, min(len(usable) // per, int(os.environ.get("DIFFPURE_CPU_WORKERS", "16"))))
    procs = []
    for w in range(nw):
        cores = ",".join(str(c) for c in usable[w * per:(w + 1) * per])
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", cores, "--workload", workload, "--t", str(t_int),
                                       "--seed", str(seed), "--cpu-budget", str(budget_s)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True))
    recs = []
    for p_ in procs:
        out, _ = p_.communicate(timeout=600)
        for line in out.splitlines():
            if line.startswith("CPUWORKER "):
                recs.append(json.loads(line[len("CPUWORKER "):]))
    if not recs:
        return dict(value=None, unit="images/s", cores=0, kind="port", sample="no oracle worker finished")
    adjoint = workload.endswith("_adjoint")
    rate = sum(1.0 / (n_steps * r["s_fwd"] + (n_steps * r["s_fb"] if adjoint else 0.0)) for r in recs)
    cpu_s = sum(r["cpu_s"] for r in recs)
    return dict(value=rate, unit="images/s", cores=len(recs) * per, kind="port",
                host={"cpu_count": os.cpu_count(), "usable_cores": len(usable), "workers": len(recs), "threads_per_worker": per,
                      "images_per_s_per_worker": [1.0 / (n_steps * r["s_fwd"] + (n_steps * r["s_fb"] if adjoint else 0.0)) for r in recs]},
                sample=(f"{len(recs)} oracle workers x {per} threads, all at once (the box's {len(usable)} usable cores of {os.cpu_count()}); "
                        f"each: {recs[0]['calls']} UNet forward(s)" + (f" + {recs[0].get('calls_fb')} forward+input-gradient pass(es)" if adjoint else "") +
                        f" at batch {recs[0]['batch']}, {cpu_s:.0f} s of CPU work in all; x{n_steps} steps" + (f" + {n_steps} adjoint steps" if adjoint else "") +
                        " extrapolated; oracle = torch-CPU fp32 restatement of the reference modules (pinned to them by tests/golden)"))


def timed_region(a, world, one_call, fence, before_first=None, after_first=None, use_dist=None):
    """W untimed warm-up calls, then exactly K timed calls bracketed by fence() on both sides; -> (seconds as the MAX over
    ranks, result of the last call).  Shared by the real bench and the CPU harness test."""
    for i in range(a.warmup):
        one_call(i)
    fence()
    t0 = time.time()
    y = None
    for i in range(a.steps):
        if i == 0 and before_first:
            before_first()
        y = one_call(a.warmup + i)
        if i == 0 and after_first:
            after_first()
    fence()
    el = time.time() - t0
    if world > 1 if use_dist is None else use_dist:
        t = torch.tensor([el], dtype=torch.float64, device=y.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = t.item()
    return el, y


def stub_main(a, rank, world):
    """--stub-engine: the distributed harness of this file (rank / world from the launcher's environment, per-rank batch,
    global sample indices, all_gather of the shards inside the timed region, barrier + MAX-over-ranks timing, one JSON line
    from rank 0) on CPU ranks over gloo.  The stand-in "purifies" x -> 0.5 x + (global sample index), so the gathered result
    is checkable."""
    if world > 1:
        dist.init_process_group("gloo")
    B, hw = a.batch or 4, 8
    gen = torch.Generator().manual_seed(a.seed + rank)
    x = torch.rand(B, 3, hw, hw, generator=gen) * 2 - 1
    gathered = torch.empty((world * B, 3, hw, hw)) if world > 1 else None

    def one_call(i):
        if a.stub_slow_rank == rank and a.stub_sleep > 0:
            time.sleep(a.stub_sleep)            # (tests: the reported time must be the MAX over ranks)
        idx = torch.arange(rank * B, rank * B + B, dtype=torch.float32).view(-1, 1, 1, 1)
        y = x * 0.5 + idx
        if world > 1:
            dist.all_gather_into_tensor(gathered, y)
        return y

    def fence():
        if world > 1:
            dist.barrier()

    el, y = timed_region(a, world, one_call, fence)
    ok = True
    if world > 1:      # every rank sees every shard, in rank order
        ok = bool(torch.equal(gathered[rank * B:(rank + 1) * B], y))
        for r in range(world):
            ok = ok and bool((gathered[r * B:(r + 1) * B].mean(dim=(1, 2, 3)) - torch.arange(r * B, r * B + B)).abs().max() < 1.0)
    if rank == 0:
        print(json.dumps({"metric": "STUB (harness test, not a measurement)", "stub": True, "value": world * B * a.steps / el,
                          "unit": "images/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": el / a.steps * 1e3,
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "stub", "data": "synthetic",
                          "config": {"workload": "stub", "per_gpu_batch": B, "global_batch": world * B}, "gather_ok": ok}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


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
    ap.add_argument("--precision", default="f16sr", choices=["f32", "f16x3", "f16x2", "f16", "f16sr"],
                    help="f32: fp32-input MFMA; f16x3: split-fp16 3-pass MFMA (fp32-class accuracy)")
    ap.add_argument("--stub-engine", action="store_true",
                    help="TEST HOOK (tests/test_bench_multirank.py): run the launch / sharding / all_gather / timing harness on CPU "
                         "ranks over gloo with a stand-in for the purification engine; the line it prints is marked stub and is "
                         "not a measurement")
    ap.add_argument("--stub-slow-rank", type=int, default=-1, help="TEST HOOK (--stub-engine): this rank sleeps --stub-sleep seconds per call")
    ap.add_argument("--stub-sleep", type=float, default=0.0)
    ap.add_argument("--force-dist", action="store_true",
                    help="TEST HOOK (tests/test_gpu_dist.py): take the multi-rank code path - init_process_group('nccl', device_id), "
                         "device-side all_gather_into_tensor of the purified shards, barrier, MAX all-reduce of the time - at world "
                         "size 1 as well, so that RCCL runs on the one GPU a test box has")
    ap.add_argument("--cpu-worker", default=None, help="INTERNAL (cpu_baseline): run one oracle worker pinned to these cores (comma list)")
    ap.add_argument("--cpu-budget", type=float, default=12.0)
    a = ap.parse_args()
    if a.cpu_worker is not None:
        return cpu_worker(a.workload, a.t, a.seed, a.cpu_budget, [int(c) for c in a.cpu_worker.split(",")])

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}")
    if a.stub_engine:
        return stub_main(a, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the purification engine has no CPU path)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = world > 1 or a.force_dist
    if use_dist:
        if "MASTER_ADDR" not in os.environ:          # --force-dist started by hand: a private single-rank rendezvous
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29533"), RANK="0", WORLD_SIZE="1")
        dist.init_process_group("nccl", device_id=dev)

    from diffpure_amd import ops
    from diffpure_amd.sde import Purifier, sde_schedule

    wl = WORKLOADS[a.workload]
    adjoint = a.workload.endswith("_adjoint")
    # per-GPU batches of BASELINE.json's configs: ImageNet 64 (configs[2], and 512 sharded 8 ways in configs[3]),
    # CIFAR 256 (configs[1]), CIFAR adjoint 128 (configs[4])
    B = a.batch or (64 if a.workload == "imagenet256_guided" else (128 if adjoint else 256))
    t_build = time.time()
    net, sd, _ = build_engine(a.workload, dev, a.seed, a.precision)
    pur = Purifier(net, wl["kind"], dev)
    torch.cuda.synchronize()
    build_s = time.time() - t_build          # synthetic weights + packing + upload: outside the timed region, reported per rank
    n_steps = len(sde_schedule(wl["kind"], a.t, a.dt))
    hw = wl["hw"]
    gen = torch.Generator().manual_seed(a.seed + rank)
    x = (torch.rand(B, 3, hw, hw, generator=gen) * 2 - 1).to(dev)      # resident in HBM before timing
    gathered = torch.empty((world * B, 3, hw, hw), device=dev) if use_dist else None

    cot = torch.randn(B, 3, hw, hw, generator=gen).to(dev) if adjoint else None
    if adjoint:
        net.enable_grad()

    def one_call(i):
        if adjoint:   # forward ODE solve, then the adjoint solve for dL/dx with a fixed cotangent
            xf = pur.ode(x, a.t, a.dt, seed=a.seed + 1000003 * i, sample0=rank * B)
            y = pur.ode_vjp(xf, cot, a.t, a.dt) * pur.diffuse_scale(a.t)
        else:
            y = pur.sde(x, a.t, a.dt, seed=a.seed + 1000003 * i, sample0=rank * B)
        if use_dist:
            dist.all_gather_into_tensor(gathered, y)
        return y

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # Instrumentation: per-launch hipEvents on the convolution launches of the FIRST timed step only (sampling: a 100-step
    # purification is 8 600 3x3 launches, the record buffer holds 65 536, and event pairs around every launch of every
    # step would sit inside the timed region for nothing); its GPU window is bracketed by two events on the same stream.
    sample = not a.no_conv_profile
    win0, win1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    clock = SclkSampler(dev)

    def before_first():
        if sample:
            win0.record()
            ops.prof_enable(True)

    def after_first():
        if sample:
            ops.prof_enable(False)
            win1.record()

    clock.start()
    el, y = timed_region(a, world, one_call, fence, before_first, after_first, use_dist)
    sclk = clock.stop()
    gather_ok = bool(torch.equal(gathered[rank * B:(rank + 1) * B], y)) if use_dist else None
    build_all = [build_s]
    if use_dist:                # so that the first real multi-GPU run explains its own wall time
        tb = torch.tensor([build_s], dtype=torch.float64, device=dev)
        allb = [torch.zeros_like(tb) for _ in range(world)]
        dist.all_gather(allb, tb)
        build_all = [float(v.item()) for v in allb]
    # the boundary hands over device tensors (runner.image_editing_sample(img) with img on the GPU), so `value` is timed with the
    # batch resident in HBM; the host->device copy of one batch is measured here, outside the timed region, for the record
    xh = x.cpu().pin_memory()
    torch.cuda.synchronize()
    t_h = time.time()
    xh.to(dev, non_blocking=True)
    torch.cuda.synchronize()
    h2d_ms = (time.time() - t_h) * 1e3
    prof = ops.prof_collect() if sample else None
    window_ms = win0.elapsed_time(win1) if sample else None
    assert torch.isfinite(y).all()

    images = world * B * a.steps
    value = images / el

    out = None
    if rank == 0:
        peak = PEAK_TFLOPS[a.precision]
        passes = MFMA_PASSES[a.precision]
        unet_tflops = value / world * wl["gflop"] * n_steps / 1e3
        roof = {"bound": "mfma",
                "kernel": "conv_igemm_f32 (3x3 implicit GEMM, v_mfma_f32_32x32x2_f32)" if a.precision == "f32" else
                          (f"conv_igemm_dw (3x3 implicit GEMM, 256x256 tile, one 8-wave workgroup per CU: two free-running waves per SIMD with 64x128 "
                           f"wave tiles sharing the tile in LDS, the OLDER wave of every SIMD staging the rows of both (LDS-DMA, separate rings for "
                           f"activations and weights, counted vmcnt, one barrier per k-tile); launches with fewer than 256 tiles run conv_igemm_sw, the "
                           f"one-wave-per-SIMD form; {passes} x v_mfma_f32_32x32x16_f16 per product)" if a.precision in ("f16", "f16sr") else
                           f"conv_igemm_h2_pp (3x3 implicit GEMM on the 8-wave ping-pong kernel; {passes} x v_mfma_f32_32x32x16_f16 per "
                           f"product; executed MFMA flops = {passes} x achieved)"),
                "achieved": None, "peak": peak, "unit": "TFLOP/s", "frac": None, "traffic": None,
                "peak_note": "dense fp16 MFMA peak at the 2.4 GHz nominal clock (MI355X_MICROARCH.md); see sclk_mhz for the clock this run held",
                "mfma_passes": passes, "sclk_mhz": sclk, "end_to_end_unet_tflops_per_gpu": unet_tflops,
                "switches": {k: os.environ[k] for k in ("DIFFPURE_LEAN", "DIFFPURE_LEAN16", "DP_H2_SW", "DP_H2_DW", "DP_H2_PP", "DIFFPURE_STREAMS") if k in os.environ}}
        if prof is not None:
            dom = prof["pp3x3"] if prof["pp3x3"]["n"] else prof["other3x3"]      # f32 / tiny shapes never reach the ping-pong kernel
            if dom["ms"] > 0:
                ach = dom["flop"] / (dom["ms"] * 1e-3) / 1e12
                launches_per_call = dom["n"]
                held = (sclk or {}).get("median")
                roof.update({
                    "achieved": ach, "frac": ach / peak, "executed_frac": ach * passes / peak,
                    # the same against the peak at the clock this run HELD (the chip is power-limited under MFMA load: 2.5 PF assumes
                    # 2.4 GHz): achieved / (peak x sclk_median / 2400).  null if the clock could not be sampled.
                    "frac_at_held_clock": (ach / (peak * held / 2400.0)) if held else None,
                    "avg_launch_ms": dom["ms"] / dom["n"],
                    "algorithmic_bytes_per_launch": dom["bytes"] / dom["n"],
                    "algorithmic_gflop_per_launch": dom["flop"] / dom["n"] / 1e9,
                    # sampling: hipEvent pairs on the launches of the first timed step only
                    "sampled_launches": dom["n"], "sampled_of_total": 1.0 / a.steps,
                    "launches_in_timed_region": launches_per_call * a.steps, "dropped_records": prof["dropped"],
                    "time_share_of_step": dom["ms"] / window_ms,
                    "sampled_step_ms": window_ms,
                    "other_kernels_share_of_step": {"3x3 on other tile variants (stem, head, split-K levels)": prof["other3x3"]["ms"] / window_ms,
                                                    "1x1 convolutions / linear": (prof["conv1x1"]["ms"] + prof["pp1x1"]["ms"]) / window_ms},
                })
                gn = prof.get("gn_apply")
                if gn and gn["ms"] > 0:     # the second kernel of the step: HBM-bound, measured with the same hipEvent pairs
                    gbs = gn["bytes"] / (gn["ms"] * 1e-3) / 1e9
                    roof["second_kernel"] = {"kernel": "GroupNorm-apply (gn_apply_h16 / gn_apply_h2q: normalise + FiLM + SiLU + resample, emits the convolution operand)",
                                             "bound": "hbm", "achieved": gbs, "peak": 8000.0, "unit": "GB/s", "frac": gbs / 8000.0,
                                             "frac_of_measured_copy_rate_6290": gbs / 6290.0, "algorithmic_bytes_per_launch": gn["bytes"] / gn["n"],
                                             "avg_launch_ms": gn["ms"] / gn["n"], "sampled_launches": gn["n"], "time_share_of_step": gn["ms"] / window_ms}
                if a.workload == "imagenet256_guided" and a.precision in ("f16", "f16sr"):
                    busy = pmc_mfma_busy()
                    if busy is not None:
                        roof["mfma_busy_by_pmc"] = busy
                row = pmc_traffic(a.workload, B, a.precision)
                if row is not None:
                    # rocprofv3 aggregates per kernel NAME: the ping-pong kernel's 3x3 and 1x1 launches together
                    allpp_n = prof["pp3x3"]["n"] + prof["pp1x1"]["n"]
                    roof["algorithmic_bytes_per_launch_all_pingpong_launches"] = (prof["pp3x3"]["bytes"] + prof["pp1x1"]["bytes"]) / max(1, allpp_n)
                    roof["traffic"] = row["hbm_bytes_per_launch"]
                    roof["traffic_detail"] = {k: row[k] for k in row if k not in ("workload", "per_gpu_batch", "precision")}
                else:
                    roof["traffic_note"] = ("no rocprofv3 --pmc pass committed for this (workload, batch, precision): "
                                            "profiles/pmc_traffic.json")
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
            "dtype_note": DTYPE_NOTES[a.precision],
            "data": "synthetic (seeded uniform images in [-1,1]; seeded non-trivial random weights of the named "
                    "architecture; Philox noise)",
            "config": {"workload": (f"{a.workload}: probability-flow ODE purification ({n_steps} Euler steps) + continuous-adjoint "
                                    f"backward for dL/dx ({n_steps} steps, each one UNet forward + one input-gradient pass), "
                                    f"t*={a.t / 1000:g}, step={a.dt:g}") if adjoint else
                                   (f"{a.workload}: reverse VP-SDE purification, t*={a.t / 1000:g}, dt={a.dt:g}, "
                                    f"{n_steps} Euler-Maruyama steps, one UNet call per step"),
                       "per_gpu_batch": B, "global_batch": world * B, "image": f"3x{hw}x{hw}",
                       "parallelism": f"batch-sharded x{world}, one all_gather of outputs"},
            "roofline": roof,
            "engine_build_s_per_rank": build_all,
            "input": {"resident_in_hbm_before_timing": True, "h2d_ms_per_batch_pinned": h2d_ms,
                      "value_if_h2d_were_inside_the_timed_region": images / (el + h2d_ms * 1e-3 * a.steps)},
        }
        if use_dist:
            out["collectives"] = {"backend": dist.get_backend(), "world_size": world, "forced_at_world_1": bool(a.force_dist and world == 1),
                                  "all_gather_into_tensor_on_device": True, "gathered_equals_local_shard": gather_ok}
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.workload, a.t, n_steps, a.seed)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
