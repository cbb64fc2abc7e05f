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
`cpu_baseline` (the reference's own score network - oracle/_ref, `kind: "reference"` - or, without that copy, the CPU oracle
restatement - `kind: "port"` - timed on this box's host cores as pinned 16-thread workers: the BEST of 1 / 2 / 4 / 8 / 16 workers
at once, with the whole sweep in the line; rank 0, N=1).

Round 5: the timed call is the DROP-IN BOUNDARY - `runner.image_editing_sample(img)` of runners/diffpure_sde.py /
diffpure_ode.py (engine pool lock, dispatch, NCHW <-> NHWC, and the host -> device copy of the batch: `img` is a pinned HOST
tensor) - as SURVEY.md section 8(d) defines the metric.  `value` is that; the rate of the bare engine loop on a batch already
resident in HBM is measured by one extra call after the timed region and reported as `input.value_resident_batch_engine_call`.
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
    # SURVEY 8f-1, what the reference's ImageNet adaptive attacks differentiate (run_scripts/imagenet/run_in_rand_inf.sh:12-24 ->
    # runners/diffpure_sde.py:236-238, torchsde.sdeint_adjoint): reverse-SDE solve + stochastic adjoint for dL/dx on the guided UNet.
    # FLOPs per image: N forward UNet calls + N x (taped forward + input-gradient pass) = 3 N F.  The reference's adjoint also
    # integrates the PARAMETER adjoints (one more F per step for weight gradients nobody reads): not formed here, not counted.
    "imagenet256_guided_sde_adjoint": dict(kind="guided", hw=256, gflop=3 * 2239.67, gflop3x3=3 * 2115.44),
}
DEFAULT_BATCH = {"imagenet256_guided": 64, "cifar32_ncsnpp": 256, "cifar32_ncsnpp_adjoint": 128,
                 # the taped forward of the guided UNet holds ~2 GiB per image (fp32 residual stream + fp32 qkv): 32 images = 63 GiB of
                 # the 288; the reference's own scripts run 4 per GPU (--batch 4)
                 "imagenet256_guided_sde_adjoint": 32}
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
        # socket power of the SAME card (hwmon power1_average / power1_input, microwatts): the chip sits at its 1 400 W cap under the
        # dominant kernel, so the clock it holds - and with it the roofline fraction - is a power figure
        self.power, self.pow_path = [], None
        if self.path:
            d = os.path.dirname(self.path)
            pc = sorted(glob.glob(os.path.join(d, "hwmon/hwmon*/power1_average"))) + sorted(glob.glob(os.path.join(d, "hwmon/hwmon*/power1_input")))
            self.pow_path = pc[0] if pc else None
            # junction / memory temperatures of the same card (millidegrees): a 20-step run holds the cap for minutes, and the clock the
            # power manager grants at the cap drifts with the temperature (round 6: first / last quarter of the clock samples are reported)
            self.temp_paths = sorted(glob.glob(os.path.join(d, "hwmon/hwmon*/temp*_input")))
            self.temps = []
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
            if self.pow_path:
                try:
                    self.power.append(float(open(self.pow_path).read().strip()) * 1e-6)
                except Exception:
                    pass
            t_ = []
            for tp in getattr(self, "temp_paths", []):
                try:
                    t_.append(float(open(tp).read().strip()) * 1e-3)
                except Exception:
                    pass
            if t_:
                self.temps.append(max(t_))
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
        out = dict(median=s_[len(s_) // 2], min=s_[0], max=s_[-1], samples=len(s_), source=self.path)
        q = max(1, len(self.samples) // 4)
        med = lambda v: sorted(v)[len(v) // 2]
        out.update(first_quarter_median=med(self.samples[:q]), last_quarter_median=med(self.samples[-q:]))
        if getattr(self, "temps", None):
            out.update(temp_c_max_first_quarter=max(self.temps[:max(1, len(self.temps) // 4)]), temp_c_max=max(self.temps))
        if self.power:
            p_ = sorted(self.power)
            out.update(power_w_median=p_[len(p_) // 2], power_w_max=p_[-1], power_source=self.pow_path)
        return out


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


def traffic_keys(row, prof, unet_calls=1):
    """roofline.traffic and the algorithmic bytes it is compared with, over the SAME launch set (round 6; round 5 printed a counter
    average over 232 launches of three kernels next to the algorithmic bytes of the 3x3 launches of one).  rocprofv3 aggregates per
    kernel NAME, the in-process profile per launch kind, and since ABI 8 a kind belongs to one kernel: the row's `per_kernel` entry of
    conv_igemm_dw covers exactly kinds pp3x3 + pp1x1 (every launch of that kernel in a UNet call, 3x3 and 1x1)."""
    pk = (row.get("per_kernel") or {}).get("conv_igemm_dw")
    n = prof["pp3x3"]["n"] + prof["pp1x1"]["n"]
    if not pk or not n:
        return {"traffic": row.get("hbm_bytes_per_launch"), "traffic_note": "committed PMC pass has no per-kernel rows: the average is over every "
                "256-wide-tile kernel's launches and is NOT comparable with algorithmic_bytes_per_launch (3x3 launches of conv_igemm_dw only)"}
    alg = (prof["pp3x3"]["bytes"] + prof["pp1x1"]["bytes"]) / n
    return {"traffic": pk["hbm_bytes_per_launch"], "traffic_launch_set": "every conv_igemm_dw launch of a UNet call (3x3 and 1x1), per launch",
            "traffic_launches_per_unet_call_pmc": pk["launches"] / max(1, row.get("unet_calls_profiled", 2)),
            "traffic_launches_per_unet_call_this_run": n / max(1, unet_calls),
            "algorithmic_bytes_per_launch_same_set": alg, "traffic_over_algorithmic": pk["hbm_bytes_per_launch"] / alg,
            "traffic_fetch_bytes_per_launch": pk["fetch_bytes_per_launch"], "traffic_write_bytes_per_launch": pk["write_bytes_per_launch"],
            "traffic_source": row.get("source", "") + " (a committed offline pass over this workload, not a measurement of this run)"}


def pmc_mfma_busy():
    """MFMA utilisation of the dominant kernel by rocprofv3's own counter, from the committed PMC pass over its most common
    shape (3x3 256->256 at 256^2, B=64; tools/pmc_conv.sh): SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs).
    Collected offline (counters cannot be read from inside the timed process), at the profiler's clock.  None if absent."""
    for rnd in ("r06", "r05", "r04"):      # the newest committed pass (the kernel itself did not change in rounds 5-6)
        rel = os.path.join("profiles", rnd, "pmc_conv_igemm_dw_256x256_256to256_b64_res16_out16.json")
        try:
            row = json.load(open(os.path.join(ROOT, rel)))["conv_igemm_dw"]
            return dict(value=row["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * row["GRBM_GUI_ACTIVE"] / 8.0),
                        source=rel + " (SQ_VALU_MFMA_BUSY_CYCLES over the SIMD cycles of the launch, 3x3 256->256 at 256^2, B=64, fp16 residual + "
                                     "fp16 output, under rocprofv3 --pmc: a committed offline pass, not a measurement of this run)")
        except Exception:
            continue
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
                            # 32t 0.199 s, 64t 0.409 s, 256t > 100 s per NCSN++ forward): the host is used as workers of 16 threads
CPU_SWEEP = (1, 2, 4, 8, 16)   # concurrent workers tried; the BEST aggregate rate is the baseline (round 4 ran 16 at once: they contend
                               # for memory bandwidth - 0.0038 images/s where 16 threads alone gave 0.0104 - the most oversubscribed
                               # configuration is not the host's best one)


def cpu_worker(workload, t_int, seed, cores):
    """One CPU worker (a process of its own: `bench.py --cpu-worker`): pinned to `cores`, `len(cores)` torch threads.  Builds the score
    network once - the REFERENCE's own nn.Module from oracle/_ref when that copy is present and matches its tracked digests
    (oracle/ref_loader.py), else the oracle restatement - warms it up, prints `CPUWORKER READY {...}` and then serves `GO <seconds>`
    lines on stdin: each times whole UNet forwards (and forward + input-gradient passes for the adjoint workloads) at a small batch
    for ~<seconds> and prints one `CPUWORKER RES {...}` line.  `QUIT` / EOF ends it."""
    os.sched_setaffinity(0, cores)
    torch.set_num_threads(len(cores))
    from diffpure_amd import guided_unet, ncsnpp, synth
    from oracle import guided_unet as og
    from oracle import ncsnpp as on
    from oracle import ref_loader
    kind = "reference" if ref_loader.available() else "port"
    guided = workload.startswith("imagenet256_guided")
    if guided:
        sd = synth.synth_state_dict(guided_unet.param_shapes(guided_unet.parse_config(IMAGENET_CFG)), seed)
        b = 1
        x = torch.rand(b, 3, 256, 256) * 2 - 1
        tt = torch.full((b,), float(t_int))
        if kind == "reference":
            mod = ref_loader.guided_unet(IMAGENET_CFG, sd)
            net = lambda xx: mod(xx, tt)
        else:
            cfg = og.parse_guided_config(IMAGENET_CFG)
            net = lambda xx: og.guided_unet_forward(sd, cfg, xx, tt)
    else:
        sd = synth.synth_state_dict(ncsnpp.param_shapes(ncsnpp.parse_config(CIFAR_CFG)), seed)
        b = 4
        x = torch.rand(b, 3, 32, 32) * 2 - 1
        tt = torch.full((b,), 99.9)
        if kind == "reference":
            mod = ref_loader.ncsnpp(CIFAR_CFG, sd)
            net = lambda xx: mod(xx, tt)
        else:
            cfg = on.parse_ncsnpp_config(CIFAR_CFG)
            net = lambda xx: on.ncsnpp_forward(sd, cfg, xx, tt)
    adjoint = workload.endswith("_adjoint")

    def fwd():
        with torch.no_grad():
            net(x)

    def fb():       # the adjoint solve costs one forward + one input-gradient pass per step (autograd on the CPU)
        xr = x.clone().requires_grad_(True)
        out = net(xr)
        torch.autograd.grad(out, xr, torch.ones_like(out))

    def timed(f, budget_s):
        calls, t0 = 0, time.time()
        while calls < 1 or (time.time() - t0 < budget_s and calls < 64):
            f()
            calls += 1
        return (time.time() - t0) / (calls * b), calls, time.time() - t0

    out_ = sys.stdout       # the tagged lines' pipe; every other print of this process (libraries, the reference modules) goes to stderr
    sys.stdout = sys.stderr
    fwd()           # warm-up (oneDNN primitive creation)
    if adjoint:
        fb()
    print("CPUWORKER READY " + json.dumps(dict(kind=kind, batch=b, threads=len(cores))), file=out_, flush=True)
    for line in sys.stdin:
        parts = line.split()
        if not parts or parts[0] == "QUIT":
            break
        if parts[0] != "GO":
            continue
        budget = float(parts[1])
        s_fwd, calls, el = timed(fwd, budget * (0.4 if adjoint else 1.0))
        rec = dict(s_fwd=s_fwd, calls=calls, cpu_s=el, batch=b, threads=len(cores), kind=kind)
        if adjoint:
            s_fb, calls2, el2 = timed(fb, budget * 0.6)
            rec.update(s_fb=s_fb, calls_fb=calls2, cpu_s=el + el2)
        print("CPUWORKER RES " + json.dumps(rec), file=out_, flush=True)


def _host_mem_available_gb():
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                return float(line.split()[1]) / 1e6
    except Exception:
        pass
    return None


def cpu_baseline(workload, t_int, n_steps, seed, budget_s=6.0, start_timeout=420.0):
    """The reference's PyTorch score network (oracle/_ref; else the oracle restatement) on THIS BOX'S HOST CORES.  Workers of
    CPU_WORKER_THREADS threads (the size at which one forward is fastest), each a process pinned to its own block of cores, are
    started ONCE; then k = 1, 2, 4, 8, 16 of them run at once for ~budget_s each and their rates add up (independent images; the sweep
    stops after two points in a row below the best so far).  The
    reported value is the BEST aggregate over k (`host.sweep` holds every point), extrapolated to images/s of the full purification
    (the UNet call is > 99.9 % of a step).  Never raises: a worker that fails or hangs is killed and the failure is recorded in
    `sample` - the GPU measurement this line belongs to is already done."""
    import select
    import subprocess
    unit, procs = "images/s", []
    try:
        usable = sorted(os.sched_getaffinity(0))
        per = max(1, min(int(os.environ.get("DIFFPURE_CPU_WORKER_THREADS", str(CPU_WORKER_THREADS))), len(usable)))
        guided = workload.startswith("imagenet256_guided")
        nw = max(1, min(len(usable) // per, int(os.environ.get("DIFFPURE_CPU_WORKERS", str(CPU_SWEEP[-1])))))
        mem = _host_mem_available_gb()
        if mem is not None:      # a guided worker holds the 2.2 GB fp32 network + autograd tape of one image (~8 GB with headroom)
            nw = max(1, min(nw, int(mem // (8.0 if guided else 2.0))))
        adjoint = workload.endswith("_adjoint")
        for w in range(nw):
            cores = ",".join(str(c) for c in usable[w * per:(w + 1) * per])
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", cores, "--workload", workload, "--t", str(t_int),
                                           "--seed", str(seed)], stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, bufsize=1))

        pending = {}        # worker -> bytes read from its pipe that are not a complete line yet

        def read_tagged(p_, tag, timeout):
            """next `CPUWORKER <tag> {...}` line of worker p_ within `timeout` seconds, else None.  The pipe is read with os.read on its
            file descriptor into a buffer of our own (round 6, advisor): select() on the descriptor followed by the TextIO wrapper's
            readline() could leave a tagged line inside Python's buffer - flushed together with a stray print of the worker - where
            select() never reports it again."""
            fd, want = p_.stdout.fileno(), "CPUWORKER " + tag + " "
            end = time.time() + timeout
            while True:
                buf = pending.get(p_, b"")
                while b"\n" in buf:
                    line, buf = buf.split(b"\n", 1)
                    pending[p_] = buf
                    text = line.decode("utf-8", "replace")
                    if text.startswith(want):
                        return json.loads(text[len(want):])
                if time.time() >= end:
                    return None
                r, _, _ = select.select([fd], [], [], max(0.0, min(1.0, end - time.time())))
                if not r:
                    if p_.poll() is not None:
                        return None
                    continue
                chunk = os.read(fd, 1 << 16)
                if not chunk:
                    return None
                pending[p_] = pending.get(p_, b"") + chunk

        t_start = time.time()
        ready = [read_tagged(p_, "READY", max(1.0, start_timeout - (time.time() - t_start))) for p_ in procs]
        live = [p_ for p_, r in zip(procs, ready) if r is not None]
        if not live:
            raise RuntimeError("no CPU worker came up")
        kind = next(r for r in ready if r is not None)["kind"]
        rate_of = lambda r: 1.0 / (n_steps * r["s_fwd"] + (n_steps * r["s_fb"] if adjoint else 0.0))
        sweep, best, stopped = [], None, False
        for k in [k for k in CPU_SWEEP if k <= len(live)] or [len(live)]:
            for p_ in live[:k]:
                p_.stdin.write(f"GO {budget_s}\n")
                p_.stdin.flush()
            recs = [read_tagged(p_, "RES", 60.0 + budget_s * (40 if guided else 8)) for p_ in live[:k]]
            recs = [r for r in recs if r is not None]
            if len(recs) < k:      # a worker died or hung: stop the sweep here, keep what was measured
                sweep.append(dict(workers=k, failed=k - len(recs)))
                break
            pt = dict(workers=k, threads_per_worker=per, value=sum(rate_of(r) for r in recs), per_worker=[rate_of(r) for r in recs],
                      calls=recs[0]["calls"], calls_fb=recs[0].get("calls_fb"), cpu_s=sum(r["cpu_s"] for r in recs), batch=recs[0]["batch"])
            sweep.append(pt)
            if best is None or pt["value"] > best["value"]:
                best = pt
            if len(sweep) >= 3 and max(sweep[-1]["value"], sweep[-2]["value"]) < best["value"]:
                stopped = True      # two points in a row below the best: more workers only contend harder (the 16-worker point of the
                break               # guided UNet is 13 minutes of CPU work for the lowest rate of the sweep)
        if best is None:
            raise RuntimeError("no sweep point finished")
        what = ("the reference's own nn.Module (oracle/_ref: a byte-exact copy of guided_diffusion/ + score_sde/models/, digests in "
                "oracle/ref_modules.sha256)" if kind == "reference" else
                "oracle = torch-CPU fp32 restatement of the reference modules (pinned to them by tests/golden); oracle/_ref absent")
        tried = [p_["workers"] for p_ in sweep]
        rates = ", ".join(f"{p_['workers']}: {p_['value']:.4g}" if "value" in p_ else f"{p_['workers']}: failed" for p_ in sweep)
        return dict(value=best["value"], unit=unit, cores=best["workers"] * per, kind=kind,
                    # scalars for the driver's parser: how many concurrent workers were tried, and the host's core count - `cores` above is the
                    # cores of the BEST point, more workers were tried and lost (they contend for memory bandwidth)
                    cpu_workers_tried=",".join(str(k) for k in tried), cpu_workers_best=best["workers"], host_cpu_count=os.cpu_count(),
                    host={"cpu_count": os.cpu_count(), "usable_cores": len(usable), "workers_started": len(live), "threads_per_worker": per,
                          "mem_available_gb": mem, "sweep": sweep, "sweep_stopped_after_two_declining_points": stopped},
                    sample=(f"best of {tried} concurrent workers x {per} threads on a {os.cpu_count()}-core host ({unit} per point - {rates}; more "
                            f"workers were tried and lost{', the sweep stopped after two declining points' if stopped else ''}): {best['workers']} worker(s); each: "
                            f"{best['calls']} UNet forward(s)" + (f" + {best['calls_fb']} forward+input-gradient pass(es)" if adjoint else "") +
                            f" at batch {best['batch']}, {sum(p_.get('cpu_s', 0.0) for p_ in sweep):.0f} s of CPU work over the sweep; x{n_steps} steps" +
                            (f" + {n_steps} adjoint steps" if adjoint else "") + " extrapolated; " + what))
    except Exception as e:      # never lose the GPU line to the CPU leg
        return dict(value=None, unit=unit, cores=0, kind="port", sample=f"cpu_baseline failed: {type(e).__name__}: {e}")
    finally:
        for p_ in procs:
            try:
                if p_.poll() is None:
                    p_.stdin.write("QUIT\n")
                    p_.stdin.flush()
            except Exception:
                pass
        for p_ in procs:
            try:
                p_.wait(timeout=5)
            except Exception:
                p_.kill()


def timed_region(a, world, one_call, fence, before_first=None, after_first=None, use_dist=None):
    """W untimed warm-up calls, then exactly K timed calls bracketed by fence() on both sides; -> (seconds as the MAX over
    ranks, result of the last call, THIS rank's seconds).  Shared by the real bench and the CPU harness test."""
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
    el_local = el = time.time() - t0
    if world > 1 if use_dist is None else use_dist:
        t = torch.tensor([el], dtype=torch.float64, device=y.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = t.item()
    return el, y, el_local


def rank_report(el_local, steps, sclk_median, all_gather_ms, build_s, world, use_dist, device):
    """So that the first real multi-GPU run explains itself: every rank's own ms per step (the line's `ms_per_step` is their MAX),
    the shader clock each rank's GPU held (ranks of one node share a power / thermal envelope), the time each rank spent inside the
    all-gather (which includes waiting for the slowest rank to arrive) and its engine build time - gathered onto every rank."""
    mine = [el_local / steps * 1e3, -1.0 if sclk_median is None else float(sclk_median), -1.0 if all_gather_ms is None else float(all_gather_ms),
            float(build_s)]
    rows = [mine]
    if use_dist:
        t = torch.tensor(mine, dtype=torch.float64, device=device)
        allr = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allr, t)
        rows = [[float(v) for v in r.tolist()] for r in allr]
    opt = lambda v: None if v < 0 else v
    return {"ms_per_step_per_rank": [r[0] for r in rows], "sclk_mhz_median_per_rank": [opt(r[1]) for r in rows],
            "all_gather_ms_per_step_per_rank": [opt(r[2]) if r[2] < 0 else r[2] / steps for r in rows],
            "engine_build_s_per_rank": [r[3] for r in rows]}


def stub_main(a, rank, world):
    """--stub-engine: the distributed harness of this file (rank / world from the launcher's environment, per-rank batch,
    global sample indices, all_gather of the shards inside the timed region, barrier + MAX-over-ranks timing, the per-rank report,
    one JSON line from rank 0) on CPU ranks over gloo.  The stand-in "purifies" x -> 0.5 x + (global sample index), so the gathered
    result is checkable."""
    if world > 1:
        dist.init_process_group("gloo")
    B, hw = a.batch or 4, 8
    gen = torch.Generator().manual_seed(a.seed + rank)
    x = torch.rand(B, 3, hw, hw, generator=gen) * 2 - 1
    gathered = torch.empty((world * B, 3, hw, hw)) if world > 1 else None
    ag = [0.0]

    def one_call(i):
        if a.stub_slow_rank == rank and a.stub_sleep > 0:
            time.sleep(a.stub_sleep)            # (tests: the reported time must be the MAX over ranks)
        idx = torch.arange(rank * B, rank * B + B, dtype=torch.float32).view(-1, 1, 1, 1)
        y = x * 0.5 + idx
        if world > 1:
            t0 = time.perf_counter()
            dist.all_gather_into_tensor(gathered, y)
            if i >= a.warmup:
                ag[0] += (time.perf_counter() - t0) * 1e3
        return y

    def fence():
        if world > 1:
            dist.barrier()

    el, y, el_local = timed_region(a, world, one_call, fence)
    ok = True
    if world > 1:      # every rank sees every shard, in rank order
        ok = bool(torch.equal(gathered[rank * B:(rank + 1) * B], y))
        for r in range(world):
            ok = ok and bool((gathered[r * B:(r + 1) * B].mean(dim=(1, 2, 3)) - torch.arange(r * B, r * B + B)).abs().max() < 1.0)
    rep = rank_report(el_local, a.steps, None, ag[0] if world > 1 else None, 0.0, world, world > 1, torch.device("cpu"))
    if rank == 0:
        print(json.dumps({"metric": "STUB (harness test, not a measurement)", "stub": True, "value": world * B * a.steps / el,
                          "unit": "images/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": el / a.steps * 1e3,
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "stub", "data": "synthetic",
                          "config": {"workload": "stub", "per_gpu_batch": B, "global_batch": world * B}, "gather_ok": ok,
                          "ranks": rep, "collectives": {"all_gather_ms": max([v for v in rep["all_gather_ms_per_step_per_rank"] if v is not None], default=None)}}),
              flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def d2n(c):
    ns = argparse.Namespace()
    for k, v in c.items():
        setattr(ns, k, d2n(v) if isinstance(v, dict) else v)
    return ns


def build_runner(workload, dev, a, rank, B):
    """The drop-in boundary the reference's driver holds (eval_sde_adv.py:44-55, :78): `RevGuidedDiffusion(args, config, device)` /
    `OdeGuidedDiffusion(...)` of runners/, built from the reference's own (args, config) fields, on seeded synthetic weights (no
    checkpoint files here).  args.sample_offset keys this rank's batch by GLOBAL sample index (weak scaling: rank r holds samples
    [r B, (r + 1) B))."""
    from runners.diffpure_ode import OdeGuidedDiffusion
    from runners.diffpure_sde import RevGuidedDiffusion
    guided = workload.startswith("imagenet256_guided")
    args = argparse.Namespace(t=a.t, rand_t=False, t_delta=15, use_bm=False, sample_step=1, log_dir=None, seed=a.seed, dt=a.dt, step_size=a.dt,
                              fix_rand=False, precision=a.precision, synthetic_weights=True, sample_offset=rank * B,
                              score_type="guided_diffusion" if guided else "score_sde")
    config = d2n(dict(data=dict(dataset="ImageNet"), model=dict(IMAGENET_CFG))) if guided else d2n(CIFAR_CFG)
    cls = OdeGuidedDiffusion if workload == "cifar32_ncsnpp_adjoint" else RevGuidedDiffusion
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):        # the constructors print their settings as the reference's do: keep stdout to the one JSON line
        runner = cls(args, config, device=dev)
    return runner


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
    ap.add_argument("--no-resident-call", action="store_true", help="skip the one extra engine call on the resident batch after the timed region")
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
    ap.add_argument("--cpu-worker", default=None, help="INTERNAL (cpu_baseline): run one CPU worker pinned to these cores (comma list)")
    ap.add_argument("--cpu-budget", type=float, default=6.0, help="seconds of CPU work per worker and sweep point of cpu_baseline")
    ap.add_argument("--engine-call", action="store_true",
                    help="time the bare engine loop (Purifier.sde / .ode + .ode_vjp / .sde + .sde_vjp) on a batch resident in HBM - what "
                         "rounds 1-4 reported - instead of the runner boundary")
    a = ap.parse_args()
    if a.cpu_worker is not None:
        return cpu_worker(a.workload, a.t, a.seed, [int(c) for c in a.cpu_worker.split(",")])

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
    from diffpure_amd.sde import sde_schedule

    wl = WORKLOADS[a.workload]
    adjoint = a.workload.endswith("_adjoint")
    sde_adj = a.workload == "imagenet256_guided_sde_adjoint"
    # per-GPU batches of BASELINE.json's configs: ImageNet 64 (configs[2], and 512 sharded 8 ways in configs[3]),
    # CIFAR 256 (configs[1]), CIFAR adjoint 128 (configs[4])
    B = a.batch or DEFAULT_BATCH[a.workload]
    t_build = time.time()
    runner = build_runner(a.workload, dev, a, rank, B)      # the engine (weights packed + uploaded) lives inside the runner's pool
    pur, net = runner.purifier, runner.model
    torch.cuda.synchronize()
    build_s = time.time() - t_build          # synthetic weights + packing + upload: outside the timed region, reported per rank
    n_steps = len(sde_schedule(wl["kind"], a.t, a.dt))
    hw = wl["hw"]
    gen = torch.Generator().manual_seed(a.seed + rank)
    xh = (torch.rand(B, 3, hw, hw, generator=gen) * 2 - 1).pin_memory()   # the batch the caller hands over: pinned HOST memory
    x = xh.to(dev)                                                         # (--engine-call: resident in HBM before timing)
    gathered = torch.empty((world * B, 3, hw, hw), device=dev) if use_dist else None

    cot = torch.randn(B, 3, hw, hw, generator=gen).to(dev) if adjoint else None
    if adjoint:
        net.enable_grad()
    ag_events = []

    def engine_call(i):
        """rounds 1-4: the engine loop itself on the resident batch"""
        seed = a.seed + 1000003 * i
        if sde_adj:       # reverse-SDE solve, then its stochastic adjoint along the same Brownian path
            xf = pur.sde(x, a.t, a.dt, seed=seed, sample0=rank * B)
            return pur.sde_vjp(xf, cot, a.t, a.dt, seed=seed, sample0=rank * B) * pur.diffuse_scale(a.t)
        if adjoint:       # forward ODE solve, then the adjoint solve for dL/dx with a fixed cotangent
            xf = pur.ode(x, a.t, a.dt, seed=seed, sample0=rank * B)
            return pur.ode_vjp(xf, cot, a.t, a.dt) * pur.diffuse_scale(a.t)
        return pur.sde(x, a.t, a.dt, seed=seed, sample0=rank * B)

    def runner_call(i):
        """the drop-in boundary: host batch in, runner.image_editing_sample (bs_id >= 2: no image logging, as in the reference's
        evaluation loop after its first two batches), purified batch / dL/dx on the device out"""
        if adjoint:       # what an adaptive attack does: autograd through the purifier (eval_sde_adv.py via the attack's backward)
            xr = xh.to(dev, non_blocking=True).requires_grad_(True)
            yy = runner.image_editing_sample(xr, bs_id=2 + i)
            (g,) = torch.autograd.grad(yy, xr, cot)
            return g
        with torch.no_grad():
            return runner.image_editing_sample(xh, bs_id=2 + i)

    def one_call(i):
        y = engine_call(i) if a.engine_call else runner_call(i)
        if use_dist:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            dist.all_gather_into_tensor(gathered, y)
            e1.record()
            if i >= a.warmup:
                ag_events.append((e0, e1))
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

    torch.cuda.reset_peak_memory_stats(dev)
    clock.start()
    el, y, el_local = timed_region(a, world, one_call, fence, before_first, after_first, use_dist)
    sclk = clock.stop()
    peak_gib = torch.cuda.max_memory_allocated(dev) / 2.0 ** 30
    gather_ok = bool(torch.equal(gathered[rank * B:(rank + 1) * B], y)) if use_dist else None
    ag_ms = sum(e0.elapsed_time(e1) for e0, e1 in ag_events) if use_dist else None
    ranks = rank_report(el_local, a.steps, (sclk or {}).get("median"), ag_ms, build_s, world, use_dist, dev)
    build_all = ranks["engine_build_s_per_rank"]
    prof = ops.prof_collect() if sample else None
    window_ms = win0.elapsed_time(win1) if sample else None
    assert torch.isfinite(y).all()
    # for the record, outside the timed region: the host -> device copy of one pinned batch on its own, and ONE call of the bare engine
    # loop on the resident batch (what rounds 1-4 reported as `value`; the runner boundary adds the copy, the pool lock, the dispatch and
    # the NCHW <-> NHWC passes to it)
    torch.cuda.synchronize()
    t_h = time.time()
    xh.to(dev, non_blocking=True)
    torch.cuda.synchronize()
    h2d_ms = (time.time() - t_h) * 1e3
    resident = None
    if not a.engine_call and not a.no_resident_call:
        t_r = time.time()
        engine_call(a.warmup + a.steps)
        torch.cuda.synchronize()
        resident = B / (time.time() - t_r)

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
                           f"activations and weights, counted vmcnt, one barrier per k-tile); launches with fewer than 256 tiles run conv_igemm_dh, the "
                           f"same wave tiles on 128x256 tiles with four waves per workgroup (booked here too); {passes} x v_mfma_f32_32x32x16_f16 per product)" if a.precision in ("f16", "f16sr") else
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
                    # the same kernel's 1x1 launches counted as well (rocprofv3 books them under the one kernel name: profiles/r06/*_kernel_stats.csv)
                    "time_share_of_step_incl_1x1_launches": (dom["ms"] + prof["pp1x1"]["ms"]) / window_ms,
                    "sampled_step_ms": window_ms,
                    # scalars (the driver's parser keeps scalars of this object, not nested ones): the clock and socket power this run held
                    "sclk_mhz_median": held, "power_w_median": (sclk or {}).get("power_w_median"),
                    "sclk_mhz_first_quarter": (sclk or {}).get("first_quarter_median"), "sclk_mhz_last_quarter": (sclk or {}).get("last_quarter_median"),
                    "temp_c_max": (sclk or {}).get("temp_c_max"),
                    "algorithmic_bytes_per_launch_note": "3x3 launches of the dominant kernel only (the set `achieved` is computed over); the set "
                                                         "`traffic` is measured over has its own key, algorithmic_bytes_per_launch_same_set",
                    "other_kernels_share_of_step": {"3x3 on other tile variants (stem, head, split-K levels)": prof["other3x3"]["ms"] / window_ms,
                                                    "1x1 convolutions / linear": (prof["conv1x1"]["ms"] + prof["pp1x1"]["ms"]) / window_ms},
                })
                dh = prof.get("dh3x3")
                if dh and dh["ms"] > 0:     # launches that leave CUs idle on 256x256 tiles (fewer than 256 of them): the 4-wave 128x256-tile kernel
                    roof["conv_igemm_dh_tflops"] = dh["flop"] / (dh["ms"] * 1e-3) / 1e12
                    roof["conv_igemm_dh_time_share_of_step"] = (dh["ms"] + prof["dh1x1"]["ms"]) / window_ms
                gn = prof.get("gn_apply")
                if gn and gn["ms"] > 0:     # the second kernel of the step: HBM-bound, measured with the same hipEvent pairs
                    gbs = gn["bytes"] / (gn["ms"] * 1e-3) / 1e9
                    roof["second_kernel"] = {"kernel": "GroupNorm-apply (gn_apply_h16 / gn_apply_h2q: normalise + FiLM + SiLU + resample, emits the convolution operand)",
                                             "bound": "hbm", "achieved": gbs, "peak": 8000.0, "unit": "GB/s", "frac": gbs / 8000.0,
                                             "frac_of_measured_copy_rate_6290": gbs / 6290.0, "algorithmic_bytes_per_launch": gn["bytes"] / gn["n"],
                                             "avg_launch_ms": gn["ms"] / gn["n"], "sampled_launches": gn["n"], "time_share_of_step": gn["ms"] / window_ms}
                    roof["gn_apply_tb_per_s"] = gbs / 1e3
                    roof["gn_apply_time_share_of_step"] = gn["ms"] / window_ms
                if a.workload == "imagenet256_guided" and a.precision in ("f16", "f16sr"):
                    busy = pmc_mfma_busy()
                    if busy is not None:
                        roof["mfma_busy_by_pmc"] = busy
                row = pmc_traffic(a.workload, B, a.precision)
                if row is not None:
                    roof.update(traffic_keys(row, prof, n_steps))     # the profiled step = one purification = n_steps UNet calls
                    roof["traffic_detail"] = {k: row[k] for k in row if k not in ("workload", "per_gpu_batch", "precision")}
                else:
                    roof["traffic_note"] = ("no rocprofv3 --pmc pass committed for this (workload, batch, precision): "
                                            "profiles/pmc_traffic.json")
        out = {
            "metric": "purified images/sec (whole node), 256x256 GuidedDiff VP-SDE t*=0.1 100-step"
            if a.workload == "imagenet256_guided" and a.t == 100 and n_steps == 100
            else (f"images/sec with input gradient (whole node), {a.workload}: reverse-SDE purification + stochastic-adjoint dL/dx, t={a.t} "
                  f"{n_steps}-step" if sde_adj else
                  f"images/sec with input gradient (whole node), {a.workload}: ODE purification + adjoint dL/dx, t={a.t} "
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
            "config": {"workload": (f"{a.workload}: reverse VP-SDE purification ({n_steps} Euler-Maruyama steps) + stochastic adjoint along the same "
                                    f"Brownian path for dL/dx ({n_steps} steps, each one taped UNet forward + one input-gradient pass; parameter "
                                    f"adjoints not formed), t*={a.t / 1000:g}, dt={a.dt:g}") if sde_adj else
                                   (f"{a.workload}: probability-flow ODE purification ({n_steps} Euler steps) + continuous-adjoint "
                                    f"backward for dL/dx ({n_steps} steps, each one UNet forward + one input-gradient pass), "
                                    f"t*={a.t / 1000:g}, step={a.dt:g}") if adjoint else
                                   (f"{a.workload}: reverse VP-SDE purification, t*={a.t / 1000:g}, dt={a.dt:g}, "
                                    f"{n_steps} Euler-Maruyama steps, one UNet call per step"),
                       "per_gpu_batch": B, "global_batch": world * B, "image": f"3x{hw}x{hw}",
                       "parallelism": f"batch-sharded x{world}, one all_gather of outputs"},
            "roofline": roof,
            "engine_build_s_per_rank": build_all,
            "ranks": ranks,
            "peak_device_memory_gib": peak_gib,
            "input": ({"timed_call": "engine loop (--engine-call): Purifier.* on a batch resident in HBM", "resident_in_hbm_before_timing": True,
                       "h2d_ms_per_batch_pinned": h2d_ms, "value_if_h2d_were_inside_the_timed_region": images / (el + h2d_ms * 1e-3 * a.steps)}
                      if a.engine_call else
                      {"timed_call": ("runner.image_editing_sample(img) of runners/ (the reference's boundary, eval_sde_adv.py:78): img is a pinned "
                                      "HOST tensor; the host -> device copy, the engine-pool lock, the dispatch and the NCHW <-> NHWC passes are "
                                      "inside the timed region" + (", and so is torch.autograd.grad through the runner (host -> device copy of the "
                                                                   "batch, the adjoint solve, dL/dx left on the device)" if adjoint else "")),
                       "resident_in_hbm_before_timing": False, "h2d_ms_per_batch_pinned": h2d_ms,
                       "value_resident_batch_engine_call": resident,
                       "value_resident_note": "ONE call of the bare engine loop on the batch already in HBM, after the timed region (rounds 1-4 "
                                              "reported this rate as `value`)"}),
        }
        if use_dist:
            ags = [v for v in ranks["all_gather_ms_per_step_per_rank"] if v is not None]
            out["collectives"] = {"backend": dist.get_backend(), "world_size": world, "forced_at_world_1": bool(a.force_dist and world == 1),
                                  "all_gather_into_tensor_on_device": True, "gathered_equals_local_shard": gather_ok,
                                  # hipEvents around the collective on the launch stream, per timed step: the MAX over ranks (a rank that arrives
                                  # early waits here for the slowest one) and the MIN (~ the transfer itself)
                                  "all_gather_ms": max(ags) if ags else None, "all_gather_ms_min_over_ranks": min(ags) if ags else None,
                                  "bytes_per_rank": B * 3 * hw * hw * 4}
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.workload, a.t, n_steps, a.seed, budget_s=a.cpu_budget)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
