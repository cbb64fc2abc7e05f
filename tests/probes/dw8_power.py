"""Energy experiment on the dominant convolution shape: the chip is power-limited under the 8-wave kernel (it
holds 1.6-1.8 of 2.4 GHz) while a pure-MFMA loop holds 2.4 GHz, so the clock is lost to MOVING OPERANDS.  Which byte stream buys it
back?  Every variant below runs the 3x3 256->256 convolution at 256^2, B=64 (fp16 residual + fp16 output) back to back for ~4 s while
a host thread samples the shader clock (sysfs pp_dpm_sclk) and the socket power (hwmon power1_average / power1_input) every 50 ms:

  dw            the shipped 8-wave kernel (conv_igemm_dw)
  sw            the one-wave-per-SIMD kernel on the same tiles (DP_H2_DW=0)
  generic       128x128 tiles, two workgroups per CU (DP_H2_PP=0)
  dw -reads     DP_H2_DW_MODE=4   no ds_read_b128 in the k-loop            (WRONG RESULTS: -DDP_ABLATE library only)
  dw -actDMA    DP_H2_DW_MODE=16  no activation LDS-DMA in the steady state (the 9x tap refetch L2 -> LDS)
  dw -wDMA      DP_H2_DW_MODE=32  no weight LDS-DMA in the steady state
  dw -DMA       DP_H2_DW_MODE=1   no LDS-DMA at all in the steady state
  dw -DMA-reads DP_H2_DW_MODE=5   neither: MFMAs + barrier only
  dw -stores    DP_H2_DW_MODE=8   no epilogue

    python tests/probes/build_ablate.py && python tests/probes/dw8_power.py [--seconds 4]
Prints per variant: ms per launch, nominal TFLOP/s, sclk median, power median (W), TFLOP/s per W, and kernel cycles per launch
(ms x sclk) - the last column separates "fewer cycles" from "higher clock"."""
import ctypes
import glob
import os
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from diffpure_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.join(ROOT, "diffpure_amd", "csrc", "libdiffpure_hip_ablate.so")
from diffpure_amd import ops  # noqa: E402

DEV = "cuda:0"


class Sampler:
    def __init__(self):
        # the sysfs node of THE GPU torch runs on (the box exposes several cards; the first attempt of this probe read card0's sensors while
        # the kernels ran on card56): matched by PCI bus id, as bench.py's SclkSampler does
        dev = None
        try:
            bus = torch.cuda.get_device_properties(0).pci_bus_id
            for c in sorted(glob.glob("/sys/class/drm/card*/device")):
                if f":{bus:02x}:" in os.path.realpath(c):
                    dev = c
        except Exception:
            pass
        if dev is None:
            cands = sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"))
            dev = os.path.dirname(cands[0]) if len(cands) == 1 else None
        self.dev = dev
        self.sclk_path = os.path.join(dev, "pp_dpm_sclk") if dev else None
        cands = (sorted(glob.glob(os.path.join(dev, "hwmon/hwmon*/power1_average"))) + sorted(glob.glob(os.path.join(dev, "hwmon/hwmon*/power1_input")))) if dev else []
        self.pow_path = cands[0] if cands else None
        self.all_power = cands
        self.sclk, self.power, self._stop = [], [], threading.Event()
        self._th = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        while not self._stop.is_set():
            try:    # several lines may carry a '*' (a sleep level next to the active one): the active clock is the largest starred level
                vals = [float(line.split(":")[1].lower().replace("mhz", "").replace("*", "").strip()) for line in open(self.sclk_path) if "*" in line]
                if vals:
                    self.sclk.append(max(vals))
            except Exception:
                pass
            try:
                self.power.append(float(open(self.pow_path).read().strip()) / 1e6)
            except Exception:
                pass
            self._stop.wait(0.05)

    def __enter__(self):
        self._th.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        self._th.join()
        return False

    @staticmethod
    def med(v):
        v = sorted(v)
        return v[len(v) // 2] if v else None


def main():
    seconds = float(sys.argv[sys.argv.index("--seconds") + 1]) if "--seconds" in sys.argv else 4.0
    B, H, ci, co = 64, 256, 256, 256
    x = torch.randn(B, H, H, ci)
    w = torch.randn(co, ci, 3, 3) * (1.0 / (9 * ci)) ** 0.5
    wh = ops.order_conv_weight_w16(w).half().to(DEV)
    xh = torch.nn.functional.pad(x, (0, 0, 1, 1, 1, 1)).half().contiguous().to(DEV)
    del x
    bias = torch.randn(co, device=DEV)
    res = torch.randn(B, H, H, co, device=DEV).half()
    M = B * H * H
    out = torch.empty((B, H, H, co), device=DEV, dtype=torch.float16)
    cs = torch.zeros(((M + 511) // 512 * 8, 2, co), device=DEV)
    tr = ctypes.c_int(0)
    s = torch.cuda.current_stream().cuda_stream
    flop = 2.0 * M * co * 9 * ci

    def call():
        _lib.call("dp_conv2d_nhwc_h2", xh.data_ptr(), ci, B, H, H, 3, wh.data_ptr(), co, bias.data_ptr(), None, 0, res.data_ptr(), co, 1.0,
                  out.data_ptr(), co, cs.data_ptr(), ctypes.addressof(tr), None, 0, 1, 1, 1, 1, 1, None, 0, None, 0, s)

    variants = [("dw", {}, "0"), ("sw", {"DP_H2_DW": 0}, "0"), ("generic", {"DP_H2_PP": 0}, "0"), ("dw -reads", {}, "4"), ("dw -actDMA", {}, "16"),
                ("dw -wDMA", {}, "32"), ("dw -DMA", {}, "1"), ("dw -DMA-reads", {}, "5"), ("dw -stores", {}, "8"), ("dw (again)", {}, "0")]
    print(f"3x3 {ci}->{co} at {H}^2, B={B}, fp16 residual + fp16 output; {seconds:.0f} s per variant; power from "
          f"{Sampler().pow_path} (all power sensors: {Sampler().all_power}), clock from {Sampler().sclk_path}")
    try:        # what the vendor tool says at idle, for the units of the sysfs sensor
        import subprocess
        print(subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=30).stdout[-1500:])
    except Exception as e:
        print("rocm-smi unavailable:", e)
    print(f"{'variant':16s} {'ms':>8s} {'TFLOP/s':>8s} {'sclk MHz':>9s} {'power W':>8s} {'TF/W':>6s} {'Mcycles/launch':>15s} {'launches':>8s}")
    for name, tune, mode in variants:
        os.environ["DP_H2_DW_MODE"] = mode
        with ops.tuning(**tune):
            for _ in range(10):
                call()
            torch.cuda.synchronize()
            time.sleep(0.5)
            with Sampler() as smp:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                n, t0 = 0, time.time()
                e0.record()
                while time.time() - t0 < seconds:
                    for _ in range(20):
                        call()
                    n += 20
                    torch.cuda.synchronize()
                e1.record()
                torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / n
            # skip the first 20 % of the samples (the clock settles within ~0.5 s of a load change)
            sk = lambda v: v[len(v) // 5:]
            sclk, pw = Sampler.med(sk(smp.sclk)), Sampler.med(sk(smp.power))
            tf = flop / ms / 1e9
            print(f"{name:16s} {ms:8.3f} {tf:8.0f} {sclk if sclk else float('nan'):9.0f} {pw if pw else float('nan'):8.0f} "
                  f"{(tf / pw) if pw else float('nan'):6.2f} {(ms * 1e-3 * sclk * 1e6 / 1e6) if sclk else float('nan'):15.2f} {n:8d}", flush=True)
    os.environ["DP_H2_DW_MODE"] = "0"


if __name__ == "__main__":
    main()
