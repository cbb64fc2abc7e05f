import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "torch default threads", torch.get_num_threads())
import bench
from oracle import ncsnpp as on
from diffpure_amd import ncsnpp as pn
from diffpure_amd.synth import synth_state_dict
cfg = on.parse_ncsnpp_config(bench.CIFAR_CFG)
sd = synth_state_dict(pn.param_shapes(pn.parse_config(bench.CIFAR_CFG)), 1234)
x = torch.rand(4, 3, 32, 32) * 2 - 1
for th in (8, 16, 32, 64):
    torch.set_num_threads(th)
    with torch.no_grad():
        on.ncsnpp_forward(sd, cfg, x, torch.full((4,), 99.9))
        t0 = time.time(); on.ncsnpp_forward(sd, cfg, x, torch.full((4,), 99.9)); el = time.time() - t0
    print(f"threads={th} ncsnpp B=4 forward {el:.3f}s", flush=True)
    if el > 20: break
