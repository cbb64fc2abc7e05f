"""GPU probe: the CIFAR-10 purification (BASELINE configs[1], B = 256) as TWO half-batches on two HIP streams (two host threads, two
engines) against one full batch on one stream.  At CIFAR sizes a convolution launch is 1-2 tiles per CU: ~30 us of every ~80-110 us
launch are dispatch, cold first k-tiles and drain (profiles/r05/cifar_conv_shapes_dh_ab.log vs the steady-state rates), and consecutive
launches of one stream cannot overlap (every GroupNorm needs its whole input).  Two independent streams can fill each other's gaps -
if the host can feed them (two Python threads share one GIL: ~770 launches per UNet call and stream).
    python tests/probes/two_stream_cifar.py [B per stream] [steps] [streams]"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from bench import CIFAR_CFG  # noqa: E402
from diffpure_amd import ncsnpp, synth  # noqa: E402
from diffpure_amd.sde import Purifier  # noqa: E402

DEV = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
T = int(sys.argv[2]) if len(sys.argv) > 2 else 20
NS = int(sys.argv[3]) if len(sys.argv) > 3 else 2

cfg = ncsnpp.parse_config(CIFAR_CFG)
sd = synth.synth_state_dict(ncsnpp.param_shapes(cfg), 1234)
nets = [ncsnpp.NCSNpp(cfg, DEV, "f16sr").load_state_dict(sd) for _ in range(NS)]
purs = [Purifier(n, "ncsnpp", DEV) for n in nets]
x = (torch.rand(NS * B, 3, 32, 32) * 2 - 1).to(DEV)


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    best, r = 1e9, None
    for _ in range(reps):
        t0 = time.time()
        r = fn()
        torch.cuda.synchronize()
        best = min(best, time.time() - t0)
    return best, r


t_full, y_full = timed(lambda: purs[0].sde(x, T, 1e-3, seed=1, sample0=0))
s = [torch.cuda.Stream() for _ in range(NS)]


def split():
    out = [None] * NS

    def work(i):
        torch.cuda.set_device(0)
        with torch.cuda.stream(s[i]):
            out[i] = purs[i].sde(x[i * B:(i + 1) * B], T, 1e-3, seed=1, sample0=i * B)

    th = [threading.Thread(target=work, args=(i,)) for i in range(NS)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for st in s:
        st.synchronize()                 # the joins only mean "enqueued"
    return torch.cat(out)


t_split, y_split = timed(split)
print(f"CIFAR NCSN++ B={NS * B}, {T} steps: one stream {t_full * 1e3 / T:.2f} ms/step ({NS * B * 100 / (t_full / T * 100) / 100:.1f} images/s at 100 steps); "
      f"{NS} streams x B={B}: {t_split * 1e3 / T:.2f} ms/step ({t_full / t_split:.3f}x); identical results: {torch.equal(y_full, y_split)}")
