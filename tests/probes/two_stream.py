"""GPU probe (measured: 1.01x - no gain; kernels of the two streams do not overlap usefully): does running TWO half-batches on two HIP streams (two host threads, two engines) beat one full batch?
The MFMA-bound convolutions of one stream could overlap the HBM-bound GroupNorm-apply passes of the other.
    python tests/probes/two_stream.py [B per stream] [steps]"""
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

cfg = guided_unet.parse_config(IMAGENET_CFG)
sd = synth.synth_state_dict(guided_unet.param_shapes(cfg), 1234)
nets = [guided_unet.GuidedUNet(cfg, DEV, "f16sr").load_state_dict(sd) for _ in range(2)]
purs = [Purifier(n, "guided", DEV) for n in nets]
x = (torch.rand(2 * B, 3, 256, 256) * 2 - 1).to(DEV)


def one(pur, xs, sample0, stream):
    with torch.cuda.stream(stream):
        return pur.sde(xs, T, 1e-3, seed=1, sample0=sample0)


def timed(fn):
    fn()
    torch.cuda.synchronize()
    t0 = time.time()
    r = fn()
    torch.cuda.synchronize()
    return time.time() - t0, r


main = torch.cuda.current_stream()
t_full, y_full = timed(lambda: purs[0].sde(x, T, 1e-3, seed=1, sample0=0))
s = [torch.cuda.Stream(), torch.cuda.Stream()]


def split():
    out = [None, None]

    def work(i):
        torch.cuda.set_device(0)
        out[i] = one(purs[i], x[i * B:(i + 1) * B], i * B, s[i])

    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for st in s:
        st.synchronize()                 # the joins only mean "enqueued"
    return torch.cat(out)


t_split, y_split = timed(split)
print(f"B={2 * B}, {T} steps: one stream {t_full * 1e3 / T:.1f} ms/step; two streams x B={B}: {t_split * 1e3 / T:.1f} ms/step "
      f"({t_full / t_split:.3f}x); identical results: {torch.equal(y_full, y_split)}")
