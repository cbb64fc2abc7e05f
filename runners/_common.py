"""Shared pieces of the drop-in runners: the logging side effects of the reference's image_editing_sample (kept
behind the same `bs_id < 2` guard), the engine pool that lets the unchanged driver's nn.DataParallel work, and the
dispatch of one purification over the ranks / the local GPU."""
import os
import random
import threading

import torch

from diffpure_amd import dist as ddist


def out_dir_for(args, bs_id, tag):
    if tag is None:
        tag = "rnd" + str(random.randint(0, 10000))
    log_dir = getattr(args, "log_dir", None)
    return None if log_dir is None else os.path.join(log_dir, "bs" + str(bs_id) + "_" + tag)


def save_image(x, path):
    """(x+1)/2 PNG via torchvision when it is installed; always the raw tensor next to it."""
    torch.save(x.detach().cpu(), os.path.splitext(path)[0] + ".pth")
    try:
        import torchvision.utils as tvu  # not installed in the ROCm image; optional
    except Exception:
        return
    tvu.save_image((x + 1) * 0.5, path)


def as_nchw(x, nhwc):
    return x.permute(0, 3, 1, 2) if nhwc else x


def as_nchw_shape(shape, nhwc):
    return (shape[0], shape[3], shape[1], shape[2]) if nhwc else tuple(shape)


def _dev_key(device):
    device = torch.device(device)
    if device.type == "cuda" and device.index is None:
        return ("cuda", torch.cuda.current_device())
    return (device.type, device.index)


class EnginePool:
    """One resident purification engine per GPU of this process.

    The reference driver wraps the whole defence in `nn.DataParallel` when several GPUs are visible
    (/root/reference/eval_sde_adv.py:227-228): `replicate()` shallow-copies the runner module for every device and
    calls `forward` from one thread per replica with that device's slice of the batch.  The engines here are not
    nn.Parameters (their weights are packed, kernel-specific panels), so replication cannot move them; instead every
    replica shares THIS pool (a plain attribute survives the shallow copy) and asks it for the engine of the device its
    input slice lives on.  The first call on a new device builds that device's engine (weights packed and uploaded
    once, then resident - no per-call broadcast as DataParallel does for module parameters)."""

    def __init__(self, build, home):
        self._build, self._lock, self._by_dev = build, threading.Lock(), {}
        self._calls_by_dev, self._locks = {}, {}
        self.home = torch.device(home)
        self.get(self.home)

    # -- noise keys under replication ---------------------------------------------------------------------------------
    # DataParallel re-replicates the runner on EVERY forward (replica.__dict__ = module.__dict__.copy()), so a counter kept
    # as a plain attribute of the runner is bumped on a throw-away copy and the original stays at 0: every call would draw
    # the same Philox path.  The counters therefore live HERE (the pool object is shared by all replicas), one per device:
    # each replica's forward runs once per DataParallel call, so device d's counter is the number of calls d has served.
    def next_call(self, device):
        """-> the index of this purification call on `device` (0, 1, 2, ...), and count it"""
        key = _dev_key(device)
        with self._lock:
            k = self._calls_by_dev.get(key, 0)
            self._calls_by_dev[key] = k + 1
            return k

    @property
    def calls(self):
        with self._lock:
            return self._calls_by_dev.get(_dev_key(self.home), 0)

    @calls.setter
    def calls(self, value):
        with self._lock:
            for key in list(self._by_dev):
                self._calls_by_dev[key] = int(value)

    def lock(self, device):
        """One purification at a time per engine: an engine owns scratch buffers, a time-table cache and the re-rounded weight
        panels of the call in flight.  DataParallel replicas on DIFFERENT GPUs hold different engines and never contend; two
        callers that alias one GPU (device_ids=[0, 0], or user threads) take turns."""
        key = _dev_key(device)
        with self._lock:
            return self._locks.setdefault(key, threading.RLock())

    def replica_offset(self, device):
        """First sample index a replica on `device` adds to its (local) batch positions: 0 on the home device (a runner that
        is not replicated keys sample i of its batch as i), (1 + GPU index) << 32 on every other GPU - replicas of one
        DataParallel call each see batch positions 0.. of THEIR slice, and without this GPU0's image i and GPU1's image i
        would share one noise path.  A pure function of the device (NOT of which engines the pool has built so far: the
        engines are built lazily from the replica threads, in whatever order those threads get there), so the offsets
        of a GPU are the same on the first call and on every later one."""
        key = _dev_key(device)
        if key == _dev_key(self.home):
            return 0
        return (1 + (key[1] or 0)) << 32

    def get(self, device):
        key = _dev_key(device)
        with self._lock:
            if key not in self._by_dev:
                self._by_dev[key] = self._build(torch.device(key[0], key[1]) if key[1] is not None else torch.device(key[0]))
            return self._by_dev[key]

    def for_input(self, img):
        """the engine of the GPU `img` lives on (DataParallel replica), else the home engine"""
        if isinstance(img, torch.Tensor) and img.is_cuda and _dev_key(img.device) != _dev_key(self.home):
            return self.get(img.device)
        return self.get(self.home)

    def devices(self):
        with self._lock:
            return sorted(self._by_dev)


def build_purifier(args, config, device):
    """The model-construction half of the sde / ode / ldsde runners' constructors (reference diffpure_sde.py:151-195):
    dataset -> score network, checked against args.score_type.  -> Purifier (with .net, .kind, .img_shape)."""
    from diffpure_amd import factory
    from diffpure_amd.sde import Purifier
    want = factory.SCORE_TYPE_TO_KIND.get(args.score_type)
    if want is None:
        raise NotImplementedError(f"Unknown score type in RevVPSDE: {args.score_type}!")
    net, kind, img_shape = factory.build_for_dataset(args, config, device)
    if want != kind:
        raise ValueError(f"score_type {args.score_type} does not match dataset {config.data.dataset}")
    pur = Purifier(net, kind, device)
    pur.img_shape = img_shape
    return pur


def sample_offset(args):
    """First GLOBAL sample index of this process's batch when the call is NOT batch-sharded.  The Philox stream is keyed
    by (args.seed + call counter, global sample index, step): ranks of a multi-process evaluation that each purify their
    own images share the seed and the call count, so their sample indices must differ or they would draw identical noise
    paths.  `args.sample_offset` overrides; default = rank * 2^40 (0 in a single process).  NOTE: the noise is a pure
    function of args.seed - pass a different seed per program run if independent randomness across runs is wanted
    (upstream draws from the global RNG; a missing or zero args.seed here means the SAME noise on every run)."""
    off = getattr(args, "sample_offset", None)
    if off is not None:
        return int(off)
    rank, ws = ddist.world()
    return rank << 40 if ws > 1 else 0


def dispatch(args, run, x, replica_offset=0):
    """One purification of the batch `x`: sharded over the ranks (`args.shard_batch`) or on this process's GPU
    (`replica_offset`: EnginePool.replica_offset of the GPU a DataParallel replica runs on)."""
    if getattr(args, "shard_batch", False):
        return ddist.sharded_purify(run, x)
    if x.shape[0] == 0:
        # an empty batch (the tail slice nn.DataParallel hands a replica when there are fewer images than GPUs never gets here -
        # scatter drops it - but a caller's empty evaluation batch does): the reference's solvers return an empty tensor of the
        # input's shape; nothing is launched (the kernels refuse B = 0 loudly)
        return x * 1.0
    return run(x, sample_offset(args) + replica_offset)


class PooledRunner:
    """Mixin of the drop-in runners: `_calls` (the purification-call counter the noise seed is derived from) is a view of
    the shared EnginePool's counter, so it survives nn.DataParallel's per-forward replication (see EnginePool)."""

    @property
    def _calls(self):
        return self._pool.calls

    @_calls.setter
    def _calls(self, value):
        self._pool.calls = value
