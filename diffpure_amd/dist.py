"""Batch sharding of the purification call over the GPUs of one node.

One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the
CPU tests).  Images are independent (GroupNorm and attention are per-sample, the Philox noise is
keyed by the GLOBAL sample index), so the only exchange step of the whole path is ONE
`all_gather_into_tensor` of the purified shard at the end: 50 MB per rank at B=512 / 256x256.
Weights stay resident on every GPU; nothing is broadcast per call - unlike the reference's
`nn.DataParallel` (/root/reference/eval_sde_adv.py:227-228), which re-broadcasts 552.8 M parameters
on every forward.
"""
import torch
import torch.distributed as dist


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_bounds(n, rank, world_size):
    """Contiguous slice [lo, hi) of a batch of n for `rank`; every rank gets ceil(n/world) rows
    except the tail ranks, which may get fewer (possibly zero)."""
    per = (n + world_size - 1) // world_size
    lo = min(n, rank * per)
    return lo, min(n, lo + per), per


def sharded_purify(fn, x, group=None):
    """Run `fn(x_local, sample0) -> y_local` on this rank's slice of the batch and reassemble the
    full result on every rank.  `x` is the FULL batch (identical on every rank, as it is when an
    attack drives all ranks with the same adversarial batch)."""
    rank, ws = world()
    if ws == 1:
        return fn(x, 0)
    n = x.shape[0]
    lo, hi, per = shard_bounds(n, rank, ws)
    if hi > lo:
        y = fn(x[lo:hi], lo)
        out_shape, dtype, device = tuple(y.shape[1:]), y.dtype, y.device
    else:  # more ranks than images: contribute padding only
        probe = fn(x[:1], 0)
        y = probe[:0]
        out_shape, dtype, device = tuple(probe.shape[1:]), probe.dtype, probe.device
    send = torch.zeros((per,) + out_shape, dtype=dtype, device=device)
    send[: hi - lo] = y
    recv = torch.empty((ws * per,) + out_shape, dtype=dtype, device=device)
    dist.all_gather_into_tensor(recv, send, group=group)
    if ws * per == n:
        return recv
    return recv[:n].contiguous()  # all padding sits at the tail because slices are contiguous
