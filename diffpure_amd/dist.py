"""Batch sharding of the purification call over the GPUs of one node.

One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the
CPU tests).  Images are independent (GroupNorm and attention are per-sample, the Philox noise is
keyed by the GLOBAL sample index), so the only exchange step of the whole path is ONE
`all_gather_into_tensor` of the purified shard at the end: 50 MB per rank at B=512 / 256x256.
Weights stay resident on every GPU; nothing is broadcast per call - unlike the reference's
`nn.DataParallel` (/root/reference/eval_sde_adv.py:227-228), which re-broadcasts 552.8 M parameters
on every forward.

Gradients: an adaptive attack differentiates through the purifier.  In the replicated-driver model this module
serves (every rank holds the same adversarial batch `x` and computes the same loss on the reassembled output),
the backward pass mirrors the forward: each rank back-propagates ITS slice of dL/dy through its shard (the
adjoint solve on its own GPU) and one more all-gather reassembles the full dL/dx on every rank.
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


def _gather_rows(local, per, n, ws, group):
    """[k <= per, ...] local rows of every rank -> [n, ...] on every rank (tail ranks pad; all padding sits at the end
    because slices are contiguous)."""
    send = local.new_zeros((per,) + tuple(local.shape[1:]))
    send[: local.shape[0]] = local
    if local.is_cuda and dist.get_backend(group) == "gloo":
        # gloo ranks driving GPUs (tests on a single-GPU box: two ranks share the device): stage through the host
        recv_h = torch.empty((ws * per,) + tuple(local.shape[1:]), dtype=local.dtype)
        dist.all_gather_into_tensor(recv_h, send.cpu().contiguous(), group=group)
        recv = recv_h.to(local.device)
    else:
        recv = local.new_empty((ws * per,) + tuple(local.shape[1:]))
        dist.all_gather_into_tensor(recv, send.contiguous(), group=group)
    return recv if ws * per == n else recv[:n].contiguous()


class _TakeShard(torch.autograd.Function):
    """x (full batch, replicated) -> x[lo:hi]; backward: all-gather of the per-rank input gradients."""

    @staticmethod
    def forward(ctx, x, lo, hi, per, ws, group):
        ctx.cfg = (x.shape[0], per, ws, group)
        return x[lo:hi].contiguous()

    @staticmethod
    def backward(ctx, g):
        n, per, ws, group = ctx.cfg
        return _gather_rows(g.contiguous(), per, n, ws, group), None, None, None, None, None


class _GatherShards(torch.autograd.Function):
    """y_local -> y (full batch on every rank); backward: this rank's rows of dL/dy (the loss is replicated: every
    rank holds the same dL/dy, so no reduction is due)."""

    @staticmethod
    def forward(ctx, y_local, lo, hi, per, n, ws, group):
        ctx.cfg = (lo, hi)
        return _gather_rows(y_local, per, n, ws, group)

    @staticmethod
    def backward(ctx, g):
        lo, hi = ctx.cfg
        return g[lo:hi].contiguous(), None, None, None, None, None, None


def sharded_purify(fn, x, group=None):
    """Run `fn(x_local, sample0) -> y_local` on this rank's slice of the batch and reassemble the
    full result on every rank.  `x` is the FULL batch (identical on every rank, as it is when an
    attack drives all ranks with the same adversarial batch); `fn` keeps the shape of its input (a purifier does).
    Differentiable w.r.t. `x` when `fn` is (see the module docstring)."""
    rank, ws = world()
    if ws == 1:
        return fn(x, 0)
    # checked on `x` BEFORE anything rank-specific runs: x is identical on every rank, so every rank raises together (a check on
    # each rank's own result would let the image-less ranks - which return `xl * 1.0` in x's dtype - raise alone while the
    # others enter the all-gather and hang)
    if x.dtype != torch.float32:
        raise ValueError(f"sharded_purify: x must be float32 (got {x.dtype}); pass the batch as float32 on the rank's engine device")
    n = x.shape[0]
    lo, hi, per = shard_bounds(n, rank, ws)
    need_grad = x.requires_grad and torch.is_grad_enabled()
    xl = _TakeShard.apply(x, lo, hi, per, ws, group) if need_grad else x[lo:hi]
    # a rank without images (more ranks than images) contributes padding only: the purified batch has the shape and
    # dtype of the input batch, so nothing has to be run to learn them (and the empty slice keeps the graph connected,
    # so that this rank still takes part in the backward all-gather)
    y = fn(xl, lo) if hi > lo else xl * 1.0
    # every rank must hand the gather a buffer of the same dtype on its own engine's device: a purifier returns float32 on the
    # device it was given its slice on, and an image-less rank contributes `xl * 1.0` - so the contract is on `x` itself
    if y.dtype != torch.float32 or y.device != x.device:
        raise ValueError(f"sharded_purify: fn must keep float32 on the device of its input (got {y.dtype} on {y.device} for x: "
                         f"{x.dtype} on {x.device}); pass x as float32 on the rank's engine device")
    if need_grad:
        return _GatherShards.apply(y, lo, hi, per, n, ws, group)
    return _gather_rows(y, per, n, ws, group)
