"""World-size-2 gloo test of the only exchange step of the path: shard the batch, purify locally,
all_gather the shards (diffpure_amd/dist.py). Uses a CPU stand-in for the per-shard purification
whose output depends on the GLOBAL sample index, exactly like the Philox-keyed engine."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_purify(x, sample0):
    idx = torch.arange(sample0, sample0 + x.shape[0], dtype=torch.float32).view(-1, 1, 1, 1)
    return x * 2 + idx


def _worker(rank, world, port, n, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from diffpure_amd import dist as ddist
        x = torch.arange(n * 3 * 4 * 4, dtype=torch.float32).reshape(n, 3, 4, 4) / 10
        out = ddist.sharded_purify(_fake_purify, x)
        ref = _fake_purify(x, 0)
        q.put((rank, bool(torch.equal(out, ref)), tuple(out.shape)))
    finally:
        dist.destroy_process_group()


def _grad_worker(rank, world, port, n, q):
    """gradients THROUGH a sharded purification: every rank ends up with the full dL/dx of the replicated loss"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from diffpure_amd import dist as ddist

        def purify(x, sample0):           # differentiable, depends on the global sample index like the Philox-keyed engine
            idx = torch.arange(sample0, sample0 + x.shape[0], dtype=torch.float32).view(-1, 1, 1, 1)
            return torch.sin(x) * (1 + idx) + idx

        x = (torch.arange(n * 3 * 4 * 4, dtype=torch.float32).reshape(n, 3, 4, 4) / 10).requires_grad_(True)
        w = torch.linspace(-1, 1, n * 3 * 4 * 4).reshape(n, 3, 4, 4)
        out = ddist.sharded_purify(purify, x)
        (g,) = torch.autograd.grad((out * w).sum(), x)
        xr = x.detach().clone().requires_grad_(True)
        ref = purify(xr, 0)
        (gr,) = torch.autograd.grad((ref * w).sum(), xr)
        q.put((rank, bool(torch.equal(out.detach(), ref.detach())), bool(torch.allclose(g, gr, rtol=0, atol=0)), tuple(g.shape)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n", [4, 5, 1])
def test_sharded_purify_is_differentiable_world2(n):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok_fwd, ok_grad, shape in res:
        assert ok_fwd and ok_grad and shape == (n, 3, 4, 4), (rank, ok_fwd, ok_grad, shape)


@pytest.mark.parametrize("n", [4, 5, 1])
def test_sharded_purify_world2(n):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, shape in res:
        assert ok and shape == (n, 3, 4, 4), (rank, ok, shape)


def test_shard_bounds():
    from diffpure_amd.dist import shard_bounds
    assert [shard_bounds(512, r, 8)[:2] for r in range(8)] == [(64 * r, 64 * r + 64) for r in range(8)]
    assert [shard_bounds(5, r, 2)[:2] for r in range(2)] == [(0, 3), (3, 5)]
    assert [shard_bounds(1, r, 2)[:2] for r in range(2)] == [(0, 1), (1, 1)]
