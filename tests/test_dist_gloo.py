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


def _run_world(target, world, n):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return sorted(res)


@pytest.mark.parametrize("n", [512 // 32, 11, 3])
def test_sharded_purify_world8_rank_order_ragged_batch_and_gradients(n):
    """BASELINE.json configs[3] shards 8 ways (ref:eval_sde_adv.py:220-228 runs adv_batch_size x ngpus).  Eight CPU ranks over gloo:
    a batch that divides evenly (16 = 2 per rank), a ragged one (11: ranks 0-4 get 2, rank 5 one, ranks 6-7 none and contribute
    padding only), and fewer images than ranks (3) - shards reassembled in rank order, gradients all-gathered back."""
    for rank, ok, shape in _run_world(_worker, 8, n):
        assert ok and shape == (n, 3, 4, 4), (rank, ok, shape)
    for rank, ok_fwd, ok_grad, shape in _run_world(_grad_worker, 8, n):
        assert ok_fwd and ok_grad and shape == (n, 3, 4, 4), (rank, ok_fwd, ok_grad, shape)


def _dtype_worker(rank, world, port, n, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from diffpure_amd import dist as ddist
        x = torch.zeros(n, 3, 4, 4, dtype=torch.float64)
        try:
            ddist.sharded_purify(lambda xl, s0: xl.float(), x)
            q.put((rank, "no error"))
        except ValueError as e:
            q.put((rank, "ValueError" if "float32" in str(e) else str(e)))
    finally:
        dist.destroy_process_group()


def test_sharded_purify_rejects_a_wrong_dtype_on_every_rank_together():
    """The dtype contract is checked on the replicated INPUT before anything rank-specific runs: with 3 ranks and 2 images the
    image-less rank used to raise alone (its result is `x * 1.0` in x's dtype) while the others entered the all-gather and hung."""
    for rank, what in _run_world(_dtype_worker, 3, 2):
        assert what == "ValueError", (rank, what)


def test_shard_bounds():
    from diffpure_amd.dist import shard_bounds
    assert [shard_bounds(512, r, 8)[:2] for r in range(8)] == [(64 * r, 64 * r + 64) for r in range(8)]
    assert [shard_bounds(5, r, 2)[:2] for r in range(2)] == [(0, 3), (3, 5)]
    assert [shard_bounds(1, r, 2)[:2] for r in range(2)] == [(0, 1), (1, 1)]
