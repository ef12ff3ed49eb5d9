"""World-size-2 (and 3) gloo tests of the batch-sharded sampling path on CPU: shard bounds, padding for uneven
shards, ordering after the all-gather.  The HIP sampler is replaced by a deterministic per-cell stand-in; the
collective logic under test is exactly what runs over RCCL on the GPUs."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from scldm_amd.sampling import sample_latents_sharded, shard_bounds


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _fake_sampler(z0, cond):
    # per-cell, order-preserving stand-in: unconditional rows then "guided" rows
    lab = cond["a"].float().view(-1, 1, 1)
    return torch.cat([z0 * 2.0, z0 * 2.0 + lab], dim=0)


def _worker(rank, world, port, B, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(0)
    z0 = torch.randn(B, 4, 3, generator=g)
    cond = {"a": torch.arange(B)}
    res = sample_latents_sharded(_fake_sampler, z0, cond)
    ref = _fake_sampler(z0, cond)
    out_q.put((rank, bool(torch.equal(res, ref)), tuple(res.shape)))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,B", [(2, 8), (2, 7), (3, 10), (2, 1)])
def test_sharded_sampling_matches_single_process(world, B):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, B, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in results) == list(range(world))
    assert all(ok for _, ok, _ in results)
    assert all(shape == (2 * B, 4, 3) for _, _, shape in results)


def test_shard_bounds_cover_and_balance():
    for n in (0, 1, 7, 8, 8192, 8191):
        for w in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _grad_worker(rank, world, port, bucket_bytes, out_q):
    from scldm_amd.training import allreduce_gradients, grad_buckets
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    shapes = [(7, 5), (3,), (64, 64), (1, 16, 8), (11,)]
    params = [torch.nn.Parameter(torch.zeros(s)) for s in shapes]
    params[1].requires_grad_(False)                      # frozen (pos_embed-like): never touched
    for i, p in enumerate(params):
        if p.requires_grad and not (i == 4 and rank == 1):   # rank 1 has no gradient for the last one
            g = torch.Generator().manual_seed(100 * rank + i)
            p.grad = torch.randn(p.shape, generator=g)
    calls = allreduce_gradients(params, bucket_bytes=bucket_bytes)
    exp = []
    for i, s in enumerate(shapes):
        tot = torch.zeros(s)
        for r in range(world):
            if not (i == 4 and r == 1):
                tot += torch.randn(s, generator=torch.Generator().manual_seed(100 * r + i))
        exp.append(tot / world)
    ok = all(torch.allclose(p.grad, e, atol=1e-6) for i, (p, e) in enumerate(zip(params, exp)) if i != 1) and params[1].grad is None
    out_q.put((rank, ok, calls, len(grad_buckets(params, bucket_bytes))))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,bucket_bytes", [(2, 256 << 20), (2, 4096), (3, 64)])
def test_gradient_allreduce_buckets(world, bucket_bytes):
    """The data-parallel training exchange (scldm_amd.training): flat-bucket all-reduce, mean over ranks."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, world, port, bucket_bytes, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _, _ in results)
    assert all(calls == nb for _, _, calls, nb in results)
    assert results[0][3] == (1 if bucket_bytes > 1 << 20 else results[0][3]) and results[0][3] >= 1


def _flat_grad_worker(rank, world, port, out_q):
    from scldm_amd.training import _shared_flat_grad, allreduce_gradients
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    shapes = [(7, 5), (64, 64), (1, 16, 8), (11,)]
    params = [torch.nn.Parameter(torch.zeros(s)) for s in shapes]
    # gradients as views of one buffer with alignment gaps, as the HIP backward returns them
    offs, total = [], 0
    for s in shapes:
        offs.append(total)
        total += (int(torch.Size(s).numel()) + 63) // 64 * 64
    flat = torch.full((total,), float("nan"))            # the gaps hold garbage: they must not leak into any gradient
    for i, (p, o) in enumerate(zip(params, offs)):
        view = flat[o:o + p.numel()].view(p.shape)
        view.copy_(torch.randn(p.shape, generator=torch.Generator().manual_seed(100 * rank + i)))
        p.grad = view
    shared = _shared_flat_grad(params)
    calls = allreduce_gradients(params)
    exp = [sum(torch.randn(s, generator=torch.Generator().manual_seed(100 * r + i)) for r in range(world)) / world for i, s in enumerate(shapes)]
    ok = shared is not None and all(torch.allclose(p.grad, e, atol=1e-6) for p, e in zip(params, exp))
    ok = ok and all(p.grad.untyped_storage().data_ptr() == flat.untyped_storage().data_ptr() for p in params)   # reduced in place
    out_q.put((rank, ok, calls))
    dist.destroy_process_group()


def test_gradient_allreduce_of_a_shared_flat_buffer():
    """When every .grad is a view of one buffer (scldm_amd.nnets._DiTTrainFn.backward), the exchange is ONE in-place all-reduce."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_flat_grad_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in results)
    assert all(calls == 1 for _, _, calls in results)
