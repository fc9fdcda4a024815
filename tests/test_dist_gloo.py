"""CPU, world_size 2 (gloo): the N>1 path -- partition, rank-independent noise, single all-gather, and the
bench's max-over-ranks timing reduction.  The sampler itself needs a GPU, so a deterministic stand-in with the
same (per-item independent) contract produces the per-rank blocks."""
import os
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    """A file rendezvous, not a port: a port picked here and released before the children bind it can be taken in between."""
    return os.path.join(tempfile.mkdtemp(prefix="ldmk_rdzv_"), "store")


def _fake_frames(x_T):
    # per-item function (no cross-item mixing), like the real sampler + decoder
    return torch.tanh(x_T * 0.5 + x_T.flatten(1).mean(1).view(-1, 1, 1, 1)).permute(0, 2, 3, 1).contiguous()


def _worker(rank, world, port, n_items, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", init_method=f"file://{port}", rank=rank, world_size=world)
    from dsml_thesis_amd.parallel import all_gather_items, batch_noise, shard_range
    lo, hi = shard_range(n_items, world, rank)
    local = _fake_frames(batch_noise(5, lo, hi, (3, 8, 8))) if hi > lo else torch.zeros(0, 8, 8, 3)
    full = all_gather_items(local, n_items)
    t = torch.tensor([0.010 * (rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        q.put((full, t.item()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_items", [6, 5, 1])
def test_sharded_equals_unsharded_world2(n_items):
    from dsml_thesis_amd.parallel import batch_noise
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_items, q)) for r in range(2)]
    for p in procs:
        p.start()
    full, tmax = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = _fake_frames(batch_noise(5, 0, n_items, (3, 8, 8)))
    assert full.shape == ref.shape and torch.equal(full, ref)
    assert abs(tmax - 0.020) < 1e-12


def _grad_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", init_method=f"file://{port}", rank=rank, world_size=world)
    from dsml_thesis_amd.train import FlatParams, UNetTrainer
    tr = UNetTrainer.__new__(UNetTrainer)          # the data-parallel reduction only touches the flat gradient buffer
    tr.P = FlatParams()
    tr.P.add("w", torch.zeros(1000))
    tr.P.add("b", torch.zeros(7))
    tr.P.finalize("cpu")
    tr.P.grad.copy_(torch.arange(tr.P.grad.numel(), dtype=torch.float32) * (rank + 1))
    tr.all_reduce_grads(world)
    if rank == 0:
        q.put(tr.P.grad.clone())
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_all_reduce_world2():
    """N1 data parallelism: one all-reduce over the flat packed gradient buffer, averaged (DDP semantics)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    g = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert torch.equal(g, torch.arange(g.numel(), dtype=torch.float32) * 1.5)
