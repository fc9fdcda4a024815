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
        q.put((full.numpy(), t.item()))         # by value (see above)
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
    full = torch.from_numpy(full)
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    ref = _fake_frames(batch_noise(5, 0, n_items, (3, 8, 8)))
    assert full.shape == ref.shape and torch.equal(full, ref)
    assert abs(tmax - 0.020) < 1e-12


class _FakeDecoder:
    num_resolutions, out_ch = 3, 3            # f = 4 like the shipped VQGAN (model.py:462-568)


class _FakeFirstStage:
    decoder = _FakeDecoder()
    policy_batch = None


class _FakeModel:
    """Stand-in with the contract sample_sharded uses (the real sampler needs a GPU): per-item independent sampler +
    decoder, `device`, `first_stage_model.decoder.{num_resolutions,out_ch}`."""
    device = torch.device("cpu")
    first_stage_model = _FakeFirstStage()

    def decode_first_stage(self, z):
        return torch.tanh(z).repeat_interleave(4, 2).repeat_interleave(4, 3)


class _FakeSampler:
    model = _FakeModel()

    def sample(self, S, batch_size, shape, conditioning, eta=0.0, x_T=None, **kw):
        assert x_T.shape[0] == batch_size == conditioning.shape[0]
        return x_T * 0.5 + conditioning.view(-1, 1, 1, 1), None


def _sharded_worker(rank, world, port, n_items, decode, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", init_method=f"file://{port}", rank=rank, world_size=world)
    from dsml_thesis_amd import parallel
    calls = []
    for name in ("all_reduce", "all_gather_into_tensor", "all_gather", "broadcast", "reduce_scatter_tensor"):
        real = getattr(dist, name)
        setattr(dist, name, (lambda real, name: lambda *a, **k: (calls.append(name), real(*a, **k))[1])(real, name))
    cond = lambda lo, hi: torch.arange(lo, hi, dtype=torch.float32)
    full = parallel.sample_sharded(_FakeSampler(), 4, n_items, (3, 8, 8), cond, seed=5, decode=decode, postprocess=False,
                                   rank=rank, world_size=world)
    q.put((rank, full.numpy(), calls))      # by value: a shared-memory tensor handle can die with the worker
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_items,decode", [(1, True), (1, False), (5, True)])
def test_sample_sharded_one_collective_also_with_an_idle_rank(n_items, decode):
    """`sample_sharded` itself on two gloo ranks: n_items = 1 leaves rank 1 idle; it must still join the job's ONE
    collective with a block of the right shape (derived from the latent shape and the first-stage factor, no probe)."""
    from dsml_thesis_amd import parallel
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sharded_worker, args=(r, 2, port, n_items, decode, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    x_T = parallel.batch_noise(5, 0, n_items, (3, 8, 8))
    ref, _ = _FakeSampler().sample(4, n_items, (3, 8, 8), torch.arange(n_items, dtype=torch.float32), x_T=x_T)
    if decode:
        ref = _FakeModel().decode_first_stage(ref)
    for rank, full, calls in got:
        full = torch.from_numpy(full)
        assert calls == ["all_gather_into_tensor"], (rank, calls)       # exactly one collective, no shape probe
        assert full.shape == ref.shape and torch.equal(full, ref)


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
        q.put(tr.P.grad.clone().numpy())
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
    g = torch.from_numpy(q.get(timeout=120))
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    assert torch.equal(g, torch.arange(g.numel(), dtype=torch.float32) * 1.5)
