"""Multi-GPU sampling: one process per GPU (torchrun), independent samples / clip frames sharded over
ranks, ONE collective at the end (RCCL all-gather of the decoded frames over xGMI; gloo on CPU for tests).

The reference has no collective on the sampling path (scripts are single-GPU,
`CUDA_VISIBLE_DEVICES=$1 python ...`, talking_face/sample.sh:27); sharding is legal because no op of the
UNet / VQGAN mixes batch items (GroupNorm and attention are per sample).  Per-item start noise is derived from
(seed, global item index), never from the rank, so results do not depend on the number of GPUs; with
`policy_batch = global batch` the GEMM tile shapes (hence the K-summation order) are pinned too and the
sharded result is bitwise equal to the single-GPU one.
"""
import numpy as np
import torch

from . import switches


def shard_range(n_items, world_size, rank):
    """Contiguous partition: rank r owns [r*ceil(N/G), min(N, (r+1)*ceil(N/G)))  (SURVEY §8e)."""
    per = -(-n_items // world_size)
    lo = min(n_items, rank * per)
    hi = min(n_items, lo + per)
    return lo, hi


def item_noise(seed, index, shape):
    """Start noise of global item `index`: RandomState(seed, index) -> independent of the sharding."""
    rs = np.random.RandomState([seed & 0x7FFFFFFF, int(index)])
    return torch.from_numpy(rs.standard_normal(tuple(shape)).astype(np.float32))


def batch_noise(seed, lo, hi, shape):
    if hi <= lo:
        return torch.empty((0,) + tuple(shape))
    return torch.stack([item_noise(seed, i, shape) for i in range(lo, hi)])


COLLECTIVES_ISSUED = 0        # all_gather_into_tensor calls this process has issued (bench.py reports it; tests count it)


def all_gather_items(local, n_items, group=None):
    """Gather the per-rank item blocks (dim 0) into the full [n_items, ...] tensor on every rank with a single
    all_gather (ranks pad to the common block size; the tail is trimmed).  One rank: nothing to gather -- unless
    LDMK_FORCE_COLLECTIVE is set and a process group exists (bench.py sets it with LDMK_BENCH_FORCE_DIST): then the REAL
    collective is issued over the one-rank group, so that communicator creation and `ncclAllGather` on device tensors next to a
    live hipGraph have run on hardware before the first multi-GPU job does."""
    import torch.distributed as dist
    global COLLECTIVES_ISSUED
    if not (dist.is_available() and dist.is_initialized()):
        return local[:n_items]
    if dist.get_world_size(group) == 1 and not switches.get("LDMK_FORCE_COLLECTIVE"):
        return local[:n_items]
    COLLECTIVES_ISSUED += 1
    world = dist.get_world_size(group)
    per = -(-n_items // world)
    pad = per - local.shape[0]
    if pad > 0:
        local = torch.cat([local, local.new_zeros((pad,) + tuple(local.shape[1:]))])
    out = local.new_empty((world * per,) + tuple(local.shape[1:]))
    dist.all_gather_into_tensor(out, local.contiguous(), group=group)
    return out[:n_items]


def gathered_item_shape(model, shape, decode=True, postprocess=True):
    """Per-item shape of what `sample_sharded` gathers, from the latent shape and the first stage alone: the latent
    itself, the decoded frame (out_ch, H*f, W*f) with f = 2^(levels-1) of the VQGAN decoder, or its channels-last
    [0,1] post-processed form (sample_affectnet.py:127,132)."""
    c, h, w = (int(v) for v in shape)
    if not decode:
        return (c, h, w)
    dec = model.first_stage_model.decoder
    f = 2 ** (dec.num_resolutions - 1)
    return (h * f, w * f, dec.out_ch) if postprocess else (dec.out_ch, h * f, w * f)


@torch.no_grad()
def sample_sharded(sampler, S, n_items, shape, conditioning_fn, seed=0, eta=0.0, decode=True, use_graph=True,
                   rank=0, world_size=1, group=None, postprocess=True, policy="job", _noise_offset=0, _policy_items=None,
                   **sample_kw):
    """Sample `n_items` independent items (class-conditional faces or fixed-identity clip frames) across ranks.

    conditioning_fn(lo, hi) -> conditioning for global items [lo, hi) (tensor or the TF dict).
    Returns the gathered frames (n_items, H, W, 3) in [0,1] (or latents when decode=False) on every rank.
    policy: which batch the launch plans (tile shapes, split-K depths, Winograd / phase routes, pre-split tiles) are made for.
      "job"   -- the whole job's n_items on every rank: the K-summation orders are those of the 1-GPU run, so the gathered
                 result equals it BIT FOR BIT whatever the rank count (stricter than the task's fp32 tolerance) -- at the price
                 of running a 16-frame shard with the plans tuned for 128 frames;
      "shard" -- the rank's own ceil(n_items / world_size) items: every rank runs the plans tuned for ITS batch; the result
                 equals the 1-GPU run within the sampling tolerance (same arithmetic, other summation orders: 1.5e-4 on a
                 short trajectory, tests/test_sampling_gpu.py) and still does not depend on which rank computed an item
                 beyond that.  What `bench.py --gpus N` times.
    """
    from . import ops
    assert policy in ("job", "shard"), policy
    lo, hi = shard_range(n_items, world_size, rank)
    model = sampler.model
    dev = model.device
    if policy == "shard":
        policy = -(-n_items // world_size)     # every rank plans for the common block size (the last block may be shorter)
    else:
        policy = _policy_items or n_items      # (test hook: emulate one rank's block of a larger job)
    if hi > lo:
        x_T = batch_noise(seed, _noise_offset + lo, _noise_offset + hi, shape).to(dev)
        z, _ = sampler.sample(S, hi - lo, list(shape), conditioning_fn(lo, hi), eta=eta, x_T=x_T, verbose=False,
                              use_graph=use_graph, policy_batch=policy, **sample_kw)
        if decode:
            model.first_stage_model.policy_batch = policy
            out = model.decode_first_stage(z)
            out = ops.postprocess_frames(out) if postprocess else out
        else:
            out = z
    else:
        out = None
    if out is None:
        # an idle rank (more ranks than items) still contributes a block of the common shape to the ONE collective of
        # the job; that shape follows from (shape, first-stage factor) alone, so no probe collective is needed
        out = torch.zeros((0,) + gathered_item_shape(model, shape, decode, postprocess), device=dev)
    return all_gather_items(out, n_items, group=group)
