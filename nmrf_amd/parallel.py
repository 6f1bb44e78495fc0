"""Multi-GPU story of the hot path: stereo pairs are independent units, so the
batch is split contiguously across one-process-per-GPU ranks (same rule as the reference's
InferenceSampler, nmrf/utils/evaluation.py:61-69); weights are replicated; the ONLY collective is the
gather of the finished disparity maps (RCCL all-gather over xGMI on GPUs, gloo on CPU in the tests).
"""
import torch
import torch.distributed as dist


def shard_range(total, rank, world):
    """Contiguous [begin, end) of `total` units for `rank`; the first total%world ranks get one extra."""
    base, extra = divmod(total, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def gather_disparity(disp_local, total=None, group=None):
    """disp_local [b_local,H,W] on every rank -> [sum b_local, H, W] on every rank, in rank order.
    Equal shards use one all_gather_into_tensor (a single RCCL ring/direct all-gather); ragged shards
    (total % world != 0) fall back to all_gather with padding to the largest shard."""
    if not (dist.is_available() and dist.is_initialized()):
        return disp_local
    world = dist.get_world_size(group)
    if world == 1:
        return disp_local
    sizes = None
    if total is not None:
        sizes = [shard_range(total, r, world) for r in range(world)]
        sizes = [e - b for b, e in sizes]
    if sizes is None or len(set(sizes)) == 1:
        out = disp_local.new_empty((world * disp_local.shape[0],) + tuple(disp_local.shape[1:]))
        if dist.get_backend(group) == "gloo":
            parts = list(out.chunk(world, 0))
            dist.all_gather(parts, disp_local.contiguous(), group=group)
            return torch.cat(parts, 0)
        dist.all_gather_into_tensor(out, disp_local.contiguous(), group=group)
        return out
    mx = max(sizes)
    pad = disp_local.new_zeros((mx,) + tuple(disp_local.shape[1:]))
    pad[: disp_local.shape[0]] = disp_local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([p[:s] for p, s in zip(parts, sizes)], 0)
