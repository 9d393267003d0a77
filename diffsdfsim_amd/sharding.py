"""Scene sharding across ranks: contiguous blocks of the batch axis, no data-path collective;
results are gathered once at the end.  Works with any torch.distributed backend (RCCL on the GPUs, gloo in tests)."""
import torch


def shard_range(n_total, rank, world):
    """Contiguous [lo, hi) of ``n_total`` scenes owned by ``rank`` (sizes differ by at most one)."""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_spec(spec, lo, hi):
    """Slice a BatchEngine spec along the scene axis (meshes are shared and kept whole)."""
    out = {}
    for k, v in spec.items():
        if k in ("meshes", "mesh_vgrad", "no_contact"):
            out[k] = v
        else:
            out[k] = v[lo:hi]
    return out


def gather_scenes(local, n_total, dist=None):
    """All-gather per-scene results [n_local, ...] into [n_total, ...] in scene order (ragged shards padded)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    sizes = [shard_range(n_total, r, world) for r in range(world)]
    nmax = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((nmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    return torch.cat([b[: hi - lo] for b, (lo, hi) in zip(bufs, sizes)])
