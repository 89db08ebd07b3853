"""Multi-GPU driver: one process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm).

The path shards trivially: voxels are independent, so the gated voxel list is cut
into contiguous blocks, one per rank; the dictionary, penalty and lambda grid (a few MB) are
replicated; there is no exchange during compute.  The only collective is the gather of the output
maps to the root rank -- direct peer->root transfers over xGMI, no ring.
"""
import os

import torch
import torch.distributed as dist


def env_rank():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init(backend=None):
    """Initialise the default process group from the torchrun environment (no-op for world size 1)."""
    rank, local_rank, world = env_rank()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = os.environ.get("MET2_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


def shard_range(nvox, rank, world):
    """Contiguous block [lo, hi) of rank; blocks differ by at most one voxel."""
    base, rem = divmod(int(nvox), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_rows(shard, nvox, root=0, dim=0):
    """Gather row-sharded `shard` (this rank's block along `dim`) into the full array on `root`.
    Returns the full tensor on root, None elsewhere.  One collective (gather) per call."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1:
        return shard
    rank = dist.get_rank()
    sizes = [shard_range(nvox, r, world) for r in range(world)]
    maxlen = max(hi - lo for lo, hi in sizes)
    # equal-size buffers for dist.gather: pad the (at most one row) shorter shards
    sh = shard.movedim(dim, 0).contiguous()
    if sh.shape[0] < maxlen:
        pad = torch.zeros((maxlen - sh.shape[0],) + tuple(sh.shape[1:]), dtype=sh.dtype, device=sh.device)
        sh = torch.cat([sh, pad], dim=0)
    bufs = [torch.empty_like(sh) for _ in range(world)] if rank == root else None
    dist.gather(sh, bufs, dst=root)
    if rank != root:
        return None
    parts = [bufs[r][: sizes[r][1] - sizes[r][0]] for r in range(world)]
    return torch.cat(parts, dim=0).movedim(0, dim)


def fit_sharded(fit_fn, data, fa_index=None, mask=None, root=0, gather=("maps", "reg")):
    """Run `fit_fn(data_block, fa_block, mask_block) -> dict of tensors` on this rank's block of the
    voxel list and gather the requested outputs on `root`.  `data` is the full [nvox, nte] array
    (every rank holds, or can generate, its own block; only the block is touched)."""
    rank, _, world = env_rank()
    nvox = data.shape[0]
    lo, hi = shard_range(nvox, rank, world)
    out = fit_fn(data[lo:hi], None if fa_index is None else fa_index[lo:hi], None if mask is None else mask[lo:hi])
    res = {}
    for k in gather:
        t = out[k]
        res[k] = gather_rows(t, nvox, root=root, dim=1 if k == "maps" else 0)
    return out, res
