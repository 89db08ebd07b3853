"""Multi-GPU driver: one process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm).

The path shards trivially: voxels are independent, so the voxel list is cut into blocks of 4 096
voxels dealt round-robin to the ranks (block b -> rank b mod world: background, CSF and white matter cluster in space
and differ 10x in iteration count, interleaving spreads them); the dictionary, penalty and lambda grid (a few MB) are
replicated; there is no exchange during compute.  The only collective is ONE gather of the requested outputs, packed
into a single buffer per rank, to the root rank -- direct peer->root transfers over xGMI, no ring.
"""
import os

import torch
import torch.distributed as dist

BLOCK = 4096


def env_rank():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init(backend=None, single=False):
    """Initialise the default process group from the torchrun environment.  World size 1 needs no group and gets none, unless
    `single` asks for one (a one-rank RCCL group: the communicator and the gather below then really run, on one GPU)."""
    rank, local_rank, world = env_rank()
    if (world > 1 or single) and not dist.is_initialized():
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


def _world():
    """(rank, world) of the initialised process group; a WORLD_SIZE > 1 environment without a process group is an error
    (a silently unsharded result would look like the full array)."""
    _, _, env_world = env_rank()
    if not dist.is_initialized():
        if env_world > 1:
            raise RuntimeError("WORLD_SIZE=%d but torch.distributed is not initialised: call dist.init() first" % env_world)
        return 0, 1
    return dist.get_rank(), dist.get_world_size()


def shard_range(nvox, rank, world):
    """Contiguous block [lo, hi) of rank; blocks differ by at most one voxel (the weak-scaling bench's split)."""
    base, rem = divmod(int(nvox), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_count(nvox, rank, world, block=BLOCK):
    """Number of voxels `shard_indices` gives to `rank`."""
    nvox, world, block = int(nvox), int(world), int(block)
    nblocks = (nvox + block - 1) // block
    mine = (nblocks - rank + world - 1) // world if nblocks > rank else 0
    n = mine * block
    last = nblocks - 1
    if nblocks > 0 and last % world == rank:
        n -= nblocks * block - nvox
    return n


def shard_indices(nvox, rank, world, block=BLOCK, device="cpu"):
    """Voxel indices (ascending int64) of the interleaved blocks owned by `rank`: block b = [b*block, (b+1)*block) -> rank b mod world."""
    nvox, world, block = int(nvox), int(world), int(block)
    nblocks = (nvox + block - 1) // block
    if nblocks <= rank:
        return torch.empty((0,), dtype=torch.int64, device=device)
    b = torch.arange(rank, nblocks, world, dtype=torch.int64, device=device)
    idx = (b.unsqueeze(1) * block + torch.arange(block, dtype=torch.int64, device=device).unsqueeze(0)).reshape(-1)
    return idx[idx < nvox]


def _pack(out, keys, n):
    """Per-voxel outputs -> one [n, W] float64 buffer (maps [6, n] goes in transposed); returns (buffer, widths)."""
    cols, widths = [], []
    for k in keys:
        t = out[k]
        if t is None:
            raise ValueError("output %r was not produced by fit_fn but is in the gather list" % k)
        t = t.to(torch.float64)
        if k == "maps":
            t = t.reshape(t.shape[0], n).t()
        w = 1
        for d in t.shape[1:]:
            w *= int(d)
        cols.append(t.reshape(n, w))           # explicit width: a rank may own no voxel at all (n = 0)
        widths.append(w)
    return torch.cat(cols, dim=1).contiguous(), widths


def fit_sharded(fit_fn, nvox, root=0, gather=("maps", "reg"), block=BLOCK, device=None):
    """Run `fit_fn(idx) -> dict of per-voxel tensors` on this rank's interleaved blocks of the voxel list [0, nvox)
    (idx: ascending int64 voxel indices on `device`) and gather the outputs named in `gather` on `root` with exactly one
    collective.  Returns (local outputs, dict of full arrays on root / None elsewhere); "maps" comes back [6, nvox], the
    rest [nvox, ...] in voxel order."""
    rank, world = _world()
    dev = device if device is not None else (torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu"))
    idx = shard_indices(nvox, rank, world, block, device=dev)
    out = fit_fn(idx)
    n = int(idx.numel())
    buf, widths = _pack(out, gather, n)
    if not dist.is_initialized():
        full = buf
        order = idx
    else:
        counts = [shard_count(nvox, r, world, block) for r in range(world)]
        maxlen = max(counts)
        if n < maxlen:                   # dist.gather needs equal shapes: pad the shorter shards
            buf = torch.cat([buf, torch.zeros((maxlen - n, buf.shape[1]), dtype=buf.dtype, device=buf.device)], dim=0)
        host = dist.get_backend() == "gloo" and buf.is_cuda     # gloo moves host memory only
        send = buf.cpu() if host else buf
        bufs = [torch.empty_like(send) for _ in range(world)] if rank == root else None
        dist.gather(send, bufs, dst=root)                        # the path's single collective
        if rank != root:
            return out, None
        full = torch.cat([bufs[r][: counts[r]] for r in range(world)], dim=0)
        order = torch.cat([shard_indices(nvox, r, world, block, device=full.device) for r in range(world)])
    res_buf = torch.empty_like(full)
    res_buf[order.to(full.device)] = full
    res, c = {}, 0
    for k, w in zip(gather, widths):
        t = res_buf[:, c:c + w]
        c += w
        res[k] = t.t().contiguous() if k == "maps" else (t.reshape(-1) if w == 1 else t.contiguous())
    return out, res
