"""Sharding of query points across ranks (one process per GPU) and the final gather.

Query points are independent (/root/reference/main_density_est.py:324: nothing crosses points until the
final np.array(...) at :342), so the path shards with no data-path collective: rank r evaluates a
contiguous block of query points and one all-gather assembles the N float64 results.  The proposal
sampler is keyed by the GLOBAL query index, so the result is bit-identical for any rank count.

torch.distributed is plumbing only (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests).
"""
import numpy as np


def shard_bounds(n, world_size, rank):
    """Contiguous block [lo, hi) of rank `rank`: ceil(n / world_size) points per rank, last ranks may be short
    or empty."""
    per = -(-int(n) // int(world_size)) if n > 0 else 0
    lo = min(n, rank * per)
    hi = min(n, lo + per)
    return lo, hi


def world():
    """(rank, world_size) of the default process group, or (0, 1) when torch.distributed is not initialised."""
    try:
        import torch.distributed as dist
    except ImportError:
        return 0, 1
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def all_gather_blocks(local, n, device=None):
    """Assembles the full (n,) float64 result from every rank's contiguous block (see shard_bounds).

    device: the HIP device index this rank computes on (the engine's).  Under the "nccl" backend the gather buffers
    live there — NOT on torch's current device, which is 0 on every rank unless the caller set it, and RCCL refuses
    two ranks on one GPU.  Ignored under "gloo" (host buffers).

    One all_gather of ceil(n / world) float64 per rank — the only collective on the path.  Payloads are tiny
    (1M points = 8 MB in total), so this is latency- not bandwidth-bound; a single direct all-gather is right.
    """
    rank, ws = world()
    if ws == 1:
        return np.asarray(local, dtype=np.float64)
    import torch
    import torch.distributed as dist
    per = -(-int(n) // ws)
    if dist.get_backend() != "nccl":
        device = torch.device("cpu")
    elif device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    else:
        device = torch.device("cuda", int(device))
    buf = torch.zeros(per, dtype=torch.float64, device=device)
    loc = torch.as_tensor(np.asarray(local, dtype=np.float64))
    buf[: loc.numel()] = loc.to(device)
    out = torch.empty(per * ws, dtype=torch.float64, device=device)
    dist.all_gather_into_tensor(out, buf)
    return out.cpu().numpy()[:n].copy()


def sharded_map(points, fn, device=None):
    """Evaluates fn(block, global_index_of_block_start) on this rank's block of `points` and gathers.

    fn returns one float64 per row of its block.  With one rank it is simply fn(points, 0)."""
    points = np.asarray(points)
    n = points.shape[0]
    rank, ws = world()
    lo, hi = shard_bounds(n, ws, rank)
    local = fn(points[lo:hi], lo) if hi > lo else np.zeros(0, dtype=np.float64)
    return all_gather_blocks(local, n, device)


def all_gather_ranges(local, total, axis=1, device=None):
    """Every rank holds the slice shard_bounds(total, world, rank) of an array along `axis`; returns the whole array on
    every rank.  Used when ONE query point's samples are split over ranks (fewer query points than GPUs): the slices are
    ranges of a point's (max, sum) pairs, 16 bytes each — again a single small all-gather."""
    rank, ws = world()
    local = np.asarray(local, dtype=np.float64)
    if ws == 1:
        return local
    import torch
    import torch.distributed as dist
    per = -(-int(total) // ws)
    moved = np.moveaxis(local, axis, 0)                       # (mine, ...)
    rest = moved.shape[1:]
    if dist.get_backend() != "nccl":
        device = torch.device("cpu")
    elif device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    else:
        device = torch.device("cuda", int(device))
    buf = np.zeros((per,) + rest, dtype=np.float64)
    buf[: moved.shape[0]] = moved
    out = torch.empty((ws * per,) + rest, dtype=torch.float64, device=device)
    dist.all_gather_into_tensor(out, torch.from_numpy(buf).to(device))
    out = out.cpu().numpy().reshape((ws, per) + rest)
    parts = []
    for r in range(ws):
        lo, hi = shard_bounds(total, ws, r)
        parts.append(out[r, : hi - lo])
    return np.ascontiguousarray(np.moveaxis(np.concatenate(parts, axis=0), 0, axis))
