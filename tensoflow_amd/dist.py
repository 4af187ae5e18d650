"""Ray / point sharding and the one exchange step of the path.

The reference is single-process (its `multi_gpus` flag is dead, train/trainer_inv.py:29,188).  Here one
process drives one GPU; units (rays in the shape stage, surface points in the material stage, pixels
in nvs) are independent in the forward pass, so each rank takes a disjoint slice of the shuffled
table and parameters are replicated.  The only collective is the per-step all-reduce of parameter
gradients (RCCL over xGMI when the backend is "nccl"; "gloo" in the CPU tests).
"""
import torch
import torch.distributed as dist


def shard_range(n_units, rank, world):
    """Contiguous, disjoint, exhaustive split of [0, n_units): sizes differ by at most one."""
    base, rem = divmod(int(n_units), int(world))
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def shard_batch(batch_start, batch_size, rank, world):
    """Slice [batch_start, batch_start+batch_size) of the (already shuffled) ray table for `rank`
    (the reference slices `train_batch_i : train_batch_i + rn`, shapeRenderer.py:777-781, materialRenderer.py:540)."""
    lo, hi = shard_range(batch_size, rank, world)
    return batch_start + lo, batch_start + hi


def allreduce_gradients(params, world=None, bucket_bytes=64 << 20, average=True):
    """Sum (then average) .grad of `params` across ranks in a few large flat buckets.
    Large buckets suit xGMI: RCCL rings are per-link bound (7 x ~153 GB/s), so fewer, bigger collectives win.
    Parameters whose grad is None on every rank (e.g. the frozen `*_copy` flows, fields.py:1054-1065) are skipped
    consistently because the decision only depends on the parameter list, not on rank-local state: a missing grad
    is treated as zeros."""
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    params = [p for p in params if p.requires_grad]
    if world == 1 or not params:
        return 0
    n_coll = 0
    bucket, size = [], 0

    def flush():
        nonlocal bucket, size, n_coll
        if not bucket:
            return
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in bucket])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        if average:
            flat /= world
        off = 0
        for p in bucket:
            n = p.numel()
            g = flat[off:off + n].view_as(p)
            if p.grad is None:
                p.grad = g.clone()
            else:
                p.grad.copy_(g)
            off += n
        n_coll += 1
        bucket, size = [], 0

    for p in params:
        nbytes = p.numel() * p.element_size()
        if bucket and (size + nbytes > bucket_bytes or p.dtype != bucket[0].dtype):
            flush()
        bucket.append(p)
        size += nbytes
    flush()
    return n_coll


def gather_rows(local_rows, n_total, rank, world):
    """Inference: all-gather per-rank row blocks (image tiles) back into [n_total, k] on every rank."""
    if world == 1:
        return local_rows
    sizes = [shard_range(n_total, r, world) for r in range(world)]
    maxn = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros(maxn, *local_rows.shape[1:], dtype=local_rows.dtype, device=local_rows.device)
    pad[: local_rows.shape[0]] = local_rows
    outs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(outs, pad)
    return torch.cat([o[: hi - lo] for o, (lo, hi) in zip(outs, sizes)], 0)
