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


def allreduce_gradients(params, world=None, bucket_bytes=64 << 20, average=True, mode="auto", stats=None):
    """Sum (then average) .grad of `params` across ranks in a few large flat buckets.
    Large buckets suit xGMI: RCCL rings are per-link bound (7 x ~153 GB/s), so fewer, bigger collectives win.

    mode "rs_ag": every bucket is reduce-scattered (each rank owns 1/world of it), scaled, then all-gathered -- the split
    SURVEY.md 8(e) sizes for the 7 direct links (the 1/world scaling runs on the owned slice only).  mode "allreduce": one
    all_reduce per bucket.  "auto" = rs_ag where the backend has reduce_scatter_tensor (nccl = RCCL), all_reduce under gloo.

    Parameters whose grad is None on THIS rank but not on others would desynchronise the replicas' Adam states; a parameter
    therefore takes part iff `requires_grad` (a decision every rank makes identically from the parameter list) and a missing
    grad counts as zeros -- callers that must not step untouched parameters (flows before nis_loss_iter, fields.py:1284) pass only
    the parameters that receive gradients at this step (MaterialTrainer.trainable(step)).

    stats (dict, optional): accumulates 'bytes', 'collectives' and -- on CUDA tensors -- a list of (start, end) events under
    'events' so that the caller can report the collective's time and bus bandwidth without a host sync inside the step."""
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    params = [p for p in params if p.requires_grad]
    if world == 1 or not params:
        return 0
    if mode == "auto":
        mode = "rs_ag" if dist.get_backend() == "nccl" else "allreduce"
    n_coll = 0
    bucket, size = [], 0

    def flush():
        nonlocal bucket, size, n_coll
        if not bucket:
            return
        n = sum(p.numel() for p in bucket)
        pad = (-n) % world
        parts = [(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in bucket]
        if pad:
            parts.append(torch.zeros(pad, dtype=parts[0].dtype, device=parts[0].device))
        flat = torch.cat(parts)
        ev = None
        if stats is not None and flat.is_cuda:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        if mode == "rs_ag":
            shard = torch.empty(flat.numel() // world, dtype=flat.dtype, device=flat.device)
            dist.reduce_scatter_tensor(shard, flat, op=dist.ReduceOp.SUM)
            if average:
                shard /= world
            dist.all_gather_into_tensor(flat, shard)
            n_coll += 2
        else:
            if flat.is_cuda and dist.get_backend() == "gloo":      # 1-GPU dry runs of the N-rank path: stage through the host
                host = flat.cpu()
                dist.all_reduce(host, op=dist.ReduceOp.SUM)
                flat.copy_(host)
            else:
                dist.all_reduce(flat, op=dist.ReduceOp.SUM)
            if average:
                flat /= world
            n_coll += 1
        if ev is not None:
            ev[1].record()
            stats.setdefault("events", []).append(ev)
        if stats is not None:
            stats["bytes"] = stats.get("bytes", 0) + flat.numel() * flat.element_size()
            stats["collectives"] = stats.get("collectives", 0) + (2 if mode == "rs_ag" else 1)
            stats["mode"] = mode
        off = 0
        for p in bucket:
            k = p.numel()
            g = flat[off:off + k].view_as(p)
            if p.grad is None:
                p.grad = g.clone()
            else:
                p.grad.copy_(g)
            off += k
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
