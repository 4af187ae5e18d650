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


class GradientExchange:
    """The per-step gradient exchange with (a) ONE persistent flat buffer per bucket -- every parameter's `.grad` is a VIEW into it, so
    backward accumulates straight into the bucket: no `torch.cat` of ~190 MB per step and no copy back (the first thing an 8-rank run
    of `allreduce_gradients` would have shown) -- and (b) every bucket's collective launched from an autograd hook as soon as its LAST
    gradient has been written (`register_post_accumulate_grad_hook`), asynchronously, under the rest of the backward pass.

    Buckets are filled in REVERSE parameter order (gradients become ready roughly in reverse order of use).  The mean is taken by
    scaling the bucket by 1 / world before the sum, so that reduce-scatter and all-gather ("rs_ag", the default on RCCL: each rank
    reduces 1 / world of the bucket over the 7 direct xGMI links) can both be queued from the hook, back to back.

        ex = GradientExchange(params, world)
        ex.zero_grad()            # instead of optimizer.zero_grad(set_to_none=True): zeroes the buckets, re-attaches the views
        loss.backward()           # hooks fire; complete buckets are already on the wire
        ex.finish(expected)       # buckets whose hooks did not all fire are sent now; waits; parameters outside `expected`
                                  # get grad = None again (the optimizer must not step them; one that DID receive a gradient
                                  # is a broken contract: the hook raises)
    A parameter that receives no gradient on THIS rank but does on another still takes part as zeros, as in allreduce_gradients."""

    def __init__(self, params, world=None, bucket_bytes=64 << 20, mode="auto", stats=None, force_collectives=False):
        if world is None:
            world = dist.get_world_size() if dist.is_initialized() else 1
        self.world = int(world)
        self.stats = stats
        # force_collectives: queue the collectives at world == 1 as well (a one-rank process group must be initialised).  The product
        # never asks for it -- one rank has nothing to exchange -- but it lets ONE GPU run the whole exchange (flat bucket views, hooks
        # firing inside a real backward, reduce-scatter + all-gather on the backend's stream, finish) before an 8-GPU node ever does:
        # tests/test_gpu_dist.py.  At one rank every collective is the identity, so the step must equal the un-exchanged one.
        self.force = bool(force_collectives)
        self.params = [p for p in params if p.requires_grad]
        if mode == "auto":
            mode = "rs_ag" if ((self.world > 1 or self.force) and dist.get_backend() == "nccl") else "allreduce"
        self.mode = mode
        self.buckets = []          # dicts: params, flat, views, shard, pending (hooks still to fire), work
        cur, size = [], 0
        for p in reversed(self.params):
            nbytes = p.numel() * p.element_size()
            if cur and (size + nbytes > bucket_bytes or p.dtype != cur[0].dtype or p.device != cur[0].device):
                self._close(cur)
                cur, size = [], 0
            cur.append(p)
            size += nbytes
        if cur:
            self._close(cur)
        self._bucket_of = {}
        self._hooks = []
        for bi, b in enumerate(self.buckets):
            for si, p in enumerate(b["params"]):
                self._bucket_of[id(p)] = (bi, si)
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
        self._fired = set()
        self._next = 0                         # index of the next bucket to go on the wire (in-order launch)
        self.launched_in_backward = 0          # collectives queued from hooks during the last backward (tests / bench line)
        self._in_backward = False

    def _close(self, plist):
        n = sum(p.numel() for p in plist)
        pad = (-n) % max(self.world, 1)
        flat = torch.zeros(n + pad, dtype=plist[0].dtype, device=plist[0].device)
        views, off = [], 0
        for p in plist:
            views.append(flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        shard = torch.empty(flat.numel() // max(self.world, 1), dtype=flat.dtype, device=flat.device) if self.mode == "rs_ag" else None
        self.buckets.append(dict(params=plist, flat=flat, views=views, shard=shard, pending=len(plist), work=[], sent=False, events=None))

    def same_params(self, params):
        want = [p for p in params if p.requires_grad]
        return len(want) == len(self.params) and all(a is b for a, b in zip(want, self.params))

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []

    def zero_grad(self, expected=None):
        """Zero every bucket and point every parameter's .grad at its slice of the bucket.  `expected`: the parameters that receive a
        gradient at this step (a decision every rank makes identically, e.g. MaterialTrainer.trainable(step)); the others are not
        waited for -- a bucket is complete when its EXPECTED gradients have landed (buckets go out in order, so a parameter that is
        never written would otherwise hold back every bucket behind it until finish())."""
        keep = None if expected is None else {id(p) for p in expected}
        for b in self.buckets:
            b["flat"].zero_()
            b["work"], b["sent"], b["events"] = [], False, None
            b["pending"] = len(b["params"]) if keep is None else sum(1 for p in b["params"] if id(p) in keep)
            for p, v in zip(b["params"], b["views"]):
                p.grad = v
        self._expected = keep
        self._fired = set()
        self._next = 0
        self.launched_in_backward = 0
        self._in_backward = True

    def _send(self, b):
        if b["sent"] or (self.world == 1 and not self.force):
            b["sent"] = True
            return
        flat = b["flat"]
        if self.stats is not None and flat.is_cuda:
            b["events"] = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            b["events"][0].record()
        flat.mul_(1.0 / self.world)
        if self.mode == "rs_ag":
            b["work"].append(dist.reduce_scatter_tensor(b["shard"], flat, op=dist.ReduceOp.SUM, async_op=True))
            b["work"].append(dist.all_gather_into_tensor(flat, b["shard"], async_op=True))
        elif flat.is_cuda and dist.get_backend() == "gloo":       # 1-GPU dry runs of the N-rank path: staged through the host at finish()
            b["work"] = None
        else:
            b["work"].append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True))
        b["sent"] = True
        if self._in_backward:
            self.launched_in_backward += 2 if self.mode == "rs_ag" else 1
        if self.stats is not None:
            self.stats["bytes"] = self.stats.get("bytes", 0) + flat.numel() * flat.element_size()
            self.stats["collectives"] = self.stats.get("collectives", 0) + (2 if self.mode == "rs_ag" else 1)
            self.stats["mode"] = self.mode

    def _on_grad(self, p):
        slot = self._bucket_of.get(id(p))
        if slot is None or id(p) in self._fired:
            return
        self._fired.add(id(p))
        counted = getattr(self, "_expected", None) is None or id(p) in self._expected
        b = self.buckets[slot[0]]
        if not counted:
            # A parameter the caller declared gradient-free at this step (zero_grad(expected=...)) received one.  Its bucket is not
            # held back for it, so the bucket may already be on the wire -- autograd would then be accumulating into a buffer an
            # asynchronous reduce-scatter / all-gather is reading and writing -- and the ranks would step it with un-averaged
            # numbers.  `expected` is a contract every rank evaluates identically: breaking it is an error, not a slow path.
            name = next((n for n, q in getattr(self, "named", {}).items() if q is p), f"shape {tuple(p.shape)}")
            raise RuntimeError(f"GradientExchange: parameter {name} is outside this step's `expected` set but received a gradient"
                               + (" after its bucket was sent" if b["sent"] else ""))
        v = b["views"][slot[1]]
        if p.grad is not v:
            # autograd replaced the view (first accumulation into an undefined / non-writable grad): put the numbers where they belong
            v.copy_(p.grad)
            p.grad = v
        if counted:
            b["pending"] -= 1
        # buckets go on the wire STRICTLY in bucket order (as torch's DDP reducer does): every rank then issues the same sequence of
        # collectives even if its autograd graph completes the buckets in another order (a rank without a single hit ray skips a
        # branch of the graph) -- a bucket that is ready early waits for its predecessors
        while self._next < len(self.buckets) and self.buckets[self._next]["pending"] == 0:
            self._send(self.buckets[self._next])
            self._next += 1

    def finish(self, expected=None):
        """After backward: send the buckets that are still waiting for a gradient (their missing slices are zeros), wait for
        every collective, and give parameters outside `expected` that received nothing their None grad back."""
        self._in_backward = False
        for b in self.buckets[self._next:]:
            self._send(b)
        self._next = len(self.buckets)
        for b in self.buckets:
            if b["work"] is None:                                  # gloo with device tensors: host staging
                host = b["flat"].cpu()
                dist.all_reduce(host, op=dist.ReduceOp.SUM)
                b["flat"].copy_(host)
            else:
                for w in b["work"]:
                    w.wait()
            if b["events"] is not None:
                b["events"][1].record()
                self.stats.setdefault("events", []).append(b["events"])
        if expected is not None:
            keep = {id(p) for p in expected}
            for p in self.params:
                if id(p) not in keep:          # (none of them can have fired: _on_grad raises)
                    p.grad = None
        return sum(1 for b in self.buckets if b["sent"])


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


def rank_world(rank=None, world=None):
    """(rank, world) of this process: the arguments if given, else torch.distributed's, else (0, 1)."""
    on = dist.is_available() and dist.is_initialized()
    if world is None:
        world = dist.get_world_size() if on else 1
    if rank is None:
        rank = dist.get_rank() if on else 0
    return int(rank), int(world)


def gather_maps(maps, n_total, rank, world):
    """Frame assembly of a tiled `nvs` (materialRenderer.py:705-752 chunks a frame in 512-ray pieces on one GPU; here the pieces
    are the multi-GPU unit): every rank holds rows shard_range(n_total, rank, world) of each per-pixel map
    {key: [n_local, C]} (float, or bool masks); ONE all-gather of the channel-concatenated rows returns {key: [n_total, C]} on every
    rank (an 800 x 800 frame with the material stage's 15 maps + hit mask is 32 floats per pixel = 82 MB)."""
    if world == 1:
        return maps
    keys = list(maps)
    widths = [maps[k].shape[1] if maps[k].dim() > 1 else 1 for k in keys]
    lo, hi = shard_range(n_total, rank, world)
    for k in keys:
        if maps[k].shape[0] != hi - lo:
            raise RuntimeError(f"gather_maps: rank {rank} holds {maps[k].shape[0]} rows of `{k}`, its shard of {n_total} is {hi - lo}")
    flat = torch.cat([maps[k].reshape(hi - lo, -1).float() for k in keys], 1)
    if flat.is_cuda and dist.get_backend() == "gloo":        # 1-GPU dry runs of the N-rank path: staged through the host
        full = gather_rows(flat.cpu(), n_total, rank, world).to(flat.device)
    else:
        full = gather_rows(flat, n_total, rank, world)
    out, c0 = {}, 0
    for k, wd in zip(keys, widths):
        v = full[:, c0:c0 + wd]
        c0 += wd
        if maps[k].dtype == torch.bool:
            v = v > 0.5
        else:
            v = v.to(maps[k].dtype)
        out[k] = v if maps[k].dim() > 1 else v[:, 0]
    return out
