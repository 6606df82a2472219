"""Data-parallel inference helpers: one process per GPU, batch sharded
contiguously over the ranks, weights replicated, NO data-path collective
(every formula is independent in eval: BN uses running statistics, attention is
per sample -- SURVEY.md 8e).  The only communication is the optional gather of
the int64 token ids at the end.  Backend "nccl" is RCCL on ROCm; the CPU tests
drive the same code over "gloo"."""
import torch


def shard_bounds(n, rank, world):
    """Contiguous [lo, hi) slice of n items for `rank` (first n % world ranks get one more)."""
    q, r = divmod(n, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def decode_sharded(decode_fn, images, rank=None, world=None, gather=True, pad_token=0):
    """Run `decode_fn(images_shard) -> int64 tokens [b, S_shard]` on this rank's shard.

    With gather=True returns the whole batch's tokens [B, S_max] on every rank
    (rows of shards that stopped earlier are right-padded with `pad_token`):
    greedy early exit is per *batch* in the reference (tfm.py:138-140), so shards
    may run different step counts; tokens up to each row's [s] are identical."""
    import torch.distributed as dist

    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
        rank = dist.get_rank() if dist.is_initialized() else 0
    lo, hi = shard_bounds(images.shape[0], rank, world)
    toks = decode_fn(images[lo:hi]) if hi > lo else torch.zeros((0, 1), dtype=torch.int64, device=images.device)
    if not gather or world == 1:
        return toks
    # exchange shapes, pad to the common width, all_gather, drop padding rows
    shape = torch.tensor([toks.shape[0], toks.shape[1]], dtype=torch.int64, device=toks.device)
    shapes = [torch.zeros_like(shape) for _ in range(world)]
    dist.all_gather(shapes, shape)
    rows = max(int(s[0]) for s in shapes)
    width = max(int(s[1]) for s in shapes)
    buf = torch.full((rows, width), pad_token, dtype=torch.int64, device=toks.device)
    buf[: toks.shape[0], : toks.shape[1]] = toks
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    return torch.cat([o[: int(s[0])] for o, s in zip(out, shapes)], dim=0)


class GradSync:
    """Data-parallel training: bucketed all-reduce-mean of the gradients, overlapped with
    the rest of the backward pass.

    The reference has no distributed path; this is the new exchange step.  One process per GPU, weights
    replicated, every rank runs forward/backward on its own shard of the batch; `loss.mean()` is taken per rank,
    so the data-parallel gradient is the mean of the ranks' gradients.  Set `model.grad_sync = GradSync()` and
    `loss.backward()` returns already-averaged gradients: the backward of `Model` (doc2tex_amd/train.py) hands every
    gradient to `collect` in the order the engine finishes them (decoder first, stem last); they are packed into flat
    buckets on a communication stream -- each copy ordered after its producing kernels by a device event
    (d2t_train_grad) -- and a bucket's all-reduce (RCCL over xGMI, backend "nccl") is issued as soon as it is full,
    while the compute stream is still working down the backbone.  Buckets are large (64 MB default: 224 MB of fp32
    gradients -> 4 collectives) because xGMI rings are per-link bound and RCCL's setup cost per collective is
    microseconds-scale against ~0.5 ms of payload time per bucket.
    """

    def __init__(self, group=None, bucket_bytes=64 << 20, always_reduce=False):
        self.group = group
        self.bucket_bytes = int(bucket_bytes)
        self.always_reduce = bool(always_reduce)  # issue the collectives even in a one-rank group (tests)
        self._comm = None

    def _world(self):
        import torch.distributed as dist
        return dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1

    def plan(self, params):
        """[(start, end)] index ranges over `params` (already in completion order), each <= bucket_bytes unless a
        single tensor is larger."""
        out, start, size = [], 0, 0
        for i, p in enumerate(params):
            n = p.numel() * 4
            if size and size + n > self.bucket_bytes:
                out.append((start, i))
                start, size = i, 0
            size += n
        if size:
            out.append((start, len(params)))
        return out

    def collect(self, fetch, names, params):
        """fetch(name, dst_flat_view) fills dst with that parameter's local gradient on the CURRENT stream.
        Returns the averaged gradients, one tensor per parameter (views into the flat buckets), in input order."""
        import torch.distributed as dist

        world = self._world()
        order = list(range(len(params)))[::-1]  # named_parameters() order is forward order; backward finishes in reverse
        ordered = [params[i] for i in order]
        dev = params[0].device
        use_streams = dev.type == "cuda"
        if use_streams and self._comm is None:
            self._comm = torch.cuda.Stream(device=dev)
        grads = [None] * len(params)
        pending = []
        import contextlib
        ctx = torch.cuda.stream(self._comm) if use_streams else contextlib.nullcontext()
        with ctx:
            for lo, hi in self.plan(ordered):
                total = sum(p.numel() for p in ordered[lo:hi])
                flat = torch.empty(total, dtype=torch.float32, device=dev)
                off = 0
                for j in range(lo, hi):
                    i = order[j]
                    n = params[i].numel()
                    view = flat[off:off + n]
                    fetch(names[i], view)
                    grads[i] = view.view(params[i].shape)
                    off += n
                if world > 1 or (self.always_reduce and dist.is_initialized()):
                    pending.append((dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True), flat))
                else:
                    pending.append((None, flat))
            for work, flat in pending:
                if work is not None:
                    work.wait()
                    flat.mul_(1.0 / world)
        if use_streams:
            torch.cuda.current_stream(dev).wait_stream(self._comm)
            for _, flat in pending:
                flat.record_stream(torch.cuda.current_stream(dev))
        return grads
