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
