"""The criterion of the reference's training step as a fused HIP kernel pair.

Mirrors doc2tex/modules/loss/builder.py:18-24 (`create_criterion("entropy", kwargs)` -> `nn.CrossEntropyLoss(**kwargs)`)
and how engine/training.py:50-53,83,90,126 uses it: `criterion(preds.view(-1, V), target.contiguous().view(-1))` with
`ignore_index = converter.ignore_idx`, `reduction = 'none'`, followed by `.mean()`.  `CrossEntropyLoss` below takes the same
constructor arguments and gives the same values and gradients; for fp32 ROCm logits it runs `d2t_ce_forward` /
`d2t_ce_backward` (one pass over the logits each way instead of torch's log_softmax + nll_loss chain).  Anything the fused
kernels do not cover (class weights, label smoothing, non-fp32 or CPU inputs) raises -- there is no eager fallback.
"""
import torch
import torch.nn as nn

from . import _lib


class _FusedCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, ignore_index):
        lib = _lib.require_device()
        if not logits.is_cuda or logits.dtype != torch.float32 or logits.dim() != 2:
            raise RuntimeError("doc2tex_amd.loss: logits must be a [rows, V] float32 ROCm tensor (the engine has no CPU path)")
        if target.shape != logits.shape[:1]:
            raise ValueError(f"target shape {tuple(target.shape)} does not match logits {tuple(logits.shape)}")
        logits = logits.contiguous()
        target = target.to(device=logits.device, dtype=torch.int64).contiguous()
        rows, V = logits.shape
        loss = torch.empty(rows, dtype=torch.float32, device=logits.device)
        lse = torch.empty(rows, dtype=torch.float32, device=logits.device)
        _lib.check(lib.d2t_ce_forward(_lib.ptr(logits), _lib.ptr(target), _lib.ptr(loss), _lib.ptr(lse), rows, V,
                                      int(ignore_index), _lib.stream_of(logits)), None, "ce_forward")
        ctx.save_for_backward(logits, target, lse)
        ctx.ignore_index = int(ignore_index)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        logits, target, lse = ctx.saved_tensors
        lib = _lib.load()
        dloss = dloss.to(torch.float32).contiguous()
        dlogits = torch.empty_like(logits)
        rows, V = logits.shape
        _lib.check(lib.d2t_ce_backward(_lib.ptr(logits), _lib.ptr(target), _lib.ptr(lse), _lib.ptr(dloss), _lib.ptr(dlogits),
                                       rows, V, ctx.ignore_index, _lib.stream_of(logits)), None, "ce_backward")
        return dlogits, None, None


class CrossEntropyLoss(nn.Module):
    """Drop-in for the `nn.CrossEntropyLoss(ignore_index=..., reduction=...)` the reference builds
    (modules/loss/builder.py:21): same arguments, same forward signature `criterion(input [rows, V], target [rows])`."""

    def __init__(self, weight=None, size_average=None, ignore_index=-100, reduce=None, reduction="mean", label_smoothing=0.0):
        super().__init__()
        if weight is not None or label_smoothing != 0.0 or size_average is not None or reduce is not None:
            raise NotImplementedError("fused cross-entropy: class weights / label smoothing / legacy reduction flags are not supported")
        if reduction not in ("none", "mean", "sum"):
            raise ValueError(f"{reduction} is not a valid value for reduction")
        self.ignore_index, self.reduction = int(ignore_index), reduction

    def forward(self, input, target):
        loss = _FusedCE.apply(input, target, self.ignore_index)
        if self.reduction == "none":
            return loss
        if self.reduction == "sum":
            return loss.sum()
        # torch's 'mean' divides by the number of non-ignored targets
        return loss.sum() / (target != self.ignore_index).sum().clamp(min=1).to(loss.dtype)


def create_criterion(loss, loss_kwargs):
    """modules/loss/builder.py:18-24 with the fused kernel behind "entropy"."""
    if loss == "entropy":
        return CrossEntropyLoss(**loss_kwargs)
    raise NotImplementedError(f"criterion '{loss}' is not on the accelerated path (the shipped configs use 'entropy')")
