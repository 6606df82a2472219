"""Training-mode forward/backward of `Model` as one autograd node.

The reference trains through PyTorch autograd (`loss.backward()`, engine/training.py:137) on the logits that
`model(image, text[:, :-1])` returns (`forward_step`, training.py:76-91).  Here the whole network is one
`torch.autograd.Function`: forward = `d2t_train_forward` (BatchNorm on batch statistics, teacher-forced decoder),
backward = `d2t_train_backward` on dL/dlogits, after which every parameter's gradient is copied out of the engine.
The caller's criterion, `clip_grad_norm_` and optimizer (torch.optim, out of scope per SURVEY section 2) run unchanged.
"""
import torch


class _TrainStep(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, image, text, names, *params):
        # the LSTM-attention head reads the engine's derived (transposed / folded) decoder weights, so it needs the
        # finalize pass; the TFM head reads the raw copies only
        eng = model.engine(finalize=model.stages["Pred"] != "TFM")
        # dropout (nn.TransformerDecoderLayer(dropout=p) / the LSTM head's droprate on its generator output): masks from
        # the engine's Philox stream, seeded like torch's generator
        pp = model.opt["Prediction"]["params"]
        tfm = model.stages["Pred"] == "TFM"
        p = float(pp.get("dropout" if tfm else "droprate", 0.0) or 0.0)
        eng.set_dropout(p, torch.initial_seed())
        if not tfm:
            # scheduled sampling (seq2seq.py:311-316): one random.random() per step but the last, drawn exactly as the
            # reference draws them, so seeding `random` reproduces its choices
            import random
            tf = float(pp.get("teacher_forcing", 1.0))
            flags = [1] + [0 if tf < random.random() else 1 for _ in range(text.shape[1] - 1)]
            eng.set_teacher_flags(None if all(flags) else flags)
        logits = eng.train_forward(image, text)
        # BatchNorm side effects of module.train(): running statistics and the batch counter
        with torch.no_grad():
            stats = [(name, buf) for name, buf in model.named_buffers() if name.endswith(("running_mean", "running_var"))]
            if stats:  # one gather kernel + one multi-tensor copy instead of a device copy per buffer
                got = eng.train_gather([n for n, _ in stats], [b for _, b in stats], source="weight")
                torch._foreach_copy_([b for _, b in stats], got)
            for name, buf in model.named_buffers():
                if name.endswith("num_batches_tracked"):
                    buf += 1
        eng._sig = None  # the buffers above changed: re-upload before the next eval forward
        ctx.model, ctx.names, ctx.params = model, names, params
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        eng = ctx.model._engine
        eng.train_backward(dlogits)  # enqueues the whole backward on the current stream and returns
        sync = getattr(ctx.model, "grad_sync", None)
        if sync is None:
            grads = tuple(eng.train_gather(ctx.names, ctx.params))  # views of one flat buffer, filled by one kernel
        else:  # data-parallel: bucketed all-reduce-mean overlapped with the rest of the backward (dist.GradSync)
            grads = tuple(sync.collect(eng.train_grad_into, ctx.names, ctx.params))
        return (None, None, None, None) + grads


def train_forward(model, image, text):
    """logits [B, L, V] with autograd history reaching every trainable parameter of `model`."""
    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    names = tuple(n for n, _ in named)
    return _TrainStep.apply(model, image, text, names, *[p for _, p in named])
