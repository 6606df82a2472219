#!/usr/bin/env python3
"""Time the training step (config C3 per-GPU shard: HybridViT + TFM-6, 128x512 crops, B=32, 150-token labels):
forward (module.train()) + CE + backward in the HIP engine + torch.optim.AdamW step.  usage: train_bench.py [B] [steps] [fp32|bf16x3] [C2|S0]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from doc2tex_amd import Model, synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
CFG = sys.argv[4] if len(sys.argv) > 4 else "C2"  # "S0" = HybridViT + Attnv2 (the LSTM head of config/train.yaml)
cfg = synth.make_config(CFG, device="cuda")
H, W = synth.crop_shape(CFG)
L = cfg["batch_max_length"]
m = Model(cfg)
tmpl = {k: v for k, v in m.state_dict().items()}
m.load_state_dict(synth.synth_state_dict(tmpl), strict=False)
m = m.cuda().train()
m.conv_precision = sys.argv[3] if len(sys.argv) > 3 else "bf16x3"
opt = torch.optim.AdamW([p for p in m.parameters() if p.requires_grad], lr=1e-4)
img = synth.synth_images(B, H, W, seed=7).cuda()
text = synth.synth_labels(B, max_len=L, seed=7)
if cfg["Prediction"]["name"] != "TFM":  # Attn converter: [GO] = 0, [s] = 1 (attn_converter.py:8)
    t = text.clone(); t[text == 1] = 0; t[text == 2] = 1; text = t
text = text.cuda()
from doc2tex_amd.loss import create_criterion
crit = create_criterion("entropy", {"ignore_index": 0, "reduction": "none"})  # fused log-softmax + NLL (d2t_ce_*)


def step():
    _, preds, _ = m(img, text[:, :-1])
    loss = crit(preds.view(-1, preds.shape[-1]), text[:, 1:].contiguous().view(-1)).mean()
    loss.backward()
    torch.nn.utils.clip_grad_norm_(m.parameters(), 5.0)
    opt.step()
    m.zero_grad()
    return loss


for _ in range(2):
    l = step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    l = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f"train step {CFG} B={B} {m.conv_precision}: {dt * 1e3:.1f} ms = {B / dt:.1f} formulas/s, loss {float(l):.4f}, "
      f"peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB (torch) ")
