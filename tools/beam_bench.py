#!/usr/bin/env python3
"""BASELINE configs[4] on one GPU: HybridViT + TFM-6, beam width 5, 160x640 crops.  The encoder runs on the whole
shard at once, beam search per sample (the reference's forward_beam is single-sample, tfm.py:146-148).
usage: beam_bench.py [n_samples] [beam] [batched|both] [shared]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from doc2tex_amd import Model, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
beam = int(sys.argv[2]) if len(sys.argv) > 2 else 5
only_batched = len(sys.argv) > 3 and sys.argv[3] == "batched"
cfg = synth.make_config("C4", device="cuda", beam_size=beam)
H, W = synth.crop_shape("C4")
m = Model(cfg)
m.load_state_dict(synth.synth_state_dict({k: v for k, v in m.state_dict().items()}), strict=False)
m = m.cuda().eval()
m.beam_shared_tile = len(sys.argv) > 4 and sys.argv[4] == "shared"
img = synth.synth_images(n, H, W, seed=11).cuda()
go = torch.ones(1, 1, dtype=torch.long, device="cuda")


def run():
    with torch.no_grad():
        mem, _, _ = m.forward_encoder(img)
        return [m.forward_decoder(mem[i:i + 1], go, is_train=False, is_test=True)[0] for i in range(n)]


def run_batched():
    with torch.no_grad():
        return [s for s, _ in m.beam_search_batch(img, beam)]


for name, fn in (("per sample (reference API)", run), ("batched (Model.beam_search_batch)", run_batched)):
    if only_batched and fn is run:
        continue
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"C4 beam {beam}, {H}x{W}, {n} samples, {name}: {dt * 1e3 / n:.1f} ms per formula = {n / dt:.1f} formulas/s "
          f"(sequence lengths {sorted(set(int(o.shape[1]) for o in out))})")
