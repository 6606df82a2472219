#!/usr/bin/env python3
"""Encoder alone (no decode at all): ms per batch of 64 crops at 128x512 -- the floor of the serving step.
usage: python tools/probe/encoder_only.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from doc2tex_amd import Model, synth
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
cfg = synth.make_config("C2", device="cuda")
m = Model(cfg)
m.load_state_dict(synth.synth_state_dict({k: v for k, v in m.state_dict().items()}), strict=False)
m.eval().to("cuda")
img = synth.synth_images(64, 128, 512, seed=1).cuda()
with torch.no_grad():
    for _ in range(5):
        m.forward_encoder(img)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        m.forward_encoder(img)
    torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f"encoder alone: {dt * 1e3:.2f} ms per batch of 64 = {64 / dt:.0f} formulas/s")
