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

# per-layer table of the encoder's GEMM launches (HIP events on the launch stream, d2t_profile_*), encoder alone
eng = m.engine()
eng.profile(True)
with torch.no_grad():
    for _ in range(10):
        m.forward_encoder(img)
torch.cuda.synchronize()
eng.profile(False)
import collections
acc = collections.OrderedDict()
for M, N, K, ms in eng.profile_read(16384):
    a = acc.setdefault((M, N, K), [0, 0.0])
    a[0] += 1
    a[1] += ms
tot = sum(v[1] for v in acc.values()) / 10
print(f"GEMM launches: {tot:.2f} ms per forward")
print(f"{'M':>8} {'N':>5} {'K':>5} {'per fwd':>7} {'avg ms':>8} {'ms/fwd':>7} {'TFLOP/s':>8} {'of 2500':>7}")
for (M, N, K), (n, ms) in acc.items():
    print(f"{M:8d} {N:5d} {K:5d} {n / 10:7.1f} {ms / n:8.4f} {ms / 10:7.3f} {2.0 * M * N * K / (ms / n) / 1e9:8.1f} {2.0 * M * N * K / (ms / n) / 1e9 / 2500:7.3f}")

# decode step loop alone: 384 rows (six batches' memories), synchronous, no encoder beside it
with torch.no_grad():
    mem, _, _ = m.forward_encoder(img)
    mem6 = torch.cat([mem] * 6).contiguous()
    go6 = torch.full((mem6.shape[0], 1), 1, dtype=torch.long, device="cuda")
    m.pipelined = False
    for _ in range(2):
        m.forward_decoder(mem6, go6, is_train=False, is_test=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        m.forward_decoder(mem6, go6, is_train=False, is_test=False)
    torch.cuda.synchronize()
    dl = (time.perf_counter() - t0) / 5
print(f"decode loop alone: {dl * 1e3:.1f} ms per {mem6.shape[0]} rows x 151 steps = {dl * 1e3 / 6:.2f} ms per batch of 64")
