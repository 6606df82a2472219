#!/usr/bin/env python3
"""Every GEMM launch of ONE encoder forward in launch order (HIP events on the launch stream, d2t_profile_*), averaged over
`reps` forwards: tells a block's conv1 (no residual) from its conv2 (+ residual records) at the same shape, and gives the
per-tile fixed cost from layers that differ in K only.
usage: python tools/probe/conv_per_launch.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from doc2tex_amd import Model, synth
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
cfg = synth.make_config("C2", device="cuda")
m = Model(cfg)
m.load_state_dict(synth.synth_state_dict({k: v for k, v in m.state_dict().items()}), strict=False)
m.eval().to("cuda")
img = synth.synth_images(64, 128, 512, seed=1).cuda()
eng = None
with torch.no_grad():
    for _ in range(3):
        m.forward_encoder(img)
    torch.cuda.synchronize()
    eng = m.engine()
    eng.profile(True)
    for _ in range(reps):
        m.forward_encoder(img)
    torch.cuda.synchronize()
    eng.profile(False)
recs = eng.profile_read(65536)
per = len(recs) // reps
assert per * reps == len(recs), (len(recs), reps)
print(f"{per} GEMM launches per forward, {reps} forwards, {sum(r[3] for r in recs) / reps:.3f} ms of GEMM launches per forward")
print(f"{'#':>3} {'M':>8} {'N':>5} {'K':>5} {'avg ms':>8} {'min ms':>8} {'tiles/256':>9} {'us/round':>8}")
for i in range(per):
    M, N, K = recs[i][:3]
    ts = [recs[r * per + i][3] for r in range(reps)]
    assert all(recs[r * per + i][:3] == (M, N, K) for r in range(reps))
    rounds = -(-M // 256) * -(-N // 128) / 256
    print(f"{i:3d} {M:8d} {N:5d} {K:5d} {sum(ts) / reps:8.4f} {min(ts):8.4f} {rounds:9.2f} {sum(ts) / reps * 1e3 / rounds:8.1f}")
