"""Soak: many forwards with changing batch sizes / crop shapes / modes; device memory must plateau."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from doc2tex_amd import Model, synth
cfg = synth.make_config("C2", device="cuda", max_seq_len=20)
m = Model(cfg); m.load_state_dict(synth.synth_state_dict({k: v for k, v in m.state_dict().items()}), strict=False); m = m.cuda().eval()
shapes = [(64, 128, 512), (8, 96, 384), (33, 128, 512), (1, 64, 256), (64, 112, 480), (17, 128, 400)]
free0 = None
t0 = time.time()
for it in range(120):
    B, H, W = shapes[it % len(shapes)]
    img = synth.synth_images(B, H, W, seed=it).cuda()
    go = torch.full((B, 1), 1, dtype=torch.long, device="cuda")
    m.pipelined = (it // 6) % 2 == 1
    m.decode_chains, m.decode_group, m.reserved_blocks = 2, 1 + (it // 12) % 3, (0 if (it // 12) % 3 else 64)
    with torch.no_grad():
        out = m(img, go, is_train=False, is_test=(it % 5 == 0 and not m.pipelined))
    if it % 6 == 5:
        m.synchronize(); torch.cuda.synchronize()
    if it == 35:
        m.synchronize(); torch.cuda.synchronize(); free0 = torch.cuda.mem_get_info()[0]
m.synchronize(); torch.cuda.synchronize()
free1 = torch.cuda.mem_get_info()[0]
print(f"soak: 120 forwards in {time.time()-t0:.1f} s; free memory after warm-up {free0/2**30:.2f} GiB, at the end {free1/2**30:.2f} GiB, delta {(free0-free1)/2**20:.1f} MiB")
