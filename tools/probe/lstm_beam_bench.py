import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from doc2tex_amd import Model, synth
n, beam = 64, 5
cfg = synth.make_config("S0", device="cuda", beam_size=beam)
H, W = synth.crop_shape("S0")
m = Model(cfg); m.load_state_dict(synth.synth_state_dict({k: v for k, v in m.state_dict().items()}), strict=False); m = m.cuda().eval()
img = synth.synth_images(n, H, W, seed=12).cuda()
text = torch.zeros(1, cfg["batch_max_length"] + 1, dtype=torch.long, device="cuda")
def per_sample():
    with torch.no_grad():
        mem, _, _ = m.forward_encoder(img)
        return [m.forward_decoder(mem[i:i+1], text, is_train=False, is_test=True)[0] for i in range(n)]
def batched():
    with torch.no_grad():
        return [s for s, _ in m.beam_search_batch(img, beam)]
for name, fn in (("per sample", per_sample), ("batched", batched)):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter(); out = fn(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"S0 (HybridViT + Attnv2) beam {beam}, {H}x{W}, {n} samples, {name}: {dt*1e3/n:.1f} ms per formula = {n/dt:.1f} formulas/s")
