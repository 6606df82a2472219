import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from doc2tex_amd import Model, synth
cfg = synth.make_config("C2", device="cuda")
m = Model(cfg)
tmpl = {k: v for k, v in m.state_dict().items()}
m.load_state_dict(synth.synth_state_dict(tmpl)); m.eval().cuda()
m.conv_precision = "bf16x3"; m.pipelined = True; m.decode_chains = int(os.environ.get("CHAINS", "2"))
img = synth.synth_images(64, 128, 512).cuda(); text = torch.ones(64, 1, dtype=torch.long, device="cuda")
with torch.no_grad():
    for _ in range(2): m(img, text, is_train=False)
    m.synchronize(); torch.cuda.synchronize()
    t0 = time.perf_counter(); ts = []
    for i in range(12):
        a = time.perf_counter()
        eng = m.engine()
        b = time.perf_counter()
        mem, _, _ = eng.encode(img)
        c = time.perf_counter()
        eng.decode_greedy_async(mem, text[:, 0])
        d = time.perf_counter()
        ts.append((b - a, c - b, d - c))
    m.synchronize(); torch.cuda.synchronize()
    tot = time.perf_counter() - t0
print("host ms per call: sync_weights %.2f encode %.2f decode_enqueue %.2f ; total/step %.2f ms" % (
    1e3 * sum(t[0] for t in ts) / 12, 1e3 * sum(t[1] for t in ts) / 12, 1e3 * sum(t[2] for t in ts) / 12, 1e3 * tot / 12))
print(["%.1f/%.1f/%.1f" % tuple(1e3 * x for x in t) for t in ts])
