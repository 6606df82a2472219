"""The shipped config/test.yaml geometry on the engine and on the oracle: HybridViT + Attnv2, max_dimension [448, 960] (1695 memory
tokens), batch_max_length 500, beam_size 5, one 448 x 960 crop.  usage: shipped_test_yaml_shape.py [end_bias]"""
import sys, time, json, torch
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from doc2tex_amd import Model, synth
from oracle import restatement as R
from conftest import oracle_state_dict
man = json.load(open(os.path.join(ROOT, "tests", "golden", "manifests.json")))
H, W, L, beam = 448, 960, 500, 5
EB = float(sys.argv[1]) if len(sys.argv) > 1 else 0.3
cfg = synth.make_config("S0", device="cuda", max_seq_len=L, beam_size=beam)
cfg["max_dimension"] = [H, W]
m = Model(cfg)
m.load_state_dict(synth.synth_state_dict({k: v for k, v in m.state_dict().items()}, end_bias=EB), strict=False)
m = m.cuda().eval()
ocfg, sd = oracle_state_dict("S0", man["S0"], L, end_bias=EB)
ocfg["max_dimension"] = [H, W]; ocfg["beam_size"] = beam
sd = dict(sd); sd["seqmodeler.SequenceModeling.pos_embed"] = R.sincos_2d_table(256, *R.vit_max_grid([H, W], (2, 2)))
img = synth.synth_images(1, H, W, seed=77)
text = torch.zeros(1, L + 1, dtype=torch.long)
with torch.no_grad():
    t0 = time.time(); seq, score, _ = m(img.cuda(), text.cuda(), is_train=False); torch.cuda.synchronize(); t1 = time.time()
    print("engine:", seq.shape, float(score), f"{t1 - t0:.2f}s", flush=True)
    t0 = time.time(); oseq, oscore, _ = R.forward(ocfg, sd, img, text, is_train=False); t1 = time.time()
    print("oracle:", oseq.shape, float(oscore), f"{t1 - t0:.1f}s", flush=True)
print("sequence equal:", seq.tolist() == oseq.tolist(), "dscore", abs(float(score) - float(oscore)))
