import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
from conftest import GOLD, engine_model, oracle_state_dict
from doc2tex_amd import synth
from oracle import restatement as R
from test_oracle_golden import _case, train_step_labels
cases = json.load(open(os.path.join(GOLD, "cases.json"))); manifests = json.load(open(os.path.join(GOLD, "manifests.json")))
name = sys.argv[1] if len(sys.argv) > 1 else "t2_train_step"
c = dict(_case(cases, "train_step", name))
if len(sys.argv) > 2:
    c["iseed"] = int(sys.argv[2])
cfg, sd = oracle_state_dict(c["config"], manifests[c["config"]], c["max_seq_len"], c["wseed"])
img = synth.synth_images(c["B"], c["H"], c["W"], seed=c["iseed"]); text = train_step_labels(c)
oloss, ologits, ograds, obn = R.train_step_grads(cfg, sd, img, text)
_, m = engine_model(c["config"], c["max_seq_len"], c["wseed"]); m.train()
_, preds, _ = m(img.cuda(), text[:, :-1].cuda())
loss = torch.nn.functional.cross_entropy(preds.view(-1, preds.shape[-1]), text[:, 1:].cuda().contiguous().view(-1), ignore_index=0, reduction="none").mean()
loss.backward(); torch.cuda.synchronize()
print("loss", float(loss), float(oloss), "logits maxdiff", float((preds.detach().cpu() - ologits).abs().max()))
params = dict(m.named_parameters())
import collections
grp = collections.OrderedDict()
def group(k):
    if "model.layers." in k: return "dec.L" + k.split("layers.")[1].split(".")[0]
    if k.startswith("predicter"): return "dec.other"
    if "blocks." in k: return "vit.block" + k.split("blocks.")[1].split(".")[0]
    if "ConvNet" in k:
        t = k.split("ConvNet.")[1]
        return "bb." + (t.split(".")[0] if t.startswith("layer") else t.split(".")[0])
    return "vit.other"
for k, g in ograds.items():
    e = params[k].grad.cpu().double(); g = g.double()
    grp.setdefault(group(k), []).append(float((e - g).norm() / max(float(g.norm()), 1e-30)))
for k, g in ograds.items():
    if "layer4" in k or "conv3" in k or "bn3" in k or "layer3.4" in k:
        e = params[k].grad.cpu().double(); g = g.double()
        print(f"   {float((e - g).norm() / max(float(g.norm()), 1e-30)):.2e} {k.split('ConvNet.')[1]}")
print("iseed", c["iseed"], "loss", float(loss), float(oloss))
for k, v in grp.items():
    print(f"  {k:14s} n={len(v):3d} relL2 max {max(v):.2e} median {sorted(v)[len(v)//2]:.2e}")
