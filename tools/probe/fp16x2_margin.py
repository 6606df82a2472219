#!/usr/bin/env python3
"""How much of the 1e-3 logits bar does fp16x2 use?  Engine (fp16x2 and split-bf16) against the CPU oracle on fresh crops and
fresh weight seeds, per config: max |dlogit| and whether the greedy tokens are exact.
usage (GPU box): [MARGIN_PRECS=bf16x3,fp16x2,mixed:4,mixed:8] python tools/probe/fp16x2_margin.py [config:B:H:W:L:n_image_seeds:n_weight_seeds ...]
("mixed:N" = conv_precision 'mixed' with N units on the two-MFMA arithmetic)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

from conftest import engine_model, oracle_state_dict
from doc2tex_amd import synth
from oracle import restatement as R

specs = sys.argv[1:] or ["T2:2:48:64:12:6:3", "C2:2:128:512:40:3:2", "C4:1:160:640:40:2:1"]
with open(os.path.join(ROOT, "tests", "golden", "manifests.json")) as f:
    man = json.load(f)
torch.set_num_threads(max(1, min(32, len(os.sched_getaffinity(0)))))
for spec in specs:
    name, B, H, W, L, ni, nw = spec.split(":")
    B, H, W, L, ni, nw = int(B), int(H), int(W), int(L), int(ni), int(nw)
    PRECS = os.environ.get("MARGIN_PRECS", "fp16x2,bf16x3").split(",")
    worst = {k: 0.0 for k in PRECS}
    exact = {k: True for k in PRECS}
    flips = {k: 0 for k in PRECS}
    for ws in range(nw):
        wseed = 1234 + 17 * ws
        cfg, m = engine_model(name, L, wseed, 0.0, beam_size=1)
        ocfg, sd = oracle_state_dict(name, man[name], L, wseed, 0.0)
        ocfg["beam_size"] = 1
        for k in range(ni):
            img = synth.synth_images(B, H, W, seed=9000 + 31 * k + ws)
            text = torch.full((B, 1), R.GO, dtype=torch.long)
            with torch.no_grad():
                op, ol, _ = R.forward(ocfg, sd, img, text, is_test=False, faithful=False)
                for prec in PRECS:
                    m.conv_precision = prec.split(":")[0]
                    if ":" in prec:
                        m.mixed_units = int(prec.split(":")[1])
                    p, l, _ = m(img.cuda(), text.cuda(), is_train=False)
                    d = float((l.cpu() - ol).abs().max())
                    worst[prec] = max(worst[prec], d)
                    exact[prec] &= bool(torch.equal(p.cpu(), op))
                    flips[prec] += int(not torch.equal(p.cpu(), op))
                    print(f"  {name} wseed {wseed} image seed {9000 + 31 * k + ws} {prec}: max |dlogit| {d:.2e}, tokens exact {torch.equal(p.cpu(), op)}", flush=True)
    print(f"{name} ({B}x{H}x{W}, {L + 1} steps, {ni} x {nw} runs): " +
          ", ".join(f"{k} worst {worst[k]:.2e} ({flips[k]} runs with a token flip)" for k in PRECS), flush=True)
