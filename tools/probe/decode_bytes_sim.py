#!/usr/bin/env python3
"""CPU study: what does a narrower K/V-cache / memory record cost the decoder in logits?  Emulates on the oracle (KV-cached
greedy decode) a decoder whose self-attention cache entries and/or encoder-memory rows are rounded to p significant bits
(16 = a 3-byte bf16 + 8 record, 11 = fp16, 8 = bf16) while everything else stays fp32.
usage: python tools/probe/decode_bytes_sim.py [config:B:H:W:L:n_image_seeds ...]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

from conftest import oracle_state_dict
from doc2tex_amd import synth
from oracle import restatement as R

MODE = {"cache": 24, "mem": 24}


def rnd(x, p):
    if p >= 24:
        return x
    m, e = torch.frexp(x)
    return torch.ldexp(torch.round(m * (1 << p)) / (1 << p), e)


_self_kv0, _cross_kv0 = R._self_kv, R.cross_kv


def self_kv(x, sd, p, heads):
    k, v = _self_kv0(x, sd, p, heads)
    return rnd(k, MODE["cache"]), rnd(v, MODE["cache"])


def cross_kv(mem, sd, p, heads):
    return _cross_kv0(rnd(mem, MODE["mem"]), sd, p, heads)  # (the engine reads the memory rows themselves: absorbed form)


R._self_kv, R.cross_kv = self_kv, cross_kv
SCHEMES = {"exact": (24, 24), "cache16": (16, 24), "mem16": (24, 16), "both16": (16, 16), "both14": (14, 14), "both11": (11, 11)}
specs = sys.argv[1:] or ["T2:2:48:64:20:3", "C2:1:128:512:40:2"]
with open(os.path.join(ROOT, "tests", "golden", "manifests.json")) as f:
    man = json.load(f)
torch.set_num_threads(max(1, min(32, len(os.sched_getaffinity(0)))))
for spec in specs:
    name, B, H, W, L, ni = spec.split(":")
    B, H, W, L, ni = int(B), int(H), int(W), int(L), int(ni)
    worst = {k: 0.0 for k in SCHEMES}
    flips = {k: 0 for k in SCHEMES}
    ocfg, sd = oracle_state_dict(name, man[name], L, 1234, 0.0)
    for k in range(ni):
        img = synth.synth_images(B, H, W, seed=9100 + 31 * k)
        text = torch.full((B, 1), R.GO, dtype=torch.long)
        with torch.no_grad():
            mem, _, _ = R.forward_encoder(ocfg, sd, img, faithful=False)
            pp = ocfg["Prediction"]["params"]
            ref = None
            for sname, (pc, pm) in SCHEMES.items():
                MODE["cache"], MODE["mem"] = pc, pm
                tok, lg = R.tfm_greedy(mem, sd, "predicter.Prediction.", pp["num_decoder_layers"], pp["nhead"], pp["max_seq_len"])
                if ref is None:
                    ref = (tok, lg)
                    continue
                d = float((lg - ref[1]).abs().max())
                worst[sname] = max(worst[sname], d)
                flips[sname] += int(not torch.equal(tok, ref[0]))
    print(f"{name} ({B}x{H}x{W}, {L + 1} steps, {ni} crops): " + ", ".join(f"{k} {worst[k]:.2e}/{flips[k]}" for k in SCHEMES if k != "exact"), flush=True)
