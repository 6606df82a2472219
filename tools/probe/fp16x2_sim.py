#!/usr/bin/env python3
"""CPU study of the two-MFMA (fp16x2) backbone arithmetic: where does its logits error come from and which storage /
per-layer choices buy margin?  Emulates, on the oracle (test infrastructure; this probe is not product code), a backbone
whose convolutions see their input rounded to fp16 (weights as fp16 hi + lo, i.e. 22 bits), under several schemes:

  cur      round-3 behaviour: every stored feature map is ONE fp16 (conv inputs AND the residual path see it)
  res      (a) residual stream kept at full precision (fp16 hi + lo in the record): only MFMA operands are rounded
  res+kN   (b) as `res`, and only layers with K = kh*kw*cin >= N run in fp16 (the others exact = split-bf16)
  cur+kN   per-layer choice alone

usage: python tools/probe/fp16x2_sim.py [config:B:H:W:L:n_image_seeds:n_weight_seeds ...]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import torch.nn.functional as F

from conftest import oracle_state_dict
from doc2tex_amd import synth
from oracle import restatement as R

MODE = {"round_in": False, "res_exact": False, "kmin": 0, "store16": False}


def h16(x):
    p = MODE.get("bits", 11)
    if p == 11:
        return x.half().float()
    m, e = torch.frexp(x)  # x = m * 2^e, 0.5 <= |m| < 1: p significant bits
    return torch.ldexp(torch.round(m * (1 << p)) / (1 << p), e)


SEEN = [0]  # running index of the K = 4608 convolutions inside one forward


_conv_bn0, _basic_block0 = R._conv_bn, R._basic_block


def conv_bn(x, sd, conv, bn, stride=1, padding=0, faithful=True, bn_train=None):
    w = sd[conv + ".weight"]
    wf, bf = R.fold_bn(w, sd, bn)
    K = w.shape[1] * w.shape[2] * w.shape[3]
    on = MODE["round_in"] and w.shape[1] >= 32 and K >= MODE["kmin"]
    if on and K == 4608 and "pick" in MODE and MODE["pick"] is not None:
        on = SEEN[0] in MODE["pick"]
    SEEN[0] += int(K == 4608)
    if on:
        wh = h16(wf)
        wf = wh + h16(wf - wh)
        x = h16(x)
    return F.conv2d(x, wf, bf, stride, padding)


def basic_block(x, sd, p, faithful, bn_train=None):
    out = F.relu(conv_bn(x, sd, p + ".conv1", p + ".bn1", 1, 1))
    out = conv_bn(out, sd, p + ".conv2", p + ".bn2", 1, 1)
    xr = x if MODE["res_exact"] or not MODE["round_in"] else h16(x)
    if (p + ".downsample.0.weight") in sd:
        xr = conv_bn(x, sd, p + ".downsample.0", p + ".downsample.1", 1, 0)
    return F.relu(out + xr)


R._conv_bn, R._basic_block = conv_bn, basic_block

SCHEMES = {
    "exact": dict(round_in=False, res_exact=True, kmin=0),
    "cur": dict(round_in=True, res_exact=False, kmin=0),
    "res": dict(round_in=True, res_exact=True, kmin=0),
    "res+k1152": dict(round_in=True, res_exact=True, kmin=1152),
    "res+k2304": dict(round_in=True, res_exact=True, kmin=2304),
    "res+k4608": dict(round_in=True, res_exact=True, kmin=4608),
    "cur+k4608": dict(round_in=True, res_exact=False, kmin=4608),
    "res12": dict(round_in=True, res_exact=True, kmin=0, bits=12),
    "res13": dict(round_in=True, res_exact=True, kmin=0, bits=13),
    "res14": dict(round_in=True, res_exact=True, kmin=0, bits=14),
    "first4": dict(round_in=True, res_exact=True, kmin=4608, pick=set(range(0, 4))),
    "first8": dict(round_in=True, res_exact=True, kmin=4608, pick=set(range(0, 8))),
    "last8": dict(round_in=True, res_exact=True, kmin=4608, pick=set(range(8, 16))),
    "last4": dict(round_in=True, res_exact=True, kmin=4608, pick=set(range(12, 16))),
}
if os.environ.get("SIM_SINGLE"):
    SCHEMES = {"exact": SCHEMES["exact"], "res+k4608": SCHEMES["res+k4608"]}
    for i in range(16):
        SCHEMES[f"only{i}"] = dict(round_in=True, res_exact=True, kmin=4608, pick={i})
    SCHEMES["lt4608"] = dict(round_in=True, res_exact=True, kmin=4608, pick=set())  # bookkeeping: nothing rounded
    SCHEMES["k<4608"] = dict(round_in=True, res_exact=True, kmin=0, pick=set())     # only the layers with K < 4608
for v in SCHEMES.values():
    v.setdefault("bits", 11)
    v.setdefault("pick", None)

specs = sys.argv[1:] or ["C2:1:128:512:24:2:2", "T2:2:48:64:12:4:3"]
with open(os.path.join(ROOT, "tests", "golden", "manifests.json")) as f:
    man = json.load(f)
torch.set_num_threads(max(1, min(32, len(os.sched_getaffinity(0)))))
for spec in specs:
    name, B, H, W, L, ni, nw = spec.split(":")
    B, H, W, L, ni, nw = int(B), int(H), int(W), int(L), int(ni), int(nw)
    worst = {k: 0.0 for k in SCHEMES}
    rms = {k: 0.0 for k in SCHEMES}
    flips = {k: 0 for k in SCHEMES}
    for ws in range(nw):
        wseed = 1234 + 17 * ws
        ocfg, sd = oracle_state_dict(name, man[name], L, wseed, 0.0)
        ocfg["beam_size"] = 1
        for k in range(ni):
            img = synth.synth_images(B, H, W, seed=9000 + 31 * k + ws)
            text = torch.full((B, 1), R.GO, dtype=torch.long)
            ref = None
            for sname, mode in SCHEMES.items():
                MODE.update(mode)
                SEEN[0] = 0
                with torch.no_grad():
                    op, ol, _ = R.forward(ocfg, sd, img, text, is_test=False, faithful=False)
                if sname == "exact":
                    ref = (op, ol)
                    continue
                d = float((ol - ref[1]).abs().max())
                worst[sname] = max(worst[sname], d)
                rms[sname] += float(((ol - ref[1]) ** 2).mean()) / (ni * nw)
                flips[sname] += int(not torch.equal(op, ref[0]))
                print(f"  {name} w{wseed} i{9000 + 31 * k + ws} {sname:10s} max|dlogit| {d:.2e} tokens {'=' if torch.equal(op, ref[0]) else 'FLIP'}", flush=True)
    print(f"{name} ({B}x{H}x{W}, {L + 1} steps, {ni}x{nw} runs): " + ", ".join(f"{k} {worst[k]:.2e}/{rms[k] ** 0.5:.2e}/{flips[k]}" for k in SCHEMES if k != "exact"), flush=True)
