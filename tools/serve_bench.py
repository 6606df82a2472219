#!/usr/bin/env python3
"""End-to-end serving loop on one MI355X: host uint8 pages -> d2t_prep_run (LANCZOS to 128x512, normalise, collate)
-> Model.forward (HybridViT + TFM-6 greedy, pipelined) -> d2t_post_decode (ids -> LaTeX string, whitespace clean-up).
The rows either side of the hot path in the loop with it; bench.py stays the headline (inputs
resident, forward only).  Prints one JSON line: formulas/s end to end, and the host milliseconds per batch spent in the
pre- and post-processing calls (one Python thread)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from doc2tex_amd import Model, synth
from doc2tex_amd.postprocess import LabelDecoder
from doc2tex_amd.preprocess import Preprocessor

SYMBOLS = ["\\frac", "{", "}", "x", "y", "a", "b", "1", "2", "^", "_", "\\mathrm", "\\operatorname", "*", "\\alpha", "+", "=",
           "(", ")", "\\,", "d", "\\hspace", "\\mathbf", "\\left", "\\right", ".", "~", "\\\\", "&", "e", "\\sum", "\\int", "|"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--group", type=int, default=6, help="decode groups (batches per decode step loop)")
    ap.add_argument("--chains", type=int, default=3, help="decode chains in flight (as bench.py)")
    ap.add_argument("--page-sets", type=int, default=4, help="distinct batches of pages cycled through")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    cfg = synth.make_config("C2", device=str(dev))
    model = Model(cfg)
    tmpl = {k: v for k, v in model.state_dict().items() if not k.endswith("image_positional_encoder.pe")}
    model.load_state_dict(synth.synth_state_dict(tmpl), strict=False)
    model.eval().to(dev)
    model.pipelined, model.decode_chains, model.decode_group, model.reserved_blocks = True, args.chains, args.group, 0 if args.group > 1 else 64
    opt = {"imgH": None, "imgW": None, "max_dimension": cfg["max_dimension"], "min_dimension": [32, 32], "mean": 0.5,
           "std": 0.5, "rgb": False, "pad": False, "device": str(dev)}
    pre = Preprocessor(opt, "demo")
    vocab = [SYMBOLS[i % len(SYMBOLS)] + ("" if i < len(SYMBOLS) else f"_{i}") for i in range(synth.VOCAB - 4)]
    dec = LabelDecoder(vocab, head="TFM")
    rng = np.random.default_rng(3)
    B = args.batch
    # pages whose aspect ratio lands every one of them on the 128x512 bucket (h/w between 97/512 and 128/512)
    sets = [[synth.synth_formula_image(int(rng.integers(310, 395)), int(rng.integers(1580, 1620)), 9000 + s * B + i)
             for i in range(B)] for s in range(args.page_sets)]
    text = torch.full((B, 1), 1, dtype=torch.long, device=dev)
    side = torch.cuda.Stream(dev)
    L = cfg["Prediction"]["params"]["max_seq_len"] + 1
    ring = [torch.empty((B, L), dtype=torch.int64).pin_memory() for _ in range(16)]
    pending, waiting, done, t_pre, t_post, sample = [], [], 0, 0.0, 0.0, None

    def consume(block):
        nonlocal done, t_post, sample
        if block and waiting:  # an incomplete last group: launch it and copy its batches out
            model.synchronize(host_sync=False)
            for k, tk in waiting:
                ring[k % len(ring)].copy_(tk, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            pending.extend((k, ev) for k, _ in waiting)
            waiting.clear()
        while pending and (block or pending[0][1].query()):
            k, ev = pending.pop(0)
            ev.synchronize()
            t = time.perf_counter()
            latex = dec.to_latex(ring[k % len(ring)], "word", postprocess=True)
            t_post += time.perf_counter() - t
            done += len(latex)
            sample = latex[0]

    def prepare(i):
        nonlocal t_pre
        t = time.perf_counter()
        tensors, errors = pre.batch(sets[i % len(sets)])
        t_pre += time.perf_counter() - t
        assert all(e is None for e in errors)
        x = tensors[0]._base
        assert x.shape == (B, 1, 128, 512), x.shape
        return x

    nxt = {"x": None}

    def step(i):
        # batch i was pre-processed during step i-1: its few kernels sit in the stream ahead of encoder i-1's successor, and
        # the host work of batch i+1 (packing pages into pinned memory) is done while encoder i runs
        x = nxt["x"] if nxt["x"] is not None else prepare(i)
        with torch.no_grad():
            tokens, _, _ = model(x, text, is_train=False, is_test=False)
        waiting.append((i, tokens))
        if (i + 1) % args.group == 0:  # this call completed a decode group: its batches can be copied out behind it
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                model.synchronize(host_sync=False, flush=False)  # the side stream waits for the decodes in flight, the main one does not
                for k, tk in waiting:
                    ring[k % len(ring)].copy_(tk, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
            pending.extend((k, ev) for k, _ in waiting)
            waiting.clear()
        nxt["x"] = prepare(i + 1)
        consume(False)

    for i in range(args.warmup):
        step(i)
    consume(True)
    torch.cuda.synchronize(dev)
    done, t_pre, t_post = 0, 0.0, 0.0
    pre.timing = {}
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    consume(True)
    torch.cuda.synchronize(dev)
    el = time.perf_counter() - t0
    assert done == B * args.steps
    print(json.dumps({"metric": "formulas/s end to end (uint8 pages -> LaTeX strings)", "value": round(done / el, 1),
                      "unit": "formulas/s", "steps": args.steps, "batch": B, "ms_per_batch": round(el / args.steps * 1e3, 2),
                      "host_ms_per_batch": {"preprocess_call": round(t_pre / args.steps * 1e3, 2),
                                            "postprocess_call": round(t_post / args.steps * 1e3, 2),
                                            # preprocess_call split up: `stage_wait` is the host waiting for the H2D copy it
                                            # queued four batches ago (the GPU is the bottleneck and the host runs ahead of
                                            # it until its staging ring is full: back-pressure, not work)
                                            "preprocess_phases": {k: round(v / args.steps * 1e3, 2) for k, v in pre.timing.items()}},
                      "page": "310-395 x 1580-1620 uint8 -> 128x512", "sample_latex_chars": len(sample)}))


if __name__ == "__main__":
    main()
