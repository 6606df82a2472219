#!/usr/bin/env python3
"""Throughput of the pre-processing row on one MI355X: B rendered-formula pages of ~3x the crop size ->
LANCZOS to max_dimension -> normalised [B,1,128,512] batch.  Reports images/s with the source bytes already in HBM
(kernels only, HIP events), the end-to-end rate from host arrays (pack + H2D + tables + kernels), the algorithmic
bytes per image (source read once + fp32 output written once) against the HBM roofline, and Pillow on the host cores
(the reference's path) on a bounded sample."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from doc2tex_amd import synth
from doc2tex_amd.preprocess import Preprocessor


def pmc_traffic():
    """HBM bytes per batch of 64 pages from the committed PMC passes (tools/probe/prep_pmc.sh), or None."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_prep.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        return json.load(f).get("hbm_bytes_per_batch_of_64")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--cpu-sample", type=int, default=64)
    args = ap.parse_args()
    opt = {"imgH": None, "imgW": None, "max_dimension": [128, 512], "min_dimension": [32, 32], "mean": 0.5, "std": 0.5,
           "rgb": False, "pad": False, "device": "cuda"}
    rng = np.random.default_rng(1)
    imgs = [synth.synth_formula_image(int(rng.integers(300, 420)), int(rng.integers(1500, 1640)), 8000 + i)
            for i in range(args.batch)]  # aspect >= 4: every page lands on 128x512 or 96x512
    pre = Preprocessor(opt, "demo")
    for _ in range(6):  # the four pinned staging blocks are allocated on first use (~0.15 s each)
        tensors, _ = pre.batch(imgs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        tensors, _ = pre.batch(imgs)
    torch.cuda.synchronize()
    e2e = (time.perf_counter() - t0) / args.steps
    # the same with page sizes the handle has not seen: every resampling table is built (libm sin on host threads)
    cold_sets = [[np.ascontiguousarray(a[:a.shape[0] - 1 - k, :a.shape[1] - 3 - 2 * k]) for a in imgs] for k in range(args.steps)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for cs in cold_sets:
        tensors, _ = pre.batch(cs)
    torch.cuda.synchronize()
    e2e_cold = (time.perf_counter() - t0) / args.steps

    # kernels only: sources resident, one bucket per output size
    plans = [pre.plan(*a.shape) for a in imgs]
    groups = {}
    for a, p in zip(imgs, plans):
        groups.setdefault((p.out_h, p.out_w), []).append((a, p))
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    import ctypes as C
    from doc2tex_amd import _lib
    staged = []
    for (oh, ow), g in groups.items():
        offs = np.zeros(len(g), np.int64)
        tot = 0
        for i, (a, _) in enumerate(g):
            offs[i] = tot
            tot += (a.size + 15) & ~15
        host = np.zeros(tot, np.uint8)
        for (a, _), o in zip(g, offs):
            host[o:o + a.size] = a.reshape(-1)
        staged.append((oh, ow, torch.from_numpy(host).cuda(), offs, (_lib.D2TPrepPlan * len(g))(*[p for _, p in g]),
                       torch.empty((len(g), 1, oh, ow), device="cuda"), len(g)))

    def kernels():
        for oh, ow, src, offs, pl, out, n in staged:
            rc = pre.lib.d2t_prep_run(pre.h, n, pl, _lib.ptr(src), offs.ctypes.data_as(C.POINTER(C.c_int64)), _lib.ptr(out),
                                      oh, ow, None, _lib.stream_of(out))
            assert rc == 0
    for _ in range(3):
        kernels()
    torch.cuda.synchronize()
    ev0.record()
    for _ in range(args.steps):
        kernels()
    ev1.record()
    torch.cuda.synchronize()
    dev = ev0.elapsed_time(ev1) / 1e3 / args.steps
    src_bytes = sum(a.size for a in imgs)
    out_bytes = sum(p.out_h * p.out_w * 4 for p in plans)
    inter = sum(p.ds_h * p.rs_w * 2 for p in plans)  # horizontal-pass image written + read once

    from PIL import Image
    lut = ((np.arange(256, dtype=np.float32) - np.float32(127.5)) * np.reciprocal(np.float32(127.5))).astype(np.float32)
    sample = imgs[:args.cpu_sample]
    t0 = time.perf_counter()
    for a in sample:
        im = Image.fromarray(a, "L")
        oh, ow = pre.plan(*a.shape).rs_h, pre.plan(*a.shape).rs_w
        r = np.asarray(im.resize((ow, oh), Image.LANCZOS).convert("RGB")).astype("uint8")
        _ = torch.from_numpy(lut[r[..., 0]])[None, None]
    cpu = (time.perf_counter() - t0) / len(sample)
    print(json.dumps({
        "metric": "images/s (pre-processing: LANCZOS to 128x512 + normalise + collate)", "unit": "images/s",
        "value_resident": round(args.batch / dev, 1), "value_from_host_arrays": round(args.batch / e2e, 1),
        "value_from_host_arrays_new_sizes": round(args.batch / e2e_cold, 1),
        "ms_per_batch_kernels": round(dev * 1e3, 3), "ms_per_batch_end_to_end": round(e2e * 1e3, 3), "batch": args.batch,
        "roofline": {"bound": "hbm", "achieved": round((src_bytes + out_bytes) / dev / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                     "frac": round((src_bytes + out_bytes) / dev / 8e12, 4), "traffic": pmc_traffic(),
                     "traffic_unit": "HBM bytes per batch of 64 (profiles/r01_pmc_prep.json)",
                     "algorithmic_bytes_per_image": (src_bytes + out_bytes) // args.batch,
                     "with_intermediate_bytes_per_image": (src_bytes + out_bytes + inter) // args.batch},
        "cpu_baseline": {"value": round(1 / cpu, 1), "unit": "images/s", "cores": 1, "kind": "reference",
                         "sample": f"{len(sample)} pages through Pillow LANCZOS + table normalise (the reference's library calls)"},
        "source_page": "300-420 x 1500-1640 uint8"}))


if __name__ == "__main__":
    main()
