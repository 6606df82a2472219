#!/usr/bin/env python3
"""Run the dominant convolution (512->512 3x3 @16x129, B=64) a few times through the C-ABI op entry
points, for rocprofv3 counter passes.  usage: prof_conv.py [bf16x3|split|fp32] [reps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from doc2tex_amd import _lib

mode = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
lib = _lib.require_device()
B, H, W, C = 64, 16, 129, 512
g = torch.Generator().manual_seed(0)
x = torch.randn(B, H, W, C, generator=g).cuda()
w = (torch.randn(C, 3, 3, C, generator=g) * (2.0 / (9 * C)) ** 0.5).cuda()
b = torch.randn(C, generator=g).cuda()
y = torch.empty(B, H, W, C, device="cuda")
fn = {"bf16x3": lib.d2t_op_conv2d_bf16x3, "split": lib.d2t_op_conv2d_bf16x3_split, "fp32": lib.d2t_op_conv2d}[mode]
st = _lib.stream_of(x)
for i in range(reps):
    t0 = time.perf_counter()
    assert fn(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), None, _lib.ptr(y), B, H, W, C, C, 3, 3, 1, 1, 1, 1, 1, st) == 0
    torch.cuda.synchronize()
    print(f"{mode} call {i}: {(time.perf_counter() - t0) * 1e3:.2f} ms (incl. weight repack)", flush=True)
