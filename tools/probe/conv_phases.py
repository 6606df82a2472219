#!/usr/bin/env python3
"""Per-tile timeline of the pipelined convolution kernel (probe build: D2T_PROBES=1 bash doc2tex_amd/csrc/build.sh): average
microseconds between consecutive marks of block 0's first compute wave, per tile, on the backbone's shapes, one shape at a
time through the C-ABI op entry point.
usage: D2T_PROBE_LIB=doc2tex_amd/csrc/libd2t_probe.so python tools/probe/conv_phases.py [reps]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from doc2tex_amd import _lib

_lib.LIB_PATH = os.path.abspath(os.environ.get("D2T_PROBE_LIB", "doc2tex_amd/csrc/libd2t_probe.so"))
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
lib = _lib.require_device()
lib.d2t_debug_conv_phases.restype = C.c_int
lib.d2t_debug_conv_phases.argtypes = [C.c_void_p, C.c_int]
NAMES = ["previous barrier .. tile start (loop overhead / kernel entry)", "setup + first two stages landed (pipeline fill)",
         "K loop (this wave)", "waiting for the other waves after the K loop", "accumulators to the LDS tile + barrier",
         "epilogue rows of this thread (reads, residual, split, stores issued)", "waiting for the slowest thread's epilogue"]
# name, B, H, W, Cin, Cout, residual
LAYERS = [("512->512 @16x129, conv1 (no residual)", 64, 16, 129, 512, 512, False),
          ("512->512 @16x129, conv2 (+ residual)", 64, 16, 129, 512, 512, True),
          ("256->256 @32x128 (no residual)", 64, 32, 128, 256, 256, False),
          ("128->256 @32x128, K = 1152", 64, 32, 128, 128, 256, False),
          ("64->128 @64x256, K = 576", 64, 64, 256, 64, 128, False)]
g = torch.Generator().manual_seed(0)
assert lib.d2t_op_set_conv_kernel(3, 0) == 0
for name, B, H, W, Cin, Cout, use_res in LAYERS:
    x = torch.randn(B, H, W, Cin, generator=g).cuda()
    w = (torch.randn(Cout, 3, 3, Cin, generator=g) * (2.0 / (9 * Cin)) ** 0.5).cuda()
    b = torch.randn(Cout, generator=g).cuda()
    res = torch.randn(B, H, W, Cout, generator=g).cuda() if use_res else None
    y = torch.empty(B, H, W, Cout, device="cuda")

    def run():
        rc = lib.d2t_op_conv2d_bf16x3_split(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(res), _lib.ptr(y), B, H, W, Cin, Cout,
                                            3, 3, 1, 1, 1, 1, 1, _lib.stream_of(x))
        assert rc == 0, rc
    run()
    torch.cuda.synchronize()
    buf = np.zeros(16, np.uint64)
    lib.d2t_debug_conv_phases(buf.ctypes.data_as(C.c_void_p), 1)
    for _ in range(reps):
        run()
    torch.cuda.synchronize()
    lib.d2t_debug_conv_phases(buf.ctypes.data_as(C.c_void_p), 0)
    n = int(buf[15])
    KT = 9 * Cin // 32
    tot = sum(float(buf[k]) for k in range(7)) / max(1, n) / 100.0
    print(f"{name}: M={B * H * W} N={Cout} K-steps={KT}; {n} tiles of block 0 in {reps} launches, {tot:.1f} us per tile")
    for k in range(7):
        us = float(buf[k]) / max(1, n) / 100.0
        print(f"  {NAMES[k]:72s} {us:7.2f} us" + (f"  ({us / KT:.3f} per K-step)" if k == 2 else ""))
