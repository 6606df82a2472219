#!/usr/bin/env python3
"""The dominant layer (512 -> 512, 3x3 @16x129, B = 64) on the 16x16x32 pipelined kernel, `reps` launches; run under
rocprofv3 --kernel-trace --stats with D2T_PROBE_LIB=<probe build> and D2T_CONV_ABL=0|1|2|4 (tools/probe/conv_abl.sh)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from doc2tex_amd import _lib

if os.environ.get("D2T_PROBE_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["D2T_PROBE_LIB"])
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
kind = int(sys.argv[2]) if len(sys.argv) > 2 else 3  # 3: pipelined 16x16x32 (split-bf16), 8: the same in fp16x2 arithmetic
lib = _lib.require_device()
B, H, W, Cin, Cout = 64, 16, 129, 512, 512
g = torch.Generator().manual_seed(0)
x = torch.randn(B, H, W, Cin, generator=g).cuda()
if os.environ.get("D2T_ABL_ZERO_X"):  # data dependence of the kernel time (power): all-zero activations
    x.zero_()
w = (torch.randn(Cout, 3, 3, Cin, generator=g) * (2.0 / (9 * Cin)) ** 0.5).cuda()
b = torch.randn(Cout, generator=g).cuda()
y = torch.empty(B, H, W, Cout, device="cuda")
assert lib.d2t_op_set_conv_kernel(kind, 0) == 0
for _ in range(reps):
    rc = lib.d2t_op_conv2d_bf16x3_split(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), None, _lib.ptr(y), B, H, W, Cin, Cout, 3, 3, 1, 1, 1, 1, 1,
                                        _lib.stream_of(x))
    assert rc == 0, rc
torch.cuda.synchronize()
print("done", float(y.float().abs().mean()))
