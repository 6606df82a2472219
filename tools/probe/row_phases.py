#!/usr/bin/env python3
"""Phase timeline of the two-row decode kernel (probe build: D2T_PROBES=1 bash doc2tex_amd/csrc/build.sh): average microseconds
between consecutive block barriers of block 0, over every launch of a few decode loops running alone.
usage: D2T_PROBE_LIB=doc2tex_amd/csrc/libd2t_probe.so python tools/probe/row_phases.py [group]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from doc2tex_amd import Model, _lib, synth

_lib.LIB_PATH = os.path.abspath(os.environ.get("D2T_PROBE_LIB", "doc2tex_amd/csrc/libd2t_probe.so"))
group = int(sys.argv[1]) if len(sys.argv) > 1 else 6
cfg = synth.make_config("C2", device="cuda")
m = Model(cfg)
m.load_state_dict(synth.synth_state_dict({k: v for k, v in m.state_dict().items()}), strict=False)
m.eval().to("cuda")
m.pipelined, m.decode_chains, m.decode_group, m.reserved_blocks = True, 1, group, 0
img = synth.synth_images(64, 128, 512, seed=1).cuda()
text = torch.full((64, 1), 1, dtype=torch.long, device="cuda")
NAMES = {19: "kernel entry", 0: "entry .. self-attention done", 1: "W_o products (+ W_q request)", 2: "reduce + residual", 3: "LN1",
         4: "W_q products (+ W_k request)", 5: "reduce q", 6: "absorbed queries", 12: "cross-attention loop (wave 0: most tiles)",
         7: "partials to LDS, W_v request, wait for the other waves", 8: "merge", 9: "W_v products (+ W_co request)", 10: "reduce",
         11: "W_co products", 20: "reduce + store"}
with torch.no_grad():
    for _ in range(2 * group):
        m(img, text, is_train=False)
    m.synchronize()
    lib = m.engine().lib
    lib.d2t_debug_row_phases.restype = C.c_int
    lib.d2t_debug_row_phases.argtypes = [C.c_void_p, C.c_int]
    buf = np.zeros(32, np.uint64)
    lib.d2t_debug_row_phases(buf.ctypes.data_as(C.c_void_p), 1)
    for _ in range(2 * group):
        m(img, text, is_train=False)
    m.synchronize()
    torch.cuda.synchronize()
    lib.d2t_debug_row_phases(buf.ctypes.data_as(C.c_void_p), 0)
n = int(buf[31])
print(f"{n} launches of the two-row kernel ({group * 64} rows)")
tot = 0.0
for k in (19, 0, 1, 2, 3, 4, 5, 6, 12, 7, 8, 9, 10, 11, 20):
    us = float(buf[k]) / max(1, n) / 100.0
    tot += us
    print(f"  {NAMES[k]:58s} {us:6.2f} us")
print(f"  {'sum':58s} {tot:6.2f} us")
print("inside the first phase (wave 0):")
for k, nm in ((19, "kernel entry (scalar state, small operands)"), (21, "first K / V groups issued"), (22, "the kernel's prefetches issued (tile DMA, W_o)"),
              (23, "further groups issued"), (24, "groups awaited + scored")):
    print(f"  {nm:58s} {float(buf[k]) / max(1, n) / 100.0:6.2f} us")
print("inside the cross-attention loop of wave 0 (per launch, all its tiles):")
for k, nm in ((13, "loop overhead"), (14, "next tile's loads issued"), (15, "score product S^T"),
              (16, "softmax + accumulator rescale"), (17, "weighted sum P.M"), (18, "next tile awaited + moved to LDS")):
    print(f"  {nm:58s} {float(buf[k]) / max(1, n) / 100.0:6.2f} us")
