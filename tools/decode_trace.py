#!/usr/bin/env python3
"""Debug timeline of ONE decode step loop running beside the encoders of the following batches (probe build of the library,
D2T_PROBES=1 bash doc2tex_amd/csrc/build.sh, selected with D2T_PROBE_LIB; env D2T_DECODE_TRACE=1):
per kernel node of the captured loop, first block start / last block end (s_memrealtime, 10 ns ticks).
usage: D2T_PROBE_LIB=doc2tex_amd/csrc/libd2t_probe.so D2T_DECODE_TRACE=1 python tools/decode_trace.py [group] [conv_kernel] [encoders_alongside]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from doc2tex_amd import Model, _lib, synth

_lib.LIB_PATH = os.path.abspath(os.environ.get("D2T_PROBE_LIB", "doc2tex_amd/csrc/libd2t_probe.so"))  # the trace needs a probe build

group = int(sys.argv[1]) if len(sys.argv) > 1 else 3
kernel = sys.argv[2] if len(sys.argv) > 2 else "pipelined16"
alongside = int(sys.argv[3]) if len(sys.argv) > 3 else 12
cfg = synth.make_config("C2", device="cuda")
m = Model(cfg)
m.load_state_dict(synth.synth_state_dict({k: v for k, v in m.state_dict().items()}), strict=False)
m.eval().to("cuda")
m.pipelined, m.decode_chains, m.decode_group, m.reserved_blocks, m.conv_kernel = True, 1, group, 0, kernel
img = synth.synth_images(64, 128, 512, seed=1).cuda()
text = torch.full((64, 1), 1, dtype=torch.long, device="cuda")
eng = None
with torch.no_grad():
    for _ in range(2 * group):  # warm-up: captures the loop graph
        m(img, text, is_train=False)
    m.synchronize()
    eng = m.engine()
    lib = eng.lib
    lib.d2t_debug_trace.restype = C.c_int32
    lib.d2t_debug_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]
    lib.d2t_debug_trace(eng.ctx, None, 0, 1)  # reset
    for _ in range(group):  # the traced loop is launched by the last of these forwards
        m(img, text, is_train=False)
    m.decode_group = alongside + 1  # the following forwards only encode (their memories pile up in a group that is never launched)
    for _ in range(alongside):
        m(img, text, is_train=False)
    eng.decode_wait(host_sync=True)
    torch.cuda.synchronize()
    buf = np.zeros((8192, 2), np.uint64)
    n = lib.d2t_debug_trace(eng.ctx, buf.ctypes.data_as(C.c_void_p), 8192, 0)
m._grp = None
t = buf[:n].astype(np.int64)
ok = t[:, 1] > 0
print(f"{n} kernel nodes, {int(ok.sum())} recorded")
t = t[ok]
t0 = t[:, 0].min()
span = (t[:, 1] - t[:, 0]) / 100.0  # us
gap = (t[1:, 0] - t[:-1, 1]) / 100.0
per_step = 6 * 4 + 2
names = []
for l in range(6):
    names += [f"L{l}.qkv", f"L{l}.row", f"L{l}.ff1", f"L{l}.ff2"]
names += ["vocab", "argmax"]
print(f"loop {(t[:, 1].max() - t0) / 100.0 / 1e3:.1f} ms for {len(t) // per_step} steps; kernel span sum {span.sum() / 1e3:.1f} ms, "
      f"gap sum {gap.sum() / 1e3:.1f} ms")
print("per kernel type: avg span us / avg gap-before us")
for k in range(per_step):
    sp = span[k::per_step]
    gp = gap[k - 1::per_step] if k else gap[per_step - 1::per_step]
    print(f"  {names[k]:8s} span {sp.mean():7.1f} (p90 {np.percentile(sp, 90):7.1f})   gap {gp.mean():7.1f} (p90 {np.percentile(gp, 90):7.1f})")
