#!/usr/bin/env python3
"""A/B of the two split-bf16 convolution kernels on the backbone's layer shapes, through the C-ABI op entry point
(d2t_op_conv2d_bf16x3_split: split-record input, residual and output).  Checks that the pipelined 256x128 kernel returns
bit-identical results to the 128x128 one, then runs every variant several times interleaved in ONE process; kernel times
come from rocprofv3 (run this under `rocprofv3 --kernel-trace --stats`) -- the op call itself includes weight repacking.

usage: conv_bench.py [reps] [reserved_cus ...]      e.g.  conv_bench.py 5 0 26
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from doc2tex_amd import _lib

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
reserves = [int(a) for a in sys.argv[2:]] or [0]
lib = _lib.require_device()
# name, B, H, W, Cin, Cout, k, stride, pad, residual
LAYERS = [
    ("512->512 3x3 @16x129 (+res)", 64, 16, 129, 512, 512, (3, 3), (1, 1), (1, 1), True),
    ("256->512 3x3 @16x129", 64, 16, 129, 256, 512, (3, 3), (1, 1), (1, 1), False),
    ("256->256 3x3 @32x128 (+res)", 64, 32, 128, 256, 256, (3, 3), (1, 1), (1, 1), True),
    ("128->128 3x3 @64x256", 64, 64, 256, 128, 128, (3, 3), (1, 1), (1, 1), False),
    ("64->128 3x3 @64x256", 64, 64, 256, 64, 128, (3, 3), (1, 1), (1, 1), False),
    ("conv4_1 2x2 s(2,1) p(0,1)", 64, 16, 129, 512, 512, (2, 2), (2, 1), (0, 1), False),
    ("1x1 shortcut 256->512", 64, 16, 129, 256, 512, (1, 1), (1, 1), (0, 0), False),
    ("ragged: 3 x 7x37, 128->384", 3, 7, 37, 128, 384, (3, 3), (1, 1), (1, 1), True),
    ("ragged: 1 x 5x9, 64->128, 2x2 s2", 1, 5, 9, 64, 128, (2, 2), (2, 2), (0, 0), False),
]
g = torch.Generator().manual_seed(0)
for name, B, H, W, Cin, Cout, k, st, pd, use_res in LAYERS:
    x = torch.randn(B, H, W, Cin, generator=g).cuda()
    w = (torch.randn(Cout, k[0], k[1], Cin, generator=g) * (2.0 / (k[0] * k[1] * Cin)) ** 0.5).cuda()
    b = torch.randn(Cout, generator=g).cuda()
    OH, OW = (H + 2 * pd[0] - k[0]) // st[0] + 1, (W + 2 * pd[1] - k[1]) // st[1] + 1
    res = torch.randn(B, OH, OW, Cout, generator=g).cuda() if use_res else None
    outs = {}
    for rep in range(reps):
        for kind, rc_ in [(0, 0)] + [(3, r) for r in reserves] + [(8, 0)]:
            assert lib.d2t_op_set_conv_kernel(kind, rc_) == 0
            y = torch.full((B, OH, OW, Cout), float("nan"), device="cuda")
            t0 = time.perf_counter()
            rc = lib.d2t_op_conv2d_bf16x3_split(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(res), _lib.ptr(y), B, H, W, Cin,
                                                Cout, k[0], k[1], st[0], st[1], pd[0], pd[1], 1, _lib.stream_of(x))
            assert rc == 0, rc
            torch.cuda.synchronize()
            if rep == 0:
                outs[(kind, rc_)] = y
    ref = outs[(0, 0)]
    assert torch.isfinite(ref).all()
    # the 16x16x32 build sums 32 k inside one MFMA: equal to fp32 rounding, not bit for bit; against float64 both are equally close
    ref64 = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w.permute(0, 3, 1, 2).double(), b.double(), st, pd)
    ref64 = ref64.permute(0, 2, 3, 1) + (res.double() if use_res else 0)
    ref64 = torch.relu(ref64)
    scale = float(ref64.abs().max())
    err = {"128x128": float((ref.double() - ref64).abs().max()) / scale}
    for r in reserves:
        err[f"p16/reserve{r}"] = float((outs[(3, r)].double() - ref64).abs().max()) / scale
        assert torch.isfinite(outs[(3, r)]).all()
        assert err[f"p16/reserve{r}"] <= 3 * max(err["128x128"], 1e-7), (name, err)
    for r in reserves[1:]:  # the grid (reserved CUs) never shows in the values
        assert torch.equal(outs[(3, r)], outs[(3, reserves[0])]), name
    err["fp16x2 / p16"] = float((outs[(8, 0)].double() - ref64).abs().max()) / scale
    assert torch.isfinite(outs[(8, 0)]).all() and err["fp16x2 / p16"] <= 2e-3, (name, err)
    print(f"{name}: M={B * OH * OW} N={Cout} K={k[0] * k[1] * Cin}  "
          f"max error / max |y| against float64: { {k_: f'{v:.2e}' for k_, v in err.items()} }", flush=True)
    if os.environ.get("D2T_CONV_ABL"):
        break  # ablation probes: the dominant shape only
print("ok")
