"""GPU: every HIP kernel, called through the C-ABI, against a plain fp32 CPU
PyTorch evaluation of the same op (tolerances written per test)."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from doc2tex_amd import _lib

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def _conv(x_nchw, w_oihw, bias, res_nchw, stride, pad, act):
    """Run d2t_op_conv2d; returns NCHW CPU tensor."""
    lib = _lib.require_device()
    B, Cin, H, W = x_nchw.shape
    Cout, _, KH, KW = w_oihw.shape
    x = x_nchw.permute(0, 2, 3, 1).contiguous().to(DEV)
    w = w_oihw.permute(0, 2, 3, 1).contiguous().to(DEV)
    OH = (H + 2 * pad[0] - KH) // stride[0] + 1
    OW = (W + 2 * pad[1] - KW) // stride[1] + 1
    y = torch.full((B, OH, OW, Cout), float("nan"), device=DEV)
    b = bias.to(DEV) if bias is not None else None
    r = res_nchw.permute(0, 2, 3, 1).contiguous().to(DEV) if res_nchw is not None else None
    rc = lib.d2t_op_conv2d(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(r), _lib.ptr(y), B, H, W, Cin, Cout, KH, KW,
                           stride[0], stride[1], pad[0], pad[1], act, _lib.stream_of(x))
    assert rc == 0
    torch.cuda.synchronize()
    return y.cpu().permute(0, 3, 1, 2)


def _ref_conv(x, w, bias, res, stride, pad, act):
    y = F.conv2d(x.double(), w.double(), None if bias is None else bias.double(), stride, pad)
    if res is not None:
        y = y + res.double()
    if act == _lib.ACT_RELU:
        y = F.relu(y)
    elif act == _lib.ACT_GELU:
        y = F.gelu(y)
    return y.float()


CONV_CASES = [
    # B, Cin, H, W, Cout, k, stride, pad, act, residual      (which reference layer)
    (2, 32, 16, 24, 64, (3, 3), (1, 1), (1, 1), 1, False),    # conv0_2 (128x64 tile path needs big M; here 64x64)
    (3, 64, 12, 20, 128, (3, 3), (1, 1), (1, 1), 1, True),    # BasicBlock conv2 + residual
    (2, 64, 9, 13, 128, (1, 1), (1, 1), (0, 0), 0, False),    # downsample 1x1
    (2, 256, 8, 17, 512, (3, 3), (1, 1), (1, 1), 1, False),   # layer3 entry
    (2, 512, 6, 11, 512, (2, 2), (2, 1), (0, 1), 1, False),   # conv4_1
    (2, 512, 4, 12, 512, (2, 2), (1, 1), (0, 0), 1, False),   # conv4_2
    (1, 512, 16, 129, 512, (3, 3), (1, 1), (1, 1), 1, True),  # the hot 16x129 conv (odd width, ragged tiles)
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_mfma_vs_torch(case):
    B, Cin, H, W, Cout, k, stride, pad, act, use_res = case
    x = _rand(B, Cin, H, W, seed=1)
    w = _rand(Cout, Cin, *k, seed=2, scale=(2.0 / (Cin * k[0] * k[1])) ** 0.5)
    bias = _rand(Cout, seed=3, scale=0.1)
    OH = (H + 2 * pad[0] - k[0]) // stride[0] + 1
    OW = (W + 2 * pad[1] - k[1]) // stride[1] + 1
    res = _rand(B, Cout, OH, OW, seed=4) if use_res else None
    y = _conv(x, w, bias, res, stride, pad, act)
    ref = _ref_conv(x, w, bias, res, stride, pad, act)
    err = float((y - ref).abs().max())
    assert err <= 2e-4, err  # fp32 accumulation over K <= 4608, |y| ~ O(3)


def test_conv_large_m_tile_paths():
    """M large enough to take the 128x128 and 128x64 tile configurations (grid >= 256 tiles)."""
    for Cin, Cout, H, W in [(32, 64, 128, 260), (128, 256, 64, 140)]:
        x = _rand(1, Cin, H, W, seed=5)
        w = _rand(Cout, Cin, 3, 3, seed=6, scale=(2.0 / (Cin * 9)) ** 0.5)
        b = _rand(Cout, seed=7, scale=0.1)
        y = _conv(x, w, b, None, (1, 1), (1, 1), 1)
        ref = _ref_conv(x, w, b, None, (1, 1), (1, 1), 1)
        assert float((y - ref).abs().max()) <= 2e-4


def test_stem_conv_cin1():
    x = _rand(2, 1, 20, 36, seed=8)
    w = _rand(32, 1, 3, 3, seed=9, scale=0.3)
    b = _rand(32, seed=10, scale=0.1)
    y = _conv(x, w, b, None, (1, 1), (1, 1), 1)
    ref = _ref_conv(x, w, b, None, (1, 1), (1, 1), 1)
    assert float((y - ref).abs().max()) <= 1e-5


@pytest.mark.parametrize("M,K,N,act,res", [(16704 // 8, 256, 768, 0, False), (300, 1024, 256, 0, True),
                                            (200, 256, 1024, 2, False), (64, 256, 768, 0, False),
                                            (64, 1024, 256, 0, True), (5, 256, 500, 0, False),
                                            (37, 512, 1536, 1, False), (64, 256, 500, 0, False)])
def test_linear_vs_torch(M, K, N, act, res):
    lib = _lib.require_device()
    x = _rand(M, K, seed=11)
    w = _rand(N, K, seed=12, scale=K ** -0.5)
    b = _rand(N, seed=13, scale=0.1)
    r = _rand(M, N, seed=14) if res else None
    xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)
    rd = r.to(DEV) if res else None
    y = torch.full((M, N), float("nan"), device=DEV)
    rc = lib.d2t_op_linear(_lib.ptr(xd), _lib.ptr(wd), _lib.ptr(bd), _lib.ptr(rd), _lib.ptr(y), M, K, N, act,
                           _lib.stream_of(xd))
    assert rc == 0
    ref = F.linear(x.double(), w.double(), b.double())
    if res:
        ref = ref + r.double()
    ref = F.relu(ref) if act == 1 else F.gelu(ref) if act == 2 else ref
    err = float((y.cpu() - ref.float()).abs().max())
    assert err <= 1e-4, err


@pytest.mark.parametrize("H,W,C,s,p", [(16, 24, 64, (2, 2), (0, 0)), (9, 11, 128, (2, 2), (0, 0)),
                                        (8, 13, 256, (2, 1), (0, 1))])
def test_maxpool_vs_torch(H, W, C, s, p):
    lib = _lib.require_device()
    x = _rand(2, C, H, W, seed=15)
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    ref = F.max_pool2d(x, 2, s, p)
    y = torch.empty((2, ref.shape[2], ref.shape[3], C), device=DEV)
    assert lib.d2t_op_maxpool2x2(_lib.ptr(xd), _lib.ptr(y), 2, H, W, C, s[0], s[1], p[0], p[1], _lib.stream_of(xd)) == 0
    assert torch.equal(y.cpu().permute(0, 3, 1, 2), ref)  # bit-exact


@pytest.mark.parametrize("D,eps", [(256, 1e-6), (512, 1e-5)])
def test_layernorm_vs_torch(D, eps):
    lib = _lib.require_device()
    x = _rand(77, D, seed=16, scale=3.0) + 0.5
    g, b = _rand(D, seed=17) * 0.2 + 1.0, _rand(D, seed=18) * 0.1
    xd, gd, bd = x.to(DEV), g.to(DEV), b.to(DEV)
    y = torch.empty_like(xd)
    assert lib.d2t_op_layernorm(_lib.ptr(xd), _lib.ptr(gd), _lib.ptr(bd), _lib.ptr(y), 77, D, eps,
                                _lib.stream_of(xd)) == 0
    ref = F.layer_norm(x.double(), (D,), g.double(), b.double(), eps).float()
    assert float((y.cpu() - ref).abs().max()) <= 2e-6


@pytest.mark.parametrize("B,N,heads", [(2, 10, 8), (2, 261, 8), (1, 406, 8), (3, 65, 16)])
def test_vit_attention_vs_torch(B, N, heads):
    lib = _lib.require_device()
    C_ = heads * 32
    qkv = _rand(B, N, 3, heads, 32, seed=19)
    qd = qkv.to(DEV)
    y = torch.empty((B, N, C_), device=DEV)
    assert lib.d2t_op_vit_attention(_lib.ptr(qd), _lib.ptr(y), B, N, heads, _lib.stream_of(qd)) == 0
    q, k, v = [t.double() for t in qkv.permute(2, 0, 3, 1, 4)]
    a = ((q @ k.transpose(-2, -1)) * 32 ** -0.5).softmax(-1)
    ref = (a @ v).transpose(1, 2).reshape(B, N, C_).float()
    assert float((y.cpu() - ref).abs().max()) <= 2e-6


@pytest.mark.parametrize("B,heads,hd,L,Lmax", [(3, 8, 32, 1, 152), (3, 8, 32, 152, 152), (2, 8, 32, 261, 261),
                                                (2, 8, 64, 195, 195), (5, 8, 64, 77, 152), (1, 8, 32, 406, 406)])
def test_decode_attention_vs_torch(B, heads, hd, L, Lmax):
    lib = _lib.require_device()
    q = _rand(B, heads * hd, seed=20)
    k = _rand(B, heads, Lmax, hd, seed=21)
    v = _rand(B, heads, Lmax, hd, seed=22)
    qd, kd, vd = q.to(DEV), k.to(DEV), v.to(DEV)
    y = torch.empty((B, heads * hd), device=DEV)
    assert lib.d2t_op_decode_attention(_lib.ptr(qd), _lib.ptr(kd), _lib.ptr(vd), _lib.ptr(y), B, heads, hd, L, Lmax,
                                       _lib.stream_of(qd)) == 0
    qq = q.view(B, heads, 1, hd).double()
    a = ((qq @ k[:, :, :L].double().transpose(-2, -1)) * hd ** -0.5).softmax(-1)
    ref = (a @ v[:, :, :L].double()).reshape(B, heads * hd).float()
    assert float((y.cpu() - ref).abs().max()) <= 2e-6


@pytest.mark.parametrize("case", [c for c in CONV_CASES if c[1] % 32 == 0])
def test_conv_bf16x3_split_planes_vs_torch(case):
    """Split-activation variant: input / residual / output as bf16 hi+lo planes."""
    lib = _lib.require_device()
    B, Cin, H, W, Cout, k, stride, pad, act, use_res = case
    x = _rand(B, Cin, H, W, seed=1)
    w = _rand(Cout, Cin, *k, seed=2, scale=(2.0 / (Cin * k[0] * k[1])) ** 0.5)
    bias = _rand(Cout, seed=3, scale=0.1)
    OH = (H + 2 * pad[0] - k[0]) // stride[0] + 1
    OW = (W + 2 * pad[1] - k[1]) // stride[1] + 1
    res = _rand(B, Cout, OH, OW, seed=4) if use_res else None
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    wd = w.permute(0, 2, 3, 1).contiguous().to(DEV)
    y = torch.full((B, OH, OW, Cout), float("nan"), device=DEV)
    bd = bias.to(DEV)
    rd = res.permute(0, 2, 3, 1).contiguous().to(DEV) if use_res else None
    rc = lib.d2t_op_conv2d_bf16x3_split(_lib.ptr(xd), _lib.ptr(wd), _lib.ptr(bd), _lib.ptr(rd), _lib.ptr(y), B, H, W, Cin,
                                        Cout, k[0], k[1], stride[0], stride[1], pad[0], pad[1], act, _lib.stream_of(xd))
    assert rc == 0
    torch.cuda.synchronize()
    ref = _ref_conv(x, w, bias, res, stride, pad, act)
    err = float((y.cpu().permute(0, 3, 1, 2) - ref).abs().max())
    assert err <= 4e-4, err


@pytest.mark.parametrize("case", [c for c in CONV_CASES if c[1] % 32 == 0 and c[4] >= 64])
def test_conv_bf16x3_vs_torch(case):
    """Split-bf16 convolution (3 bf16 MFMAs per product): ~2^-17 relative product error, so the
    result must sit ~two orders of magnitude closer to fp64 than a plain-bf16 convolution would."""
    lib = _lib.require_device()
    B, Cin, H, W, Cout, k, stride, pad, act, use_res = case
    x = _rand(B, Cin, H, W, seed=1)
    w = _rand(Cout, Cin, *k, seed=2, scale=(2.0 / (Cin * k[0] * k[1])) ** 0.5)
    bias = _rand(Cout, seed=3, scale=0.1)
    OH = (H + 2 * pad[0] - k[0]) // stride[0] + 1
    OW = (W + 2 * pad[1] - k[1]) // stride[1] + 1
    res = _rand(B, Cout, OH, OW, seed=4) if use_res else None
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    wd = w.permute(0, 2, 3, 1).contiguous().to(DEV)
    y = torch.full((B, OH, OW, Cout), float("nan"), device=DEV)
    bd = bias.to(DEV)
    rd = res.permute(0, 2, 3, 1).contiguous().to(DEV) if use_res else None
    rc = lib.d2t_op_conv2d_bf16x3(_lib.ptr(xd), _lib.ptr(wd), _lib.ptr(bd), _lib.ptr(rd), _lib.ptr(y), B, H, W, Cin, Cout,
                                  k[0], k[1], stride[0], stride[1], pad[0], pad[1], act, _lib.stream_of(xd))
    assert rc == 0
    torch.cuda.synchronize()
    ref = _ref_conv(x, w, bias, res, stride, pad, act)
    err = float((y.cpu().permute(0, 3, 1, 2) - ref).abs().max())
    assert err <= 3e-4, err  # plain bf16 would be ~2e-2 here


def test_conv_bf16x3_large_tiles():
    lib = _lib.require_device()
    for Cin, Cout, H, W in [(32, 64, 128, 260), (128, 256, 64, 140), (512, 512, 16, 129)]:
        x = _rand(2, Cin, H, W, seed=5)
        w = _rand(Cout, Cin, 3, 3, seed=6, scale=(2.0 / (Cin * 9)) ** 0.5)
        b = _rand(Cout, seed=7, scale=0.1)
        xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
        wd = w.permute(0, 2, 3, 1).contiguous().to(DEV)
        bd = b.to(DEV)
        y = torch.full((2, H, W, Cout), float("nan"), device=DEV)
        assert lib.d2t_op_conv2d_bf16x3(_lib.ptr(xd), _lib.ptr(wd), _lib.ptr(bd), None, _lib.ptr(y), 2, H, W, Cin, Cout,
                                        3, 3, 1, 1, 1, 1, 1, _lib.stream_of(xd)) == 0
        torch.cuda.synchronize()
        ref = _ref_conv(x, w, b, None, (1, 1), (1, 1), 1)
        assert float((y.cpu().permute(0, 3, 1, 2) - ref).abs().max()) <= 3e-4


@pytest.mark.parametrize("shape", [
    # B, H, W, Cin, Cout, k, stride, pad, residual
    (1, 257, 256, 32, 128, (3, 3), (1, 1), (1, 1), True),    # 257 tiles: one whole round on 256 CUs + a one-tile last round
    (2, 129, 256, 64, 256, (3, 3), (1, 1), (1, 1), False),   # 516 tiles: two rounds + four
    (2, 16, 129, 512, 512, (3, 3), (1, 1), (1, 1), True),    # the dominant layer's geometry at B = 2
    (3, 7, 37, 128, 384, (3, 3), (1, 1), (1, 1), True),      # ragged rows and a partial last row tile
    (1, 16, 129, 256, 512, (2, 2), (2, 1), (0, 1), False),   # conv4_1's strided, asymmetrically padded window
    (1, 5, 9, 64, 128, (2, 2), (2, 2), (0, 0), False),       # 8 output rows
    (3, 9, 131, 64, 256, (3, 3), (1, 1), (1, 1), True),      # an odd width, two column tiles
    (5, 13, 50, 96, 128, (3, 3), (1, 1), (1, 1), False),     # three channel chunks, tiles that straddle images and rows
    (33, 16, 129, 64, 512, (3, 3), (1, 1), (1, 1), True),    # 267 x 4 tiles: four whole rounds + 44 tiles for the 64 x 128 build
])
def test_pipelined_convolution_vs_the_128_row_kernel_and_float64(shape):
    """The pipelined 256x128 kernel on 16x16x32 MFMAs (default, kind 3) against the 128x128 LDS-DMA kernel on 32x32x16 (kind 0):
    the same records, the same three products per output element in the same K order; the sum inside one MFMA spans 32 k
    instead of 16, so the two agree to fp32 rounding and are equally close to float64.  And per sample: a sample's rows do not
    depend on the batch they are computed in (one kernel per layer shape; the leftover rows of a sparsely filled last round of
    tiles go to the 64 x 128 build of the same body, bit for bit)."""
    lib = _lib.require_device()
    B, H, W, Cin, Cout, k, st, pd, use_res = shape
    x = _rand(B, Cin, H, W, seed=31)
    w = _rand(Cout, Cin, *k, seed=32, scale=(2.0 / (Cin * k[0] * k[1])) ** 0.5)
    b = _rand(Cout, seed=33, scale=0.1)
    OH, OW = (H + 2 * pd[0] - k[0]) // st[0] + 1, (W + 2 * pd[1] - k[1]) // st[1] + 1
    res = _rand(B, Cout, OH, OW, seed=34) if use_res else None
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    wd = w.permute(0, 2, 3, 1).contiguous().to(DEV)
    bd = b.to(DEV)
    rd = res.permute(0, 2, 3, 1).contiguous().to(DEV) if use_res else None

    def run(kind, xs, rs, n):
        assert lib.d2t_op_set_conv_kernel(kind, 0) == 0
        y = torch.full((n, OH, OW, Cout), float("nan"), device=DEV)
        assert lib.d2t_op_conv2d_bf16x3_split(_lib.ptr(xs), _lib.ptr(wd), _lib.ptr(bd), _lib.ptr(rs) if use_res else None, _lib.ptr(y),
                                              n, H, W, Cin, Cout, k[0], k[1], st[0], st[1], pd[0], pd[1], 1, _lib.stream_of(xd)) == 0
        torch.cuda.synchronize()
        return y.cpu()

    try:
        y32 = run(0, xd, rd, B)
        y16 = run(3, xd, rd, B)
        y16_0 = run(3, xd[B - 1:], rd[B - 1:] if use_res else None, 1)
        assert lib.d2t_op_set_conv_kernel(1, 0) != 0  # (the variants of rounds 1-3 are gone from the library)
    finally:
        lib.d2t_op_set_conv_kernel(3, 0)
    assert torch.isfinite(y32).all() and torch.isfinite(y16).all()
    ref = _ref_conv(x, w, b, res, st, pd, 1)
    e32 = float((y32.permute(0, 3, 1, 2) - ref).abs().max())
    e16 = float((y16.permute(0, 3, 1, 2) - ref).abs().max())
    assert e32 <= 4e-4 and e16 <= 4e-4 and e16 <= 1.5 * e32 + 1e-6, (e32, e16)
    assert torch.equal(y16[B - 1:], y16_0)


@pytest.mark.parametrize("shape", [
    # B, H, W, Cin, Cout
    (2, 64, 96, 32, 64),      # conv0_2's kernel (128x64 tile, Cout = 64)
    (3, 33, 47, 32, 64),      # odd height and width: the last row / column belongs to no window
    (2, 32, 64, 64, 128),     # conv1's kernel (pipelined 16x16x32, Cout = 128)
    (1, 45, 63, 128, 256),    # odd sizes, two column tiles
])
def test_convolution_with_the_max_pool_fused(shape):
    """ConvP::pool2: the 2x2 / stride 2 max-pool inside the convolution's epilogue (pooled-order GEMM rows).  Every pooled
    VALUE equals max_pool2d of the unfused kernel's output exactly (max and the (hi, lo) split are monotone), and float64."""
    lib = _lib.require_device()
    B, H, W, Cin, Cout = shape
    x = _rand(B, Cin, H, W, seed=51)
    w = _rand(Cout, Cin, 3, 3, seed=52, scale=(2.0 / (Cin * 9)) ** 0.5)
    b = _rand(Cout, seed=53, scale=0.1)
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    wd = w.permute(0, 2, 3, 1).contiguous().to(DEV)
    bd = b.to(DEV)
    full = torch.full((B, H, W, Cout), float("nan"), device=DEV)
    pooled = torch.full((B, H // 2, W // 2, Cout), float("nan"), device=DEV)
    assert lib.d2t_op_set_conv_kernel(3, 0) == 0
    assert lib.d2t_op_conv2d_bf16x3_split(_lib.ptr(xd), _lib.ptr(wd), _lib.ptr(bd), None, _lib.ptr(full), B, H, W, Cin, Cout,
                                          3, 3, 1, 1, 1, 1, 1, _lib.stream_of(xd)) == 0
    assert lib.d2t_op_conv2d_bf16x3_split_pool(_lib.ptr(xd), _lib.ptr(wd), _lib.ptr(bd), _lib.ptr(pooled), B, H, W, Cin, Cout,
                                               3, 3, 1, 1, 1, 1, 1, _lib.stream_of(xd)) == 0
    torch.cuda.synchronize()
    want = torch.nn.functional.max_pool2d(full.permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1)
    assert torch.isfinite(pooled).all()
    assert torch.equal(pooled, want), float((pooled - want).abs().max())
    ref = torch.nn.functional.max_pool2d(_ref_conv(x, w, b, None, (1, 1), (1, 1), 1), 2, 2)
    assert float((pooled.cpu().permute(0, 3, 1, 2) - ref).abs().max()) <= 4e-4


@pytest.mark.parametrize("shape", [
    # B, H, W, Cin, Cout, k, stride, pad, residual
    (2, 16, 129, 512, 512, (3, 3), (1, 1), (1, 1), True),    # the dominant layer's geometry (pipelined kernel)
    (2, 64, 96, 32, 64, (3, 3), (1, 1), (1, 1), False),      # conv0_2's (128 x 64 kernel)
    (3, 7, 37, 128, 384, (3, 3), (1, 1), (1, 1), True),      # ragged rows and a partial last row tile
    (1, 16, 129, 256, 512, (2, 2), (2, 1), (0, 1), False),   # conv4_1's strided, asymmetrically padded window
    (5, 13, 50, 96, 96, (3, 3), (1, 1), (1, 1), True),       # 64 < Cout < 128: the 128 x 128 kernel
    (2, 9, 33, 256, 512, (1, 1), (1, 1), (0, 0), False),     # a 1x1 shortcut
])
def test_fp16x2_convolution_vs_float64(shape):
    """fp16x2 arithmetic (ConvP::f16; d2t_op_set_conv_kernel kind 8): activations and residual are fp16 records, the weights
    fp16 hi + fp16 lo, a product is x*w_lo + x*w_hi with fp32 accumulation, the result is stored as fp16.  Against float64
    on the SAME rounded inputs the only error left is the fp32 accumulation and the final rounding to fp16 (half an ulp:
    2^-11 relative); a sample's rows do not depend on the batch they are computed in."""
    lib = _lib.require_device()
    B, H, W, Cin, Cout, k, st, pd, use_res = shape
    x = _rand(B, Cin, H, W, seed=41)
    w = _rand(Cout, Cin, *k, seed=42, scale=(2.0 / (Cin * k[0] * k[1])) ** 0.5)
    b = _rand(Cout, seed=43, scale=0.1)
    OH, OW = (H + 2 * pd[0] - k[0]) // st[0] + 1, (W + 2 * pd[1] - k[1]) // st[1] + 1
    res = _rand(B, Cout, OH, OW, seed=44) if use_res else None
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    wd = w.permute(0, 2, 3, 1).contiguous().to(DEV)
    bd = b.to(DEV)
    rd = res.permute(0, 2, 3, 1).contiguous().to(DEV) if use_res else None
    try:
        assert lib.d2t_op_set_conv_kernel(8, 0) == 0
        y = torch.full((B, OH, OW, Cout), float("nan"), device=DEV)
        assert lib.d2t_op_conv2d_bf16x3_split(_lib.ptr(xd), _lib.ptr(wd), _lib.ptr(bd), _lib.ptr(rd), _lib.ptr(y), B, H, W,
                                              Cin, Cout, k[0], k[1], st[0], st[1], pd[0], pd[1], 1, _lib.stream_of(xd)) == 0
        y0 = torch.full((1, OH, OW, Cout), float("nan"), device=DEV)
        assert lib.d2t_op_conv2d_bf16x3_split(_lib.ptr(xd[B - 1:]), _lib.ptr(wd), _lib.ptr(bd), _lib.ptr(rd[B - 1:]) if use_res else None,
                                              _lib.ptr(y0), 1, H, W, Cin, Cout, k[0], k[1], st[0], st[1], pd[0], pd[1], 1,
                                              _lib.stream_of(xd)) == 0
        torch.cuda.synchronize()
    finally:
        lib.d2t_op_set_conv_kernel(3, 0)
    y, y0 = y.cpu(), y0.cpu()
    assert torch.isfinite(y).all() and torch.equal(y[B - 1:], y0)
    x16 = x.half().double()
    wh = w.half()
    w22 = wh.double() + (w - wh.float()).half().double()
    ref = F.conv2d(x16, w22, b.double(), st, pd)
    if use_res:
        ref = ref + res.half().double()
    ref = F.relu(ref)
    got = y.permute(0, 3, 1, 2).double()
    assert torch.equal(got.float().half().float(), got.float())  # what is stored IS fp16
    err = (got - ref).abs()
    # fp16 rounding of the result (2^-11 relative, one more ulp where the fp32 sum lands next to a rounding boundary)
    assert bool((err <= ref.abs() * 2.0 ** -10 + 2e-4).all()), float((err - ref.abs() * 2.0 ** -10).max())


@pytest.mark.parametrize("shape", [(2, 64, 96, 32, 64), (2, 32, 64, 64, 128)])
def test_fp16x2_convolution_with_the_max_pool_fused(shape):
    """The fused 2x2 max-pool in fp16x2 arithmetic equals max_pool2d of the unfused fp16x2 convolution exactly (the maximum of
    fp16 values is one of them)."""
    lib = _lib.require_device()
    B, H, W, Cin, Cout = shape
    x = _rand(B, Cin, H, W, seed=51)
    w = _rand(Cout, Cin, 3, 3, seed=52, scale=(2.0 / (Cin * 9)) ** 0.5)
    b = _rand(Cout, seed=53, scale=0.1)
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    wd = w.permute(0, 2, 3, 1).contiguous().to(DEV)
    bd = b.to(DEV)
    try:
        assert lib.d2t_op_set_conv_kernel(8, 0) == 0
        y = torch.full((B, H, W, Cout), float("nan"), device=DEV)
        assert lib.d2t_op_conv2d_bf16x3_split(_lib.ptr(xd), _lib.ptr(wd), _lib.ptr(bd), None, _lib.ptr(y), B, H, W, Cin, Cout, 3, 3,
                                              1, 1, 1, 1, 1, _lib.stream_of(xd)) == 0
        yp = torch.full((B, H // 2, W // 2, Cout), float("nan"), device=DEV)
        assert lib.d2t_op_conv2d_bf16x3_split_pool(_lib.ptr(xd), _lib.ptr(wd), _lib.ptr(bd), _lib.ptr(yp), B, H, W, Cin, Cout, 3, 3,
                                                   1, 1, 1, 1, 1, _lib.stream_of(xd)) == 0
        torch.cuda.synchronize()
    finally:
        lib.d2t_op_set_conv_kernel(3, 0)
    ref = F.max_pool2d(y.cpu().permute(0, 3, 1, 2), 2, 2)
    assert torch.isfinite(yp).all() and torch.equal(yp.cpu().permute(0, 3, 1, 2), ref)


def test_fp16x2_records_saturate_instead_of_overflowing():
    """fp16 records hold at most 65504: a feature map beyond that saturates (no inf / NaN reaches the next layer).  Trained
    models stay orders of magnitude below (the fixtures' largest feature-map value is a few hundred)."""
    lib = _lib.require_device()
    B, H, W, Cin, Cout = 1, 8, 40, 32, 64
    x = torch.full((B, H, W, Cin), 300.0, device=DEV)
    w = torch.full((Cout, 3, 3, Cin), 1.0, device=DEV)  # sums of up to 9 * 32 * 300 = 86400 > 65504
    b = torch.zeros(Cout, device=DEV)
    try:
        assert lib.d2t_op_set_conv_kernel(8, 0) == 0
        y = torch.full((B, H, W, Cout), float("nan"), device=DEV)
        assert lib.d2t_op_conv2d_bf16x3_split(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), None, _lib.ptr(y), B, H, W, Cin, Cout, 3, 3,
                                              1, 1, 1, 1, 1, _lib.stream_of(x)) == 0
        torch.cuda.synchronize()
    finally:
        lib.d2t_op_set_conv_kernel(3, 0)
    y = y.cpu()
    assert torch.isfinite(y).all() and float(y.max()) == 65504.0  # interior pixels: 86400 -> saturated
    assert float(y[0, 0, 0, 0]) == 4 * 32 * 300.0               # a corner pixel (4 taps): 38400, exact in fp16
