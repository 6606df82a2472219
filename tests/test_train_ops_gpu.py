"""GPU: every BACKWARD kernel of the training step, one op at a time through the C-ABI test entry points
(d2t_op_train_*: one node of the training tape, forward + backward, on caller tensors), against float64 torch autograd
of the same op.  Both arithmetic modes where the op has them (fp32 MFMA / split-bf16 for the convolution, data-gradient
and weight-gradient GEMMs).

Tolerance = max |engine - fp64| relative to max |fp64| of that tensor: 1e-4 for fp32 arithmetic (fp32 accumulation
over up to ~10^5 terms in the weight gradients), 1e-4 for split-bf16 (2^-16 per product, averaged over the same
sums).  A ReLU inside the op is applied on the ENGINE's own forward values in both the engine and the reference
(reference mask = sign of the fp64 pre-activation; inputs are generated so that no pre-activation is within 1e-3 of
zero), so no decision can flip and the comparison is tight everywhere.
"""
import pytest
import torch
import torch.nn.functional as F

from doc2tex_amd import _lib

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = 1e-4


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def _rel(a, b):
    b = b.double()
    return float((a.double().cpu() - b).abs().max() / max(float(b.abs().max()), 1e-30))


def _nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().to(DEV)


def _nchw(t):
    return t.cpu().permute(0, 3, 1, 2)


def _away_from_zero(pre, margin=1e-3):
    """Shift a pre-activation tensor so that no element is within `margin` of the ReLU kink."""
    return pre + torch.where(pre.abs() < margin, torch.sign(pre) * 2 * margin + (pre == 0) * 2 * margin, torch.zeros_like(pre))


CONV_CASES = [
    # B, Cin, H, W, Cout, k, stride, pad, bn, relu, residual       (which reference layer)
    (2, 32, 16, 24, 64, (3, 3), (1, 1), (1, 1), True, True, False),    # conv0_2 + bn0_2
    (3, 64, 12, 20, 128, (3, 3), (1, 1), (1, 1), True, True, True),    # BasicBlock conv2 + bn2 + residual + ReLU
    (2, 64, 9, 13, 128, (1, 1), (1, 1), (0, 0), True, False, False),   # downsample 1x1 + BN, no ReLU
    (2, 256, 8, 17, 512, (3, 3), (1, 1), (1, 1), True, True, False),   # layer3 entry
    (2, 512, 6, 11, 512, (2, 2), (2, 1), (0, 1), True, True, False),   # conv4_1: stride (2,1), pad (0,1) -> dilated dgrad
    (2, 512, 4, 12, 512, (2, 2), (1, 1), (0, 0), True, True, False),   # conv4_2
    (1, 512, 16, 129, 512, (3, 3), (1, 1), (1, 1), True, True, True),  # the hot 16x129 layer (odd width)
    (2, 512, 7, 9, 256, (2, 2), (2, 2), (0, 0), False, False, False),  # patch embedding: stride 2, bias, no BN (odd H: floor)
    (4, 128, 24, 37, 256, (3, 3), (1, 1), (1, 1), True, True, False),  # layer2 entry: 3552 pixels, several row chunks of the
                                                                       # record weight-gradient kernel, the last one ragged
    (3, 128, 10, 18, 128, (3, 3), (1, 1), (1, 1), True, True, True),   # layer1.conv2: the 128 x 128 tile of that kernel (the
                                                                       # cases above take 64 x 32, 128 x 64, 256 x 128, 256 x 256)
]


@pytest.mark.parametrize("bf16x3", [0, 1], ids=["fp32", "bf16x3"])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_bn_relu_backward(case, bf16x3):
    B, Cin, H, W, Cout, k, stride, pad, bn, relu, use_res = case
    lib = _lib.require_device()
    x = _rand(B, Cin, H, W, seed=1)
    w = _rand(Cout, Cin, *k, seed=2, scale=(2.0 / (Cin * k[0] * k[1])) ** 0.5)
    OH = (H + 2 * pad[0] - k[0]) // stride[0] + 1
    OW = (W + 2 * pad[1] - k[1]) // stride[1] + 1
    bias = None if bn else _rand(Cout, seed=3, scale=0.1)
    gamma = (torch.rand(Cout, generator=torch.Generator().manual_seed(4)) + 0.5) if bn else None
    beta = _rand(Cout, seed=5, scale=0.1) if bn else None
    res = _rand(B, Cout, OH, OW, seed=6) if use_res else None
    dy = _rand(B, Cout, OH, OW, seed=7)

    # float64 reference (training-mode BatchNorm: batch statistics, biased variance, eps 1e-5)
    xd, wd = x.double().requires_grad_(True), w.double().requires_grad_(True)
    leaves = [xd, wd]
    z = F.conv2d(xd, wd, None if bias is None else bias.double(), stride, pad)
    if bn:
        gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
        leaves += [gd, bd]
        z = F.batch_norm(z, None, None, gd, bd, training=True, eps=1e-5)
    else:
        bid = bias.double().requires_grad_(True)
        leaves.append(bid)
        z = F.conv2d(xd, wd, bid, stride, pad)
    if use_res:
        # choose the residual so that the ReLU input stays clear of zero: no decision can differ between fp32 and fp64
        rd = res.double()
        if relu:
            rd = _away_from_zero(z.detach() + rd) - z.detach()
            res = rd.float()
            rd = res.double()
        rd = rd.requires_grad_(True)
        leaves.append(rd)
        z = z + rd
    elif relu:
        # no residual to adjust: move beta instead (a per-channel shift cannot clear every element, so mask the gradient of
        # the few near-zero ones out of the comparison by zeroing dy there)
        near = z.detach().abs() < 1e-3
        dy = dy.masked_fill(near.float().bool(), 0.0)
    y_ref = F.relu(z) if relu else z
    grads = torch.autograd.grad(y_ref, leaves, dy.double())

    xg, wg = _nhwc(x), w.contiguous().to(DEV)
    args = dict(bias=bias, gamma=gamma, beta=beta)
    dev = {k_: (None if v is None else v.contiguous().to(DEV)) for k_, v in args.items()}
    rg = _nhwc(res) if use_res else None
    dyg = _nhwc(dy)
    y = torch.full((B, OH, OW, Cout), float("nan"), device=DEV)
    dx = torch.full((B, H, W, Cin), float("nan"), device=DEV)
    dw = torch.full_like(wg, float("nan"))
    dbias = torch.full((Cout,), float("nan"), device=DEV)
    dgam, dbet = torch.full_like(dbias, float("nan")), torch.full_like(dbias, float("nan"))
    dres = torch.full_like(y, float("nan")) if use_res else None
    rc = lib.d2t_op_train_conv(_lib.ptr(xg), _lib.ptr(wg), _lib.ptr(dev["bias"]), _lib.ptr(dev["gamma"]), _lib.ptr(dev["beta"]),
                               _lib.ptr(rg), _lib.ptr(dyg), _lib.ptr(y), _lib.ptr(dx), _lib.ptr(dw), _lib.ptr(dbias),
                               _lib.ptr(dgam), _lib.ptr(dbet), _lib.ptr(dres), B, H, W, Cin, Cout, k[0], k[1], stride[0],
                               stride[1], pad[0], pad[1], int(relu), bf16x3, _lib.stream_of(xg))
    assert rc == 0
    torch.cuda.synchronize()
    tol = TOL if not bf16x3 else 2 * TOL
    assert _rel(_nchw(y), y_ref.detach()) <= tol
    got = {"dx": _nchw(dx), "dw": dw.cpu()}
    want = {"dx": grads[0], "dw": grads[1]}
    if bn:
        got.update(dgamma=dgam.cpu(), dbeta=dbet.cpu())
        want.update(dgamma=grads[2], dbeta=grads[3])
    else:
        got["dbias"], want["dbias"] = dbias.cpu(), grads[2]
    if use_res:
        got["dres"], want["dres"] = _nchw(dres), grads[-1]
    errs = {k_: _rel(got[k_], want[k_]) for k_ in got}
    assert max(errs.values()) <= tol, errs


def test_stem_conv_backward():
    """conv0_1 (1 -> 32 channels, 3x3) + BatchNorm + ReLU: its own forward and weight-gradient kernels."""
    lib = _lib.require_device()
    B, H, W, Cout = 2, 20, 36, 32
    x = _rand(B, 1, H, W, seed=8)
    w = _rand(Cout, 1, 3, 3, seed=9, scale=0.3)
    gamma = torch.rand(Cout, generator=torch.Generator().manual_seed(10)) + 0.5
    beta = _rand(Cout, seed=11, scale=0.1)
    dy = _rand(B, Cout, H, W, seed=12)
    wd, gd, bd = (t.double().requires_grad_(True) for t in (w, gamma, beta))
    z = F.batch_norm(F.conv2d(x.double(), wd, None, 1, 1), None, None, gd, bd, training=True, eps=1e-5)
    dy = dy.masked_fill(z.detach().abs() < 1e-3, 0.0)
    grads = torch.autograd.grad(F.relu(z), [wd, gd, bd], dy.double())
    xg, wg, dyg = _nhwc(x), w.to(DEV), _nhwc(dy)
    gg, bg = gamma.to(DEV), beta.to(DEV)  # named: a temporary would be freed (and its block reused) before the call runs
    y = torch.empty((B, H, W, Cout), device=DEV)
    dw, dgam, dbet = torch.empty_like(wg), torch.empty(Cout, device=DEV), torch.empty(Cout, device=DEV)
    rc = lib.d2t_op_train_conv(_lib.ptr(xg), _lib.ptr(wg), None, _lib.ptr(gg), _lib.ptr(bg), None,
                               _lib.ptr(dyg), _lib.ptr(y), None, _lib.ptr(dw), None, _lib.ptr(dgam), _lib.ptr(dbet), None,
                               B, H, W, 1, Cout, 3, 3, 1, 1, 1, 1, 1, 0, _lib.stream_of(xg))
    assert rc == 0
    torch.cuda.synchronize()
    assert _rel(_nchw(y), F.relu(z).detach()) <= TOL
    assert _rel(dw.cpu(), grads[0]) <= TOL and _rel(dgam.cpu(), grads[1]) <= TOL and _rel(dbet.cpu(), grads[2]) <= TOL


@pytest.mark.parametrize("bf16x3", [0, 1], ids=["fp32", "bf16x3"])
@pytest.mark.parametrize("M,K,N,relu,use_res", [(302, 256, 768, False, False), (302, 256, 256, False, True),
                                                 (453, 256, 1024, True, False), (453, 1024, 256, False, True),
                                                 (151, 256, 500, False, False)])  # the vocabulary projection: N % 32 != 0
def test_linear_backward(M, K, N, relu, use_res, bf16x3):
    lib = _lib.require_device()
    x, w, b = _rand(M, K, seed=20), _rand(N, K, seed=21, scale=K ** -0.5), _rand(N, seed=22, scale=0.1)
    res = _rand(M, N, seed=23) if use_res else None
    dy = _rand(M, N, seed=24)
    xd, wd, bd = (t.double().requires_grad_(True) for t in (x, w, b))
    z = F.linear(xd, wd, bd)
    if relu:
        dy = dy.masked_fill(z.detach().abs() < 1e-3, 0.0)
        z = F.relu(z)
    leaves = [xd, wd, bd]
    if use_res:
        rd = res.double().requires_grad_(True)
        leaves.append(rd)
        z = z + rd
    grads = torch.autograd.grad(z, leaves, dy.double())
    xg, wg, bg, dyg = x.to(DEV), w.to(DEV), b.to(DEV), dy.to(DEV)
    rg = res.to(DEV) if use_res else None
    y, dx, dw, db = torch.empty(M, N, device=DEV), torch.empty_like(xg), torch.empty_like(wg), torch.empty_like(bg)
    dres = torch.empty(M, N, device=DEV) if use_res else None
    rc = lib.d2t_op_train_linear(_lib.ptr(xg), _lib.ptr(wg), _lib.ptr(bg), _lib.ptr(rg), _lib.ptr(dyg), _lib.ptr(y),
                                 _lib.ptr(dx), _lib.ptr(dw), _lib.ptr(db), _lib.ptr(dres), M, K, N, int(relu), bf16x3,
                                 _lib.stream_of(xg))
    assert rc == 0
    torch.cuda.synchronize()
    tol = TOL if not bf16x3 else 2 * TOL
    errs = {"y": _rel(y, z.detach()), "dx": _rel(dx, grads[0]), "dw": _rel(dw, grads[1]), "db": _rel(db, grads[2])}
    if use_res:
        errs["dres"] = _rel(dres, grads[3])
    assert max(errs.values()) <= tol, errs


@pytest.mark.parametrize("rows,D,eps", [(522, 256, 1e-6), (302, 256, 1e-5), (390, 512, 1e-5)])
def test_layernorm_backward(rows, D, eps):
    lib = _lib.require_device()
    x, g, b, dy = _rand(rows, D, seed=30, scale=2.0), _rand(D, seed=31) * 0.2 + 1.0, _rand(D, seed=32, scale=0.1), _rand(rows, D, seed=33)
    xd, gd, bd = (t.double().requires_grad_(True) for t in (x, g, b))
    yr = F.layer_norm(xd, (D,), gd, bd, eps)
    grads = torch.autograd.grad(yr, [xd, gd, bd], dy.double())
    xg, gg, bg, dyg = x.to(DEV), g.to(DEV), b.to(DEV), dy.to(DEV)
    y, dx, dg, db = torch.empty_like(xg), torch.empty_like(xg), torch.empty_like(gg), torch.empty_like(bg)
    rc = lib.d2t_op_train_layernorm(_lib.ptr(xg), _lib.ptr(gg), _lib.ptr(bg), _lib.ptr(dyg), _lib.ptr(y), _lib.ptr(dx),
                                    _lib.ptr(dg), _lib.ptr(db), rows, D, eps, _lib.stream_of(xg))
    assert rc == 0
    torch.cuda.synchronize()
    errs = {"y": _rel(y, yr.detach()), "dx": _rel(dx, grads[0]), "dg": _rel(dg, grads[1]), "db": _rel(db, grads[2])}
    assert max(errs.values()) <= TOL, errs


@pytest.mark.parametrize("nb,Lq,Lk,heads,hd,causal,pad", [
    (3, 37, 37, 8, 32, 0, False),    # ViT block attention (no mask)
    (2, 261, 261, 8, 32, 0, False),  # ViT at the headline token count
    (3, 25, 25, 8, 32, 1, True),     # decoder self-attention: causal + PAD key-padding mask
    (3, 25, 70, 8, 32, 0, False),    # decoder cross-attention over the memory
    (2, 23, 23, 8, 64, 1, True),     # d_model 512 (config C1): head_dim 64
])
def test_attention_backward(nb, Lq, Lk, heads, hd, causal, pad):
    lib = _lib.require_device()
    D = heads * hd
    q, kv, dy = _rand(nb * Lq, D, seed=40), _rand(nb * Lk, 2 * D, seed=41), _rand(nb * Lq, D, seed=42)
    keytok = None
    if pad:  # rows end in PAD (0) tokens; the first key of a row is never PAD ([GO])
        keytok = torch.randint(4, 100, (nb, Lk), generator=torch.Generator().manual_seed(43))
        for i in range(nb):
            keytok[i, Lk - 3 * i - 2:] = 0
    qd, kvd = q.double().requires_grad_(True), kv.double().requires_grad_(True)
    Q = qd.view(nb, Lq, heads, hd).transpose(1, 2)
    Kk = kvd[:, :D].reshape(nb, Lk, heads, hd).transpose(1, 2)
    Vv = kvd[:, D:].reshape(nb, Lk, heads, hd).transpose(1, 2)
    s = Q @ Kk.transpose(-1, -2) / hd ** 0.5
    if causal:
        s = s + torch.full((Lq, Lk), float("-inf"), dtype=torch.float64).triu(1)
    if pad:
        s = s.masked_fill((keytok == 0)[:, None, None, :], float("-inf"))
    yr = (torch.softmax(s, -1) @ Vv).transpose(1, 2).reshape(nb * Lq, D)
    grads = torch.autograd.grad(yr, [qd, kvd], dy.double())
    qg, kvg, dyg = q.to(DEV), kv.to(DEV), dy.to(DEV)
    ktg = keytok.to(DEV) if pad else None
    y, dq, dkv = torch.empty_like(qg), torch.empty_like(qg), torch.empty_like(kvg)
    rc = lib.d2t_op_train_attention(_lib.ptr(qg), _lib.ptr(kvg), _lib.ptr(ktg), _lib.ptr(dyg), _lib.ptr(y), _lib.ptr(dq),
                                    _lib.ptr(dkv), nb, Lq, Lk, heads, hd, causal, _lib.stream_of(qg))
    assert rc == 0
    torch.cuda.synchronize()
    errs = {"y": _rel(y, yr.detach()), "dq": _rel(dq, grads[0]), "dkv": _rel(dkv, grads[1])}
    assert max(errs.values()) <= TOL, errs


@pytest.mark.parametrize("B,H,W,C,stride,pad", [(2, 16, 24, 64, (2, 2), (0, 0)), (2, 9, 13, 128, (2, 2), (0, 0)),
                                                (2, 8, 17, 256, (2, 1), (0, 1))])  # the overlapping k2 s(2,1) p(0,1) pool
def test_maxpool_backward(B, H, W, C, stride, pad):
    lib = _lib.require_device()
    x = _rand(B, C, H, W, seed=50)
    x = torch.relu(x)  # post-ReLU input, as in the network: ties at zero exercise torch's first-maximum rule
    OH, OW = (H + 2 * pad[0] - 2) // stride[0] + 1, (W + 2 * pad[1] - 2) // stride[1] + 1
    dy = _rand(B, C, OH, OW, seed=51)
    xd = x.double().requires_grad_(True)
    yr = F.max_pool2d(xd, 2, stride, pad)
    (gx,) = torch.autograd.grad(yr, [xd], dy.double())
    xg, dyg = _nhwc(x), _nhwc(dy)
    y, dx = torch.empty((B, OH, OW, C), device=DEV), torch.empty_like(xg)
    rc = lib.d2t_op_train_maxpool(_lib.ptr(xg), _lib.ptr(dyg), _lib.ptr(y), _lib.ptr(dx), B, H, W, C, stride[0], stride[1],
                                  pad[0], pad[1], _lib.stream_of(xg))
    assert rc == 0
    torch.cuda.synchronize()
    assert torch.equal(_nchw(y), yr.detach().float())
    assert _rel(_nchw(dx), gx) <= 1e-6


@pytest.mark.parametrize("rows,V", [(453, 500), (4832, 500), (37, 93), (64, 1000)])
def test_fused_cross_entropy(rows, V):
    """d2t_ce_forward / d2t_ce_backward (doc2tex_amd.loss.CrossEntropyLoss) against float64 torch: per-row losses with PAD
    rows ignored, the reference's `cost.mean()` over ALL positions (engine/training.py:126), and 'mean' / 'sum' reductions."""
    from doc2tex_amd.loss import CrossEntropyLoss, create_criterion
    g = torch.Generator().manual_seed(rows + V)
    logits = _rand(rows, V, seed=60, scale=3.0)
    target = torch.randint(0, V, (rows,), generator=g)
    target[::5] = 0  # PAD
    for reduction in ("none", "mean", "sum"):
        x = logits.clone().to(DEV).requires_grad_(True)
        crit = create_criterion("entropy", {"ignore_index": 0, "reduction": reduction})
        assert isinstance(crit, CrossEntropyLoss)
        out = crit(x, target.to(DEV))
        xd = logits.double().requires_grad_(True)
        ref = F.cross_entropy(xd, target, ignore_index=0, reduction=reduction)
        if reduction == "none":
            assert out.shape == (rows,) and float(out[::5].abs().max()) == 0.0
            assert _rel(out, ref.detach()) <= 1e-6
            out.mean().backward()   # training.py:126: mean over all positions, ignored ones included
            ref.mean().backward()
        else:
            assert abs(float(out) - float(ref)) <= 1e-6 * max(1.0, abs(float(ref)))
            out.backward()
            ref.backward()
        torch.cuda.synchronize()
        assert _rel(x.grad, xd.grad) <= 1e-6
        assert float(x.grad[::5].abs().max()) == 0.0
    with pytest.raises(NotImplementedError):
        create_criterion("smooth", {"classes": V})
    with pytest.raises(RuntimeError):
        CrossEntropyLoss(ignore_index=0, reduction="none")(logits, target)  # CPU tensors: no fallback
