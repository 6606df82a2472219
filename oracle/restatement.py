"""CPU restatement of the doc2tex recognizer forward pass -- TEST INFRASTRUCTURE.

This file is the parity oracle for the HIP engine.  It is NOT part of the
product: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import it.  It restates, in plain functional fp32 PyTorch on the CPU, what
the reference computes on the hot path named by BASELINE.json,
operating directly on a state_dict with the reference's key names.

Pinning: the reference ships no tests or golden vectors for this path
(SURVEY.md 4 / 8c), so this restatement is pinned against the reference itself,
imported in the authoring container by tools/make_golden.py, which (1) checks
every stage of this file against the reference modules on the same seeded
weights/inputs and (2) writes tests/golden/*.npz from the REFERENCE's outputs.
tests/test_oracle_golden.py re-checks this file against those fixtures
anywhere (no reference needed).

Two modes where the reference's algorithm is wasteful:
  faithful=True   op-for-op what the reference does: unfused conv+BN+ReLU
                  (resnet.py:205-245), full-prefix re-decode every step with no
                  KV cache (tfm.py:125-140).  Used for the CPU baseline timing.
  faithful=False  same arithmetic, BN folded into the conv and a KV cache /
                  one-time cross-attention K,V projection (what the engine does).

All file:line citations are relative to /root/reference/doc2tex/.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

PAD, GO, END, UNK = 0, 1, 2, 3  # modules/converter/tfm_converter.py:8,20-34
RESNET_LAYERS = (1, 2, 5, 3)  # feature_extractor/resnet.py:262


# ---------------------------------------------------------------------------
# ResNet backbone  (modules/component/feature_extractor/resnet.py)
# ---------------------------------------------------------------------------
def _bn(x, sd, p, eps=1e-5, bn_train=None):
    # nn.BatchNorm2d.  Eval mode: running statistics.  Train mode (bn_train is a dict): batch statistics
    # (biased variance) and the running-statistics update with momentum 0.1 / unbiased variance, whose new
    # values are left in bn_train[<buffer key>] (module.train() semantics, engine/training.py:94-164).
    if bn_train is None:
        return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"],
                            sd[p + ".weight"], sd[p + ".bias"], False, 0.0, eps)
    rm, rv = sd[p + ".running_mean"].detach().clone(), sd[p + ".running_var"].detach().clone()
    y = F.batch_norm(x, rm, rv, sd[p + ".weight"], sd[p + ".bias"], True, 0.1, eps)
    bn_train[p + ".running_mean"], bn_train[p + ".running_var"] = rm, rv
    return y


def fold_bn(w, sd, p, eps=1e-5):
    """conv weight/bias with eval BatchNorm folded in: y = conv(x, w*s) + (b - m*s)."""
    s = sd[p + ".weight"] / torch.sqrt(sd[p + ".running_var"] + eps)
    return w * s.view(-1, 1, 1, 1), sd[p + ".bias"] - sd[p + ".running_mean"] * s


def _conv_bn(x, sd, conv, bn, stride=1, padding=0, faithful=True, bn_train=None):
    w = sd[conv + ".weight"]
    if faithful or bn_train is not None:
        return _bn(F.conv2d(x, w, None, stride, padding), sd, bn, bn_train=bn_train)
    wf, bf = fold_bn(w, sd, bn)
    return F.conv2d(x, wf, bf, stride, padding)


def _basic_block(x, sd, p, faithful, bn_train=None):
    # BasicBlock.forward, resnet.py:32-48 (ReLU after the residual add, :45-46)
    out = F.relu(_conv_bn(x, sd, p + ".conv1", p + ".bn1", 1, 1, faithful, bn_train))
    out = _conv_bn(out, sd, p + ".conv2", p + ".bn2", 1, 1, faithful, bn_train)
    if (p + ".downsample.0.weight") in sd:  # 1x1 conv + BN, resnet.py:181-192
        x = _conv_bn(x, sd, p + ".downsample.0", p + ".downsample.1", 1, 0, faithful, bn_train)
    return F.relu(out + x)


GC_DROP = 0.25  # ConvMLP's nn.Dropout(drop=0.25), visual_attention.py:86,94: hard-wired, active under module.train()


def _global_context(x, sd, p, drop=None):
    """GlobalContext.forward with use_attn / fuse_add (addon_module/visual_attention.py:147-165): 1x1 conv -> softmax
    over H*W -> attention-pooled channel vector -> ConvMLP (fc1, LayerNorm2d, ReLU, dropout p = 0.25 (eval: off), fc2; the
    hidden width equals the channel count, :88-89) -> added to every position.  `drop(shape, "gc")` returns the scaled keep
    mask of the training-mode dropout (None: evaluation)."""
    B, C, H, W = x.shape
    attn = F.conv2d(x, sd[p + "global_cxt.weight"], sd[p + "global_cxt.bias"]).reshape(B, H * W)
    attn = F.softmax(attn, dim=-1).unsqueeze(-1)
    ctx = torch.bmm(x.reshape(B, C, H * W), attn).unsqueeze(-1)  # [B,C,1,1]
    m = p + "bottleneck_add."
    h = F.conv2d(ctx, sd[m + "fc1.weight"], sd[m + "fc1.bias"])
    h = F.layer_norm(h.permute(0, 2, 3, 1), (h.shape[1],), sd[m + "norm.weight"], sd[m + "norm.bias"], 1e-5).permute(0, 3, 1, 2)
    h = F.relu(h)
    if drop is not None:
        h = h * drop(h.shape, "gc").to(h.dtype)
    h = F.conv2d(h, sd[m + "fc2.weight"], sd[m + "fc2.bias"])
    return x + h


def resnet(x, sd, p, faithful=True, bn_train=None, drop=None):
    """ResNet.forward, resnet.py:205-245.  x [B,1,H,W] -> [B,512,H',W'] (NCHW)."""
    cb = lambda x, c, b, s=1, pd=1: F.relu(_conv_bn(x, sd, p + c, p + b, s, pd, faithful, bn_train))
    x = cb(x, "conv0_1", "bn0_1")
    x = cb(x, "conv0_2", "bn0_2")
    x = F.max_pool2d(x, 2, 2, 0)  # :94
    for i in range(RESNET_LAYERS[0]):
        x = _basic_block(x, sd, f"{p}layer1.{i}", faithful, bn_train)
    if f"{p}layer1.{RESNET_LAYERS[0]}.global_cxt.weight" in sd:  # gcb: GlobalContext closes the stage (resnet.py:200-201)
        x = _global_context(x, sd, f"{p}layer1.{RESNET_LAYERS[0]}.", drop)
    x = cb(x, "conv1", "bn1")
    x = F.max_pool2d(x, 2, 2, 0)  # :106
    for i in range(RESNET_LAYERS[1]):
        x = _basic_block(x, sd, f"{p}layer2.{i}", faithful, bn_train)
    if f"{p}layer2.{RESNET_LAYERS[1]}.global_cxt.weight" in sd:  # gcb: GlobalContext closes the stage (resnet.py:200-201)
        x = _global_context(x, sd, f"{p}layer2.{RESNET_LAYERS[1]}.", drop)
    x = cb(x, "conv2", "bn2")
    x = F.max_pool2d(x, 2, (2, 1), (0, 1))  # :120, implicit -inf padding
    for i in range(RESNET_LAYERS[2]):
        x = _basic_block(x, sd, f"{p}layer3.{i}", faithful, bn_train)
    if f"{p}layer3.{RESNET_LAYERS[2]}.global_cxt.weight" in sd:  # gcb: GlobalContext closes the stage (resnet.py:200-201)
        x = _global_context(x, sd, f"{p}layer3.{RESNET_LAYERS[2]}.", drop)
    x = cb(x, "conv3", "bn3")
    for i in range(RESNET_LAYERS[3]):
        x = _basic_block(x, sd, f"{p}layer4.{i}", faithful, bn_train)
    if f"{p}layer4.{RESNET_LAYERS[3]}.global_cxt.weight" in sd:  # gcb: GlobalContext closes the stage (resnet.py:200-201)
        x = _global_context(x, sd, f"{p}layer4.{RESNET_LAYERS[3]}.", drop)
    x = cb(x, "conv4_1", "bn4_1", (2, 1), (0, 1))  # :139-147
    x = cb(x, "conv4_2", "bn4_2", 1, 0)  # :149-157
    return x


def resnet_out_hw(h, w):
    """Spatial size of resnet() output for an h x w crop."""
    h, w = h // 2, w // 2
    h, w = h // 2, w // 2
    h, w = (h - 2) // 2 + 1, (w + 2 - 2) // 1 + 1  # maxpool3 k2 s(2,1) p(0,1)
    h, w = (h - 2) // 2 + 1, (w + 2 - 2) // 1 + 1  # conv4_1 k2 s(2,1) p(0,1)
    return h - 1, w - 1  # conv4_2 k2 s1 p0


# ---------------------------------------------------------------------------
# Position tables
# ---------------------------------------------------------------------------
def sincos_2d_table(dim, grid_h, grid_w):
    """get_2d_sincos_pos_embed(cls_token=True), common/mae_posembed.py:20-70.

    meshgrid(grid_w, grid_h) puts w first (:28): channels [0, D/2) encode the
    COLUMN index, [D/2, D) the ROW index; each half is [sin | cos] concatenated;
    row 0 is the zero cls row.  float32 numpy like the reference.
    """
    gh = np.arange(grid_h, dtype=np.float32)
    gw = np.arange(grid_w, dtype=np.float32)
    col, row = np.meshgrid(gw, gh)  # each [grid_h, grid_w]

    def one(d, pos):
        omega = np.arange(d // 2, dtype=np.float32)
        omega /= d / 2.0
        omega = 1.0 / 10000 ** omega
        out = np.einsum("m,d->md", pos.reshape(-1), omega)
        return np.concatenate([np.sin(out), np.cos(out)], axis=1)

    emb = np.concatenate([one(dim // 2, col), one(dim // 2, row)], axis=1)
    emb = np.concatenate([np.zeros([1, dim]), emb], axis=0)
    return torch.from_numpy(emb).float().unsqueeze(0)  # [1, 1+gh*gw, dim]


def word_pos_table(d_model, max_len=500, temperature=10000.0):
    """WordPosEnc, prediction_head/addon_module/position_encoding.py:7-22."""
    pe = torch.zeros(max_len, d_model)
    position = torch.arange(0, max_len, dtype=torch.float)
    dim_t = torch.arange(0, d_model, 2, dtype=torch.float)
    div_term = 1.0 / (temperature ** (dim_t / d_model))
    ang = position[:, None] * div_term[None, :]
    pe[:, 0::2] = ang.sin()
    pe[:, 1::2] = ang.cos()
    return pe


def posenc2d_crop(d_model, h, w):
    """PositionalEncoding2D.pe[:, :h, :w], common/postional_encoding.py:105-134,146-157.

    First d/2 channels encode the row, last d/2 the column, sin/cos interleaved.
    Built directly at [d,h,w] instead of the reference's 2000x2000 table."""
    def pe1d(n, d):
        pe = torch.zeros(n, d)
        position = torch.arange(0, n, dtype=torch.float).unsqueeze(1)
        div = torch.exp(torch.arange(0, d, 2).float() * (-math.log(10000.0) / d))
        pe[:, 0::2] = torch.sin(position * div)
        pe[:, 1::2] = torch.cos(position * div)
        return pe
    half = d_model // 2
    pe_h = pe1d(h, half).t()[:, :, None].expand(half, h, w)
    pe_w = pe1d(w, half).t()[:, None, :].expand(half, h, w)
    return torch.cat([pe_h, pe_w], dim=0).contiguous()


# ---------------------------------------------------------------------------
# HybridViT encoder
# ---------------------------------------------------------------------------
def hybrid_embed(x, sd, p, patch=(2, 2), faithful=True, bn_train=None, drop=None):
    """HybridEmbed.forward, seq_modeling/addon_module/patchembed.py:115-141."""
    x = resnet(x, sd, p + "backbone.ConvNet.", faithful, bn_train, drop)
    fh, fw = x.shape[2:]
    pad_h = (-fh) % patch[0]
    pad_w = (-fw) % patch[1]
    x = F.pad(x, (0, pad_w, 0, pad_h))  # zeros right/bottom, :121-126
    y = F.conv2d(x, sd[p + "proj.weight"], sd[p + "proj.bias"], stride=patch)
    return y.flatten(2).transpose(1, 2), (pad_w, pad_h), {"height": x.shape[2], "width": x.shape[3]}


def _vit_block(x, sd, p, heads):
    """Block.forward vision_transformer.py:119-122; Attention :61-81; Mlp :26-32."""
    B, N, C = x.shape
    hd = C // heads
    h = F.layer_norm(x, (C,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-6)  # eps :175
    qkv = F.linear(h, sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"])
    qkv = qkv.reshape(B, N, 3, heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    attn = ((q @ k.transpose(-2, -1)) * hd ** -0.5).softmax(dim=-1)
    h = (attn @ v).transpose(1, 2).reshape(B, N, C)
    x = x + F.linear(h, sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"])
    h = F.layer_norm(x, (C,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-6)
    h = F.gelu(F.linear(h, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"]))  # exact erf
    return x + F.linear(h, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])


def interpolated_pos_embed(pos, max_grid, size, patch):
    """ViTEncoder.interpolating_pos_embedding, seq_modeling/vit_encoder.py:58-95: the learned [1, 1 + GH*GW, D] table resized
    to the patch grid of a padded feature map of `size` (F.interpolate bicubic, align_corners False, scale factors
    (gh + 0.1) / GH and (gw + 0.1) / GW); the table itself when the token counts agree and the feature map is square (:66-67)."""
    GH, GW = max_grid
    gh, gw = size["height"] // patch[0], size["width"] // patch[1]
    if gh * gw == pos.shape[1] - 1 and size["height"] == size["width"]:
        return pos
    dim = pos.shape[-1]
    h0, w0 = gh + 0.1, gw + 0.1
    grid = F.interpolate(pos[:, 1:].reshape(1, GH, GW, dim).permute(0, 3, 1, 2), scale_factor=(h0 / GH, w0 / GW),
                         mode="bicubic", align_corners=False)
    assert int(h0) == grid.shape[-2] and int(w0) == grid.shape[-1]
    return torch.cat((pos[:, 0].unsqueeze(0), grid.permute(0, 2, 3, 1).reshape(1, -1, dim)), dim=1)


def vit_max_grid(max_dimension, patch):
    """HybridEmbed.__init__ (patchembed.py:74-113): patch grid of the max_dimension crop (the reference measures the backbone's
    output size with a dry run; resnet_out_hw is the same arithmetic)."""
    fh, fw = resnet_out_hw(*max_dimension)
    return -(-fh // patch[0]), -(-fw // patch[1])


def vit_encoder_v3(img, sd, p, depth, heads, patch=(2, 2), faithful=True, taps=None, bn_train=None, drop=None, pos_mode="prefix",
                   max_grid=None):
    """ViTEncoderV3.forward / ViTEncoderV2.forward (seq_modeling/vit_encoder.py:249-268, :208-226; pos_mode "prefix") and
    ViTEncoder.forward_features (:97-118; pos_mode "interp")."""
    x, pad_info, size = hybrid_embed(img, sd, p + "patch_embed.", patch, faithful, bn_train, drop)
    if taps is not None:
        taps["patch"] = x
    B, n, C = x.shape
    x = torch.cat((sd[p + "cls_token"].expand(B, -1, -1), x), dim=1)
    if pos_mode == "interp":
        GH, GW = max_grid
        if size["height"] != GH * patch[0] or size["width"] != GW * patch[1]:  # HybridEmbed's flag, patchembed.py:140
            x = x + interpolated_pos_embed(sd[p + "pos_embed"], max_grid, size, patch)  # :108-109
        else:
            x = x + sd[p + "pos_embed"]  # :110-111
    else:
        x = x + sd[p + "pos_embed"][:, : n + 1]  # flat prefix slice, :260 / :219
    for i in range(depth):
        x = _vit_block(x, sd, f"{p}blocks.{i}.", heads)
        if taps is not None:
            taps[f"block{i}"] = x
    x = F.layer_norm(x, (C,), sd[p + "norm.weight"], sd[p + "norm.bias"], 1e-6)
    return x, pad_info, size


# ---------------------------------------------------------------------------
# Transformer decoder (nn.TransformerDecoder, post-norm, ReLU, no final norm)
# prediction_head/tfm.py:12-32
# ---------------------------------------------------------------------------
def _split_heads(x, heads):  # [B,L,d] -> [B,H,L,hd]
    B, L, d = x.shape
    return x.view(B, L, heads, d // heads).transpose(1, 2)


def _attend(q, k, v, mask=None, drop=None):
    """q [B,H,Lq,hd], k/v [B,H,Lk,hd], additive mask broadcastable to [B,H,Lq,Lk].  drop: see _decoder_layer."""
    s = (q @ k.transpose(-2, -1)) * (q.shape[-1] ** -0.5)
    if mask is not None:
        s = s + mask
    pr = s.softmax(dim=-1)
    if drop is not None:  # nn.MultiheadAttention(dropout=p): on the attention probabilities
        pr = pr * drop(pr.shape, "attn")
    o = pr @ v
    B, H, L, hd = o.shape
    return o.transpose(1, 2).reshape(B, L, H * hd)


def _ln(x, sd, p, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], eps)


def cross_kv(mem, sd, p, heads):
    """K,V of the memory for one layer's multihead_attn (in_proj rows d:2d, 2d:3d)."""
    d = mem.shape[-1]
    w, b = sd[p + "multihead_attn.in_proj_weight"], sd[p + "multihead_attn.in_proj_bias"]
    k = F.linear(mem, w[d:2 * d], b[d:2 * d])
    v = F.linear(mem, w[2 * d:], b[2 * d:])
    return _split_heads(k, heads), _split_heads(v, heads)


def _decoder_layer(x, sd, p, heads, self_k, self_v, ck, cv, mask, drop=None):
    """One nn.TransformerDecoderLayer (norm_first=False) on queries x [B,Lq,d].

    self_k/self_v [B,H,Lk,hd] are the self-attention keys/values (already
    including x's own positions); ck/cv the cross-attention K,V of the memory.

    drop (training with dropout > 0): callable(shape, kind) -> keep mask already scaled by 1/(1-p), asked for in
    the order torch draws them: self-attention probabilities ("attn"), dropout1 ("hidden"), cross-attention
    probabilities, dropout2, the feed-forward block's inner dropout, dropout3 (nn.TransformerDecoderLayer
    _sa_block / _mha_block / _ff_block)."""
    d = x.shape[-1]
    dr = (lambda t: t * drop(t.shape, "hidden")) if drop is not None else (lambda t: t)
    w, b = sd[p + "self_attn.in_proj_weight"], sd[p + "self_attn.in_proj_bias"]
    q = _split_heads(F.linear(x, w[:d], b[:d]), heads)
    a = _attend(q, self_k, self_v, mask, drop)
    x = _ln(x + dr(F.linear(a, sd[p + "self_attn.out_proj.weight"], sd[p + "self_attn.out_proj.bias"])), sd, p + "norm1")
    w, b = sd[p + "multihead_attn.in_proj_weight"], sd[p + "multihead_attn.in_proj_bias"]
    q = _split_heads(F.linear(x, w[:d], b[:d]), heads)
    a = _attend(q, ck, cv, None, drop)
    x = _ln(x + dr(F.linear(a, sd[p + "multihead_attn.out_proj.weight"], sd[p + "multihead_attn.out_proj.bias"])), sd, p + "norm2")
    h = dr(F.relu(F.linear(x, sd[p + "linear1.weight"], sd[p + "linear1.bias"])))
    return _ln(x + dr(F.linear(h, sd[p + "linear2.weight"], sd[p + "linear2.bias"])), sd, p + "norm3")


def _self_kv(x, sd, p, heads):
    d = x.shape[-1]
    w, b = sd[p + "self_attn.in_proj_weight"], sd[p + "self_attn.in_proj_bias"]
    return (_split_heads(F.linear(x, w[d:2 * d], b[d:2 * d]), heads),
            _split_heads(F.linear(x, w[2 * d:], b[2 * d:]), heads))


def _causal_mask(L):
    # _build_attention_mask, tfm.py:74-84: 0 on/below the diagonal, -inf above
    return torch.full((L, L), float("-inf")).triu(1)


def _embed(tgt, sd, p, pos0=0):
    # _embedd_tgt tfm.py:86-94: Embedding * sqrt(d) + WordPosEnc (position_encoding.py:24-28)
    d = sd[p + "word_embed.weight"].shape[1]
    e = F.embedding(tgt, sd[p + "word_embed.weight"]) * math.sqrt(d)
    return e + sd[p + "pos_enc.pe"][pos0:pos0 + tgt.shape[1]][None]


def tfm_full_pass(tgt, mem, sd, p, layers, heads, key_padding=False, drop=None):
    """Decoder over a whole token prefix tgt [B,L] (what the reference runs at
    every greedy step, and the teacher-forced training pass tfm.py:103-118)."""
    B, L = tgt.shape
    x = _embed(tgt, sd, p)
    mask = _causal_mask(L)[None, None]
    if key_padding:  # tgt_key_padding_mask = (tgt == PAD), training only (tfm.py:88-91)
        mask = mask + torch.zeros(B, 1, 1, L).masked_fill((tgt == PAD)[:, None, None, :], float("-inf"))
    for i in range(layers):
        lp = f"{p}model.layers.{i}."
        sk, sv = _self_kv(x, sd, lp, heads)
        ck, cv = cross_kv(mem, sd, lp, heads)
        x = _decoder_layer(x, sd, lp, heads, sk, sv, ck, cv, mask, drop)
    return F.linear(x, sd[p + "proj.weight"], sd[p + "proj.bias"])


def tfm_greedy(mem, sd, p, layers, heads, max_seq_len, is_test=False, faithful=False, tgt=None):
    """TransformerPrediction.forward_greedy eval branch, tfm.py:119-143.

    Returns (preds_index [B,S], logits [B,S,V]).  faithful=True re-decodes the
    whole prefix each step and returns the final pass's logits, as the
    reference does; faithful=False keeps a KV cache and records each step's
    last-position logits (equal up to fp32 accumulation noise)."""
    B = mem.shape[0]
    if tgt is None:
        tgt = torch.full((B, 1), GO, dtype=torch.long)
    end = torch.zeros(B, dtype=torch.bool)
    if faithful:
        out = None
        for _ in range(max_seq_len + 1):
            out = tfm_full_pass(tgt, mem, sd, p, layers, heads)
            nxt = out[:, -1:].softmax(-1).argmax(-1)
            tgt = torch.cat([tgt, nxt], dim=-1)
            end |= nxt[:, 0] == END
            if end.all() and is_test:
                break
        return out.argmax(dim=2), out
    ckv = [cross_kv(mem, sd, f"{p}model.layers.{i}.", heads) for i in range(layers)]
    sk = [None] * layers
    sv = [None] * layers
    logits = []
    tok = tgt[:, -1:]
    for step in range(max_seq_len + 1):
        x = _embed(tok, sd, p, pos0=step)
        for i in range(layers):
            lp = f"{p}model.layers.{i}."
            k, v = _self_kv(x, sd, lp, heads)
            sk[i] = k if sk[i] is None else torch.cat([sk[i], k], dim=2)
            sv[i] = v if sv[i] is None else torch.cat([sv[i], v], dim=2)
            x = _decoder_layer(x, sd, lp, heads, sk[i], sv[i], ckv[i][0], ckv[i][1], None)
        lg = F.linear(x, sd[p + "proj.weight"], sd[p + "proj.bias"])
        logits.append(lg)
        tok = lg.argmax(-1)
        end |= tok[:, 0] == END
        if end.all() and is_test:
            break
    out = torch.cat(logits, dim=1)
    return out.argmax(dim=2), out


def tfm_beam(mem, sd, p, layers, heads, max_seq_len, beam_size):
    """TransformerPrediction.forward_beam (tfm.py:145-186) + Beam (tools/beam.py:38-140)
    for ONE sample (mem [1,T,d]) with a fresh Beam (demo reset_beam semantics).

    Full-length re-decode of every live hypothesis, log_softmax at `step`,
    flat top-k over (hyp x V) with k = beam - completed.  Returns
    (LongTensor [1,len], score float)."""
    assert mem.shape[0] == 1
    hyps = torch.full((1, max_seq_len + 2), PAD, dtype=torch.long)
    hyps[:, 0] = GO
    scores = torch.zeros(1)
    completed = []  # (seq list, score)
    for step in range(max_seq_len + 1):
        n = hyps.shape[0]
        out = tfm_full_pass(hyps, mem.expand(n, -1, -1), sd, p, layers, heads)
        logp = F.log_softmax(out[:, step, :], dim=-1)
        V = logp.shape[1]
        live = beam_size - len(completed)
        top_s, top_i = torch.topk((scores[:, None] + logp).reshape(-1), k=live)
        prev, word = top_i // V, top_i % V
        new_h, new_s = [], []
        for pi, wi, sc in zip(prev.tolist(), word.tolist(), top_s.tolist()):
            hyps[pi, step + 1] = wi  # in-place write before clone, beam.py:90
            if wi == END:
                completed.append((hyps[pi, 1:step + 2].tolist(), sc))
            else:
                new_h.append(hyps[pi].clone())
                new_s.append(sc)
        if len(completed) == beam_size:
            break
        hyps = torch.stack(new_h, dim=0)
        scores = torch.tensor(new_s, dtype=torch.float)
    if not completed:  # set_hypothesis, beam.py:132-140
        completed.append((hyps[0, 1:].tolist(), float(scores[0])))
    best = max(completed, key=lambda h: h[1] / max(len(h[0]), 1))
    return torch.LongTensor(best[0]).unsqueeze(0), best[1]


# ---------------------------------------------------------------------------
# VGG extractor, BiLSTM, LSTM-attention decoder (config C0 and the shipped ViT+Attnv2 configs)
# ---------------------------------------------------------------------------
ATTN_GO, ATTN_END = 0, 1  # modules/converter/attn_converter.py:8,19-29


def vgg(x, sd, p, faithful=True, bn_train=None):
    """VGG_FeatureExtractor.forward, feature_extractor/vgg.py:16-44 (p = '...ConvNet.'); bn_train: module.train() batch
    statistics for its two BatchNorm layers (as _bn)."""
    c = lambda x, i, pad=1: F.conv2d(x, sd[f"{p}{i}.weight"], sd.get(f"{p}{i}.bias"), 1, pad)
    x = F.max_pool2d(F.relu(c(x, 0)), 2, 2)
    x = F.max_pool2d(F.relu(c(x, 3)), 2, 2)
    x = F.relu(c(x, 6))
    x = F.max_pool2d(F.relu(c(x, 8)), (2, 1), (2, 1))
    for conv, bn in ((11, 12), (14, 15)):
        if faithful or bn_train is not None:
            x = F.relu(_bn(c(x, conv), sd, f"{p}{bn}", bn_train=bn_train))
        else:
            wf, bf = fold_bn(sd[f"{p}{conv}.weight"], sd, f"{p}{bn}")
            x = F.relu(F.conv2d(x, wf, bf, 1, 1))
        if conv == 14:
            x = F.max_pool2d(x, (2, 1), (2, 1))
    return F.relu(c(x, 18, 0))


def bilstm(x, sd, p):
    """BidirectionalLSTM.forward, seq_modeling/bilstm.py:14-24: nn.LSTM(bidirectional) + Linear.
    x [B,T,in]; gate order i,f,g,o."""
    B, T, _ = x.shape
    outs = []
    for sfx, order in (("", range(T)), ("_reverse", range(T - 1, -1, -1))):
        wi, wh = sd[f"{p}rnn.weight_ih_l0{sfx}"], sd[f"{p}rnn.weight_hh_l0{sfx}"]
        bi, bh = sd[f"{p}rnn.bias_ih_l0{sfx}"], sd[f"{p}rnn.bias_hh_l0{sfx}"]
        H = wh.shape[1]
        h = torch.zeros(B, H, dtype=x.dtype)
        c = torch.zeros(B, H, dtype=x.dtype)
        out = [None] * T
        for t in order:
            g = F.linear(x[:, t], wi, bi) + F.linear(h, wh, bh)
            i, f, gg, o = g.chunk(4, dim=1)
            c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
            h = torch.sigmoid(o) * torch.tanh(c)
            out[t] = h
        outs.append(torch.stack(out, dim=1))
    return F.linear(torch.cat(outs, dim=2), sd[p + "linear.weight"], sd[p + "linear.bias"])


class LuongResetMemError(AttributeError):
    """attn_type 'luong': Attention.forward_greedy / forward_beam call `self.attention_cell.reset_mem()` unconditionally
    (seq2seq.py:114,285; seq2seq_v2.py:65,247) and LuongAttention (addon_module/attention1D.py:8-35) defines no such
    method -- the reference raises AttributeError before the first step.  The restatement raises the same type."""


def _attn_keys_proj(keys, sd, a, attn_type):
    """Projection of the encoder outputs inside the score: LocationAwareAttentionCell.key_proj (attention1D.py:132,145) or
    BahdanauAttentionCell.i2h (bias-free, :75,80)."""
    if attn_type in ("coverage", "loc_aware"):
        return F.linear(keys, sd[a + "attn.key_proj.weight"], sd[a + "attn.key_proj.bias"])
    return F.linear(keys, sd[a + "attn.i2h.weight"])


def _attn_alpha(kp, h, mem, sd, a, attn_type):
    """Alignment of one step, [M,T,1].  Location-aware cell (attention1D.py:121-161,216-226): tanh(key_proj + query_proj +
    loc_proj(loc_conv(last alignment))) -> score (with bias); Bahdanau cell (:71-85,104-106): tanh(i2h + h2h) -> score
    (no bias), no memory."""
    if attn_type in ("coverage", "loc_aware"):
        hq = F.linear(h, sd[a + "attn.query_proj.weight"], sd[a + "attn.query_proj.bias"]).unsqueeze(1)
        pad = (sd[a + "attn.loc_conv.weight"].shape[2] - 1) // 2
        last = torch.zeros(h.shape[0], kp.shape[-2], 1, dtype=kp.dtype) if mem is None else mem
        loc = F.conv1d(last.permute(0, 2, 1), sd[a + "attn.loc_conv.weight"], sd[a + "attn.loc_conv.bias"], padding=pad)
        loc = F.linear(loc.transpose(1, 2), sd[a + "attn.loc_proj.weight"], sd[a + "attn.loc_proj.bias"])
        e = F.linear(torch.tanh(kp + hq + loc), sd[a + "attn.score.weight"], sd[a + "attn.score.bias"])
    else:
        hq = F.linear(h, sd[a + "attn.h2h.weight"], sd[a + "attn.h2h.bias"]).unsqueeze(1)
        e = F.linear(torch.tanh(kp + hq), sd[a + "attn.score.weight"])
    return F.softmax(e, dim=1)


def _attn_embed(targets, sd, p, V, embed_target, dtype):
    """Decoder input of a step: nn.Embedding(padding_idx=[GO]) (seq2seq.py:33-35,69-70) or the one-hot vector of
    _char_to_onehot (:72-78) when embed_target is False (the constructor's default)."""
    if embed_target:
        return F.embedding(targets, sd[p + "embedding.weight"], padding_idx=ATTN_GO)
    return F.one_hot(targets, V).to(dtype)


def attn_greedy(batch_H, sd, p, num_steps, seqmodel, attn_type="coverage", enc_init=True, is_test=False, teacher=None,
                flags=None, drop=None, embed_target=True):
    """Attention.forward_greedy (prediction_head/seq2seq.py:224-331) / AttentionV2.forward_greedy
    (seq2seq_v2.py:176-293) in eval mode (is_train=False), on the LocationAwareAttention cell
    (addon_module/attention1D.py:121-161,203-242; attn_type 'coverage' / 'loc_aware') or the BahdanauAttention cell
    (:71-118; any other attn_type except 'luong', which raises), with embedded or one-hot targets.

    seqmodel: 'BiLSTM' (keys = all tokens, init from their mean), 'TFM' (AttentionV2: keys without the
    cls token, init from the cls token), 'first' (Attention v1 on a non-BiLSTM encoder: keys = all
    tokens, init from token 0)."""
    if attn_type == "luong":
        raise LuongResetMemError("'LuongAttention' object has no attribute 'reset_mem'")
    B = batch_H.shape[0]
    a = p + "attention_cell."
    keys = batch_H[:, 1:] if seqmodel == "TFM" else batch_H
    init = batch_H.mean(dim=1) if seqmodel == "BiLSTM" else batch_H[:, 0]
    Hd = sd[a + "rnn.weight_hh"].shape[1]
    if enc_init:
        h = F.linear(init, sd[p + "proj_init_h.weight"], sd[p + "proj_init_h.bias"])
        c = F.linear(init, sd[p + "proj_init_c.weight"], sd[p + "proj_init_c.bias"])
    else:
        h, c = torch.zeros(B, Hd, dtype=keys.dtype), torch.zeros(B, Hd, dtype=keys.dtype)
    V = sd[a + "generator.weight"].shape[0]
    T = keys.shape[1]
    probs = torch.zeros(B, num_steps, V, dtype=keys.dtype)
    targets = torch.zeros(B, dtype=torch.long)  # [GO]
    mem = None  # attention memory: None on the first step (-> zeros, attention1D.py:147-148)
    alpha_cum = torch.zeros(B, T, 1, dtype=keys.dtype)
    end = torch.zeros(B, dtype=torch.bool)
    kp = _attn_keys_proj(keys, sd, a, attn_type)
    for i in range(num_steps):
        emb = _attn_embed(targets, sd, p, V, embed_target, keys.dtype)
        alpha = _attn_alpha(kp, h, mem, sd, a, attn_type)
        context = torch.bmm(alpha.permute(0, 2, 1), keys).squeeze(1)
        g = (F.linear(torch.cat([context, emb], 1), sd[a + "rnn.weight_ih"], sd[a + "rnn.bias_ih"])
             + F.linear(h, sd[a + "rnn.weight_hh"], sd[a + "rnn.bias_hh"]))
        gi, gf, gg, go = g.chunk(4, dim=1)
        c = torch.sigmoid(gf) * c + torch.sigmoid(gi) * torch.tanh(gg)
        h = torch.sigmoid(go) * torch.tanh(c)
        out = F.linear(h, sd[a + "generator.weight"], sd[a + "generator.bias"])
        if attn_type == "coverage":
            alpha_cum = alpha_cum + alpha
            mem = alpha_cum
        elif attn_type == "loc_aware":
            mem = alpha
        if teacher is not None:
            # is_train (seq2seq.py:298-316): dropout on the generator output, then the next input is the label
            # text[:, i + 1] unless scheduled sampling (teacher_forcing < random.random(), flags[i + 1] == 0 here)
            # picks the arg-max of the (dropped) output
            if drop is not None:
                out = out * drop(out.shape, "hidden")
            probs[:, i] = out
            if i == num_steps - 1:
                break
            targets = teacher[:, i + 1] if (flags is None or flags[i + 1]) else out.argmax(1)
            continue
        probs[:, i] = out
        if i == num_steps - 1:
            break
        targets = out.argmax(1)
        if is_test:
            end |= targets == ATTN_END
            if end.all():
                break
    return probs.argmax(2), probs


def attn_beam(batch_H, sd, p, num_steps, seqmodel, beam_size, enc_init=True, attn_type="coverage", embed_target=True):
    """Attention.forward_beam (prediction_head/seq2seq.py:83-222) / AttentionV2.forward_beam (seq2seq_v2.py:12-174)
    for one sample; coverage attention or the memory-free Bahdanau cell, embedded or one-hot targets ('loc_aware' hands
    the un-reordered alignment of the previous beam to the next step, :207, a shape error as soon as the beam shrinks:
    not restated).  Quirks kept: all `beam_size` rows start identical and
    step 0 takes the top-k of row 0 only (:145-146); the hidden state is re-ordered by prev_word_inds[incomplete]
    but the coverage memory only by `incomplete` (:197-207); a hypothesis ends on [s] = 1 and is stored with its
    [GO]; when the LAST executed step completed nothing the first live sequence is returned even if earlier steps
    completed some (:209-216); otherwise the best score/len sequence is returned together with the MAXIMUM raw
    score (:218-224).  Returns (seq LongTensor [1, n], score float)."""
    assert batch_H.shape[0] == 1
    if attn_type == "luong":
        raise LuongResetMemError("'LuongAttention' object has no attribute 'reset_mem'")
    assert attn_type != "loc_aware"
    a = p + "attention_cell."
    H1 = batch_H[0]
    keys1 = H1[1:] if seqmodel == "TFM" else H1
    init = H1.mean(dim=0) if seqmodel == "BiLSTM" else H1[0]
    Hd = sd[a + "rnn.weight_hh"].shape[1]
    k = beam_size
    if enc_init:
        h = F.linear(init, sd[p + "proj_init_h.weight"], sd[p + "proj_init_h.bias"])[None].repeat(k, 1)
        c = F.linear(init, sd[p + "proj_init_c.weight"], sd[p + "proj_init_c.bias"])[None].repeat(k, 1)
    else:
        h, c = torch.zeros(k, Hd), torch.zeros(k, Hd)
    keys = keys1[None].repeat(k, 1, 1)
    T = keys.shape[1]
    kp1 = _attn_keys_proj(keys1, sd, a, attn_type)
    Vv = sd[a + "generator.weight"].shape[0]
    alpha_cum = torch.zeros(k, T, 1)
    mem = None
    seqs = torch.zeros(k, 1, dtype=torch.long)  # [GO] = 0 (attn_converter.py:8)
    targets = seqs[:, 0]
    top_scores = torch.zeros(k, 1)
    complete, complete_scores, complete_inds = [], [], []
    for step in range(num_steps):
        M = h.shape[0]
        emb = _attn_embed(targets, sd, p, Vv, embed_target, keys.dtype)
        alpha = _attn_alpha(kp1[None], h, mem, sd, a, attn_type)
        context = torch.bmm(alpha.permute(0, 2, 1), keys[:M]).squeeze(1)
        g = (F.linear(torch.cat([context, emb], 1), sd[a + "rnn.weight_ih"], sd[a + "rnn.bias_ih"])
             + F.linear(h, sd[a + "rnn.weight_hh"], sd[a + "rnn.bias_hh"]))
        gi, gf, gg, go = g.chunk(4, dim=1)
        c = torch.sigmoid(gf) * c + torch.sigmoid(gi) * torch.tanh(gg)
        h = torch.sigmoid(go) * torch.tanh(c)
        out = F.linear(h, sd[a + "generator.weight"], sd[a + "generator.bias"])
        V = out.shape[1]
        scores = top_scores.expand_as(out) + F.log_softmax(out, dim=-1)
        if step == 0:
            top_v, top_w = scores[0].topk(k, 0, True, True)
        else:
            top_v, top_w = scores.reshape(-1).topk(k, 0, True, True)
        prev = top_w // V
        nxt = top_w % V
        seqs = torch.cat([seqs[prev], nxt.unsqueeze(1)], dim=1)
        incomplete = [i for i, w in enumerate(nxt.tolist()) if w != ATTN_END]
        complete_inds = sorted(set(range(len(nxt))) - set(incomplete))
        if complete_inds:
            complete.extend(seqs[complete_inds].tolist())
            complete_scores.extend(top_v[complete_inds].tolist())
        k -= len(complete_inds)
        if k == 0:
            break
        seqs = seqs[incomplete]
        h, c = h[prev[incomplete]], c[prev[incomplete]]
        top_scores = top_v[incomplete].unsqueeze(1)
        targets = nxt[incomplete]
        if attn_type == "coverage":
            alpha_cum = (alpha_cum + alpha)[incomplete]
            mem = alpha_cum
    if not complete_inds:
        return torch.tensor(seqs[0][1:].tolist(), dtype=torch.long)[None], float(top_scores[0])
    best = max(range(len(complete)), key=lambda i: complete_scores[i] / len(complete[i]))
    return torch.tensor(complete[best][1:], dtype=torch.long)[None], float(max(complete_scores))


# ---------------------------------------------------------------------------
# Model.forward  (modules/build_model.py:36-79)
# ---------------------------------------------------------------------------
def forward_encoder(cfg, sd, image, faithful=True, taps=None, bn_train=None, drop=None):
    """Model.forward_encoder: returns (contextual_feature [B,T,d], output_shape, feat_pad)."""
    seq = cfg["SequenceModeling"]
    if seq["name"] == "ViT":
        sp = seq["params"]
        # create_vit_modeling, vit_encoder.py:292-302
        patch = tuple(sp["patch_size"])
        interp = not sp.get("fix_embed", False) and sp.get("interpolate_embed", True)
        max_dim = (cfg["imgH"], cfg["max_dimension"][1]) if cfg.get("imgH") else cfg["max_dimension"]
        x, pad_info, size = vit_encoder_v3(image, sd, "seqmodeler.SequenceModeling.", sp["depth"],
                                           sp["num_heads"], patch, faithful, taps, bn_train, drop,
                                           pos_mode="interp" if interp else "prefix",
                                           max_grid=vit_max_grid(max_dim, patch) if interp else None)
        shape = (size["height"] // sp["patch_size"][0], size["width"] // sp["patch_size"][1])
        return x, shape, pad_info
    if seq["name"] == "BiLSTM":
        # Feat=VGG|ResNet -> AdaptiveAvgPool2d((None,1)) over the height (build_feat.py:50-55) -> 2x BiLSTM
        if cfg["FeatureExtraction"]["name"] == "ResNet":
            f = resnet(image, sd, "featextractor.FeatureExtraction.ConvNet.", faithful, bn_train, drop)
        else:
            f = vgg(image, sd, "featextractor.FeatureExtraction.ConvNet.", faithful, bn_train)
        if taps is not None:
            taps["backbone"] = f
        x = f.permute(0, 3, 1, 2).mean(dim=3)  # [B,W,C]
        for i in range(2):
            x = bilstm(x, sd, f"seqmodeler.SequenceModeling.{i}.")
        return x, None, None
    # Feat=ResNet, Seq=None, Pred=TFM: PositionalEncoding2D add then B,C,H,W -> B,HW,C
    # (recognizers/build_seq.py:69-76)
    f = resnet(image, sd, "featextractor.FeatureExtraction.ConvNet.", faithful, bn_train, drop)
    if taps is not None:
        taps["backbone"] = f
    f = f + posenc2d_crop(f.shape[1], f.shape[2], f.shape[3])
    return f.flatten(2).transpose(1, 2).contiguous(), None, None


def forward(cfg, sd, image, text, is_train=True, is_test=False, faithful=False, training=False):
    """Model.forward.  `training` mirrors module.training (teacher forcing vs
    autoregressive, tfm.py:103,189); is_train is the (unused by TFM) call flag."""
    pp = cfg["Prediction"]["params"]
    mem, shape, pad = forward_encoder(cfg, sd, image, faithful)
    p = "predicter.Prediction."
    if cfg["Prediction"]["name"] in ("Attn", "Attnv2"):
        sm = pp.get("seqmodel", "ViT")
        if cfg["Prediction"]["name"] == "Attn" and sm != "BiLSTM":
            sm = "first"  # seq2seq.py:229-238: keys = all tokens, init from token 0
        if cfg.get("beam_size", 1) > 1:  # Attention.forward (seq2seq.py:333-347): beam only when not is_train
            seq, score = attn_beam(mem, sd, p, cfg["batch_max_length"] + 1, sm, cfg["beam_size"], pp.get("enc_init", False),
                                   pp.get("attn_type", "coverage"), pp.get("embed_target", False))
            return seq, score, {}
        preds, probs = attn_greedy(mem, sd, p, cfg["batch_max_length"] + 1, sm, pp.get("attn_type", "coverage"),
                                   pp.get("enc_init", False), is_test, embed_target=pp.get("embed_target", False))
        return preds, probs, {}
    if training:
        logits = tfm_full_pass(text, mem, sd, p, pp["num_decoder_layers"], pp["nhead"], key_padding=True)
        return logits.argmax(2), logits, {}
    if cfg.get("beam_size", 1) > 1:
        seq, score = tfm_beam(mem, sd, p, pp["num_decoder_layers"], pp["nhead"], pp["max_seq_len"], cfg["beam_size"])
        return seq, score, {}
    preds, logits = tfm_greedy(mem, sd, p, pp["num_decoder_layers"], pp["nhead"], pp["max_seq_len"],
                               is_test=is_test, faithful=faithful, tgt=text)
    return preds, logits, {}


def ce_loss(logits, target):
    """forward_step + train_one_step loss (engine/training.py:83,126):
    CrossEntropyLoss(ignore_index=PAD, reduction='none') then .mean() over ALL B*L."""
    V = logits.shape[-1]
    return F.cross_entropy(logits.reshape(-1, V), target.reshape(-1), ignore_index=PAD, reduction="none").mean()


# ---------------------------------------------------------------------------
# Training step (engine/training.py:76-164): module.train() forward + CE + backward
# ---------------------------------------------------------------------------
def is_trainable(key, cfg=None):
    """state_dict keys that are nn.Parameters with requires_grad=True in the reference: everything except
    BatchNorm buffers, the sinusoid tables and the frozen sincos pos_embed of ViTEncoderV3 (vit_encoder.py:235-237);
    the pos_embed of ViTEncoder / ViTEncoderV2 (`cfg` without fix_embed) is a trained table (:44-50)."""
    tail = key.rsplit(".", 1)[-1]
    if tail == "pos_embed" and cfg is not None:
        seq = cfg.get("SequenceModeling") or {}
        return seq.get("name") == "ViT" and not seq["params"].get("fix_embed", False)
    return tail not in ("running_mean", "running_var", "num_batches_tracked", "pe", "pos_embed")


def train_forward(cfg, sd, image, text_in, bn_train, drop=None, flags=None):
    """Model.forward under module.train(): BatchNorm on batch statistics (their running updates are left in
    `bn_train`); TFM head: teacher-forced decoder pass with causal + PAD key-padding masks (tfm.py:103-118);
    Attn / Attnv2 heads: the LSTM-attention loop fed with the label tokens (teacher_forcing = 1.0,
    seq2seq.py:311-316; droprate 0).  Returns logits [B,L,V]."""
    pp = cfg["Prediction"]["params"]
    mem, _, _ = forward_encoder(cfg, sd, image, faithful=True, bn_train=bn_train, drop=drop)
    if cfg["Prediction"]["name"] in ("Attn", "Attnv2"):
        sm = pp.get("seqmodel", "ViT")
        if cfg["Prediction"]["name"] == "Attn" and sm != "BiLSTM":
            sm = "first"
        return attn_greedy(mem, sd, "predicter.Prediction.", cfg["batch_max_length"] + 1, sm, pp.get("attn_type", "coverage"),
                           pp.get("enc_init", False), teacher=text_in, flags=flags, drop=drop,
                           embed_target=pp.get("embed_target", False))[1]
    return tfm_full_pass(text_in, mem, sd, "predicter.Prediction.", pp["num_decoder_layers"], pp["nhead"],
                         key_padding=True, drop=drop)


def train_step_grads(cfg, sd, image, text, drop=None, flags=None):
    """forward_step + loss.backward() (engine/training.py:83-88,126,137): text [B,L+1] with [GO] first;
    the model sees text[:, :-1], the target is text[:, 1:].  Returns (loss, logits, {key: grad}, bn_train)."""
    params = {k: (v.detach().clone().requires_grad_(True) if (v.is_floating_point() and is_trainable(k, cfg)) else v)
              for k, v in sd.items()}
    bn_train = {}
    logits = train_forward(cfg, params, image, text[:, :-1], bn_train, drop, flags)
    # CE ignore_index: PAD = 0 for the TFM converter; for the Attn converter index 0 is [GO], also ignored
    # (attn_converter.py:17) -- the same call either way
    loss = ce_loss(logits, text[:, 1:])
    names = [k for k, v in params.items() if v.is_floating_point() and v.requires_grad]
    grads = torch.autograd.grad(loss, [params[k] for k in names], allow_unused=True)
    return loss.detach(), logits.detach(), {k: g for k, g in zip(names, grads)}, bn_train
