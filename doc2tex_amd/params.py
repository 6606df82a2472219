"""Parameter containers with the reference's state_dict key names and shapes.

These nn.Modules only HOLD tensors (so state_dict()/load_state_dict(),
.parameters(), optimizers and the reference's load_checkpoint keep working,
SURVEY.md 8b); they have no forward.  All compute happens in libd2t.  Built
from stock torch.nn classes, initialised the way the reference initialises
(citations relative to /root/reference/doc2tex/modules/component/).
"""
import math

import numpy as np
import torch
import torch.nn as nn

RESNET_LAYERS = (1, 2, 5, 3)  # feature_extractor/resnet.py:262


class _Holder(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("parameter container: compute runs in libd2t, not in this module")


class BasicBlockParams(_Holder):
    """BasicBlock, feature_extractor/resnet.py:10-30."""

    def __init__(self, inplanes, planes, downsample):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, 1, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample


class _ConvMLPParams(_Holder):
    """ConvMLP, addon_module/visual_attention.py:85-102.  Its `hidden_channels = in_channels or hidden_channels`
    (:89) makes the hidden width equal to the input width whatever `rd_channels` says."""

    def __init__(self, channels):
        super().__init__()
        self.fc1 = nn.Conv2d(channels, channels, 1, bias=True)
        self.norm = nn.LayerNorm(channels)  # LayerNorm2d
        self.fc2 = nn.Conv2d(channels, channels, 1, bias=True)


class GlobalContextParams(_Holder):
    """GlobalContext(channel) with its defaults use_attn=True, fuse_add=True, fuse_scale=False
    (visual_attention.py:105-165)."""

    def __init__(self, channel):
        super().__init__()
        self.global_cxt = nn.Conv2d(channel, 1, 1, bias=True)
        self.bottleneck_add = _ConvMLPParams(channel)
        nn.init.kaiming_normal_(self.global_cxt.weight, mode="fan_in", nonlinearity="relu")
        nn.init.zeros_(self.bottleneck_add.fc2.weight)


class ResNetParams(_Holder):
    """ResNet, feature_extractor/resnet.py:51-177."""

    def __init__(self, input_channel, output_channel, with_gcb=False):
        super().__init__()
        self.with_gcb = with_gcb
        oc = [output_channel // 4, output_channel // 2, output_channel, output_channel]
        self.inplanes = output_channel // 8
        self.conv0_1 = nn.Conv2d(input_channel, output_channel // 16, 3, 1, 1, bias=False)
        self.bn0_1 = nn.BatchNorm2d(output_channel // 16)
        self.conv0_2 = nn.Conv2d(output_channel // 16, self.inplanes, 3, 1, 1, bias=False)
        self.bn0_2 = nn.BatchNorm2d(self.inplanes)
        self.layer1 = self._make_layer(oc[0], RESNET_LAYERS[0])
        self.conv1 = nn.Conv2d(oc[0], oc[0], 3, 1, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(oc[0])
        self.layer2 = self._make_layer(oc[1], RESNET_LAYERS[1])
        self.conv2 = nn.Conv2d(oc[1], oc[1], 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(oc[1])
        self.layer3 = self._make_layer(oc[2], RESNET_LAYERS[2])
        self.conv3 = nn.Conv2d(oc[2], oc[2], 3, 1, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(oc[2])
        self.layer4 = self._make_layer(oc[3], RESNET_LAYERS[3])
        self.conv4_1 = nn.Conv2d(oc[3], oc[3], 2, (2, 1), (0, 1), bias=False)
        self.bn4_1 = nn.BatchNorm2d(oc[3])
        self.conv4_2 = nn.Conv2d(oc[3], oc[3], 2, 1, 0, bias=False)
        self.bn4_2 = nn.BatchNorm2d(oc[3])
        for m in self.modules():  # init_weights, resnet.py:164-177
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)

    def _make_layer(self, planes, blocks):
        downsample = None
        if self.inplanes != planes:  # resnet.py:181-192
            downsample = nn.Sequential(nn.Conv2d(self.inplanes, planes, 1, 1, bias=False), nn.BatchNorm2d(planes))
        layers = [BasicBlockParams(self.inplanes, planes, downsample)]
        self.inplanes = planes
        layers += [BasicBlockParams(planes, planes, None) for _ in range(1, blocks)]
        if self.with_gcb:  # resnet.py:200-201
            layers.append(GlobalContextParams(planes))
        return nn.Sequential(*layers)


class ResNetFeatureExtractorParams(_Holder):
    """ResNet_FeatureExtractor, resnet.py:248-271 -> key prefix '<...>.ConvNet.'"""

    def __init__(self, input_channel=3, output_channel=512, gcb=False, pretrained=False, weight_dir=None, debug=False):
        super().__init__()
        if pretrained:
            raise NotImplementedError("load weights through load_state_dict / load_checkpoint")
        self.ConvNet = ResNetParams(input_channel, output_channel, with_gcb=bool(gcb))
        self.in_chans = input_channel


def backbone_out_hw(h, w):
    """Spatial size of the ResNet output (resnet.py:205-245) for an h x w crop."""
    h, w = h // 2, w // 2
    h, w = h // 2, w // 2
    h, w = (h - 2) // 2 + 1, w + 1
    h, w = (h - 2) // 2 + 1, w + 1
    return h - 1, w - 1


def sincos_2d_table(dim, grid_h, grid_w):
    """Frozen 2-D sincos table with a zero cls row (common/mae_posembed.py:20-70):
    w-index first (:28), each half [sin | cos] concatenated, float32 numpy."""
    gh = np.arange(grid_h, dtype=np.float32)
    gw = np.arange(grid_w, dtype=np.float32)
    col, row = np.meshgrid(gw, gh)

    def one(d, pos):
        omega = np.arange(d // 2, dtype=np.float32)
        omega /= d / 2.0
        omega = 1.0 / 10000 ** omega
        out = np.einsum("m,d->md", pos.reshape(-1), omega)
        return np.concatenate([np.sin(out), np.cos(out)], axis=1)

    emb = np.concatenate([one(dim // 2, col), one(dim // 2, row)], axis=1)
    emb = np.concatenate([np.zeros([1, dim]), emb], axis=0)
    return torch.from_numpy(emb).float().unsqueeze(0)


class HybridEmbedParams(_Holder):
    """HybridEmbed, seq_modeling/addon_module/patchembed.py:51-113."""

    def __init__(self, backbone, img_size, patch_size, embed_dim):
        super().__init__()
        self.img_size = tuple(img_size)
        self.patch_size = tuple(patch_size)
        self.backbone = backbone
        fh, fw = backbone_out_hw(*self.img_size)  # the reference measures this with a dry run (:74-85)
        if fh < self.patch_size[0] or fw < self.patch_size[1]:
            raise AssertionError("max_dimension too small for the backbone + patch size")
        self.feature_size = (-(-fh // self.patch_size[0]) * self.patch_size[0],
                             -(-fw // self.patch_size[1]) * self.patch_size[1])
        self.grid_size = (self.feature_size[0] // self.patch_size[0], self.feature_size[1] // self.patch_size[1])
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.proj = nn.Conv2d(backbone.ConvNet.conv4_2.out_channels, embed_dim, self.patch_size, self.patch_size)


class _VitAttnParams(_Holder):
    def __init__(self, dim):
        super().__init__()
        self.qkv = nn.Linear(dim, dim * 3, bias=True)  # qkv_bias=True, vision_transformer.py:142
        self.proj = nn.Linear(dim, dim)


class _VitMlpParams(_Holder):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)


class _VitBlockParams(_Holder):
    def __init__(self, dim, mlp_ratio=4.0):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = _VitAttnParams(dim)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = _VitMlpParams(dim, int(dim * mlp_ratio))


class ViTEncoderParams(_Holder):
    """ViTEncoder / ViTEncoderV2 / ViTEncoderV3 (seq_modeling/vit_encoder.py:22-118, :207-226, :229-268) over
    VisionTransformer (seq_modeling/vit/vision_transformer.py:132-228): the same parameter tree; `fix_embed` makes
    pos_embed the frozen sincos table of V3, otherwise it is a learned table (trunc-normal, std 0.02, :44-50)."""

    def __init__(self, img_size, patch_size, in_chans, depth, embed_dim, num_heads, hybrid_backbone, fix_embed=True):
        super().__init__()
        self.num_features = self.embed_dim = embed_dim
        self.num_heads = num_heads
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.patch_embed = HybridEmbedParams(hybrid_backbone, img_size, patch_size, embed_dim)
        gh, gw = self.patch_embed.grid_size
        if fix_embed:
            self.pos_embed = nn.Parameter(sincos_2d_table(embed_dim, gh, gw), requires_grad=False)
        else:
            self.pos_embed = nn.Parameter(torch.zeros(1, gh * gw + 1, embed_dim))
            nn.init.trunc_normal_(self.pos_embed, std=0.02)
        self.blocks = nn.ModuleList([_VitBlockParams(embed_dim) for _ in range(depth)])
        self.norm = nn.LayerNorm(embed_dim, eps=1e-6)
        self.emb_height, self.emb_width = gh, gw
        nn.init.trunc_normal_(self.cls_token, std=0.02)
        for m in self.modules():  # _init_weights, vision_transformer.py:230-237
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=0.02)
                nn.init.zeros_(m.bias)
            elif isinstance(m, nn.LayerNorm):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)


class WordPosEncParams(_Holder):
    """WordPosEnc, prediction_head/addon_module/position_encoding.py:7-22."""

    def __init__(self, d_model=512, max_len=500, temperature=10000.0):
        super().__init__()
        pe = torch.zeros(max_len, d_model)
        position = torch.arange(0, max_len, dtype=torch.float)
        dim_t = torch.arange(0, d_model, 2, dtype=torch.float)
        div_term = 1.0 / (temperature ** (dim_t / d_model))
        ang = position[:, None] * div_term[None, :]
        pe[:, 0::2] = ang.sin()
        pe[:, 1::2] = ang.cos()
        self.register_buffer("pe", pe)


class TransformerPredictionParams(_Holder):
    """TransformerPrediction.__init__, prediction_head/tfm.py:36-72."""

    def __init__(self, d_model, nhead, num_decoder_layers, dim_feedforward, dropout, num_classes, max_seq_len,
                 padding_idx, device="cuda"):
        super().__init__()
        self.max_seq_len = max_seq_len
        self.padding_idx = padding_idx
        self.num_classes = num_classes
        self.device = device
        self.d_model = d_model
        self.nhead = nhead
        self.num_decoder_layers = num_decoder_layers
        self.dim_feedforward = dim_feedforward
        self.dropout = dropout
        self.word_embed = nn.Embedding(num_classes, d_model, padding_idx=padding_idx)
        self.pos_enc = WordPosEncParams(d_model=d_model)
        layer = nn.TransformerDecoderLayer(d_model=d_model, nhead=nhead, dim_feedforward=dim_feedforward,
                                           dropout=dropout)
        self.model = nn.TransformerDecoder(layer, num_decoder_layers)
        for p in self.model.parameters():  # tfm.py:28-30
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        self.proj = nn.Linear(d_model, num_classes)


class PositionalEncoding2DParams(_Holder):
    """PositionalEncoding2D (common/postional_encoding.py:91-134).

    The reference registers a (d_model, 2000, 2000) fp32 buffer (8 GB at d=512).
    Here the same state_dict key exists as a zero-stride expanded view (no
    storage); the engine evaluates the closed form for the crop it needs.
    load_state_dict ignores an incoming 'pe' (it is a constant table)."""

    def __init__(self, d_model, max_h=2000, max_w=2000):
        super().__init__()
        self.d_model = d_model
        self.register_buffer("pe", torch.zeros(1, 1, 1).expand(d_model, max_h, max_w), persistent=True)

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        state_dict.pop(prefix + "pe", None)
        # the key is still expected: report nothing missing for it
        missing_keys = args[2] if len(args) > 2 else kwargs.get("missing_keys")
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)
        if missing_keys is not None and (prefix + "pe") in missing_keys:
            missing_keys.remove(prefix + "pe")


class VGGFeatureExtractorParams(_Holder):
    """VGG_FeatureExtractor, feature_extractor/vgg.py:5-41: same nn.Sequential indices -> same keys."""

    def __init__(self, input_channel, output_channel=512):
        super().__init__()
        oc = [output_channel // 8, output_channel // 4, output_channel // 2, output_channel]
        self.output_channel = oc
        self.ConvNet = nn.Sequential(
            nn.Conv2d(input_channel, oc[0], 3, 1, 1), nn.ReLU(True), nn.MaxPool2d(2, 2),
            nn.Conv2d(oc[0], oc[1], 3, 1, 1), nn.ReLU(True), nn.MaxPool2d(2, 2),
            nn.Conv2d(oc[1], oc[2], 3, 1, 1), nn.ReLU(True),
            nn.Conv2d(oc[2], oc[2], 3, 1, 1), nn.ReLU(True), nn.MaxPool2d((2, 1), (2, 1)),
            nn.Conv2d(oc[2], oc[3], 3, 1, 1, bias=False), nn.BatchNorm2d(oc[3]), nn.ReLU(True),
            nn.Conv2d(oc[3], oc[3], 3, 1, 1, bias=False), nn.BatchNorm2d(oc[3]), nn.ReLU(True),
            nn.MaxPool2d((2, 1), (2, 1)),
            nn.Conv2d(oc[3], oc[3], 2, 1, 0), nn.ReLU(True))


class BidirectionalLSTMParams(_Holder):
    """BidirectionalLSTM, seq_modeling/bilstm.py:6-12."""

    def __init__(self, input_size, hidden_size, output_size):
        super().__init__()
        self.rnn = nn.LSTM(input_size, hidden_size, bidirectional=True, batch_first=True)
        self.linear = nn.Linear(hidden_size * 2, output_size)


class _LocationAwareAttentionCellParams(_Holder):
    """LocationAwareAttentionCell, prediction_head/addon_module/attention1D.py:121-133."""

    def __init__(self, kernel_size, kernel_dim, hidden_dim, input_dim):
        super().__init__()
        self.loc_conv = nn.Conv1d(1, kernel_dim, kernel_size=2 * kernel_size + 1, padding=kernel_size, bias=True)
        self.loc_proj = nn.Linear(kernel_dim, hidden_dim)
        self.query_proj = nn.Linear(hidden_dim, hidden_dim)
        self.key_proj = nn.Linear(input_dim, hidden_dim)
        self.score = nn.Linear(hidden_dim, 1)


class _LocationAwareAttentionParams(_Holder):
    """LocationAwareAttention over BahdanauAttention, attention1D.py:88-97,203-214."""

    def __init__(self, kernel_size, kernel_dim, input_size, hidden_size, num_embeddings, num_classes):
        super().__init__()
        self.attn = _LocationAwareAttentionCellParams(kernel_size, kernel_dim, hidden_size, input_size)
        self.rnn = nn.LSTMCell(input_size + num_embeddings, hidden_size)
        self.generator = nn.Linear(hidden_size, num_classes)


class _BahdanauAttentionCellParams(_Holder):
    """BahdanauAttentionCell, attention1D.py:71-77."""

    def __init__(self, input_dim, hidden_dim):
        super().__init__()
        self.i2h = nn.Linear(input_dim, hidden_dim, bias=False)
        self.h2h = nn.Linear(hidden_dim, hidden_dim)
        self.score = nn.Linear(hidden_dim, 1, bias=False)


class _BahdanauAttentionParams(_Holder):
    """BahdanauAttention, attention1D.py:88-97."""

    def __init__(self, input_size, hidden_size, num_embeddings, num_classes):
        super().__init__()
        self.attn = _BahdanauAttentionCellParams(input_size, hidden_size)
        self.rnn = nn.LSTMCell(input_size + num_embeddings, hidden_size)
        self.generator = nn.Linear(hidden_size, num_classes)


class _LuongAttentionCellParams(_Holder):
    """LuongAttentionCell, attention1D.py:38-50."""

    def __init__(self, hidden_size, method):
        super().__init__()
        if method in ("general", "concat"):
            self.fc = nn.Linear(hidden_size, hidden_size, bias=False)
        if method == "concat":
            self.weight = nn.Parameter(torch.zeros(1, hidden_size))  # the reference leaves it uninitialised (:50)


class _LuongAttentionParams(_Holder):
    """LuongAttention, attention1D.py:8-16.  Holds the reference's parameters so that checkpoints load; the reference
    cannot run this cell (Attention.forward_* call attention_cell.reset_mem(), which it does not define)."""

    def __init__(self, input_size, hidden_size, num_embeddings, num_classes, method="dot"):
        super().__init__()
        self.attn = _LuongAttentionCellParams(hidden_size, method)
        self.rnn = nn.LSTMCell(num_embeddings, hidden_size)
        self.generator = nn.Linear(2 * hidden_size, num_classes)


class AttentionParams(_Holder):
    """Attention.__init__ / AttentionV2, prediction_head/seq2seq.py:11-82: the attention cell is chosen by `attn_type`
    ('luong' -> Luong, 'loc_aware' / 'coverage' -> location-aware, anything else -> Bahdanau, :44-53), the decoder input is
    an nn.Embedding (embed_target) or the one-hot vector of the previous token (:72-78)."""

    def __init__(self, kernel_size, kernel_dim, input_size, hidden_size, num_classes, embed_dim=None,
                 attn_type="coverage", embed_target=False, enc_init=False, teacher_forcing=1.0, droprate=0.1,
                 method="concat", seqmodel="ViT", viz_attn=False, device="cuda"):
        super().__init__()
        if embed_dim is None:
            embed_dim = input_size
        if input_size != 256 or hidden_size != 256 or (embed_target and embed_dim != 256):
            raise NotImplementedError("the Attn kernel is built for input_size = hidden_size = embed_dim = 256")
        if not embed_target and num_classes > 1024:
            raise NotImplementedError("one-hot targets: the Attn kernel handles up to 1024 classes")
        if embed_target:
            self.embedding = nn.Embedding(num_classes, embed_dim, padding_idx=0)  # ATTN.START() = 0
        num_embeddings = embed_dim if embed_target else num_classes
        if attn_type == "luong":
            self.attention_cell = _LuongAttentionParams(input_size, hidden_size, num_embeddings, num_classes, method)
        elif attn_type in ("loc_aware", "coverage"):
            self.attention_cell = _LocationAwareAttentionParams(kernel_size, kernel_dim, input_size, hidden_size,
                                                                num_embeddings, num_classes)
        else:
            self.attention_cell = _BahdanauAttentionParams(input_size, hidden_size, num_embeddings, num_classes)
        self.hidden_size, self.input_size, self.num_classes = hidden_size, input_size, num_classes
        self.kernel_size, self.kernel_dim = kernel_size, kernel_dim
        self.attn_type, self.enc_init, self.seqmodel, self.device = attn_type, enc_init, seqmodel, device
        self.embed_target, self.teacher_forcing = embed_target, teacher_forcing
        if enc_init:  # init_hidden, seq2seq.py:80-82
            self.proj_init_h = nn.Linear(input_size, hidden_size, bias=True)
            self.proj_init_c = nn.Linear(input_size, hidden_size, bias=True)
