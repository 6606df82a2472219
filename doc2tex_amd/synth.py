"""Benchmark configurations C0..C4 and seeded synthetic weights / crops.

No weights, vocabulary or dataset ship with the reference ("Facts"),
so parity and throughput runs use seeded synthetic state_dicts and crops.  The
generator is keyed by the state_dict entry NAME (crc32) so that the reference,
the oracle and the engine regenerate identical tensors without shipping 225 MB
files.  Config dictionaries follow the reference's flat YAML schema
(config/train.yaml:1-49 in the reference).
"""
import copy
import zlib

import numpy as np
import torch

VOCAB = 500  # 496 symbols + [PAD],[GO],[s],[UNK]  (tfm_converter.py:8)
MAX_LEN = 150


def _vit_seq(depth=6, heads=8, hidden=256):
    return {
        "name": "ViT",
        "params": {
            "backbone": {"name": "resnet", "input_channel": 1, "output_channel": 512, "gcb": False},
            "fix_embed": True,
            "input_channel": 1,
            "patching_style": "2d",
            "patch_size": [2, 2],
            "depth": depth,
            "num_heads": heads,
            "hidden_size": hidden,
        },
    }


def _tfm(d_model, layers, ff=1024, heads=8):
    return {
        "name": "TFM",
        "params": {
            "d_model": d_model,
            "nhead": heads,
            "num_decoder_layers": layers,
            "dim_feedforward": ff,
            "dropout": 0.0,
            "max_seq_len": MAX_LEN,
            "padding_idx": 0,
        },
    }


_CONFIGS = {
    # C1: ResNet conv encoder + 2-layer transformer decoder, 64x256 crops, bs=32
    "C1": {
        "FeatureExtraction": {"name": "ResNet", "params": {"input_channel": 1, "output_channel": 512}},
        "SequenceModeling": {"name": "None", "params": {}},
        "Prediction": _tfm(512, 2),
        "max_dimension": [64, 256],
        "_crop": (64, 256),
        "_batch": 32,
    },
    # C2 (headline): HybridViT encoder + 6-layer transformer decoder, 128x512, bs=64
    "C2": {
        "FeatureExtraction": {"name": "None", "params": {}},
        "SequenceModeling": _vit_seq(),
        "Prediction": _tfm(256, 6),
        "max_dimension": [128, 512],
        "_crop": (128, 512),
        "_batch": 64,
    },
    # C4: C2 model sized for 160x640, beam width 5
    "C4": {
        "FeatureExtraction": {"name": "None", "params": {}},
        "SequenceModeling": _vit_seq(),
        "Prediction": _tfm(256, 6),
        "max_dimension": [160, 640],
        "beam_size": 5,
        "_crop": (160, 640),
        "_batch": 128,
    },
    # tiny variants for fast CPU-side oracle tests (same code paths, small shapes)
    "T2": {
        "FeatureExtraction": {"name": "None", "params": {}},
        "SequenceModeling": _vit_seq(depth=2, heads=8, hidden=256),
        "Prediction": _tfm(256, 2),
        "max_dimension": [48, 64],
        "_crop": (48, 64),
        "_batch": 2,
    },
    "T1": {
        "FeatureExtraction": {"name": "ResNet", "params": {"input_channel": 1, "output_channel": 512}},
        "SequenceModeling": {"name": "None", "params": {}},
        "Prediction": _tfm(512, 2),
        "max_dimension": [32, 64],
        "_crop": (32, 64),
        "_batch": 2,
    },
}
def _attn(name, seqmodel):
    return {"name": name, "params": {"input_size": 256, "hidden_size": 256, "kernel_size": 2, "kernel_dim": 128,
                                     "embed_target": True, "enc_init": True, "attn_type": "coverage",
                                     "seqmodel": seqmodel, "droprate": 0.0}}


# C0: CNN feature extractor + recurrent (LSTMCell) attention decoder, the reference's CPU case, bs=4
_CONFIGS["C0"] = {
    "FeatureExtraction": {"name": "VGG", "params": {"input_channel": 1, "output_channel": 512}},
    "SequenceModeling": {"name": "BiLSTM", "params": {"hidden_size": 256}},
    "Prediction": _attn("Attn", "BiLSTM"),
    "max_dimension": [32, 320],
    "_crop": (32, 320),
    "_batch": 4,
}
# S0: what every shipped YAML uses (config/train.yaml:17-49): HybridViT + Attnv2 coverage-LSTM decoder
_CONFIGS["S0"] = {
    "FeatureExtraction": {"name": "None", "params": {}},
    "SequenceModeling": _vit_seq(),
    "Prediction": _attn("Attnv2", "TFM"),
    "max_dimension": [128, 512],
    "_crop": (128, 512),
    "_batch": 4,
}
# GlobalContext blocks on (gcb: True; off in every shipped config): tiny HybridViT and tiny ResNet + TFM
_CONFIGS["T2G"] = copy.deepcopy(_CONFIGS["T2"])
_CONFIGS["T2G"]["SequenceModeling"]["params"]["backbone"]["gcb"] = True
_CONFIGS["T1G"] = copy.deepcopy(_CONFIGS["T1"])
_CONFIGS["T1G"]["FeatureExtraction"]["params"]["gcb"] = True
_CONFIGS["T2D"] = copy.deepcopy(_CONFIGS["T2"])  # training with dropout in the decoder layers
_CONFIGS["T2D"]["Prediction"]["params"]["dropout"] = 0.1
# The ViT encoders beside ViTEncoderV3 (create_vit_modeling, vit_encoder.py:295-302; in no shipped config): a LEARNED position
# table, read through bicubic interpolation (ViTEncoder: T2V1, and T2V1P with a non-square patch) or as a prefix slice
# (ViTEncoderV2: T2V2).  max_dimension [96, 128] -> patch grid 3 x 17.
_CONFIGS["T2V1"] = copy.deepcopy(_CONFIGS["T2"])
_CONFIGS["T2V1"]["SequenceModeling"]["params"]["fix_embed"] = False
_CONFIGS["T2V1"]["max_dimension"] = [96, 128]
_CONFIGS["T2V1"]["_crop"] = (48, 64)
_CONFIGS["T2V2"] = copy.deepcopy(_CONFIGS["T2V1"])
_CONFIGS["T2V2"]["SequenceModeling"]["params"]["interpolate_embed"] = False
_CONFIGS["T2V1P"] = copy.deepcopy(_CONFIGS["T2V1"])  # patch 1 x 2, max grid 1 x 8: an 80 x 12 crop has the same token count
_CONFIGS["T2V1P"]["SequenceModeling"]["params"]["patch_size"] = [1, 2]  # on a square feature map (vit_encoder.py:66-67)
_CONFIGS["T2V1P"]["max_dimension"] = [32, 60]
_CONFIGS["T2V1P"]["_crop"] = (32, 60)
_CONFIGS["TS0"] = copy.deepcopy(_CONFIGS["S0"])  # tiny S0
_CONFIGS["TS0"]["SequenceModeling"] = _vit_seq(depth=2)
_CONFIGS["TS0"]["max_dimension"] = [48, 64]
_CONFIGS["TS0"]["_crop"] = (48, 64)
# the shipped training recipe of the LSTM head (config/train.yaml:38-49): droprate 0.25, plus scheduled sampling
_CONFIGS["TS0D"] = copy.deepcopy(_CONFIGS["TS0"])
_CONFIGS["TS0D"]["Prediction"]["params"]["droprate"] = 0.25
_CONFIGS["TS0D"]["Prediction"]["params"]["teacher_forcing"] = 0.7
# the other attention cells / decoder inputs of Attention.__init__ (seq2seq.py:11-68): Bahdanau cell (any attn_type that is
# not luong / loc_aware / coverage), one-hot targets (embed_target False, the constructor's default), and Luong
_CONFIGS["B0"] = copy.deepcopy(_CONFIGS["C0"])  # VGG + BiLSTM + Attn, Bahdanau cell, embedded targets
_CONFIGS["B0"]["Prediction"]["params"]["attn_type"] = "bahdanau"
_CONFIGS["TB0"] = copy.deepcopy(_CONFIGS["TS0"])  # tiny HybridViT + Attnv2, Bahdanau cell, one-hot targets, zero initial state
_CONFIGS["TB0"]["Prediction"]["params"].update({"attn_type": "bahdanau", "embed_target": False, "enc_init": False})
_CONFIGS["TO0"] = copy.deepcopy(_CONFIGS["TS0"])  # tiny HybridViT + Attnv2, coverage cell, one-hot targets
_CONFIGS["TO0"]["Prediction"]["params"]["embed_target"] = False
_CONFIGS["TL0"] = copy.deepcopy(_CONFIGS["TS0"])  # Luong cell: constructs, every forward raises (attention_cell.reset_mem)
_CONFIGS["TL0"]["Prediction"]["params"].update({"attn_type": "luong", "method": "general"})
_CONFIGS["C3"] = copy.deepcopy(_CONFIGS["C2"])
_CONFIGS["C3"]["_batch"] = 32  # per GPU; 256 global over 8 GPUs


def make_config(name, device="cpu", max_seq_len=None, beam_size=None):
    """Return a fresh (deep-copied) reference-schema config dict.

    The reference's constructors mutate the dict (build_feat.py:16,
    build_pred.py:16-25), so every Model() needs its own copy.
    """
    cfg = copy.deepcopy(_CONFIGS[name])
    cfg.setdefault("beam_size", 1)
    cfg["num_class"] = VOCAB
    cfg["device"] = device
    cfg["imgH"] = None
    cfg["batch_max_length"] = MAX_LEN
    if max_seq_len is not None:
        if cfg["Prediction"]["name"] == "TFM":
            cfg["Prediction"]["params"]["max_seq_len"] = max_seq_len
        cfg["batch_max_length"] = max_seq_len
    if beam_size is not None:
        cfg["beam_size"] = beam_size
    return cfg


def crop_shape(name):
    return _CONFIGS[name]["_crop"]


def batch_size(name):
    return _CONFIGS[name]["_batch"]


# ---------------------------------------------------------------------------
# seeded tensors
# ---------------------------------------------------------------------------
def _rng(seed, name):
    key = (zlib.crc32(name.encode()) << 32) | (seed & 0xFFFFFFFF)
    return np.random.Generator(np.random.Philox(key=key))


def _fans(shape):
    if len(shape) == 4:  # conv OIHW
        rf = shape[2] * shape[3]
        return shape[1] * rf, shape[0] * rf
    return shape[1], shape[0]


# entries that are deterministic tables, not weights (kept as constructed)
_TABLES = ("pos_embed", "pos_enc.pe", "image_positional_encoder.pe")


def learned_pos_embed(cfg):
    """True when the configuration's ViT encoder trains its position table (ViTEncoder / ViTEncoderV2): `pos_embed` is then a
    weight like any other and gets a seeded value; ViTEncoderV3's sincos table is kept as constructed."""
    seq = cfg.get("SequenceModeling") or {}
    return seq.get("name") == "ViT" and not seq["params"].get("fix_embed", False)


def synth_tensor(name, shape, dtype, seed=1234, end_bias=0.0, learned_pos=False):
    """Seeded value for one state_dict entry, or None for constructed tables."""
    if any(name.endswith(t) for t in _TABLES) and not (learned_pos and name.endswith("pos_embed")):
        return None
    shape = tuple(shape)
    g = _rng(seed, name)
    leaf = name.rsplit(".", 1)[-1]
    parent = name.rsplit(".", 2)[-2] if name.count(".") >= 1 else ""
    if leaf == "num_batches_tracked":
        return torch.zeros(shape, dtype=dtype)
    # BatchNorm2d: bnX / bn0_1 / downsample.1 ; LayerNorm: norm / norm1..3
    is_norm = parent.startswith("bn") or parent.startswith("norm") or ".downsample.1." in name
    if leaf == "running_mean":
        v = g.standard_normal(shape) * 0.1
    elif leaf == "running_var":
        v = g.uniform(0.5, 1.5, shape)
    elif is_norm and leaf == "weight":
        v = g.uniform(0.5, 1.5, shape)
    elif is_norm and leaf == "bias":
        v = g.standard_normal(shape) * 0.1
    elif leaf == "cls_token":
        v = g.standard_normal(shape) * 0.02
    elif leaf == "pos_embed":  # a trained table (learned_pos): as large as the patch tokens, so a wrong resize shows
        v = g.standard_normal(shape) * 0.5
    elif "word_embed" in name or name.endswith("Prediction.embedding.weight"):
        v = g.standard_normal(shape)
        v[0] = 0.0  # padding_idx row
    elif len(shape) == 4:  # conv: kaiming-normal fan_out (resnet.py:168), patch proj included
        _, fan_out = _fans(shape)
        v = g.standard_normal(shape) * np.sqrt(2.0 / fan_out)
    elif len(shape) == 2:  # linear / in_proj: xavier-normal
        fi, fo = _fans(shape)
        v = g.standard_normal(shape) * np.sqrt(2.0 / (fi + fo))
    elif len(shape) == 1:  # bias
        v = g.standard_normal(shape) * 0.02
        if end_bias and name.endswith("Prediction.proj.bias"):
            v[2] += end_bias  # raise the [s] logit so greedy rows terminate
        if end_bias and name.endswith("attention_cell.generator.bias"):
            v[1] += end_bias  # [s] = 1 for the Attn converter
    else:
        v = g.standard_normal(shape) * 0.02
    return torch.from_numpy(np.ascontiguousarray(v)).to(dtype)


def synth_state_dict(template, seed=1234, end_bias=0.0, learned_pos=False):
    """Fill a state_dict-shaped mapping {name: tensor} with seeded values.

    `template` is any mapping with the reference's key names and shapes (the
    reference Model's or the engine Model's state_dict()).  Table entries
    (sincos pos_embed, sinusoid pe) are passed through unchanged; `learned_pos` (= learned_pos_embed(cfg)) makes
    pos_embed a seeded weight instead.
    """
    out = {}
    for name, t in template.items():
        v = synth_tensor(name, t.shape, t.dtype, seed=seed, end_bias=end_bias, learned_pos=learned_pos)
        out[name] = t.detach().clone() if v is None else v
    return out


def synth_images(batch, h, w, seed=1000):
    """Seeded crops, uniform in [-1, 1] after the reference normalisation
    x/255 -> (x-0.5)/0.5 (transform/math_transform.py:35-38).  [B,1,H,W] f32."""
    g = _rng(seed, "images")
    u8 = g.integers(0, 256, size=(batch, 1, h, w), dtype=np.int64).astype(np.float32)
    x = (u8 / np.float32(255.0) - np.float32(0.5)) / np.float32(0.5)
    return torch.from_numpy(x.astype(np.float32))


def synth_labels(batch, max_len=MAX_LEN, seed=2000, vocab=VOCAB):
    """Teacher-forcing labels as converter.encode lays them out
    (tfm_converter.py:36-57): [GO] tokens... [s] [PAD]...; shape [B, max_len+2]."""
    g = _rng(seed, "labels")
    text = np.zeros((batch, max_len + 2), dtype=np.int64)
    text[:, 0] = 1
    for b in range(batch):
        n = int(g.integers(20, max_len + 1))
        text[b, 1:1 + n] = g.integers(4, vocab, size=n)
        text[b, 1 + n] = 2
    return torch.from_numpy(text)


def synth_formula_image(h, w, seed=4000, zero_border=False, blank=None):
    """uint8 [h, w] grayscale stand-in for a rendered formula image (what `Image.open(path).convert("L")` holds before
    utils/predict_utils.py:14 resizes it): white page, dark strokes of a few pixels with grey anti-aliased edges.
    `zero_border` blackens the first row (the case in which the reference's paste raises); `blank` = a constant page."""
    rng = np.random.default_rng(seed)
    if blank is not None:
        return np.full((h, w), blank, np.uint8)
    img = np.full((h, w), 255, np.int32)
    for _ in range(max(4, h * w // 600)):
        y, x = int(rng.integers(0, h)), int(rng.integers(0, w))
        dy, dx = (int(rng.integers(1, 4)), int(rng.integers(2, 24))) if rng.random() < 0.5 else \
                 (int(rng.integers(2, 24)), int(rng.integers(1, 4)))
        img[y:y + dy, x:x + dx] = int(rng.integers(0, 60))
    img = img + rng.integers(-12, 1, (h, w))  # page noise keeps the resampler's rounding exercised
    img = np.clip(img, 0, 255).astype(np.uint8)
    if zero_border:
        img[0, :] = 0
    return img
