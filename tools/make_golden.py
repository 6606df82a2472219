#!/usr/bin/env python3
"""Generate tests/golden/* from the REFERENCE and validate oracle/restatement.py.

Authoring-container only: imports /root/reference (never shipped, never read at
test/bench time).  For each case it
  1. builds the reference Model(opt) and loads the seeded synthetic state_dict
     (doc2tex_amd/synth.py),
  2. runs the reference forward on seeded crops,
  3. runs oracle/restatement.py (faithful and KV-cached modes) on the same
     tensors and asserts agreement (tokens exact, floats <= TOL),
  4. writes the reference's outputs as small fixtures (npz + json).

Usage:  PYTHONDONTWRITEBYTECODE=1 python tools/make_golden.py
"""
import contextlib
import io
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True

import numpy as np
import torch

from doc2tex_amd import synth
from oracle import restatement as R

with contextlib.redirect_stdout(io.StringIO()):
    from doc2tex.modules.build_model import Model as RefModel
    from doc2tex.tools.beam import Beam as RefBeam
    from doc2tex.modules.converter.tfm_converter import TFMLabelConverter as RefTFM

GOLD = os.path.join(ROOT, "tests", "golden")
TOL = 2e-5  # restatement vs reference, fp32 CPU both

# name, config, B, H, W, max_seq_len, weight seed, input seed, end_bias, is_test
GREEDY_CASES = [
    ("t2_greedy", "T2", 2, 48, 64, 12, 1234, 1000, 0.0, False),
    ("t2_greedy_early", "T2", 3, 48, 64, 40, 1234, 1001, 1.81, True),
    ("t2_greedy_late", "T2", 3, 48, 64, 40, 1234, 1001, 1.8, True),
    ("t1_greedy", "T1", 2, 32, 64, 12, 1234, 1002, 0.0, False),
    ("c2_greedy", "C2", 2, 128, 512, 150, 1234, 1003, 0.0, False),
    ("c2_small_crop", "C2", 1, 96, 384, 10, 1234, 1004, 0.0, False),
    ("c1_greedy", "C1", 2, 64, 256, 30, 1234, 1005, 0.0, False),
    # recurrent family: BASELINE configs[0] (VGG + BiLSTM + LSTM-attention, bs=4) and the shipped
    # HybridViT + Attnv2 configuration (config/train.yaml:17-49)
    ("c0_greedy", "C0", 4, 32, 320, 150, 1234, 1006, 0.0, False),
    ("c0_greedy_early", "C0", 3, 32, 320, 150, 1234, 1007, 0.0, True),
    ("ts0_greedy", "TS0", 2, 48, 64, 12, 1234, 1008, 0.0, False),
    ("s0_greedy", "S0", 2, 128, 512, 30, 1234, 1009, 0.0, False),
    ("s0_small_crop", "S0", 1, 96, 384, 10, 1234, 1013, 0.0, True),
    # GlobalContext blocks on (gcb: True, addon_module/visual_attention.py:105-165)
    ("t2g_greedy", "T2G", 2, 48, 64, 12, 1234, 1014, 0.0, False),
    ("t1g_greedy", "T1G", 2, 32, 64, 12, 1234, 1015, 0.0, False),
    # round 2 -- BASELINE configs[4] geometry: the C2 model built for max_dimension [160, 640] (9x161 backbone grid, 406 memory
    # tokens), at its largest bucket and at the two smaller buckets of SURVEY 8d (flat-prefix slice of the 160x640 table)
    ("c4_greedy_160", "C4", 2, 160, 640, 150, 1234, 1070, 0.0, False),
    ("c4_greedy_128", "C4", 1, 128, 512, 20, 1234, 1071, 0.0, False),
    ("c4_greedy_96", "C4", 2, 96, 384, 20, 1234, 1072, 0.0, False),
    # BASELINE configs[1] at its full decode length (the GPU test decodes B=32 and checks these first rows)
    ("c1_greedy_full", "C1", 2, 64, 256, 150, 1234, 1073, 0.0, False),
    # round 2 -- the other attention cells / decoder inputs of Attention.__init__ (seq2seq.py:31-53): Bahdanau cell with
    # embedded targets on the VGG + BiLSTM encoder; Bahdanau + one-hot targets + zero initial state and coverage + one-hot
    # targets on the tiny HybridViT + Attnv2 stack
    ("b0_greedy", "B0", 3, 32, 320, 20, 1234, 1080, 0.0, False),
    ("b0_greedy_early", "B0", 3, 32, 320, 40, 1234, 1081, 0.0, True),
    ("tb0_greedy", "TB0", 2, 48, 64, 12, 1234, 1082, 0.0, False),
    ("to0_greedy", "TO0", 2, 48, 64, 12, 1234, 1083, 0.0, False),
]
# round 4 -- the ViT encoders beside ViTEncoderV3 (vit_encoder.py:22-118, :207-226): learned position table read through
# bicubic interpolation (T2V1: max grid 3 x 17; crops whose grid is the table's own, smaller in both directions, smaller in
# one, LARGER than max_dimension; T2V1P: 1 x 2 patches -- a crop with the table's token count on a square feature map takes
# the table as is, :66-67, another is interpolated) or as a prefix slice (T2V2)
VITPOS_GREEDY_CASES = [
    ("v1_greedy_full", "T2V1", 2, 96, 128, 12, 1234, 1090, 0.0, False),
    ("v1_greedy_small", "T2V1", 2, 48, 64, 12, 1234, 1091, 0.0, False),
    ("v1_greedy_mid", "T2V1", 1, 64, 96, 12, 1234, 1092, 0.0, False),
    ("v1_greedy_narrow", "T2V1", 1, 96, 64, 12, 1234, 1093, 0.0, False),
    ("v1_greedy_big", "T2V1", 1, 128, 160, 12, 1234, 1094, 0.0, False),
    ("v1p_greedy_samecount", "T2V1P", 2, 80, 12, 12, 1234, 1095, 0.0, False),
    ("v1p_greedy_interp", "T2V1P", 1, 48, 64, 12, 1234, 1096, 0.0, False),
    ("v2_greedy_small", "T2V2", 2, 48, 64, 12, 1234, 1097, 0.0, False),
    ("v2_greedy_full", "T2V2", 1, 96, 128, 12, 1234, 1098, 0.0, False),
]
GREEDY_CASES += VITPOS_GREEDY_CASES
A15_GREEDY = ("b0_greedy", "b0_greedy_early", "tb0_greedy", "to0_greedy")
ROUND2_GREEDY = ("c4_greedy_160", "c4_greedy_128", "c4_greedy_96", "c1_greedy_full")
BEAM_CASES = [
    ("t2_beam5", "T2", 48, 64, 16, 1234, 1010, 1.8, 5),
    ("c2_beam5", "C2", 96, 384, 12, 1234, 1011, 1.8, 5),
    ("t2_beam3_nofinish", "T2", 48, 64, 6, 1234, 1012, 0.0, 3),
    # round 2 -- config C4 itself (beam_size 5 under max_dimension [160, 640]), every bucket; the last runs all 151 steps
    ("c4_beam5_160", "C4", 160, 640, 24, 1234, 1074, 1.8, 5),
    ("c4_beam5_128", "C4", 128, 512, 24, 1234, 1075, 1.8, 5),
    ("c4_beam5_96", "C4", 96, 384, 24, 1234, 1076, 1.8, 5),
    ("c4_beam5_160_full", "C4", 160, 640, 150, 1234, 1077, 0.0, 5),
]
ROUND2_BEAM = ("c4_beam5_160", "c4_beam5_128", "c4_beam5_96", "c4_beam5_160_full")
# LSTM-attention beam search (seq2seq.py:83-222 / seq2seq_v2.py:12-174): name, config, H, W, batch_max_length, wseed, iseed, end_bias, beam
ATTN_BEAM_CASES = [
    ("ts0_beam5", "TS0", 48, 64, 14, 1234, 1050, 0.3, 5),
    ("c0_beam3", "C0", 32, 320, 12, 1234, 1051, 0.15, 3),
    ("c0_beam3_end", "C0", 32, 320, 12, 1234, 1051, 0.2, 3),  # [s] wins at step 0
    ("s0_beam10", "S0", 96, 384, 10, 1234, 1052, 0.3, 10),
    ("s0_beam10_late", "S0", 96, 384, 10, 1234, 1052, 0.4, 10),  # completions early, none in the last step
    ("ts0_beam4_nofinish", "TS0", 48, 64, 6, 1234, 1053, 0.0, 4),
    # round 2 -- beam search on the Bahdanau cell (no alignment memory) and with one-hot targets
    ("b0_beam3", "B0", 32, 320, 12, 1234, 1084, 0.15, 3),
    ("tb0_beam4", "TB0", 48, 64, 10, 1234, 1085, 0.3, 4),
    ("to0_beam5", "TO0", 48, 64, 12, 1234, 1086, 0.3, 5),
    ("to0_beam5_end", "TO0", 48, 64, 12, 1234, 1086, 0.38, 5),
    ("b0_beam3_end", "B0", 32, 320, 12, 1234, 1084, 0.18, 3),
]
A15_BEAM = ("b0_beam3", "tb0_beam4", "to0_beam5", "to0_beam5_end", "b0_beam3_end")
# configurations whose every forward raises in the reference: name, config, H, W, beam_size
RAISES_CASES = [("tl0_luong_greedy", "TL0", 48, 64, 1), ("tl0_luong_beam", "TL0", 48, 64, 3)]
TRAIN_CASES = [("t2_train", "T2", 2, 48, 64, 20, 1234, 1020)]
# full module.train() steps (BN batch statistics, teacher forcing, CE, backward): name, config, B, H, W, L, wseed, iseed
TRAIN_STEP_CASES = [("t2_train_step", "T2", 3, 48, 64, 24, 1234, 1030), ("t1_train_step", "T1", 2, 32, 64, 22, 1234, 1031),
                    # HybridViT + Attnv2 (the shipped config/train.yaml stack), teacher_forcing 1.0, droprate 0
                    ("ts0_train_step", "TS0", 3, 48, 64, 24, 1234, 1032),
                    # round 2 -- BASELINE configs[3] at its own crop size and label length (four rows of the per-GPU shard:
                    # 8256 pixels per BatchNorm channel in the deepest stage instead of ~100 in the toys above)
                    ("c3_train_step", "C3", 4, 128, 512, 150, 1234, 1033),
                    # GlobalContext blocks on (gcb: True): the two TFM stacks
                    ("t2g_train_step", "T2G", 3, 48, 64, 24, 1234, 1034), ("t1g_train_step", "T1G", 2, 32, 64, 22, 1234, 1035),
                    # BASELINE configs[0]'s stack in training: VGG + 2x BidirectionalLSTM + Attn (coverage cell, init from the mean)
                    ("c0_train_step", "C0", 3, 32, 160, 24, 1234, 1036),
                    # the other attention cells / target encodings of the LSTM heads: Bahdanau cell (VGG + BiLSTM + Attn),
                    # Bahdanau cell with one-hot targets and a zero initial state, coverage cell with one-hot targets
                    ("b0_train_step", "B0", 3, 32, 160, 24, 1234, 1037), ("tb0_train_step", "TB0", 3, 48, 64, 24, 1234, 1038),
                    ("to0_train_step", "TO0", 3, 48, 64, 24, 1234, 1039)]
# round 4 -- training the learned position tables: through the bicubic resize (48x64 crops under a 96x128 table), with the
# table read as it is (crop = max_dimension), and the prefix slice of ViTEncoderV2 (rows past the crop's tokens get zero)
VITPOS_TRAIN_CASES = [("v1_train_step", "T2V1", 3, 48, 64, 24, 1234, 1100), ("v1_train_step_full", "T2V1", 2, 96, 128, 24, 1234, 1101),
                      ("v2_train_step", "T2V2", 3, 48, 64, 24, 1234, 1102)]
TRAIN_STEP_CASES += VITPOS_TRAIN_CASES
LOGIT_STRIDE = {"c3_train_step": 8}  # store every 8th position of the [B, 151, V] logits (fixture size)
GRAD_SAMPLES = 48
GC_MASK_SEED = 99  # seeded keep masks of the GlobalContext blocks' dropout in the *g_train_step fixtures
# dropout placement (p = 0.1 in the decoder layers): name, config, B, H, W, L, wseed, iseed, mask seed
TRAIN_DROPOUT_CASES = [("t2d_train_dropout", "T2D", 3, 48, 64, 24, 1234, 1060, 77),
                       # LSTM head: droprate 0.25 on the generator output + scheduled sampling (teacher_forcing 0.7)
                       ("ts0d_train_dropout", "TS0D", 3, 48, 64, 24, 1234, 1061, 78)]


def build_ref(cfg_name, max_seq_len, beam_size=None, wseed=1234, end_bias=0.0):
    cfg = synth.make_config(cfg_name, max_seq_len=max_seq_len, beam_size=beam_size)
    with contextlib.redirect_stdout(io.StringIO()):
        m = RefModel(cfg)
    tmpl = m.state_dict()
    # the C1 table is (512,2000,2000) fp32 = 8 GB: never copy it
    sd = {}
    learned = synth.learned_pos_embed(cfg)  # ViTEncoder / ViTEncoderV2: pos_embed is a trained table, seeded like a weight
    for k, v in tmpl.items():
        t = synth.synth_tensor(k, v.shape, v.dtype, seed=wseed, end_bias=end_bias, learned_pos=learned)
        sd[k] = v if t is None else t
    m.load_state_dict(sd)
    m.eval()
    return cfg, m, sd


def manifest(sd):
    return {k: list(v.shape) for k, v in sd.items()}


def maxdiff(a, b):
    """max |a-b| relative to max(1, max|b|) (backbone features are O(10-100))."""
    return float((a.double() - b.double()).abs().max() / max(1.0, float(b.double().abs().max())))


def slim_sd(sd):
    """state_dict view for the restatement (drops nothing; the 8 GB C1 table is
    only referenced, the restatement builds its own crop)."""
    return sd


def check_tables(cfg, sd, report):
    p = "seqmodeler.SequenceModeling."
    if p + "pos_embed" in sd and synth.learned_pos_embed(cfg):
        sp = cfg["SequenceModeling"]["params"]
        GH, GW = R.vit_max_grid(cfg["max_dimension"], tuple(sp["patch_size"]))
        assert tuple(sd[p + "pos_embed"].shape) == (1, GH * GW + 1, sp["hidden_size"]), sd[p + "pos_embed"].shape
        report["pos_embed_sum"] = float(sd[p + "pos_embed"].double().sum())
        report["pos_embed_abs"] = float(sd[p + "pos_embed"].double().abs().sum())
        report["max_grid"] = [GH, GW]
    elif p + "pos_embed" in sd:
        gh, gw = R.resnet_out_hw(*cfg["max_dimension"])
        gh, gw = -(-gh // 2), -(-gw // 2)
        t = R.sincos_2d_table(sd[p + "pos_embed"].shape[-1], gh, gw)
        d = maxdiff(t, sd[p + "pos_embed"])
        assert d == 0.0, f"sincos table differs {d}"
        report["pos_embed_sum"] = float(sd[p + "pos_embed"].double().sum())
        report["pos_embed_abs"] = float(sd[p + "pos_embed"].double().abs().sum())
    if "predicter.Prediction.pos_enc.pe" in sd:
        pe = sd["predicter.Prediction.pos_enc.pe"]
        d = maxdiff(R.word_pos_table(pe.shape[1], pe.shape[0]), pe)
        assert d == 0.0, f"word pos table differs {d}"
        report["pe_sum"] = float(pe.double().sum())
    if "seqmodeler.image_positional_encoder.pe" in sd:
        big = sd["seqmodeler.image_positional_encoder.pe"]
        crop = big[:, :9, :80]
        d = maxdiff(R.posenc2d_crop(big.shape[0], 9, 80), crop)
        assert d == 0.0, f"posenc2d differs {d}"
        report["pe2d_crop_sum"] = float(crop.double().sum())


def run_greedy(case):
    name, cname, B, H, W, L, wseed, iseed, end_bias, is_test = case
    t0 = time.time()
    cfg, m, sd = build_ref(cname, L, beam_size=1, wseed=wseed, end_bias=end_bias)  # greedy: beam_size 1 (C4 defaults to 5)
    img = synth.synth_images(B, H, W, seed=iseed)
    text = torch.full((B, 1), R.GO, dtype=torch.long)
    rep = {"case": name, "config": cname, "B": B, "H": H, "W": W, "max_seq_len": L, "wseed": wseed, "beam_size": 1,
           "iseed": iseed, "end_bias": end_bias, "is_test": is_test, "torch": torch.__version__}
    check_tables(cfg, sd, rep)
    with torch.no_grad():
        mem, shape, pad = m.forward_encoder(img)
        preds, logits, _ = m(img, text, is_train=False, is_test=is_test)
        # restatement, both modes
        taps = {}
        mem_f, shape_f, pad_f = R.forward_encoder(cfg, sd, img, faithful=True, taps=taps)
        mem_a, _, _ = R.forward_encoder(cfg, sd, img, faithful=False)
        pf, lf, _ = R.forward(cfg, sd, img, text, is_test=is_test, faithful=True)
        pa, la, _ = R.forward(cfg, sd, img, text, is_test=is_test, faithful=False)
    assert (shape_f, pad_f) == (shape, pad), (shape_f, pad_f, shape, pad)
    rep["diff_mem_faithful"] = maxdiff(mem_f, mem)
    rep["diff_mem_folded"] = maxdiff(mem_a, mem)
    assert rep["diff_mem_faithful"] <= TOL and rep["diff_mem_folded"] <= TOL, rep
    assert pf.shape == preds.shape == pa.shape, (pf.shape, preds.shape, pa.shape)
    assert torch.equal(pf, preds), "faithful restatement tokens differ"
    assert torch.equal(pa, preds), "cached restatement tokens differ"
    rep["diff_logits_faithful"] = maxdiff(lf, logits)
    rep["diff_logits_cached"] = maxdiff(la, logits)
    assert rep["diff_logits_faithful"] <= TOL and rep["diff_logits_cached"] <= 10 * TOL, rep
    top2 = logits.topk(2, dim=-1).values
    rep["min_top2_gap"] = float((top2[..., 0] - top2[..., 1]).min())
    rep["steps"] = int(preds.shape[1])
    rep["output_shape"] = list(shape) if shape is not None else None
    rep["feat_pad"] = list(pad) if pad is not None else None
    rep["mem_shape"] = list(mem.shape)
    rep["mem_sum"] = float(mem.double().sum())
    rep["mem_abs"] = float(mem.double().abs().sum())
    rep["mem_absmax"] = float(mem.abs().max())
    rep["logits_sum"] = float(logits.double().sum())
    rep["logits_absmax"] = float(logits.abs().max())
    rows = sorted(set([0, 1, mem.shape[1] // 2, mem.shape[1] - 1]))
    steps = sorted(set(min(i, preds.shape[1] - 1) for i in [0, 1, preds.shape[1] // 2, preds.shape[1] - 1]))
    arrays = {
        "tokens": preds.numpy().astype(np.int32),
        "mem_rows": np.array(rows, dtype=np.int32),
        "mem_sample": mem[:, rows].numpy(),
        "logit_steps": np.array(steps, dtype=np.int32),
        "logits_sample": logits[:, steps].numpy(),
        "top2_gap": (top2[..., 0] - top2[..., 1]).numpy(),
    }
    if mem.numel() <= 64 * 1024:
        arrays["mem_full"] = mem.numpy()
    if "patch" in taps:  # per-stage checksums for bisecting engine mismatches
        rep["patch_sum"] = float(taps["patch"].double().sum())
        for k in sorted(k for k in taps if k.startswith("block")):
            rep[k + "_sum"] = float(taps[k].double().sum())
    if "backbone" in taps:
        rep["backbone_sum"] = float(taps["backbone"].double().sum())
        rep["backbone_shape"] = list(taps["backbone"].shape)
    rep["seconds"] = round(time.time() - t0, 1)
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **arrays)
    return rep, manifest(sd), cname


def run_beam(case):
    name, cname, H, W, L, wseed, iseed, end_bias, beam = case
    cfg, m, sd = build_ref(cname, L, beam_size=beam, wseed=wseed, end_bias=end_bias)
    img = synth.synth_images(1, H, W, seed=iseed)
    text = torch.full((1, 1), R.GO, dtype=torch.long)
    pred = m.predicter.Prediction
    # fresh Beam per sample (demo reset_beam semantics)
    pred.beam = RefBeam(ignore_w=RefTFM.PAD(), start_w=RefTFM.START(), stop_w=RefTFM.END(),
                        max_len=pred.max_seq_len, device="cpu")
    with torch.no_grad():
        seq, score, _ = m(img, text, is_train=False, is_test=True)
        n_completed = len(pred.beam.completed_hypotheses)
        rseq, rscore, _ = R.forward(cfg, sd, img, text, is_test=True)
    assert torch.equal(seq, rseq), (seq, rseq)
    assert abs(score - rscore) <= 1e-4, (score, rscore)
    rep = {"case": name, "config": cname, "H": H, "W": W, "max_seq_len": L, "wseed": wseed, "iseed": iseed,
           "end_bias": end_bias, "beam_size": beam, "seq": seq[0].tolist(), "score": float(score),
           "completed": n_completed, "torch": torch.__version__}
    return rep


def run_attn_beam(case):
    name, cname, H, W, L, wseed, iseed, end_bias, beam = case
    cfg, m, sd = build_ref(cname, L, beam_size=beam, wseed=wseed, end_bias=end_bias)
    img = synth.synth_images(1, H, W, seed=iseed)
    text = torch.zeros(1, L + 1, dtype=torch.long)  # engine/inferencing.py:58-63 (ignored by the beam path)
    with torch.no_grad():
        seq, score, _ = m(img, text, is_train=False, is_test=True)
        rseq, rscore, _ = R.forward(cfg, slim_sd(sd), img, text, is_train=False, is_test=True)
    assert torch.equal(seq, rseq), (seq, rseq)
    assert abs(float(score) - rscore) <= 1e-4, (float(score), rscore)
    return {"case": name, "config": cname, "H": H, "W": W, "max_seq_len": L, "wseed": wseed, "iseed": iseed,
            "end_bias": end_bias, "beam_size": beam, "seq": seq[0].tolist(), "score": float(score),
            "ended": bool(len(seq[0]) and int(seq[0][-1]) == 1), "torch": torch.__version__}


def run_raises(case):
    """Run the reference on a configuration it cannot run and record the exception it ends in (type and message), after
    asserting that the restatement ends in the same one."""
    name, cname, H, W, beam = case
    cfg, m, sd = build_ref(cname, 6, beam_size=beam)
    img = synth.synth_images(1, H, W, seed=1)
    text = torch.zeros(1, 7, dtype=torch.long)
    rep = {"case": name, "config": cname, "H": H, "W": W, "beam_size": beam, "torch": torch.__version__}
    for who, fn in (("reference", lambda: m(img, text, is_train=False, is_test=True)),
                    ("oracle", lambda: R.forward(cfg, sd, img, text, is_train=False, is_test=True))):
        try:
            with torch.no_grad():
                fn()
            raise SystemExit(f"{name}: the {who} did not raise")
        except Exception as e:  # noqa: BLE001 -- the point is to record whatever it is
            rep[who] = {"type": next(c.__name__ for c in type(e).__mro__ if c.__module__ == "builtins"), "message": str(e)}
    assert rep["reference"] == rep["oracle"], rep
    rep.update(rep.pop("reference"))
    rep.pop("oracle")
    return rep, manifest(sd), cname


def run_train(case):
    name, cname, B, H, W, L, wseed, iseed = case
    cfg, m, sd = build_ref(cname, L, wseed=wseed)
    m.train()  # teacher forcing (tfm.py:103); BN switches to batch statistics
    img = synth.synth_images(B, H, W, seed=iseed)
    text = synth.synth_labels(B, max_len=L, seed=iseed)
    text[:, 8:] = 0
    text[:, 7] = R.END  # short labels so PAD masking is exercised
    # The reference train path uses BN batch statistics; the engine/oracle
    # parity target for the teacher-forced pass is the eval-BN encoder +
    # training-mode decoder.  Golden: decoder pass on the eval encoder memory.
    m.eval()
    with torch.no_grad():
        mem, _, _ = m.forward_encoder(img)
    m.predicter.train()
    logits = m.predicter(mem, text[:, :-1], True, False)[1]
    loss = torch.nn.functional.cross_entropy(logits.reshape(-1, logits.shape[-1]), text[:, 1:].reshape(-1),
                                             ignore_index=0, reduction="none").mean()
    with torch.no_grad():
        rl = R.tfm_full_pass(text[:, :-1], mem, sd, "predicter.Prediction.",
                             cfg["Prediction"]["params"]["num_decoder_layers"],
                             cfg["Prediction"]["params"]["nhead"], key_padding=True)
    d = maxdiff(rl, logits.detach())
    assert d <= TOL, d
    rep = {"case": name, "config": cname, "B": B, "H": H, "W": W, "max_seq_len": L, "wseed": wseed,
           "iseed": iseed, "loss": float(loss), "diff_logits": d, "logits_sum": float(logits.double().sum()),
           "text": text.tolist()}
    return rep


def grad_sample_index(key, numel):
    """Fixed pseudo-random positions of a tensor whose values are stored in the fixture."""
    import zlib
    g = torch.Generator().manual_seed(zlib.crc32(key.encode()))
    return torch.randint(0, numel, (min(GRAD_SAMPLES, numel),), generator=g)


def train_labels(cfg, B, L, iseed):
    """Teacher-forcing labels of a train_step case.  TFM converter: [GO]=1 first, [s]=2, PAD=0 (tfm_converter.py);
    Attn converter: [GO]=0 first AND as padding, [s]=1 (attn_converter.py:8-17,31-50)."""
    text = synth.synth_labels(B, max_len=L, seed=iseed)
    text[0, L // 2:] = 0
    text[0, L // 2 - 1] = R.END  # one short label so PAD masking / ignore_index are exercised
    if cfg["Prediction"]["name"] in ("Attn", "Attnv2"):
        t = text.clone()
        t[text == 1] = 0   # [GO]
        t[text == 2] = 1   # [s]
        text = t           # ordinary tokens (>= 4) and the padding zeros stay
    return text


def run_train_step(case):
    """forward_step + loss.backward() of the REFERENCE in module.train() mode (engine/training.py:76-91,126,137);
    checks the oracle's train_step_grads against it and stores loss, logits / gradient samples, gradient norms and
    the BatchNorm running statistics after the step."""
    name, cname, B, H, W, L, wseed, iseed = case
    cfg, m, sd = build_ref(cname, L, wseed=wseed)
    m.train()
    img = synth.synth_images(B, H, W, seed=iseed)
    text = train_labels(cfg, B, L, iseed)
    t0 = time.time()
    # GlobalContext blocks carry an nn.Dropout(0.25) that module.train() switches on (visual_attention.py:86-101): the
    # reference runs with torch.nn.functional.dropout replaced by a seeded mask source and the oracle gets the same masks
    gc_seed = GC_MASK_SEED if cname.endswith("G") else None
    src = SeqFirstMasks(R.GC_DROP, gc_seed) if gc_seed is not None else None
    real = torch.nn.functional.dropout

    def fake(input, p=0.5, training=True, inplace=False):
        if not training or p == 0.0:
            return input
        assert src is not None and p == R.GC_DROP, p
        return input * src.draw(input.shape)

    torch.nn.functional.dropout = fake
    try:
        _, preds, _ = m(img, text[:, :-1])  # is_train defaults to True (training.py:88)
    finally:
        torch.nn.functional.dropout = real
    cost = torch.nn.functional.cross_entropy(preds.view(-1, preds.shape[-1]), text[:, 1:].contiguous().view(-1),
                                             ignore_index=0, reduction="none")
    loss = cost.mean()
    loss.backward()
    ref_grads = {k: p.grad for k, p in m.named_parameters() if p.grad is not None}
    ref_frozen = [k for k, p in m.named_parameters() if p.grad is None]
    after = {k: v.detach().clone() for k, v in m.state_dict().items() if k.endswith(("running_mean", "running_var"))}
    osrc = SeqFirstMasks(R.GC_DROP, gc_seed) if gc_seed is not None else None
    oloss, ologits, ograds, obn = R.train_step_grads(cfg, slim_sd(sd), img, text,
                                                     drop=(lambda shape, kind: osrc.draw(shape) if kind == "gc"
                                                           else torch.ones(tuple(shape))) if osrc else None)
    assert abs(float(oloss) - float(loss)) <= 1e-5 * max(1.0, abs(float(loss))), (float(oloss), float(loss))
    assert maxdiff(ologits, preds.detach()) <= TOL
    assert sorted(ograds) == sorted(ref_grads), (set(ograds) ^ set(ref_grads), ref_frozen)
    worst, worst_key = 0.0, None
    for k, g in ref_grads.items():
        if k.endswith("global_cxt.bias"):  # a softmax ignores a shift of its logits: this gradient is rounding noise (1e-9)
            assert float(g.abs().max()) <= 1e-7 and float(ograds[k].abs().max()) <= 1e-7, k
            continue
        e = float((ograds[k].double() - g.double()).abs().max() / max(1e-6, float(g.double().abs().max())))
        if e > worst:
            worst, worst_key = e, k
    assert worst <= 5e-4, (worst, worst_key, float(ref_grads[worst_key].abs().max()))
    for k, v in after.items():
        assert maxdiff(obn[k], v) <= 1e-5, k
    arrays = {"logits": preds.detach()[:, ::LOGIT_STRIDE.get(name, 1)].numpy(), "text": text.numpy()}
    norms = {}
    for k, g in ref_grads.items():
        idx = grad_sample_index(k, g.numel())
        arrays["g:" + k] = g.reshape(-1)[idx].numpy()
        norms[k] = [float(g.double().norm()), float(g.double().sum())]
    for k, v in after.items():
        arrays["bn:" + k] = v.numpy()
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **arrays)
    return {"case": name, "config": cname, "B": B, "H": H, "W": W, "max_seq_len": L, "wseed": wseed, "iseed": iseed,
            "loss": float(loss), "oracle_worst_rel_grad_diff": worst, "n_grads": len(ref_grads), "frozen": ref_frozen,
            "logit_stride": LOGIT_STRIDE.get(name, 1), "gc_mask_seed": gc_seed,
            "grad_norms": norms, "seconds": round(time.time() - t0, 1), "torch": torch.__version__}


class SeqFirstMasks:
    """Keep masks drawn in call order from a seeded generator, in the (L, B, D) layout nn.TransformerDecoderLayer
    (batch_first=False) presents to its nn.Dropout modules."""

    def __init__(self, p, seed):
        self.p, self.g = p, torch.Generator().manual_seed(seed)

    def draw(self, shape):
        return (torch.rand(tuple(shape), generator=self.g) >= self.p).float() / (1.0 - self.p)


def run_train_dropout(case):
    """Pins WHERE the oracle applies dropout: the reference runs with torch.nn.functional.dropout replaced by a
    seeded mask source (its attention-probability dropout, which lives inside scaled_dot_product_attention and cannot
    be intercepted, is switched off for this run); the oracle gets the same masks at its "hidden" sites."""
    name, cname, B, H, W, L, wseed, iseed, mseed = case
    cfg, m, sd = build_ref(cname, L, wseed=wseed)
    lstm = cfg["Prediction"]["name"] != "TFM"
    p = cfg["Prediction"]["params"]["droprate" if lstm else "dropout"]
    m.train()
    if not lstm:
        for layer in m.predicter.Prediction.model.layers:
            layer.self_attn.dropout = 0.0
            layer.multihead_attn.dropout = 0.0
    img = synth.synth_images(B, H, W, seed=iseed)
    text = train_labels(cfg, B, L, iseed) if lstm else synth.synth_labels(B, max_len=L, seed=iseed)
    import random
    random.seed(mseed)  # scheduled sampling of the LSTM head draws random.random() once per step (seq2seq.py:312)
    src = SeqFirstMasks(p, mseed)
    real = torch.nn.functional.dropout

    def fake(input, p=0.5, training=True, inplace=False):
        if not training or p == 0.0:
            return input
        return input * src.draw(input.shape)

    torch.nn.functional.dropout = fake
    try:
        _, preds, _ = m(img, text[:, :-1])
    finally:
        torch.nn.functional.dropout = real
    loss = torch.nn.functional.cross_entropy(preds.view(-1, preds.shape[-1]), text[:, 1:].contiguous().view(-1),
                                             ignore_index=0, reduction="none").mean()
    loss.backward()
    ref_grads = {k: q.grad for k, q in m.named_parameters() if q.grad is not None}
    osrc = SeqFirstMasks(p, mseed)

    def drop(shape, kind):
        if kind == "attn":
            return torch.ones(tuple(shape))
        if len(shape) == 2:  # LSTM head: one [B, V] mask per step, drawn in that layout
            return osrc.draw(shape)
        Bq, Lq, D = shape
        return osrc.draw((Lq, Bq, D)).transpose(0, 1)

    flags = None
    if lstm:
        random.seed(mseed)
        tf = cfg["Prediction"]["params"].get("teacher_forcing", 1.0)
        flags = [1] + [0 if tf < random.random() else 1 for _ in range(L)]
    oloss, ologits, ograds, _ = R.train_step_grads(cfg, slim_sd(sd), img, text, drop=drop, flags=flags)
    assert abs(float(oloss) - float(loss)) <= 1e-5 * max(1.0, abs(float(loss))), (float(oloss), float(loss))
    assert maxdiff(ologits, preds.detach()) <= TOL
    worst = max(float((ograds[k].double() - g.double()).abs().max() / max(1e-6, float(g.double().abs().max())))
                for k, g in ref_grads.items())
    assert worst <= 5e-4, worst
    return {"case": name, "config": cname, "B": B, "H": H, "W": W, "max_seq_len": L, "wseed": wseed, "iseed": iseed,
            "mask_seed": mseed, "p": p, "flags": flags, "loss": float(loss), "logits_sum": float(preds.detach().double().sum()),
            "oracle_worst_rel_grad_diff": worst, "torch": torch.__version__}


def main():
    os.makedirs(GOLD, exist_ok=True)
    torch.manual_seed(0)
    summary = {"greedy": [], "beam": [], "train": [], "train_step": []}
    manifests = {}
    if os.environ.get("GOLDEN_ONLY") == "train_dropout":
        with open(os.path.join(GOLD, "cases.json")) as f:
            summary = json.load(f)
        with open(os.path.join(GOLD, "manifests.json")) as f:
            manifests = json.load(f)
        summary["train_dropout"] = []
        for case in TRAIN_DROPOUT_CASES:
            rep = run_train_dropout(case)
            summary["train_dropout"].append(rep)
            print("train_dropout", rep["case"], rep["loss"], rep["oracle_worst_rel_grad_diff"], flush=True)
        for cn in ("T2D", "TS0D"):
            manifests[cn] = manifest(build_ref(cn, 24)[2])
        with open(os.path.join(GOLD, "cases.json"), "w") as f:
            json.dump(summary, f, indent=1)
        with open(os.path.join(GOLD, "manifests.json"), "w") as f:
            json.dump(manifests, f)
        return
    if os.environ.get("GOLDEN_ONLY") == "vitpos":  # add / refresh only the ViTEncoder / ViTEncoderV2 fixtures (round 4)
        with open(os.path.join(GOLD, "cases.json")) as f:
            summary = json.load(f)
        with open(os.path.join(GOLD, "manifests.json")) as f:
            manifests = json.load(f)
        for case in VITPOS_GREEDY_CASES:
            rep, man, cname = run_greedy(case)
            manifests[cname] = man
            summary["greedy"] = [r for r in summary["greedy"] if r["case"] != rep["case"]] + [rep]
            print("greedy", rep["case"], "steps", rep["steps"], "dmem", rep["diff_mem_folded"], "dlogit", rep["diff_logits_cached"],
                  "gap", rep["min_top2_gap"], f'{rep["seconds"]}s', flush=True)
        for case in VITPOS_TRAIN_CASES:
            rep = run_train_step(case)
            summary["train_step"] = [r for r in summary["train_step"] if r["case"] != rep["case"]] + [rep]
            print("train_step", rep["case"], rep["loss"], rep["oracle_worst_rel_grad_diff"], f'{rep["seconds"]}s', flush=True)
        with open(os.path.join(GOLD, "cases.json"), "w") as f:
            json.dump(summary, f, indent=1)
        with open(os.path.join(GOLD, "manifests.json"), "w") as f:
            json.dump(manifests, f)
        return
    if os.environ.get("GOLDEN_ONLY") == "round2":  # add / refresh only the round-2 cases (C4 geometry, C1 full length, C3 size)
        with open(os.path.join(GOLD, "cases.json")) as f:
            summary = json.load(f)
        with open(os.path.join(GOLD, "manifests.json")) as f:
            manifests = json.load(f)
        for case in GREEDY_CASES:
            if case[0] not in ROUND2_GREEDY:
                continue
            rep, man, cname = run_greedy(case)
            manifests[cname] = man
            summary["greedy"] = [r for r in summary["greedy"] if r["case"] != rep["case"]] + [rep]
            print("greedy", rep["case"], "steps", rep["steps"], "dmem", rep["diff_mem_folded"], "dlogit", rep["diff_logits_cached"],
                  "gap", rep["min_top2_gap"], f'{rep["seconds"]}s', flush=True)
        for case in BEAM_CASES:
            if case[0] not in ROUND2_BEAM:
                continue
            rep = run_beam(case)
            summary["beam"] = [r for r in summary["beam"] if r["case"] != rep["case"]] + [rep]
            print("beam", rep["case"], rep["seq"], rep["score"], "completed", rep["completed"], flush=True)
        for case in TRAIN_STEP_CASES:
            if case[0] != "c3_train_step":
                continue
            rep = run_train_step(case)
            manifests["C3"] = manifests["C2"]
            summary["train_step"] = [r for r in summary["train_step"] if r["case"] != rep["case"]] + [rep]
            print("train_step", rep["case"], rep["loss"], rep["oracle_worst_rel_grad_diff"], f'{rep["seconds"]}s', flush=True)
        with open(os.path.join(GOLD, "cases.json"), "w") as f:
            json.dump(summary, f, indent=1)
        with open(os.path.join(GOLD, "manifests.json"), "w") as f:
            json.dump(manifests, f)
        return
    if os.environ.get("GOLDEN_ONLY") == "a15":  # add / refresh only the Bahdanau / one-hot / Luong fixtures
        with open(os.path.join(GOLD, "cases.json")) as f:
            summary = json.load(f)
        with open(os.path.join(GOLD, "manifests.json")) as f:
            manifests = json.load(f)
        for case in GREEDY_CASES:
            if case[0] not in A15_GREEDY:
                continue
            rep, man, cname = run_greedy(case)
            manifests[cname] = man
            summary["greedy"] = [r for r in summary["greedy"] if r["case"] != rep["case"]] + [rep]
            print("greedy", rep["case"], "steps", rep["steps"], "dmem", rep["diff_mem_folded"], "dlogit", rep["diff_logits_cached"],
                  "gap", rep["min_top2_gap"], f'{rep["seconds"]}s', flush=True)
        for case in ATTN_BEAM_CASES:
            if case[0] not in A15_BEAM:
                continue
            rep = run_attn_beam(case)
            summary["attn_beam"] = [r for r in summary["attn_beam"] if r["case"] != rep["case"]] + [rep]
            print("attn_beam", rep["case"], rep["seq"], rep["score"], "ended", rep["ended"], flush=True)
        summary["raises"] = []
        for case in RAISES_CASES:
            rep, man, cname = run_raises(case)
            manifests[cname] = man
            summary["raises"].append(rep)
            print("raises", rep["case"], rep["type"], rep["message"], flush=True)
        with open(os.path.join(GOLD, "cases.json"), "w") as f:
            json.dump(summary, f, indent=1)
        with open(os.path.join(GOLD, "manifests.json"), "w") as f:
            json.dump(manifests, f)
        return
    if os.environ.get("GOLDEN_ONLY") == "gcb":  # add / refresh only the GlobalContext greedy fixtures
        with open(os.path.join(GOLD, "cases.json")) as f:
            summary = json.load(f)
        with open(os.path.join(GOLD, "manifests.json")) as f:
            manifests = json.load(f)
        for case in GREEDY_CASES:
            if not case[0].endswith("g_greedy"):
                continue
            rep, man, cname = run_greedy(case)
            manifests[cname] = man
            summary["greedy"] = [r for r in summary["greedy"] if r["case"] != rep["case"]] + [rep]
            print("greedy", rep["case"], "steps", rep["steps"], "dmem", rep["diff_mem_folded"], "dlogit", rep["diff_logits_cached"], flush=True)
        with open(os.path.join(GOLD, "cases.json"), "w") as f:
            json.dump(summary, f, indent=1)
        with open(os.path.join(GOLD, "manifests.json"), "w") as f:
            json.dump(manifests, f)
        return
    if os.environ.get("GOLDEN_ONLY") == "attn_beam":  # refresh only the LSTM beam fixtures
        with open(os.path.join(GOLD, "cases.json")) as f:
            summary = json.load(f)
        summary["attn_beam"] = []
        for case in ATTN_BEAM_CASES:
            rep = run_attn_beam(case)
            summary["attn_beam"].append(rep)
            print("attn_beam", rep["case"], rep["seq"], rep["score"], "ended", rep["ended"], flush=True)
        with open(os.path.join(GOLD, "cases.json"), "w") as f:
            json.dump(summary, f, indent=1)
        return
    if os.environ.get("GOLDEN_ONLY") == "gcb_train":  # add / refresh only the GlobalContext training fixtures
        with open(os.path.join(GOLD, "cases.json")) as f:
            summary = json.load(f)
        for case in TRAIN_STEP_CASES:
            if case[0] not in os.environ.get("GOLDEN_CASES", "t2g_train_step,t1g_train_step").split(","):
                continue
            rep = run_train_step(case)
            summary["train_step"] = [r for r in summary["train_step"] if r["case"] != rep["case"]] + [rep]
            print("train_step", rep["case"], rep["loss"], rep["oracle_worst_rel_grad_diff"], f'{rep["seconds"]}s', flush=True)
        with open(os.path.join(GOLD, "cases.json"), "w") as f:
            json.dump(summary, f, indent=1)
        return
    if os.environ.get("GOLDEN_ONLY") == "train_step":  # refresh only the training fixtures
        with open(os.path.join(GOLD, "cases.json")) as f:
            summary = json.load(f)
        summary["train_step"] = []
        for case in TRAIN_STEP_CASES:
            rep = run_train_step(case)
            summary["train_step"].append(rep)
            print("train_step", rep["case"], rep["loss"], rep["oracle_worst_rel_grad_diff"], f'{rep["seconds"]}s', flush=True)
        with open(os.path.join(GOLD, "cases.json"), "w") as f:
            json.dump(summary, f, indent=1)
        return
    for case in GREEDY_CASES:
        rep, man, cname = run_greedy(case)
        manifests[cname] = man
        summary["greedy"].append(rep)
        print("greedy", rep["case"], "steps", rep["steps"], "dmem", rep["diff_mem_folded"],
              "dlogit", rep["diff_logits_cached"], "gap", rep["min_top2_gap"], f'{rep["seconds"]}s', flush=True)
    for case in BEAM_CASES:
        rep = run_beam(case)
        summary["beam"].append(rep)
        print("beam", rep["case"], rep["seq"], rep["score"], "completed", rep["completed"], flush=True)
    summary["attn_beam"] = []
    for case in ATTN_BEAM_CASES:
        rep = run_attn_beam(case)
        summary["attn_beam"].append(rep)
        print("attn_beam", rep["case"], rep["seq"], rep["score"], "ended", rep["ended"], flush=True)
    for case in TRAIN_CASES:
        rep = run_train(case)
        summary["train"].append(rep)
        print("train", rep["case"], rep["loss"], rep["diff_logits"], flush=True)
    for case in TRAIN_STEP_CASES:
        rep = run_train_step(case)
        summary["train_step"].append(rep)
        print("train_step", rep["case"], rep["loss"], rep["oracle_worst_rel_grad_diff"], f'{rep["seconds"]}s', flush=True)
    summary["raises"] = []
    for case in RAISES_CASES:
        rep, man, cname = run_raises(case)
        manifests[cname] = man
        summary["raises"].append(rep)
        print("raises", rep["case"], rep["type"], rep["message"], flush=True)
    summary["train_dropout"] = []
    for case in TRAIN_DROPOUT_CASES:
        rep = run_train_dropout(case)
        summary["train_dropout"].append(rep)
        print("train_dropout", rep["case"], rep["loss"], rep["oracle_worst_rel_grad_diff"], flush=True)
    for cn in ("T2D", "TS0D"):
        manifests[cn] = manifest(build_ref(cn, 24)[2])
    with open(os.path.join(GOLD, "cases.json"), "w") as f:
        json.dump(summary, f, indent=1)
    with open(os.path.join(GOLD, "manifests.json"), "w") as f:
        json.dump(manifests, f)
    print("wrote", GOLD)


if __name__ == "__main__":
    main()
