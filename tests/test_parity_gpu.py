"""GPU: the drop-in Model (HIP engine through the C-ABI) against the CPU oracle
on the same seeded weights/crops, and against the fixtures generated from the
reference.  Bar: greedy token ids bit-exact, logits within 1e-3 (north_star)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLD, engine_model, oracle_state_dict
from doc2tex_amd import synth
from oracle import restatement as R

pytestmark = pytest.mark.gpu
LOGIT_TOL = 1e-3  # BASELINE.json north_star: "logits within 1e-3 fp32"
MEM_TOL = {"fp32": 1e-4, "bf16x3": 5e-4, "fp16x2": 2e-3, "mixed": 1e-3}  # encoder memory, relative to its largest magnitude (our own bar; the
# opt-in arithmetics only when the suite is run under D2T_CONV_PRECISION=fp16x2 / mixed)


def _case(cases, kind, name):
    return next(c for c in cases[kind] if c["case"] == name)


def _score_tol(m, n):
    """Bar for a beam score = a SUM of n log-probabilities.  Split-bf16 / fp32 encoders hold each to ~1e-5 (the logits bar is
    1e-3 per value): 1e-3 for the short fixtures, 2e-5 per token for the 151-token one (a sum around -520, whose own fp32
    spacing is 6e-5).  fp16x2 encoder memory (opt-in; these tests run in it under D2T_CONV_PRECISION=fp16x2) moves a logit by
    a few 1e-4: 5e-3 on the sum."""
    return 5e-3 if m.effective_conv_precision() == "fp16x2" else max(1e-3, 2e-5 * n)


def _run_engine(c, B=None):
    cfg, m = engine_model(c["config"], c["max_seq_len"], c["wseed"], c["end_bias"], beam_size=c.get("beam_size"))
    B = B or c["B"]
    img = synth.synth_images(B, c["H"], c["W"], seed=c["iseed"]).cuda()
    text = torch.full((B, 1), R.GO, dtype=torch.long, device="cuda")
    with torch.no_grad():
        mem, shape, pad = m.forward_encoder(img)
        preds, logits, extra = m(img, text, is_train=False, is_test=c["is_test"])
    torch.cuda.synchronize()
    return cfg, m, mem.cpu(), shape, pad, preds.cpu(), logits.cpu()


@pytest.mark.parametrize("name", ["t2_greedy", "t2_greedy_early", "t2_greedy_late", "t1_greedy", "c2_small_crop",
                                  "c2_greedy", "c1_greedy", "c0_greedy", "c0_greedy_early", "ts0_greedy",
                                  "s0_greedy", "s0_small_crop", "t2g_greedy", "t1g_greedy",
                                  # BASELINE configs[4] geometry (max_dimension [160, 640]: 9x161 grid, 406 memory tokens) at
                                  # each bucket of SURVEY 8d, and configs[1] at its full 151 steps
                                  "c4_greedy_160", "c4_greedy_128", "c4_greedy_96", "c1_greedy_full",
                                  # Bahdanau cell, one-hot targets (the remaining branches of Attention.__init__)
                                  "b0_greedy", "b0_greedy_early", "tb0_greedy", "to0_greedy",
                                  # ViTEncoder: learned position table through the bicubic resize kernel (the table's own
                                  # grid / smaller / one direction / beyond max_dimension; 1 x 2 patches: same token count on
                                  # a square feature map, and resized); ViTEncoderV2: learned table, prefix slice
                                  "v1_greedy_full", "v1_greedy_small", "v1_greedy_mid", "v1_greedy_narrow", "v1_greedy_big",
                                  "v1p_greedy_samecount", "v1p_greedy_interp", "v2_greedy_small", "v2_greedy_full"])
def test_greedy_vs_reference_fixture(cases, name):
    c = _case(cases, "greedy", name)
    z = np.load(os.path.join(GOLD, name + ".npz"))
    cfg, m, mem, shape, pad, preds, logits = _run_engine(c)
    assert list(mem.shape) == c["mem_shape"]
    assert (list(shape) if shape else None) == c["output_shape"]
    assert (list(pad) if pad else None) == c["feat_pad"]
    scale = max(1.0, c["mem_absmax"])
    rows = z["mem_rows"].tolist()
    dmem = float(np.abs(mem[:, rows].numpy() - z["mem_sample"]).max()) / scale
    # fp32 arithmetic: 1e-4; split-bf16 arithmetic (2^-16 relative per product through 32 convolutions): 5e-4; fp16x2 (feature
    # maps rounded to 11 bits where a layer stores them): 2e-3 -- the north_star bar is the tokens and the logits below
    assert dmem <= MEM_TOL[m.effective_conv_precision()], f"encoder memory rel err {dmem}"
    assert preds.shape[1] == c["steps"], (preds.shape, c["steps"])
    assert np.array_equal(preds.numpy(), z["tokens"]), "greedy token ids differ from the reference"
    steps = z["logit_steps"].tolist()
    dl = float(np.abs(logits[:, steps].numpy() - z["logits_sample"]).max())
    assert dl <= LOGIT_TOL, f"logits differ by {dl}"


@pytest.mark.parametrize("name", ["t2_greedy", "c2_small_crop"])
def test_full_tensors_vs_oracle(cases, manifests, name):
    """Whole memory / logits tensors against the oracle run here on the CPU."""
    c = _case(cases, "greedy", name)
    cfg, m, mem, shape, pad, preds, logits = _run_engine(c)
    ocfg, sd = oracle_state_dict(c["config"], manifests[c["config"]], c["max_seq_len"], c["wseed"], c["end_bias"])
    img = synth.synth_images(c["B"], c["H"], c["W"], seed=c["iseed"])
    with torch.no_grad():
        omem, oshape, opad = R.forward_encoder(ocfg, sd, img, faithful=False)
        op, ol, _ = R.forward(ocfg, sd, img, torch.full((c["B"], 1), R.GO, dtype=torch.long), is_test=c["is_test"])
    assert float((mem - omem).abs().max()) <= (2e-3 if m.effective_conv_precision() == "fp16x2" else 1e-4) * max(1.0, float(omem.abs().max()))
    assert torch.equal(preds, op)
    assert float((logits - ol).abs().max()) <= LOGIT_TOL


def test_batch_shard_invariance(cases):
    """Data-parallel contract: decoding a batch in shards gives the same
    tokens as decoding it whole (kernels are batch-size invariant)."""
    c = dict(_case(cases, "greedy", "t2_greedy"))
    cfg, m = engine_model(c["config"], c["max_seq_len"], c["wseed"], c["end_bias"])
    img = synth.synth_images(6, c["H"], c["W"], seed=77).cuda()
    text = torch.full((6, 1), R.GO, dtype=torch.long, device="cuda")
    with torch.no_grad():
        p_all, l_all, _ = m(img, text, is_train=False)
        parts = [m(img[i:i + 2], text[i:i + 2], is_train=False) for i in range(0, 6, 2)]
    p_sh = torch.cat([p[0] for p in parts])
    l_sh = torch.cat([p[1] for p in parts])
    assert torch.equal(p_all, p_sh)
    assert torch.equal(l_all, l_sh), "logits must be bit-identical across shardings"


def test_two_row_decode_blocks_do_not_couple_their_rows(cases):
    """The d_model-512 decoder (ResNet + TFM stacks: T1 / C1) steps two rows per block (decode.hip decoder_row2_kernel): a row's
    logits must not depend on which row shares its block, nor on the batch being odd (the last block repeats its row)."""
    c = dict(_case(cases, "greedy", "t1_greedy"))
    cfg, m = engine_model(c["config"], c["max_seq_len"], c["wseed"], c["end_bias"])
    img = synth.synth_images(5, c["H"], c["W"], seed=78).cuda()
    text = torch.full((5, 1), R.GO, dtype=torch.long, device="cuda")
    with torch.no_grad():
        p_all, l_all, _ = m(img, text, is_train=False)
        parts = [m(img[i:i + n], text[i:i + n], is_train=False) for i, n in ((0, 1), (1, 3), (4, 1))]
        swapped = m(img.flip(0), text, is_train=False)
    assert torch.equal(p_all, torch.cat([q[0] for q in parts])) and torch.equal(l_all, torch.cat([q[1] for q in parts]))
    assert torch.equal(l_all, swapped[1].flip(0))


def test_determinism_and_weight_reload(cases):
    c = _case(cases, "greedy", "t2_greedy")
    cfg, m = engine_model(c["config"], c["max_seq_len"], c["wseed"], 0.0)
    img = synth.synth_images(2, c["H"], c["W"], seed=c["iseed"]).cuda()
    text = torch.full((2, 1), R.GO, dtype=torch.long, device="cuda")
    with torch.no_grad():
        a = m(img, text, is_train=False)
        b = m(img, text, is_train=False)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
        # in-place weight change must be picked up (engine re-packs)
        m.predicter.Prediction.proj.bias[5] += 50.0  # in-place under no_grad bumps ._version
        d = m(img, text, is_train=False)
    assert (d[0] == 5).all()


def test_headline_shape_properties(cases):
    """BASELINE configs[2] at full size (B=64, 128x512, 151 steps): the first two
    rows reproduce the reference fixture, logits are finite, every row has 151 tokens."""
    c = _case(cases, "greedy", "c2_greedy")
    z = np.load(os.path.join(GOLD, "c2_greedy.npz"))
    cfg, m, mem, shape, pad, preds, logits = _run_engine(c, B=64)
    assert preds.shape == (64, 151) and logits.shape == (64, 151, synth.VOCAB)
    assert torch.isfinite(logits).all()
    assert np.array_equal(preds[:2].numpy(), z["tokens"])
    assert torch.equal(preds, logits.argmax(-1))  # tokens are the argmax of the returned logits


def test_headline_shape_in_the_benchmarked_serving_mode(cases):
    """What bench.py times -- B=64, 128x512, pipelined, THREE decode chains, decode groups of SIX batches (384 rows per step
    loop), no reserved block slots -- and the earlier serving configurations (two chains; groups of three / two; ungrouped
    with 64 reserved slots).  Seven batches: with groups of six one full 384-row group and an incomplete one (64 rows, launched by
    synchronize()) both run.  Every batch comes back exactly as the synchronous path returns it, and the fixture rows
    (computed by the reference) stay exact."""
    c = _case(cases, "greedy", "c2_greedy")
    z = np.load(os.path.join(GOLD, "c2_greedy.npz"))
    cfg, m = engine_model(c["config"], c["max_seq_len"], c["wseed"], c["end_bias"])
    base = synth.synth_images(c["B"], c["H"], c["W"], seed=c["iseed"])
    imgs = []
    for i in range(7):
        x = synth.synth_images(64, c["H"], c["W"], seed=300 + i)
        x[: c["B"]] = base  # the fixture rows ride in every batch
        imgs.append(x.cuda())
    text = torch.full((64, 1), R.GO, dtype=torch.long, device="cuda")
    steps = z["logit_steps"].tolist()
    with torch.no_grad():
        ref = [m(x, text, is_train=False) for x in imgs]
        ref = [(p.clone(), l.clone()) for p, l, _ in ref]
        for chains, group, reserve in ((3, 6, 0), (2, 6, 0), (4, 3, 0), (2, 2, 0), (2, 1, 64)):
            m.pipelined, m.decode_chains, m.decode_group, m.reserved_blocks = True, chains, group, reserve
            got = [m(x, text, is_train=False) for x in imgs]
            m.synchronize()
            torch.cuda.synchronize()
            for (p, l, extra), (rp, rl) in zip(got, ref):
                assert torch.equal(p, rp) and torch.equal(l, rl), (chains, group, reserve)
                assert np.array_equal(p[: c["B"]].cpu().numpy(), z["tokens"])
                assert float(np.abs(l[: c["B"], steps].cpu().numpy() - z["logits_sample"]).max()) <= LOGIT_TOL
                hp, hl = extra["decode"].result()  # the documented way to consume a pipelined forward, at every group index
                assert hp.shape == rp.shape and torch.equal(hp, rp) and torch.equal(hl, rl)
    m.pipelined, m.decode_group = False, 1


def test_all_64_rows_of_a_headline_batch_vs_the_oracle(cases, manifests):
    """BASELINE configs[2], one whole batch (B=64, 128x512, 151 steps) in the benchmarked serving mode: EVERY row's token
    ids equal the KV-cached CPU oracle's and every logit is within 1e-3 (the oracle is pinned on the reference by
    tests/test_oracle_golden.py; its first rows are the reference's own fixture).  ~20 s of host time."""
    c = _case(cases, "greedy", "c2_greedy")
    z = np.load(os.path.join(GOLD, "c2_greedy.npz"))
    cfg, m = engine_model(c["config"], c["max_seq_len"], c["wseed"], c["end_bias"])
    ocfg, sd = oracle_state_dict(c["config"], manifests[c["config"]], c["max_seq_len"], c["wseed"], c["end_bias"])
    img = synth.synth_images(64, c["H"], c["W"], seed=4100)
    img[: c["B"]] = synth.synth_images(c["B"], c["H"], c["W"], seed=c["iseed"])
    text = torch.full((64, 1), R.GO, dtype=torch.long)
    torch.set_num_threads(max(1, min(32, len(os.sched_getaffinity(0)))))
    with torch.no_grad():
        op, ol, _ = R.forward(ocfg, sd, img, text, is_test=False, faithful=False)
        m.pipelined, m.decode_chains, m.decode_group, m.reserved_blocks = True, 3, 6, 0
        p, l, extra = m(img.cuda(), text.cuda(), is_train=False)
        p, l = extra["decode"].result()
        torch.cuda.synchronize()
    m.pipelined, m.decode_group = False, 1
    assert np.array_equal(op[: c["B"]].numpy(), z["tokens"])  # the oracle's rows 0.. are the reference's
    assert p.shape == (64, 151) and torch.equal(p.cpu(), op), "greedy token ids differ from the oracle on some of the 64 rows"
    dl = float((l.cpu() - ol).abs().max())
    assert dl <= LOGIT_TOL, f"logits differ by {dl}"


def test_config_c1_at_its_own_batch_and_length(cases):
    """BASELINE configs[1] at full size (ResNet + PositionalEncoding2D + TFM-2, 64x256 crops, B=32, 151 steps): the first
    rows reproduce the reference fixture (run there with B=2), every row decodes 151 tokens, halves decode identically."""
    c = _case(cases, "greedy", "c1_greedy_full")
    z = np.load(os.path.join(GOLD, "c1_greedy_full.npz"))
    cfg, m = engine_model(c["config"], c["max_seq_len"], c["wseed"], c["end_bias"])
    img = synth.synth_images(32, c["H"], c["W"], seed=310)
    img[: c["B"]] = synth.synth_images(c["B"], c["H"], c["W"], seed=c["iseed"])
    img = img.cuda()
    text = torch.full((32, 1), R.GO, dtype=torch.long, device="cuda")
    with torch.no_grad():
        preds, logits, _ = m(img, text, is_train=False)
        halves = [m(img[i:i + 16], text[:16], is_train=False) for i in (0, 16)]
    torch.cuda.synchronize()
    assert preds.shape == (32, 151) and logits.shape == (32, 151, synth.VOCAB)
    assert torch.isfinite(logits).all()
    assert np.array_equal(preds[: c["B"]].cpu().numpy(), z["tokens"])
    steps = z["logit_steps"].tolist()
    assert float(np.abs(logits[: c["B"], steps].cpu().numpy() - z["logits_sample"]).max()) <= LOGIT_TOL
    assert torch.equal(preds, logits.argmax(-1))
    assert torch.equal(preds, torch.cat([h[0] for h in halves])) and torch.equal(logits, torch.cat([h[1] for h in halves]))


@pytest.mark.parametrize("name", ["t2_beam5", "c2_beam5", "t2_beam3_nofinish",
                                  # config C4 itself (beam 5 under max_dimension [160, 640]), every bucket; _full = all 151 steps
                                  "c4_beam5_160", "c4_beam5_128", "c4_beam5_96", "c4_beam5_160_full"])
def test_beam_vs_reference_fixture(cases, name):
    """forward_beam + Beam (tfm.py:145-186, tools/beam.py) for one sample: the best
    hypothesis' token ids are exact and its score is within 1e-3."""
    c = _case(cases, "beam", name)
    cfg, m = engine_model(c["config"], c["max_seq_len"], c["wseed"], c["end_bias"], beam_size=c["beam_size"])
    img = synth.synth_images(1, c["H"], c["W"], seed=c["iseed"]).cuda()
    text = torch.full((1, 1), R.GO, dtype=torch.long, device="cuda")
    with torch.no_grad():
        seq, score, _ = m(img, text, is_train=False, is_test=True)
        seq2, score2, _ = m(img, text, is_train=False, is_test=True)  # fresh beam state per call
    assert seq.shape[0] == 1 and seq[0].tolist() == c["seq"], (seq, c["seq"])
    assert abs(score - c["score"]) <= _score_tol(m, len(c["seq"])), (score, c["score"])
    assert torch.equal(seq, seq2) and score == score2


def test_beam_vs_oracle_other_seeds(cases, manifests):
    c = _case(cases, "beam", "t2_beam5")
    for iseed, eb in [(501, 1.8), (502, 1.75), (503, 0.0)]:
        cfg, m = engine_model(c["config"], c["max_seq_len"], c["wseed"], eb, beam_size=5)
        ocfg, sd = oracle_state_dict(c["config"], manifests[c["config"]], c["max_seq_len"], c["wseed"], eb)
        ocfg["beam_size"] = 5
        img = synth.synth_images(1, c["H"], c["W"], seed=iseed)
        text = torch.full((1, 1), R.GO, dtype=torch.long)
        with torch.no_grad():
            seq, score, _ = m(img.cuda(), text.cuda(), is_train=False, is_test=True)
            oseq, oscore, _ = R.forward(ocfg, sd, img, text, is_test=True)
        assert seq[0].tolist() == oseq[0].tolist(), (iseed, eb)
        assert abs(score - oscore) <= _score_tol(m, seq.shape[1])


@pytest.mark.parametrize("name", ["ts0_beam5", "c0_beam3", "c0_beam3_end", "s0_beam10", "s0_beam10_late", "ts0_beam4_nofinish",
                                  "b0_beam3", "b0_beam3_end", "tb0_beam4", "to0_beam5", "to0_beam5_end"])
def test_attn_beam_vs_reference_fixture(cases, name):
    """LSTM-attention beam search (Attention.forward_beam seq2seq.py:83-222, AttentionV2.forward_beam
    seq2seq_v2.py:12-174; what config/test.yaml runs): token ids exact, score within 1e-3."""
    c = _case(cases, "attn_beam", name)
    cfg, m = engine_model(c["config"], c["max_seq_len"], c["wseed"], c["end_bias"], beam_size=c["beam_size"])
    img = synth.synth_images(1, c["H"], c["W"], seed=c["iseed"]).cuda()
    text = torch.zeros(1, c["max_seq_len"] + 1, dtype=torch.long, device="cuda")
    with torch.no_grad():
        seq, score, _ = m(img, text, is_train=False, is_test=True)
        seq2, score2, _ = m(img, text, is_train=False, is_test=True)
    assert seq.shape[0] == 1 and seq[0].tolist() == c["seq"], (seq, c["seq"])
    assert abs(float(score) - c["score"]) <= 1e-3, (score, c["score"])
    assert torch.equal(seq, seq2) and float(score) == float(score2)


def test_attn_beam_vs_oracle_other_seeds(cases, manifests):
    c = _case(cases, "attn_beam", "ts0_beam5")
    for iseed, eb, beam in [(601, 0.3, 5), (602, 0.25, 7), (603, 0.35, 3), (604, 0.0, 2)]:
        cfg, m = engine_model(c["config"], c["max_seq_len"], c["wseed"], eb, beam_size=beam)
        ocfg, sd = oracle_state_dict(c["config"], manifests[c["config"]], c["max_seq_len"], c["wseed"], eb)
        ocfg["beam_size"] = beam
        img = synth.synth_images(1, c["H"], c["W"], seed=iseed)
        text = torch.zeros(1, c["max_seq_len"] + 1, dtype=torch.long)
        with torch.no_grad():
            seq, score, _ = m(img.cuda(), text.cuda(), is_train=False, is_test=True)
            oseq, oscore, _ = R.forward(ocfg, sd, img, text, is_train=False, is_test=True)
        assert seq[0].tolist() == oseq[0].tolist(), (iseed, eb, beam)
        assert abs(float(score) - oscore) <= 1e-3


def test_batched_beam_equals_per_sample_beam(cases):
    """Model.beam_search_batch (hypotheses of all samples in one step loop) returns, for every sample, exactly what
    the reference-shaped single-sample call returns -- including samples that finish at different steps."""
    c = _case(cases, "beam", "t2_beam5")
    for eb, beam in [(1.8, 5), (1.75, 3), (0.0, 4)]:
        cfg, m = engine_model(c["config"], c["max_seq_len"], c["wseed"], eb, beam_size=beam)
        img = synth.synth_images(7, c["H"], c["W"], seed=c["iseed"]).cuda()
        text = torch.full((1, 1), R.GO, dtype=torch.long, device="cuda")
        with torch.no_grad():
            single = [m(img[i:i + 1], text, is_train=False, is_test=True)[:2] for i in range(7)]
            batch = m.beam_search_batch(img)
        assert len({tuple(s[0].tolist()) for s, _ in single}) > 1 or eb == 0.0
        for (s1, v1), (s2, v2) in zip(single, batch):
            assert torch.equal(s1, s2), (eb, beam)
            assert v1 == v2


def test_config_c4_batched_beam_at_full_size(cases):
    """BASELINE configs[4] at its own workload: the C2 model under max_dimension [160, 640], beam 5, B=128 crops of one
    bucket per batch (160x640 -> 406 memory tokens; 96x384 -> the flat-prefix slice of the big table).  The batched
    extension equals the reference-shaped single-sample call on every sample checked, and the fixture sample (computed
    by the reference) rides in row 0.  Rows differ in contrast and brightness so that, at this [s] bias, some samples
    finish at the first step and others run to the length limit (seeded noise alone gives one behaviour for all)."""
    for name, picks in (("c4_beam5_160", [0, 1, 16, 31, 63, 127]), ("c4_beam5_96", [0, 24, 31, 127])):
        c = _case(cases, "beam", name)
        cfg, m = engine_model(c["config"], c["max_seq_len"], c["wseed"], c["end_bias"], beam_size=c["beam_size"])
        assert cfg["max_dimension"] == [160, 640] and cfg["beam_size"] == 5
        img = synth.synth_images(128, c["H"], c["W"], seed=1300)
        i = torch.arange(128)
        scale = (0.1 + 0.9 * (i % 8).float() / 7).view(-1, 1, 1, 1)
        offs = (((i // 8) % 4).float() - 1.5).view(-1, 1, 1, 1) * 0.3
        img = (img * scale + offs).clamp(-1, 1)
        img[:1] = synth.synth_images(1, c["H"], c["W"], seed=c["iseed"])
        img = img.cuda()
        text = torch.full((1, 1), R.GO, dtype=torch.long, device="cuda")
        with torch.no_grad():
            batch = m.beam_search_batch(img)
            single = {k: m(img[k:k + 1], text, is_train=False, is_test=True)[:2] for k in picks}
        assert len(batch) == 128
        assert batch[0][0][0].tolist() == c["seq"] and abs(batch[0][1] - c["score"]) <= 1e-3
        for k, (s1, v1) in single.items():
            assert torch.equal(s1, batch[k][0]), (name, k)
            assert v1 == batch[k][1], (name, k)
        if name == "c4_beam5_160":
            assert len({len(b[0][0]) for b in batch}) > 1  # both kinds of sample present (checked with the oracle: rows 24 / 31)


def test_beam_with_one_cross_attention_block_per_sample(cases):
    """Model.beam_shared_tile: the cross-attention of all live hypotheses of a sample in ONE block that stages the sample's
    memory tiles once per layer and step (north_star's LDS-resident K/V, in the absorbed form).  Same sequences as the
    per-row path on the reference fixtures (single-sample API) and in the batched API, scores within the beam bar."""
    for name in ("t2_beam5", "c2_beam5", "c4_beam5_96"):
        c = _case(cases, "beam", name)
        cfg, m = engine_model(c["config"], c["max_seq_len"], c["wseed"], c["end_bias"], beam_size=c["beam_size"])
        m.beam_shared_tile = True
        img = synth.synth_images(1, c["H"], c["W"], seed=c["iseed"]).cuda()
        text = torch.full((1, 1), R.GO, dtype=torch.long, device="cuda")
        with torch.no_grad():
            seq, score, _ = m(img, text, is_train=False, is_test=True)
        assert seq[0].tolist() == c["seq"], (name, seq, c["seq"])
        assert abs(score - c["score"]) <= _score_tol(m, len(c["seq"]))
    c = _case(cases, "beam", "t2_beam5")
    for eb, beam in [(1.8, 5), (1.75, 3), (0.0, 6)]:
        cfg, m = engine_model(c["config"], c["max_seq_len"], c["wseed"], eb, beam_size=beam)
        img = synth.synth_images(7, c["H"], c["W"], seed=c["iseed"]).cuda()
        with torch.no_grad():
            ref = m.beam_search_batch(img)
            m.beam_shared_tile = True
            got = m.beam_search_batch(img)
        for (s1, v1), (s2, v2) in zip(ref, got):
            assert torch.equal(s1, s2), (eb, beam)
            assert abs(v1 - v2) <= _score_tol(m, s1.shape[1])


def test_batched_attn_beam_equals_per_sample_beam(cases):
    """The same for the LSTM-attention head (Attnv2 on the ViT encoder, Attn on VGG + BiLSTM)."""
    for cname, H, W, L, eb, beam in [("TS0", 48, 64, 14, 0.3, 5), ("TS0", 48, 64, 8, 0.0, 3), ("C0", 32, 320, 12, 0.17, 4)]:
        cfg, m = engine_model(cname, L, 1234, eb, beam_size=beam)
        img = synth.synth_images(6, H, W, seed=1200).cuda()
        text = torch.zeros(1, L + 1, dtype=torch.long, device="cuda")
        with torch.no_grad():
            single = [m(img[i:i + 1], text, is_train=False, is_test=True)[:2] for i in range(6)]
            batch = m.beam_search_batch(img)
        for (s1, v1), (s2, v2) in zip(single, batch):
            assert torch.equal(s1, s2), (cname, eb, beam)
            assert float(v1) == float(v2)


def test_beam_rejects_batches():
    cfg, m = engine_model("T2", 8, beam_size=3)
    img = synth.synth_images(2, 48, 64).cuda()
    with pytest.raises(AssertionError):  # tfm.py:146-148
        m(img, torch.ones(2, 1, dtype=torch.long, device="cuda"), is_train=False, is_test=True)


@pytest.mark.parametrize("chains,precision", [(1, "fp32"), (2, "fp32"), (2, "bf16x3")])
def test_pipelined_decode_equals_synchronous(cases, chains, precision):
    """Opt-in cross-batch pipelining (decode of batch i overlaps the encoder of batch i+1; with two decode
    chains also the decode of batch i-1) returns exactly what the synchronous path returns."""
    c = _case(cases, "greedy", "t2_greedy")
    cfg, m = engine_model(c["config"], c["max_seq_len"], c["wseed"], c["end_bias"])
    m.conv_precision = precision
    imgs = [synth.synth_images(3, c["H"], c["W"], seed=900 + i).cuda() for i in range(7)]
    text = torch.full((3, 1), R.GO, dtype=torch.long, device="cuda")
    with torch.no_grad():
        ref = [m(x, text, is_train=False) for x in imgs]
        ref = [(p.clone(), l.clone()) for p, l, _ in ref]
        m.pipelined = True
        m.decode_chains = chains
        got = []
        for x in imgs:
            p, l, _ = m(x, text, is_train=False)
            got.append((p, l))
            if len(got) >= 2:  # consume with one batch of lag, as a serving loop would
                m.synchronize(host_sync=False)
        m.synchronize()
        torch.cuda.synchronize()
    # every result stays valid for as long as the caller holds it (fresh tensors per decode, no ring to overrun)
    for (p, l), (rp, rl) in zip(got, ref):
        assert torch.equal(p, rp) and torch.equal(l, rl)
    m.pipelined = False
    # a synchronous call right after pipelined ones still matches
    with torch.no_grad():
        p, l, _ = m(imgs[0], text, is_train=False)
    assert torch.equal(p, ref[0][0]) and torch.equal(l, ref[0][1])


@pytest.mark.parametrize("group", [2, 3])
def test_grouped_pipelined_decode_equals_synchronous(cases, group):
    """model.decode_group: the rows of `group` consecutive forwards share one decode step loop; every forward still
    returns exactly its synchronous result (rows are independent, kernels are dispatched by layer shape only).  Seven
    forwards: the last group is incomplete and is launched by synchronize()."""
    c = _case(cases, "greedy", "t2_greedy")
    cfg, m = engine_model(c["config"], c["max_seq_len"], c["wseed"], c["end_bias"])
    imgs = [synth.synth_images(3, c["H"], c["W"], seed=950 + i).cuda() for i in range(7)]
    text = torch.full((3, 1), R.GO, dtype=torch.long, device="cuda")
    with torch.no_grad():
        ref = [m(x, text, is_train=False) for x in imgs]
        ref = [(p.clone(), l.clone()) for p, l, _ in ref]
        m.pipelined, m.decode_chains, m.decode_group = True, 2, group
        outs = [m(x, text, is_train=False) for x in imgs]
        got = [o[:2] for o in outs]
        # result() -- the documented way to consume a pipelined forward -- at EVERY position of a group, without early exit
        for o, (rp, rl) in zip(outs, ref):
            hp, hl = o[2]["decode"].result()
            assert o[2]["decode"].steps() == rp.shape[1]
            assert hp.shape == rp.shape and torch.equal(hp, rp) and torch.equal(hl, rl)
        m.synchronize()
        torch.cuda.synchronize()
    for (p, l), (rp, rl) in zip(got, ref):  # all seven, however many groups ago they were decoded
        assert p.shape == rp.shape and torch.equal(p, rp) and torch.equal(l, rl)
    m.pipelined, m.decode_group = False, 1
    with torch.no_grad():
        p, l, _ = m(imgs[0], text, is_train=False)
    assert torch.equal(p, ref[0][0]) and torch.equal(l, ref[0][1])


def test_pipelined_results_carry_a_completion_handle(cases):
    """Serving safety: a pipelined forward returns addition_outputs["decode"], a handle on exactly that batch's decode.
    wait() makes the batch readable while later batches are still in flight (launching an incomplete group if need be),
    done() polls, results of earlier batches are never overwritten by later ones, and the C-ABI tickets behind it count
    up by one per launched decode."""
    c = _case(cases, "greedy", "t2_greedy")
    cfg, m = engine_model(c["config"], c["max_seq_len"], c["wseed"], c["end_bias"])
    imgs = [synth.synth_images(3, c["H"], c["W"], seed=980 + i).cuda() for i in range(12)]
    text = torch.full((3, 1), R.GO, dtype=torch.long, device="cuda")
    with torch.no_grad():
        ref = [tuple(t.clone() for t in m(x, text, is_train=False)[:2]) for x in imgs]
        for group in (1, 2):
            m.pipelined, m.decode_chains, m.decode_group = True, 2, group
            eng = m.engine()
            t0 = int(eng.lib.d2t_decode_last_ticket(eng.ctx))
            outs = []
            for i, x in enumerate(imgs):
                p, l, extra = m(x, text, is_train=False)
                outs.append((p, l, extra["decode"]))
                if i == 4:  # consume batch 3 right now: five forwards issued, batch 4's group (group 2) not even launched
                    h = outs[3][2]
                    h.wait(host_sync=True)
                    assert h.done() and torch.equal(outs[3][0], ref[3][0]) and torch.equal(outs[3][1], ref[3][1])
            outs[-1][2].wait(host_sync=True)
            assert outs[-1][2].done()
            m.synchronize()
            assert int(eng.lib.d2t_decode_last_ticket(eng.ctx)) - t0 == len(imgs) // group
            assert all(h.done() for _, _, h in outs)
            for (p, l, _), (rp, rl) in zip(outs, ref):  # twelve batches later the first ones are still intact
                assert torch.equal(p, rp) and torch.equal(l, rl)
            with pytest.raises(RuntimeError):
                eng.wait_ticket(10 ** 9)
    m.pipelined, m.decode_group = False, 1


@pytest.mark.parametrize("group", [1, 2, 3])
def test_pipelined_early_exit_decided_on_the_device(cases, group):
    """What api/infer.py calls -- model(image, text, is_test=True) (early exit, tfm.py:138-140) -- in pipelined serving
    mode: the loop is one graph launch whose kernels stop working once every batch of the decode group has ended; each
    batch comes back cut at ITS OWN first all-ended step, exactly as its synchronous is_test call returns it.  Batches end
    at different steps (and one never does), so groups mix them."""
    c = _case(cases, "greedy", "t2_greedy_early")  # end_bias 1.81: rows end at different steps
    cfg, m = engine_model(c["config"], c["max_seq_len"], c["wseed"], c["end_bias"])
    imgs = [synth.synth_images(3, c["H"], c["W"], seed=c["iseed"] + i).cuda() for i in range(6)]
    imgs.append((synth.synth_images(3, c["H"], c["W"], seed=77) * 0.05).cuda())  # a faint crop: different dynamics
    text = torch.full((3, 1), R.GO, dtype=torch.long, device="cuda")
    with torch.no_grad():
        ref = [tuple(t.clone() for t in m(x, text, is_train=False, is_test=True)[:2]) for x in imgs]
        steps = [p.shape[1] for p, _ in ref]
        assert len(set(steps)) > 1, steps  # the batches really stop at different steps
        m.pipelined, m.decode_chains, m.decode_group = True, 2, group
        outs = [m(x, text, is_train=False, is_test=True) for x in imgs]
        full = c["max_seq_len"] + 1
        for (p, l, extra), (rp, rl) in zip(outs, ref):
            assert p.shape == (3, full) and l.shape[:2] == (3, full)  # full-size buffers while the decode is in flight
            cp, cl = extra["decode"].result()
            assert cp.shape == rp.shape, (cp.shape, rp.shape)
            assert torch.equal(cp, rp) and torch.equal(cl, rl)
            # past the group's stop step the full-size tensors are defined (PAD ids, zero logits), never stale memory;
            # between this batch's own cut and the group's stop they hold real decode steps
            n = cp.shape[1]
            tail_p, tail_l = p[:, n:], l[:, n:]
            if tail_p.numel():
                assert torch.isfinite(tail_l).all() and int(tail_p.min()) >= 0 and int(tail_p.max()) < synth.VOCAB
                if group == 1:  # the batch is its own group: nothing ran past its cut
                    assert int(tail_p.abs().max()) == 0 and float(tail_l.abs().max()) == 0.0
        m.synchronize()
        # a mixed stream: is_test and plain forwards interleaved keep their own groups
        a = m(imgs[0], text, is_train=False, is_test=True)
        b = m(imgs[1], text, is_train=False, is_test=False)
        pa, la = a[2]["decode"].result()
        pb, lb = b[2]["decode"].result()
        assert torch.equal(pa, ref[0][0]) and pb.shape[1] == full
        assert torch.equal(pb[:, :steps[1]], ref[1][0])
    m.pipelined, m.decode_group = False, 1


@pytest.mark.parametrize("name", ["t2_greedy", "c2_small_crop", "c2_greedy", "c1_greedy", "s0_greedy"])
def test_fp32_convolutions_keep_parity(cases, name):
    """The exact-fp32 arithmetic mode (conv_precision = 'fp32'; the default is the split-bf16 path every other test
    runs): tokens bit-exact, logits within 1e-3."""
    c = _case(cases, "greedy", name)
    z = np.load(os.path.join(GOLD, name + ".npz"))
    cfg, m = engine_model(c["config"], c["max_seq_len"], c["wseed"], c["end_bias"])
    m.conv_precision = "fp32"
    img = synth.synth_images(c["B"], c["H"], c["W"], seed=c["iseed"]).cuda()
    text = torch.full((c["B"], 1), R.GO, dtype=torch.long, device="cuda")
    with torch.no_grad():
        mem, _, _ = m.forward_encoder(img)
        preds, logits, _ = m(img, text, is_train=False, is_test=c["is_test"])
    torch.cuda.synchronize()
    rows = z["mem_rows"].tolist()
    dmem = float(np.abs(mem[:, rows].cpu().numpy() - z["mem_sample"]).max()) / max(1.0, c["mem_absmax"])
    assert dmem <= 5e-4, dmem
    assert np.array_equal(preds.cpu().numpy(), z["tokens"])
    steps = z["logit_steps"].tolist()
    dl = float(np.abs(logits[:, steps].cpu().numpy() - z["logits_sample"]).max())
    assert dl <= LOGIT_TOL, dl


@pytest.mark.parametrize("cname,H,W,B,precision", [
    ("T2", 48, 64, 1, "fp32"), ("T2", 45, 63, 3, "bf16x3"), ("T2", 35, 61, 2, "fp32"), ("T2", 47, 61, 5, "bf16x3"),
    ("T1", 32, 64, 1, "bf16x3"), ("T1", 45, 99, 3, "fp32"), ("T1", 63, 130, 2, "bf16x3"),
    ("TS0", 48, 64, 3, "bf16x3"), ("T2G", 41, 59, 2, "bf16x3"),
])
def test_odd_crop_shapes_and_batch_sizes(manifests, cname, H, W, B, precision):
    """Ragged geometry: odd heights / widths (every pool and strided conv floors differently), batch sizes that are not
    a multiple of anything, both arithmetic modes -- encoder memory and greedy tokens against the oracle."""
    L = 10
    cfg, m = engine_model(cname, L)
    # max_dimension of the tiny configs is their nominal crop; larger crops need a larger positional table
    ocfg, sd = oracle_state_dict(cname, manifests[cname], L)
    if cfg["SequenceModeling"]["name"] == "ViT" and (H > cfg["max_dimension"][0] or W > cfg["max_dimension"][1]):
        pytest.skip("crop exceeds max_dimension")
    m.conv_precision = precision
    img = synth.synth_images(B, H, W, seed=700 + H + W)
    attn = cfg["Prediction"]["name"] != "TFM"
    text = torch.zeros(B, L + 1, dtype=torch.long) if attn else torch.full((B, 1), R.GO, dtype=torch.long)
    with torch.no_grad():
        omem, oshape, opad = R.forward_encoder(ocfg, sd, img, faithful=False)
        opreds, ologits, _ = R.forward(ocfg, sd, img, text, is_train=False, is_test=False)
        mem, shape, pad = m.forward_encoder(img.cuda())
        preds, logits, _ = m(img.cuda(), text.cuda(), is_train=False, is_test=False)
    torch.cuda.synchronize()
    assert tuple(mem.shape) == tuple(omem.shape)
    assert (tuple(shape) if shape else None) == (tuple(oshape) if oshape else None)
    assert (tuple(pad) if pad else None) == (tuple(opad) if opad else None)
    scale = max(1.0, float(omem.abs().max()))
    assert float((mem.cpu() - omem).abs().max()) / scale <= (5e-4 if precision == "bf16x3" else 1e-4)
    assert torch.equal(preds.cpu(), opreds)
    assert float((logits.cpu() - ologits).abs().max()) <= LOGIT_TOL


@pytest.mark.parametrize("cname,H,W,B", [("T2", 48, 64, 3), ("T2", 45, 63, 2), ("T1", 63, 130, 2), ("C2", 128, 512, 2)])
def test_fused_max_pools_equal_the_pool_kernels(cname, H, W, B):
    """The two 2x2 / stride 2 max-pools of the backbone (resnet.py:94,106) run inside the epilogue of the convolution in
    front of them (pooled-order GEMM rows, ConvP::pool2).  Every pooled VALUE equals the separate pool kernel's
    (Model.conv_fusion = (False, True): d2t_set_conv_fusion): max and the fp32 -> (hi, lo) split are both monotone.  The stored record
    can differ in representation only -- where lo rounds up to the next hi step the pool kernel re-splits hi + lo as
    (hi', 0) while the fused epilogue keeps (hi, lo) -- which moves a consumer's products by the dropped lo*lo term
    (2^-16 relative) and, through the thirty layers behind it, every later value at that level: tokens equal, memory within
    1e-4 relative (a fifth of its bar against the oracle), logits within 5e-4 (the ResNet + TFM-2 stack amplifies a
    2^-16 perturbation most: 5e-4 for the direct split-bf16 form itself there, docs/winograd_study_r03.txt; the
    pooled VALUES are asserted exactly equal at the op level, test_convolution_with_the_max_pool_fused).  Odd crop sizes (a last row / column that
    belongs to no window) included."""
    L = 6
    img = synth.synth_images(B, H, W, seed=4200 + H).cuda()
    text = torch.full((B, 1), R.GO, dtype=torch.long, device="cuda")
    outs = []
    for fused in (True, False):
        cfg, m = engine_model(cname, L)
        m.conv_fusion = (fused, True)
        with torch.no_grad():
            mem, _, _ = m.forward_encoder(img)
            p, l, _ = m(img, text, is_train=False)
        torch.cuda.synchronize()
        outs.append((mem.cpu(), p.cpu(), l.cpu()))
    assert torch.isfinite(outs[0][0]).all()
    scale = max(1.0, float(outs[1][0].abs().max()))
    assert float((outs[0][0] - outs[1][0]).abs().max()) <= 1e-4 * scale
    assert torch.equal(outs[0][1], outs[1][1])
    assert float((outs[0][2] - outs[1][2]).abs().max()) <= 5e-4


@pytest.mark.parametrize("cname,H,W,B", [("T2", 48, 64, 3), ("T1", 63, 130, 2), ("C2", 128, 512, 2)])
def test_shortcut_inside_conv2_equals_the_separate_shortcut_kernel(cname, H, W, B):
    """A BasicBlock's 1x1 shortcut (resnet.py:181-192) runs inside its conv2's launch: K-steps over the block input appended
    behind the filter taps, weights concatenated along K, ONE accumulator (ConvP::in2_hi).  Against the path with its own
    shortcut kernel and a residual add in the epilogue (Model.conv_fusion = (True, False)): the same sums in another order, and no
    rounding of the shortcut to a record in between -- tokens equal, memory within 1e-4 relative, logits within 5e-4."""
    L = 6
    img = synth.synth_images(B, H, W, seed=4300 + H).cuda()
    text = torch.full((B, 1), R.GO, dtype=torch.long, device="cuda")
    outs = []
    for fused in (True, False):
        cfg, m = engine_model(cname, L)
        m.conv_fusion = (True, fused)
        with torch.no_grad():
            mem, _, _ = m.forward_encoder(img)
            p, l, _ = m(img, text, is_train=False)
        torch.cuda.synchronize()
        outs.append((mem.cpu(), p.cpu(), l.cpu()))
    assert torch.isfinite(outs[0][0]).all()
    assert not torch.equal(outs[0][0], outs[1][0])  # (the two paths really differ: otherwise the switch is dead)
    scale = max(1.0, float(outs[1][0].abs().max()))
    # (fp16x2, under D2T_CONV_PRECISION=fp16x2: the unfused path also rounds the shortcut to an fp16 record in between)
    f16 = m.effective_conv_precision() == "fp16x2"
    assert float((outs[0][0] - outs[1][0]).abs().max()) <= (3e-3 if f16 else 1e-4) * scale
    assert torch.equal(outs[0][1], outs[1][1])
    assert float((outs[0][2] - outs[1][2]).abs().max()) <= (1e-3 if f16 else 5e-4)


def test_error_paths_raise_instead_of_crashing():
    """Misuse is reported through status codes / Python exceptions (the library never aborts): a crop larger than the
    positional table, a CPU tensor, a wrong channel count, a missing weight, a second backward without a forward."""
    cfg, m = engine_model("T2", 8)
    text = torch.full((1, 1), R.GO, dtype=torch.long, device="cuda")
    with pytest.raises(RuntimeError):  # 96x128 > max_dimension 48x64: pos_embed too short (the reference fails here too)
        m(synth.synth_images(1, 96, 128).cuda(), text, is_train=False)
    with pytest.raises(RuntimeError):
        m(synth.synth_images(1, 48, 64), text.cpu(), is_train=False)
    with pytest.raises(ValueError):
        m(torch.zeros(1, 3, 48, 64, device="cuda"), text, is_train=False)
    # the engine keeps working after the failures
    preds, logits, _ = m(synth.synth_images(1, 48, 64).cuda(), text, is_train=False)
    assert preds.shape == (1, 9)
    eng = m.engine()
    with pytest.raises(RuntimeError):
        eng.train_backward(torch.zeros(1, 9, synth.VOCAB, device="cuda"))  # no preceding training forward
    with pytest.raises(RuntimeError):
        eng.read_weight("no.such.tensor", torch.zeros(4, device="cuda"))


@pytest.mark.parametrize("name", ["tl0_luong_greedy", "tl0_luong_beam"])
def test_luong_configuration_raises_what_the_reference_raises(cases, name):
    """attn_type 'luong' constructs (checkpoints load: same state_dict keys) and every forward ends in the reference's own
    AttributeError -- Attention.forward_* call attention_cell.reset_mem(), which LuongAttention does not define."""
    c = next(r for r in cases["raises"] if r["case"] == name)
    cfg, m = engine_model(c["config"], 6, beam_size=c["beam_size"])
    img = synth.synth_images(1, c["H"], c["W"], seed=1).cuda()
    with pytest.raises(AttributeError) as e:
        m(img, torch.zeros(1, 7, dtype=torch.long, device="cuda"), is_train=False, is_test=True)
    assert c["type"] == "AttributeError" and str(e.value) == c["message"]


def test_other_attention_cells_vs_oracle_other_seeds(manifests):
    """Bahdanau cell / one-hot targets on inputs the fixtures do not hold: greedy tokens exact, logits within 1e-3; beam
    sequences exact; and the batched beam extension equals the per-sample call."""
    for cname, beam in (("TB0", 4), ("TO0", 3), ("B0", 2)):
        H, W = synth.crop_shape(cname)
        for iseed, eb in ((611, 0.0),) if cname == "B0" else ((612, 0.3),):
            cfg, m = engine_model(cname, 10, 1234, eb)
            ocfg, sd = oracle_state_dict(cname, manifests[cname], 10, 1234, eb)
            img = synth.synth_images(2, H, W, seed=iseed)
            text = torch.zeros(2, 11, dtype=torch.long)
            with torch.no_grad():
                p, l, _ = m(img.cuda(), text.cuda(), is_train=False, is_test=False)
                op, ol, _ = R.forward(ocfg, sd, img, text, is_train=False, is_test=False)
            assert torch.equal(p.cpu(), op), (cname, iseed)
            assert float((l.cpu() - ol).abs().max()) <= LOGIT_TOL
            cfgb, mb = engine_model(cname, 10, 1234, eb, beam_size=beam)
            ocfg["beam_size"] = beam
            with torch.no_grad():
                seq, score, _ = mb(img[:1].cuda(), text[:1].cuda(), is_train=False, is_test=True)
                oseq, oscore, _ = R.forward(ocfg, sd, img[:1], text[:1], is_train=False, is_test=True)
                both = mb.beam_search_batch(img.cuda(), beam_size=beam)
            assert seq[0].tolist() == oseq[0].tolist(), (cname, iseed, eb)
            assert abs(float(score) - oscore) <= 1e-3
            assert both[0][0][0].tolist() == seq[0].tolist() and abs(float(both[0][1]) - float(score)) <= 1e-5


def test_resized_position_table_follows_the_weights(cases, manifests):
    """ViTEncoder (vit_encoder.py:58-95): the engine keeps one bicubic resize of the learned position table per patch grid.
    When the table changes (an optimizer step, load_state_dict) the resized copies must be rebuilt: encoder memory of a 48x64
    crop (grid 1 x 9 under the 3 x 17 table) before and after the table is edited in place, each against the oracle."""
    c = _case(cases, "greedy", "v1_greedy_small")
    cfg, m = engine_model(c["config"], c["max_seq_len"], c["wseed"], c["end_bias"])
    ocfg, sd = oracle_state_dict(c["config"], manifests[c["config"]], c["max_seq_len"], c["wseed"], c["end_bias"])
    img = synth.synth_images(c["B"], c["H"], c["W"], seed=c["iseed"])
    key = "seqmodeler.SequenceModeling.pos_embed"
    tol = MEM_TOL[m.effective_conv_precision()]
    with torch.no_grad():
        mem0, _, _ = m.forward_encoder(img.cuda())
        o0, _, _ = R.forward_encoder(ocfg, sd, img, faithful=False)
        assert float((mem0.cpu() - o0).abs().max()) / max(1.0, float(o0.abs().max())) <= tol
        g = torch.Generator().manual_seed(5)
        delta = torch.randn(sd[key].shape, generator=g) * 0.3
        dict(m.named_parameters())[key].add_(delta.cuda())  # in place: the version counter tells the engine
        sd2 = dict(sd)
        sd2[key] = sd[key] + delta
        mem1, _, _ = m.forward_encoder(img.cuda())
        o1, _, _ = R.forward_encoder(ocfg, sd2, img, faithful=False)
    assert float((o1 - o0).abs().max()) > 1e-2  # the edit shows
    assert float((mem1.cpu() - o1).abs().max()) / max(1.0, float(o1.abs().max())) <= tol


def test_smallest_crop_and_crops_the_model_cannot_take(manifests):
    """Edges of the crop range on the headline stack (HybridViT + TFM-6, max_dimension [128, 512]): the smallest crop the
    reference's min_dimension [32, 32] lets through (backbone output 1 x 9 -> a 1 x 5 patch grid, six memory tokens) against the
    oracle; a crop beyond max_dimension, for which ViTEncoderV3's prefix slice `pos_embed[:, :N+1]` is shorter than the token
    sequence -- the reference raises a RuntimeError on the addition (vit_encoder.py:260), the engine raises one naming the limit;
    and a crop the backbone reduces to nothing (the reference's convolution raises a RuntimeError there too)."""
    L = 6
    cfg, m = engine_model("C2", L)
    ocfg, sd = oracle_state_dict("C2", manifests["C2"], L)
    img = synth.synth_images(2, 32, 32, seed=4711)
    text = torch.full((2, 1), R.GO, dtype=torch.long)
    with torch.no_grad():
        omem, oshape, opad = R.forward_encoder(ocfg, sd, img, faithful=False)
        opreds, ologits, _ = R.forward(ocfg, sd, img, text, is_train=False, is_test=False)
        mem, shape, pad = m.forward_encoder(img.cuda())
        preds, logits, _ = m(img.cuda(), text.cuda(), is_train=False, is_test=False)
    assert tuple(mem.shape) == tuple(omem.shape) == (2, 6, 256)
    assert tuple(shape) == tuple(oshape) == (1, 5) and tuple(pad) == tuple(opad)
    assert float((mem.cpu() - omem).abs().max()) / max(1.0, float(omem.abs().max())) <= MEM_TOL[m.effective_conv_precision()]
    assert torch.equal(preds.cpu(), opreds)
    assert float((logits.cpu() - ologits).abs().max()) <= LOGIT_TOL
    with torch.no_grad():
        with pytest.raises(RuntimeError, match="max_dimension"):
            m.forward_encoder(synth.synth_images(1, 160, 640, seed=1).cuda())
        with pytest.raises(RuntimeError, match="crop"):
            m.forward_encoder(synth.synth_images(1, 8, 8, seed=1).cuda())
        with pytest.raises(RuntimeError):  # the oracle (torch ops, like the reference) cannot take them either
            R.forward_encoder(ocfg, sd, synth.synth_images(1, 160, 640, seed=1), faithful=False)
        with pytest.raises(RuntimeError):
            R.forward_encoder(ocfg, sd, synth.synth_images(1, 8, 8, seed=1), faithful=False)
    # the engine is still usable after the refused calls
    with torch.no_grad():
        mem2, _, _ = m.forward_encoder(img.cuda())
    assert torch.equal(mem2, mem)


@pytest.mark.parametrize("H,W,B,L", [(192, 768, 2, 6), (448, 960, 1, 4), (800, 800, 1, 3)])
def test_crops_beyond_512_memory_tokens(manifests, H, W, B, L):
    """The shipped configurations allow crops up to 448 x 960 (config/test.yaml:3: 14 x 121 patches = 1695 memory tokens) and
    800 x 800; up to round 3 the engine stopped at 512 tokens (the ViT attention kernel held a head's whole K / V in LDS).  The
    HybridViT + TFM stack on such crops against the oracle: 192 x 768 (583 tokens: two query blocks, three key chunks per head)
    448 x 960 itself, and 800 x 800 (config/train.yaml:3: 2526 tokens) -- encoder memory, greedy tokens and logits.  At
    192 x 768 also: TFM beam search, the shipped HybridViT + Attnv2 stack (the LSTM head's alignment rows: 4096 entries now) and
    the d_model-512 decoder over 903 keys (ResNet + TFM-2 on a 128 x 512 crop)."""
    from doc2tex_amd import Model
    cfg = synth.make_config("C2", device="cuda", max_seq_len=L)
    cfg["max_dimension"] = [H, W]
    m = Model(cfg)
    tmpl = {k: v for k, v in m.state_dict().items()}
    m.load_state_dict(synth.synth_state_dict(tmpl), strict=False)
    m = m.cuda().eval()
    ocfg, sd = oracle_state_dict("C2", manifests["C2"], L)
    ocfg["max_dimension"] = [H, W]
    gh, gw = R.vit_max_grid([H, W], (2, 2))
    sd = dict(sd)
    sd["seqmodeler.SequenceModeling.pos_embed"] = R.sincos_2d_table(256, gh, gw)
    assert torch.equal(sd["seqmodeler.SequenceModeling.pos_embed"], m.state_dict()["seqmodeler.SequenceModeling.pos_embed"].cpu())
    img = synth.synth_images(B, H, W, seed=900 + H)
    text = torch.full((B, 1), R.GO, dtype=torch.long)
    with torch.no_grad():
        omem, oshape, opad = R.forward_encoder(ocfg, sd, img, faithful=False)
        opreds, ologits, _ = R.forward(ocfg, sd, img, text, is_train=False, is_test=False)
        mem, shape, pad = m.forward_encoder(img.cuda())
        preds, logits, _ = m(img.cuda(), text.cuda(), is_train=False, is_test=False)
    assert tuple(mem.shape) == tuple(omem.shape) == (B, gh * gw + 1, 256) and mem.shape[1] > 512
    assert tuple(shape) == tuple(oshape) and tuple(pad) == tuple(opad)
    assert float((mem.cpu() - omem).abs().max()) / max(1.0, float(omem.abs().max())) <= MEM_TOL[m.effective_conv_precision()]
    assert torch.equal(preds.cpu(), opreds)
    assert float((logits.cpu() - ologits).abs().max()) <= LOGIT_TOL
    if H == 192:
        # beam search over the same long memory (one sample per call, tfm.py:146-148) against the oracle's
        bcfg = synth.make_config("C2", device="cuda", max_seq_len=L, beam_size=3)
        bcfg["max_dimension"] = [H, W]
        mb = Model(bcfg)
        mb.load_state_dict(synth.synth_state_dict({k: v for k, v in mb.state_dict().items()}, end_bias=1.8), strict=False)
        mb = mb.cuda().eval()
        osd = dict(oracle_state_dict("C2", manifests["C2"], L, end_bias=1.8)[1])
        osd["seqmodeler.SequenceModeling.pos_embed"] = sd["seqmodeler.SequenceModeling.pos_embed"]
        ob = dict(ocfg)
        ob["beam_size"] = 3
        with torch.no_grad():
            seq, score, _ = mb(img[:1].cuda(), text[:1].cuda(), is_train=False)
            oseq, oscore, _ = R.forward(ob, osd, img[:1], text[:1], is_train=False)
        assert seq.tolist() == oseq.tolist() and abs(float(score) - float(oscore)) <= 1e-3
        # HybridViT + Attnv2 (the shipped stack): the LSTM-attention decoder keeps two alignment rows of the memory's length in
        # LDS, 4096 entries since round 4 -- 582 keys against the oracle
        scfg = synth.make_config("S0", device="cuda", max_seq_len=L)
        scfg["max_dimension"] = [H, W]
        ms = Model(scfg)
        ms.load_state_dict(synth.synth_state_dict({k: v for k, v in ms.state_dict().items()}), strict=False)
        ms = ms.cuda().eval()
        socfg, ssd = oracle_state_dict("S0", manifests["S0"], L)
        socfg["max_dimension"] = [H, W]
        ssd = dict(ssd)
        ssd["seqmodeler.SequenceModeling.pos_embed"] = sd["seqmodeler.SequenceModeling.pos_embed"]
        ztext = torch.zeros(B, L + 1, dtype=torch.long)
        with torch.no_grad():
            sp, sl, _ = ms(img.cuda(), ztext.cuda(), is_train=False)
            osp, osl, _ = R.forward(socfg, ssd, img, ztext, is_train=False, is_test=False)
        assert torch.equal(sp.cpu(), osp)
        assert float((sl.cpu() - osl).abs().max()) <= LOGIT_TOL
        # the d_model-512 decoder (ResNet + None + TFM-2, projected K / V): a 128 x 512 crop is 7 x 129 = 903 keys
        c1 = synth.make_config("C1", device="cuda", max_seq_len=L)
        m1 = Model(c1)
        m1.load_state_dict(synth.synth_state_dict({k: v for k, v in m1.state_dict().items()
                                                   if not k.endswith("image_positional_encoder.pe")}), strict=False)
        m1 = m1.cuda().eval()
        o1cfg, o1sd = oracle_state_dict("C1", manifests["C1"], L)
        img1 = synth.synth_images(1, 128, 512, seed=3)
        with torch.no_grad():
            p1, l1, _ = m1(img1.cuda(), text[:1].cuda(), is_train=False)
            op1, ol1, _ = R.forward(o1cfg, o1sd, img1, text[:1], is_train=False, is_test=False)
            mem1, _, _ = m1.forward_encoder(img1.cuda())
        assert mem1.shape[1] == 903
        assert torch.equal(p1.cpu(), op1)
        assert float((l1.cpu() - ol1).abs().max()) <= LOGIT_TOL


@pytest.mark.parametrize("end_bias", [0.3, 0.0])
def test_shipped_test_yaml_geometry(manifests, end_bias):
    """config/test.yaml as shipped: HybridViT + Attnv2 (coverage LSTM head), max_dimension [448, 960] (1695 memory tokens),
    batch_max_length 500, beam_size 5 -- one 448 x 960 crop through Model.forward against the oracle's beam search: the same
    sequence and score, with the [s] bias raised (hypotheses complete after a few steps) and as seeded (all 501 steps run)."""
    from doc2tex_amd import Model
    H, W, L, beam = 448, 960, 500, 5
    cfg = synth.make_config("S0", device="cuda", max_seq_len=L, beam_size=beam)
    cfg["max_dimension"] = [H, W]
    m = Model(cfg)
    m.load_state_dict(synth.synth_state_dict({k: v for k, v in m.state_dict().items()}, end_bias=end_bias), strict=False)
    m = m.cuda().eval()
    ocfg, sd = oracle_state_dict("S0", manifests["S0"], L, end_bias=end_bias)
    ocfg["max_dimension"] = [H, W]
    ocfg["beam_size"] = beam
    sd = dict(sd)
    sd["seqmodeler.SequenceModeling.pos_embed"] = R.sincos_2d_table(256, *R.vit_max_grid([H, W], (2, 2)))
    img = synth.synth_images(1, H, W, seed=77)
    text = torch.zeros(1, L + 1, dtype=torch.long)
    with torch.no_grad():
        seq, score, _ = m(img.cuda(), text.cuda(), is_train=False)
        oseq, oscore, _ = R.forward(ocfg, sd, img, text, is_train=False)
    assert seq.tolist() == oseq.tolist()
    assert (seq.shape[1] == L + 1) == (end_bias == 0.0)
    assert abs(float(score) - float(oscore)) <= _score_tol(m, seq.shape[1])
