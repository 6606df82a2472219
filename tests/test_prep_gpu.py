"""GPU parity of the pre-processing row: doc2tex_amd.preprocess (libd2t's d2t_prep_run) against
oracle/preprocess.py -- which is pinned on Pillow and on the reference's own minmax_size (tests/test_prep_post_cpu.py).
Integer pixel work: the uint8 image behind the float tensor must be bit-exact, so the float tensors are compared with ==."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import ROOT, engine_model, oracle_state_dict
from doc2tex_amd import synth
from oracle import preprocess as P

pytestmark = pytest.mark.gpu
GOLD = os.path.join(ROOT, "tests", "golden")
META = json.load(open(os.path.join(GOLD, "prep_cases.json")))
ARR = np.load(os.path.join(GOLD, "prep_cases.npz"))


def _opt(maxd, mind=(32, 32), **kw):
    o = {"imgH": None, "imgW": None, "max_dimension": list(maxd), "min_dimension": list(mind), "mean": 0.5, "std": 0.5,
         "rgb": False, "pad": False, "use_resizer": False, "device": "cuda"}
    o.update(kw)
    return o


def _pre(opt, variant):
    from doc2tex_amd.preprocess import Preprocessor
    return Preprocessor(opt, variant)


@pytest.mark.parametrize("case", [c for c in META["cases"] if c["variant"] == "demo"], ids=lambda c: c["name"])
def test_reference_fixture_cases(case, capsys):
    img = synth.synth_formula_image(case["h"], case["w"], case["seed"], zero_border=case["zero_border"], blank=case["blank"])
    opt = _opt(case["max_dimension"], case["min_dimension"])
    got = _pre(opt, "demo")(img).cpu().numpy()
    want = P.resize(img, opt, variant="demo")
    assert got.shape == want.shape and got.dtype == np.float32
    assert np.array_equal(got, want)
    if case["raises"] is None:  # the REFERENCE's minmax_size output, through the normalisation table
        assert np.array_equal(got[0, 0], P.normalize_lut(0.5, 0.5)[ARR[case["name"]]])
    else:  # ValueError inside the reference -> its except branch: max_dimension canvas of ones
        assert got.shape[2:] == tuple(case["max_dimension"]) and "Error:" in capsys.readouterr().out


def test_api_variant_raises_what_the_reference_raises():
    pre = _pre(_opt([128, 512], downsample=None), "api")
    for c in [c for c in META["cases"] if c["variant"] == "api"]:
        img = synth.synth_formula_image(c["h"], c["w"], c["seed"], zero_border=c["zero_border"], blank=c["blank"])
        if c["max_dimension"] != [128, 512]:
            continue
        if c["raises"] == "UnboundLocalError":
            with pytest.raises(UnboundLocalError):
                pre(img)
        else:
            assert np.array_equal(pre(img).cpu().numpy()[0, 0], P.normalize_lut(0.5, 0.5)[ARR[c["name"]]])


@pytest.mark.parametrize("maxd", [(128, 512), (448, 960), (64, 256)])
def test_random_sizes_bit_exact(maxd):
    rng = np.random.default_rng(maxd[0])
    pre = _pre(_opt(maxd), "demo")
    imgs = []
    for i in range(48):
        h = int(rng.integers(1, 4 * maxd[0]))
        w = int(rng.integers(1, 4 * maxd[1]))
        imgs.append(synth.synth_formula_image(h, w, 5000 + i, zero_border=(i % 11 == 0), blank=(0 if i % 13 == 0 else None)))
    tensors, errors = pre.batch(imgs)
    assert all(e is None for e in errors)
    buckets = set()
    for img, t in zip(imgs, tensors):
        want = P.resize(img, _opt(maxd), variant="demo")
        got = t.cpu().numpy()
        assert got.shape == want.shape, (img.shape, got.shape, want.shape)
        assert np.array_equal(got, want), img.shape
        buckets.add(t._base.data_ptr() if t._base is not None else t.data_ptr())
        assert np.array_equal(pre(img).cpu().numpy(), got)  # single-image call == batched call
    assert len(buckets) < len(imgs)  # images of one output size share one [n,1,H,W] batch


def test_api_variant_downsample_matches_oracle():
    """cv2.INTER_AREA is restated, not pinned (cv2 is not in the image): engine == oracle on even (2x2 mean), odd
    (fractional area weights) and factor-3 sizes."""
    for ds in (2, 3):
        opt = _opt((448, 960), downsample=ds)
        pre = _pre(opt, "api")
        n_ok = 0
        for i, (h, w) in enumerate([(200, 600), (201, 601), (199, 600), (90, 301), (64, 64), (63, 65), (40, 500), (333, 777)]):
            img = synth.synth_formula_image(h, w, 6000 + i)
            try:
                want = P.resize(img, opt, variant="api")
            except UnboundLocalError:
                with pytest.raises(UnboundLocalError):
                    pre(img)
                continue
            got = pre(img).cpu().numpy()
            assert got.shape == want.shape and np.array_equal(got, want), (ds, h, w)
            n_ok += 1
        assert n_ok >= 4


def test_very_wide_and_very_tall_pages():
    """Widths beyond 7.6 k pixels take the one-row-per-block horizontal pass (eight rows no longer fit in LDS); a page
    that is only resampled vertically skips the horizontal pass altogether (Pillow does the same)."""
    pre = _pre(_opt((128, 512)), "demo")
    for h, w, seed in ((40, 9000, 7100), (61, 20011, 7101), (3000, 96, 7102), (2500, 512, 7103), (128, 7000, 7104)):
        img = synth.synth_formula_image(h, w, seed)
        want = P.resize(img, _opt((128, 512)), variant="demo")
        got = pre(img).cpu().numpy()
        assert got.shape == want.shape and np.array_equal(got, want), (h, w)


def test_table_arena_survives_many_sizes():
    """More distinct page sizes than one upload holds: tables are appended to the resident arena call after call and every
    result stays exact (a stale or misplaced table would show up as wrong pixels)."""
    pre = _pre(_opt((64, 256)), "demo")
    rng = np.random.default_rng(11)
    for rnd in range(6):
        imgs = [synth.synth_formula_image(int(rng.integers(70, 400)), int(rng.integers(260, 1200)), 7200 + 40 * rnd + i)
                for i in range(24)]
        tensors, errors = pre.batch(imgs)
        for img, t in zip(imgs[::5], tensors[::5]):
            assert np.array_equal(t.cpu().numpy(), P.resize(img, _opt((64, 256)), variant="demo"))
    again, _ = pre.batch(imgs)  # sizes seen before: served from the arena
    for a, b in zip(again, tensors):
        assert torch.equal(a, b)


def _page_with_margins(h, w, seed, mt, ml, inverted=False, rule=False):
    """A formula somewhere on a larger blank page (what `pad: True` is for)."""
    page = np.full((h, w), 255, np.uint8)
    body = synth.synth_formula_image(h - 2 * mt, w - 2 * ml, seed)
    page[mt:h - mt, ml:w - ml] = np.where(body < 200, body, 255)
    if rule:  # a solid black rule as the topmost text row: the crop's first row is all zero -> the paste check fails
        page[mt - 1, ml:w - ml] = 0
    return (255 - page) if inverted else page


@pytest.mark.parametrize("variant", ["demo", "api"])
def test_pad_option(variant, capsys):
    """`pad: True` (data_utils.py:10-45): contrast-normalise, crop to the text, extend to multiples of 32, then minmax_size.
    Engine == oracle for dark-on-light and light-on-dark pages, low-contrast pages, pages that need the LANCZOS resize
    afterwards, the paste failure (-> the except branch) and the blank page (-> the reference's cv2 error)."""
    opt = _opt((128, 512), pad=True, downsample=None)
    pre = _pre(opt, variant)
    pages = [_page_with_margins(200, 700, 7300, 30, 40), _page_with_margins(150, 400, 7301, 20, 33, inverted=True),
             _page_with_margins(400, 2000, 7302, 50, 100), _page_with_margins(90, 310, 7303, 10, 10, rule=True),
             (_page_with_margins(120, 380, 7304, 16, 24) // 2 + 60).astype(np.uint8),
             _page_with_margins(64, 64, 7305, 20, 20), np.full((50, 80), 255, np.uint8), np.full((50, 80), 7, np.uint8)]
    tensors, errors = pre.batch(pages)
    seen = set()
    for img, t, e in zip(pages, tensors, errors):
        try:
            want = P.resize(img, opt, variant=variant)
        except (UnboundLocalError, AssertionError, RuntimeError) as exc:  # what the reference raises for this page
            kind = next(k for k in (UnboundLocalError, AssertionError, RuntimeError) if isinstance(exc, k))
            assert t is None and isinstance(e, kind), (img.shape, exc, e)
            seen.add(kind.__name__)
            continue
        assert e is None, (img.shape, e)
        got = t.cpu().numpy()
        assert got.shape == want.shape and np.array_equal(got, want), img.shape
        seen.add("ok")
    assert "ok" in seen or variant == "api"
    assert "RuntimeError" in seen  # the blank pages


def test_fixed_height_branch():
    """`imgH` set (predict_utils.py:98-114): no resize at all, torchvision's Normalize on the raw 0..255 values."""
    opt = _opt((128, 512), imgH=32, mean=0.5, std=0.5)
    pre = _pre(opt, "api")
    for h, w, seed in ((32, 100, 7400), (77, 301, 7401), (200, 900, 7402)):
        img = synth.synth_formula_image(h, w, seed)
        got = pre(img).cpu().numpy()
        assert got.shape == (1, 1, h, w) and np.array_equal(got, P.resize(img, opt, variant="api"))
    assert float(got.max()) > 400  # not divided by 255, as in the reference


def test_full_size_batch_properties():
    """The bench / serving shape (64 pages of ~360x1600 -> 128x512) through size-independent properties: constant pages stay
    constant under the resampling (its integer coefficients sum to one), outputs lie in the normalised range, a page's
    result does not depend on what else is in the batch, and three sampled pages match the oracle bit for bit."""
    opt = _opt((128, 512))
    pre = _pre(opt, "demo")
    rng = np.random.default_rng(17)
    pages = [synth.synth_formula_image(int(rng.integers(310, 395)), int(rng.integers(1580, 1620)), 7700 + i) for i in range(64)]
    for k, v in ((5, 0), (9, 255), (13, 131)):
        pages[k] = np.full_like(pages[k], v)
    tensors, errors = pre.batch(pages)
    assert all(e is None for e in errors) and all(t.shape == (1, 1, 128, 512) for t in tensors)
    lut = P.normalize_lut(0.5, 0.5)
    for k, v in ((5, 0), (9, 255), (13, 131)):
        assert torch.all(tensors[k] == float(lut[v]))
    allv = torch.cat(tensors)
    assert float(allv.min()) >= -1.0 and float(allv.max()) <= 1.0
    again, _ = pre.batch(pages[40:] + pages[:7])  # other batch composition, other order
    for a, b in zip(again, tensors[40:] + tensors[:7]):
        assert torch.equal(a, b)
    for k in (0, 31, 63):
        assert np.array_equal(tensors[k].cpu().numpy(), P.resize(pages[k], opt, variant="demo"))


def test_table_arena_start_over():
    """A deliberately tiny table arena (D2T_PREP_ARENA_WORDS, read once per process: run in a child process) overflows after
    a few sizes and starts over; results stay exact across the resets."""
    import subprocess
    import sys
    code = r"""
import os, sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
from doc2tex_amd import synth
from doc2tex_amd.preprocess import Preprocessor
from oracle import preprocess as P
opt = {"imgH": None, "imgW": None, "max_dimension": [64, 256], "min_dimension": [32, 32], "mean": 0.5, "std": 0.5,
       "rgb": False, "pad": False, "device": "cuda"}
pre = Preprocessor(opt, "demo")
rng = np.random.default_rng(3)
for rnd in range(10):
    imgs = [synth.synth_formula_image(int(rng.integers(70, 300)), int(rng.integers(260, 900)), 7600 + 10 * rnd + i) for i in range(6)]
    ts, es = pre.batch(imgs)
    for img, t in zip(imgs, ts):
        assert np.array_equal(t.cpu().numpy(), P.resize(img, opt, variant="demo")), (rnd, img.shape)
print("arena ok")
""" % (ROOT, ROOT)
    env = dict(os.environ, D2T_PREP_ARENA_WORDS="60000")  # ~4 pages' worth of tables
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0 and "arena ok" in r.stdout, r.stdout + r.stderr


def test_c_abi_misuse_is_reported_not_executed():
    """A plan that is not what d2t_prep_plan_image produces for the handle's configuration, or an output size that does not
    match, is refused with an error string; the handle keeps working."""
    import ctypes as C
    from doc2tex_amd import _lib
    pre = _pre(_opt((128, 512)), "demo")
    img = synth.synth_formula_image(300, 1400, 7500)
    src, offs = pre._upload([img])
    plan = pre.plan(300, 1400)
    out = torch.empty((1, 1, plan.out_h, plan.out_w), device="cuda")
    call = lambda pl, oh, ow: pre.lib.d2t_prep_run(pre.h, 1, (_lib.D2TPrepPlan * 1)(pl), _lib.ptr(src),
                                                    offs.ctypes.data_as(C.POINTER(C.c_int64)), _lib.ptr(out), oh, ow, None,
                                                    _lib.stream_of(out))
    bad = pre.plan(300, 1400)
    bad.rs_w += 32
    assert call(bad, plan.out_h, plan.out_w) == 1 and b"does not match" in pre.lib.d2t_prep_last_error(pre.h)
    assert call(plan, plan.out_h, plan.out_w + 32) == 1 and b"produces" in pre.lib.d2t_prep_last_error(pre.h)
    assert call(plan, plan.out_h, plan.out_w) == 0
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), P.resize(img, _opt((128, 512)), variant="demo"))


def test_unsupported_options_raise():
    from doc2tex_amd.preprocess import Preprocessor, resize
    with pytest.raises(NotImplementedError):
        Preprocessor(_opt((128, 512), pad=True, downsample=2), "api")
    with pytest.raises(UnboundLocalError):  # the demo copy of resize() has no branch for a set imgH
        Preprocessor(_opt((128, 512), imgH=32), "demo")
    with pytest.raises(NotImplementedError):
        resize(object(), np.zeros((4, 4), np.uint8), _opt((128, 512)))


def test_resize_feeds_the_recognizer():
    """resize() -> Model.forward on the GPU == oracle pre-processing -> oracle forward on the CPU (tokens exact)."""
    from doc2tex_amd.preprocess import resize
    from oracle import restatement as R
    man = json.load(open(os.path.join(GOLD, "manifests.json")))
    L = 10
    cfg, m = engine_model("T2", L)
    opt = _opt(cfg["max_dimension"])
    img = synth.synth_formula_image(3 * cfg["max_dimension"][0] + 5, 3 * cfg["max_dimension"][1] + 11, 7000)
    x = resize(None, img, opt)
    want_x = torch.from_numpy(P.resize(img, opt, variant="demo"))
    assert torch.equal(x.cpu(), want_x)
    text = torch.full((1, 1), R.GO, dtype=torch.long)
    with torch.no_grad():
        preds, logits, _ = m(x, text.cuda(), is_train=False)
        ocfg, sd = oracle_state_dict("T2", man["T2"], L)
        op, ol, _ = R.forward(ocfg, sd, want_x, text)
    assert torch.equal(preds.cpu(), op)
    assert float((logits.cpu() - ol).abs().max()) <= 1e-3


# ---- the rows either side of the recognizer, end to end (SURVEY 8f.1 + 8f.2) --------------------------------------------
def _reference_cleanup(s):
    """Postprocessing.remove_unused_whitespace (utils/data_utils.py:433-455) through Python's re: the checker of the
    native scanner (pinned on the reference's own function by tests/golden/post_cases.json)."""
    import re
    noletter, letter = r"[\W_^\d]", "[a-zA-Z]"
    news = s
    while True:
        s = news
        news = re.sub(r"(?!\\ )(%s)\s+?(%s)" % (noletter, noletter), r"\1\2", s)
        news = re.sub(r"(?!\\ )(%s)\s+?(%s)" % (noletter, letter), r"\1\2", news)
        news = re.sub(r"(%s)\s+?(%s)" % (letter, noletter), r"\1\2", news)
        if news == s:
            return s


def test_pages_to_latex_end_to_end():
    """Page images -> resize() (d2t_prep_run) -> Model.forward (HIP engine) -> LabelDecoder.to_latex (d2t_post_decode)
    against oracle pre-processing -> oracle forward -> the reference's decode / "[s]" cut / regex clean-up in Python:
    identical LaTeX strings for a batch of pages of one size bucket, rows ending at different steps."""
    from doc2tex_amd.postprocess import LabelDecoder
    from doc2tex_amd.preprocess import resize
    from oracle import restatement as R
    man = json.load(open(os.path.join(GOLD, "manifests.json")))
    post = json.load(open(os.path.join(GOLD, "post_cases.json")))
    L = 40
    cfg, m = engine_model("T2", L, end_bias=2.1)  # at this [s] bias the plain pages end at once, the altered ones never do
    ocfg, sd = oracle_state_dict("T2", man["T2"], L, end_bias=2.1)
    opt = _opt(cfg["max_dimension"])
    # a 496-symbol vocabulary made of the fixture's LaTeX tokens (repeats get a numeric suffix, so "\\frac3" etc. also
    # exercise the letter / non-letter boundaries of the clean-up)
    base = post["vocab"]
    vocab = [base[i % len(base)] + (str(i // len(base)) if i >= len(base) else "") for i in range(synth.VOCAB - 4)]
    dec = LabelDecoder(vocab, head="TFM")
    chars = ["[PAD]", "[GO]", "[s]", "[UNK]"] + vocab
    H0, W0 = cfg["max_dimension"]
    pages = [synth.synth_formula_image(2 * H0 + 3 + i, 3 * W0 + 7 * i, 7100 + i) for i in range(5)]
    pages[1] = 255 - pages[1]                                # white strokes on a dark page
    pages[3] = (pages[3] // 4 + 190).astype(np.uint8)        # a faint, low-contrast scan
    xs = [resize(None, pg, opt) for pg in pages]
    assert len({tuple(x.shape) for x in xs}) == 1  # one (H, W) bucket -> one batch (data/collate_fn.py:15-47)
    x = torch.cat(xs)
    want_x = torch.cat([torch.from_numpy(P.resize(pg, opt, variant="demo")) for pg in pages])
    assert torch.equal(x.cpu(), want_x)
    text = torch.full((len(pages), 1), R.GO, dtype=torch.long)
    with torch.no_grad():
        preds, logits, _ = m(x, text.cuda(), is_train=False, is_test=True)
        op, ol, _ = R.forward(ocfg, sd, want_x, text, is_test=True)
    assert torch.equal(preds.cpu(), op)
    got = dec.to_latex(preds, "word", postprocess=True)
    want = []
    for row in op.tolist():
        s = " ".join(chars[i] for i in row)  # TFMLabelConverter.decode, token_level "word" (tfm_converter.py:59-70)
        cut = s.find("[s]")                    # engine/inferencing.py:119-121: pred[:pred.find("[s]")] -- a row that never
        want.append(_reference_cleanup(s[:cut]))  # emitted "[s]" gives find() == -1 and loses its last character (quirk kept)
    assert got == want
    assert "" in want and any(len(w) > 40 for w in want)  # both kinds of row: "[s]" first -> empty string; never ended


def test_post_processing_fixtures_in_the_gpu_suite():
    """The reference-generated post-processing fixtures (419 strings, 80 id matrices; tests/test_prep_post_cpu.py) once
    more in the suite the driver runs on the GPU box, so that row 8f.2 is covered there as well."""
    import test_prep_post_cpu as T
    T.test_whitespace_pass_matches_reference_fixture()
    T.test_decode_cut_and_cleanup_match_reference_fixture()
    T.test_whitespace_pass_matches_python_re_live()
