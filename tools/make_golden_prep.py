#!/usr/bin/env python3
"""Generate tests/golden/prep_cases.npz for the pre-processing row and validate oracle/preprocess.py.

Authoring-container only.  `doc2tex.utils.data_utils` cannot be imported here (its module header imports cv2, which
is not in the image: ModuleNotFoundError; the demo copy imports cv2 and gradio too), but the two functions on this path -- `get_divisible_size` and
`minmax_size` (utils/data_utils.py:48-83) -- use only numpy, math and Pillow.  This script compiles exactly those two
function definitions out of the reference file at generation time (nothing of the file is stored in the repo), runs
them on Pillow images made from seeded synthetic pages, asserts that oracle/preprocess.py agrees, and stores the
REFERENCE's outputs (or the exception type it raised) as the fixture.  Pillow version is recorded in the fixture.

Usage:  PYTHONDONTWRITEBYTECODE=1 python tools/make_golden_prep.py
"""
import ast
import hashlib
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

import numpy as np
import PIL
from PIL import Image

from doc2tex_amd import synth
from oracle import preprocess as P

REF_FILES = {"api": "/root/reference/doc2tex/utils/data_utils.py",      # what api/infer.py:62 reaches
             "demo": "/root/reference/demo/HybridViT/helper.py"}        # what demo/HybridViT/recog_flow.py reaches
GOLD = os.path.join(ROOT, "tests", "golden")


def reference_functions(variant):
    REF_FILE = REF_FILES[variant]
    tree = ast.parse(open(REF_FILE).read())
    keep = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in ("get_divisible_size", "minmax_size")]
    assert len(keep) == 2
    ns = {"np": np, "Image": Image, "math": math}
    exec(compile(ast.Module(body=keep, type_ignores=[]), REF_FILE, "exec"), ns)
    return ns["minmax_size"]


# name, h, w, seed, max_dimension, min_dimension, zero_border, blank
CASES = [
    ("both_too_large", 300, 1400, 4001, [128, 512], [32, 32], False, None),
    ("width_too_large", 100, 1100, 4002, [128, 512], [32, 32], False, None),
    ("height_too_large", 500, 300, 4003, [128, 512], [32, 32], False, None),
    ("fits_untouched", 96, 384, 4004, [128, 512], [32, 32], False, None),
    ("fits_odd_untouched", 77, 301, 4005, [128, 512], [32, 32], False, None),
    ("floor_to_max", 131, 1030, 4006, [128, 512], [32, 32], False, None),
    ("exact_multiple_raises", 256, 1024, 4007, [128, 512], [32, 32], False, None),
    ("too_small_padded", 20, 50, 4008, [128, 512], [32, 32], False, None),
    ("too_small_height_only", 18, 200, 4009, [128, 512], [32, 32], False, None),
    ("too_small_zero_border", 20, 50, 4010, [128, 512], [32, 32], True, None),
    ("too_small_black", 20, 50, 4011, [128, 512], [32, 32], False, 0),
    ("large_then_small", 12, 3000, 4012, [128, 512], [32, 32], False, None),
    ("test_yaml_dims", 700, 2400, 4013, [448, 960], [32, 32], False, None),
    ("c1_dims", 140, 500, 4014, [64, 256], [32, 32], False, None),
    ("width_only_resampled", 128, 2000, 4015, [128, 512], [32, 32], False, None),
]


def main():
    out, meta = {}, {"pillow": PIL.__version__, "numpy": np.__version__, "cases": []}
    for variant, (name, h, w, seed, maxd, mind, zb, blank) in [(v, c) for v in ("demo", "api") for c in CASES]:
        ref_minmax = reference_functions(variant)
        name = f"{variant}_{name}"
        src = synth.synth_formula_image(h, w, seed, zero_border=zb, blank=blank)
        try:
            ref = np.asarray(ref_minmax(Image.fromarray(src, "L"), maxd, mind, True))
            ref_err = None
        except Exception as e:  # the drop-in mirrors the exception type
            ref, ref_err = None, type(e).__name__
        try:
            got = P.minmax_size(src, maxd, mind, variant=variant)
            got_err = None
        except Exception as e:
            got, got_err = None, type(e).__mro__[1].__name__
        assert ref_err == got_err, (name, ref_err, got_err)
        if ref is not None:
            assert ref.shape == got.shape and np.array_equal(ref, got), (name, ref.shape, got.shape)
            out[name] = ref
        meta["cases"].append({"name": name, "variant": variant, "h": h, "w": w, "seed": seed, "max_dimension": maxd, "min_dimension": mind,
                              "zero_border": zb, "blank": blank, "raises": ref_err,
                              "out_shape": None if ref is None else list(ref.shape),
                              "sha256": None if ref is None else hashlib.sha256(ref.tobytes()).hexdigest()})
        print(f"{name:26s} {h}x{w} -> {None if ref is None else ref.shape} {ref_err or ''}")
    np.savez_compressed(os.path.join(GOLD, "prep_cases.npz"), **out)
    with open(os.path.join(GOLD, "prep_cases.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print("wrote", len(out), "arrays,", os.path.getsize(os.path.join(GOLD, "prep_cases.npz")), "bytes")




# ---------------------------------------------------------------------------------------------------------------------
# Post-processing fixtures: the REFERENCE's Postprocessing.remove_unused_whitespace
# (doc2tex/utils/data_utils.py:433-455), MathRecognition._postprocess (demo/HybridViT/recog_flow.py:84-105) and
# TFMLabelConverter.decode (imported normally) on seeded token streams.
# ---------------------------------------------------------------------------------------------------------------------
POST_TOKENS = ["\\frac", "{", "}", "x", "y", "a", "b", "1", "2", "^", "_", "\\mathrm", "\\operatorname", "*", "\\alpha", "+",
               "=", "(", ")", "\\ ", "\\,", "d", "\\hspace", "\\vspace", "ố", "α", "²", "\t", " ", "\\mathbf", "\\left",
               "\\right", ".", "~", "\\\\", "&", "\n", "e", "\\mathit", "\\mathfrak", "\\mathnormal", "\\mathsf", "0.5",
               "c m", "　", "\\sum", "\\int", "|", "[", "]", "s", "[s"]


def reference_postprocessors():
    import re
    from collections import deque
    tree = ast.parse(open("/root/reference/doc2tex/utils/data_utils.py").read())
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "Postprocessing"]
    ns = {"re": re, "deque": deque, "List": list}
    exec(compile(ast.Module(body=cls, type_ignores=[]), "data_utils.py", "exec"), ns)
    tree = ast.parse(open("/root/reference/demo/HybridViT/recog_flow.py").read())
    fn = [m for n in tree.body if isinstance(n, ast.ClassDef) for m in n.body
          if isinstance(m, ast.FunctionDef) and m.name == "_postprocess"]
    ns2 = {"re": re}
    exec(compile(ast.Module(body=fn, type_ignores=[]), "recog_flow.py", "exec"), ns2)
    return ns["Postprocessing"].remove_unused_whitespace, (lambda s: ns2["_postprocess"](None, s))


def post_main():
    import contextlib
    import io
    import random
    import warnings
    warnings.simplefilter("ignore")
    sys.path.insert(0, "/root/reference")
    with contextlib.redirect_stdout(io.StringIO()):
        from doc2tex.modules.converter.tfm_converter import TFMLabelConverter
        from doc2tex.modules.converter.attn_converter import AttnLabelConverter
    api, demo = reference_postprocessors()
    rnd = random.Random(77)
    strings = ["", " ", "x", "\\mathrm { d } x", "\\operatorname * { a r g m a x } _ { x }", "\\mathrm  { a }",
               "\\mathrm \t { a b }", "\\mathrm{ a }", "\\mathbf { a \n b }", "a b c", "1 2 3", "\\ a", "\\  a", "x ^ { 2 }",
               "\\hspace { 1 c m } x \\vspace { 2 p t }", "\\hspace { 1 c m", "\\frac { a } { b } ", "a  b", "a \t\n b"]
    for i in range(400):
        n = rnd.randint(0, 30)
        if i % 3 == 0:
            strings.append("".join(rnd.choice(POST_TOKENS + [" ", " "]) for _ in range(n)))
        else:
            strings.append(" ".join(rnd.choice(POST_TOKENS) for _ in range(n)))
    cases = [{"s": s, "api": api(s), "demo": demo(s)} for s in strings]
    # decode + cut + whitespace pass, exactly as engine/inferencing.py:93,119-125 chains them
    vocab = [t for t in POST_TOKENS if t not in ("[s",)]
    conv = TFMLabelConverter(vocab, "cpu")
    V = len(conv.character)
    rng = np.random.default_rng(78)
    decode = []
    for r in range(40):
        ids = rng.integers(3, V, (3, int(rng.integers(1, 24))))
        if r % 2 == 0:
            ids[0, int(rng.integers(0, ids.shape[1]))] = 2  # [s]
        if r % 5 == 0:
            ids[1, :] = 0
        for level in ("word", "char"):
            full = conv.decode(ids, level)
            cut = [p[: p.find("[s]")] for p in full]
            decode.append({"ids": ids.tolist(), "token_level": level, "decode": full,
                           "latex_api": [api(p) for p in cut], "latex_demo": [demo(p) for p in cut], "latex_none": cut})
    aconv = AttnLabelConverter(vocab, "cpu")
    attn = []
    for r in range(10):
        ids = rng.integers(2, len(aconv.character), (2, int(rng.integers(1, 24))))
        if r % 2 == 0:
            ids[0, int(rng.integers(0, ids.shape[1]))] = 1  # [s]
        full = aconv.decode(ids, "word")
        attn.append({"ids": ids.tolist(), "decode": full, "detokenize": aconv.detokenize(ids),
                     "latex_api": [api(p[: p.find("[s]")]) for p in full]})
    for c in decode:
        c["detokenize"] = conv.detokenize(np.array(c["ids"]))
    with open(os.path.join(GOLD, "post_cases.json"), "w") as f:
        json.dump({"python": sys.version.split()[0], "vocab": vocab, "strings": cases, "decode": decode, "attn": attn}, f,
                  ensure_ascii=True, indent=0)
    print("post:", len(cases), "strings,", len(decode), "decode cases,",
          os.path.getsize(os.path.join(GOLD, "post_cases.json")), "bytes")


if __name__ == "__main__":
    main()
    post_main()
