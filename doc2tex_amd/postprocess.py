"""Token ids -> LaTeX strings: the step after `Model.forward`.

Mirrors, through the native `d2t_post_*` entry points of libd2t.so (include/d2t_prep.h; host code, no GPU needed):
  TFMLabelConverter.decode / AttnLabelConverter.decode   doc2tex/modules/converter/tfm_converter.py:59-70, attn_converter.py
  the `pred[: pred.find("[s]")]` cut                     doc2tex/engine/inferencing.py:119-121 (api/infer.py:185-188)
  Postprocessing.remove_unused_whitespace                doc2tex/utils/data_utils.py:433-455
  MathRecognition._postprocess                           demo/HybridViT/recog_flow.py:84-105
The reference runs `re` substitutions to a fixed point per formula; here a formula is a few linear scans in C++.
"""
import ctypes as C

import numpy as np

from . import _lib

_MODES = {None: _lib.POST_NONE, False: _lib.POST_NONE, "none": _lib.POST_NONE, True: _lib.POST_API, "api": _lib.POST_API,
          "demo": _lib.POST_DEMO}


def _strip(s, mode):
    raw = s.encode("utf-8")
    if b"\0" in raw:
        raise ValueError("embedded NUL")
    buf = C.create_string_buffer(len(raw) + 1)
    rc = _lib.load().d2t_post_strip_whitespace(raw, mode, buf, len(raw) + 1)
    if rc != _lib.D2T_OK:
        raise RuntimeError(f"d2t_post_strip_whitespace failed (code {rc})")
    return buf.value.decode("utf-8")


class Postprocessing:
    """Same name and static method as the reference class (utils/data_utils.py:433)."""

    @staticmethod
    def remove_unused_whitespace(s: str):
        return _strip(s, _lib.POST_API)


def demo_postprocess(s: str):
    """MathRecognition._postprocess (demo/HybridViT/recog_flow.py:84-105)."""
    return _strip(s, _lib.POST_DEMO)


class LabelDecoder:
    """`character`: the vocabulary WITHOUT the special tokens, as the reference converters take it; `head` "TFM"
    prepends [PAD] [GO] [s] [UNK] (tfm_converter.py:8), "Attn" prepends [GO] [s] [UNK] (attn_converter.py:8)."""

    def __init__(self, character, head="TFM"):
        special = ["[PAD]", "[GO]", "[s]", "[UNK]"] if head == "TFM" else ["[GO]", "[s]", "[UNK]"]
        self.character = special + list(character)
        enc = [t.encode("utf-8") for t in self.character]
        arr = (C.c_char_p * len(enc))(*enc)
        self.lib = _lib.load()
        h = C.c_void_p()
        if self.lib.d2t_vocab_create(arr, len(enc), C.byref(h)) != _lib.D2T_OK:
            raise RuntimeError("d2t_vocab_create failed")
        self.h = h

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.d2t_vocab_destroy(self.h)
            self.h = None

    def _decode(self, text_index, token_level, cut, mode):
        ids = text_index.detach().cpu().numpy() if hasattr(text_index, "detach") else np.asarray(text_index)
        ids = np.ascontiguousarray(ids, dtype=np.int64)
        rows, cols = ids.shape
        sep = b" " if token_level == "word" else b""
        offs = np.zeros(max(rows, 1), np.int64)
        need = C.c_int64(0)
        cap = rows * (cols * 12 + 8)
        for _ in range(2):
            buf = C.create_string_buffer(max(cap, 1))
            rc = self.lib.d2t_post_decode(self.h, ids.ctypes.data_as(C.POINTER(C.c_int64)), rows, cols, sep, int(cut), mode,
                                          buf, cap, offs.ctypes.data_as(C.POINTER(C.c_int64)), C.byref(need))
            if rc != 2:  # D2T_ENOMEM: retry with the size the library asked for
                break
            cap = need.value
        if rc == 1:
            raise IndexError("list index out of range")
        if rc != _lib.D2T_OK:
            raise RuntimeError(f"d2t_post_decode failed (code {rc})")
        raw = buf.raw
        return [raw[offs[r]:raw.index(b"\0", offs[r])].decode("utf-8") for r in range(rows)]

    def decode(self, text_index, token_level="word"):
        """converter.decode(text_index, token_level): every token of every row, joined."""
        return self._decode(text_index, token_level, False, _lib.POST_NONE)

    def detokenize(self, token_ids):
        """converter.detokenize(token_ids): per row the token strings up to (not including) the first "[s]"."""
        ids = token_ids.detach().cpu().numpy() if hasattr(token_ids, "detach") else np.asarray(token_ids)
        out = []
        for row in ids:
            toks = []
            for i in row:
                if self.character[i] == "[s]":
                    break
                toks.append(self.character[i])
            out.append(toks)
        return out

    def to_latex(self, preds_index, token_level="word", postprocess=True):
        """decode -> cut at "[s]" -> whitespace clean-up, in one native call (inferencing.py:93,119-125).
        `postprocess`: True / "api" (remove_unused_whitespace), "demo" (recog_flow._postprocess), False."""
        return self._decode(preds_index, token_level, True, _MODES[postprocess])
