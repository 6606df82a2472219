"""CPU restatement of doc2tex's inference pre-processing -- TEST INFRASTRUCTURE.

Parity oracle for doc2tex_amd/csrc/prep.hip.  NOT part of the product: only tests/ and tools/ import it.

What it restates (citations relative to /root/reference/doc2tex/):
  resize()              utils/predict_utils.py:14-115   the `imgH is None`, `use_resizer False` branch that
                                                        api/infer.py:62 runs with config/test.yaml
  minmax_size()         utils/data_utils.py:63-83
  get_divisible_size()  utils/data_utils.py:48-60       (variant "api": raises UnboundLocalError whenever a scaled size is
                                                        already a multiple of 32 -- which is nearly every image that
                                                        needs resizing) and /root/reference/demo/HybridViT/helper.py:95-107
                                                        (variant "demo": the repaired copy demo/HybridViT/recog_flow.py uses)
  get_test_transform()  transform/math_transform.py:43-52  ToGray + Normalize + ToTensorV2 on an R=G=B image
and, because the reference delegates the arithmetic to Pillow (`Image.resize(..., Image.LANCZOS)`,
`Image.paste`, `Image.getbbox`), Pillow's published 8-bit resampling algorithm (src/libImaging/Resample.c,
Pillow 12.2.0 is the version in this image; the reference pins none): double-precision windowed-sinc
coefficients normalised per output pixel, converted to 22-bit fixed point, a horizontal pass then a vertical
pass (each only when that dimension changes) with the intermediate rounded to uint8.

Pinning:
  * the resampling / paste / bbox arithmetic is pinned against Pillow itself, live in tests/test_prep_oracle.py
    (Pillow is in the image on both boxes) and through the fixtures tests/golden/prep_*.npz;
  * the control flow (minmax_size, get_divisible_size) is pinned on the REFERENCE's own function bodies:
    tools/make_golden_prep.py compiles those two functions out of the reference files (both variants) and runs
    them on Pillow images; their outputs are the fixtures;
  * NOT pinned (the libraries are absent from the image, `ModuleNotFoundError`): cv2.resize(INTER_AREA) of the
    `downsample` option and albumentations' Normalize.  Both are restated from their documented arithmetic
    (2x2 block mean with round-half-up for exact halving, general area weights otherwise; `(v - mean*255) *
    float32(1/(std*255))` in float32) and the header of each function says "parity unpinned".
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2  # Resample.c


class ReferenceUnboundLocal(UnboundLocalError):
    """get_divisible_size() leaves new_h / new_w unassigned when the size already is a multiple of 32
    (data_utils.py:49-59 only assigns inside `if ori % scale_factor`), so the reference raises here."""


def get_divisible_size(ori_h, ori_w, max_dimension, scale_factor=32, variant="demo"):
    # variant "api":  utils/data_utils.py:48-60 -- new_h / new_w are only assigned inside `if ori % scale_factor`
    # variant "demo": /root/reference/demo/HybridViT/helper.py:95-107 -- the same function with
    #                 `new_h, new_w = ori_h, ori_w` first (the copy demo/HybridViT/recog_flow.py runs)
    new_h, new_w = (ori_h, ori_w) if variant == "demo" else (None, None)
    if ori_h % scale_factor:
        new_h = math.ceil(ori_h / scale_factor) * scale_factor
        if new_h > max_dimension[0]:
            new_h = math.floor(ori_h / scale_factor) * scale_factor
    if ori_w % scale_factor:
        new_w = math.ceil(ori_w / scale_factor) * scale_factor
        if new_w > max_dimension[1]:
            new_w = math.floor(ori_w / scale_factor) * scale_factor
    if new_h is None or new_w is None:
        raise ReferenceUnboundLocal("cannot access local variable 'new_h'/'new_w' (size is a multiple of 32)")
    return int(new_h), int(new_w)


# ---------------------------------------------------------------------------
# Pillow 8-bit resampling (Resample.c: precompute_coeffs, normalize_coeffs_8bpc, ImagingResample*_8bpc)
# ---------------------------------------------------------------------------
def _sinc(x):
    if x == 0.0:
        return 1.0
    x = x * math.pi
    return math.sin(x) / x


def _lanczos(x):
    if -3.0 <= x < 3.0:
        return _sinc(x) * _sinc(x / 3)
    return 0.0


def lanczos_coeffs(in_size, out_size):
    """-> (ksize, bounds int32 [out,2] (xmin, count), kk int32 [out,ksize]) exactly as Resample.c builds them."""
    scale = float(np.float32(in_size) - np.float32(0.0)) / out_size
    filterscale = max(scale, 1.0)
    support = 3.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = [_lanczos((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            k = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + k * (1 << PRECISION_BITS)) if k < 0 else int(0.5 + k * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return ksize, bounds, kk


def _clip8(ss):
    return np.clip(ss >> PRECISION_BITS, 0, 255).astype(np.uint8)


def _resample_axis1(img, out_size):
    """One pass along axis 1 of a uint8 [rows, in_size] array."""
    rows, in_size = img.shape
    _, bounds, kk = lanczos_coeffs(in_size, out_size)
    out = np.empty((rows, out_size), np.uint8)
    src = img.astype(np.int64)
    for xx in range(out_size):
        xmin, n = bounds[xx]
        ss = (src[:, xmin:xmin + n] * kk[xx, :n].astype(np.int64)).sum(axis=1) + (1 << (PRECISION_BITS - 1))
        ss = ((ss + 2 ** 31) % 2 ** 32) - 2 ** 31  # the C accumulator is a 32-bit int
        out[:, xx] = _clip8(ss)
    return out


def lanczos_resize(img, new_w, new_h):
    """PIL.Image.resize((new_w, new_h), Image.LANCZOS) of a mode-"L" image given as uint8 [h, w]."""
    h, w = img.shape
    if (new_w, new_h) == (w, h):
        return img.copy()
    if new_w != w:  # horizontal pass first (ImagingResample), only when the width changes
        img = _resample_axis1(img, new_w)
    if new_h != h:
        img = _resample_axis1(np.ascontiguousarray(img.T), new_h).T
    return np.ascontiguousarray(img)


def getbbox(img):
    """PIL.Image.getbbox(): bounding box (left, upper, right, lower) of the non-zero pixels, None if all zero."""
    ys, xs = np.nonzero(img)
    if len(ys) == 0:
        return None
    return int(xs.min()), int(ys.min()), int(xs.max()) + 1, int(ys.max()) + 1


class ReferencePasteMismatch(ValueError):
    """`padded_im.paste(img, img.getbbox())` (data_utils.py:79) raises ValueError("images do not match") when
    the bounding box of the non-zero pixels is not the whole image (a zero border row / column)."""


def minmax_size(img, max_dimensions=None, min_dimensions=None, variant="demo"):
    # data_utils.py:63-83 == demo/HybridViT/helper.py:110-131, on a uint8 [h, w] array (mode "L", is_gray=True)
    if max_dimensions is not None:
        h, w = img.shape
        ratios = [h / max_dimensions[0], w / max_dimensions[1]]
        if any(r > 1 for r in ratios):
            size = np.array((w, h)) / max(ratios)
            new_h, new_w = get_divisible_size(size[1], size[0], max_dimensions, variant=variant)
            img = lanczos_resize(img, new_w, new_h)
    if min_dimensions is not None:
        h, w = img.shape
        ratios = [h / min_dimensions[0], w / min_dimensions[1]]
        if any(r < 1 for r in ratios):
            new_h, new_w = h / min(ratios), w / min(ratios)
            new_h, new_w = get_divisible_size(new_h, new_w, max_dimensions, variant=variant)
            box = getbbox(img)
            if box is not None and box != (0, 0, w, h):
                raise ReferencePasteMismatch("images do not match")
            if new_h < h or new_w < w:
                # Image.paste of an image larger than the canvas: Pillow raises for a 4-tuple box and crops for
                # None; neither happens for min_dimensions = (32, 32) with sizes rounded up to 32
                raise ReferencePasteMismatch("images do not match")
            padded = np.full((new_h, new_w), 255, np.uint8)
            padded[:h, :w] = img
            img = padded
    return img


def pad(img, divable=32, variant="demo"):
    """pad() -- utils/data_utils.py:10-45 (variant "api": black canvas) == demo/HybridViT/helper.py:52-92 (variant "demo":
    white canvas) on a uint8 [h, w] array.  cv2.findNonZero + cv2.boundingRect are restated as the bounding rectangle of the
    non-zero pixels (PARITY UNPINNED: cv2 is absent); everything else is numpy / Pillow semantics."""
    data = np.stack([img, np.full_like(img, 255)], axis=-1)  # img.convert("LA"): alpha 255
    with np.errstate(divide="ignore", invalid="ignore"):
        data = (data - data.min()) / (data.max() - data.min()) * 255
    if data[..., 0].mean() > 128:
        gray = 255 * (data[..., 0] < 128).astype(np.uint8)  # text = the dark pixels
    else:
        gray = 255 * (data[..., 0] > 128).astype(np.uint8)
        data[..., 0] = 255 - data[..., 0]
    ys, xs = np.nonzero(gray)
    if len(ys) == 0:
        raise RuntimeError("cv2.boundingRect(None): no text pixels")  # cv2.error in the reference
    a, b = int(xs.min()), int(ys.min())
    w, h = int(xs.max()) - a + 1, int(ys.max()) - b + 1
    rect = data[b:b + h, a:a + w]
    if rect[..., -1].var() == 0:
        im = rect[..., 0].astype(np.uint8)
    else:
        im = (255 - rect[..., -1]).astype(np.uint8)
    dw, dh = [divable * (x // divable + (1 if x % divable > 0 else 0)) for x in (w, h)]
    box = getbbox(im)
    if box is not None and box != (0, 0, w, h):
        raise ReferencePasteMismatch("images do not match")
    padded = np.full((dh, dw), 255 if variant == "demo" else 0, np.uint8)
    padded[:h, :w] = im
    return padded


def area_downsample(img, ratio):
    """cv2.resize(img, (int(w/ratio), int(h/ratio)), interpolation=cv2.INTER_AREA) -- PARITY UNPINNED (cv2 absent).

    OpenCV's uint8 INTER_AREA: when both scale factors are the integer 2 the result is (a+b+c+d+2)>>2; for other
    integer factors round-half-even of sum/area (saturate_cast of a float product); otherwise fractional area
    weights in float32, rounded half-even.  predict_utils.py:31-44."""
    h, w = img.shape
    dw, dh = int(w / ratio), int(h / ratio)
    sx, sy = w / dw, h / dh
    if sx == 2 and sy == 2:
        s = img.astype(np.int32)
        return ((s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    if sx == int(sx) and sy == int(sy):
        ix, iy = int(sx), int(sy)
        s = img.astype(np.float32).reshape(dh, iy, dw, ix).sum(axis=(1, 3))
        return np.clip(np.rint(s * np.float32(1.0 / (ix * iy))), 0, 255).astype(np.uint8)

    def tab(ssize, dsize, scale):
        # computeResizeAreaTab (OpenCV imgproc/resize.cpp): per destination index the source taps, in order
        rows = []
        for d in range(dsize):
            f1 = d * scale
            f2 = f1 + scale
            cw = min(scale, ssize - f1)
            s1, s2 = math.ceil(f1), math.floor(f2)
            s2 = min(s2, ssize - 1)
            s1 = min(s1, s2)
            taps = []
            if s1 - f1 > 1e-3:
                taps.append((s1 - 1, np.float32((s1 - f1) / cw)))
            for sx in range(s1, s2):
                taps.append((sx, np.float32(1.0 / cw)))
            if f2 - s2 > 1e-3:
                taps.append((s2, np.float32(min(min(f2 - s2, 1.0), cw) / cw)))
            rows.append(taps)
        return rows
    # ResizeArea_Invoker: buf = sum_x alpha*S (float32, tap order), sum = beta*buf for the first row, += afterwards
    tx, ty = tab(w, dw, sx), tab(h, dh, sy)
    src = img.astype(np.float32)
    hbuf = np.zeros((h, dw), np.float32)
    for d, taps in enumerate(tx):
        acc = np.zeros(h, np.float32)
        for si, a in taps:
            acc = acc + a * src[:, si]
        hbuf[:, d] = acc
    out = np.zeros((dh, dw), np.float32)
    for d, taps in enumerate(ty):
        for j, (si, b) in enumerate(taps):
            out[d] = b * hbuf[si] if j == 0 else out[d] + b * hbuf[si]
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


def normalize_lut(mean, std):
    """albumentations Normalize(mean, std, max_pixel_value=255) on a uint8 value, float32 arithmetic:
    (v - mean*255) * reciprocal(std*255).  PARITY UNPINNED (albumentations absent); math_transform.py:43-52."""
    m = np.float32(mean) * np.float32(255.0)
    d = np.reciprocal(np.float32(std) * np.float32(255.0), dtype=np.float32)
    return ((np.arange(256, dtype=np.float32) - m) * d).astype(np.float32)


def resize(img, opt, variant="demo"):
    """predict_utils.py:14-115 (variant "api") == demo/HybridViT/helper.py:134-200 (variant "demo") for `imgH None`, `use_resizer False`, `pad False`, `rgb False`: uint8 [h, w]
    (what `Image.open(path).convert("L")` holds) -> float32 [1, 1, H, W]."""
    if opt.get("imgH", None) is not None:
        # predict_utils.py:98-114 (variant "api" only): torchvision Normalize(mean, std) on the raw 0..255 values, no resize.
        # PARITY UNPINNED (torchvision absent): tensor.sub_(mean).div_(std) in float32
        if variant == "demo":
            raise UnboundLocalError("new_img")  # helper.py has no else-branch
        x = img.astype(np.float32)
        return ((x - np.float32(opt["mean"])) / np.float32(opt["std"]))[None, None]
    if opt.get("downsample", None) is not None:
        ratio = opt["downsample"]
        h, w = img.shape
        if h / ratio >= opt["min_dimension"][0] and w / ratio >= opt["min_dimension"][1]:
            img = area_downsample(img, ratio)
    lut = normalize_lut(opt["mean"], opt["std"])
    try:
        out = minmax_size(pad(img, variant=variant) if opt.get("pad", False) else img,
                          opt["max_dimension"], opt["min_dimension"], variant=variant)
    except ValueError:
        # predict_utils.py:85-97 / helper.py:193-205: the image as it is at this point, normalised, then
        # F.pad(..., (0, max_w - w, 0, max_h - h), value=1) -- a negative amount crops.  The "api" copy asserts a 3-D
        # array on a grayscale image first, i.e. it raises AssertionError.
        if variant == "api":
            raise AssertionError()
        mh, mw = opt["max_dimension"]
        x = lut[img]
        canvas = np.full((mh, mw), np.float32(1.0), np.float32)
        h, w = min(x.shape[0], mh), min(x.shape[1], mw)
        canvas[:h, :w] = x[:h, :w]
        return canvas[None, None]
    return lut[out][None, None]
