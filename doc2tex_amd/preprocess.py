"""Image pre-processing on the MI355X: the step before `Model.forward`.

Host-side mirror of the reference's `resize()` -- `doc2tex/utils/predict_utils.py:14-115` (what `api/infer.py:62`
calls with an image path; variant "api") and `demo/HybridViT/helper.py:134-207` (what `demo/HybridViT/recog_flow.py:81`
calls with a PIL image; variant "demo") -- for the configuration the shipped YAMLs use: `imgH: null`, no learned
resizer, grayscale; `pad: True` (contrast-normalise, crop to the text's bounding rectangle, extend to multiples of 32:
`utils/data_utils.py:10-45`) is supported too.  Same argument meaning, same return value (a float32 `[1,1,H,W]` tensor, here on the
GPU), same exceptions.  All pixel work runs in `libd2t.so` (`d2t_prep_run`, include/d2t_prep.h): there is no CPU path.

The two reference copies differ in `get_divisible_size`: the "api" copy (`utils/data_utils.py:48-60`) leaves its result
unassigned when a scaled size is already a multiple of 32 and raises `UnboundLocalError` -- for nearly every image that
needs resizing -- while the "demo" copy (`helper.py:95-107`) is repaired.  Both behaviours are mirrored; "demo" is the
default for arrays / PIL images, "api" for paths, as in the reference.

`Preprocessor.batch()` is the serving extension: many images per call, grouped by output size into `[n,1,H,W]` batches
(the bucketed batches `data/collate_fn.py:15-47` builds), each image bit-identical to its single-image `resize()`.
"""
import ctypes as C
import os
import time

import numpy as np

from . import _lib


def _as_gray_array(img):
    """uint8 [h, w]: what `Image.open(path).convert("L")` / `img.convert("L")` holds (predict_utils.py:16, helper.py:136)."""
    if isinstance(img, np.ndarray):
        if img.dtype != np.uint8 or img.ndim != 2:
            raise TypeError("expected a uint8 [h, w] grayscale array")
        return np.ascontiguousarray(img)
    from PIL import Image
    if isinstance(img, (str, os.PathLike)):
        img = Image.open(img)
    return np.ascontiguousarray(np.asarray(img.convert("L"), dtype=np.uint8))


class Preprocessor:
    def __init__(self, opt, variant="demo", device=None):
        import torch
        assert isinstance(opt, dict)
        assert "imgH" in opt and "imgW" in opt  # predict_utils.py:18-19
        self.fixed_height = opt["imgH"] is not None
        if self.fixed_height and variant == "demo":
            # helper.py:142-207 has no `else:` for a set imgH: `new_img` is never assigned
            raise UnboundLocalError("local variable 'new_img' referenced before assignment (demo/HybridViT/helper.py:206)")
        if opt.get("rgb", False):
            raise NotImplementedError("doc2tex_amd.preprocess: grayscale only (rgb: False in every shipped config)")
        self.pad = bool(opt.get("pad", False))
        if opt.get("use_resizer", False):
            raise NotImplementedError("doc2tex_amd.preprocess: the learned resizer loop is not on this path")
        if variant not in ("demo", "api"):
            raise ValueError("variant must be 'demo' or 'api'")
        ds = opt.get("downsample", None) if variant == "api" else None
        if ds is not None and (int(ds) != ds or ds < 1):
            raise NotImplementedError("doc2tex_amd.preprocess: integer `downsample` ratios only")
        if self.pad and ds:
            raise NotImplementedError("doc2tex_amd.preprocess: `pad: True` together with `downsample` is not built")
        self.variant = variant
        self.lib = _lib.require_device()
        self.device = torch.device(device if device is not None else opt.get("device", "cuda"))
        if self.device.type != "cuda":
            raise RuntimeError("doc2tex_amd.preprocess runs on the GPU only (no CPU path)")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        if self.fixed_height:
            # predict_utils.py:98-114: no downsample, no resize, no padding -- the pixels as they are through torchvision's
            # Normalize(mean, std) on the 0..255 values
            self.pad = False
            self.cfg = _lib.D2TPrepConfig(max_h=1 << 30, max_w=1 << 30, min_h=0, min_w=0, downsample=0, variant=_lib.PREP_API,
                                          mean=float(opt["mean"]), std=float(opt["std"]), norm_mode=_lib.NORM_RAW)
        else:
            self.cfg = _lib.D2TPrepConfig(
                max_h=opt["max_dimension"][0], max_w=opt["max_dimension"][1],
                min_h=opt["min_dimension"][0], min_w=opt["min_dimension"][1],
                downsample=int(ds) if ds else 0, variant=_lib.PREP_API if variant == "api" else _lib.PREP_DEMO,
                mean=float(opt["mean"]), std=float(opt["std"]), norm_mode=_lib.NORM_ALB)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            rc = self.lib.d2t_prep_create(C.byref(self.cfg), C.byref(h))
        self.h = h
        self._slot, self._stage = -1, [None] * 4  # rotating pinned staging blocks (_upload)
        self.timing = None  # set to a dict to accumulate seconds per phase of batch()
        self._check(rc, "d2t_prep_create")

    def _check(self, rc, what):
        if rc != _lib.D2T_OK:
            raise RuntimeError(f"libd2t {what} failed (code {rc}) {self.lib.d2t_prep_last_error(self.h).decode()}")

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.d2t_prep_destroy(self.h)
            self.h = None

    def plan(self, h, w, fallback=False):
        p = _lib.D2TPrepPlan()
        fn = self.lib.d2t_prep_plan_fallback if fallback else self.lib.d2t_prep_plan_image
        if fn(C.byref(self.cfg), int(h), int(w), C.byref(p)) != _lib.D2T_OK:
            raise ValueError(f"bad image size {h}x{w}")
        return p

    def _upload(self, arrays):
        """Pack the pixels of `arrays` into one device buffer: (uint8 tensor, int64 offsets)."""
        import torch
        n = len(arrays)
        offs = np.zeros(n, np.int64)
        total = 0
        for i, a in enumerate(arrays):
            offs[i] = total
            total += (a.size + 15) & ~15
        with torch.cuda.device(self.device):
            # Everything is queued on the caller's stream (a private stream was tried: its kernels cannot get block slots
            # next to the persistent convolution and the host then waits a whole encoder for its staging memory).
            # Pinned staging blocks rotate; a block is reused only after the copy out of it has finished, which with four
            # blocks is a copy queued three batches ago -- the host does not wait behind the encoder in flight.
            slot = self._slot = (self._slot + 1) % len(self._stage)
            stage = self._stage
            t0 = time.perf_counter()
            if stage[slot] is None or stage[slot][0].numel() < total:
                stage[slot] = (torch.empty(max(total, 1 << 20) * 5 // 4, dtype=torch.uint8).pin_memory(), torch.cuda.Event())
            else:
                stage[slot][1].synchronize()
            t1 = time.perf_counter()
            host, copied = stage[slot]
            hv = host.numpy()
            for a, o in zip(arrays, offs):
                hv[o:o + a.size] = a.reshape(-1)
            t2 = time.perf_counter()
            src = host[:total].to(self.device, non_blocking=True)
            copied.record()
            if self.timing is not None:  # where a serving loop's host time goes (tools/serve_bench.py)
                self.timing["stage_wait"] = self.timing.get("stage_wait", 0.0) + (t1 - t0)
                self.timing["pack"] = self.timing.get("pack", 0.0) + (t2 - t1)
                self.timing["h2d_enqueue"] = self.timing.get("h2d_enqueue", 0.0) + (time.perf_counter() - t2)
        return src, offs

    def _run(self, src, offs, plans, out_h, out_w):
        """One d2t_prep_run over device-resident sources -> ([n,1,H,W] float32 on the device, flags tensor or None)."""
        import torch
        n = len(plans)
        offs = np.ascontiguousarray(offs, dtype=np.int64)
        with torch.cuda.device(self.device):
            out = torch.empty((n, 1, out_h, out_w), dtype=torch.float32, device=self.device)
            need_flags = any(p.min_branch for p in plans)
            flags = torch.empty(n, dtype=torch.int32, device=self.device) if need_flags else None
            rc = self.lib.d2t_prep_run(self.h, n, (_lib.D2TPrepPlan * n)(*plans), _lib.ptr(src),
                                       offs.ctypes.data_as(C.POINTER(C.c_int64)), _lib.ptr(out), out_h, out_w,
                                       _lib.ptr(flags), _lib.stream_of(out))
        self._check(rc, "d2t_prep_run")
        return out, flags

    def _pad(self, arrays, src, offs):
        """pad() of every image on the device (data_utils.py:10-45).  -> (padded uint8 buffer, offsets, [(h, w)], errors):
        errors[i] is None, a ValueError (the reference's paste raised: resize() takes its except branch) or the exception
        that propagates out of the reference."""
        import torch
        n = len(arrays)
        I32, I64 = C.POINTER(C.c_int32), C.POINTER(C.c_int64)
        hs = np.array([a.shape[0] for a in arrays], np.int32)
        ws = np.array([a.shape[1] for a in arrays], np.int32)
        st = _lib.stream_of(src)
        common = (self.h, n, _lib.ptr(src), offs.ctypes.data_as(I64), hs.ctypes.data_as(I32), ws.ctypes.data_as(I32))
        with torch.cuda.device(self.device):
            hist = torch.empty((n, 256), dtype=torch.int32, device=self.device)
            self._check(self.lib.d2t_prep_pad_hist(*common, _lib.ptr(hist), st), "d2t_prep_pad_hist")
            hist = hist.cpu().numpy().astype(np.int64)
            # the reference's float64 arithmetic, per pixel VALUE instead of per pixel (data_utils.py:20-29)
            masks = np.zeros((n, 256), np.uint8)
            luts = np.zeros((n, 256), np.uint8)
            errors = [None] * n
            v = np.arange(256, dtype=np.uint8)
            for i in range(n):
                present = np.nonzero(hist[i])[0]
                mn, mx = np.uint8(min(int(present[0]), 255)), np.uint8(255)  # the alpha plane of "LA" is 255 everywhere
                if mx == mn:  # division by zero -> nan -> no text pixels -> cv2.boundingRect(None) fails in the reference
                    errors[i] = RuntimeError("pad(): blank image (cv2.boundingRect of no points)")
                    continue
                f = (v - mn) / (mx - mn) * 255  # uint8 subtraction, float64 division: as numpy evaluates the reference line
                mean = float((hist[i] * f)[present].sum() / hist[i].sum())
                if mean > 128:
                    masks[i], g = f < 128, f
                else:
                    masks[i], g = f > 128, 255 - f
                luts[i, present] = g[present].astype(np.uint8)
            bbox = torch.empty((n, 4), dtype=torch.int32, device=self.device)
            self._check(self.lib.d2t_prep_pad_bbox(*common, masks.ctypes.data_as(C.c_void_p), _lib.ptr(bbox), st),
                        "d2t_prep_pad_bbox")
            bbox = bbox.cpu().numpy()
            rects = np.zeros((n, 4), np.int32)
            dh, dw = np.zeros(n, np.int32), np.zeros(n, np.int32)
            doffs = np.zeros(n, np.int64)
            total = 0
            for i in range(n):
                x0, y0, x1, y1 = (int(t) for t in bbox[i])
                if errors[i] is None and x1 < x0:
                    errors[i] = RuntimeError("pad(): no text pixels (cv2.boundingRect of no points)")
                if errors[i] is not None:
                    rects[i] = (0, 0, 1, 1)
                    dh[i] = dw[i] = 32
                else:
                    w, h = x1 - x0 + 1, y1 - y0 + 1
                    rects[i] = (x0, y0, w, h)
                    dw[i], dh[i] = -(-w // 32) * 32, -(-h // 32) * 32
                doffs[i] = total
                total += (int(dh[i]) * int(dw[i]) + 15) & ~15
            dst = torch.empty(total, dtype=torch.uint8, device=self.device)
            flags = torch.empty(n, dtype=torch.int32, device=self.device)
            background = 255 if self.variant == "demo" else 0  # helper.py:89 / data_utils.py:43
            self._check(self.lib.d2t_prep_pad_apply(*common, rects.ctypes.data_as(I32), luts.ctypes.data_as(C.c_void_p),
                                                    background, _lib.ptr(dst), doffs.ctypes.data_as(I64),
                                                    dh.ctypes.data_as(I32), dw.ctypes.data_as(I32), _lib.ptr(flags), st),
                        "d2t_prep_pad_apply")
            fl = flags.cpu().numpy()
        for i in range(n):
            if errors[i] is None and fl[i] & _lib.PREP_FLAG_PASTE_MISMATCH:
                errors[i] = ValueError("images do not match")  # padded.paste(im, im.getbbox()), data_utils.py:44
        return dst, doffs, [(int(dh[i]), int(dw[i])) for i in range(n)], errors

    def batch(self, images):
        """images: paths / PIL images / uint8 [h,w] arrays -> (tensors, errors): tensors[i] is image i's [1,1,H,W] result
        (a view into the [n,1,H,W] batch of its size bucket; `tensors[i]._base` is the bucket), errors[i] the exception
        instance the reference's resize() raises for image i (tensors[i] is None then)."""
        t_in = time.perf_counter()
        arrays = [_as_gray_array(im) for im in images]
        n = len(arrays)
        tensors, errors = [None] * n, [None] * n
        if self.timing is not None:
            self.timing["to_gray"] = self.timing.get("to_gray", 0.0) + (time.perf_counter() - t_in)
        src, offs = self._upload(arrays)
        t_up = time.perf_counter()
        # per image: which device buffer holds its current source, at which offset and size (pad() replaces the source)
        cur = [(src, int(offs[i]), arrays[i].shape) for i in range(n)]
        fallback = [False] * n
        if self.pad:
            psrc, poffs, psizes, perr = self._pad(arrays, src, offs)
            for i in range(n):
                if perr[i] is None:
                    cur[i] = (psrc, int(poffs[i]), psizes[i])
                elif isinstance(perr[i], ValueError):  # caught by resize()'s `except ValueError` (predict_utils.py:85)
                    print("Error:", perr[i])
                    fallback[i] = True
                else:
                    errors[i] = perr[i]
        plans = [None if errors[i] is not None else self.plan(*cur[i][2], fallback=fallback[i]) for i in range(n)]
        pending = [i for i in range(n) if errors[i] is None]
        for attempt in range(2):
            buckets = {}
            for i in pending:
                p = plans[i]
                if p.status == _lib.PREP_UNBOUND_LOCAL:
                    errors[i] = UnboundLocalError("local variable 'new_h' referenced before assignment "
                                                  "(get_divisible_size, utils/data_utils.py:48-60)")
                elif p.status == _lib.PREP_FALLBACK and self.variant == "api":
                    # predict_utils.py:87-88: the grayscale array is 2-D, the assert on its shape fails
                    errors[i] = AssertionError()
                else:
                    buckets.setdefault((p.out_h, p.out_w, cur[i][0].data_ptr()), []).append(i)
            redo = []
            for (oh, ow, _), idx in buckets.items():
                out, flags = self._run(cur[idx[0]][0], [cur[i][1] for i in idx], [plans[i] for i in idx], oh, ow)
                bad = set()
                if flags is not None:  # only tiny images pasted on the min_dimension canvas get here
                    fl = flags.cpu().numpy()
                    bad = {k for k in range(len(idx)) if fl[k] & _lib.PREP_FLAG_PASTE_MISMATCH}
                for k, i in enumerate(idx):
                    if k in bad:  # `padded_im.paste(img, img.getbbox())` raised ValueError -> the except branch
                        print("Error:", "images do not match")
                        fallback[i] = True
                        redo.append(i)
                    else:
                        tensors[i] = out[k:k + 1]
            for i in redo:  # the except branch works on the ORIGINAL image (predict_utils.py:87)
                cur[i] = (src, int(offs[i]), arrays[i].shape)
                plans[i] = self.plan(*arrays[i].shape, fallback=True)
            pending = redo
            if not pending:
                break
        if self.timing is not None:
            self.timing["plan_and_launch"] = self.timing.get("plan_and_launch", 0.0) + (time.perf_counter() - t_up)
        return tensors, errors

    def __call__(self, img):
        tensors, errors = self.batch([img])
        if errors[0] is not None:
            raise errors[0]
        return tensors[0]


_cache = {}


def resize(resizer, img, opt, variant=None):
    """Drop-in for `resize(resizer, img_path, opt)` (predict_utils.py:14) / `resize(resizer, img, opt)` (helper.py:134)."""
    if resizer is not None and resizer is not False:
        raise NotImplementedError("doc2tex_amd.preprocess: the learned resizer loop is not on this path")
    if variant is None:
        variant = "api" if isinstance(img, (str, os.PathLike)) else "demo"
    key = (variant, opt["imgH"] is not None, bool(opt.get("pad", False)), tuple(opt["max_dimension"]),
           tuple(opt["min_dimension"]), opt.get("downsample", None),
           float(opt["mean"]), float(opt["std"]), str(opt.get("device", "cuda")))
    pre = _cache.get(key)
    if pre is None:
        pre = _cache[key] = Preprocessor(opt, variant)
    return pre(img)
