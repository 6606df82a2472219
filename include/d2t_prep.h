/*
 * libd2t, pre- and post-processing entry points: the steps immediately before and after the
 * recognizer forward pass of include/d2t.h.  Same library (libd2t.so), same conventions: plain C, caller-owned
 * buffers, integer status + a last-error string, no CPU fallback for the device work.
 *
 * The reference is Python and binds nothing; each entry point below replaces these reference functions
 * (citations relative to /root/reference/):
 *
 *   d2t_prep_plan_image  <- get_divisible_size / minmax_size size arithmetic
 *                              doc2tex/utils/data_utils.py:48-83        (variant D2T_PREP_API: api/infer.py:62)
 *                              demo/HybridViT/helper.py:95-131          (variant D2T_PREP_DEMO: demo/HybridViT/recog_flow.py:81)
 *                           the `downsample` gate of resize()  doc2tex/utils/predict_utils.py:31-44
 *   d2t_prep_run         <- the pixel work of resize()  doc2tex/utils/predict_utils.py:14-115 == demo/HybridViT/helper.py:134-207
 *                              cv2.resize(INTER_AREA)                   predict_utils.py:38-43
 *                              PIL Image.resize(LANCZOS)                data_utils.py:68
 *                              Image.new("L", size, 255) + paste        data_utils.py:78-80 (and its "images do not match" check)
 *                              ToGray + Normalize + ToTensorV2, [:1]    transform/math_transform.py:43-52, predict_utils.py:52-57
 *                              the `except ValueError` fallback          predict_utils.py:85-97 (F.pad to max_dimension, value 1)
 *                           batched: n images of one output size per call, written as one [n,1,H,W] float32 tensor --
 *                           the layout d2t_encode takes (the bucketed batch of data/collate_fn.py:15-47).
 *   d2t_post_*           <- TFMLabelConverter.decode / detokenize       doc2tex/modules/converter/tfm_converter.py:59-82
 *                           Postprocessing.remove_unused_whitespace     doc2tex/utils/data_utils.py:433-455
 *                           (host-side string work; declared below)
 *
 * Arithmetic: the resampling is Pillow's 8-bit algorithm (double-precision LANCZOS coefficients built on the host exactly
 * as Pillow builds them, 22-bit fixed point, horizontal pass then vertical pass, uint8 intermediate) -- results are
 * bit-identical to Pillow's; the final float is a 256-entry table `(v - mean*255) * float32(1 / (std*255))`.
 */
#ifndef D2T_PREP_H
#define D2T_PREP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct d2t_prep d2t_prep;

enum { D2T_PREP_DEMO = 0, D2T_PREP_API = 1 };
enum { D2T_NORM_ALB = 0, D2T_NORM_RAW = 1 };

/* d2t_prep_plan.status */
enum {
  D2T_PREP_OK = 0,
  D2T_PREP_UNBOUND_LOCAL = 1, /* variant API only: get_divisible_size leaves new_h / new_w unassigned (the reference raises
                                 UnboundLocalError); the Python mirror raises the same                                   */
  D2T_PREP_FALLBACK = 2       /* the reference's `except ValueError` branch: the original image, normalised, padded /
                                 cropped to max_dimension with value 1 (predict_utils.py:85-97)                          */
};

/* d2t_prep_run flags_dev bits (per image) */
enum {
  D2T_PREP_FLAG_PASTE_MISMATCH = 1 /* min_dimension padding was planned, but the bounding box of the non-zero pixels is not
                                      the whole image: `padded_im.paste(img, img.getbbox())` raises ValueError in the
                                      reference, so the caller must re-run this image with status = D2T_PREP_FALLBACK   */
};

typedef struct {
  int32_t max_h, max_w; /* opt["max_dimension"] */
  int32_t min_h, min_w; /* opt["min_dimension"] */
  int32_t downsample;   /* opt["downsample"] (variant API only), 0 = none */
  int32_t variant;      /* D2T_PREP_DEMO / D2T_PREP_API */
  float mean, std;      /* opt["mean"], opt["std"] (grayscale) */
  int32_t norm_mode;    /* D2T_NORM_ALB: (v - mean*255) * float32(1/(std*255))  -- albumentations, the `imgH: null` branch;
                           D2T_NORM_RAW: (v - mean) / std on the 0..255 values  -- torchvision Normalize, the `imgH` branch
                           (predict_utils.py:98-114, which neither resizes nor divides by 255)                          */
} d2t_prep_config;

typedef struct {
  int32_t src_h, src_w; /* the image as decoded (`Image.open(path).convert("L")`) */
  int32_t ds_h, ds_w;   /* after the INTER_AREA downsample (== src when it does not apply) */
  int32_t rs_h, rs_w;   /* after the LANCZOS resize (== ds when the image already fits max_dimension) */
  int32_t out_h, out_w; /* tensor height / width: rs, or the min_dimension canvas, or max_dimension for the fallback */
  int32_t min_branch;   /* 1: pasted on a 255 canvas (data_utils.py:70-81); the paste check applies */
  int32_t status;       /* D2T_PREP_OK / _UNBOUND_LOCAL / _FALLBACK */
} d2t_prep_plan;

/* Host only, no device needed.  Fills every field of *plan from src_h, src_w.  Returns D2T_OK or D2T_EINVAL. */
int d2t_prep_plan_image(const d2t_prep_config* cfg, int src_h, int src_w, d2t_prep_plan* plan);
/* The same plan for status = D2T_PREP_FALLBACK (what the reference computes after the ValueError). */
int d2t_prep_plan_fallback(const d2t_prep_config* cfg, int src_h, int src_w, d2t_prep_plan* plan);

int d2t_prep_create(const d2t_prep_config* cfg, d2t_prep** out);
void d2t_prep_destroy(d2t_prep* p);
const char* d2t_prep_last_error(const d2t_prep* p);

/*
 * Pre-process n images into out_dev[n][1][out_h][out_w] (float32).
 *   plans        [host]   n plans from d2t_prep_plan_image / _fallback; every plan must have status OK or FALLBACK and
 *                         (out_h, out_w) equal to the call's
 *   src_dev      [device] the uint8 pixels of all images, row-major, image i at src_dev + src_offsets[i]
 *   src_offsets  [host]   n byte offsets
 *   flags_dev    [device] n int32, written (D2T_PREP_FLAG_*); may be NULL when no plan has min_branch
 * Asynchronous on `stream`; the host part (coefficient tables) is done before the call returns.
 */
int d2t_prep_run(d2t_prep* p, int n, const d2t_prep_plan* plans, const uint8_t* src_dev, const int64_t* src_offsets,
                 float* out_dev, int out_h, int out_w, int32_t* flags_dev, void* stream);

/*
 * pad() (doc2tex/utils/data_utils.py:10-45 == demo/HybridViT/helper.py:52-92; `pad: True`): contrast-normalise, find the
 * bounding rectangle of the text pixels, crop to it and extend to multiples of 32.  The rectangle -- and with it every size
 * downstream -- depends on the pixels, so the work is split into three device passes with the (tiny, 256-entry) float64
 * table arithmetic between them on the host, where the reference does it:
 *   d2t_prep_pad_hist    per-image histogram of the uint8 pixels                       -> hist_dev [n][256] int32
 *   d2t_prep_pad_bbox    bounding rectangle of the pixels whose value is marked in mask -> bbox_dev [n][4] int32
 *                        (x0, y0, x1, y1 inclusive; x0 > x1 when no pixel is marked: cv2.boundingRect(None) fails)
 *   d2t_prep_pad_apply   dst = zero/`background`-extended crop of lut[src]             -> uint8 images + flags_dev [n]
 *                        (flags bit 0: the crop's non-zero bounding box is not the whole crop, i.e. the reference's
 *                         `padded.paste(im, im.getbbox())` raises ValueError)
 * mask_host / lut_host: [n][256] uint8 on the host.  rects [n][4] = (a, b, w, h) of the crop; dst image i has
 * dst_h[i] x dst_w[i] pixels at dst_dev + dst_offsets[i] and is then an ordinary source image for d2t_prep_run.
 */
int d2t_prep_pad_hist(d2t_prep* p, int n, const uint8_t* src_dev, const int64_t* src_offsets, const int32_t* src_h,
                      const int32_t* src_w, int32_t* hist_dev, void* stream);
int d2t_prep_pad_bbox(d2t_prep* p, int n, const uint8_t* src_dev, const int64_t* src_offsets, const int32_t* src_h,
                      const int32_t* src_w, const uint8_t* mask_host, int32_t* bbox_dev, void* stream);
int d2t_prep_pad_apply(d2t_prep* p, int n, const uint8_t* src_dev, const int64_t* src_offsets, const int32_t* src_h,
                       const int32_t* src_w, const int32_t* rects, const uint8_t* lut_host, int32_t background,
                       uint8_t* dst_dev, const int64_t* dst_offsets, const int32_t* dst_h, const int32_t* dst_w,
                       int32_t* flags_dev, void* stream);

/* Test hook: Pillow's integer LANCZOS coefficient table for one axis (host only).  ksize_out = coefficients per output
 * position; bounds[2*i] = first source index, bounds[2*i+1] = count; kk[i*ksize + k].  kk may be NULL to query ksize. */
int d2t_prep_lanczos_coeffs(int in_size, int out_size, int32_t* ksize_out, int32_t* bounds, int32_t* kk);

/* ---- post-processing (host only) ------------------------------------------------------------------------------ */
typedef struct d2t_vocab d2t_vocab;

/* tokens: n_tokens UTF-8 strings INCLUDING the special tokens at their ids (TFM: [PAD] [GO] [s] [UNK] first,
 * tfm_converter.py:8; Attn: [GO] [s] [UNK], attn_converter.py:8). */
int d2t_vocab_create(const char* const* tokens, int n_tokens, d2t_vocab** out);
void d2t_vocab_destroy(d2t_vocab* v);

enum {
  D2T_POST_NONE = 0,         /* no whitespace pass (config `postprocess: False`)                                      */
  D2T_POST_API = 1,          /* Postprocessing.remove_unused_whitespace   doc2tex/utils/data_utils.py:433-455         */
  D2T_POST_DEMO = 2          /* MathRecognition._postprocess              demo/HybridViT/recog_flow.py:84-105         */
};

/*
 * What engine/inferencing.py:93,119-125 (api/infer.py:137,185-193; demo/HybridViT/recog_flow.py:119-126) does per row:
 *   text = sep.join(vocab[i] for i in ids[row])          sep = " " (token_level "word") or "" ("char")
 *   text = text[: text.find("[s]")]                      when cut_at_end: a STRING search, and -- as in the reference --
 *                                                        a row without "[s]" loses its last character (find() == -1)
 *   text = whitespace pass `mode`
 * ids [rows][cols] int64 (host).  The strings are written back to back, NUL-terminated, into out (capacity out_cap
 * bytes); out_offsets[r] = start of row r.  Returns D2T_OK, D2T_EINVAL (id out of range) or D2T_ENOMEM (out_cap too
 * small; *needed holds the size required).
 */
int d2t_post_decode(const d2t_vocab* v, const int64_t* ids, int rows, int cols, const char* sep, int cut_at_end, int mode,
                    char* out, int64_t out_cap, int64_t* out_offsets, int64_t* needed);

/* The whitespace pass alone on one NUL-terminated UTF-8 string.  The result is never longer than the input: out_cap must
 * be >= strlen(s) + 1.  Returns D2T_EINVAL for malformed UTF-8. */
int d2t_post_strip_whitespace(const char* s, int mode, char* out, int64_t out_cap);

#ifdef __cplusplus
}
#endif
#endif
