// Pre-processing of formula images on the MI355X (include/d2t_prep.h).
//
// Replaces the pixel work of resize() (doc2tex/utils/predict_utils.py:14-115, demo/HybridViT/helper.py:134-207): optional
// cv2 INTER_AREA downsample, Pillow LANCZOS resize to max_dimension, paste on a 255 canvas for min_dimension, normalise,
// and the collate into one [n,1,H,W] float32 batch.  Byte / integer work, HBM-bound: every source byte is read once into
// LDS (horizontal pass) or by coalesced column-parallel loads (vertical pass); the output is written once, coalesced.
//
// Host side: the size arithmetic (double precision, the same IEEE operations the Python reference performs) and Pillow's
// coefficient tables (double-precision windowed sinc through libm's sin, as Pillow computes them; cached per
// (in_size, out_size)).  Device side: integer multiply-accumulate in 32 bits, 22-bit fixed point, uint8 intermediate.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <vector>

#include "../../include/d2t.h"
#include "../../include/d2t_prep.h"

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;  // Pillow Resample.c

// One resampling table for one axis: bounds[2*i] = first source index, bounds[2*i+1] = taps, coef[i*ksize + k] = tap k
// (int32 fixed point for LANCZOS, float bits for INTER_AREA).
struct AxisTab {
  int ksize = 0;
  std::vector<int32_t> bounds, coef;
};

double sinc_filter(double x) {
  if (x == 0.0) return 1.0;
  x = x * M_PI;
  return sin(x) / x;
}
double lanczos_filter(double x) { return (-3.0 <= x && x < 3.0) ? sinc_filter(x) * sinc_filter(x / 3) : 0.0; }

// Pillow precompute_coeffs + normalize_coeffs_8bpc for the box (0, in_size).
void build_lanczos(int in_size, int out_size, AxisTab& t) {
  const float in0 = 0.f, in1 = (float)in_size;
  double scale = (double)(in1 - in0) / out_size, filterscale = scale;
  if (filterscale < 1.0) filterscale = 1.0;
  const double support = 3.0 * filterscale;
  t.ksize = (int)ceil(support) * 2 + 1;
  t.bounds.assign((size_t)out_size * 2, 0);
  t.coef.assign((size_t)out_size * t.ksize, 0);
  std::vector<double> k(t.ksize);
  const double ss = 1.0 / filterscale;
  for (int xx = 0; xx < out_size; ++xx) {
    const double center = in0 + (xx + 0.5) * scale;
    double ww = 0.0;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    for (int x = 0; x < xmax; ++x) {
      const double w = lanczos_filter((x + xmin - center + 0.5) * ss);
      k[x] = w;
      ww += w;
    }
    for (int x = 0; x < xmax; ++x) {
      if (ww != 0.0) k[x] /= ww;
      t.coef[(size_t)xx * t.ksize + x] = k[x] < 0 ? (int)(-0.5 + k[x] * (1 << PRECISION_BITS))
                                                  : (int)(0.5 + k[x] * (1 << PRECISION_BITS));
    }
    t.bounds[2 * xx] = xmin;
    t.bounds[2 * xx + 1] = xmax;
  }
}

// OpenCV computeResizeAreaTab (imgproc/resize.cpp) regrouped per destination index.
void build_area(int ssize, int dsize, AxisTab& t) {
  const double scale = (double)ssize / dsize;
  t.ksize = (int)ceil(scale) + 2;
  t.bounds.assign((size_t)dsize * 2, 0);
  t.coef.assign((size_t)dsize * t.ksize, 0);
  auto put = [&](int d, int& n, int& first, int si, float a) {
    if (n == 0) first = si;
    int32_t bits;
    memcpy(&bits, &a, 4);
    t.coef[(size_t)d * t.ksize + n++] = bits;
  };
  for (int dx = 0; dx < dsize; ++dx) {
    const double fsx1 = dx * scale, fsx2 = fsx1 + scale, cell = std::min(scale, ssize - fsx1);
    int sx1 = (int)ceil(fsx1), sx2 = (int)floor(fsx2);
    sx2 = std::min(sx2, ssize - 1);
    sx1 = std::min(sx1, sx2);
    int n = 0, first = sx1;
    if (sx1 - fsx1 > 1e-3) put(dx, n, first, sx1 - 1, (float)((sx1 - fsx1) / cell));
    for (int sx = sx1; sx < sx2; ++sx) put(dx, n, first, sx, float(1.0 / cell));
    if (fsx2 - sx2 > 1e-3) put(dx, n, first, sx2, (float)(std::min(std::min(fsx2 - sx2, 1.), cell) / cell));
    t.bounds[2 * dx] = first;
    t.bounds[2 * dx + 1] = n;
  }
}

// ---- size arithmetic (doubles, as Python computes them) ----------------------------------------------------------
// get_divisible_size: returns false where the reference leaves a result unassigned (variant API).
bool divisible_size(double ori_h, double ori_w, int max_h, int max_w, int variant, int64_t* nh, int64_t* nw) {
  const double sf = 32.0;
  bool has_h = variant == D2T_PREP_DEMO, has_w = has_h;
  double new_h = ori_h, new_w = ori_w;
  if (fmod(ori_h, sf) != 0.0) {
    new_h = ceil(ori_h / sf) * sf;
    if (new_h > max_h) new_h = floor(ori_h / sf) * sf;
    has_h = true;
  }
  if (fmod(ori_w, sf) != 0.0) {
    new_w = ceil(ori_w / sf) * sf;
    if (new_w > max_w) new_w = floor(ori_w / sf) * sf;
    has_w = true;
  }
  if (!has_h || !has_w) return false;
  *nh = (int64_t)new_h;
  *nw = (int64_t)new_w;
  return true;
}

void plan_fallback(const d2t_prep_config* c, d2t_prep_plan* p) {
  p->rs_h = p->ds_h;  // predict_utils.py:87 normalises `img` as it is at that point (after the downsample)
  p->rs_w = p->ds_w;
  p->out_h = c->max_h;
  p->out_w = c->max_w;
  p->min_branch = 0;
  p->status = D2T_PREP_FALLBACK;
}

int plan_image(const d2t_prep_config* c, int src_h, int src_w, d2t_prep_plan* p) {
  if (!c || !p || src_h <= 0 || src_w <= 0 || c->max_h <= 0 || c->max_w <= 0 || c->downsample < 0) return D2T_EINVAL;
  memset(p, 0, sizeof(*p));
  p->src_h = p->ds_h = src_h;
  p->src_w = p->ds_w = src_w;
  if (c->variant == D2T_PREP_API && c->downsample > 0) {  // predict_utils.py:31-44
    const double r = c->downsample;
    if (src_h / r >= c->min_h && src_w / r >= c->min_w) {
      p->ds_h = (int)(src_h / r);
      p->ds_w = (int)(src_w / r);
    }
  }
  int64_t h = p->ds_h, w = p->ds_w;
  {  // data_utils.py:64-68
    const double rh = (double)h / c->max_h, rw = (double)w / c->max_w;
    if (rh > 1 || rw > 1) {
      const double m = std::max(rh, rw);
      int64_t nh, nw;
      if (!divisible_size((double)h / m, (double)w / m, c->max_h, c->max_w, c->variant, &nh, &nw)) {
        p->status = D2T_PREP_UNBOUND_LOCAL;
        return D2T_OK;
      }
      if (nh <= 0 || nw <= 0) {  // Image.resize raises ValueError("height and width must be > 0")
        plan_fallback(c, p);
        return D2T_OK;
      }
      h = nh;
      w = nw;
    }
  }
  p->rs_h = p->out_h = (int)h;
  p->rs_w = p->out_w = (int)w;
  if (c->min_h > 0 && c->min_w > 0) {  // data_utils.py:70-81
    const double rh = (double)h / c->min_h, rw = (double)w / c->min_w;
    if (rh < 1 || rw < 1) {
      const double m = std::min(rh, rw);
      int64_t nh, nw;
      if (!divisible_size((double)h / m, (double)w / m, c->max_h, c->max_w, c->variant, &nh, &nw)) {
        p->status = D2T_PREP_UNBOUND_LOCAL;
        return D2T_OK;
      }
      if (nh < h || nw < w || nh <= 0 || nw <= 0) {  // canvas smaller than the image: paste raises ValueError
        plan_fallback(c, p);
        return D2T_OK;
      }
      p->out_h = (int)nh;
      p->out_w = (int)nw;
      p->min_branch = 1;
    }
  }
  return D2T_OK;
}

// albumentations Normalize(mean, std, max_pixel_value=255) in float32 (math_transform.py:43-52), or torchvision's
// Normalize on the raw 0..255 values (predict_utils.py:110): tensor.sub_(mean).div_(std)
float normalise_value(const d2t_prep_config& c, float v) {
  if (c.norm_mode == D2T_NORM_RAW) {
    volatile float t = v - c.mean;
    return t / c.std;
  }
  const float m = c.mean * 255.0f;
  volatile float sd = c.std * 255.0f;
  const float dnm = 1.0f / sd;
  volatile float t = v - m;
  return t * dnm;
}

// ---- device side -------------------------------------------------------------------------------------------------
struct PrepDesc {
  int64_t src_off, ds_off, hp_off;  // byte offsets: source pixels; work buffer (downsampled image, horizontal-pass image)
  int32_t src_h, src_w, ds_h, ds_w, rs_h, rs_w;
  int32_t do_ds, do_h, do_v;  // do_ds: 1 = 2x2 mean, 2 = integer factor, 3 = fractional area tables
  int32_t ds_fx, ds_fy;       // integer factors (do_ds == 2)
  int32_t axb, axk, axs, ayb, ayk, ays;  // area tables: bounds / coefficient word offsets, ksize
  int32_t hb, hk, hks, vb, vk, vks;      // LANCZOS tables; the horizontal coefficients are stored transposed [k][out]
  int32_t min_branch, fallback;
  float fill;
};

__device__ __forceinline__ uint8_t clip8(int ss) {
  int v = ss >> PRECISION_BITS;
  return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// cv2.resize(..., INTER_AREA): one thread per destination pixel (the footprint is a few source pixels).
__global__ __launch_bounds__(256) void prep_area_kernel(const PrepDesc* __restrict__ descs, const uint8_t* __restrict__ src,
                                                        uint8_t* __restrict__ work, const int32_t* __restrict__ tabs) {
  // separate float32 multiplies and adds, as in the CPU code this restates: HIP's default would contract them into fused
  // multiply-adds (the __fmul_rn / __fadd_rn wrappers of this ROCm inline to contractable operators too), which moves results that sit on a rounding tie
#pragma clang fp contract(off)
  const PrepDesc d = descs[blockIdx.z];
  if (!d.do_ds) return;
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= d.ds_w || y >= d.ds_h) return;
  const uint8_t* S = src + d.src_off;
  uint8_t* D = work + d.ds_off;
  const int sw = d.src_w;
  if (d.do_ds == 1) {
    const uint8_t* r0 = S + (size_t)(2 * y) * sw + 2 * x;
    D[(size_t)y * d.ds_w + x] = (uint8_t)((r0[0] + r0[1] + r0[sw] + r0[sw + 1] + 2) >> 2);
    return;
  }
  float v;
  if (d.do_ds == 2) {
    int sum = 0;
    for (int ky = 0; ky < d.ds_fy; ++ky)
      for (int kx = 0; kx < d.ds_fx; ++kx) sum += S[(size_t)(y * d.ds_fy + ky) * sw + x * d.ds_fx + kx];
    v = (float)sum * (1.f / (float)(d.ds_fx * d.ds_fy));
  } else {
    const int x0 = tabs[d.axb + 2 * x], nx = tabs[d.axb + 2 * x + 1];
    const int y0 = tabs[d.ayb + 2 * y], ny = tabs[d.ayb + 2 * y + 1];
    const float* ax = reinterpret_cast<const float*>(tabs + d.axk) + (size_t)x * d.axs;
    const float* ay = reinterpret_cast<const float*>(tabs + d.ayk) + (size_t)y * d.ays;
    v = 0.f;
    for (int ky = 0; ky < ny; ++ky) {
      float buf = 0.f;
      for (int kx = 0; kx < nx; ++kx) buf = buf + ax[kx] * (float)S[(size_t)(y0 + ky) * sw + x0 + kx];
      v = ky == 0 ? ay[0] * buf : v + ay[ky] * buf;
    }
  }
  int r = (int)rintf(v);  // saturate_cast<uchar>(float): round half to even, clamp
  D[(size_t)y * d.ds_w + x] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
}

// Horizontal LANCZOS pass: one block per R consecutive source rows, staged in LDS once (4-byte loads over the aligned
// middle of each row).  A thread owns output columns and keeps R accumulators, so a coefficient (4 bytes, from L2) is
// fetched once per R pixels (1 byte each, from LDS) -- with one row per block the table traffic was 4x the pixel traffic.
template <int R>
__global__ __launch_bounds__(256) void prep_hpass_kernel(const PrepDesc* __restrict__ descs, const uint8_t* __restrict__ src,
                                                         uint8_t* __restrict__ work, const int32_t* __restrict__ tabs,
                                                         int lds_stride) {
  extern __shared__ uint8_t rows[];
  const PrepDesc d = descs[blockIdx.y];
  const int y0 = blockIdx.x * R;
  if (!d.do_h || y0 >= d.ds_h) return;
  const int w = d.ds_w;
  const uint8_t* base = d.do_ds ? work + d.ds_off : src + d.src_off;
  int pads[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    pads[r] = 0;
    if (y0 + r >= d.ds_h) continue;
    const uint8_t* in = base + (size_t)(y0 + r) * w;
    uint8_t* row = rows + (size_t)r * lds_stride;
    const int head = min(w, (int)((4 - (reinterpret_cast<uintptr_t>(in) & 3)) & 3));
    const int words = (w - head) >> 2;
    const int pad = (4 - head) & 3;  // LDS byte pad + i holds source byte i; pad + head is 4-byte aligned
    pads[r] = pad;
    for (int i = threadIdx.x; i < head; i += 256) row[pad + i] = in[i];
    const uint32_t* in4 = reinterpret_cast<const uint32_t*>(in + head);
    uint32_t* row4 = reinterpret_cast<uint32_t*>(row + pad + head);
    for (int i = threadIdx.x; i < words; i += 256) row4[i] = in4[i];
    for (int i = head + 4 * words + threadIdx.x; i < w; i += 256) row[pad + i] = in[i];
  }
  __syncthreads();
  const int ow = d.rs_w;
  const int32_t* hb = tabs + d.hb;
  const int32_t* hk = tabs + d.hk;
  for (int xx = threadIdx.x; xx < ow; xx += 256) {
    const int xmin = hb[2 * xx], cnt = hb[2 * xx + 1];
    int ss[R];
#pragma unroll
    for (int r = 0; r < R; ++r) ss[r] = 1 << (PRECISION_BITS - 1);
    for (int k = 0; k < cnt; ++k) {
      const int c = hk[(size_t)k * ow + xx];
#pragma unroll
      for (int r = 0; r < R; ++r) ss[r] += (int)rows[(size_t)r * lds_stride + pads[r] + xmin + k] * c;
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
      if (y0 + r < d.ds_h) work[d.hp_off + (size_t)(y0 + r) * ow + xx] = clip8(ss[r]);
  }
}

// Vertical LANCZOS pass (when the height changes) + canvas / padding + normalisation table + paste check.  One block per
// FR consecutive output rows x 256 columns: the intermediate rows those FR outputs need (about taps + (FR-1) x scale of
// them) are staged in LDS once -- one block per output row re-read every intermediate row ~taps times, 4.4x its size in L2
// misses (rocprofv3 FETCH_SIZE) because neighbouring rows ran on different XCDs.  The row index is block-uniform within an
// iteration, so the coefficients are scalar loads.  Strips whose rows do not fit the LDS window read global memory directly.
constexpr int FR = 8, FIN_ROWS = 96;  // output rows per block; LDS window (rows x 256 bytes = 24 KB)
__global__ __launch_bounds__(256) void prep_finish_kernel(const PrepDesc* __restrict__ descs, const uint8_t* __restrict__ src,
                                                          const uint8_t* __restrict__ work, const int32_t* __restrict__ tabs,
                                                          const float* __restrict__ lut, float* __restrict__ out, int out_h,
                                                          int out_w, int32_t* __restrict__ bits) {
  __shared__ uint8_t win[FIN_ROWS][256];
  const PrepDesc d = descs[blockIdx.z];
  const int y0 = blockIdx.y * FR, x = blockIdx.x * 256 + threadIdx.x;
  const uint8_t* in = d.do_h ? work + d.hp_off : (d.do_ds ? work + d.ds_off : src + d.src_off);
  const int iw = d.rs_w;  // width of `in` (the horizontal pass, when there is one, has already produced rs_w columns)
  // input rows [lo, hi) needed by the resampled output rows of this strip (block-uniform)
  int lo = 0, hi = 0;
  bool windowed = false;
  if (d.do_v && y0 < d.rs_h) {
    const int yl = min(y0 + FR, d.rs_h) - 1;
    lo = tabs[d.vb + 2 * y0];
    hi = tabs[d.vb + 2 * yl] + tabs[d.vb + 2 * yl + 1];
    windowed = hi - lo <= FIN_ROWS;
    if (windowed) {
      if (x < iw)
        for (int rr = lo; rr < hi; ++rr) win[rr - lo][threadIdx.x] = in[(size_t)rr * iw + x];
    }
  }
  // each thread reads back only the column it wrote: no barrier needed
  int mybits = 0;
  for (int y = y0; y < min(y0 + FR, out_h); ++y) {
    float o = d.fill;
    if (x < out_w && y < d.rs_h && x < d.rs_w) {
      int v;
      if (d.do_v) {
        const int ymin = tabs[d.vb + 2 * y], cnt = tabs[d.vb + 2 * y + 1];
        const int32_t* vk = tabs + d.vk + (size_t)y * d.vks;
        int ss = 1 << (PRECISION_BITS - 1);
        if (windowed)
          for (int k = 0; k < cnt; ++k) ss += (int)win[ymin + k - lo][threadIdx.x] * vk[k];
        else
          for (int k = 0; k < cnt; ++k) ss += (int)in[(size_t)(ymin + k) * iw + x] * vk[k];
        v = clip8(ss);
      } else {
        v = in[(size_t)y * iw + x];
      }
      o = lut[v];  // the fallback normalises with the same transform (predict_utils.py:89)
      if (d.min_branch && v) mybits |= 1 | (y == 0 ? 2 : 0) | (y == d.rs_h - 1 ? 4 : 0) | (x == 0 ? 8 : 0) | (x == d.rs_w - 1 ? 16 : 0);
    }
    if (x < out_w) out[((size_t)blockIdx.z * out_h + y) * out_w + x] = o;
  }
  if (d.min_branch) {
    for (int s = 32; s; s >>= 1) mybits |= __shfl_xor(mybits, s);
    if ((threadIdx.x & 63) == 0 && mybits) atomicOr(&bits[blockIdx.z], mybits);
  }
}

__global__ void prep_flags_kernel(const PrepDesc* __restrict__ descs, const int32_t* __restrict__ bits, int32_t* __restrict__ flags,
                                  int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int b = bits[i];
  // getbbox() is None (all zero): paste(img, None) succeeds; otherwise the box must be the whole image
  flags[i] = (descs[i].min_branch && (b & 1) && (b & 30) != 30) ? D2T_PREP_FLAG_PASTE_MISMATCH : 0;
}


// ---- pad() passes (data_utils.py:10-45) ------------------------------------------------------------------------------
struct PadDesc {
  int64_t src_off, dst_off;
  int32_t h, w;              // source
  int32_t a, b, cw, ch;      // crop rectangle
  int32_t dh, dw;            // destination (multiples of 32)
  int32_t background;
};

__global__ __launch_bounds__(256) void pad_hist_kernel(const PadDesc* __restrict__ descs, const uint8_t* __restrict__ src,
                                                       int32_t* __restrict__ hist) {
  __shared__ int32_t bins[256];
  const PadDesc d = descs[blockIdx.y];
  bins[threadIdx.x] = 0;
  __syncthreads();
  const size_t npx = (size_t)d.h * d.w;
  const uint8_t* S = src + d.src_off;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < npx; i += (size_t)gridDim.x * 256) atomicAdd(&bins[S[i]], 1);
  __syncthreads();
  if (bins[threadIdx.x]) atomicAdd(&hist[blockIdx.y * 256 + threadIdx.x], bins[threadIdx.x]);
}

__global__ __launch_bounds__(256) void pad_bbox_kernel(const PadDesc* __restrict__ descs, const uint8_t* __restrict__ src,
                                                       const uint8_t* __restrict__ masks, int32_t* __restrict__ bbox) {
  __shared__ uint8_t mk[256];
  __shared__ int32_t red[4];
  const PadDesc d = descs[blockIdx.y];
  mk[threadIdx.x] = masks[blockIdx.y * 256 + threadIdx.x];
  if (threadIdx.x < 4) red[threadIdx.x] = threadIdx.x < 2 ? 0x7fffffff : -1;
  __syncthreads();
  const uint8_t* S = src + d.src_off;
  int x0 = 0x7fffffff, y0 = 0x7fffffff, x1 = -1, y1 = -1;
  for (int y = blockIdx.x; y < d.h; y += gridDim.x)
    for (int x = threadIdx.x; x < d.w; x += 256)
      if (mk[S[(size_t)y * d.w + x]]) {
        x0 = min(x0, x), x1 = max(x1, x), y0 = min(y0, y), y1 = max(y1, y);
      }
  if (x1 >= 0) {
    atomicMin(&red[0], x0), atomicMin(&red[1], y0), atomicMax(&red[2], x1), atomicMax(&red[3], y1);
  }
  __syncthreads();
  if (threadIdx.x == 0 && red[2] >= 0) {
    int32_t* o = bbox + blockIdx.y * 4;
    atomicMin(&o[0], red[0]), atomicMin(&o[1], red[1]), atomicMax(&o[2], red[2]), atomicMax(&o[3], red[3]);
  }
}

__global__ __launch_bounds__(256) void pad_apply_kernel(const PadDesc* __restrict__ descs, const uint8_t* __restrict__ src,
                                                        const uint8_t* __restrict__ luts, uint8_t* __restrict__ dst,
                                                        int32_t* __restrict__ bits) {
  __shared__ uint8_t lut[256];
  const PadDesc d = descs[blockIdx.z];
  lut[threadIdx.x] = luts[blockIdx.z * 256 + threadIdx.x];
  __syncthreads();
  const int y = blockIdx.y, x = blockIdx.x * 256 + threadIdx.x;
  int mybits = 0;
  if (y < d.dh && x < d.dw) {
    int v = d.background;
    if (y < d.ch && x < d.cw) {
      v = lut[src[d.src_off + (size_t)(d.b + y) * d.w + d.a + x]];
      if (v) mybits = 1 | (y == 0 ? 2 : 0) | (y == d.ch - 1 ? 4 : 0) | (x == 0 ? 8 : 0) | (x == d.cw - 1 ? 16 : 0);
    }
    dst[d.dst_off + (size_t)y * d.dw + x] = (uint8_t)v;
  }
  for (int s = 32; s; s >>= 1) mybits |= __shfl_xor(mybits, s);
  if ((threadIdx.x & 63) == 0 && mybits) atomicOr(&bits[blockIdx.z], mybits);
}

__global__ void pad_flags_kernel(const int32_t* __restrict__ bits, int32_t* __restrict__ flags, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) flags[i] = ((bits[i] & 1) && (bits[i] & 30) != 30) ? D2T_PREP_FLAG_PASTE_MISMATCH : 0;
}

}  // namespace

struct TabKey {
  int in, out, kind;
  bool operator<(const TabKey& o) const { return in != o.in ? in < o.in : (out != o.out ? out < o.out : kind < o.kind); }
  bool operator==(const TabKey& o) const { return in == o.in && out == o.out && kind == o.kind; }
};
struct DevTab {
  int32_t b, k, ks;  // word offsets of bounds / coefficients in the arena, coefficients per output position
};

struct d2t_prep {
  d2t_prep_config cfg;
  std::string err;
  float* lut = nullptr;  // [256] normalisation table
  std::map<TabKey, DevTab> dev_tabs;
  int32_t* d_arena = nullptr;  // resident resampling tables
  size_t arena_cap = 0, arena_used = 0;  // in int32 words
  // per-call staging (descriptors + new tables): pinned host block + device mirror, NSTAGE sets used in rotation so that
  // the host waits for a copy queued NSTAGE calls ago, not for the one just behind the work in flight
  static constexpr int NSTAGE = 4;
  struct Stage {
    char* h = nullptr;
    char* d = nullptr;
    size_t cap = 0;
    hipEvent_t done = nullptr;
    bool pending = false;
  } stage[NSTAGE];
  uint64_t calls = 0;
  uint8_t* d_work = nullptr;
  size_t work_cap = 0;
  int32_t* d_bits = nullptr;
  int bits_cap = 0;
  char* d_pad = nullptr;  // pad() passes: descriptors + per-image 256-entry tables
  size_t pad_cap = 0;
};

namespace {
int fail(d2t_prep* p, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (p) p->err = buf;
  return code;
}
#define PHIP(p, expr)                                                                              \
  do {                                                                                             \
    hipError_t e_ = (expr);                                                                        \
    if (e_ != hipSuccess) return fail(p, D2T_EHIP, "%s: %s", #expr, hipGetErrorString(e_));        \
  } while (0)

}  // namespace

extern "C" {

int d2t_prep_plan_image(const d2t_prep_config* cfg, int src_h, int src_w, d2t_prep_plan* plan) {
  return plan_image(cfg, src_h, src_w, plan);
}

int d2t_prep_plan_fallback(const d2t_prep_config* cfg, int src_h, int src_w, d2t_prep_plan* plan) {
  int rc = plan_image(cfg, src_h, src_w, plan);
  if (rc != D2T_OK) return rc;
  plan_fallback(cfg, plan);
  return D2T_OK;
}

int d2t_prep_lanczos_coeffs(int in_size, int out_size, int32_t* ksize_out, int32_t* bounds, int32_t* kk) {
  if (in_size <= 0 || out_size <= 0 || !ksize_out) return D2T_EINVAL;
  AxisTab t;
  build_lanczos(in_size, out_size, t);
  *ksize_out = t.ksize;
  if (bounds) memcpy(bounds, t.bounds.data(), t.bounds.size() * 4);
  if (kk) memcpy(kk, t.coef.data(), t.coef.size() * 4);
  return D2T_OK;
}

int d2t_prep_create(const d2t_prep_config* cfg, d2t_prep** out) {
  if (!cfg || !out) return D2T_EINVAL;
  d2t_prep* p = new d2t_prep();
  p->cfg = *cfg;
  *out = p;
  if (cfg->max_h <= 0 || cfg->max_w <= 0 || cfg->downsample < 0 || !(cfg->std != 0.f))
    return fail(p, D2T_EINVAL, "bad pre-processing config");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return fail(p, D2T_EHIP, "no HIP device visible (the pre-processing has no CPU path)");
  if (cfg->norm_mode != D2T_NORM_ALB && cfg->norm_mode != D2T_NORM_RAW) return fail(p, D2T_EINVAL, "bad norm_mode");
  float host[256];
  for (int v = 0; v < 256; ++v) host[v] = normalise_value(*cfg, (float)v);
  PHIP(p, hipMalloc(&p->lut, sizeof host));
  PHIP(p, hipMemcpy(p->lut, host, sizeof host, hipMemcpyHostToDevice));
  for (auto& st : p->stage) PHIP(p, hipEventCreateWithFlags(&st.done, hipEventDisableTiming));
  return D2T_OK;
}

void d2t_prep_destroy(d2t_prep* p) {
  if (!p) return;
  for (auto& st : p->stage) {
    if (st.pending) hipEventSynchronize(st.done);
    if (st.d) hipFree(st.d);
    if (st.h) hipHostFree(st.h);
    if (st.done) hipEventDestroy(st.done);
  }
  if (p->lut) hipFree(p->lut);
  if (p->d_work) hipFree(p->d_work);
  if (p->d_arena) hipFree(p->d_arena);
  if (p->d_bits) hipFree(p->d_bits);
  if (p->d_pad) hipFree(p->d_pad);
  delete p;
}

const char* d2t_prep_last_error(const d2t_prep* p) { return p ? p->err.c_str() : "null handle"; }

int d2t_prep_run(d2t_prep* p, int n, const d2t_prep_plan* plans, const uint8_t* src_dev, const int64_t* src_offsets,
                 float* out_dev, int out_h, int out_w, int32_t* flags_dev, void* stream_) {
  if (!p) return D2T_EINVAL;
  if (n <= 0 || !plans || !src_dev || !src_offsets || !out_dev || out_h <= 0 || out_w <= 0)
    return fail(p, D2T_EINVAL, "d2t_prep_run: bad argument");
  if (!p->lut) return fail(p, D2T_ESTATE, "d2t_prep_run: handle was not created on a HIP device");
  hipStream_t stream = (hipStream_t)stream_;
  const d2t_prep_config& c = p->cfg;

  // ---- validate the plans against the configuration (a plan must be what d2t_prep_plan_image would produce) -------
  for (int i = 0; i < n; ++i) {
    const d2t_prep_plan& pl = plans[i];
    d2t_prep_plan want;
    int rc = pl.status == D2T_PREP_FALLBACK ? d2t_prep_plan_fallback(&c, pl.src_h, pl.src_w, &want)
                                            : plan_image(&c, pl.src_h, pl.src_w, &want);
    if (rc != D2T_OK || memcmp(&want, &pl, sizeof want) != 0)
      return fail(p, D2T_EINVAL, "d2t_prep_run: plan %d does not match the configuration (status %d)", i, pl.status);
    if (pl.status == D2T_PREP_UNBOUND_LOCAL)
      return fail(p, D2T_EINVAL, "d2t_prep_run: plan %d has status UNBOUND_LOCAL (the reference raises for this image)", i);
    if (pl.out_h != out_h || pl.out_w != out_w)
      return fail(p, D2T_EINVAL, "d2t_prep_run: plan %d produces %dx%d, the call's output is %dx%d", i, pl.out_h, pl.out_w,
                  out_h, out_w);
    if (pl.min_branch && !flags_dev) return fail(p, D2T_EINVAL, "d2t_prep_run: plan %d needs flags_dev (paste check)", i);
    if (pl.ds_w > 65000) return fail(p, D2T_EINVAL, "d2t_prep_run: image %d is wider than 65000 pixels", i);
  }

  // ---- resampling tables: resident in a device arena, keyed by (in, out, kind); the ones this call needs and the arena
  // does not hold yet are built on a few host threads (two libm sin() per LANCZOS tap, ~25 k per image = 0.3 ms on one
  // core) and appended with one copy ------------------------------------------------------------------------------------
  enum { K_VERT = 0, K_HORZ = 1, K_AREA = 2 };  // K_HORZ: coefficients transposed [k][out]
  struct Miss {
    TabKey key;
    AxisTab t;
  };
  std::vector<Miss> miss;
  std::vector<std::pair<TabKey, DevTab>> new_tabs;  // this call's new tables (committed to dev_tabs after their upload)
  auto need_tab = [&](int in, int out, int kind) {
    TabKey key{in, out, kind};
    if (p->dev_tabs.count(key)) return;
    for (auto& m : miss)
      if (m.key == key) return;
    miss.push_back(Miss{key, AxisTab()});
  };
  auto collect = [&]() {
    miss.clear();
    for (int i = 0; i < n; ++i) {
      const d2t_prep_plan& pl = plans[i];
      if (pl.ds_h != pl.src_h || pl.ds_w != pl.src_w) {
        const double sx = (double)pl.src_w / pl.ds_w, sy = (double)pl.src_h / pl.ds_h;
        if (!(sx == floor(sx) && sy == floor(sy))) need_tab(pl.src_w, pl.ds_w, K_AREA), need_tab(pl.src_h, pl.ds_h, K_AREA);
      }
      if (pl.status != D2T_PREP_OK) continue;
      if (pl.rs_w != pl.ds_w) need_tab(pl.ds_w, pl.rs_w, K_HORZ);
      if (pl.rs_h != pl.ds_h) need_tab(pl.ds_h, pl.rs_h, K_VERT);
    }
  };
  collect();
  size_t new_words = 0;
  for (auto& m : miss) {
    const double scale = (double)m.key.in / m.key.out;
    const int ks = m.key.kind == K_AREA ? (int)ceil(scale) + 2 : (int)ceil(3.0 * std::max(scale, 1.0)) * 2 + 1;
    new_words += (size_t)m.key.out * (2 + ks);
  }
  if (p->arena_used + new_words > p->arena_cap) {  // start over (rare): everything queued so far must have finished
    PHIP(p, hipStreamSynchronize(stream));
    p->dev_tabs.clear();
    p->arena_used = 0;
    collect();
    new_words = 0;
    for (auto& m : miss) {
      const double scale = (double)m.key.in / m.key.out;
      const int ks = m.key.kind == K_AREA ? (int)ceil(scale) + 2 : (int)ceil(3.0 * std::max(scale, 1.0)) * 2 + 1;
      new_words += (size_t)m.key.out * (2 + ks);
    }
    if (new_words > p->arena_cap) {
      if (p->d_arena) PHIP(p, hipFree(p->d_arena));
      p->d_arena = nullptr, p->arena_cap = 0;
      // >= 64 MB of int32 (D2T_PREP_ARENA_WORDS: a small arena for the test of the start-over path)
      static const size_t min_words = getenv("D2T_PREP_ARENA_WORDS") ? (size_t)atoll(getenv("D2T_PREP_ARENA_WORDS")) : (size_t)16 << 20;
      const size_t cap = std::max<size_t>(new_words * 2, min_words);
      PHIP(p, hipMalloc((void**)&p->d_arena, cap * 4));
      p->arena_cap = cap;
    }
  }
  if (miss.size() > 1) {
    const int nt = (int)std::min<size_t>(std::min<size_t>(miss.size(), 8), std::max(1u, std::thread::hardware_concurrency()));
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t)
      th.emplace_back([&, t] {
        for (size_t k = t; k < miss.size(); k += nt)
          (miss[k].key.kind == K_AREA ? build_area : build_lanczos)(miss[k].key.in, miss[k].key.out, miss[k].t);
      });
    for (auto& x : th) x.join();
  } else if (miss.size() == 1) {
    (miss[0].key.kind == K_AREA ? build_area : build_lanczos)(miss[0].key.in, miss[0].key.out, miss[0].t);
  }

  // ---- stage descriptors + new tables ---------------------------------------------------------------------------------
  const size_t desc_bytes = ((size_t)n * sizeof(PrepDesc) + 255) & ~(size_t)255;
  const size_t need = desc_bytes + new_words * 4 + 256;
  d2t_prep::Stage& st = p->stage[p->calls++ % d2t_prep::NSTAGE];
  if (st.pending) {  // the copies out of this pinned block (queued NSTAGE calls ago) must have finished
    PHIP(p, hipEventSynchronize(st.done));
    st.pending = false;
  }
  if (need > st.cap) {
    if (st.d) PHIP(p, hipFree(st.d));
    if (st.h) PHIP(p, hipHostFree(st.h));
    st.d = st.h = nullptr;
    st.cap = 0;
    const size_t cap = need * 2;
    PHIP(p, hipHostMalloc((void**)&st.h, cap, hipHostMallocDefault));
    PHIP(p, hipMalloc((void**)&st.d, cap));
    st.cap = cap;
  }
  {
    int32_t* hw = reinterpret_cast<int32_t*>(st.h + desc_bytes);
    size_t at = 0;
    new_tabs.reserve(miss.size());
    for (auto& m : miss) {
      const AxisTab& t = m.t;
      DevTab dt;
      dt.ks = t.ksize;
      dt.b = (int32_t)(p->arena_used + at);
      memcpy(hw + at, t.bounds.data(), t.bounds.size() * 4);
      at += t.bounds.size();
      dt.k = (int32_t)(p->arena_used + at);
      if (m.key.kind == K_HORZ) {
        const int ow = m.key.out;
        for (int xx = 0; xx < ow; ++xx)
          for (int k = 0; k < t.ksize; ++k) hw[at + (size_t)k * ow + xx] = t.coef[(size_t)xx * t.ksize + k];
      } else {
        memcpy(hw + at, t.coef.data(), t.coef.size() * 4);
      }
      at += t.coef.size();
      new_tabs.emplace_back(m.key, dt);
    }
    if (at != new_words) return fail(p, D2T_EINVAL, "d2t_prep_run: internal table size mismatch");
  }
  // tables of this call that are not resident yet: visible to the descriptor loop below, but entered into the
  // resident map only once their upload has been enqueued (an error return in between must not leave map entries
  // that point at arena words which were never written)
  auto tab_at = [&](const TabKey& k) -> const DevTab& {
    for (auto& kv : new_tabs)
      if (kv.first == k) return kv.second;
    return p->dev_tabs.at(k);
  };

  // ---- descriptors ----------------------------------------------------------------------------------------------------
  PrepDesc* descs = reinterpret_cast<PrepDesc*>(st.h);
  const float canvas = normalise_value(c, 255.0f);  // canvas colour 255 through the normalisation table
  size_t work = 0;
  int max_ds_w = 1, max_ds_h = 1, max_rows_h = 0, any_min = 0, any_ds = 0, any_h = 0;
  for (int i = 0; i < n; ++i) {
    const d2t_prep_plan& pl = plans[i];
    PrepDesc& d = descs[i];
    memset(&d, 0, sizeof d);
    d.src_off = src_offsets[i];
    d.src_h = pl.src_h, d.src_w = pl.src_w, d.ds_h = pl.ds_h, d.ds_w = pl.ds_w, d.rs_h = pl.rs_h, d.rs_w = pl.rs_w;
    d.fallback = pl.status == D2T_PREP_FALLBACK;
    d.min_branch = pl.min_branch;
    d.fill = d.fallback ? 1.0f : canvas;
    if (pl.ds_h != pl.src_h || pl.ds_w != pl.src_w) {
      const double sx = (double)pl.src_w / pl.ds_w, sy = (double)pl.src_h / pl.ds_h;
      if (sx == 2.0 && sy == 2.0) {
        d.do_ds = 1;
      } else if (sx == floor(sx) && sy == floor(sy)) {
        d.do_ds = 2, d.ds_fx = (int)sx, d.ds_fy = (int)sy;
      } else {
        d.do_ds = 3;
        const DevTab& ax = tab_at(TabKey{pl.src_w, pl.ds_w, K_AREA});
        d.axb = ax.b, d.axk = ax.k, d.axs = ax.ks;
        const DevTab& ay = tab_at(TabKey{pl.src_h, pl.ds_h, K_AREA});
        d.ayb = ay.b, d.ayk = ay.k, d.ays = ay.ks;
      }
      d.ds_off = (int64_t)work;
      work += ((size_t)pl.ds_h * pl.ds_w + 15) & ~(size_t)15;
      any_ds = 1;
      max_ds_w = std::max(max_ds_w, pl.ds_w), max_ds_h = std::max(max_ds_h, pl.ds_h);
    }
    if (!d.fallback && pl.rs_w != pl.ds_w) {  // ImagingResample: horizontal pass only when the width changes
      d.do_h = 1;
      const DevTab& t = tab_at(TabKey{pl.ds_w, pl.rs_w, K_HORZ});
      d.hb = t.b, d.hk = t.k, d.hks = t.ks;
      d.hp_off = (int64_t)work;
      work += ((size_t)pl.ds_h * pl.rs_w + 15) & ~(size_t)15;
      any_h = 1;
      max_rows_h = std::max(max_rows_h, pl.ds_h);
    }
    if (!d.fallback && pl.rs_h != pl.ds_h) {
      d.do_v = 1;
      const DevTab& t = tab_at(TabKey{pl.ds_h, pl.rs_h, K_VERT});
      d.vb = t.b, d.vk = t.k, d.vks = t.ks;
    }
    if (d.fallback) d.rs_h = std::min(pl.ds_h, out_h), d.rs_w = pl.ds_w;  // F.pad with a negative amount crops
    any_min |= pl.min_branch;
  }
  if (work > p->work_cap) {
    if (p->d_work) PHIP(p, hipFree(p->d_work));
    p->d_work = nullptr, p->work_cap = 0;
    PHIP(p, hipMalloc((void**)&p->d_work, work * 2));
    p->work_cap = work * 2;
  }
  if (n > p->bits_cap) {
    if (p->d_bits) PHIP(p, hipFree(p->d_bits));
    p->d_bits = nullptr, p->bits_cap = 0;
    PHIP(p, hipMalloc((void**)&p->d_bits, (size_t)n * 2 * 4));
    p->bits_cap = n * 2;
  }
  PHIP(p, hipMemcpyAsync(st.d, st.h, (size_t)n * sizeof(PrepDesc), hipMemcpyHostToDevice, stream));
  if (new_words)
    PHIP(p, hipMemcpyAsync(p->d_arena + p->arena_used, st.h + desc_bytes, new_words * 4, hipMemcpyHostToDevice, stream));
  p->arena_used += new_words;
  for (auto& kv : new_tabs) p->dev_tabs.emplace(kv.first, kv.second);
  PHIP(p, hipEventRecord(st.done, stream));
  st.pending = true;
  const PrepDesc* d_descs = reinterpret_cast<const PrepDesc*>(st.d);
  const int32_t* d_tabs = p->d_arena;

  // ---- launches -------------------------------------------------------------------------------------------------------
  if (any_ds)
    hipLaunchKernelGGL(prep_area_kernel, dim3((max_ds_w + 63) / 64, (max_ds_h + 3) / 4, n), dim3(256), 0, stream, d_descs,
                       src_dev, p->d_work, d_tabs);
  if (any_h) {
    int max_in_w = 1;
    for (int i = 0; i < n; ++i)
      if (descs[i].do_h) max_in_w = std::max(max_in_w, descs[i].ds_w);
    const int stride = (max_in_w + 8 + 15) & ~15;
    if (stride * 8 <= 60 * 1024)
      hipLaunchKernelGGL(prep_hpass_kernel<8>, dim3((max_rows_h + 7) / 8, n), dim3(256), (size_t)stride * 8, stream, d_descs,
                         src_dev, p->d_work, d_tabs, stride);
    else
      hipLaunchKernelGGL(prep_hpass_kernel<1>, dim3(max_rows_h, n), dim3(256), (size_t)stride, stream, d_descs, src_dev,
                         p->d_work, d_tabs, stride);
  }
  if (any_min) PHIP(p, hipMemsetAsync(p->d_bits, 0, (size_t)n * 4, stream));
  hipLaunchKernelGGL(prep_finish_kernel, dim3((out_w + 255) / 256, (out_h + FR - 1) / FR, n), dim3(256), 0, stream, d_descs, src_dev,
                     (const uint8_t*)p->d_work, d_tabs, (const float*)p->lut, out_dev, out_h, out_w, p->d_bits);
  if (flags_dev) {
    if (!any_min) PHIP(p, hipMemsetAsync(p->d_bits, 0, (size_t)n * 4, stream));
    hipLaunchKernelGGL(prep_flags_kernel, dim3((n + 63) / 64), dim3(64), 0, stream, d_descs, (const int32_t*)p->d_bits,
                       flags_dev, n);
  }
  PHIP(p, hipGetLastError());
  return D2T_OK;
}

// ---- pad() entry points: descriptors / tables go through a small synchronous upload (these calls are followed by a
// device-to-host read of their result anyway) --------------------------------------------------------------------------
namespace {
int pad_upload(d2t_prep* p, const std::vector<PadDesc>& descs, const uint8_t* tab_host, int n, PadDesc** d_descs,
               uint8_t** d_tab, hipStream_t stream) {
  const size_t db = ((size_t)n * sizeof(PadDesc) + 255) & ~(size_t)255, tb = tab_host ? (size_t)n * 256 : 0;
  if (db + tb > p->pad_cap) {
    if (p->d_pad) PHIP(p, hipFree(p->d_pad));
    p->d_pad = nullptr, p->pad_cap = 0;
    PHIP(p, hipMalloc((void**)&p->d_pad, (db + tb) * 2));
    p->pad_cap = (db + tb) * 2;
  }
  // pageable source: the copy is staged by the runtime before the call returns
  PHIP(p, hipMemcpyAsync(p->d_pad, descs.data(), (size_t)n * sizeof(PadDesc), hipMemcpyHostToDevice, stream));
  if (tab_host) PHIP(p, hipMemcpyAsync(p->d_pad + db, tab_host, tb, hipMemcpyHostToDevice, stream));
  PHIP(p, hipStreamSynchronize(stream));
  *d_descs = reinterpret_cast<PadDesc*>(p->d_pad);
  if (d_tab) *d_tab = reinterpret_cast<uint8_t*>(p->d_pad + db);
  return D2T_OK;
}
int pad_common(d2t_prep* p, int n, const uint8_t* src_dev, const int64_t* offs, const int32_t* hs, const int32_t* ws,
               std::vector<PadDesc>& descs, int* max_h, int* max_w) {
  if (!p) return D2T_EINVAL;
  if (n <= 0 || !src_dev || !offs || !hs || !ws) return fail(p, D2T_EINVAL, "pad: bad argument");
  if (!p->lut) return fail(p, D2T_ESTATE, "pad: handle was not created on a HIP device");
  descs.assign(n, PadDesc());
  *max_h = *max_w = 1;
  for (int i = 0; i < n; ++i) {
    if (hs[i] <= 0 || ws[i] <= 0) return fail(p, D2T_EINVAL, "pad: image %d has size %dx%d", i, hs[i], ws[i]);
    memset(&descs[i], 0, sizeof(PadDesc));
    descs[i].src_off = offs[i], descs[i].h = hs[i], descs[i].w = ws[i];
    *max_h = std::max(*max_h, hs[i]), *max_w = std::max(*max_w, ws[i]);
  }
  return D2T_OK;
}
}  // namespace

int d2t_prep_pad_hist(d2t_prep* p, int n, const uint8_t* src_dev, const int64_t* src_offsets, const int32_t* src_h,
                      const int32_t* src_w, int32_t* hist_dev, void* stream_) {
  std::vector<PadDesc> descs;
  int mh, mw;
  int rc = pad_common(p, n, src_dev, src_offsets, src_h, src_w, descs, &mh, &mw);
  if (rc) return rc;
  if (!hist_dev) return fail(p, D2T_EINVAL, "pad: hist_dev is null");
  hipStream_t stream = (hipStream_t)stream_;
  PadDesc* dd;
  if ((rc = pad_upload(p, descs, nullptr, n, &dd, nullptr, stream))) return rc;
  PHIP(p, hipMemsetAsync(hist_dev, 0, (size_t)n * 256 * 4, stream));
  const int bx = (int)std::min<size_t>(256, ((size_t)mh * mw + 65535) / 65536 + 1);
  hipLaunchKernelGGL(pad_hist_kernel, dim3(bx, n), dim3(256), 0, stream, (const PadDesc*)dd, src_dev, hist_dev);
  PHIP(p, hipGetLastError());
  return D2T_OK;
}

int d2t_prep_pad_bbox(d2t_prep* p, int n, const uint8_t* src_dev, const int64_t* src_offsets, const int32_t* src_h,
                      const int32_t* src_w, const uint8_t* mask_host, int32_t* bbox_dev, void* stream_) {
  std::vector<PadDesc> descs;
  int mh, mw;
  int rc = pad_common(p, n, src_dev, src_offsets, src_h, src_w, descs, &mh, &mw);
  if (rc) return rc;
  if (!mask_host || !bbox_dev) return fail(p, D2T_EINVAL, "pad: mask / bbox is null");
  hipStream_t stream = (hipStream_t)stream_;
  PadDesc* dd;
  uint8_t* dm;
  if ((rc = pad_upload(p, descs, mask_host, n, &dd, &dm, stream))) return rc;
  std::vector<int32_t> init((size_t)n * 4);
  for (int i = 0; i < n; ++i) init[4 * i] = init[4 * i + 1] = 0x7fffffff, init[4 * i + 2] = init[4 * i + 3] = -1;
  PHIP(p, hipMemcpyAsync(bbox_dev, init.data(), init.size() * 4, hipMemcpyHostToDevice, stream));
  PHIP(p, hipStreamSynchronize(stream));
  hipLaunchKernelGGL(pad_bbox_kernel, dim3(std::min(mh, 128), n), dim3(256), 0, stream, (const PadDesc*)dd, src_dev,
                     (const uint8_t*)dm, bbox_dev);
  PHIP(p, hipGetLastError());
  return D2T_OK;
}

int d2t_prep_pad_apply(d2t_prep* p, int n, const uint8_t* src_dev, const int64_t* src_offsets, const int32_t* src_h,
                       const int32_t* src_w, const int32_t* rects, const uint8_t* lut_host, int32_t background,
                       uint8_t* dst_dev, const int64_t* dst_offsets, const int32_t* dst_h, const int32_t* dst_w,
                       int32_t* flags_dev, void* stream_) {
  std::vector<PadDesc> descs;
  int mh, mw;
  int rc = pad_common(p, n, src_dev, src_offsets, src_h, src_w, descs, &mh, &mw);
  if (rc) return rc;
  if (!rects || !lut_host || !dst_dev || !dst_offsets || !dst_h || !dst_w || !flags_dev || background < 0 || background > 255)
    return fail(p, D2T_EINVAL, "pad_apply: bad argument");
  int odh = 1, odw = 1;
  for (int i = 0; i < n; ++i) {
    PadDesc& d = descs[i];
    d.a = rects[4 * i], d.b = rects[4 * i + 1], d.cw = rects[4 * i + 2], d.ch = rects[4 * i + 3];
    d.dh = dst_h[i], d.dw = dst_w[i], d.dst_off = dst_offsets[i], d.background = background;
    if (d.a < 0 || d.b < 0 || d.cw <= 0 || d.ch <= 0 || d.a + d.cw > d.w || d.b + d.ch > d.h || d.dh < d.ch || d.dw < d.cw)
      return fail(p, D2T_EINVAL, "pad_apply: rectangle %d is outside its image or larger than its destination", i);
    odh = std::max(odh, d.dh), odw = std::max(odw, d.dw);
  }
  hipStream_t stream = (hipStream_t)stream_;
  PadDesc* dd;
  uint8_t* dl;
  if ((rc = pad_upload(p, descs, lut_host, n, &dd, &dl, stream))) return rc;
  if (n > p->bits_cap) {
    if (p->d_bits) PHIP(p, hipFree(p->d_bits));
    p->d_bits = nullptr, p->bits_cap = 0;
    PHIP(p, hipMalloc((void**)&p->d_bits, (size_t)n * 2 * 4));
    p->bits_cap = n * 2;
  }
  PHIP(p, hipMemsetAsync(p->d_bits, 0, (size_t)n * 4, stream));
  hipLaunchKernelGGL(pad_apply_kernel, dim3((odw + 255) / 256, odh, n), dim3(256), 0, stream, (const PadDesc*)dd, src_dev,
                     (const uint8_t*)dl, dst_dev, p->d_bits);
  hipLaunchKernelGGL(pad_flags_kernel, dim3((n + 63) / 64), dim3(64), 0, stream, (const int32_t*)p->d_bits, flags_dev, n);
  PHIP(p, hipGetLastError());
  return D2T_OK;
}

}  // extern "C"
