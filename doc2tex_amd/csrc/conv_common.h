// Shared pieces of the implicit-GEMM convolution kernels (fp32 and bf16x3 variants).
#pragma once
#include "kernels.h"

namespace d2t {

using f32x16 = __attribute__((ext_vector_type(16))) float;

__device__ __forceinline__ float apply_act(float v, int act) {
  if (act == ACT_RELU) return fmaxf(v, 0.f);
  if (act == ACT_GELU) return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
  return v;
}

// fp32 <-> split bf16 (hi = upper 16 bits, lo = bf16(x - hi), round to nearest even)
__device__ __forceinline__ float bf16_bits_to_f32(uint16_t b) { return __uint_as_float((unsigned)b << 16); }
__device__ __forceinline__ void split_f32(float x, uint16_t& hi, uint16_t& lo) {
  const unsigned u = __float_as_uint(x);
  hi = (uint16_t)(u >> 16);
  const __bf16 l = (__bf16)(x - __uint_as_float(u & 0xFFFF0000u));
  lo = *reinterpret_cast<const uint16_t*>(&l);
}

// fp16x2 mode (ConvP::f16, round 3): a record holds the activation ONCE, as fp16, in its hi half (the lo half is unused);
// the convolution is x16 * w_lo + x16 * w_hi with the weights as fp16 hi + fp16 lo -- two MFMAs per product instead of three.
// FINITE values beyond the fp16 range saturate (a feature map of a trained model stays far below 65504); NaN and +-Inf pass
// through the conversion unchanged, so a numerically broken forward stays visible in the logits as it does in the other modes
// (fmaxf / fminf alone would turn a NaN into -65504 and the next ReLU into 0).
__device__ __forceinline__ uint16_t f32_to_f16_bits(float x) {
  const float c = fminf(fmaxf(x, -65504.f), 65504.f);
  const _Float16 h = (_Float16)(fabsf(x) <= 3.402823466e38f ? c : x);  // v_cvt_f16_f32: round to nearest even; NaN -> NaN, Inf -> Inf
  return *reinterpret_cast<const uint16_t*>(&h);
}
__device__ __forceinline__ float f16_bits_to_f32(uint16_t b) { return (float)*reinterpret_cast<const _Float16*>(&b); }
// record formats: 0 = bf16 hi | lo (x ~ hi + lo), 1 = ONE fp16 in the hi half (lo half unused), 2 = fp16 hi | fp16 lo (x ~ hi + lo)
enum : int { REC_BF16 = 0, REC_F16 = 1, REC_F16_PAIR = 2 };
__device__ __forceinline__ int out_fmt(const ConvP& p) { return p.out_fmt ? p.out_fmt - 1 : p.f16; }
__device__ __forceinline__ int res_fmt(const ConvP& p) { return p.res_fmt ? p.res_fmt - 1 : p.f16; }
// one element of a record: (hi, lo) <-> fp32
__device__ __forceinline__ void split_rec(float x, uint16_t& hi, uint16_t& lo, int fmt) {
  if (fmt == REC_BF16) { split_f32(x, hi, lo); return; }
  hi = f32_to_f16_bits(x);
  lo = fmt == REC_F16_PAIR ? f32_to_f16_bits(x - f16_bits_to_f32(hi)) : (uint16_t)0;  // (x - hi is exact in fp32)
}
__device__ __forceinline__ float join_rec(uint16_t hi, uint16_t lo, int fmt) {
  if (fmt == REC_BF16) return bf16_bits_to_f32(hi) + bf16_bits_to_f32(lo);
  return fmt == REC_F16_PAIR ? f16_bits_to_f32(hi) + f16_bits_to_f32(lo) : f16_bits_to_f32(hi);
}
// four consecutive elements: rh / rl = the 8-byte hi / lo words of the record
__device__ __forceinline__ void add_rec4(float (&v)[4], const uint2 rh, const uint2 rl, int fmt) {
  if (fmt != REC_BF16) {
    const unsigned h[2] = {rh.x, rh.y}, l[2] = {rl.x, rl.y};
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      float a = f16_bits_to_f32((uint16_t)(h[e] & 0xFFFFu)), b = f16_bits_to_f32((uint16_t)(h[e] >> 16));
      if (fmt == REC_F16_PAIR) { a += f16_bits_to_f32((uint16_t)(l[e] & 0xFFFFu)); b += f16_bits_to_f32((uint16_t)(l[e] >> 16)); }
      v[2 * e] += a;
      v[2 * e + 1] += b;
    }
  } else {
    v[0] += __uint_as_float(rh.x << 16) + __uint_as_float(rl.x << 16);
    v[1] += __uint_as_float(rh.x & 0xFFFF0000u) + __uint_as_float(rl.x & 0xFFFF0000u);
    v[2] += __uint_as_float(rh.y << 16) + __uint_as_float(rl.y << 16);
    v[3] += __uint_as_float(rh.y & 0xFFFF0000u) + __uint_as_float(rl.y & 0xFFFF0000u);
  }
}

// Split-activation layout ("planes"): per row and per 32-channel group one 128-byte record
// [32 x hi | 32 x lo] (bf16), so both halves of a K-step's operand share a cache line.
// Index (in uint16 units) of the hi part of element (row, c); the lo part sits 32 elements further.
__device__ __forceinline__ size_t plane_idx(size_t row, int c, int C) {
  return (row * C + (size_t)(c & ~31)) * 2 + (c & 31);
}

// GEMM row m -> output pixel (b, oh, ow): row-major, or pooled order when a 2x2 max-pool is fused (ConvP::pool2)
__device__ __forceinline__ void conv_row_coords(const ConvP& p, int m, int& b, int& oh, int& ow) {
  if (p.pool2) {
    const int q = m >> 2, sb = m & 3, pw2 = p.OW >> 1, ph2 = p.OH >> 1;
    const int t = q / pw2;
    ow = 2 * (q - t * pw2) + (sb & 1);
    b = t / ph2;
    oh = 2 * (t - b * ph2) + (sb >> 1);
  } else {
    const int ohow = p.OH * p.OW;
    b = m / ohow;
    const int rem = m - b * ohow;
    oh = rem / p.OW;
    ow = rem - oh * p.OW;
  }
}

// XCD-aware tile order: consecutive logical tiles (same A rows, neighbouring pixels) run on the
// same XCD so they share its L2 (bijective remap, cdna guide T1).
__device__ __forceinline__ int xcd_logical_tile() {
  const int nwg = gridDim.x, orig = blockIdx.x, q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
}

// Epilogue of one wave's MI x NJ grid of 32x32 accumulators whose top-left output element is
// (mw, nw): bias (folded BN), residual, activation, positional-table add, row / head-split remaps.
// C/D map of v_mfma_*_32x32*: col = lane&31 -> n, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) -> m.
// (Records here are bf16 hi | lo or, with ConvP::f16, one fp16: the 32x32 kernels never see the mixed-precision formats
// ConvP::out_fmt / res_fmt -- launch_conv / launch_conv_bf16x3 reject them -- and the three-way format code costs the fp32
// kernel, which runs at the VGPR limit, 700 bytes of scratch per lane and half its speed.)
template <int MI, int NJ>
__device__ __forceinline__ void conv_epilogue(const ConvP& p, f32x16 (&acc)[MI][NJ], int mw, int nw, int r, int h) {
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int n = nw + j * 32 + r;
    if (n >= p.Cout) continue;
    const float bias = p.bias ? p.bias[n] : 0.f;
    int slab = 0, head = 0, e = 0;
    if (p.store_mode == STORE_KV) {
      const int d = p.kv_heads * p.kv_hd;
      slab = n / d;
      const int within = n - slab * d;
      head = within / p.kv_hd;
      e = within - head * p.kv_hd;
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int m = mw + i * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
        if (m >= p.M) continue;
        float v = acc[i][j][reg] + bias;
        if (p.store_mode == STORE_KV) {
          const int bb = m / p.kv_T, jj = m - bb * p.kv_T;
          p.out[((((size_t)slab * p.kv_B + bb) * p.kv_heads + head) * p.kv_T + jj) * p.kv_hd + e] = v;
          continue;
        }
        size_t row = (size_t)m;
        int in_img = 0;
        if (p.rows_per_img > 0) {
          const int img = m / p.rows_per_img;
          in_img = m - img * p.rows_per_img;
          row = (size_t)img * p.img_stride + p.row_off + in_img;
        }
        const size_t off = row * p.Cout + n;
        if (p.res) v += p.res[off];
        if (p.res_hi) {
          const size_t ri = plane_idx(row, n, p.Cout);
          v += p.f16 ? f16_bits_to_f32(p.res_hi[ri]) : bf16_bits_to_f32(p.res_hi[ri]) + bf16_bits_to_f32(p.res_hi[ri + 32]);
        }
        v = apply_act(v, p.act);
        if (p.row_add) v += p.row_add[(size_t)(p.row_add_off + in_img) * p.Cout + n];
        if (p.out_hi) {
          uint16_t hi, lo;
          if (p.f16) { hi = f32_to_f16_bits(v); lo = 0; } else split_f32(v, hi, lo);
          const size_t oi = plane_idx(row, n, p.Cout);
          p.out_hi[oi] = hi;
          p.out_hi[oi + 32] = lo;
        } else {
          p.out[off] = v;
        }
      }
    }
  }
}

// Wide epilogue for block tiles whose LDS staging area is free after the K loop (the split-bf16 LDS-DMA kernel): the
// accumulators go through LDS once so that a thread afterwards owns FOUR CONSECUTIVE CHANNELS of one output row, and the
// residual is read and the result written with 8/16-byte accesses instead of one 2-byte (or 4-byte) access per element --
// the element-wise epilogue above costs a residual layer ~10 % of its run time and dominates the short-K layers.
// Same arithmetic per element, in the same order, as conv_epilogue: v = acc + bias; v += res | (res_hi + res_lo);
// activation; split.  Requires: no row remap, no positional add, no head-split store, Cout % 32 == 0.
// LDS tile: fp32 [BM][BN], the 32-float column block XOR-ed with bit 2 of the row so that the two half-waves of an MFMA
// result (rows 4 apart, same columns) hit different banks.
__device__ __forceinline__ bool wide_epilogue_ok(const ConvP& p) {
  return p.store_mode == STORE_ROWS && p.rows_per_img == 0 && !p.row_add && (p.Cout & 31) == 0;
}

template <int BM, int BN, int NT, int MI, int NJ>
__device__ __forceinline__ void conv_epilogue_wide(const ConvP& p, f32x16 (&acc)[MI][NJ], unsigned char* smem, int m0, int n0,
                                                   int wrow, int wcol, int r, int h, int tid) {
  float* tile = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int row = wrow + i * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
        const int col = (wcol + j * 32 + r) ^ (((row >> 2) & 1) << 5);
        tile[row * BN + col] = acc[i][j][reg];
      }
  __syncthreads();
  constexpr int QPR = BN / 4;  // 4-channel quads per tile row
#pragma unroll 2  // more would cost the registers that let a decode wave share the SIMD with four convolution waves
  for (int idx = tid; idx < BM * QPR; idx += NT) {
    const int row = idx / QPR, q = idx % QPR;
    const int m = m0 + row, n = n0 + q * 4;
    if (m >= p.M || n >= p.Cout) continue;
    const int col = (q * 4) ^ (((row >> 2) & 1) << 5);
    const float4 a = *reinterpret_cast<const float4*>(tile + row * BN + col);
    float v[4] = {a.x, a.y, a.z, a.w};
    if (p.bias) {
      const float4 b = *reinterpret_cast<const float4*>(p.bias + n);
      v[0] += b.x, v[1] += b.y, v[2] += b.z, v[3] += b.w;
    }
    const size_t off = (size_t)m * p.Cout + n;
    const size_t pi = plane_idx((size_t)m, n, p.Cout);
    if (p.res) {
      const float4 rr = *reinterpret_cast<const float4*>(p.res + off);
      v[0] += rr.x, v[1] += rr.y, v[2] += rr.z, v[3] += rr.w;
    }
    if (p.res_hi) {
      const uint2 rh = *reinterpret_cast<const uint2*>(p.res_hi + pi), rl = *reinterpret_cast<const uint2*>(p.res_hi + pi + 32);
      add_rec4(v, rh, rl, res_fmt(p));
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], p.act);
    if (p.out_hi) {
      uint16_t hi[4], lo[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) split_rec(v[e], hi[e], lo[e], out_fmt(p));
      uint2 oh, ol;
      oh.x = (unsigned)hi[0] | ((unsigned)hi[1] << 16), oh.y = (unsigned)hi[2] | ((unsigned)hi[3] << 16);
      ol.x = (unsigned)lo[0] | ((unsigned)lo[1] << 16), ol.y = (unsigned)lo[2] | ((unsigned)lo[3] << 16);
      *reinterpret_cast<uint2*>(p.out_hi + pi) = oh;
      *reinterpret_cast<uint2*>(p.out_hi + pi + 32) = ol;
    } else {
      *reinterpret_cast<float4*>(p.out + off) = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
  __syncthreads();  // the tile is staging memory again (next tile's LDS-DMA)
}

// The fused 2x2 max-pool form (ConvP::pool2): tile rows 4r .. 4r+3 are one pooling window; a thread owns four channels of one
// POOLED row.  max, then bias, then activation = the pool of the activated convolution outputs (both monotone), bit for bit.
template <int BM, int BN, int NT, int MI, int NJ>
__device__ __forceinline__ void conv_epilogue_wide_pool(const ConvP& p, f32x16 (&acc)[MI][NJ], unsigned char* smem, int m0, int n0,
                                                        int wrow, int wcol, int r, int h, int tid) {
  float* tile = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int row = wrow + i * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
        const int col = (wcol + j * 32 + r) ^ (((row >> 2) & 1) << 5);
        tile[row * BN + col] = acc[i][j][reg];
      }
  __syncthreads();
  constexpr int QPR = BN / 4;
  for (int idx = tid; idx < (BM / 4) * QPR; idx += NT) {
    const int pr = idx / QPR, q = idx % QPR;
    const int m = m0 + 4 * pr, n = n0 + q * 4;
    if (m >= p.M || n >= p.Cout) continue;
    const int col = (q * 4) ^ ((pr & 1) << 5);  // rows 4 pr .. 4 pr + 3 share (row >> 2) & 1 = pr & 1
    float4 a = *reinterpret_cast<const float4*>(tile + (4 * pr) * BN + col);
#pragma unroll
    for (int k = 1; k < 4; ++k) {
      const float4 t = *reinterpret_cast<const float4*>(tile + (4 * pr + k) * BN + col);
      a.x = fmaxf(a.x, t.x); a.y = fmaxf(a.y, t.y); a.z = fmaxf(a.z, t.z); a.w = fmaxf(a.w, t.w);
    }
    float v[4] = {a.x, a.y, a.z, a.w};
    if (p.bias) {
      const float4 b = *reinterpret_cast<const float4*>(p.bias + n);
      v[0] += b.x, v[1] += b.y, v[2] += b.z, v[3] += b.w;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], p.act);
    const size_t mp = (size_t)(m >> 2);
    const size_t pi = plane_idx(mp, n, p.Cout);
    if (p.out_hi) {
      uint16_t hi[4], lo[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) split_rec(v[e], hi[e], lo[e], out_fmt(p));
      uint2 oh, ol;
      oh.x = (unsigned)hi[0] | ((unsigned)hi[1] << 16), oh.y = (unsigned)hi[2] | ((unsigned)hi[3] << 16);
      ol.x = (unsigned)lo[0] | ((unsigned)lo[1] << 16), ol.y = (unsigned)lo[2] | ((unsigned)lo[3] << 16);
      *reinterpret_cast<uint2*>(p.out_hi + pi) = oh;
      *reinterpret_cast<uint2*>(p.out_hi + pi + 32) = ol;
    } else {
      *reinterpret_cast<float4*>(p.out + mp * p.Cout + n) = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
  __syncthreads();
}

// The same with the row remap (rows_per_img), the positional-table add and the head-split K/V store, for the kernels whose
// register budget does not matter (the on-the-fly split-bf16 kernel, the fp32 kernel).  Kept apart from the lean function
// above on purpose: with the remap code compiled in, the split-bf16 LDS-DMA kernel needs 107-121 VGPRs instead of 101; above
// 104 a decode wave no longer fits on a SIMD next to four convolution waves and the decode streams starve (measured: 1139
// instead of 1210 formulas/s).  Requires Cout % 32 == 0 and, for the head-split store, head_dim % 4 == 0.
__device__ __forceinline__ bool wide_epilogue_full_ok(const ConvP& p) {
  return (p.Cout & 31) == 0 && (p.store_mode == STORE_ROWS || (p.kv_hd & 3) == 0);
}

template <int BM, int BN, int NT, int MI, int NJ>
__device__ __forceinline__ void conv_epilogue_wide_full(const ConvP& p, f32x16 (&acc)[MI][NJ], unsigned char* smem, int m0, int n0,
                                                   int wrow, int wcol, int r, int h, int tid) {
  float* tile = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int row = wrow + i * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
        const int col = (wcol + j * 32 + r) ^ (((row >> 2) & 1) << 5);
        tile[row * BN + col] = acc[i][j][reg];
      }
  __syncthreads();
  constexpr int QPR = BN / 4;  // 4-channel quads per tile row
#pragma unroll 2  // more would cost the registers that let a decode wave share the SIMD with four convolution waves
  for (int idx = tid; idx < BM * QPR; idx += NT) {
    const int row = idx / QPR, q = idx % QPR;
    const int m = m0 + row, n = n0 + q * 4;
    if (m >= p.M || n >= p.Cout) continue;
    const int col = (q * 4) ^ (((row >> 2) & 1) << 5);
    const float4 a = *reinterpret_cast<const float4*>(tile + row * BN + col);
    float v[4] = {a.x, a.y, a.z, a.w};
    if (p.bias) {
      const float4 b = *reinterpret_cast<const float4*>(p.bias + n);
      v[0] += b.x, v[1] += b.y, v[2] += b.z, v[3] += b.w;
    }
    if (p.store_mode == STORE_KV) {  // n -> (slab, head, e), m -> (b, j): four consecutive e of one head
      const int d = p.kv_heads * p.kv_hd, slab = n / d, within = n - slab * d;
      const int head = within / p.kv_hd, e = within - head * p.kv_hd;
      const int bb = m / p.kv_T, jj = m - bb * p.kv_T;
      *reinterpret_cast<float4*>(p.out + ((((size_t)slab * p.kv_B + bb) * p.kv_heads + head) * p.kv_T + jj) * p.kv_hd + e) =
          make_float4(v[0], v[1], v[2], v[3]);
      continue;
    }
    size_t orow = (size_t)m;
    int in_img = 0;
    if (p.rows_per_img > 0) {
      const int img = m / p.rows_per_img;
      in_img = m - img * p.rows_per_img;
      orow = (size_t)img * p.img_stride + p.row_off + in_img;
    }
    const size_t off = orow * p.Cout + n;
    const size_t pi = plane_idx(orow, n, p.Cout);
    if (p.res) {
      const float4 rr = *reinterpret_cast<const float4*>(p.res + off);
      v[0] += rr.x, v[1] += rr.y, v[2] += rr.z, v[3] += rr.w;
    }
    if (p.res_hi) {
      const uint2 rh = *reinterpret_cast<const uint2*>(p.res_hi + pi), rl = *reinterpret_cast<const uint2*>(p.res_hi + pi + 32);
      add_rec4(v, rh, rl, res_fmt(p));
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], p.act);
    if (p.row_add) {
      const float4 ra = *reinterpret_cast<const float4*>(p.row_add + (size_t)(p.row_add_off + in_img) * p.Cout + n);
      v[0] += ra.x, v[1] += ra.y, v[2] += ra.z, v[3] += ra.w;
    }
    if (p.out_hi) {
      uint16_t hi[4], lo[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) split_rec(v[e], hi[e], lo[e], out_fmt(p));
      uint2 oh, ol;
      oh.x = (unsigned)hi[0] | ((unsigned)hi[1] << 16), oh.y = (unsigned)hi[2] | ((unsigned)hi[3] << 16);
      ol.x = (unsigned)lo[0] | ((unsigned)lo[1] << 16), ol.y = (unsigned)lo[2] | ((unsigned)lo[3] << 16);
      *reinterpret_cast<uint2*>(p.out_hi + pi) = oh;
      *reinterpret_cast<uint2*>(p.out_hi + pi + 32) = ol;
    } else {
      *reinterpret_cast<float4*>(p.out + off) = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
  __syncthreads();  // the tile is staging memory again (next tile's LDS-DMA)
}

}  // namespace d2t
