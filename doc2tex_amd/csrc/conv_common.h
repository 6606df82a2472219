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

// Split-activation layout ("planes"): per row and per 32-channel group one 128-byte record
// [32 x hi | 32 x lo] (bf16), so both halves of a K-step's operand share a cache line.
// Index (in uint16 units) of the hi part of element (row, c); the lo part sits 32 elements further.
__device__ __forceinline__ size_t plane_idx(size_t row, int c, int C) {
  return (row * C + (size_t)(c & ~31)) * 2 + (c & 31);
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
          v += bf16_bits_to_f32(p.res_hi[ri]) + bf16_bits_to_f32(p.res_hi[ri + 32]);
        }
        v = apply_act(v, p.act);
        if (p.row_add) v += p.row_add[(size_t)(p.row_add_off + in_img) * p.Cout + n];
        if (p.out_hi) {
          uint16_t hi, lo;
          split_f32(v, hi, lo);
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

}  // namespace d2t
