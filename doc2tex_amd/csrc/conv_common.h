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
        v = apply_act(v, p.act);
        if (p.row_add) v += p.row_add[(size_t)(p.row_add_off + in_img) * p.Cout + n];
        p.out[off] = v;
      }
    }
  }
}

}  // namespace d2t
