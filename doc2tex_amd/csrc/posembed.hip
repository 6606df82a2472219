// Learned position tables of the ViT encoders that are not ViTEncoderV3 (seq_modeling/vit_encoder.py:22-118, :207-226):
// ViTEncoder resizes its [1 + GH*GW][D] table to the crop's patch grid with F.interpolate(mode="bicubic",
// align_corners=False, scale_factor=((gh + 0.1) / GH, (gw + 0.1) / GW)) (:58-95).  The kernels below restate
// ATen's upsample_bicubic2d for that call: source coordinate scale * (dst + 0.5) - 0.5 with scale = 1 / scale_factor,
// cubic-convolution coefficients with A = -0.75, taps clamped to the table, rows summed as sum_i wy[i] * (sum_j wx[j] * v).
// Tables are [cell][D] (channels innermost), so one thread per (cell, channel) reads coalesced rows.
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace d2t {
namespace {
struct Taps {
  int idx[4];
  float w[4];
};
// ATen UpSample.h: area_pixel_compute_source_index(cubic) + guard_index_and_lambda + get_cubic_upsample_coefficients
__device__ __forceinline__ Taps cubic_taps(float scale, int dst, int n_in) {
  const float A = -0.75f;
  const float real = __fsub_rn(__fmul_rn(scale, (float)dst + 0.5f), 0.5f);  // no contraction: the CPU op rounds the product
  const int i0 = min((int)floorf(real), n_in - 1);              // guard_index_and_lambda
  const float t = fminf(fmaxf(real - (float)i0, 0.f), 1.f);
  Taps r;
  const float x0 = t + 1.f, x1 = t, x2 = 1.f - t, x3 = x2 + 1.f;
  r.w[0] = ((A * x0 - 5.f * A) * x0 + 8.f * A) * x0 - 4.f * A;
  r.w[1] = ((A + 2.f) * x1 - (A + 3.f)) * x1 * x1 + 1.f;
  r.w[2] = ((A + 2.f) * x2 - (A + 3.f)) * x2 * x2 + 1.f;
  r.w[3] = ((A * x3 - 5.f * A) * x3 + 8.f * A) * x3 - 4.f * A;
#pragma unroll
  for (int k = 0; k < 4; ++k) r.idx[k] = min(max(i0 - 1 + k, 0), n_in - 1);
  return r;
}

__global__ __launch_bounds__(256) void bicubic_table_kernel(const float* __restrict__ src, float* __restrict__ dst, int GH,
                                                            int GW, int gh, int gw, int D, float sh, float sw) {
  const long long total = (long long)gh * gw * D;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(e % D);
    const int cell = (int)(e / D), ox = cell % gw, oy = cell / gw;
    const Taps ty = cubic_taps(sh, oy, GH), tx = cubic_taps(sw, ox, GW);
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float* row = src + ((size_t)ty.idx[i] * GW) * D + c;
      float r = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) r += row[(size_t)tx.idx[j] * D] * tx.w[j];
      acc += r * ty.w[i];
    }
    dst[e] = acc;
  }
}

// transpose of the above, as a gather so that the sum has a fixed order: dsrc[y][x][c] = sum over the destination cells whose
// taps touch (y, x) of wy * wx * ddst
__global__ __launch_bounds__(256) void bicubic_table_bwd_kernel(const float* __restrict__ ddst, float* __restrict__ dsrc, int GH,
                                                                int GW, int gh, int gw, int D, float sh, float sw) {
  const long long total = (long long)GH * GW * D;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(e % D);
    const int cell = (int)(e / D), x = cell % GW, y = cell / GW;
    float acc = 0.f;
    for (int oy = 0; oy < gh; ++oy) {
      const Taps ty = cubic_taps(sh, oy, GH);
      float wy = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) wy += ty.idx[i] == y ? ty.w[i] : 0.f;
      if (wy == 0.f) continue;
      float r = 0.f;
      for (int ox = 0; ox < gw; ++ox) {
        const Taps tx = cubic_taps(sw, ox, GW);
        float wx = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) wx += tx.idx[j] == x ? tx.w[j] : 0.f;
        if (wx != 0.f) r += wx * ddst[((size_t)oy * gw + ox) * D + c];
      }
      acc += wy * r;
    }
    dsrc[e] = acc;
  }
}

// out[i] = sum_b x[b * stride + i], i < n   (gradient of a table broadcast over the batch)
__global__ __launch_bounds__(256) void sum_over_batch_kernel(const float* __restrict__ x, float* __restrict__ out, int B,
                                                             long long stride, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float a = 0.f;
    for (int b = 0; b < B; ++b) a += x[(size_t)b * stride + i];
    out[i] = a;
  }
}
unsigned blocks_for(long long total) { return (unsigned)std::min<long long>((total + 255) / 256, 4096); }
}  // namespace

hipError_t launch_bicubic_table(const float* src, float* dst, int GH, int GW, int gh, int gw, int D, float scale_h,
                                float scale_w, hipStream_t s) {
  hipLaunchKernelGGL(bicubic_table_kernel, dim3(blocks_for((long long)gh * gw * D)), dim3(256), 0, s, src, dst, GH, GW, gh, gw, D,
                     scale_h, scale_w);
  return hipGetLastError();
}
hipError_t launch_bicubic_table_bwd(const float* ddst, float* dsrc, int GH, int GW, int gh, int gw, int D, float scale_h,
                                    float scale_w, hipStream_t s) {
  hipLaunchKernelGGL(bicubic_table_bwd_kernel, dim3(blocks_for((long long)GH * GW * D)), dim3(256), 0, s, ddst, dsrc, GH, GW, gh,
                     gw, D, scale_h, scale_w);
  return hipGetLastError();
}
hipError_t launch_sum_over_batch(const float* x, float* out, int B, long long stride, long long n, hipStream_t s) {
  hipLaunchKernelGGL(sum_over_batch_kernel, dim3(blocks_for(n)), dim3(256), 0, s, x, out, B, stride, n);
  return hipGetLastError();
}
}  // namespace d2t
