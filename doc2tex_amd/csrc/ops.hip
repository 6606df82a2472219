// Non-GEMM kernels of the recognizer path (gfx950, 64-wide wavefronts, fp32).
// Each kernel names the reference op sequence it replaces (paths relative to
// /root/reference/doc2tex/modules/component/).
#include "conv_common.h"

namespace d2t {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---------------------------------------------------------------------------
// conv0_1 + bn0_1 + ReLU (feature_extractor/resnet.py:206-208): Cin = 1, 3x3 pad 1.
// HBM-write-bound: one thread = one pixel x 4 output channels, 16-B coalesced stores.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void stem_kernel(const float* __restrict__ img, const float* __restrict__ w,
                                                   const float* __restrict__ bias, float* __restrict__ out, int B,
                                                   int H, int W, int Cout, int act) {
  const int cq = Cout >> 2;
  const long long total = (long long)B * H * W * cq;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int c4 = (int)(idx % cq);
    const long long pix = idx / cq;
    const int x = (int)(pix % W);
    const int y = (int)((pix / W) % H);
    const long long b = pix / ((long long)W * H);
    const float* im = img + b * H * W;
    float v[9];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int yy = y + kh - 1, xx = x + kw - 1;
        v[kh * 3 + kw] = ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) ? im[(long long)yy * W + xx] : 0.f;
      }
    float o[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int oc = c4 * 4 + c;
      float a = 0.f;
#pragma unroll
      for (int t = 0; t < 9; ++t) a = fmaf(v[t], w[oc * 9 + t], a);
      a += bias ? bias[oc] : 0.f;
      o[c] = act == ACT_RELU ? fmaxf(a, 0.f) : a;
    }
    *reinterpret_cast<float4*>(out + pix * Cout + c4 * 4) = make_float4(o[0], o[1], o[2], o[3]);
  }
}

__global__ __launch_bounds__(256) void stem_split_kernel(const float* __restrict__ img, const float* __restrict__ w,
                                                         const float* __restrict__ bias, uint16_t* __restrict__ out,
                                                         int B, int H, int W, int Cout, int act, int f16) {
  const int cq = Cout >> 2;  // == 8: the stride of the loop below is a multiple of 8, so a thread keeps its 4 channels
  const long long total = (long long)B * H * W * cq;
  const int c4 = (int)(((long long)blockIdx.x * blockDim.x + threadIdx.x) % cq);
  float wr[4][9], br[4];  // this thread's filter taps, loaded once (36 loads per pixel otherwise)
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    br[c] = bias ? bias[c4 * 4 + c] : 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) wr[c][t] = w[(c4 * 4 + c) * 9 + t];
  }
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const long long pix = idx / cq;
    const int x = (int)(pix % W);
    const int y = (int)((pix / W) % H);
    const long long b = pix / ((long long)W * H);
    const float* im = img + b * H * W;
    float v[9];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int yy = y + kh - 1, xx = x + kw - 1;
        v[kh * 3 + kw] = ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) ? im[(long long)yy * W + xx] : 0.f;
      }
    uint16_t hi[4], lo[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float a = 0.f;
#pragma unroll
      for (int t = 0; t < 9; ++t) a = fmaf(v[t], wr[c][t], a);
      a += br[c];
      split_rec(act == ACT_RELU ? fmaxf(a, 0.f) : a, hi[c], lo[c], f16);
    }
    const size_t o = plane_idx((size_t)pix, c4 * 4, Cout);
    *reinterpret_cast<uint2*>(out + o) = make_uint2(hi[0] | ((unsigned)hi[1] << 16), hi[2] | ((unsigned)hi[3] << 16));
    *reinterpret_cast<uint2*>(out + o + 32) = make_uint2(lo[0] | ((unsigned)lo[1] << 16), lo[2] | ((unsigned)lo[3] << 16));
  }
}

hipError_t launch_stem_split(const float* img, const float* w, const float* bias, uint16_t* out, int B, int H, int W,
                             int Cout, int act, hipStream_t s, int f16) {
  if (Cout != 32) return hipErrorInvalidValue;  // 8 channel quads per pixel: see the kernel's hoisted weights
  const long long total = (long long)B * H * W * (Cout / 4);
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(stem_split_kernel, dim3(blocks), dim3(256), 0, s, img, w, bias, out, B, H, W, Cout, act, f16);
  return hipGetLastError();
}

hipError_t launch_stem(const float* img, const float* w, const float* bias, float* out, int B, int H, int W, int Cout,
                       int act, hipStream_t s) {
  if (Cout % 4) return hipErrorInvalidValue;
  const long long total = (long long)B * H * W * (Cout / 4);
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(stem_kernel, dim3(blocks), dim3(256), 0, s, img, w, bias, out, B, H, W, Cout, act);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// nn.MaxPool2d(kernel 2) NHWC (resnet.py:94,106,120); padding behaves as -inf.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void maxpool_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int H,
                                                      int W, int C, int OH, int OW, int SH, int SW, int PH, int PW) {
  const int cq = C >> 2;
  const long long total = (long long)B * OH * OW * cq;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int c4 = (int)(idx % cq);
    const long long pix = idx / cq;
    const int ow = (int)(pix % OW);
    const int oh = (int)((pix / OW) % OH);
    const long long b = pix / ((long long)OW * OH);
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
      for (int kw = 0; kw < 2; ++kw) {
        const int ih = oh * SH - PH + kh, iw = ow * SW - PW + kw;
        if ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W) {
          const float4 v = *reinterpret_cast<const float4*>(x + ((b * H + ih) * W + iw) * C + c4 * 4);
          m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
        }
      }
    *reinterpret_cast<float4*>(y + pix * C + c4 * 4) = m;
  }
}

__global__ __launch_bounds__(256) void maxpool_split_kernel(const uint16_t* __restrict__ xp, uint16_t* __restrict__ yp,
                                                            int B, int H, int W, int C, int OH, int OW, int SH, int SW,
                                                            int PH, int PW, int f16) {
  const int cq = C >> 2;
  const long long total = (long long)B * OH * OW * cq;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int c4 = (int)(idx % cq);
    const long long pix = idx / cq;
    const int ow = (int)(pix % OW);
    const int oh = (int)((pix / OW) % OH);
    const long long b = pix / ((long long)OW * OH);
    float m[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
      for (int kw = 0; kw < 2; ++kw) {
        const int ih = oh * SH - PH + kh, iw = ow * SW - PW + kw;
        if ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W) {
          const size_t o = plane_idx((size_t)((b * H + ih) * W + iw), c4 * 4, C);
          const uint2 h2 = *reinterpret_cast<const uint2*>(xp + o), l2 = *reinterpret_cast<const uint2*>(xp + o + 32);
          float v[4] = {0.f, 0.f, 0.f, 0.f};
          add_rec4(v, h2, l2, f16);
          m[0] = fmaxf(m[0], v[0]); m[1] = fmaxf(m[1], v[1]); m[2] = fmaxf(m[2], v[2]); m[3] = fmaxf(m[3], v[3]);
        }
      }
    uint16_t hi[4], lo[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) split_rec(m[c], hi[c], lo[c], f16);
    const size_t o = plane_idx((size_t)pix, c4 * 4, C);
    *reinterpret_cast<uint2*>(yp + o) = make_uint2(hi[0] | ((unsigned)hi[1] << 16), hi[2] | ((unsigned)hi[3] << 16));
    *reinterpret_cast<uint2*>(yp + o + 32) = make_uint2(lo[0] | ((unsigned)lo[1] << 16), lo[2] | ((unsigned)lo[3] << 16));
  }
}

hipError_t launch_maxpool_split(const uint16_t* x, uint16_t* y, int B, int H, int W, int C, int SH, int SW, int PH,
                                int PW, hipStream_t s, int f16) {
  if (C % 32) return hipErrorInvalidValue;
  const int OH = (H + 2 * PH - 2) / SH + 1, OW = (W + 2 * PW - 2) / SW + 1;
  const long long total = (long long)B * OH * OW * (C / 4);
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(maxpool_split_kernel, dim3(blocks), dim3(256), 0, s, x, y, B, H, W, C, OH, OW, SH, SW, PH, PW, f16);
  return hipGetLastError();
}

hipError_t launch_maxpool(const float* x, float* y, int B, int H, int W, int C, int SH, int SW, int PH, int PW,
                          hipStream_t s) {
  if (C % 4) return hipErrorInvalidValue;
  const int OH = (H + 2 * PH - 2) / SH + 1, OW = (W + 2 * PW - 2) / SW + 1;
  const long long total = (long long)B * OH * OW * (C / 4);
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(maxpool_kernel, dim3(blocks), dim3(256), 0, s, x, y, B, H, W, C, OH, OW, SH, SW, PH, PW);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// LayerNorm over the last dim, one wave per row, row in registers
// (vision_transformer.py:119-122 eps 1e-6; nn.TransformerDecoderLayer norms eps 1e-5).
// ---------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                        const float* __restrict__ b, float* __restrict__ y, int rows,
                                                        float eps) {
  constexpr int V = D / 256;  // float4 per lane
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* xr = x + (size_t)row * D;
  float4 v[V];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < V; ++i) {
    v[i] = *reinterpret_cast<const float4*>(xr + (i * 64 + lane) * 4);
    s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  }
  const float mean = wave_sum(s) * (1.f / D);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < V; ++i) {
    v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
    q += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
  }
  const float rstd = 1.f / sqrtf(wave_sum(q) * (1.f / D) + eps);
  float* yr = y + (size_t)row * D;
#pragma unroll
  for (int i = 0; i < V; ++i) {
    const int c = (i * 64 + lane) * 4;
    const float4 gg = *reinterpret_cast<const float4*>(g + c);
    const float4 bb = *reinterpret_cast<const float4*>(b + c);
    float4 o;
    o.x = v[i].x * rstd * gg.x + bb.x;
    o.y = v[i].y * rstd * gg.y + bb.y;
    o.z = v[i].z * rstd * gg.z + bb.z;
    o.w = v[i].w * rstd * gg.w + bb.w;
    *reinterpret_cast<float4*>(yr + c) = o;
  }
}

hipError_t launch_layernorm(const float* x, const float* g, const float* b, float* y, int rows, int D, float eps,
                            hipStream_t s) {
  if (rows <= 0) return hipSuccess;
  const dim3 grid((rows + 3) / 4);
  if (D == 256) hipLaunchKernelGGL(layernorm_kernel<256>, grid, dim3(256), 0, s, x, g, b, y, rows, eps);
  else if (D == 512) hipLaunchKernelGGL(layernorm_kernel<512>, grid, dim3(256), 0, s, x, g, b, y, rows, eps);
  else if (D == 1024) hipLaunchKernelGGL(layernorm_kernel<1024>, grid, dim3(256), 0, s, x, g, b, y, rows, eps);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// ViT self-attention (vision_transformer.py:61-81), head_dim 32.
// qkv [B,N,3,heads,32] -> y [B,N,heads*32].  One block = one (b, head) x 64
// query rows; K (rows padded to 33 floats) and V of that head live in LDS; each
// of the 8 waves takes 8 query rows: lanes own keys for q.k^T and softmax,
// then (key-parity, channel) pairs for P.V.
// ---------------------------------------------------------------------------
constexpr int VA_QB = 64;
constexpr int VA_MAXKPL = 8;  // keys per lane -> N <= 512

__global__ __launch_bounds__(512) void vit_attention_kernel(const float* __restrict__ qkv, float* __restrict__ y, int B,
                                                            int N, int heads) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* Ks = sm;                                  // [N][33]
  float* Vs = Ks + (((size_t)N * 33 + 3) & ~(size_t)3);  // [N][32], 16-B aligned
  float* Ps = Vs + (size_t)N * 32;                 // [8 waves][N]
  const int bh = blockIdx.x, b = bh / heads, hd = bh % heads;
  const int C = heads * 32;
  const float* base = qkv + (size_t)b * N * 3 * C;
  const int tid = threadIdx.x;
  for (int i = tid; i < N * 8; i += 512) {  // 8 float4 per key row
    const int j = i >> 3, c = (i & 7) * 4;
    const float4 kv = *reinterpret_cast<const float4*>(base + (size_t)j * 3 * C + C + hd * 32 + c);
    const float4 vv = *reinterpret_cast<const float4*>(base + (size_t)j * 3 * C + 2 * C + hd * 32 + c);
    float* kd = Ks + j * 33 + c;
    kd[0] = kv.x; kd[1] = kv.y; kd[2] = kv.z; kd[3] = kv.w;
    *reinterpret_cast<float4*>(Vs + j * 32 + c) = vv;
  }
  __syncthreads();
  const int wave = tid >> 6, lane = tid & 63;
  float* P = Ps + (size_t)wave * N;
  const float scale = 0.17677669529663687f;  // 32^-0.5
  const int q0 = blockIdx.y * VA_QB;
  for (int qi = wave; qi < VA_QB; qi += 8) {
    const int qrow = q0 + qi;
    if (qrow >= N) break;  // wave-uniform
    const float* qp = base + (size_t)qrow * 3 * C + hd * 32;
    float q[32];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const float4 t = *reinterpret_cast<const float4*>(qp + c * 4);
      q[c * 4 + 0] = t.x; q[c * 4 + 1] = t.y; q[c * 4 + 2] = t.z; q[c * 4 + 3] = t.w;
    }
    float sc[VA_MAXKPL];
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < VA_MAXKPL; ++t) {
      const int j = lane + 64 * t;
      float a = -INFINITY;
      if (j < N) {
        const float* kr = Ks + j * 33;
        a = 0.f;
#pragma unroll
        for (int c = 0; c < 32; ++c) a = fmaf(q[c], kr[c], a);
        a *= scale;
      }
      sc[t] = a;
      mx = fmaxf(mx, a);
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < VA_MAXKPL; ++t) {
      const int j = lane + 64 * t;
      if (j < N) {
        const float e = expf(sc[t] - mx);
        sum += e;
        P[j] = e;
      }
    }
    sum = wave_sum(sum);
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): P visible to the whole wave
    const int c = lane & 31, par = lane >> 5;
    float acc = 0.f;
    for (int j = par; j < N; j += 2) acc = fmaf(P[j], Vs[j * 32 + c], acc);
    acc += __shfl_xor(acc, 32, 64);
    if (par == 0) y[((size_t)b * N + qrow) * C + hd * 32 + c] = acc / sum;
    __builtin_amdgcn_wave_barrier();
  }
}

// ---------------------------------------------------------------------------
// ViT attention on the fp32 matrix cores.  One block per (image, head); wave w owns the 32 queries [32w, 32w+32) and
// walks the 32-key tiles flash-style: S = (scale Q) K^T by 16 v_mfma_f32_32x32x2_f32, running row maximum (the
// accumulator holds a row in one register across 32 lanes, so the row maximum is a 5-step lane reduction), rescale,
// P = exp(S - max) staged through a per-wave LDS tile to turn accumulator layout into the A-operand layout, O += P V by
// 16 more MFMAs.  K rows are padded to 33 floats (conflict-free B-operand reads), V rows are read lane-contiguous.
// head_dim = 32.  Up to 512 tokens the block holds the head's whole K / V in LDS (one chunk; grid = images x heads).  Longer
// sequences (crops beyond about 180 x 720: the shipped configurations allow up to 448 x 960 = 1695 tokens) take the same
// loop in CHUNKS of `kct` key tiles staged one after the other -- the running maximum / sum / O carry over, the keys are
// visited in the same order -- and the queries are split over `qchunks` blocks of up to 16 waves per (image, head).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void vit_attention_mfma_kernel(const float* __restrict__ qkv, float* __restrict__ y,
                                                                  int N, int heads, int tiles, int kct, int qchunks) {
  // Round 3: the score product is taken TRANSPOSED (S^T = K Q^T: rows = keys, columns = this wave's 32 queries), so a lane
  // holds, for ONE query (its column), 16 of the tile's 32 keys in its accumulator registers -- and that is already the
  // A-operand layout of the second product when MFMA step kk is made to mean "key (kk & 3) + 8 (kk >> 2) + 4 h" (the V rows are
  // simply read in that order).  No LDS tile for P (113 -> 75 KB: two blocks per CU), the row maximum is 16 in-register
  // maxima and one lane exchange instead of sixteen 5-step lane reductions, and the exps are 16 per lane as before.
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int Np = kct * 32;                 // keys per staged chunk
  float* Ks = sm;                          // [Np][33]
  float* Vs = Ks + (size_t)Np * 33;        // [Np][32]
  const int bh = blockIdx.x / qchunks, qc = blockIdx.x % qchunks;
  const int b = bh / heads, hh = bh % heads, C = heads * 32;
  const float* base = qkv + (size_t)b * N * 3 * C;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, nthreads = blockDim.x;
  const int r = lane & 31, h = lane >> 5, q0 = (qc * 16 + wave) * 32;
  // B operand of S^T: lane (k = h, column = query r) supplies scale * Q[q0 + r][2 kk + h]
  float qb[16];
  {
    const int row = q0 + r < N ? q0 + r : N - 1;  // clamp: queries beyond N are computed and discarded
    const float* qp = base + (size_t)row * 3 * C + hh * 32;
    const float scale = 0.17677669529663687f;     // 32^-0.5
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) qb[kk] = qp[2 * kk + h] * scale;
  }
  f32x16 o;
#pragma unroll
  for (int e = 0; e < 16; ++e) o[e] = 0.f;
  float mrun = -INFINITY, lrun = 0.f;  // of query q0 + r (both lane halves hold the same values)
  for (int t0 = 0; t0 < tiles; t0 += kct) {
  if (t0) __syncthreads();  // everyone is done with the previous chunk
  for (int i = tid; i < Np * 8; i += nthreads) {  // 8 float4 per key row; rows >= N are zero
    const int jl = i >> 3, j = t0 * 32 + jl, c = (i & 7) * 4;
    float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
    if (j < N) {
      kv = *reinterpret_cast<const float4*>(base + (size_t)j * 3 * C + C + hh * 32 + c);
      vv = *reinterpret_cast<const float4*>(base + (size_t)j * 3 * C + 2 * C + hh * 32 + c);
    }
    float* kd = Ks + jl * 33 + c;
    kd[0] = kv.x; kd[1] = kv.y; kd[2] = kv.z; kd[3] = kv.w;
    *reinterpret_cast<float4*>(Vs + jl * 32 + c) = vv;
  }
  __syncthreads();
  const int tend = t0 + kct < tiles ? t0 + kct : tiles;
  for (int t = t0; t < tend; ++t) {
    const int tl = t - t0;  // tile index inside the staged chunk
    f32x16 sacc;
#pragma unroll
    for (int e = 0; e < 16; ++e) sacc[e] = 0.f;
    // A operand: lane (row = key r of the tile, k = h) supplies K[t*32 + r][2 kk + h]
    const float* ka = Ks + (size_t)(tl * 32 + r) * 33 + h;
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[2 * kk], qb[kk], sacc, 0, 0, 0);
    // register e of this lane: key t*32 + (e & 3) + 8 (e >> 2) + 4 h, query q0 + r
    float sv[16], mx = -INFINITY;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int key = t * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
      sv[e] = key < N ? sacc[e] : -INFINITY;
      mx = fmaxf(mx, sv[e]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));  // the other half of the tile's keys
    const float mnew = fmaxf(mrun, mx);      // (every tile holds at least one valid key: finite)
    const float corr = expf(mrun - mnew);    // 0 on the first tile
    float pe[16], ps = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      pe[e] = expf(sv[e] - mnew);
      ps += pe[e];
    }
    lrun = lrun * corr + ps;
    mrun = mnew;
    // O rows are queries (e & 3) + 8 (e >> 2) + 4 h: their correction lives in the lane of that query
#pragma unroll
    for (int e = 0; e < 16; ++e) o[e] *= __shfl(corr, (e & 3) + 8 * (e >> 2) + 4 * h, 64);
    // O += P V_t with MFMA step kk <-> key (kk & 3) + 8 (kk >> 2) + 4 h: A[query r][k = h] = pe[kk], B[k = h][d = r] = V[that key][r]
    const float* vb = Vs + (size_t)(tl * 32 + 4 * h) * 32 + r;
#pragma unroll
    for (int kk = 0; kk < 16; ++kk)
      o = __builtin_amdgcn_mfma_f32_32x32x2f32(pe[kk], vb[(size_t)((kk & 3) + 8 * (kk >> 2)) * 32], o, 0, 0, 0);
  }
  }
  lrun += __shfl_xor(lrun, 32, 64);  // each half summed its own 16 keys per tile
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int qi = (e & 3) + 8 * (e >> 2) + 4 * h;
    const float l = __shfl(lrun, qi, 64);
    const int row = q0 + qi;
    if (row < N) y[((size_t)b * N + row) * C + hh * 32 + r] = o[e] / l;
  }
}

hipError_t launch_vit_attention(const float* qkv, float* y, int B, int N, int heads, hipStream_t s) {
  static const bool valu = D2T_PROBE_ENV_STR("D2T_VIT_ATTN_VALU") != nullptr;
  if (!valu) {
    const int tiles = (N + 31) / 32;
    // up to 512 tokens: the whole head in one chunk, one block per (image, head) -- the launch of rounds 2-3, unchanged;
    // beyond: 256-key chunks (66 KB: two blocks per CU) and the queries in blocks of sixteen waves
    const int kct = tiles <= 16 ? tiles : 8, qchunks = (tiles + 15) / 16, waves = tiles < 16 ? tiles : 16;
    const size_t lds2 = ((size_t)kct * 32 * 33 + (size_t)kct * 32 * 32) * sizeof(float);
    static bool attr2 = false;
    if (!attr2) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(vit_attention_mfma_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) return e;
      attr2 = true;
    }
    if (lds2 <= 160 * 1024) {
      hipLaunchKernelGGL(vit_attention_mfma_kernel, dim3(B * heads * qchunks), dim3(waves * 64), lds2, s, qkv, y, N, heads, tiles,
                         kct, qchunks);
      return hipGetLastError();
    }
  }
  if (N > 64 * VA_MAXKPL) return hipErrorInvalidValue;
  const size_t lds = ((((size_t)N * 33 + 3) & ~(size_t)3) + (size_t)N * 32 + 8 * (size_t)N) * sizeof(float);
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(vit_attention_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  dim3 grid(B * heads, (N + VA_QB - 1) / VA_QB);
  hipLaunchKernelGGL(vit_attention_kernel, grid, dim3(512), lds, s, qkv, y, B, N, heads);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Weight packing: OIHW -> OHWI with eval-BatchNorm folded
//   w'[o][kh][kw][c] = w[o][c][kh][kw] * s[o],  s = gamma / sqrt(var + eps)
//   b'[o] = beta - mean * s  (+ conv_bias * s)
// ---------------------------------------------------------------------------
__global__ void pack_conv_kernel(const float* __restrict__ w, const float* __restrict__ cb,
                                 const float* __restrict__ g, const float* __restrict__ be,
                                 const float* __restrict__ mu, const float* __restrict__ var, float eps,
                                 float* __restrict__ wo, float* __restrict__ bo, int Cout, int Cin, int KH, int KW) {
  const long long total = (long long)Cout * Cin * KH * KW;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % Cin);
    const int kw = (int)((idx / Cin) % KW);
    const int kh = (int)((idx / ((long long)Cin * KW)) % KH);
    const int o = (int)(idx / ((long long)Cin * KW * KH));
    const float s = g ? g[o] / sqrtf(var[o] + eps) : 1.f;
    // K order of the packed weights = the order the conv kernels sweep K: 32-channel chunk outermost,
    // filter tap inside it, channel within the chunk innermost (see conv_common.h: advance_k)
    const size_t kdst = ((size_t)(c >> 5) * KH * KW + (size_t)kh * KW + kw) * 32 + (c & 31);
    const size_t dst = Cin % 32 == 0 ? (size_t)o * Cin * KH * KW + kdst : (size_t)idx;
    wo[dst] = w[(((size_t)o * Cin + c) * KH + kh) * KW + kw] * s;
    if (c == 0 && kw == 0 && kh == 0) {
      float bias = cb ? cb[o] * s : 0.f;
      if (g) bias += be[o] - mu[o] * s;
      bo[o] = bias;
    }
  }
}

hipError_t launch_pack_conv(const float* w_oihw, const float* conv_bias, const float* bn_w, const float* bn_b,
                            const float* bn_mean, const float* bn_var, float eps, float* w_out, float* bias_out,
                            int Cout, int Cin, int KH, int KW, hipStream_t s) {
  const long long total = (long long)Cout * Cin * KH * KW;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(pack_conv_kernel, dim3(blocks), dim3(256), 0, s, w_oihw, conv_bias, bn_w, bn_b, bn_mean, bn_var,
                     eps, w_out, bias_out, Cout, Cin, KH, KW);
  return hipGetLastError();
}

__global__ void repack_ohwi_kernel(const float* __restrict__ w, float* __restrict__ wo, int Cout, int KH, int KW, int Cin) {
  const long long total = (long long)Cout * KH * KW * Cin;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % Cin);
    const int tap = (int)((idx / Cin) % (KH * KW));
    const long long o = idx / ((long long)Cin * KH * KW);
    wo[o * Cin * KH * KW + ((size_t)(c >> 5) * KH * KW + tap) * 32 + (c & 31)] = w[idx];
  }
}
hipError_t launch_repack_ohwi(const float* w_ohwi, float* w_out, int Cout, int KH, int KW, int Cin, hipStream_t s) {
  const long long total = (long long)Cout * KH * KW * Cin;
  hipLaunchKernelGGL(repack_ohwi_kernel, dim3((unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096)),
                     dim3(256), 0, s, w_ohwi, w_out, Cout, KH, KW, Cin);
  return hipGetLastError();
}

hipError_t launch_copy(const float* src, float* dst, size_t n, hipStream_t s) {
  return hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, s);
}

__global__ void add_rows_kernel(const float* a, const float* b, float* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = a[i] + b[i];
}
hipError_t launch_add_rows(const float* a, const float* b, float* out, int n, hipStream_t s) {
  hipLaunchKernelGGL(add_rows_kernel, dim3((n + 255) / 256), dim3(256), 0, s, a, b, out, n);
  return hipGetLastError();
}

__global__ void fill_cls_kernel(const float* row, float* out, long long img_stride, int D) {
  for (int i = threadIdx.x; i < D; i += blockDim.x) out[(long long)blockIdx.x * img_stride + i] = row[i];
}
hipError_t launch_fill_cls(const float* row, float* out, int B, long long img_stride_floats, int D, hipStream_t s) {
  hipLaunchKernelGGL(fill_cls_kernel, dim3(B), dim3(256), 0, s, row, out, img_stride_floats, D);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// GlobalContext block (addon_module/visual_attention.py:105-165; gcb: True): per image a 1x1-conv attention map over
// the H*W positions, softmax, attention-pooled channel vector, ConvMLP (fc1 -> LayerNorm -> ReLU -> fc2), added to
// every position.  x is NHWC fp32 [B][HW][C], modified in place.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gc_logits_kernel(const float* __restrict__ x, const float* __restrict__ wg,
                                                        const float* __restrict__ bg, float* __restrict__ logits,
                                                        long long rows, int C) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  float a = 0.f;
  for (int c = lane; c < C; c += 64) a = fmaf(x[row * C + c], wg[c], a);
  a = wave_sum(a);
  if (lane == 0) logits[row] = a + bg[0];
}
// ctx[b][c] = sum_p softmax_p(logits[b])[p] * x[b][p][c];  one block per image
__global__ __launch_bounds__(512) void gc_pool_kernel(const float* __restrict__ x, const float* __restrict__ logits,
                                                      float* __restrict__ ctx, int HW, int C) {
  __shared__ float red[16], wts[512];
  const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const float* l = logits + (size_t)b * HW;
  float m = -INFINITY;
  for (int p = tid; p < HW; p += 512) m = fmaxf(m, l[p]);
  m = wave_max(m);
  if (lane == 0) red[wave] = m;
  __syncthreads();
  m = red[0];
#pragma unroll
  for (int w = 1; w < 8; ++w) m = fmaxf(m, red[w]);
  __syncthreads();
  float sum = 0.f;
  for (int p = tid; p < HW; p += 512) sum += expf(l[p] - m);
  sum = wave_sum(sum);
  if (lane == 0) red[8 + wave] = sum;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int w = 0; w < 8; ++w) tot += red[8 + w];
  const float inv = 1.f / tot;
  float acc = 0.f;  // thread = channel (C <= 512)
  for (int p0 = 0; p0 < HW; p0 += 512) {
    __syncthreads();
    if (p0 + tid < HW) wts[tid] = expf(l[p0 + tid] - m) * inv;
    __syncthreads();
    const int n = HW - p0 < 512 ? HW - p0 : 512;
    if (tid < C)
      for (int p = 0; p < n; ++p) acc = fmaf(wts[p], x[((size_t)b * HW + p0 + p) * C + tid], acc);
  }
  if (tid < C) ctx[(size_t)b * C + tid] = acc;
}
// y[b] = fc2(relu(LayerNorm(fc1(ctx[b]))));  one block per image, thread = output channel (C <= 512)
__global__ __launch_bounds__(512) void gc_mlp_kernel(const float* __restrict__ ctx, const float* __restrict__ w1,
                                                     const float* __restrict__ b1, const float* __restrict__ g,
                                                     const float* __restrict__ be, const float* __restrict__ w2,
                                                     const float* __restrict__ b2, float* __restrict__ y, int C) {
  __shared__ float v[512], h[512], red[16];
  const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  if (tid < C) v[tid] = ctx[(size_t)b * C + tid];
  __syncthreads();
  float a = 0.f;
  if (tid < C) {
    a = b1[tid];
    for (int k = 0; k < C; ++k) a = fmaf(v[k], w1[(size_t)tid * C + k], a);
  }
  float s = wave_sum(tid < C ? a : 0.f);
  if (lane == 0) red[wave] = s;
  __syncthreads();
  float mean = 0.f;
#pragma unroll
  for (int w = 0; w < 8; ++w) mean += red[w];
  mean /= C;
  const float d = tid < C ? a - mean : 0.f;
  float q = wave_sum(d * d);
  if (lane == 0) red[8 + wave] = q;
  __syncthreads();
  float var = 0.f;
#pragma unroll
  for (int w = 0; w < 8; ++w) var += red[8 + w];
  const float rstd = 1.f / sqrtf(var / C + 1e-5f);
  if (tid < C) h[tid] = fmaxf(d * rstd * g[tid] + be[tid], 0.f);
  __syncthreads();
  if (tid < C) {
    float o = b2[tid];
    for (int k = 0; k < C; ++k) o = fmaf(h[k], w2[(size_t)tid * C + k], o);
    y[(size_t)b * C + tid] = o;
  }
}
__global__ void gc_add_kernel(float* __restrict__ x, const float* __restrict__ y, size_t n4, int HW, int C) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const size_t e = i * 4;
    const int c = (int)(e % C);
    const size_t b = e / ((size_t)HW * C);
    float4 v = reinterpret_cast<float4*>(x)[i];
    const float4 a = *reinterpret_cast<const float4*>(y + b * C + c);
    v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
    reinterpret_cast<float4*>(x)[i] = v;
  }
}
hipError_t launch_gc_logits(const float* x, const float* wg, const float* bg, float* logits, long long rows, int C, hipStream_t s) {
  hipLaunchKernelGGL(gc_logits_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, x, wg, bg, logits, rows, C);
  return hipGetLastError();
}
hipError_t launch_gc_pool(const float* x, const float* logits, float* ctx, int B, int HW, int C, hipStream_t s) {
  if (C > 512) return hipErrorInvalidValue;
  hipLaunchKernelGGL(gc_pool_kernel, dim3(B), dim3(512), 0, s, x, logits, ctx, HW, C);
  return hipGetLastError();
}
hipError_t launch_global_context(float* x, const GCParams& w, float* logits, float* ctx, float* y, int B, int HW, int C,
                                 hipStream_t s) {
  if (C > 512 || C % 4) return hipErrorInvalidValue;
  const long long rows = (long long)B * HW;
  hipLaunchKernelGGL(gc_logits_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, x, w.wg, w.bg, logits, rows, C);
  hipLaunchKernelGGL(gc_pool_kernel, dim3(B), dim3(512), 0, s, x, logits, ctx, HW, C);
  hipLaunchKernelGGL(gc_mlp_kernel, dim3(B), dim3(512), 0, s, ctx, w.w1, w.b1, w.ln_g, w.ln_b, w.w2, w.b2, y, C);
  const size_t n4 = (size_t)rows * C / 4;
  hipLaunchKernelGGL(gc_add_kernel, dim3((unsigned)std::min<size_t>((n4 + 255) / 256, 1u << 20)), dim3(256), 0, s, x, y, n4,
                     HW, C);
  return hipGetLastError();
}

}  // namespace d2t
