// Decode-step kernels (gfx950, fp32): everything the KV-cached greedy / beam loop
// launches once per generated token.  Replaces, for the newest position only, what
// nn.TransformerDecoder recomputes over the whole prefix at every step in the
// reference (prediction_head/tfm.py:125-140).  All position-dependent values are
// read from a device-side step counter so one captured hipGraph replays for every step.
#include <cstdlib>

#include "kernels.h"

namespace d2t {

using f32x4 = __attribute__((ext_vector_type(4))) float;
#ifndef D2T_DECODE_PRIO
#define D2T_DECODE_PRIO 3
#endif

// The decode step is a dependent chain of small latency-bound kernels that runs BESIDE the next batches' convolutions
// (pipelined serving).  Instruction issue on a SIMD is arbitrated by priority, then age (MI355X_MICROARCH.md "Two waves per
// SIMD" item 2): a decode wave is always younger than the persistent convolution waves it shares the SIMD with and would
// get only the issue slots they leave.  Its few instructions cost the matrix-bound convolution next to nothing, so the
// decode kernels simply outrank it.
__device__ __forceinline__ void decode_wave_priority() { __builtin_amdgcn_s_setprio(D2T_DECODE_PRIO); }

// debug timeline (D2T_DECODE_TRACE): every block folds its own start / end time into the launch's record
struct TraceScope {
  unsigned long long* t;
  __device__ __forceinline__ explicit TraceScope(unsigned long long* tr) : t(tr) {
    if (t && threadIdx.x == 0) atomicMin(t, (unsigned long long)__builtin_amdgcn_s_memrealtime());
  }
  __device__ __forceinline__ ~TraceScope() {
    if (t) {
      __syncthreads();
      if (threadIdx.x == 0) atomicMax(t + 1, (unsigned long long)__builtin_amdgcn_s_memrealtime());
    }
  }
};

__device__ __forceinline__ float act_fn(float v, int act) {
  if (act == ACT_RELU) return fmaxf(v, 0.f);
  if (act == ACT_GELU) return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
  return v;
}

// ---------------------------------------------------------------------------
// Skinny GEMM  y[M,N] = act(A @ w[N,K]^T + bias + res),  A = x or LayerNorm(x).
// M <= 64 rows per block row (the decode batch), weights streamed from L2 straight
// into registers.  Block tile 64 x 16; the K dimension is split over the NW waves
// of the block (each wave: all 64 rows x its K-slice, v_mfma_f32_16x16x4_f32, every
// load issued before the first MFMA), partial tiles are summed through LDS.
// The optional LayerNorm prologue (post-norm decoder: the previous sub-layer's
// LayerNorm) is evaluated on the register-resident rows: two-pass mean / variance
// with a cross-wave LDS reduction, so no standalone LayerNorm launch is needed.
// ---------------------------------------------------------------------------
template <int NW, int NCH, int MT>  // K == NW * NCH * 16; block tile (16*MT) rows x 16 columns
__global__ __launch_bounds__(NW * 64) void skinny_splitk_kernel(const SkinnyP p) {
  if (p.stop_at && *p.stop_at && *p.cur_step >= *p.stop_at) return;  // block-uniform
  decode_wave_priority();
  TraceScope trace_(p.trace);
  constexpr int KS = NCH * 16;
  constexpr int ROWS = 16 * MT;
  __shared__ float part[NW][ROWS][17];
  __shared__ float stat[NW][ROWS];
  __shared__ float s_mean[ROWS], s_rstd[ROWS];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r = lane & 15, q = lane >> 4;
  const int mbase = blockIdx.y * ROWS;
  const int n = blockIdx.x * 16 + r;
  const bool nok = n < p.N;
  const int k0 = wave * KS + q * 4;

  float4 a[MT][NCH], b[NCH];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int m = mbase + i * 16 + r;
    const bool mok = m < p.M;
    const float* src = p.x + (size_t)(mok ? m : 0) * p.ldx + k0;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      a[i][c] = *reinterpret_cast<const float4*>(src + c * 16);
      if (!mok) a[i][c] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  {
    const float* src = p.w + (size_t)(nok ? n : 0) * (p.wld ? p.wld : p.K) + k0;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      b[c] = *reinterpret_cast<const float4*>(src + c * 16);
      if (!nok) b[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }

  if (p.ln_g) {
    const float invK = 1.f / (float)p.K;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < NCH; ++c) s += (a[i][c].x + a[i][c].y) + (a[i][c].z + a[i][c].w);
      s += __shfl_xor(s, 16, 64);
      s += __shfl_xor(s, 32, 64);
      if (q == 0) stat[wave][i * 16 + r] = s;
    }
    __syncthreads();
    if (tid < ROWS) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) t += stat[w][tid];
      s_mean[tid] = t * invK;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const float mu = s_mean[i * 16 + r];
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        a[i][c].x -= mu; a[i][c].y -= mu; a[i][c].z -= mu; a[i][c].w -= mu;
        s += (a[i][c].x * a[i][c].x + a[i][c].y * a[i][c].y) + (a[i][c].z * a[i][c].z + a[i][c].w * a[i][c].w);
      }
      s += __shfl_xor(s, 16, 64);
      s += __shfl_xor(s, 32, 64);
      if (q == 0) stat[wave][i * 16 + r] = s;
    }
    __syncthreads();
    if (tid < ROWS) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) t += stat[w][tid];
      s_rstd[tid] = 1.f / sqrtf(t * invK + p.ln_eps);
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const float4 g4 = *reinterpret_cast<const float4*>(p.ln_g + k0 + c * 16);
      const float4 b4 = *reinterpret_cast<const float4*>(p.ln_b + k0 + c * 16);
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const float rs = s_rstd[i * 16 + r];
        a[i][c].x = a[i][c].x * rs * g4.x + b4.x;
        a[i][c].y = a[i][c].y * rs * g4.y + b4.y;
        a[i][c].z = a[i][c].z * rs * g4.z + b4.z;
        a[i][c].w = a[i][c].w * rs * g4.w + b4.w;
      }
    }
  }

  f32x4 acc[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][c].x, b[c].x, acc[i], 0, 0, 0);
      acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][c].y, b[c].y, acc[i], 0, 0, 0);
      acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][c].z, b[c].z, acc[i], 0, 0, 0);
      acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][c].w, b[c].w, acc[i], 0, 0, 0);
    }
  }
  // C/D map: col = lane&15 -> n, row = (lane>>4)*4 + reg -> m
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) part[wave][i * 16 + q * 4 + reg][r] = acc[i][reg];
  __syncthreads();

  float* y = p.y;
  if (p.step_ptr) y += (long long)(*p.step_ptr) * p.out_step_stride;
  for (int idx = tid; idx < ROWS * 16; idx += NW * 64) {
    const int row = idx >> 4, col = idx & 15;
    const int m = mbase + row, nn = blockIdx.x * 16 + col;
    if (m >= p.M || nn >= p.N) continue;
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) v += part[w][row][col];
    v += p.bias ? p.bias[nn] : 0.f;
    if (p.res) v += p.res[(size_t)m * p.ldres + nn];
    y[(size_t)m * p.ldy + nn] = act_fn(v, p.act);
    if (p.ln_out && nn < p.K) {
      const float xv = p.x[(size_t)m * p.ldx + nn];
      p.ln_out[(size_t)m * p.K + nn] = (xv - s_mean[row]) * s_rstd[row] * p.ln_g[nn] + p.ln_b[nn];
    }
  }
}

// generic fallback (any K % 16 == 0, no LayerNorm prologue): one wave per 16 rows
__global__ __launch_bounds__(256) void skinny_generic_kernel(const SkinnyP p) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r = lane & 15, q = lane >> 4;
  const int m = blockIdx.y * 64 + wave * 16 + r;
  const int n = blockIdx.x * 16 + r;
  const bool mok = m < p.M, nok = n < p.N;
  const float* xa = p.x + (size_t)(mok ? m : 0) * p.ldx + q * 4;
  const float* wb = p.w + (size_t)(nok ? n : 0) * p.K + q * 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const int KC = p.K >> 4;
  for (int c = 0; c < KC; ++c) {
    float4 a = *reinterpret_cast<const float4*>(xa + c * 16);
    float4 b = *reinterpret_cast<const float4*>(wb + c * 16);
    if (!mok) a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!nok) b = make_float4(0.f, 0.f, 0.f, 0.f);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc, 0, 0, 0);
  }
  const int nn = blockIdx.x * 16 + r;
  if (nn >= p.N) return;
  float* y = p.y;
  if (p.step_ptr) y += (long long)(*p.step_ptr) * p.out_step_stride;
  const float bias = p.bias ? p.bias[nn] : 0.f;
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) {
    const int mm = blockIdx.y * 64 + wave * 16 + q * 4 + reg;
    if (mm >= p.M) continue;
    float v = acc[reg] + bias;
    if (p.res) v += p.res[(size_t)mm * p.ldres + nn];
    y[(size_t)mm * p.ldy + nn] = act_fn(v, p.act);
  }
}

// small = 1: the 256-thread forms of the decode kernels (one wave per SIMD, <= 128 VGPRs), which fit on a CU beside the
// pipelined convolution kernel's three 128-register waves per SIMD; identical arithmetic per row, so identical results
static int decode_small() {
  static const int v = D2T_PROBE_ENV("D2T_DECODE_SMALL");
  return v;
}

hipError_t launch_skinny(const SkinnyP& p, hipStream_t s) {
  if (p.M <= 0 || p.N <= 0) return hipSuccess;
  if (p.K % 16 != 0 || p.ldx % 4 != 0) return hipErrorInvalidValue;
  // narrow outputs (N <= 512) take 16-row tiles so that the few column tiles still fill >= 64 CUs
  const int mt = p.N <= 512 ? 1 : 2;
  const dim3 grid((p.N + 15) / 16, (p.M + 16 * mt - 1) / (16 * mt));
#define D2T_SK(NW, NCH)                                                                                  \
  do {                                                                                                   \
    if (mt == 1) hipLaunchKernelGGL((skinny_splitk_kernel<NW, NCH, 1>), grid, dim3(NW * 64), 0, s, p);   \
    else hipLaunchKernelGGL((skinny_splitk_kernel<NW, NCH, 2>), grid, dim3(NW * 64), 0, s, p);           \
  } while (0)
  if (p.K == 256) D2T_SK(4, 4);
  else if (p.K == 512) D2T_SK(8, 4);
  else if (p.K == 1024 && decode_small() && !p.ln_g && !p.step_ptr) {
    // two 256-thread passes over the halves of K (second pass accumulates onto the first through the residual input)
    SkinnyP a = p, b = p;
    a.K = 512; a.ldx = p.ldx; a.w = p.w; a.wld = p.K;
    b.K = 512; b.x = p.x + 512; b.w = p.w + 512; b.wld = p.K; b.bias = nullptr; b.res = p.y; b.ldres = p.ldy;
    a.act = ACT_NONE;
    if (p.act != ACT_NONE) return hipErrorInvalidValue;
    hipLaunchKernelGGL((skinny_splitk_kernel<4, 8, 1>), dim3((p.N + 15) / 16, (p.M + 15) / 16), dim3(256), 0, s, a);
    hipLaunchKernelGGL((skinny_splitk_kernel<4, 8, 1>), dim3((p.N + 15) / 16, (p.M + 15) / 16), dim3(256), 0, s, b);
  }
  else if (p.K == 1024) D2T_SK(8, 8);
  else {
    if (p.ln_g) return hipErrorInvalidValue;
    hipLaunchKernelGGL(skinny_generic_kernel, dim3((p.N + 15) / 16, (p.M + 63) / 64), dim3(256), 0, s, p);
  }
#undef D2T_SK
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Single-query multi-head attention for one decode step: one block (4 waves) per
// (row, head).  K and V rows of a head are contiguous, so a wave-wide 16-B load
// covers 64/(HD/4) whole keys: fully coalesced.  Each wave takes every 4th group of
// keys, keeps a wave-local (max, sum, weighted V) triple and the four triples are
// merged through LDS (log-sum-exp combine).  Self-attention mode appends this
// step's k,v to the cache.  Replaces nn.MultiheadAttention inside
// nn.TransformerDecoderLayer (tfm.py:130) for the newest position.
// ---------------------------------------------------------------------------
template <int HD>
__global__ __launch_bounds__(256) void decode_attention_kernel(const DecAttnP p) {
  constexpr int LPK = HD / 4;          // lanes per key
  constexpr int KPI = 64 / LPK;        // keys per wave-iteration
  constexpr int MAXIT = 512 / (KPI * 4);
  __shared__ float w_m[4], w_s[4];
  __shared__ __attribute__((aligned(16))) float w_acc[4][HD];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int kig = lane / LPK, ch = lane % LPK;
  const int b = blockIdx.x / p.heads, head = blockIdx.x % p.heads;
  int L = p.L, t = -1;
  if (p.step_ptr) { t = *p.step_ptr; L = t + 1; }
  float* Kc = p.k + (size_t)b * p.kv_batch_stride + (size_t)head * p.Lmax * HD;
  float* Vc = p.v + (size_t)b * p.kv_batch_stride + (size_t)head * p.Lmax * HD;
  const float* curk = p.cur_k ? p.cur_k + (size_t)b * p.cur_stride + head * HD : nullptr;
  const float* curv = p.cur_v ? p.cur_v + (size_t)b * p.cur_stride + head * HD : nullptr;
  if (curk && t >= 0 && tid < HD) {  // append this step's k,v to the cache
    Kc[(size_t)t * HD + tid] = curk[tid];
    Vc[(size_t)t * HD + tid] = curv[tid];
  }
  const float4 q4 = *reinterpret_cast<const float4*>(p.q + (size_t)b * p.q_stride + head * HD + ch * 4);
  const float scale = HD == 32 ? 0.17677669529663687f : 0.125f;
  const int nit = (L + KPI * 4 - 1) / (KPI * 4);  // iterations of this block (wave-uniform)

  float sc[MAXIT];
  float mloc = -INFINITY;
#pragma unroll
  for (int it = 0; it < MAXIT; ++it) {
    sc[it] = -INFINITY;
    if (it < nit) {
      const int j = (it * 4 + wave) * KPI + kig;
      float d = 0.f;
      if (j < L) {
        const float* kr = (curk && j == t) ? curk : Kc + (size_t)j * HD;
        const float4 k4 = *reinterpret_cast<const float4*>(kr + ch * 4);
        d = (q4.x * k4.x + q4.y * k4.y) + (q4.z * k4.z + q4.w * k4.w);
      }
#pragma unroll
      for (int o = 1; o < LPK; o <<= 1) d += __shfl_xor(d, o, 64);
      if (j < L) {
        sc[it] = d * scale;
        mloc = fmaxf(mloc, sc[it]);
      }
    }
  }
#pragma unroll
  for (int o = LPK; o < 64; o <<= 1) mloc = fmaxf(mloc, __shfl_xor(mloc, o, 64));
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  float ssum = 0.f;
#pragma unroll
  for (int it = 0; it < MAXIT; ++it) {
    if (it < nit) {
      const int j = (it * 4 + wave) * KPI + kig;
      if (j < L) {
        const float pj = expf(sc[it] - mloc);
        const float* vr = (curv && j == t) ? curv : Vc + (size_t)j * HD;
        const float4 v4 = *reinterpret_cast<const float4*>(vr + ch * 4);
        ssum += pj;
        acc.x = fmaf(pj, v4.x, acc.x); acc.y = fmaf(pj, v4.y, acc.y);
        acc.z = fmaf(pj, v4.z, acc.z); acc.w = fmaf(pj, v4.w, acc.w);
      }
    }
  }
#pragma unroll
  for (int o = LPK; o < 64; o <<= 1) {
    ssum += __shfl_xor(ssum, o, 64);
    acc.x += __shfl_xor(acc.x, o, 64); acc.y += __shfl_xor(acc.y, o, 64);
    acc.z += __shfl_xor(acc.z, o, 64); acc.w += __shfl_xor(acc.w, o, 64);
  }
  if (kig == 0) *reinterpret_cast<float4*>(&w_acc[wave][ch * 4]) = acc;
  if (lane == 0) { w_m[wave] = mloc; w_s[wave] = ssum; }
  __syncthreads();
  if (tid < HD) {
    const float m = fmaxf(fmaxf(w_m[0], w_m[1]), fmaxf(w_m[2], w_m[3]));
    float tot = 0.f, o = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float f = w_s[w] > 0.f ? expf(w_m[w] - m) : 0.f;  // waves without keys contribute nothing
      tot += w_s[w] * f;
      o += w_acc[w][tid] * f;
    }
    p.y[(size_t)b * p.y_stride + head * HD + tid] = o / tot;
  }
}

hipError_t launch_decode_attention(const DecAttnP& p, hipStream_t s) {
  const int Lcap = p.step_ptr ? p.Lmax : p.L;
  if (Lcap > 512 || Lcap < 1) return hipErrorInvalidValue;
  const dim3 grid(p.B * p.heads);
  if (p.hd == 32) hipLaunchKernelGGL(decode_attention_kernel<32>, grid, dim3(256), 0, s, p);
  else if (p.hd == 64) hipLaunchKernelGGL(decode_attention_kernel<64>, grid, dim3(256), 0, s, p);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// _embedd_tgt (tfm.py:86-94) for the newest position: Embedding * sqrt(d) + WordPosEnc row t.
// ---------------------------------------------------------------------------
__global__ void embed_kernel(const float* __restrict__ emb, const float* __restrict__ pe,
                             const int64_t* __restrict__ start, const int64_t* __restrict__ tokens, int tok_stride,
                             const int* __restrict__ step_ptr, float* __restrict__ x, int d, float sqrt_d) {
  const int b = blockIdx.x, t = *step_ptr;
  const int64_t tok = t == 0 ? start[b] : tokens[(size_t)b * tok_stride + t - 1];
  for (int c = threadIdx.x; c < d; c += blockDim.x)
    x[(size_t)b * d + c] = emb[(size_t)tok * d + c] * sqrt_d + pe[(size_t)t * d + c];
}

hipError_t launch_embed(const float* emb, const float* pe, const int64_t* start, const int64_t* tokens,
                        int tok_stride, const int* step_ptr, float* x, int B, int d, hipStream_t s) {
  hipLaunchKernelGGL(embed_kernel, dim3(B), dim3(256), 0, s, emb, pe, start, tokens, tok_stride, step_ptr, x, d,
                     sqrtf((float)d));
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Greedy next token (tfm.py:134-139) for every row + the next step's input
// embedding + step counter increment, one block.  First maximum wins ties;
// end-of-sequence bookkeeping stays on the device.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void argmax_embed_kernel(const ArgmaxP p) {
  const int t = *p.step_ptr;
  if (p.stop_at && *p.stop_at && t >= *p.stop_at) return;  // block-uniform; set by an EARLIER launch only (t + 1 > t)
  decode_wave_priority();
  TraceScope trace_(p.trace);
  const int b = blockIdx.x, lane = threadIdx.x;
  const float* row = p.logits + (size_t)b * p.row_stride + (size_t)t * p.step_stride;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int i = lane; i < p.V; i += 64) {
    const float v = row[i];
    if (v > best || (v == best && i < bi)) { best = v; bi = i; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  if (bi >= p.V) bi = 0;  // all-NaN row: keep indexing in range
  if (p.x) {
    const float sqrt_d = sqrtf((float)p.d);
    const float* e = p.emb + (size_t)bi * p.d;
    const float* pe = p.pe + (size_t)(t + 1) * p.d;
    for (int c = lane; c < p.d; c += 64) p.x[(size_t)b * p.d + c] = e[c] * sqrt_d + pe[c];
  }
  if (lane == 0) {
    p.tokens[(size_t)b * p.tok_stride + t] = bi;
    if (bi == p.end_token && !p.ended[b]) {
      p.ended[b] = 1;
      const int c = atomicAdd(p.end_count, 1) + 1;
      if (c == p.B) *p.steps_done = t + 1;
      if (p.n_batches > 0) {  // decode group: this row's encoder batch may have finished
        const int k = b / p.rows_per_batch;
        if (atomicAdd(p.batch_end_count + k, 1) + 1 == p.rows_per_batch) {
          p.batch_steps_done[k] = t + 1;
          // every block of THIS launch read step t before the stop takes effect (t < t + 1); the kernels of step t + 1 see it
          if (atomicAdd(p.batches_done, 1) + 1 == p.n_batches && p.stop_at) *p.stop_at = t + 1;
        }
      }
    }
    // every block has read the step counter before it arrives here, so the
    // last arriver may advance it (the kernel boundary publishes the store)
    __threadfence();
    if (atomicAdd(p.done_count, 1) == p.B - 1) {
      *p.done_count = 0;
      *p.step_ptr = t + 1;
    }
  }
}

hipError_t launch_argmax_embed(const ArgmaxP& p, hipStream_t s) {
  hipLaunchKernelGGL(argmax_embed_kernel, dim3(p.B), dim3(64), 0, s, p);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Fused row kernel (see DecRowP).  512 threads = 8 waves = 8 heads.
// ---------------------------------------------------------------------------
template <int HD, int U>  // U key groups fetched together: 2*U 16-B loads in flight per lane
__device__ __forceinline__ void row_attention(const float* q, const float* Kc, const float* Vc, const float* curk,
                                              const float* curv, int t, int L, float* out, int lane,
                                              const int* anc = nullptr, long long anc_stride = 0) {
  // anc != nullptr (beam search, DecRowP::anc): position j < t of this hypothesis lives in cache row anc[j]; Kc / Vc then
  // point at cache row 0 of the head and anc_stride is the cache's row stride
  constexpr int LPK = HD / 4, KPI = 64 / LPK;
  const int kig = lane / LPK, ch = lane % LPK;
  const float4 q4 = *reinterpret_cast<const float4*>(q + ch * 4);
  const float scale = HD == 32 ? 0.17677669529663687f : 0.125f;
  float m = -INFINITY, l = 0.f;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const int nit = (L + KPI - 1) / KPI;
  for (int it0 = 0; it0 < nit; it0 += U) {
    float4 k4[U], v4[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int j = (it0 + u) * KPI + kig;
      const int jj = j < L ? j : 0;
      const size_t ro = (anc && jj != t) ? (size_t)anc[jj] * (size_t)anc_stride : 0;
      const float* kr = (curk && jj == t) ? curk : Kc + ro + (size_t)jj * HD;
      const float* vr = (curv && jj == t) ? curv : Vc + ro + (size_t)jj * HD;
      k4[u] = *reinterpret_cast<const float4*>(kr + ch * 4);
      v4[u] = *reinterpret_cast<const float4*>(vr + ch * 4);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int j = (it0 + u) * KPI + kig;
      float d = (q4.x * k4[u].x + q4.y * k4[u].y) + (q4.z * k4[u].z + q4.w * k4[u].w);
#pragma unroll
      for (int o = 1; o < LPK; o <<= 1) d += __shfl_xor(d, o, 64);
      if (j < L) {
        const float sj = d * scale;
        const float mn = fmaxf(m, sj);
        const float f = expf(m - mn);  // exp(-inf) = 0 on the first key
        const float pj = expf(sj - mn);
        l = l * f + pj;
        acc.x = acc.x * f + pj * v4[u].x; acc.y = acc.y * f + pj * v4[u].y;
        acc.z = acc.z * f + pj * v4[u].z; acc.w = acc.w * f + pj * v4[u].w;
        m = mn;
      }
    }
  }
#pragma unroll
  for (int o = LPK; o < 64; o <<= 1) {  // merge the key groups (log-sum-exp combine)
    const float mo = __shfl_xor(m, o, 64), lo = __shfl_xor(l, o, 64);
    const float ax = __shfl_xor(acc.x, o, 64), ay = __shfl_xor(acc.y, o, 64);
    const float az = __shfl_xor(acc.z, o, 64), aw = __shfl_xor(acc.w, o, 64);
    const float mn = fmaxf(m, mo);
    const float f1 = l > 0.f ? expf(m - mn) : 0.f, f2 = lo > 0.f ? expf(mo - mn) : 0.f;
    l = l * f1 + lo * f2;
    acc.x = acc.x * f1 + ax * f2; acc.y = acc.y * f1 + ay * f2;
    acc.z = acc.z * f1 + az * f2; acc.w = acc.w * f1 + aw * f2;
    m = mn;
  }
  if (kig == 0) {
    const float inv = 1.f / l;
    *reinterpret_cast<float4*>(out + ch * 4) = make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
  }
}

// out_part[g][n] = sum_{k in group g} in[k] * Wt[k][n]; caller reduces over g after a barrier
template <int D, int NTH>
__device__ __forceinline__ void row_gemv(const float* in_s, const float* __restrict__ Wt, float* part_s, int tid) {
  constexpr int LPR = D / 4, G = NTH / LPR, KG = D / G;
  const int lr = tid % LPR, g = tid / LPR;
  const float* w = Wt + (size_t)(g * KG) * D + lr * 4;
  const float* in = in_s + g * KG;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int k = 0; k < KG; ++k) {
    const float4 w4 = *reinterpret_cast<const float4*>(w + (size_t)k * D);
    const float a = in[k];
    acc.x = fmaf(a, w4.x, acc.x); acc.y = fmaf(a, w4.y, acc.y);
    acc.z = fmaf(a, w4.z, acc.z); acc.w = fmaf(a, w4.w, acc.w);
  }
  *reinterpret_cast<float4*>(part_s + g * D + lr * 4) = acc;
}

template <int D, int HD, int NTH>  // NTH = 512: one wave per head; 256: four waves, two heads each (<= 128 VGPRs, so that a
                                    // block fits on a CU next to the pipelined convolution's three waves per SIMD)
__global__ __launch_bounds__(NTH, NTH == 256 ? 4 : 2) void decoder_row_kernel(const DecRowP p) {
  constexpr int G = NTH / (D / 4);
  constexpr int HPW = 8 * 64 / NTH;  // heads per wave
  if (p.stop_at && *p.stop_at && *p.step_ptr >= *p.stop_at) return;  // block-uniform
  decode_wave_priority();
  TraceScope trace_(p.trace);
  __shared__ __attribute__((aligned(16))) float a_s[D], y_s[D], x1_s[D], q2_s[D], part_s[G * D];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int b = blockIdx.x;
  const int t = *p.step_ptr;
  const float* qkv = p.qkv + (size_t)b * p.qkv_stride;
  // ---- self-attention, one head per wave (and pass) ----
#pragma unroll
  for (int hp = 0; hp < HPW; ++hp) {
    const int head = wave + hp * (NTH / 64);
    float* Kc = p.sk + (size_t)b * p.s_batch_stride + (size_t)head * p.s_Lmax * HD;
    float* Vc = p.sv + (size_t)b * p.s_batch_stride + (size_t)head * p.s_Lmax * HD;
    const float* curk = qkv + D + head * HD;
    const float* curv = qkv + 2 * D + head * HD;
    if (lane < HD) {
      Kc[(size_t)t * HD + lane] = curk[lane];
      Vc[(size_t)t * HD + lane] = curv[lane];
    }
    row_attention<HD, 4>(qkv + head * HD, Kc, Vc, curk, curv, t, t + 1, a_s + head * HD, lane);
  }
  __syncthreads();
  row_gemv<D, NTH>(a_s, p.wo_t, part_s, tid);
  __syncthreads();
  if (tid < D) {
    float v = p.bo[tid] + p.xres[(size_t)b * D + tid];
#pragma unroll
    for (int g = 0; g < G; ++g) v += part_s[g * D + tid];
    y_s[tid] = v;
  }
  __syncthreads();
  if (wave == 0) {  // LN1, two-pass, one wave
    constexpr int V = D / 64;
    float v[V], s = 0.f;
#pragma unroll
    for (int i = 0; i < V; ++i) { v[i] = y_s[i * 64 + lane]; s += v[i]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float mean = s * (1.f / D);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < V; ++i) { v[i] -= mean; q += v[i] * v[i]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
    const float rstd = 1.f / sqrtf(q * (1.f / D) + p.eps);
#pragma unroll
    for (int i = 0; i < V; ++i) {
      const int c = i * 64 + lane;
      x1_s[c] = v[i] * rstd * p.ln1_g[c] + p.ln1_b[c];
    }
  }
  __syncthreads();
  row_gemv<D, NTH>(x1_s, p.wq_t, part_s, tid);
  __syncthreads();
  if (tid < D) {
    float v = p.bq[tid];
#pragma unroll
    for (int g = 0; g < G; ++g) v += part_s[g * D + tid];
    q2_s[tid] = v;
  }
  __syncthreads();
  // ---- cross-attention over the memory K/V, one head per wave (and pass) ----
#pragma unroll
  for (int hp = 0; hp < HPW; ++hp) {
    const int head = wave + hp * (NTH / 64);
    const int cb = p.c_row_map ? p.c_row_map[b] : b;
    const float* Kc = p.ck + (size_t)cb * p.c_batch_stride + (size_t)head * p.T * HD;
    const float* Vc = p.cv + (size_t)cb * p.c_batch_stride + (size_t)head * p.T * HD;
    row_attention<HD, 8>(q2_s + head * HD, Kc, Vc, nullptr, nullptr, -1, p.T, a_s + head * HD, lane);
  }
  __syncthreads();
  row_gemv<D, NTH>(a_s, p.wco_t, part_s, tid);
  __syncthreads();
  if (tid < D) {
    float v = p.bco[tid] + x1_s[tid];
#pragma unroll
    for (int g = 0; g < G; ++g) v += part_s[g * D + tid];
    p.y2[(size_t)b * D + tid] = v;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Row kernel, round 3: cross-attention over the encoder MEMORY itself ("absorbed" projections).
//
// The kernel above streams, per row and layer, the projected K and V of its sample: 2 * T * D floats (534 KB at T = 261,
// D = 256) that are different for every layer -- 1.23 GB per decode step at 384 rows, the whole cost of the step.  But
//     score_h[j] = q_h . (W_k,h m_j + b_k,h) = (W_k,h^T q_h) . m_j + const_h          (the constant cancels in the softmax)
//     out_h      = sum_j p_hj (W_v,h m_j + b_v,h) = W_v,h (sum_j p_hj m_j) + b_v,h     (sum_j p_hj = 1)
// so a head can attend over the D-wide memory rows directly with the absorbed query q'_h = W_k,h^T q_h (D floats per head),
// and project the attention-weighted memory row afterwards.  Per row and layer that reads T * D floats (267 KB) -- the SAME
// bytes for all six layers, 102 MB for 384 rows: resident in the 256 MB Infinity Cache for the whole step loop -- and the
// 2 x 205 MB projected-K/V buffers, their GEMM per batch and the per-layer streams disappear.  The price is arithmetic:
// 8 heads x T x D multiply-adds for the scores and again for the weighted sum (8 x the head-dim form).  It goes to the
// matrix cores in exact fp32 (v_mfma_f32_16x16x4_f32, an fp32 fma chain per output like the VALU code it replaces):
//     S^T [16 keys x 16 (8 heads + 8 idle)] = M_tile [16 x D] . Q'^T [D x 16]            64 MFMAs per 16-key tile
//     ctx [16 (8 heads + 8 idle) x D]      += P [16 x 16 keys] . M_tile [16 x D]          64 MFMAs per tile
// A wave owns the key tiles w, w + NW, ...: it copies a tile (16 rows x 1 KB) into its private 16 KB of LDS with LDS-DMA
// (the 16-byte chunks of row i XOR-ed with i on the source side, so that both fragment patterns -- 16 keys x 4 channel
// groups for the scores, 4 keys x 16 channel groups for the weighted sum -- read without bank conflicts:
// tools/probe/lds_swizzle_check.py), keeps an online softmax per head (running max and sum, flash-decoding), and the waves'
// partial (max, sum, ctx) triples are merged through LDS.  The S^T product is taken transposed on purpose: its result
// layout (lane = (key group g, head), registers = keys 4g..4g+3) IS the A-operand layout of the second product with
// k-step s <-> register s, so P never moves between lanes.
// Everything else (self-attention over the cache, the three row GEMVs, LN1) is the code of decoder_row_kernel.
// ---------------------------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void* lds_ptr_dec;

// this wave's share of the cross-attention of ONE query row set: heads x T keys of `mem` [T][256].
// qp_s: LDS [8][256] absorbed queries (already scaled by 1/sqrt(head_dim); the 16-byte chunks of row h XOR-ed with h like
// the tile rows); stage: this wave's 16 KB; results: the wave's
// running max / sum per head (lanes with col < 8, reduced over the key groups) and ctx accumulators acc[w][e] (D layout).
template <int NW>
__device__ __forceinline__ void cross_absorbed_wave(const float* __restrict__ mem, int T, const float* qp_s, unsigned char* stage,
                                                    int wave, int lane, float& m_run, float& l_run, f32x4 (&acc)[4][4]) {
  const int col = lane & 15, g = lane >> 4;
  // B operand of the score product: q'[head = col][16u + 4g + t], re-read from LDS per tile (16 reads against 128 MFMAs; in
  // registers it would cost 64 VGPRs and the second block per CU).  The idle half of the MFMA tile (columns 8..15) repeats
  // heads 0..7: its results are never stored, and identical addresses broadcast.
  const int hrow = col & 7;
  m_run = -INFINITY;
  l_run = 0.f;
#pragma unroll
  for (int w = 0; w < 4; ++w)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[w][e] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int ntiles = (T + 15) >> 4;
  for (int tile = wave; tile < ntiles; tile += NW) {
    const int j0 = tile << 4;
    // ---- stage the tile: row i -> stage + i * 1024, physical 16-byte chunk p holds logical chunk p ^ i ----
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the previous tile's fragment reads have returned
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int j = j0 + i < T ? j0 + i : T - 1;  // rows past the end: a valid row, its probability is forced to zero
      const float* src = mem + (size_t)j * 256 + ((lane ^ i) << 2);
      __builtin_amdgcn_global_load_lds(src, (lds_ptr_dec)(stage + i * 1024), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // ---- S^T[key = 4g' + reg][head = col] = sum_c m[key][c] q'[head][c] ----
    f32x4 sacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      // A operand: lane (row = key col, k = g) -> m[key][16u + 4g + t]; logical chunk 4u + g of row `col`
      const float4 a4 = *reinterpret_cast<const float4*>(stage + col * 1024 + (((4 * u + g) ^ col) << 4));
      const float4 q4 = *reinterpret_cast<const float4*>(reinterpret_cast<const unsigned char*>(qp_s) + hrow * 1024 + (((4 * u + g) ^ hrow) << 4));
      sacc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, q4.x, sacc, 0, 0, 0);
      sacc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, q4.y, sacc, 0, 0, 0);
      sacc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, q4.z, sacc, 0, 0, 0);
      sacc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, q4.w, sacc, 0, 0, 0);
    }
    // ---- online softmax: this lane holds keys j0 + 4g + reg of head `col` ----
    float sv[4], mx = -INFINITY;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      sv[reg] = (j0 + 4 * g + reg < T) ? sacc[reg] : -INFINITY;
      mx = fmaxf(mx, sv[reg]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));  // a tile holds at least one valid key: finite
    const float m_new = fmaxf(m_run, mx);
    const float alpha = expf(m_run - m_new);  // exp(-inf) = 0 for the first tile
    float pv[4], ps = 0.f;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      pv[reg] = expf(sv[reg] - m_new);  // exp(-inf) = 0 for keys past the end
      ps += pv[reg];
    }
    l_run = l_run * alpha + ps;
    m_run = m_new;
    // the accumulators hold ctx[head = 4g + reg][...]: their scale is the alpha of THAT head (lane 4g + reg has it)
    float ar[4];
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) ar[reg] = __shfl(alpha, 4 * g + reg, 64);
#pragma unroll
    for (int w = 0; w < 4; ++w)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) acc[w][e][reg] *= ar[reg];
    // ---- ctx[head][64w + 4 col' + e] += sum_keys P[head][key] m[key][chan]; k-step s <-> keys 4g + s ----
#pragma unroll
    for (int sk = 0; sk < 4; ++sk) {
      const int key = 4 * g + sk;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        // B operand: lane (k = g, col) -> m[key][64w + 4 col + e]; logical chunk 16w + col of row `key`
        const float4 b4 = *reinterpret_cast<const float4*>(stage + key * 1024 + (((16 * w + col) ^ key) << 4));
        acc[w][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(pv[sk], b4.x, acc[w][0], 0, 0, 0);
        acc[w][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(pv[sk], b4.y, acc[w][1], 0, 0, 0);
        acc[w][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(pv[sk], b4.z, acc[w][2], 0, 0, 0);
        acc[w][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(pv[sk], b4.w, acc[w][3], 0, 0, 0);
      }
    }
  }
  l_run += __shfl_xor(l_run, 16, 64);
  l_run += __shfl_xor(l_run, 32, 64);
}

#ifdef D2T_PROBES
// probe builds: block 0 / thread 0 adds the time between consecutive marks (s_memrealtime, 10 ns ticks) to d2t_row_phase[k]
__device__ unsigned long long d2t_row_phase[32];
#define ROW_PHASE(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) { const unsigned long long now_ = __builtin_amdgcn_s_memrealtime(); \
    atomicAdd(&d2t_row_phase[k], now_ - phase_t_); phase_t_ = now_; } } while (0)
#define ROW_PHASE_INIT() unsigned long long phase_t_ = __builtin_amdgcn_s_memrealtime(); if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&d2t_row_phase[31], 1ull)
#else
#define ROW_PHASE(k) do { } while (0)
#define ROW_PHASE_INIT() do { } while (0)
#endif
#ifdef D2T_PROBES
#define WAVE_PHASE(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) { const unsigned long long now_ = __builtin_amdgcn_s_memrealtime(); \
    atomicAdd(&d2t_row_phase[k], now_ - wave_t_); wave_t_ = now_; } } while (0)
#define WAVE_PHASE_INIT() unsigned long long wave_t_ = __builtin_amdgcn_s_memrealtime()
#else
#define WAVE_PHASE(k) do { } while (0)
#define WAVE_PHASE_INIT() do { } while (0)
#endif
// ---------------------------------------------------------------------------------------------------------------------
// Cross-attention on split-bf16 MFMAs (round 4).  Inside the loop below the fp32-MFMA form is bound by the matrix pipe:
// 128 v_mfma_f32_16x16x4_f32 per tile at 32 cycles, two waves per SIMD, and only 8 of the 16 columns (heads) of every tile
// useful -- 15 of the loop's 28 us (probe build, tools/probe/row_phases.py).  The same products on the bf16 pipe, every fp32
// operand as hi + lo (three MFMAs per product, lo*hi + hi*lo + hi*hi: the arithmetic of the encoder's GEMMs):
//     S^T [16 keys x 16 (8 heads + 8 idle)] = M_tile [16 x 256] . Q'^T      8 K-steps x 3 v_mfma_f32_16x16x32_bf16   (24)
//     ctx [16 (8 heads + 8 idle) x 256]    += P [16 x 16 keys] . M_tile     16 column blocks x 3 v_mfma_f32_16x16x16_bf16 (48)
// 72 MFMAs of 16 cycles instead of 128 of 32.  The memory rows arrive as two bf16 planes (hi = upper 16 bits, lo = bf16(x - hi):
// x to 16 significant bits; launch_split_bf16 at cross_kv time); a tile = 16 rows x 512 B of hi | 16 x 512 B of lo in the wave's
// 16 KB stage, 16-byte chunk c of row r at position c ^ r (conflict-free ds_read_b128 of the score product's A operand; the
// weighted sum's B operand [keys x channels] comes out of the same image with ds_read_b64_tr_b16).  The absorbed queries
// are split the same way when they are stored.  The score product's result layout (lane = (key group, head), registers =
// keys 4g..4g+3) is the A-operand layout of the 16x16x16 form, so P never moves between lanes, as before.
// ---------------------------------------------------------------------------------------------------------------------
typedef __bf16 dbf16x8 __attribute__((ext_vector_type(8)));
typedef short ds4 __attribute__((ext_vector_type(4)));
typedef unsigned du32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) ds4* lds_ds4_ptr;
__device__ __forceinline__ void split16(float x, unsigned& hi, unsigned& lo) {  // hi = upper 16 bits, lo = bf16_rne(x - hi)
  const unsigned u = __float_as_uint(x);
  hi = u >> 16;
  const __bf16 l = (__bf16)(x - __uint_as_float(u & 0xFFFF0000u));
  lo = *reinterpret_cast<const unsigned short*>(&l);
}
// global source of the 16 bytes lane `lane` holds of piece i (0..15) of tile `tile`: pieces 0-7 = two hi rows each, 8-15 = lo
__device__ __forceinline__ const uint16_t* bx3_piece_src(const uint16_t* __restrict__ mh, const uint16_t* __restrict__ ml, int T, int tile,
                                                         int i, int lane) {
  const int row = 2 * (i & 7) + (lane >> 5);
  const int jr = (tile << 4) + row, j = jr < T ? jr : T - 1;  // rows past the end: a valid row, its probability is forced to zero
  return (i < 8 ? mh : ml) + (size_t)j * 256 + (((lane & 31) ^ row) << 3);
}
__device__ __forceinline__ void bx3_tile_dma(const uint16_t* __restrict__ mh, const uint16_t* __restrict__ ml, int T, int tile,
                                             unsigned char* stage, int lane) {
#pragma unroll
  for (int i = 0; i < 16; ++i)
    __builtin_amdgcn_global_load_lds(bx3_piece_src(mh, ml, T, tile, i, lane), (lds_ptr_dec)(stage + i * 1024), 16, 0, 0);
}
// this wave's share of the cross-attention of one query row: acc[cb][reg] = ctx[head 4 g + reg][channel 16 cb + col] (unnormalised)
template <int NW>
__device__ __forceinline__ void cross_absorbed_wave_bx3(const uint16_t* __restrict__ mh, const uint16_t* __restrict__ ml, int T,
                                                        const unsigned char* qp_b, unsigned char* stage, int wave, int lane,
                                                        float& m_run, float& l_run, f32x4 (&acc)[16]) {
  const int col = lane & 15, g = lane >> 4, hrow = col & 7;
  m_run = -INFINITY;
  l_run = 0.f;
#pragma unroll
  for (int cb = 0; cb < 16; ++cb) acc[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int ntiles = (T + 15) >> 4;
  // transposing-read addresses of the weighted sum's B operand: lane 4 q' + p of a 16-lane group supplies row 4 g + q' of the
  // block, channels 4 p .. 4 p + 3 of the column block; with the row's chunk swizzle: chunk (2 cb + (p >> 1)) ^ row
  const int trow = 4 * g + ((lane & 15) >> 2), tp = lane & 3;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the first tile (issued at kernel entry) has landed
  WAVE_PHASE_INIT();
  for (int tile = wave; tile < ntiles; tile += NW) {
    const int j0 = tile << 4;
    const int nxt = tile + NW;
    WAVE_PHASE(13);
    du32x4 pf[16];
    if (nxt < ntiles) {  // wave-uniform: the next tile travels to registers during this tile's arithmetic
#pragma unroll
      for (int i = 0; i < 16; ++i) pf[i] = *reinterpret_cast<const du32x4*>(bx3_piece_src(mh, ml, T, nxt, i, lane));
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) pf[i] = du32x4{0u, 0u, 0u, 0u};
    }
    WAVE_PHASE(14);
    // ---- S^T[key = 4 g' + reg][head = col]: A = tile rows (key = col), B = absorbed queries (head = col & 7) ----
    f32x4 sacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const int ch = 4 * ks + g;  // logical 16-byte chunk (8 channels) of this lane's k-group
      const dbf16x8 ah = *reinterpret_cast<const dbf16x8*>(stage + col * 512 + ((ch ^ col) << 4));
      const dbf16x8 al = *reinterpret_cast<const dbf16x8*>(stage + 8192 + col * 512 + ((ch ^ col) << 4));
      const dbf16x8 bh = *reinterpret_cast<const dbf16x8*>(qp_b + hrow * 512 + ((ch ^ hrow) << 4));
      const dbf16x8 bl = *reinterpret_cast<const dbf16x8*>(qp_b + 4096 + hrow * 512 + ((ch ^ hrow) << 4));
      sacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, sacc, 0, 0, 0);
      sacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, sacc, 0, 0, 0);
      sacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, sacc, 0, 0, 0);
    }
    // ---- online softmax: this lane holds keys j0 + 4 g + reg of head `col` ----
    float sv[4], mx = -INFINITY;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      sv[reg] = (j0 + 4 * g + reg < T) ? sacc[reg] : -INFINITY;
      mx = fmaxf(mx, sv[reg]);
    }
    WAVE_PHASE(15);
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));  // a tile holds at least one valid key: finite
    const float m_new = fmaxf(m_run, mx);
    const float alpha = expf(m_run - m_new);  // exp(-inf) = 0 for the first tile
    float ps = 0.f;
    unsigned ph[4], pl[4];
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const float pv = expf(sv[reg] - m_new);  // exp(-inf) = 0 for keys past the end
      ps += pv;
      split16(pv, ph[reg], pl[reg]);
    }
    l_run = l_run * alpha + ps;
    m_run = m_new;
    float ar[4];  // the accumulators hold ctx[head = 4 g + reg][...]: their scale is the alpha of THAT head (lane 4 g + reg has it)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) ar[reg] = __shfl(alpha, 4 * g + reg, 64);
#pragma unroll
    for (int cb = 0; cb < 16; ++cb)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) acc[cb][reg] *= ar[reg];
    const ds4 pah = {(short)ph[0], (short)ph[1], (short)ph[2], (short)ph[3]}, pal = {(short)pl[0], (short)pl[1], (short)pl[2], (short)pl[3]};
    WAVE_PHASE(16);
    // ---- ctx[head][16 cb + col] += sum_keys P[head][key] m[key][channel]: A = P (head = col, keys 4 g .. 4 g + 3), B by transposing reads ----
#pragma unroll
    for (int cb = 0; cb < 16; ++cb) {
      const int off = trow * 512 + (((2 * cb + (tp >> 1)) ^ trow) << 4) + (tp & 1) * 8;
      const ds4 bh = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ds4_ptr)(stage + off));
      const ds4 bl = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ds4_ptr)(stage + 8192 + off));
      acc[cb] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(pal, bh, acc[cb], 0, 0, 0);
      acc[cb] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(pah, bl, acc[cb], 0, 0, 0);
      acc[cb] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(pah, bh, acc[cb], 0, 0, 0);
    }
    WAVE_PHASE(17);
    if (nxt < ntiles) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this tile's fragment reads have returned: the stage may be overwritten
#pragma unroll
      for (int i = 0; i < 16; ++i) *reinterpret_cast<du32x4*>(stage + i * 1024 + lane * 16) = pf[i];
    }
    WAVE_PHASE(18);
  }
  l_run += __shfl_xor(l_run, 16, 64);
  l_run += __shfl_xor(l_run, 32, 64);
}

struct DecRow2P {
  DecRowP r;            // as decoder_row_kernel (ck / cv unused)
  const float* mem;     // [samples][T][D] encoder memory (engine-owned copy)
  long long mem_stride; // floats per sample
  const float* wk;      // [D][D] cross-attention key projection as stored (rows = output feature h*hd + e)
  const float* wv_t;    // [D (c)][D (o)] value projection, transposed
  const float* bv;      // [D]
  // beam search (MODE 1 / 2): the row kernel split around the per-SAMPLE cross-attention kernel
  float* qp;            // [rows][8][D]: MODE 1 writes the absorbed queries, beam_cross_kernel replaces them with the context rows
  float* x1;            // [rows][D]: LN1 output (the residual of the cross-attention block), MODE 1 -> MODE 2
  // round 4: the memory rows as two bf16 planes (hi = upper 16 bits, lo = bf16(x - hi); launch_split_bf16) for the greedy
  // two-row kernel's split-bf16 cross-attention (nullptr: the fp32 rows above on the fp32 MFMA)
  const uint16_t* mem_hi;   // [samples][T][D]
  const uint16_t* mem_lo;   // [samples][T][D]
};

#ifdef D2T_PROBES
#define ROW_PROBE(bit) (p.probe & (bit))
#else
#define ROW_PROBE(bit) 0
#endif
// MODE 0: the whole row step (greedy).  MODE 1: up to the absorbed queries, which go to q.qp (+ x1 to q.x1).  MODE 2: from
// the context rows in q.qp on (value projection, output projection, residual) -- the two halves around beam_cross_kernel.
constexpr int ANC_MAX = 512;  // longest ancestry row held in LDS (DecRowP::anc needs s_Lmax <= ANC_MAX)
template <int NTH, int MODE, bool BX3 = false>  // D = 256, 8 heads of 32; BX3 (MODE 0): the cross-attention on split-bf16 MFMAs
__global__ __launch_bounds__(NTH, 2) void decoder_row_absorbed_kernel(const DecRow2P q) {
  constexpr int D = 256, HD = 32, NW = NTH / 64, G = NTH / (D / 4), HPW = 8 / NW;
  const DecRowP& p = q.r;
  if (p.stop_at && *p.stop_at && *p.step_ptr >= *p.stop_at) return;  // block-uniform
  if (p.rows_ptr && (int)blockIdx.x >= *p.rows_ptr) return;        // device-side beam search: a dead row slot
  decode_wave_priority();
  TraceScope trace_(p.trace);
  // 77 KB: two blocks per CU.  The GEMV partial sums live in the (then idle) tile staging area, the merged context rows
  // in the absorbed queries' place (the queries are in registers by then).
  __shared__ __attribute__((aligned(1024))) unsigned char stage_s[NW * 16384];
  __shared__ __attribute__((aligned(1024))) float qp_s[8 * D];
  __shared__ __attribute__((aligned(16))) float a_s[D], y_s[D], x1_s[D], q2_s[D];
  __shared__ float wm_s[NW][8], wl_s[NW][8];
  float* const part_s = reinterpret_cast<float*>(stage_s);
  float* const ctx_s = qp_s;
  static_assert(G * D * 4 <= 16384, "GEMV partial sums must fit into one wave's staging area");
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int b = blockIdx.x;
  const int t = *p.step_ptr;
  const float* qkv = p.qkv + (size_t)b * p.qkv_stride;
  if (MODE != 2) {
  // beam search: the hypothesis' earlier positions stay in the cache rows they were written to (no cache copy per step)
  __shared__ int anc_s[ANC_MAX];
  if (p.anc) {
    for (int j = tid; j < t; j += NTH) anc_s[j] = p.anc[(size_t)b * p.anc_stride + j];
    __syncthreads();
  }
  // ---- self-attention over the cache (as decoder_row_kernel) ----
#pragma unroll
  for (int hp = 0; hp < (ROW_PROBE(1) ? 0 : HPW); ++hp) {
    const int head = wave + hp * NW;
    float* Kc = p.sk + (size_t)b * p.s_batch_stride + (size_t)head * p.s_Lmax * HD;
    float* Vc = p.sv + (size_t)b * p.s_batch_stride + (size_t)head * p.s_Lmax * HD;
    const float* curk = qkv + D + head * HD;
    const float* curv = qkv + 2 * D + head * HD;
    if (lane < HD) {
      Kc[(size_t)t * HD + lane] = curk[lane];
      Vc[(size_t)t * HD + lane] = curv[lane];
    }
    if (p.anc)
      row_attention<HD, 4>(qkv + head * HD, p.sk + (size_t)head * p.s_Lmax * HD, p.sv + (size_t)head * p.s_Lmax * HD, curk, curv,
                           t, t + 1, a_s + head * HD, lane, anc_s, p.s_batch_stride);
    else
      row_attention<HD, 4>(qkv + head * HD, Kc, Vc, curk, curv, t, t + 1, a_s + head * HD, lane);
  }
  __syncthreads();
  if (!ROW_PROBE(2)) row_gemv<D, NTH>(a_s, p.wo_t, part_s, tid);
  __syncthreads();
  if (tid < D) {
    float v = p.bo[tid] + p.xres[(size_t)b * D + tid];
#pragma unroll
    for (int gI = 0; gI < G; ++gI) v += part_s[gI * D + tid];
    y_s[tid] = v;
  }
  __syncthreads();
  if (wave == 0) {  // LN1, two-pass, one wave
    constexpr int V = D / 64;
    float v[V], s = 0.f;
#pragma unroll
    for (int i = 0; i < V; ++i) { v[i] = y_s[i * 64 + lane]; s += v[i]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float mean = s * (1.f / D);
    float qq = 0.f;
#pragma unroll
    for (int i = 0; i < V; ++i) { v[i] -= mean; qq += v[i] * v[i]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) qq += __shfl_xor(qq, o, 64);
    const float rstd = 1.f / sqrtf(qq * (1.f / D) + p.eps);
#pragma unroll
    for (int i = 0; i < V; ++i) {
      const int c = i * 64 + lane;
      x1_s[c] = v[i] * rstd * p.ln1_g[c] + p.ln1_b[c];
    }
  }
  __syncthreads();
  if (!ROW_PROBE(2)) row_gemv<D, NTH>(x1_s, p.wq_t, part_s, tid);
  __syncthreads();
  if (tid < D) {
    float v = p.bq[tid];
#pragma unroll
    for (int gI = 0; gI < G; ++gI) v += part_s[gI * D + tid];
    q2_s[tid] = v;
  }
  __syncthreads();
  // ---- absorbed queries: q'[h][c] = scale * sum_e q2[h*32 + e] * W_k[h*32 + e][c] ----
  if (!ROW_PROBE(2)) {
    const float scale = 0.17677669529663687f;  // 1 / sqrt(32)
    for (int idx = tid; idx < 8 * (D / 4); idx += NTH) {
      const int h = idx / (D / 4), c4 = (idx % (D / 4)) * 4;
      const float* w = q.wk + (size_t)(h * HD) * D + c4;
      float4 accq = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
      for (int e = 0; e < HD; ++e) {
        const float4 w4 = *reinterpret_cast<const float4*>(w + (size_t)e * D);
        const float a = q2_s[h * HD + e];
        accq.x = fmaf(a, w4.x, accq.x); accq.y = fmaf(a, w4.y, accq.y);
        accq.z = fmaf(a, w4.z, accq.z); accq.w = fmaf(a, w4.w, accq.w);
      }
      const float4 qv = make_float4(accq.x * scale, accq.y * scale, accq.z * scale, accq.w * scale);
      if (MODE == 1) *reinterpret_cast<float4*>(q.qp + ((size_t)b * 8 + h) * D + c4) = qv;
      else if (BX3) {  // hi / lo bf16 planes (cross_absorbed_wave_bx3): head h's row of 512 B, chunk (c4 >> 3) ^ h, half (c4 >> 2) & 1
        const float v[4] = {qv.x, qv.y, qv.z, qv.w};
        unsigned hi[4], lo[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) split16(v[e], hi[e], lo[e]);
        unsigned char* base = reinterpret_cast<unsigned char*>(qp_s) + h * 512 + (((c4 >> 3) ^ h) << 4) + ((c4 >> 2) & 1) * 8;
        *reinterpret_cast<uint2*>(base) = make_uint2(hi[0] | hi[1] << 16, hi[2] | hi[3] << 16);
        *reinterpret_cast<uint2*>(base + 4096) = make_uint2(lo[0] | lo[1] << 16, lo[2] | lo[3] << 16);
      } else *reinterpret_cast<float4*>(qp_s + h * D + ((((c4 >> 2) ^ h)) << 2)) = qv;
    }
  }
  __syncthreads();
  }
  if (MODE == 1) {
    if (tid < D) q.x1[(size_t)b * D + tid] = x1_s[tid];
    return;
  }
  if (MODE == 2) {  // the context rows of this hypothesis and its LN1 output come back from global memory
    for (int idx = tid; idx < 8 * (D / 4); idx += NTH)
      *reinterpret_cast<float4*>(ctx_s + idx * 4) = *reinterpret_cast<const float4*>(q.qp + (size_t)b * 8 * D + idx * 4);
    if (tid < D) x1_s[tid] = q.x1[(size_t)b * D + tid];
    __syncthreads();
  }
  // ---- cross-attention over the memory rows of this row's sample ----
  if (MODE == 0 && !ROW_PROBE(4)) {
    const int cb = p.c_row_map ? p.c_row_map[b] : b;
    float m_run, l_run;
    unsigned char* stage = stage_s + wave * 16384;
    const int col = lane & 15, g = lane >> 4;
    float* mine = reinterpret_cast<float*>(stage);
    if constexpr (BX3) {
      const uint16_t* mh = q.mem_hi + (size_t)cb * q.mem_stride;
      const uint16_t* ml = q.mem_lo + (size_t)cb * q.mem_stride;
      f32x4 acc[16];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (wave < ((p.T + 15) >> 4)) bx3_tile_dma(mh, ml, p.T, wave, stage, lane);  // (the stage held GEMV partials until now)
      cross_absorbed_wave_bx3<NW>(mh, ml, p.T, reinterpret_cast<const unsigned char*>(qp_s), stage, wave, lane, m_run, l_run, acc);
      if (g == 0 && col < 8) { wm_s[wave][col] = m_run; wl_s[wave][col] = l_run; }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (g < 2) {
#pragma unroll
        for (int cbk = 0; cbk < 16; ++cbk)
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) mine[(4 * g + reg) * D + 16 * cbk + col] = acc[cbk][reg];
      }
    } else {
      f32x4 acc[4][4];
      cross_absorbed_wave<NW>(q.mem + (size_t)cb * q.mem_stride, p.T, qp_s, stage, wave, lane, m_run, l_run, acc);
      if (g == 0 && col < 8) { wm_s[wave][col] = m_run; wl_s[wave][col] = l_run; }
      // this wave's un-normalised ctx [8 heads][256] into its (now idle) staging area: lane (g, col) holds heads 4g + reg
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (g < 2) {
#pragma unroll
        for (int w = 0; w < 4; ++w)
#pragma unroll
          for (int reg = 0; reg < 4; ++reg)
            *reinterpret_cast<float4*>(mine + (4 * g + reg) * D + 64 * w + 4 * col) =
                make_float4(acc[w][0][reg], acc[w][1][reg], acc[w][2][reg], acc[w][3][reg]);
      }
    }
  }
  __syncthreads();
  for (int idx = tid; idx < (MODE != 0 || ROW_PROBE(4) ? 0 : 8 * (D / 4)); idx += NTH) {  // merge the waves' partial softmaxes (log-sum-exp combine)
    const int h = idx / (D / 4), c4 = (idx % (D / 4)) * 4;
    float M = -INFINITY;
#pragma unroll
    for (int w = 0; w < NW; ++w) M = fmaxf(M, wm_s[w][h]);
    float L = 0.f;
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const float f = wl_s[w][h] > 0.f ? expf(wm_s[w][h] - M) : 0.f;  // waves without keys contribute nothing
      L += wl_s[w][h] * f;
      const float4 c = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(stage_s + w * 16384) + h * D + c4);
      o.x += c.x * f; o.y += c.y * f; o.z += c.z * f; o.w += c.w * f;
    }
    const float inv = 1.f / L;
    *reinterpret_cast<float4*>(ctx_s + h * D + c4) = make_float4(o.x * inv, o.y * inv, o.z * inv, o.w * inv);
  }
  __syncthreads();
  // ---- a2[o] = b_v[o] + sum_c ctx[head(o)][c] * W_v^T[c][o] ----
  if (!ROW_PROBE(2)) {
    constexpr int LPR = D / 4, KG = D / G;
    const int lr = tid % LPR, gI = tid / LPR;
    const float* w = q.wv_t + (size_t)(gI * KG) * D + lr * 4;
    const float* in = ctx_s + ((lr * 4) / HD) * D + gI * KG;
    float4 accv = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < KG; ++k) {
      const float4 w4 = *reinterpret_cast<const float4*>(w + (size_t)k * D);
      const float a = in[k];
      accv.x = fmaf(a, w4.x, accv.x); accv.y = fmaf(a, w4.y, accv.y);
      accv.z = fmaf(a, w4.z, accv.z); accv.w = fmaf(a, w4.w, accv.w);
    }
    *reinterpret_cast<float4*>(part_s + gI * D + lr * 4) = accv;
  }
  __syncthreads();
  if (tid < D) {
    float v = q.bv[tid];
#pragma unroll
    for (int gI = 0; gI < G; ++gI) v += part_s[gI * D + tid];
    a_s[tid] = v;
  }
  __syncthreads();
  if (!ROW_PROBE(2)) row_gemv<D, NTH>(a_s, p.wco_t, part_s, tid);
  __syncthreads();
  if (tid < D) {
    float v = p.bco[tid] + x1_s[tid];
#pragma unroll
    for (int gI = 0; gI < G; ++gI) v += part_s[gI * D + tid];
    p.y2[(size_t)b * D + tid] = v;
  }
}

#ifdef D2T_PROBES  // round 3's issue order of the two-row kernel, kept for A/B timing (D2T_DECODE_ROW2_NO_PREFETCH) in probe builds
// ---------------------------------------------------------------------------------------------------------------------
// The same step for TWO rows per block (greedy decode).  The five row GEMVs read 5 x 256 KB of weights per row, and with two
// one-row blocks on a CU that is 2.56 MB through the CU's 64 B/clk L2 path per launch -- 18 of the kernel's 69 us (probe
// build).  Here the block's 512 threads fetch every weight element ONCE and apply it to both rows' inputs (two
// accumulators, eight K groups of 32 instead of four of 64), the absorbed-query product shares W_k the same way, and the
// attention phases run as before: waves 0-3 on row 2b, waves 4-7 on row 2b + 1, each wave its own key tiles and 16 KB stage.
// Per row the arithmetic and its order are those of decoder_row_absorbed_kernel (the K groups of a GEMV are summed in the
// same ascending order: groups of 32 instead of 64 change the association) -- results agree to fp32 rounding.
// 152 KB of LDS: one block per CU.
// ---------------------------------------------------------------------------------------------------------------------
template <int D>
__device__ __forceinline__ void row2_gemv(const float* in0, const float* in1, const float* __restrict__ Wt, float* part_s, int tid) {
  // part_s[row][g][n]; thread (lr = n / 4, g): k in [g KG, (g + 1) KG)
  constexpr int LPR = D / 4, G = 512 / LPR, KG = D / G;
  const int lr = tid % LPR, g = tid / LPR;
  const float* w = Wt + (size_t)(g * KG) * D + lr * 4;
  float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
#pragma unroll
  for (int k = 0; k < KG; ++k) {
    const float4 w4 = *reinterpret_cast<const float4*>(w + (size_t)k * D);
    const float x0 = in0[g * KG + k], x1 = in1[g * KG + k];
    a0.x = fmaf(x0, w4.x, a0.x); a0.y = fmaf(x0, w4.y, a0.y); a0.z = fmaf(x0, w4.z, a0.z); a0.w = fmaf(x0, w4.w, a0.w);
    a1.x = fmaf(x1, w4.x, a1.x); a1.y = fmaf(x1, w4.y, a1.y); a1.z = fmaf(x1, w4.z, a1.z); a1.w = fmaf(x1, w4.w, a1.w);
  }
  *reinterpret_cast<float4*>(part_s + (0 * G + g) * D + lr * 4) = a0;
  *reinterpret_cast<float4*>(part_s + (1 * G + g) * D + lr * 4) = a1;
}

__global__ __launch_bounds__(512, 1) void decoder_row2_absorbed_kernel(const DecRow2P q) {
  constexpr int D = 256, HD = 32, G = 8;
  const DecRowP& p = q.r;
  if (p.stop_at && *p.stop_at && *p.step_ptr >= *p.stop_at) return;  // block-uniform
  decode_wave_priority();
  TraceScope trace_(p.trace);
  __shared__ __attribute__((aligned(1024))) unsigned char stage_s[8 * 16384];
  __shared__ __attribute__((aligned(1024))) float qp_s[2][8 * D];
  __shared__ __attribute__((aligned(16))) float a_s[2][D], y_s[2][D], x1_s[2][D], q2_s[2][D];
  __shared__ float wm_s[2][4][8], wl_s[2][4][8];
  float* const part_s = reinterpret_cast<float*>(stage_s);  // [2][G][D] = 16 KB of the (then idle) staging area
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int half = wave >> 2, w4i = wave & 3;   // this wave's row of the pair, its index among the row's four waves
  // an odd row count: the last block's second half repeats the last row (same kernel for EVERY row, so a row's result never
  // depends on how many rows share the launch) and keeps its stores to itself
  const bool valid = 2 * (int)blockIdx.x + half < p.M;
  const int b = valid ? 2 * blockIdx.x + half : p.M - 1;
  const int trow = tid >> 8, tcol = tid & 255;  // element-wise phases: thread -> (row of the pair, channel)
  const bool tvalid = 2 * (int)blockIdx.x + trow < p.M;
  const int brow = tvalid ? 2 * blockIdx.x + trow : p.M - 1;
  const int t = *p.step_ptr;
  // ---- self-attention over the cache, two heads per wave ----
  {
    const float* qkv = p.qkv + (size_t)b * p.qkv_stride;
#pragma unroll
    for (int hp = 0; hp < 2; ++hp) {
      const int head = w4i + hp * 4;
      float* Kc = p.sk + (size_t)b * p.s_batch_stride + (size_t)head * p.s_Lmax * HD;
      float* Vc = p.sv + (size_t)b * p.s_batch_stride + (size_t)head * p.s_Lmax * HD;
      const float* curk = qkv + D + head * HD;
      const float* curv = qkv + 2 * D + head * HD;
      if (lane < HD && valid) {
        Kc[(size_t)t * HD + lane] = curk[lane];
        Vc[(size_t)t * HD + lane] = curv[lane];
      }
      row_attention<HD, 4>(qkv + head * HD, Kc, Vc, curk, curv, t, t + 1, a_s[half] + head * HD, lane);
    }
  }
  __syncthreads();
  row2_gemv<D>(a_s[0], a_s[1], p.wo_t, part_s, tid);
  __syncthreads();
  {
    float v = p.bo[tcol] + p.xres[(size_t)brow * D + tcol];
#pragma unroll
    for (int g = 0; g < G; ++g) v += part_s[(trow * G + g) * D + tcol];
    y_s[trow][tcol] = v;
  }
  __syncthreads();
  if (w4i == 0) {  // LN1, two-pass, one wave per row
    constexpr int V = D / 64;
    float v[V], s = 0.f;
#pragma unroll
    for (int i = 0; i < V; ++i) { v[i] = y_s[half][i * 64 + lane]; s += v[i]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float mean = s * (1.f / D);
    float qq = 0.f;
#pragma unroll
    for (int i = 0; i < V; ++i) { v[i] -= mean; qq += v[i] * v[i]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) qq += __shfl_xor(qq, o, 64);
    const float rstd = 1.f / sqrtf(qq * (1.f / D) + p.eps);
#pragma unroll
    for (int i = 0; i < V; ++i) {
      const int c = i * 64 + lane;
      x1_s[half][c] = v[i] * rstd * p.ln1_g[c] + p.ln1_b[c];
    }
  }
  __syncthreads();
  row2_gemv<D>(x1_s[0], x1_s[1], p.wq_t, part_s, tid);
  __syncthreads();
  {
    float v = p.bq[tcol];
#pragma unroll
    for (int g = 0; g < G; ++g) v += part_s[(trow * G + g) * D + tcol];
    q2_s[trow][tcol] = v;
  }
  __syncthreads();
  // ---- absorbed queries of both rows: q'[h][c] = scale * sum_e q2[h*32 + e] * W_k[h*32 + e][c]; thread -> (head, 4 channels) ----
  {
    const float scale = 0.17677669529663687f;  // 1 / sqrt(32)
    const int h = tid / (D / 4), c4 = (tid % (D / 4)) * 4;
    const float* w = q.wk + (size_t)(h * HD) * D + c4;
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
#pragma unroll 8
    for (int e = 0; e < HD; ++e) {
      const float4 w4 = *reinterpret_cast<const float4*>(w + (size_t)e * D);
      const float x0 = q2_s[0][h * HD + e], x1 = q2_s[1][h * HD + e];
      a0.x = fmaf(x0, w4.x, a0.x); a0.y = fmaf(x0, w4.y, a0.y); a0.z = fmaf(x0, w4.z, a0.z); a0.w = fmaf(x0, w4.w, a0.w);
      a1.x = fmaf(x1, w4.x, a1.x); a1.y = fmaf(x1, w4.y, a1.y); a1.z = fmaf(x1, w4.z, a1.z); a1.w = fmaf(x1, w4.w, a1.w);
    }
    const int o = h * D + ((((c4 >> 2) ^ h)) << 2);
    *reinterpret_cast<float4*>(qp_s[0] + o) = make_float4(a0.x * scale, a0.y * scale, a0.z * scale, a0.w * scale);
    *reinterpret_cast<float4*>(qp_s[1] + o) = make_float4(a1.x * scale, a1.y * scale, a1.z * scale, a1.w * scale);
  }
  __syncthreads();
  // ---- cross-attention over the memory rows of each row's sample: four waves per row ----
  {
    const int cb = p.c_row_map ? p.c_row_map[b] : b;
    float m_run, l_run;
    f32x4 acc[4][4];
    unsigned char* stage = stage_s + wave * 16384;
    cross_absorbed_wave<4>(q.mem + (size_t)cb * q.mem_stride, p.T, qp_s[half], stage, w4i, lane, m_run, l_run, acc);
    const int col = lane & 15, g = lane >> 4;
    if (g == 0 && col < 8) { wm_s[half][w4i][col] = m_run; wl_s[half][w4i][col] = l_run; }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float* mine = reinterpret_cast<float*>(stage);
    if (g < 2) {
#pragma unroll
      for (int w = 0; w < 4; ++w)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg)
          *reinterpret_cast<float4*>(mine + (4 * g + reg) * D + 64 * w + 4 * col) =
              make_float4(acc[w][0][reg], acc[w][1][reg], acc[w][2][reg], acc[w][3][reg]);
    }
  }
  __syncthreads();
  for (int idx = tid; idx < 2 * 8 * (D / 4); idx += 512) {  // merge each row's four partial softmaxes (log-sum-exp combine)
    const int row = idx / (8 * (D / 4)), rem = idx % (8 * (D / 4));
    const int h = rem / (D / 4), c4 = (rem % (D / 4)) * 4;
    float M = -INFINITY;
#pragma unroll
    for (int w = 0; w < 4; ++w) M = fmaxf(M, wm_s[row][w][h]);
    float L = 0.f;
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float f = wl_s[row][w][h] > 0.f ? expf(wm_s[row][w][h] - M) : 0.f;
      L += wl_s[row][w][h] * f;
      const float4 c = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(stage_s + (row * 4 + w) * 16384) + h * D + c4);
      o.x += c.x * f; o.y += c.y * f; o.z += c.z * f; o.w += c.w * f;
    }
    const float inv = 1.f / L;
    *reinterpret_cast<float4*>(qp_s[row] + h * D + c4) = make_float4(o.x * inv, o.y * inv, o.z * inv, o.w * inv);  // ctx in the queries' place
  }
  __syncthreads();
  // ---- a2[o] = b_v[o] + sum_c ctx[head(o)][c] * W_v^T[c][o], both rows ----
  {
    constexpr int LPR = D / 4, KG = D / G;
    const int lr = tid % LPR, g = tid / LPR;
    const float* w = q.wv_t + (size_t)(g * KG) * D + lr * 4;
    const int hoff = ((lr * 4) / HD) * D + g * KG;
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
#pragma unroll
    for (int k = 0; k < KG; ++k) {
      const float4 w4 = *reinterpret_cast<const float4*>(w + (size_t)k * D);
      const float x0 = qp_s[0][hoff + k], x1 = qp_s[1][hoff + k];
      a0.x = fmaf(x0, w4.x, a0.x); a0.y = fmaf(x0, w4.y, a0.y); a0.z = fmaf(x0, w4.z, a0.z); a0.w = fmaf(x0, w4.w, a0.w);
      a1.x = fmaf(x1, w4.x, a1.x); a1.y = fmaf(x1, w4.y, a1.y); a1.z = fmaf(x1, w4.z, a1.z); a1.w = fmaf(x1, w4.w, a1.w);
    }
    // (the merge above still reads the staging area of waves 0 .. 7: it is behind the barrier)
    *reinterpret_cast<float4*>(part_s + (0 * G + g) * D + lr * 4) = a0;
    *reinterpret_cast<float4*>(part_s + (1 * G + g) * D + lr * 4) = a1;
  }
  __syncthreads();
  {
    float v = q.bv[tcol];
#pragma unroll
    for (int g = 0; g < G; ++g) v += part_s[(trow * G + g) * D + tcol];
    a_s[trow][tcol] = v;
  }
  __syncthreads();
  row2_gemv<D>(a_s[0], a_s[1], p.wco_t, part_s, tid);
  __syncthreads();
  {
    float v = p.bco[tcol] + x1_s[trow][tcol];
#pragma unroll
    for (int g = 0; g < G; ++g) v += part_s[(trow * G + g) * D + tcol];
    if (tvalid) p.y2[(size_t)brow * D + tcol] = v;
  }
}

#endif  // D2T_PROBES

// ---------------------------------------------------------------------------------------------------------------------
// The row step for TWO rows per block (greedy decode; the shipped form since round 4).
//
// Two rows per block (round 3): the five row GEMVs read 5 x 256 KB of weights per row through the CU's 64 B/clk L2 path; the
// block's 512 threads fetch every weight element ONCE and apply it to both rows' inputs (two accumulators, eight K groups of
// 32), the absorbed-query product shares W_k the same way, and the attention phases run per row: waves 0-3 on row 2b, waves
// 4-7 on row 2b + 1, each wave its own key tiles and 16 KB stage.  Per row the arithmetic and its order are those of
// decoder_row_absorbed_kernel (K groups of 32 instead of 64 change the association: equal to fp32 rounding).  An odd row
// count: the last block's second half repeats the last row and keeps its stores to itself -- the same kernel for EVERY row, so
// a row's result never depends on how many rows share the launch.  156 KB of LDS: one block per CU.
//
// Issue order (round 4).  PMC of round 3's form at 384 rows (profiles/r04_pmc_decode.txt): matrix pipes 12 % busy, SQ_WAIT_ANY
// 49 % of the wave cycles -- a serial chain of phases that each began with an exposed round trip.  Same arithmetic per element
// in the same order (bit-identical results), but
//   * a wave's FIRST memory tile is on its way (LDS-DMA) from the kernel's first instruction -- the staging area is idle until
//     the cross-attention (the GEMV partials of the first two projections live in the absorbed queries' LDS instead);
//   * inside the cross-attention the NEXT tile travels to registers (16 x 16 bytes per lane) while the matrix cores work on the
//     current one, and moves to LDS when its reads are done: a tile costs max(arithmetic, round trip), not their sum;
//   * a GEMV's weight rows (32 x 16 bytes per thread) are requested one phase early: W_o before the self-attention, W_q behind
//     the W_o products, W_k behind the W_q products, W_v behind the cross-attention loop, W_co behind the W_v products; the
//     element-wise phases' few global operands even earlier (the vector-memory counter retires in order);
//   * the self-attention runs a wave's two heads in ONE loop (twice the K / V groups in flight per round trip);
//   * block barriers are raw s_barrier behind lgkmcnt(0) (LDS only): a __syncthreads() fence would drain the prefetches.
// Phase timeline at 384 rows, alone on the chip (probe build, tools/probe/row_phases.py; us per launch), with the cross-attention
// on split-bf16 MFMAs (cross_absorbed_wave_bx3): entry .. self-attention 21.9, five GEMVs 10.6, cross-attention 18.9, element-wise
// 3.9 -- 55.3 in all (round 3's kernel: 64.4; this issue order on the fp32 MFMA: 62.7).
// ---------------------------------------------------------------------------------------------------------------------
#define ROW_SYNC() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)

// weight rows of one two-row GEMV: thread (lr = tid % 64, g = tid / 64) holds Wt[32 g + k][4 lr .. 4 lr + 3], k = 0 .. 31
__device__ __forceinline__ void gemv2_load(const float* __restrict__ Wt, int g, int lr, float4 (&w)[32]) {
  const float* src = Wt + (size_t)(g * 32) * 256 + lr * 4;
#pragma unroll
  for (int k = 0; k < 32; ++k) w[k] = *reinterpret_cast<const float4*>(src + (size_t)k * 256);
}
// part_s[row][g][n] += ... exactly row2_gemv<256>'s products in its order; in0 / in1 = the two rows' inputs at k = 32 g
__device__ __forceinline__ void gemv2_fma(const float4 (&w)[32], const float* in0, const float* in1, float4& a0, float4& a1) {
  a0 = make_float4(0.f, 0.f, 0.f, 0.f);
  a1 = a0;
#pragma unroll
  for (int k = 0; k < 32; ++k) {
    const float4 w4 = w[k];
    const float x0 = in0[k], x1 = in1[k];
    a0.x = fmaf(x0, w4.x, a0.x); a0.y = fmaf(x0, w4.y, a0.y); a0.z = fmaf(x0, w4.z, a0.z); a0.w = fmaf(x0, w4.w, a0.w);
    a1.x = fmaf(x1, w4.x, a1.x); a1.y = fmaf(x1, w4.y, a1.y); a1.z = fmaf(x1, w4.z, a1.z); a1.w = fmaf(x1, w4.w, a1.w);
  }
}

// row_attention<32, U> for TWO heads of one row in one loop (per head the same keys per lane in the same order)
struct NoHook { __device__ __forceinline__ void operator()() const {} };
// `after_first_issue` runs once, right behind the first iteration's K / V loads: the caller's own prefetches (first memory tile,
// first GEMV's weight rows) then queue BEHIND those loads instead of in front of them, and their issue time is covered by the
// first round trip (the vector-memory counter retires in order: what is issued first is awaited first)
template <int U, int HD = 32, class Hook = NoHook>
__device__ __forceinline__ void row_attention_2h(const float* const (&q)[2], const float* const (&Kc)[2], const float* const (&Vc)[2],
                                                 const float* const (&curk)[2], const float* const (&curv)[2], int t, int L,
                                                 float* const (&out)[2], int lane, Hook after_first_issue = Hook()) {
  constexpr int LPK = HD / 4, KPI = 64 / LPK;
  const int kig = lane / LPK, ch = lane % LPK;
  float4 q4[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) q4[h] = *reinterpret_cast<const float4*>(q[h] + ch * 4);
  const float scale = HD == 32 ? 0.17677669529663687f : 0.125f;
  float m[2] = {-INFINITY, -INFINITY}, l[2] = {0.f, 0.f};
  float4 acc[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
  const int nit = (L + KPI - 1) / KPI;
  float4 k4[2][U], v4[2][U];
  auto issue = [&](int it0) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int j = (it0 + u) * KPI + kig;
      const int jj = j < L ? j : 0;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const float* kr = (curk[h] && jj == t) ? curk[h] : Kc[h] + (size_t)jj * HD;
        const float* vr = (curv[h] && jj == t) ? curv[h] : Vc[h] + (size_t)jj * HD;
        k4[h][u] = *reinterpret_cast<const float4*>(kr + ch * 4);
        v4[h][u] = *reinterpret_cast<const float4*>(vr + ch * 4);
      }
    }
  };
  WAVE_PHASE_INIT();
  issue(0);
  __builtin_amdgcn_sched_barrier(0);
  WAVE_PHASE(21);  // first K / V groups issued
  after_first_issue();
  __builtin_amdgcn_sched_barrier(0);
  WAVE_PHASE(22);  // the caller's prefetches issued
  for (int it0 = 0; it0 < nit; it0 += U) {
    if (it0 > 0) issue(it0);
    WAVE_PHASE(23);  // (further groups issued)
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int j = (it0 + u) * KPI + kig;
        float d = (q4[h].x * k4[h][u].x + q4[h].y * k4[h][u].y) + (q4[h].z * k4[h][u].z + q4[h].w * k4[h][u].w);
#pragma unroll
        for (int o = 1; o < LPK; o <<= 1) d += __shfl_xor(d, o, 64);
        if (j < L) {
          const float sj = d * scale;
          const float mn = fmaxf(m[h], sj);
          const float f = expf(m[h] - mn);
          const float pj = expf(sj - mn);
          l[h] = l[h] * f + pj;
          acc[h].x = acc[h].x * f + pj * v4[h][u].x; acc[h].y = acc[h].y * f + pj * v4[h][u].y;
          acc[h].z = acc[h].z * f + pj * v4[h][u].z; acc[h].w = acc[h].w * f + pj * v4[h][u].w;
          m[h] = mn;
        }
      }
    WAVE_PHASE(24);  // groups awaited + scored
  }
#pragma unroll
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int o = LPK; o < 64; o <<= 1) {
      const float mo = __shfl_xor(m[h], o, 64), lo = __shfl_xor(l[h], o, 64);
      const float ax = __shfl_xor(acc[h].x, o, 64), ay = __shfl_xor(acc[h].y, o, 64);
      const float az = __shfl_xor(acc[h].z, o, 64), aw = __shfl_xor(acc[h].w, o, 64);
      const float mn = fmaxf(m[h], mo);
      const float f1 = l[h] > 0.f ? expf(m[h] - mn) : 0.f, f2 = lo > 0.f ? expf(mo - mn) : 0.f;
      l[h] = l[h] * f1 + lo * f2;
      acc[h].x = acc[h].x * f1 + ax * f2; acc[h].y = acc[h].y * f1 + ay * f2;
      acc[h].z = acc[h].z * f1 + az * f2; acc[h].w = acc[h].w * f1 + aw * f2;
      m[h] = mn;
    }
    if (kig == 0) {
      const float inv = 1.f / l[h];
      *reinterpret_cast<float4*>(out[h] + ch * 4) = make_float4(acc[h].x * inv, acc[h].y * inv, acc[h].z * inv, acc[h].w * inv);
    }
  }
}

// LDS-DMA of key tile `tile` of `mem` into a wave's 16 KB stage (the layout cross_absorbed_wave reads)
__device__ __forceinline__ void cross_tile_dma(const float* __restrict__ mem, int T, int tile, unsigned char* stage, int lane) {
  const int j0 = tile << 4;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int j = j0 + i < T ? j0 + i : T - 1;
    __builtin_amdgcn_global_load_lds(mem + (size_t)j * 256 + ((lane ^ i) << 2), (lds_ptr_dec)(stage + i * 1024), 16, 0, 0);
  }
}

// cross_absorbed_wave<NW> with the tile stream pipelined: the wave's first tile was started with cross_tile_dma long before;
// every following tile is loaded to registers during the arithmetic on the current one.  Same products, same order.
template <int NW>
__device__ __forceinline__ void cross_absorbed_wave_pf(const float* __restrict__ mem, int T, const float* qp_s, unsigned char* stage,
                                                       int wave, int lane, float& m_run, float& l_run, f32x4 (&acc)[4][4]) {
  const int col = lane & 15, g = lane >> 4;
  const int hrow = col & 7;
  m_run = -INFINITY;
  l_run = 0.f;
#pragma unroll
  for (int w = 0; w < 4; ++w)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[w][e] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int ntiles = (T + 15) >> 4;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the first tile (issued at kernel entry) has landed
  WAVE_PHASE_INIT();
  for (int tile = wave; tile < ntiles; tile += NW) {
    const int j0 = tile << 4;
    const int nxt = tile + NW;
    WAVE_PHASE(13);
    f32x4 pf[16];
    if (nxt < ntiles) {  // wave-uniform
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int j = (nxt << 4) + i < T ? (nxt << 4) + i : T - 1;
        pf[i] = *reinterpret_cast<const f32x4*>(mem + (size_t)j * 256 + ((lane ^ i) << 2));
      }
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) pf[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    WAVE_PHASE(14);
    f32x4 sacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const float4 a4 = *reinterpret_cast<const float4*>(stage + col * 1024 + (((4 * u + g) ^ col) << 4));
      const float4 q4 = *reinterpret_cast<const float4*>(reinterpret_cast<const unsigned char*>(qp_s) + hrow * 1024 + (((4 * u + g) ^ hrow) << 4));
      sacc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, q4.x, sacc, 0, 0, 0);
      sacc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, q4.y, sacc, 0, 0, 0);
      sacc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, q4.z, sacc, 0, 0, 0);
      sacc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, q4.w, sacc, 0, 0, 0);
    }
    float sv[4], mx = -INFINITY;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      sv[reg] = (j0 + 4 * g + reg < T) ? sacc[reg] : -INFINITY;
      mx = fmaxf(mx, sv[reg]);
    }
    WAVE_PHASE(15);
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = expf(m_run - m_new);
    float pv[4], ps = 0.f;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      pv[reg] = expf(sv[reg] - m_new);
      ps += pv[reg];
    }
    l_run = l_run * alpha + ps;
    m_run = m_new;
    float ar[4];
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) ar[reg] = __shfl(alpha, 4 * g + reg, 64);
#pragma unroll
    for (int w = 0; w < 4; ++w)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) acc[w][e][reg] *= ar[reg];
    WAVE_PHASE(16);
#pragma unroll
    for (int sk = 0; sk < 4; ++sk) {
      const int key = 4 * g + sk;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const float4 b4 = *reinterpret_cast<const float4*>(stage + key * 1024 + (((16 * w + col) ^ key) << 4));
        acc[w][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(pv[sk], b4.x, acc[w][0], 0, 0, 0);
        acc[w][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(pv[sk], b4.y, acc[w][1], 0, 0, 0);
        acc[w][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(pv[sk], b4.z, acc[w][2], 0, 0, 0);
        acc[w][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(pv[sk], b4.w, acc[w][3], 0, 0, 0);
      }
    }
    WAVE_PHASE(17);
    if (nxt < ntiles) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this tile's fragment reads have returned: the stage may be overwritten
#pragma unroll
      for (int i = 0; i < 16; ++i) *reinterpret_cast<f32x4*>(stage + i * 1024 + lane * 16) = pf[i];
    }
    WAVE_PHASE(18);
  }
  l_run += __shfl_xor(l_run, 16, 64);
  l_run += __shfl_xor(l_run, 32, 64);
}

template <bool BX3>  // BX3: the cross-attention on split-bf16 MFMAs over DecRow2P::mem_hi / mem_lo (cross_absorbed_wave_bx3)
__device__ __forceinline__ void decoder_row2_absorbed_pf_body(const DecRow2P& q) {
  constexpr int D = 256, HD = 32, G = 8;
  const DecRowP& p = q.r;
  if (p.stop_at && *p.stop_at && *p.step_ptr >= *p.stop_at) return;  // block-uniform
  decode_wave_priority();
  TraceScope trace_(p.trace);
  ROW_PHASE_INIT();
  __shared__ __attribute__((aligned(1024))) unsigned char stage_s[8 * 16384];
  __shared__ __attribute__((aligned(1024))) float qp_s[2][8 * D];
  __shared__ __attribute__((aligned(16))) float a_s[2][D], y_s[2][D], x1_s[2][D], q2_s[2][D];
  __shared__ float wm_s[2][4][8], wl_s[2][4][8];
  float* const part_late = reinterpret_cast<float*>(stage_s);   // [2][G][D]: the GEMVs behind the cross-attention (stages idle again)
  float* const part_early = &qp_s[0][0];                        // [2][G][D]: the GEMVs in front of it (the first tiles are landing in the stages)
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;  // (wave-uniform values in SGPRs)
  const int half = wave >> 2, w4i = wave & 3;
  const bool valid = 2 * (int)blockIdx.x + half < p.M;
  const int b = valid ? 2 * blockIdx.x + half : p.M - 1;
  const int trow = wave >> 2, tcol = tid & 255;
  const bool tvalid = valid;
  const int brow = b;
  const int t = *p.step_ptr;
  const int lr = lane, gg = wave;  // GEMV thread coordinates: columns 4 lr .. 4 lr + 3, K group gg
  // ---- this wave's first memory tile, and the first GEMV's weight rows, are requested before anything else ----
  const int cb = p.c_row_map ? p.c_row_map[b] : b;
  const float* const mem = q.mem + (size_t)cb * q.mem_stride;
  unsigned char* const stage = stage_s + wave * 16384;
  const uint16_t* const mh = BX3 ? q.mem_hi + (size_t)cb * q.mem_stride : nullptr;
  const uint16_t* const ml = BX3 ? q.mem_lo + (size_t)cb * q.mem_stride : nullptr;
  // the element-wise phases' few global operands, requested BEFORE the weight prefetches: the vector-memory counter retires in
  // order, so a small load issued behind a 256 KB prefetch would wait for all of it
  const float bo_v = p.bo[tcol] + p.xres[(size_t)brow * D + tcol], bq_v = p.bq[tcol], bv_v = q.bv[tcol], bco_v = p.bco[tcol];
  float ln_g[4], ln_b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { ln_g[i] = p.ln1_g[i * 64 + lane]; ln_b[i] = p.ln1_b[i * 64 + lane]; }
  float4 W[32];
  // this wave's first memory tile (LDS-DMA) and the first GEMV's weight rows: issued from inside the self-attention, right
  // behind its first K / V loads (their 48 instructions' issue time -- ~100 ns each with the CU's vector-memory path busy --
  // then passes under that first round trip instead of in front of it)
  auto prefetch = [&]() {
    if (w4i < ((p.T + 15) >> 4)) {
      if constexpr (BX3) bx3_tile_dma(mh, ml, p.T, w4i, stage, lane);
      else cross_tile_dma(mem, p.T, w4i, stage, lane);
    }
    gemv2_load(p.wo_t, gg, lr, W);
  };
  ROW_PHASE(19);  // kernel entry: scalar state, small operands
  // ---- self-attention over the cache, this wave's two heads in one loop ----
  {
    const float* qkv = p.qkv + (size_t)b * p.qkv_stride;
    const float *qh[2], *Kh[2], *Vh[2], *ck[2], *cv[2];
    float* oh[2];
#pragma unroll
    for (int hp = 0; hp < 2; ++hp) {
      const int head = w4i + hp * 4;
      float* Kc = p.sk + (size_t)b * p.s_batch_stride + (size_t)head * p.s_Lmax * HD;
      float* Vc = p.sv + (size_t)b * p.s_batch_stride + (size_t)head * p.s_Lmax * HD;
      ck[hp] = qkv + D + head * HD;
      cv[hp] = qkv + 2 * D + head * HD;
      if (lane < HD && valid) {
        Kc[(size_t)t * HD + lane] = ck[hp][lane];
        Vc[(size_t)t * HD + lane] = cv[hp][lane];
      }
      qh[hp] = qkv + head * HD; Kh[hp] = Kc; Vh[hp] = Vc; oh[hp] = a_s[half] + head * HD;
    }
    row_attention_2h<4, 32>(qh, Kh, Vh, ck, cv, t, t + 1, oh, lane, prefetch);
  }
  ROW_SYNC();
  ROW_PHASE(0);
  {
    float4 a0, a1;
    gemv2_fma(W, a_s[0] + gg * 32, a_s[1] + gg * 32, a0, a1);
    gemv2_load(p.wq_t, gg, lr, W);  // next projection's rows: on their way during the reduction and LN1
    *reinterpret_cast<float4*>(part_early + (0 * G + gg) * D + lr * 4) = a0;
    *reinterpret_cast<float4*>(part_early + (1 * G + gg) * D + lr * 4) = a1;
  }
  ROW_SYNC();
  ROW_PHASE(1);
  {
    float v = bo_v;
#pragma unroll
    for (int g = 0; g < G; ++g) v += part_early[(trow * G + g) * D + tcol];
    y_s[trow][tcol] = v;
  }
  ROW_SYNC();
  ROW_PHASE(2);
  if (w4i == 0) {  // LN1, two-pass, one wave per row
    constexpr int V = D / 64;
    float v[V], s = 0.f;
#pragma unroll
    for (int i = 0; i < V; ++i) { v[i] = y_s[half][i * 64 + lane]; s += v[i]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float mean = s * (1.f / D);
    float qq = 0.f;
#pragma unroll
    for (int i = 0; i < V; ++i) { v[i] -= mean; qq += v[i] * v[i]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) qq += __shfl_xor(qq, o, 64);
    const float rstd = 1.f / sqrtf(qq * (1.f / D) + p.eps);
#pragma unroll
    for (int i = 0; i < V; ++i) {
      const int c = i * 64 + lane;
      x1_s[half][c] = v[i] * rstd * ln_g[i] + ln_b[i];
    }
  }
  ROW_SYNC();
  ROW_PHASE(3);
  {
    float4 a0, a1;
    gemv2_fma(W, x1_s[0] + gg * 32, x1_s[1] + gg * 32, a0, a1);
    gemv2_load(q.wk, gg, lr, W);  // W_k rows of head gg (= tid / 64), columns 4 lr ..: the absorbed-query product's operand
    *reinterpret_cast<float4*>(part_early + (0 * G + gg) * D + lr * 4) = a0;
    *reinterpret_cast<float4*>(part_early + (1 * G + gg) * D + lr * 4) = a1;
  }
  ROW_SYNC();
  ROW_PHASE(4);
  {
    float v = bq_v;
#pragma unroll
    for (int g = 0; g < G; ++g) v += part_early[(trow * G + g) * D + tcol];
    q2_s[trow][tcol] = v;
  }
  ROW_SYNC();
  ROW_PHASE(5);
  // ---- absorbed queries of both rows: q'[h][c] = scale * sum_e q2[h*32 + e] * W_k[h*32 + e][c]; thread -> (head gg, 4 channels) ----
  {
    const float scale = 0.17677669529663687f;  // 1 / sqrt(32)
    float4 a0, a1;
    gemv2_fma(W, q2_s[0] + gg * HD, q2_s[1] + gg * HD, a0, a1);
    if constexpr (BX3) {  // hi / lo bf16 planes: head gg's row of 512 B, 16-byte chunk (lr >> 1) ^ gg, its half lr & 1
      const float v0[4] = {a0.x * scale, a0.y * scale, a0.z * scale, a0.w * scale}, v1[4] = {a1.x * scale, a1.y * scale, a1.z * scale, a1.w * scale};
      const int ob = gg * 512 + (((lr >> 1) ^ gg) << 4) + (lr & 1) * 8;
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        unsigned hi[4], lo[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) split16(r ? v1[e] : v0[e], hi[e], lo[e]);
        unsigned char* base = reinterpret_cast<unsigned char*>(qp_s[r]);
        *reinterpret_cast<uint2*>(base + ob) = make_uint2(hi[0] | hi[1] << 16, hi[2] | hi[3] << 16);
        *reinterpret_cast<uint2*>(base + 4096 + ob) = make_uint2(lo[0] | lo[1] << 16, lo[2] | lo[3] << 16);
      }
    } else {
      const int o = gg * D + ((lr ^ gg) << 2);
      *reinterpret_cast<float4*>(qp_s[0] + o) = make_float4(a0.x * scale, a0.y * scale, a0.z * scale, a0.w * scale);
      *reinterpret_cast<float4*>(qp_s[1] + o) = make_float4(a1.x * scale, a1.y * scale, a1.z * scale, a1.w * scale);
    }
  }
  ROW_SYNC();
  ROW_PHASE(6);
  // ---- cross-attention over the memory rows of each row's sample: four waves per row ----
  {
    float m_run, l_run;
    const int col = lane & 15, g = lane >> 4;
    float* mine = reinterpret_cast<float*>(stage);
    if constexpr (BX3) {
      f32x4 acc[16];
      cross_absorbed_wave_bx3<4>(mh, ml, p.T, reinterpret_cast<const unsigned char*>(qp_s[half]), stage, w4i, lane, m_run, l_run, acc);
      ROW_PHASE(12);
      if (g == 0 && col < 8) { wm_s[half][w4i][col] = m_run; wl_s[half][w4i][col] = l_run; }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (g < 2) {  // acc[cb][reg] = ctx[head 4 g + reg][channel 16 cb + col]
#pragma unroll
        for (int cb = 0; cb < 16; ++cb)
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) mine[(4 * g + reg) * D + 16 * cb + col] = acc[cb][reg];
      }
    } else {
      f32x4 acc[4][4];
      cross_absorbed_wave_pf<4>(mem, p.T, qp_s[half], stage, w4i, lane, m_run, l_run, acc);
      ROW_PHASE(12);
      if (g == 0 && col < 8) { wm_s[half][w4i][col] = m_run; wl_s[half][w4i][col] = l_run; }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (g < 2) {
#pragma unroll
        for (int w = 0; w < 4; ++w)
#pragma unroll
          for (int reg = 0; reg < 4; ++reg)
            *reinterpret_cast<float4*>(mine + (4 * g + reg) * D + 64 * w + 4 * col) =
                make_float4(acc[w][0][reg], acc[w][1][reg], acc[w][2][reg], acc[w][3][reg]);
      }
    }
  }
  gemv2_load(q.wv_t, gg, lr, W);  // value projection's rows: on their way during the merge
  ROW_SYNC();
  ROW_PHASE(7);
  for (int idx = tid; idx < 2 * 8 * (D / 4); idx += 512) {  // merge each row's four partial softmaxes (log-sum-exp combine)
    const int row = idx / (8 * (D / 4)), rem = idx % (8 * (D / 4));
    const int h = rem / (D / 4), c4 = (rem % (D / 4)) * 4;
    float M = -INFINITY;
#pragma unroll
    for (int w = 0; w < 4; ++w) M = fmaxf(M, wm_s[row][w][h]);
    float L = 0.f;
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float f = wl_s[row][w][h] > 0.f ? expf(wm_s[row][w][h] - M) : 0.f;
      L += wl_s[row][w][h] * f;
      const float4 c = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(stage_s + (row * 4 + w) * 16384) + h * D + c4);
      o.x += c.x * f; o.y += c.y * f; o.z += c.z * f; o.w += c.w * f;
    }
    const float inv = 1.f / L;
    *reinterpret_cast<float4*>(qp_s[row] + h * D + c4) = make_float4(o.x * inv, o.y * inv, o.z * inv, o.w * inv);  // ctx in the queries' place
  }
  ROW_SYNC();
  ROW_PHASE(8);
  // ---- a2[o] = b_v[o] + sum_c ctx[head(o)][c] * W_v^T[c][o], both rows ----
  {
    const int hoff = ((lr * 4) / HD) * D + gg * 32;
    float4 a0, a1;
    gemv2_fma(W, qp_s[0] + hoff, qp_s[1] + hoff, a0, a1);
    gemv2_load(p.wco_t, gg, lr, W);
    *reinterpret_cast<float4*>(part_late + (0 * G + gg) * D + lr * 4) = a0;
    *reinterpret_cast<float4*>(part_late + (1 * G + gg) * D + lr * 4) = a1;
  }
  ROW_SYNC();
  ROW_PHASE(9);
  {
    float v = bv_v;
#pragma unroll
    for (int g = 0; g < G; ++g) v += part_late[(trow * G + g) * D + tcol];
    a_s[trow][tcol] = v;
  }
  ROW_SYNC();
  ROW_PHASE(10);
  {
    float4 a0, a1;
    gemv2_fma(W, a_s[0] + gg * 32, a_s[1] + gg * 32, a0, a1);
    ROW_SYNC();  // (every thread has read part_late's previous contents)
    *reinterpret_cast<float4*>(part_late + (0 * G + gg) * D + lr * 4) = a0;
    *reinterpret_cast<float4*>(part_late + (1 * G + gg) * D + lr * 4) = a1;
  }
  ROW_SYNC();
  ROW_PHASE(11);
  {
    float v = bco_v + x1_s[trow][tcol];
#pragma unroll
    for (int g = 0; g < G; ++g) v += part_late[(trow * G + g) * D + tcol];
    if (tvalid) p.y2[(size_t)brow * D + tcol] = v;
  }
  ROW_PHASE(20);
}

__global__ __launch_bounds__(512, 1) void decoder_row2_absorbed_pf_kernel(const DecRow2P q) { decoder_row2_absorbed_pf_body<false>(q); }
__global__ __launch_bounds__(512, 1) void decoder_row2_absorbed_bx3_kernel(const DecRow2P q) { decoder_row2_absorbed_pf_body<true>(q); }

#ifdef D2T_PROBES
}  // namespace d2t
extern "C" int d2t_debug_row_phases(unsigned long long* out, int reset) {  // probe builds: read (and clear) d2t_row_phase
  hipDeviceSynchronize();
  if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(d2t::d2t_row_phase), 32 * 8) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[32] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(d2t::d2t_row_phase), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
namespace d2t {
#endif
hipError_t launch_decoder_row_absorbed(const DecRowP& r, const float* mem, long long mem_stride, const float* wk, const float* wv_t,
                                       const float* bv, hipStream_t s, const uint16_t* mem_hi, const uint16_t* mem_lo) {
  if (r.heads != 8 || r.D != 256 || r.T < 1) return hipErrorInvalidValue;
  DecRow2P q{r, mem, mem_stride, wk, wv_t, bv, nullptr, nullptr, mem_hi, mem_lo};
  static const int probe = D2T_PROBE_ENV("D2T_ROW_PROBE");  // probe builds only: skip phases (results are garbage by construction)
  q.r.probe = probe;
  static const bool one_row = D2T_PROBE_ENV_STR("D2T_DECODE_ONE_ROW_BLOCKS") != nullptr;  // A/B: the one-row-per-block form for every row
  if (r.anc && (!r.one_row || r.s_Lmax > ANC_MAX)) return hipErrorInvalidValue;  // the two-row kernel reads the cache directly
  if (one_row || r.one_row) {
    if (mem_hi && mem_lo) hipLaunchKernelGGL((decoder_row_absorbed_kernel<256, 0, true>), dim3(r.M), dim3(256), 0, s, q);
    else hipLaunchKernelGGL((decoder_row_absorbed_kernel<256, 0>), dim3(r.M), dim3(256), 0, s, q);
    return hipGetLastError();
  }
#ifdef D2T_PROBES
  static const bool no_pf = getenv("D2T_DECODE_ROW2_NO_PREFETCH") != nullptr;  // A/B: the round-3 issue order
  if (no_pf) { hipLaunchKernelGGL(decoder_row2_absorbed_kernel, dim3((r.M + 1) / 2), dim3(512), 0, s, q); return hipGetLastError(); }
#endif
  if (mem_hi && mem_lo) hipLaunchKernelGGL(decoder_row2_absorbed_bx3_kernel, dim3((r.M + 1) / 2), dim3(512), 0, s, q);
  else hipLaunchKernelGGL(decoder_row2_absorbed_pf_kernel, dim3((r.M + 1) / 2), dim3(512), 0, s, q);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
// Beam search: ONE block per SAMPLE runs the cross-attention of all its live hypotheses (north_star's "K/V resident in LDS
// per wavefront", in the absorbed form: the sample's memory rows are staged in LDS ONCE per layer and step and serve every
// hypothesis and head; the reference repeats the memory per hypothesis, tfm.py:163, and the per-row kernel re-read it per
// hypothesis).  M <= 6 hypotheses x 8 heads = up to 48 query rows = three 16-row MFMA tiles (83 % filled at beam 5 against
// 50 % for a single row).  The four waves work through the key tiles TOGETHER: a ring of four 16 KB tile buffers with the
// LDS-DMA three tiles ahead; per tile wave w multiplies channel quarter w of the tile with the same slice of every q' row
// (the A fragment is read once for all row tiles), the partial score tiles are summed through LDS, each wave runs the online
// softmax of all rows and accumulates channel quarter w of the weighted memory rows.  (For ONE row this cooperative form
// lost to independent waves -- two block barriers per tile; with three row tiles per barrier pair it is the better split,
// and it needs no merge.)  In: q.qp = absorbed queries [rows][8][256] (from decoder_row_absorbed_kernel MODE 1); out: the
// normalised context rows in their place.
struct BeamCrossP {
  const float* mem; long long mem_stride; int T;
  float* qp;            // [rows][8][256]
  const int* seg;       // [samples][3]: first row, live hypotheses, (unused) per sample; nullptr: one sample, rows [0, M)
  int M;                // seg == nullptr: live hypotheses
  const int* step_ptr; const int* stop_at;
};

__global__ __launch_bounds__(256, 1) void beam_cross_kernel(const BeamCrossP p) {
  if (p.stop_at && *p.stop_at && *p.step_ptr >= *p.stop_at) return;
  constexpr int NR = 3, D = 256;
  int off = 0, M = p.M;
  if (p.seg) { off = p.seg[blockIdx.x * 3]; M = p.seg[blockIdx.x * 3 + 1]; }
  if (M <= 0) return;  // block-uniform (a finished sample)
  decode_wave_priority();
  __shared__ __attribute__((aligned(1024))) unsigned char stage[4 * 16384];
  __shared__ __attribute__((aligned(1024))) float qp_s[NR * 16 * D];
  __shared__ __attribute__((aligned(16))) float part[4 * 16 * NR * 16];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int col = lane & 15, g = lane >> 4;
  const int R = 8 * M, nrt = (R + 15) >> 4;
  const float* mem = p.mem + (size_t)blockIdx.x * p.mem_stride;
  float* qrows = p.qp + (size_t)off * 8 * D;
  for (int idx = tid; idx < nrt * 16 * (D / 4); idx += 256) {  // q' rows, 16-byte chunks of row r XOR-ed with r & 15; rows >= R zero
    const int row = idx / (D / 4), c4 = idx % (D / 4);
    const float4 v = row < R ? *reinterpret_cast<const float4*>(qrows + (size_t)row * D + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    *reinterpret_cast<float4*>(qp_s + row * D + ((c4 ^ (row & 15)) << 2)) = v;
  }
  float m_run[NR], l_run[NR];
  f32x4 acc[NR][4];
#pragma unroll
  for (int rt = 0; rt < NR; ++rt) {
    m_run[rt] = -INFINITY;
    l_run[rt] = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[rt][e] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int T = p.T, ntiles = (T + 15) >> 4;
  auto issue = [&](int tile) {  // this wave's four rows of the tile: row i -> buffer + i * 1024, physical chunk c holds logical c ^ i
    unsigned char* buf = stage + (tile & 3) * 16384;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = wave * 4 + k;
      const int j = (tile << 4) + i < T ? (tile << 4) + i : T - 1;  // rows past the end: a valid row, its probability is forced to zero
      __builtin_amdgcn_global_load_lds(mem + (size_t)j * D + ((lane ^ i) << 2), (lds_ptr_dec)(buf + i * 1024), 16, 0, 0);
    }
  };
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the q' loads above are done: from here on vmcnt counts tile pieces only
  issue(0);
  if (ntiles > 1) issue(1);
  if (ntiles > 2) issue(2);
  for (int tile = 0; tile < ntiles; ++tile) {
    const int j0 = tile << 4;
    const int ahead = ntiles - 1 - tile;  // block-uniform
    if (ahead >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (ahead == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // the tile has landed for everyone (and, at tile 0, the q' rows are in LDS); nobody reads tile - 1 any more
    if (tile + 3 < ntiles) issue(tile + 3);
    const unsigned char* buf = stage + (tile & 3) * 16384;
    // ---- partial S^T[key = 4g' + reg][row = 16 rt + col] over this wave's channel quarter ----
    f32x4 sacc[NR];
#pragma unroll
    for (int rt = 0; rt < NR; ++rt) sacc[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int uu = 0; uu < 4; ++uu) {
      const int u = wave * 4 + uu;
      const float4 a4 = *reinterpret_cast<const float4*>(buf + col * 1024 + (((4 * u + g) ^ col) << 4));
#pragma unroll
      for (int rt = 0; rt < NR; ++rt) {
        if (rt < nrt) {
          const float4 q4 = *reinterpret_cast<const float4*>(reinterpret_cast<const unsigned char*>(qp_s) + (rt * 16 + col) * 1024 +
                                                              (((4 * u + g) ^ col) << 4));
          sacc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, q4.x, sacc[rt], 0, 0, 0);
          sacc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, q4.y, sacc[rt], 0, 0, 0);
          sacc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, q4.z, sacc[rt], 0, 0, 0);
          sacc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, q4.w, sacc[rt], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int rt = 0; rt < NR; ++rt)
      if (rt < nrt) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) part[(wave * 16 + 4 * g + reg) * (NR * 16) + rt * 16 + col] = sacc[rt][reg];
      }
    __syncthreads();
    // ---- online softmax of every row (all waves, identically): this lane holds keys j0 + 4g + reg of row 16 rt + col ----
    float pv[NR][4];
#pragma unroll
    for (int rt = 0; rt < NR; ++rt) {
      if (rt < nrt) {
        float sv[4], mx = -INFINITY;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int o = (4 * g + reg) * (NR * 16) + rt * 16 + col;
          const float sc = (part[o] + part[16 * NR * 16 + o]) + (part[2 * 16 * NR * 16 + o] + part[3 * 16 * NR * 16 + o]);
          sv[reg] = (j0 + 4 * g + reg < T) ? sc : -INFINITY;
          mx = fmaxf(mx, sv[reg]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run[rt], mx);
        const float alpha = expf(m_run[rt] - m_new);
        float ps = 0.f;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          pv[rt][reg] = expf(sv[reg] - m_new);
          ps += pv[rt][reg];
        }
        l_run[rt] = l_run[rt] * alpha + ps;
        m_run[rt] = m_new;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {  // the accumulators hold ctx[row 16 rt + 4g + reg]: lane 4g + reg has that row's alpha
          const float ar = __shfl(alpha, 4 * g + reg, 64);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[rt][e][reg] *= ar;
        }
      } else {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) pv[rt][reg] = 0.f;
      }
    }
    // ---- ctx[row][64 wave + 4 col + e] += sum_keys P[row][key] m[key][chan]; k-step sk <-> keys 4g + sk ----
#pragma unroll
    for (int sk = 0; sk < 4; ++sk) {
      const int key = 4 * g + sk;
      const float4 b4 = *reinterpret_cast<const float4*>(buf + key * 1024 + (((16 * wave + col) ^ key) << 4));
#pragma unroll
      for (int rt = 0; rt < NR; ++rt)
        if (rt < nrt) {
          acc[rt][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(pv[rt][sk], b4.x, acc[rt][0], 0, 0, 0);
          acc[rt][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(pv[rt][sk], b4.y, acc[rt][1], 0, 0, 0);
          acc[rt][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(pv[rt][sk], b4.z, acc[rt][2], 0, 0, 0);
          acc[rt][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(pv[rt][sk], b4.w, acc[rt][3], 0, 0, 0);
        }
    }
  }
  // ---- normalise and hand the context rows back in the queries' place ----
#pragma unroll
  for (int rt = 0; rt < NR; ++rt) {
    if (rt < nrt) {
      float l = l_run[rt];
      l += __shfl_xor(l, 16, 64);
      l += __shfl_xor(l, 32, 64);
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const float inv = 1.f / __shfl(l, 4 * g + reg, 64);  // lane 4g + reg holds the denominator of row 16 rt + 4g + reg
        const int row = rt * 16 + 4 * g + reg;
        if (row < R)
          *reinterpret_cast<float4*>(qrows + (size_t)row * D + 64 * wave + 4 * col) =
              make_float4(acc[rt][0][reg] * inv, acc[rt][1][reg] * inv, acc[rt][2][reg] * inv, acc[rt][3][reg] * inv);
      }
    }
  }
}

// one beam step of a layer's row work: pre (per hypothesis) -> cross (per sample) -> post (per hypothesis).
// seg / nsamples: the samples' row segments (batched beam), or nullptr / 1 for a single sample whose rows are [0, r.M).
hipError_t launch_decoder_row_beam(const DecRowP& r, const float* mem, long long mem_stride, const float* wk, const float* wv_t,
                                   const float* bv, float* qp, float* x1, const int* seg, int nsamples, hipStream_t s) {
  if (r.heads != 8 || r.D != 256 || r.T < 1 || !qp || !x1 || nsamples < 1) return hipErrorInvalidValue;
  DecRow2P q{r, mem, mem_stride, wk, wv_t, bv, qp, x1, nullptr, nullptr};
  hipLaunchKernelGGL((decoder_row_absorbed_kernel<256, 1>), dim3(r.M), dim3(256), 0, s, q);
  BeamCrossP c{mem, mem_stride, r.T, qp, seg, r.M, r.step_ptr, r.stop_at};
  hipLaunchKernelGGL(beam_cross_kernel, dim3(nsamples), dim3(256), 0, s, c);
  hipLaunchKernelGGL((decoder_row_absorbed_kernel<256, 2>), dim3(r.M), dim3(256), 0, s, q);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
// decoder_row_kernel for TWO rows per block (round 4; d_model 512 = config C1's decoder).  rocprofv3 of C1's serving run: the
// one-row kernel is 43 % of the kernel time at 62.8 us per launch, and a third of that is the three row GEMVs pulling 3 x 1 MB
// of weights through the CU's 64 B/clk L2 path for ONE row.  Here the block's 512 threads fetch every weight element once and
// apply it to both rows' inputs (two accumulators; the same K groups of D / 4 in the same order: per row bit-identical to the
// one-row kernel), and each wave runs ITS head of both rows in one attention loop (twice the K / V groups in flight per round
// trip, row_attention_2h).  An odd row count: the last block's second half repeats the last row and keeps its stores to itself.
// ---------------------------------------------------------------------------------------------------------------------
template <int D, int HD>
__global__ __launch_bounds__(512, 1) void decoder_row2_kernel(const DecRowP p) {
  constexpr int NTH = 512, LPR = D / 4, G = NTH / LPR, KG = D / G;
  static_assert(D / HD == 8 && NTH / 64 == 8, "one wave per head");
  if (p.stop_at && *p.stop_at && *p.step_ptr >= *p.stop_at) return;  // block-uniform
  decode_wave_priority();
  TraceScope trace_(p.trace);
  __shared__ __attribute__((aligned(16))) float a_s[2][D], y_s[2][D], x1_s[2][D], q2_s[2][D], part_s[2][G * D];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int t = *p.step_ptr;
  int b[2];
  bool valid[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    valid[r] = 2 * (int)blockIdx.x + r < p.M;
    b[r] = valid[r] ? 2 * blockIdx.x + r : p.M - 1;
  }
  // one GEMV for both rows: thread (lr, g) -> columns 4 lr .. 4 lr + 3, k in [g KG, (g + 1) KG)
  auto gemv2 = [&](const float* in0, const float* in1, const float* __restrict__ Wt) {
    const int lr = tid % LPR, g = tid / LPR;
    const float* w = Wt + (size_t)(g * KG) * D + lr * 4;
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
#pragma unroll 16
    for (int k = 0; k < KG; ++k) {
      const float4 w4 = *reinterpret_cast<const float4*>(w + (size_t)k * D);
      const float x0 = in0[g * KG + k], x1 = in1[g * KG + k];
      a0.x = fmaf(x0, w4.x, a0.x); a0.y = fmaf(x0, w4.y, a0.y); a0.z = fmaf(x0, w4.z, a0.z); a0.w = fmaf(x0, w4.w, a0.w);
      a1.x = fmaf(x1, w4.x, a1.x); a1.y = fmaf(x1, w4.y, a1.y); a1.z = fmaf(x1, w4.z, a1.z); a1.w = fmaf(x1, w4.w, a1.w);
    }
    *reinterpret_cast<float4*>(part_s[0] + g * D + lr * 4) = a0;
    *reinterpret_cast<float4*>(part_s[1] + g * D + lr * 4) = a1;
  };
  // ---- self-attention: wave = head, both rows in one loop ----
  {
    const int head = wave;
    const float *qh[2], *Kh[2], *Vh[2], *ck[2], *cv[2];
    float* oh[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const float* qkv = p.qkv + (size_t)b[r] * p.qkv_stride;
      float* Kc = p.sk + (size_t)b[r] * p.s_batch_stride + (size_t)head * p.s_Lmax * HD;
      float* Vc = p.sv + (size_t)b[r] * p.s_batch_stride + (size_t)head * p.s_Lmax * HD;
      ck[r] = qkv + D + head * HD;
      cv[r] = qkv + 2 * D + head * HD;
      if (lane < HD && valid[r]) {
        Kc[(size_t)t * HD + lane] = ck[r][lane];
        Vc[(size_t)t * HD + lane] = cv[r][lane];
      }
      qh[r] = qkv + head * HD; Kh[r] = Kc; Vh[r] = Vc; oh[r] = a_s[r] + head * HD;
    }
    row_attention_2h<4, HD>(qh, Kh, Vh, ck, cv, t, t + 1, oh, lane);
  }
  __syncthreads();
  gemv2(a_s[0], a_s[1], p.wo_t);
  __syncthreads();
  for (int i = tid; i < 2 * D; i += NTH) {
    const int r = i / D, c = i % D;
    float v = p.bo[c] + p.xres[(size_t)(r ? b[1] : b[0]) * D + c];
#pragma unroll
    for (int g = 0; g < G; ++g) v += part_s[r][g * D + c];
    y_s[r][c] = v;
  }
  __syncthreads();
  if (wave < 2) {  // LN1, two-pass, one wave per row
    const int r = wave;
    constexpr int V = D / 64;
    float v[V], sm = 0.f;
#pragma unroll
    for (int i = 0; i < V; ++i) { v[i] = y_s[r][i * 64 + lane]; sm += v[i]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sm += __shfl_xor(sm, o, 64);
    const float mean = sm * (1.f / D);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < V; ++i) { v[i] -= mean; q += v[i] * v[i]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
    const float rstd = 1.f / sqrtf(q * (1.f / D) + p.eps);
#pragma unroll
    for (int i = 0; i < V; ++i) {
      const int c = i * 64 + lane;
      x1_s[r][c] = v[i] * rstd * p.ln1_g[c] + p.ln1_b[c];
    }
  }
  __syncthreads();
  gemv2(x1_s[0], x1_s[1], p.wq_t);
  __syncthreads();
  for (int i = tid; i < 2 * D; i += NTH) {
    const int r = i / D, c = i % D;
    float v = p.bq[c];
#pragma unroll
    for (int g = 0; g < G; ++g) v += part_s[r][g * D + c];
    q2_s[r][c] = v;
  }
  __syncthreads();
  // ---- cross-attention over the projected memory K / V: wave = head, both rows in one loop ----
  {
    const int head = wave;
    const float *qh[2], *Kh[2], *Vh[2];
    const float* none[2] = {nullptr, nullptr};
    float* oh[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int cb = p.c_row_map ? p.c_row_map[b[r]] : b[r];
      Kh[r] = p.ck + (size_t)cb * p.c_batch_stride + (size_t)head * p.T * HD;
      Vh[r] = p.cv + (size_t)cb * p.c_batch_stride + (size_t)head * p.T * HD;
      qh[r] = q2_s[r] + head * HD; oh[r] = a_s[r] + head * HD;
    }
    row_attention_2h<8, HD>(qh, Kh, Vh, none, none, -1, p.T, oh, lane);
  }
  __syncthreads();
  gemv2(a_s[0], a_s[1], p.wco_t);
  __syncthreads();
  for (int i = tid; i < 2 * D; i += NTH) {
    const int r = i / D, c = i % D;
    float v = p.bco[c] + x1_s[r][c];
#pragma unroll
    for (int g = 0; g < G; ++g) v += part_s[r][g * D + c];
    if (r ? valid[1] : valid[0]) p.y2[(size_t)(r ? b[1] : b[0]) * D + c] = v;
  }
}

hipError_t launch_decoder_row(const DecRowP& p, hipStream_t s) {
  if (p.heads != 8) return hipErrorInvalidValue;
  const bool small = decode_small() != 0;
  if (p.D == 256 && small) hipLaunchKernelGGL((decoder_row_kernel<256, 32, 256>), dim3(p.M), dim3(256), 0, s, p);
  else if (p.D == 256) hipLaunchKernelGGL((decoder_row_kernel<256, 32, 512>), dim3(p.M), dim3(512), 0, s, p);
  else if (p.D == 512 && !p.anc && !p.one_row && !D2T_PROBE_ENV_STR("D2T_DECODE_ONE_ROW_BLOCKS"))
    hipLaunchKernelGGL((decoder_row2_kernel<512, 64>), dim3((p.M + 1) / 2), dim3(512), 0, s, p);
  else if (p.D == 512) hipLaunchKernelGGL((decoder_row_kernel<512, 64, 512>), dim3(p.M), dim3(512), 0, s, p);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Beam search device side (reference: tfm.py:145-186 runs the model, tools/beam.py:68-105
// does log_softmax + flat top-k on the host).
// ---------------------------------------------------------------------------
__global__ void embed_tokens_kernel(const float* __restrict__ emb, const float* __restrict__ pe,
                                    const int64_t* __restrict__ tok, const int* __restrict__ step_ptr,
                                    float* __restrict__ x, int d, float sqrt_d, const int* __restrict__ rows_ptr,
                                    const int* __restrict__ stop) {
  const int b = blockIdx.x, t = *step_ptr;
  if ((rows_ptr && b >= *rows_ptr) || (stop && *stop && t >= *stop)) return;
  const int64_t tk = tok[b];
  for (int c = threadIdx.x; c < d; c += blockDim.x)
    x[(size_t)b * d + c] = emb[(size_t)tk * d + c] * sqrt_d + pe[(size_t)t * d + c];
}
hipError_t launch_embed_tokens(const float* emb, const float* pe, const int64_t* tok, const int* step_ptr, float* x,
                               int M, int d, hipStream_t s, const int* rows_ptr, const int* stop) {
  hipLaunchKernelGGL(embed_tokens_kernel, dim3(M), dim3(256), 0, s, emb, pe, tok, step_ptr, x, d, sqrtf((float)d), rows_ptr, stop);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void beam_topk_kernel(const float* __restrict__ logits,
                                                        const float* __restrict__ scores, int M, int V, int k,
                                                        float* __restrict__ topv, int* __restrict__ topi,
                                                        const int* __restrict__ seg, int kmax,
                                                        const int* __restrict__ step, const int* __restrict__ stop) {
  if (stop && *stop && *step >= *stop) return;  // device-side beam loop: every sample has finished
  if (seg) {  // batched form: this block's segment of rows
    const int off = seg[blockIdx.x * 3];
    M = seg[blockIdx.x * 3 + 1];
    k = seg[blockIdx.x * 3 + 2];
    if (M <= 0 || k <= 0) return;  // block-uniform
    logits += (size_t)off * V;
    scores += off;
    topv += (size_t)blockIdx.x * kmax;
    topi += (size_t)blockIdx.x * kmax;
  }
  __shared__ float s_lse[16];
  __shared__ float r_v[4];
  __shared__ int r_i[4];
  __shared__ float sel_v;
  __shared__ int sel_i;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  // log-sum-exp of every row (F.log_softmax, tfm.py:169)
  for (int i = wave; i < M; i += 4) {
    const float* row = logits + (size_t)i * V;
    float mx = -INFINITY;
    for (int v = lane; v < V; v += 64) mx = fmaxf(mx, row[v]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float sum = 0.f;
    for (int v = lane; v < V; v += 64) sum += expf(row[v] - mx);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    if (lane == 0) s_lse[i] = mx + logf(sum);
  }
  __syncthreads();
  const int total = M * V;
  float last_v = INFINITY;
  int last_i = -1;
  for (int sel = 0; sel < k; ++sel) {
    // best candidate strictly after (last_v, last_i) in (value desc, index asc) order
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int idx = tid; idx < total; idx += 256) {
      const int i = idx / V;
      const float c = scores[i] + (logits[idx] - s_lse[i]);
      const bool after = c < last_v || (c == last_v && idx > last_i);
      if (after && (c > bv || (c == bv && idx < bi))) { bv = c; bi = idx; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if (lane == 0) { r_v[wave] = bv; r_i[wave] = bi; }
    __syncthreads();
    if (tid == 0) {
      for (int w = 1; w < 4; ++w)
        if (r_v[w] > bv || (r_v[w] == bv && r_i[w] < bi)) { bv = r_v[w]; bi = r_i[w]; }
      sel_v = bv; sel_i = bi;
      topv[sel] = bv; topi[sel] = bi;
    }
    __syncthreads();
    last_v = sel_v; last_i = sel_i;
    __syncthreads();
  }
}
hipError_t launch_beam_topk(const float* logits, const float* scores, int M, int V, int k, float* topv, int* topi,
                            hipStream_t s) {
  if (M < 1 || M > 16 || k < 1) return hipErrorInvalidValue;
  hipLaunchKernelGGL(beam_topk_kernel, dim3(1), dim3(256), 0, s, logits, scores, M, V, k, topv, topi, nullptr, 0, nullptr, nullptr);
  return hipGetLastError();
}
hipError_t launch_beam_topk_batch(const float* logits, const float* scores, const int* seg, int N, int V, int kmax,
                                  float* topv, int* topi, hipStream_t s, const int* step, const int* stop) {
  if (N < 1 || kmax < 1 || kmax > 16) return hipErrorInvalidValue;
  hipLaunchKernelGGL(beam_topk_kernel, dim3(N), dim3(256), 0, s, logits, scores, 0, V, 0, topv, topi, seg, kmax, step, stop);
  return hipGetLastError();
}

__global__ void cache_gather_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                    const int* __restrict__ prev, int cap, int M, int heads, int Lmax, int hd,
                                    int rows) {
  // grid: (slab, i, head); copies rows*hd contiguous floats
  const int slab = blockIdx.x, i = blockIdx.y, h = blockIdx.z;
  const size_t per_row = (size_t)heads * Lmax * hd;
  const float* s = src + ((size_t)slab * cap + prev[i]) * per_row + (size_t)h * Lmax * hd;
  float* d = dst + ((size_t)slab * cap + i) * per_row + (size_t)h * Lmax * hd;
  const int n4 = rows * hd / 4;
  for (int c = threadIdx.x; c < n4; c += blockDim.x)
    reinterpret_cast<float4*>(d)[c] = reinterpret_cast<const float4*>(s)[c];
}
hipError_t launch_cache_gather(const float* src, float* dst, const int* prev, int slabs, int cap, int M, int heads,
                               int Lmax, int hd, int rows, hipStream_t s) {
  hipLaunchKernelGGL(cache_gather_kernel, dim3(slabs, M, heads), dim3(256), 0, s, src, dst, prev, cap, M, heads, Lmax,
                     hd, rows);
  return hipGetLastError();
}

// Beam search without the cache copy: anc[row][j] = cache row that holds position j of hypothesis `row`.  Before step t
// (read from the host's step pack) the survivors inherit their parent's ancestry and add the parent's row for position
// t - 1; the kernel also publishes t in the engine's step counter.  One block per row.
__global__ void beam_ancestry_kernel(const int* __restrict__ anc_old, int* __restrict__ anc_new, const int* __restrict__ prev,
                                     int stride, const int* __restrict__ step_in, int* __restrict__ step_out,
                                     const int* __restrict__ rows_ptr, const int* __restrict__ stop) {
  const int t = *step_in, row = blockIdx.x;
  if (row == 0 && threadIdx.x == 0) *step_out = t;
  if (!anc_new || t < 1) return;
  if ((rows_ptr && row >= *rows_ptr) || (stop && *stop && t >= *stop)) return;
  const int p = prev[row];
  const int* src = anc_old + (size_t)p * stride;
  int* dst = anc_new + (size_t)row * stride;
  for (int j = threadIdx.x; j < t - 1; j += blockDim.x) dst[j] = src[j];
  if (threadIdx.x == 0) dst[t - 1] = p;
}
hipError_t launch_beam_ancestry(const int* anc_old, int* anc_new, const int* prev, int rows, int stride, const int* step_in,
                                int* step_out, hipStream_t s, const int* rows_ptr, const int* stop) {
  if (rows < 1) return hipErrorInvalidValue;
  hipLaunchKernelGGL(beam_ancestry_kernel, dim3(rows), dim3(64), 0, s, anc_old, anc_new, prev, stride, step_in, step_out, rows_ptr, stop);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
// Device-side beam bookkeeping (round 4): Beam.advance (tools/beam.py:68-105) for every sample of a batched beam search in ONE
// single-block launch per step, so that the step loop needs no host round trip and can be captured as a graph.
// Thread i owns sample i (i, i + 256, ...).  Pass 1: its `live` candidates in order -- prev = idx / V, word = idx % V; a
// candidate ending in [s] joins the sample's completed set (step, parent row, score), the others become next step's
// hypotheses; a sample with `beam` completed hypotheses is finished (Beam.done) and its rows drop out.  Pass 2: the samples'
// new row counts are prefix-summed (rows stay compact, in sample order, as the host version kept them).  Pass 3: the new rows'
// token / score / sample / parent, and the (parent, token) history record the host walks back through at the end.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void beam_dev_init_kernel(const BeamDev b, long long go_token) {
  for (int i = threadIdx.x; i < b.N; i += 256) {
    b.seg[3 * i] = i; b.seg[3 * i + 1] = 1; b.seg[3 * i + 2] = b.beam;
    b.tok[i] = go_token; b.scores[i] = 0.f; b.map[i] = i; b.prev[i] = i;
    b.comp_n[i] = 0; b.fin[i] = 0;
  }
  if (threadIdx.x == 0) { b.ctrl[0] = 0; b.ctrl[1] = b.N; b.ctrl[2] = 0; b.ctrl[3] = 0; }
}
hipError_t launch_beam_dev_init(const BeamDev& b, int64_t go_token, hipStream_t s) {
  hipLaunchKernelGGL(beam_dev_init_kernel, dim3(1), dim3(256), 0, s, b, (long long)go_token);
  return hipGetLastError();
}
__global__ __launch_bounds__(256) void beam_dev_advance_kernel(const BeamDev b) {
  constexpr int KMAX = 16, SPT = 4;  // beam <= 16; up to 4 samples per thread (N <= 1024)
  __shared__ int s_new[1024], s_off[1024];
  const int t = b.ctrl[0];
  if (b.ctrl[2] && t >= b.ctrl[2]) return;  // every sample finished at an earlier step
  int n_par[SPT][KMAX], n_word[SPT][KMAX];
  float n_val[SPT][KMAX];
#pragma unroll
  for (int q = 0; q < SPT; ++q) {
    const int i = threadIdx.x + 256 * q;
    if (i >= b.N) break;
    int newM = 0;
    if (!b.fin[i] && b.seg[3 * i + 1] > 0) {
      const int off = b.seg[3 * i], live = b.seg[3 * i + 2];
      int cn = b.comp_n[i];
      for (int r = 0; r < live; ++r) {
        const int idx = b.topi[(size_t)i * b.beam + r], prev = idx / b.V, word = idx - prev * b.V;
        const float val = b.topv[(size_t)i * b.beam + r];
        if (word == b.end_token) {
          b.comp_t[(size_t)i * b.beam + cn] = t; b.comp_par[(size_t)i * b.beam + cn] = off + prev; b.comp_score[(size_t)i * b.beam + cn] = val;
          ++cn;
        } else {
          n_par[q][newM] = off + prev; n_word[q][newM] = word; n_val[q][newM] = val;
          ++newM;
        }
      }
      b.comp_n[i] = cn;
      if (cn == b.beam) { b.fin[i] = 1; newM = 0; }
    }
    s_new[i] = newM;
  }
  __syncthreads();
  if (threadIdx.x == 0) {  // (N <= 1024 small integers: a serial scan is a microsecond)
    int run = 0;
    for (int i = 0; i < b.N; ++i) { s_off[i] = run; run += s_new[i]; }
    b.ctrl[0] = t + 1;
    b.ctrl[1] = run;
    b.ctrl[3] = t + 1;
    if (run == 0) b.ctrl[2] = t + 1;  // nothing left to extend: the remaining steps of the loop return at once
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < SPT; ++q) {
    const int i = threadIdx.x + 256 * q;
    if (i >= b.N) break;
    const int off = s_off[i], newM = s_new[i];
    for (int j = 0; j < newM; ++j) {
      const int row = off + j;
      b.tok[row] = n_word[q][j]; b.scores[row] = n_val[q][j]; b.map[row] = i; b.prev[row] = n_par[q][j];
      b.hist_par[(size_t)t * b.cap + row] = n_par[q][j];
      b.hist_tok[(size_t)t * b.cap + row] = n_word[q][j];
    }
    b.seg[3 * i] = off; b.seg[3 * i + 1] = newM; b.seg[3 * i + 2] = b.fin[i] ? 0 : b.beam - b.comp_n[i];
  }
}
hipError_t launch_beam_dev_advance(const BeamDev& b, hipStream_t s) {
  if (b.N < 1 || b.N > 1024 || b.beam < 1 || b.beam > 16) return hipErrorInvalidValue;
  hipLaunchKernelGGL(beam_dev_advance_kernel, dim3(1), dim3(256), 0, s, b);
  return hipGetLastError();
}

__global__ void transpose_kernel(const float* __restrict__ src, float* __restrict__ dst, int rows, int cols) {
  const long long total = (long long)rows * cols;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i / rows), r = (int)(i % rows);
    dst[i] = src[(size_t)r * cols + c];
  }
}
hipError_t launch_transpose(const float* src, float* dst, int rows, int cols, hipStream_t s) {
  const long long total = (long long)rows * cols;
  hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048)), dim3(256),
                     0, s, src, dst, rows, cols);
  return hipGetLastError();
}

}  // namespace d2t
