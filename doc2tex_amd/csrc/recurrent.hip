// Recurrent pieces of the recognizer path (gfx950, fp32): VGG pooling variants, the
// height-mean, the bidirectional LSTM sequence model (seq_modeling/bilstm.py:6-24) and the
// LSTMCell attention decoder (prediction_head/seq2seq.py:224-331, seq2seq_v2.py:176-293,
// addon_module/attention1D.py:121-161,203-242).  These paths are sequential in time and
// row-local, so each is ONE launch that loops over all time steps inside the kernel.
#include "kernels.h"

namespace d2t {

__global__ __launch_bounds__(256) void maxpool_k_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int H,
                                                        int W, int C, int OH, int OW, int KH, int KW, int SH, int SW,
                                                        int PH, int PW) {
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
    for (int kh = 0; kh < KH; ++kh)
      for (int kw = 0; kw < KW; ++kw) {
        const int ih = oh * SH - PH + kh, iw = ow * SW - PW + kw;
        if ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W) {
          const float4 v = *reinterpret_cast<const float4*>(x + ((b * H + ih) * W + iw) * C + c4 * 4);
          m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
        }
      }
    *reinterpret_cast<float4*>(y + pix * C + c4 * 4) = m;
  }
}

hipError_t launch_maxpool_k(const float* x, float* y, int B, int H, int W, int C, int KH, int KW, int SH, int SW,
                            int PH, int PW, hipStream_t s) {
  if (C % 4) return hipErrorInvalidValue;
  const int OH = (H + 2 * PH - KH) / SH + 1, OW = (W + 2 * PW - KW) / SW + 1;
  const long long total = (long long)B * OH * OW * (C / 4);
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(maxpool_k_kernel, dim3(blocks), dim3(256), 0, s, x, y, B, H, W, C, OH, OW, KH, KW, SH, SW, PH, PW);
  return hipGetLastError();
}

__global__ void mean_h_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int H, int W, int C) {
  const long long total = (long long)B * W * C;
  const float inv = 1.f / (float)H;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % C);
    const int w = (int)((idx / C) % W);
    const long long b = idx / ((long long)C * W);
    float s = 0.f;
    for (int h = 0; h < H; ++h) s += x[((b * H + h) * W + w) * C + c];
    y[idx] = s * inv;
  }
}
hipError_t launch_mean_h(const float* x, float* y, int B, int H, int W, int C, hipStream_t s) {
  const long long total = (long long)B * W * C;
  hipLaunchKernelGGL(mean_h_kernel, dim3((unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096)),
                     dim3(256), 0, s, x, y, B, H, W, C);
  return hipGetLastError();
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// ---------------------------------------------------------------------------
// Bidirectional LSTM recurrence.  grid = (2 directions, ceil(B / RB)); 1024 threads = the 4H
// gate rows (H = 256).  W_hh^T [H][4H] is streamed from L2 each step with coalesced rows.
// ---------------------------------------------------------------------------
constexpr int LSTM_RB = 4;  // batch rows per block

__global__ __launch_bounds__(1024) void bilstm_kernel(const float* __restrict__ g, const float* __restrict__ whh_t,
                                                      float* __restrict__ out, int B, int T, int H) {
  __shared__ float h_s[LSTM_RB][256], c_s[LSTM_RB][256], gate_s[LSTM_RB][1024];
  const int dir = blockIdx.x, b0 = blockIdx.y * LSTM_RB, r = threadIdx.x;
  const int G4 = 4 * H;
  const float* W = whh_t + (size_t)dir * H * G4;
  for (int i = r; i < LSTM_RB * H; i += 1024) { (&h_s[0][0])[i] = 0.f; (&c_s[0][0])[i] = 0.f; }
  __syncthreads();
  for (int step = 0; step < T; ++step) {
    const int t = dir == 0 ? step : T - 1 - step;
    float acc[LSTM_RB];
#pragma unroll
    for (int b = 0; b < LSTM_RB; ++b)
      acc[b] = (b0 + b < B) ? g[((size_t)(b0 + b) * T + t) * (2 * G4) + dir * G4 + r] : 0.f;
#pragma unroll 8
    for (int k = 0; k < H; ++k) {
      const float w = W[(size_t)k * G4 + r];
#pragma unroll
      for (int b = 0; b < LSTM_RB; ++b) acc[b] = fmaf(h_s[b][k], w, acc[b]);
    }
#pragma unroll
    for (int b = 0; b < LSTM_RB; ++b) gate_s[b][r] = acc[b];
    __syncthreads();
    {
      const int b = r >> 8, j = r & 255;  // 4 rows x 256 hidden units = 1024 threads
      if (b0 + b < B) {
        const float ig = sigmoidf_(gate_s[b][j]), fg = sigmoidf_(gate_s[b][H + j]);
        const float gg = tanhf(gate_s[b][2 * H + j]), og = sigmoidf_(gate_s[b][3 * H + j]);
        const float c = fg * c_s[b][j] + ig * gg;
        const float h = og * tanhf(c);
        c_s[b][j] = c;
        h_s[b][j] = h;
        out[((size_t)(b0 + b) * T + t) * (2 * H) + dir * H + j] = h;
      }
    }
    __syncthreads();
  }
}

hipError_t launch_bilstm(const float* g, const float* whh_t, float* out, int B, int T, int H, hipStream_t s) {
  if (H != 256) return hipErrorInvalidValue;
  hipLaunchKernelGGL(bilstm_kernel, dim3(2, (B + LSTM_RB - 1) / LSTM_RB), dim3(1024), 0, s, g, whh_t, out, B, T, H);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// LSTM-attention greedy decoder: one block (1024 threads) per batch row runs every step.
// D = E = H = 256, V <= 1024, Tk <= 512.
// ---------------------------------------------------------------------------
constexpr int AD_MAXT = 512;

__global__ __launch_bounds__(1024) void attn_decode_kernel(const AttnDecP p) {
  constexpr int H = 256;
  __shared__ float x_s[3 * H];  // [context | embedding | h]  = LSTMCell input
  __shared__ float c_s[H], hq_s[H];
  __shared__ float mem_s[AD_MAXT + 16], alpha_s[AD_MAXT], red_s[32];
  __shared__ float gate_s[4 * H];
  __shared__ float logit_s[1024];
  __shared__ int tok_s;
  __shared__ __attribute__((aligned(16))) float wloc_s[11 * H];  // folded location filter, [tap][n]
  const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int Tk = p.T - p.key_off;
  const int bm = p.step_mode ? 0 : b;  // beam search: every hypothesis attends over sample 0
  const float* keys = p.mem + ((size_t)bm * p.T + p.key_off) * p.D;
  const float* kp = p.kp + ((size_t)bm * p.T + p.key_off) * H;
  float* ctx_s = x_s;
  float* emb_s = x_s + H;
  float* h_s = x_s + 2 * H;
  const int half = p.taps / 2;

  // ---- initial state (seq2seq.py:229-238) ----
  if (tid < H) {
    float init = 0.f;
    if (p.init_mode == 1) {
      for (int t = 0; t < p.T; ++t) init += p.mem[((size_t)bm * p.T + t) * p.D + tid];
      init /= (float)p.T;
    } else if (p.init_mode == 2) {
      init = p.mem[(size_t)bm * p.T * p.D + tid];
    }
    hq_s[tid] = init;  // scratch
  }
  for (int i = tid; i < AD_MAXT + 16; i += 1024) mem_s[i] = 0.f;
  for (int i = tid; i < p.taps * H; i += 1024) wloc_s[i] = p.wloc[(i % H) * p.taps + i / H];
  __syncthreads();
  if (tid < H) {
    float hh = 0.f, cc = 0.f;
    if (p.init_mode != 0) {
      hh = p.bih[tid]; cc = p.bic[tid];
      for (int k = 0; k < p.D; ++k) {
        const float v = hq_s[k];
        hh = fmaf(v, p.wih_t[(size_t)k * H + tid], hh);
        cc = fmaf(v, p.wic_t[(size_t)k * H + tid], cc);
      }
    }
    h_s[tid] = hh;
    c_s[tid] = cc;
  }
  if (tid == 0) tok_s = 0;  // [GO]
  int ended = 0;
  __syncthreads();
  if (p.step_mode && !p.first) {  // resume a hypothesis from its stored state
    if (tid < H) { h_s[tid] = p.st_h_in[(size_t)b * H + tid]; c_s[tid] = p.st_c_in[(size_t)b * H + tid]; }
    for (int t = tid; t < Tk; t += 1024) mem_s[t] = p.st_mem_in[(size_t)b * Tk + t];
    if (tid == 0) tok_s = (int)p.tok_in[b];
    __syncthreads();
  }

  for (int step = 0; step < p.S; ++step) {
    // (1) query projection and target embedding
    if (tid < H) {
      float a = p.bq[tid];
#pragma unroll 8
      for (int k = 0; k < H; ++k) a = fmaf(h_s[k], p.wq_t[(size_t)k * H + tid], a);
      hq_s[tid] = a;
      emb_s[tid] = p.emb[(size_t)tok_s * p.E + tid];
    }
    __syncthreads();
    // (2) scores: e[t] = w . tanh(key_proj[t] + query_proj + loc(mem)[t]) + b ; one wave per key
    {
      const int n0 = lane * 4;
      const float4 hq4 = *reinterpret_cast<const float4*>(hq_s + n0);
      const float4 ws4 = *reinterpret_cast<const float4*>(p.wscore + n0);
      const float4 bl4 = *reinterpret_cast<const float4*>(p.bloc + n0);
      for (int t = wave; t < Tk; t += 16) {
        const float4 k4 = *reinterpret_cast<const float4*>(kp + (size_t)t * H + n0);
        float4 lc = bl4;
        for (int j = 0; j < p.taps; ++j) {
          const int tt = t + j - half;
          const float m = (tt >= 0 && tt < Tk) ? mem_s[tt] : 0.f;
          const float4 wl = *reinterpret_cast<const float4*>(wloc_s + j * H + n0);
          lc.x = fmaf(wl.x, m, lc.x); lc.y = fmaf(wl.y, m, lc.y);
          lc.z = fmaf(wl.z, m, lc.z); lc.w = fmaf(wl.w, m, lc.w);
        }
        float e = ws4.x * tanhf(k4.x + hq4.x + lc.x) + ws4.y * tanhf(k4.y + hq4.y + lc.y) +
                  ws4.z * tanhf(k4.z + hq4.z + lc.z) + ws4.w * tanhf(k4.w + hq4.w + lc.w);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) e += __shfl_xor(e, o, 64);
        if (lane == 0) alpha_s[t] = e + p.bscore;
      }
    }
    __syncthreads();
    // (3) softmax over the keys
    {
      float m = -INFINITY;
      for (int t = tid; t < Tk; t += 1024) m = fmaxf(m, alpha_s[t]);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
      if (lane == 0) red_s[wave] = m;
      __syncthreads();
      m = red_s[0];
#pragma unroll
      for (int w = 1; w < 16; ++w) m = fmaxf(m, red_s[w]);
      __syncthreads();
      float sum = 0.f;
      for (int t = tid; t < Tk; t += 1024) {
        const float ex = expf(alpha_s[t] - m);
        alpha_s[t] = ex;
        sum += ex;
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
      if (lane == 0) red_s[16 + wave] = sum;
      __syncthreads();
      float tot = 0.f;
#pragma unroll
      for (int w = 0; w < 16; ++w) tot += red_s[16 + w];
      const float inv = 1.f / tot;
      for (int t = tid; t < Tk; t += 1024) {
        const float a = alpha_s[t] * inv;
        alpha_s[t] = a;
        mem_s[t] = p.coverage ? mem_s[t] + a : a;  // coverage: accumulated alignment (seq2seq.py:302-304)
      }
    }
    __syncthreads();
    // (4) context = alpha^T keys : 4 key groups x 256 channels, reduced through LDS
    {
      const int c = tid & 255, gq = tid >> 8;
      float a = 0.f;
      for (int t = gq; t < Tk; t += 4) a = fmaf(alpha_s[t], keys[(size_t)t * p.D + c], a);
      gate_s[gq * H + c] = a;
    }
    __syncthreads();
    if (tid < H) ctx_s[tid] = (gate_s[tid] + gate_s[H + tid]) + (gate_s[2 * H + tid] + gate_s[3 * H + tid]);
    __syncthreads();
    // (5) LSTMCell gates: thread = gate row, [ctx ; emb ; h] . W^T (coalesced transposed weights)
    {
      float a = p.bx[tid];
      const float* w = p.wx_t + tid;
#pragma unroll 8
      for (int k = 0; k < 3 * H; ++k) a = fmaf(x_s[k], w[(size_t)k * 4 * H], a);
      gate_s[tid] = a;
    }
    __syncthreads();
    if (tid < H) {
      const float ig = sigmoidf_(gate_s[tid]), fg = sigmoidf_(gate_s[H + tid]);
      const float gg = tanhf(gate_s[2 * H + tid]), og = sigmoidf_(gate_s[3 * H + tid]);
      const float c = fg * c_s[tid] + ig * gg;
      c_s[tid] = c;
      h_s[tid] = og * tanhf(c);
    }
    __syncthreads();
    // (6) generator logits + argmax (first maximum)
    float v = -INFINITY;
    if (tid < p.V) {
      v = p.bg[tid];
#pragma unroll 8
      for (int k = 0; k < H; ++k) v = fmaf(h_s[k], p.wg_t[(size_t)k * p.V + tid], v);
      p.probs[((size_t)b * p.S + step) * p.V + tid] = v;
    }
    logit_s[tid] = v;
    __syncthreads();
    if (wave == 0) {
      float best = -INFINITY;
      int bi = 0x7fffffff;
      for (int i = lane; i < p.V; i += 64) {
        const float x = logit_s[i];
        if (x > best || (x == best && i < bi)) { best = x; bi = i; }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
      }
      if (bi >= p.V) bi = 0;
      if (lane == 0) {
        tok_s = bi;
        p.tokens[(size_t)b * p.S + step] = bi;
        if (bi == p.end_token && !ended) { ended = 1; p.end_step[b] = step; }
      }
    }
    __syncthreads();
  }
  if (p.step_mode) {
    if (tid < H) { p.st_h_out[(size_t)b * H + tid] = h_s[tid]; p.st_c_out[(size_t)b * H + tid] = c_s[tid]; }
    for (int t = tid; t < Tk; t += 1024) p.st_mem_out[(size_t)b * Tk + t] = mem_s[t];
  }
}

__global__ void gather_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, const int* __restrict__ idx,
                                   int width) {
  const int i = blockIdx.x;
  for (int c = threadIdx.x; c < width; c += blockDim.x) dst[(size_t)i * width + c] = src[(size_t)idx[i] * width + c];
}
hipError_t launch_gather_rows(const float* src, float* dst, const int* idx, int rows, int width, hipStream_t s) {
  if (rows <= 0) return hipSuccess;
  hipLaunchKernelGGL(gather_rows_kernel, dim3(rows), dim3(256), 0, s, src, dst, idx, width);
  return hipGetLastError();
}

hipError_t launch_attn_decode(const AttnDecP& p, hipStream_t s) {
  if (p.H != 256 || p.D != 256 || p.E != 256 || p.V > 1024 || p.T - p.key_off > AD_MAXT || p.taps > 11 ||
      p.T - p.key_off < 1)
    return hipErrorInvalidValue;
  hipLaunchKernelGGL(attn_decode_kernel, dim3(p.B), dim3(1024), 0, s, p);
  return hipGetLastError();
}

__global__ void transpose_into_kernel(const float* __restrict__ src, int rows, int cols, float* __restrict__ dst,
                                      int ld, int row_off) {
  const long long total = (long long)rows * cols;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i / rows), r = (int)(i % rows);
    dst[(size_t)(row_off + c) * ld + r] = src[(size_t)r * cols + c];
  }
}
hipError_t launch_transpose_into(const float* src, int rows, int cols, float* dst, int ld, int row_off,
                                 hipStream_t s) {
  const long long total = (long long)rows * cols;
  hipLaunchKernelGGL(transpose_into_kernel, dim3((unsigned)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048)),
                     dim3(256), 0, s, src, rows, cols, dst, ld, row_off);
  return hipGetLastError();
}

}  // namespace d2t
