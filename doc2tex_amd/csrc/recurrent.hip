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
// D = E = H = 256, V <= 1024, Tk <= 4096 keys (two alignment rows of that length in LDS: 32 of the block's 58 KB; the
// shipped max_dimension [800, 800] gives 2525).  The backward kernel of the training step keeps six such rows (96 of its 134 KB).
// ---------------------------------------------------------------------------
constexpr int AD_MAXT = 4096, AD_MAXT_TRAIN = 4096;

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
  // beam search: every hypothesis attends over its sample's keys (sample 0 unless a row map is given)
  const int bm = p.step_mode ? (p.row_sample ? p.row_sample[b] : 0) : b;
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
    if (p.teacher && tid == 0) {
      if (step == 0 || !p.use_teacher || p.use_teacher[step]) tok_s = (int)p.teacher[(size_t)b * p.S + step];
      if (p.sv_tok) p.sv_tok[(size_t)b * p.S + step] = tok_s;  // otherwise: the argmax of the previous step
    }
    if (p.sv_hprev && tid < H) {
      p.sv_hprev[((size_t)b * p.S + step) * H + tid] = h_s[tid];
      p.sv_cprev[((size_t)b * p.S + step) * H + tid] = c_s[tid];
    }
    if (p.teacher) __syncthreads();
    // (1) query projection and target embedding
    if (tid < H) {
      float a = p.bq[tid];
#pragma unroll 8
      for (int k = 0; k < H; ++k) a = fmaf(h_s[k], p.wq_t[(size_t)k * H + tid], a);
      hq_s[tid] = a;
      emb_s[tid] = p.emb ? p.emb[(size_t)tok_s * p.E + tid] : 0.f;  // one-hot targets: see tokgate below
      if (p.sv_hq) p.sv_hq[((size_t)b * p.S + step) * H + tid] = a;
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
        if (p.sv_alpha) p.sv_alpha[((size_t)b * p.S + step) * Tk + t] = a;
        mem_s[t] = p.coverage ? mem_s[t] + a : a;  // coverage: accumulated alignment (seq2seq.py:302-304)
      }
    }
    __syncthreads();
    // (4) context = alpha^T keys : 4 key groups x 256 channels, reduced through LDS
    {
      const int c = tid & 255, gq = tid >> 8;
      float a = 0.f;
#pragma unroll 8
      for (int t = gq; t < Tk; t += 4) a = fmaf(alpha_s[t], keys[(size_t)t * p.D + c], a);
      gate_s[gq * H + c] = a;
    }
    __syncthreads();
    if (tid < H) {
      ctx_s[tid] = (gate_s[tid] + gate_s[H + tid]) + (gate_s[2 * H + tid] + gate_s[3 * H + tid]);
      if (p.sv_x) {
        p.sv_x[((size_t)b * p.S + step) * (p.D + p.E) + tid] = ctx_s[tid];
        p.sv_x[((size_t)b * p.S + step) * (p.D + p.E) + p.D + tid] = emb_s[tid];
      }
    }
    __syncthreads();
    // (5) LSTMCell gates: thread = gate row, [ctx ; emb ; h] . W^T (coalesced transposed weights)
    {
      float a = p.bx[tid];
      if (p.tokgate) a += p.tokgate[(size_t)tok_s * 4 * H + tid];  // W_ih . onehot(token) = one column of W_ih
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
      if (p.sv_gates) {
        float* gs = p.sv_gates + ((size_t)b * p.S + step) * 4 * H;
        gs[tid] = ig; gs[H + tid] = fg; gs[2 * H + tid] = gg; gs[3 * H + tid] = og;
        p.sv_hafter[((size_t)b * p.S + step) * H + tid] = h_s[tid];
        p.sv_cafter[((size_t)b * p.S + step) * H + tid] = c;
      }
    }
    __syncthreads();
    // (6) generator logits + argmax (first maximum)
    float v = -INFINITY;
    if (tid < p.V) {
      v = p.bg[tid];
#pragma unroll 8
      for (int k = 0; k < H; ++k) v = fmaf(h_s[k], p.wg_t[(size_t)k * p.V + tid], v);
      if (p.out_dropmask) v = p.out_dropmask[((size_t)b * p.S + step) * p.V + tid] ? v * p.out_dropscale : 0.f;
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

// ---------------------------------------------------------------------------
// Backward of the teacher-forced LSTM-attention loop (Attention / AttentionV2.forward_greedy with is_train,
// teacher_forcing = 1, coverage or location-aware memory): one block per batch row walks the steps in reverse.
// H = D = E = 256.  See AttnTrainBwdP for what is produced.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void attn_train_lstm_bwd_kernel(const AttnTrainBwdP p) {
  constexpr int H = 256;
  __shared__ float dh_s[H], dc_s[H], dgate_s[4 * H], dctx_s[H], dhprev_s[H], hq_s[H], dhq_s[H];
  __shared__ float alpha_s[AD_MAXT_TRAIN], mem_s[AD_MAXT_TRAIN + 16], dal_s[AD_MAXT_TRAIN], de_s[AD_MAXT_TRAIN], dcov_s[AD_MAXT_TRAIN], dmem_s[AD_MAXT_TRAIN + 16];
  __shared__ __attribute__((aligned(16))) float part_s[16][H];
  __shared__ float dl_s[1024], red_s[32];
  __shared__ __attribute__((aligned(16))) float wloc_s[11 * H];  // [tap][n]
  const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int Tk = p.T - p.key_off, half = p.taps / 2;
  const float* keys = p.mem + ((size_t)b * p.T + p.key_off) * p.D;
  const float* kp = p.kp + ((size_t)b * p.T + p.key_off) * H;
  float* dkeys = p.dmem + ((size_t)b * p.T + p.key_off) * p.D;
  float* dkp = p.dkp + ((size_t)b * p.T + p.key_off) * H;
  for (int i = tid; i < p.taps * H; i += 1024) wloc_s[i] = p.wloc[(i % H) * p.taps + i / H];
  if (tid < H) { dh_s[tid] = 0.f; dc_s[tid] = 0.f; }
  for (int i = tid; i < AD_MAXT_TRAIN; i += 1024) { dcov_s[i] = 0.f; mem_s[i] = 0.f; dmem_s[i] = 0.f; }
  if (tid < 16) { mem_s[AD_MAXT_TRAIN + tid] = 0.f; dmem_s[AD_MAXT_TRAIN + tid] = 0.f; }
  __syncthreads();
  // memory after the last step = sum of all alignments (coverage) / the last alignment (location-aware)
  for (int t = 0; t < p.S; ++t)
    for (int j = tid; j < Tk; j += 1024) {
      const float a = p.sv_alpha[((size_t)b * p.S + t) * Tk + j];
      mem_s[j] = p.coverage ? mem_s[j] + a : a;
    }
  // persistent per-lane accumulators over all steps (lane -> channels n0..n0+3 of the score layer)
  const int n0 = lane * 4;
  float acc_ws[4] = {0, 0, 0, 0}, acc_bl[4] = {0, 0, 0, 0}, acc_wl[11][4];
#pragma unroll
  for (int a = 0; a < 11; ++a)
#pragma unroll
    for (int k = 0; k < 4; ++k) acc_wl[a][k] = 0.f;
  float acc_bs = 0.f;
  __syncthreads();

  for (int t = p.S - 1; t >= 0; --t) {
    const size_t bt = (size_t)b * p.S + t;
    // A. this step's alignment; memory BEFORE the step
    for (int j = tid; j < Tk; j += 1024) {
      const float a = p.sv_alpha[bt * Tk + j];
      alpha_s[j] = a;
      if (p.coverage) mem_s[j] -= a;
      else mem_s[j] = t > 0 ? p.sv_alpha[(bt - 1) * Tk + j] : 0.f;
    }
    if (tid < p.V) dl_s[tid] = p.dlogits[bt * p.V + tid];
    if (tid < H) hq_s[tid] = p.sv_hq[bt * H + tid];
    __syncthreads();
    // B. dh += generator^T dlogits   (4 threads per hidden unit)
    // (p.dhl: that product for every (row, step) from one GEMM before this kernel -- it does not depend on the recurrence,
    // and inside the loop it cost half a megabyte of generator weights per row and step)
    if (p.dhl) {
      if (tid < H) dh_s[tid] += p.dhl[bt * H + tid];
    } else if (!(p.probe & 1)) {
      const int n = tid >> 2, q = tid & 3;
      float a = 0.f;
      for (int v = q; v < p.V; v += 4) a = fmaf(dl_s[v], p.wg_t[(size_t)n * p.V + v], a);
      a += __shfl_xor(a, 1, 64);
      a += __shfl_xor(a, 2, 64);
      if (q == 0) dh_s[n] += a;
    }
    __syncthreads();
    // C. LSTMCell backward
    if (tid < H) {
      const float* gs = p.sv_gates + bt * 4 * H;
      const float ig = gs[tid], fg = gs[H + tid], gg = gs[2 * H + tid], og = gs[3 * H + tid];
      const float cp = p.sv_cprev[bt * H + tid], tc = tanhf(p.sv_cafter[bt * H + tid]);
      const float dh = dh_s[tid];
      const float dc = dc_s[tid] + dh * og * (1.f - tc * tc);
      const float dai = dc * gg * ig * (1.f - ig), daf = dc * cp * fg * (1.f - fg);
      const float dag = dc * ig * (1.f - gg * gg), dao = dh * tc * og * (1.f - og);
      dgate_s[tid] = dai; dgate_s[H + tid] = daf; dgate_s[2 * H + tid] = dag; dgate_s[3 * H + tid] = dao;
      float* dg = p.dgates + bt * 4 * H;
      dg[tid] = dai; dg[H + tid] = daf; dg[2 * H + tid] = dag; dg[3 * H + tid] = dao;
      dc_s[tid] = dc * fg;
    }
    __syncthreads();
    // D. gradient of the LSTMCell input and of h_prev: dgates (1 x 4H) times W_ih (4H x 2H: context | embedding columns)
    // and W_hh (4H x H).  All 1024 threads take part: a thread owns four consecutive columns (16-byte loads) and one
    // part of the 4H rows, the parts are added through LDS in a fixed order.  The embedding half is not part of the
    // recurrence: when p.demb is null the caller computes it for all (row, step) with one GEMM on the saved dgates, and
    // only the context half (1 MB of the 2 MB of W_ih) is streamed here.  The phase is bound by the CU's L2 ingest.
    if (!(p.probe & 2)) {
      if (p.demb) {
        const int cg = tid & 127, rp = tid >> 7;  // 128 column groups x 8 row parts of 128 rows
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        const float* w = p.wih_raw + (size_t)(rp * 128) * 2 * H + cg * 4;
#pragma unroll 8
        for (int r = 0; r < 128; ++r) {
          const float4 w4 = *reinterpret_cast<const float4*>(w + (size_t)r * 2 * H);
          const float g = dgate_s[rp * 128 + r];
          a.x = fmaf(g, w4.x, a.x); a.y = fmaf(g, w4.y, a.y); a.z = fmaf(g, w4.z, a.z); a.w = fmaf(g, w4.w, a.w);
        }
        *reinterpret_cast<float4*>(&part_s[0][0] + rp * 2 * H + cg * 4) = a;  // part_s as [8][2H]
      } else {
        const int cg = tid & 63, rp = tid >> 6;  // context columns only: 64 column groups x 16 row parts of 64 rows
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        const float* w = p.wih_raw + (size_t)(rp * 64) * 2 * H + cg * 4;
#pragma unroll 8
        for (int r = 0; r < 64; ++r) {
          const float4 w4 = *reinterpret_cast<const float4*>(w + (size_t)r * 2 * H);
          const float g = dgate_s[rp * 64 + r];
          a.x = fmaf(g, w4.x, a.x); a.y = fmaf(g, w4.y, a.y); a.z = fmaf(g, w4.z, a.z); a.w = fmaf(g, w4.w, a.w);
        }
        *reinterpret_cast<float4*>(&part_s[rp][cg * 4]) = a;
      }
    }
    __syncthreads();
    if (p.demb) {
      if (tid < 2 * H) {
        float a = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) a += (&part_s[0][0])[q * 2 * H + tid];
        if (tid < H) dctx_s[tid] = a;
        else p.demb[bt * p.E + (tid - H)] = a;
      }
    } else if (tid < H) {
      float a = 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) a += part_s[q][tid];
      dctx_s[tid] = a;
    }
    __syncthreads();
    if (!(p.probe & 2)) {
      const int cg = tid & 63, rp = tid >> 6;  // W_hh: 64 column groups x 16 row parts of 64 rows
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
      const float* w = p.whh_raw + (size_t)(rp * 64) * H + cg * 4;
#pragma unroll 8
      for (int r = 0; r < 64; ++r) {
        const float4 w4 = *reinterpret_cast<const float4*>(w + (size_t)r * H);
        const float g = dgate_s[rp * 64 + r];
        a.x = fmaf(g, w4.x, a.x); a.y = fmaf(g, w4.y, a.y); a.z = fmaf(g, w4.z, a.z); a.w = fmaf(g, w4.w, a.w);
      }
      *reinterpret_cast<float4*>(&part_s[rp][cg * 4]) = a;
    }
    __syncthreads();
    if (tid < H) {
      float a = 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) a += part_s[q][tid];
      dhprev_s[tid] = a;
    }
    __syncthreads();
    // E. context backward: dalpha_j = dctx . key_j ; dkeys_j += alpha_j * dctx   (wave per key)
    if (!(p.probe & 4)) {
      const float4 dc4 = *reinterpret_cast<const float4*>(dctx_s + n0);
      for (int j = wave; j < Tk; j += 16) {
        const float4 k4 = *reinterpret_cast<const float4*>(keys + (size_t)j * p.D + n0);
        float e = dc4.x * k4.x + dc4.y * k4.y + dc4.z * k4.z + dc4.w * k4.w;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) e += __shfl_xor(e, o, 64);
        if (lane == 0) dal_s[j] = e;
        const float a = alpha_s[j];
        float4* dk = reinterpret_cast<float4*>(dkeys + (size_t)j * p.D + n0);
        float4 v = *dk;
        v.x = fmaf(a, dc4.x, v.x); v.y = fmaf(a, dc4.y, v.y); v.z = fmaf(a, dc4.z, v.z); v.w = fmaf(a, dc4.w, v.w);
        *dk = v;
      }
    }
    __syncthreads();
    // F. softmax backward (with the coverage gradient of later steps)
    {
      float s = 0.f;
      for (int j = tid; j < Tk; j += 1024) s = fmaf(alpha_s[j], dal_s[j] + dcov_s[j], s);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
      if (lane == 0) red_s[wave] = s;
      __syncthreads();
      float tot = 0.f;
#pragma unroll
      for (int w = 0; w < 16; ++w) tot += red_s[w];
      for (int j = tid; j < Tk; j += 1024) {
        const float de = alpha_s[j] * (dal_s[j] + dcov_s[j] - tot);
        de_s[j] = de;
        acc_bs += de;
      }
    }
    __syncthreads();
    // G. score backward: u = key_proj_j + query + loc_j ; du = de_j * w * (1 - tanh(u)^2)   (wave per key)
    if (!(p.probe & 8)) {
      const float4 hq4 = *reinterpret_cast<const float4*>(hq_s + n0);
      const float4 ws4 = *reinterpret_cast<const float4*>(p.wscore + n0);
      const float4 bl4 = *reinterpret_cast<const float4*>(p.bloc + n0);
      float dq[4] = {0, 0, 0, 0};
      for (int j = wave; j < Tk; j += 16) {
        const float4 k4 = *reinterpret_cast<const float4*>(kp + (size_t)j * H + n0);
        float lc[4] = {bl4.x, bl4.y, bl4.z, bl4.w};
        for (int a = 0; a < p.taps; ++a) {
          const int tt = j + a - half;
          const float m = (tt >= 0 && tt < Tk) ? mem_s[tt] : 0.f;
          const float4 wl = *reinterpret_cast<const float4*>(wloc_s + a * H + n0);
          lc[0] = fmaf(wl.x, m, lc[0]); lc[1] = fmaf(wl.y, m, lc[1]); lc[2] = fmaf(wl.z, m, lc[2]); lc[3] = fmaf(wl.w, m, lc[3]);
        }
        const float th[4] = {tanhf(k4.x + hq4.x + lc[0]), tanhf(k4.y + hq4.y + lc[1]), tanhf(k4.z + hq4.z + lc[2]),
                             tanhf(k4.w + hq4.w + lc[3])};
        const float w4[4] = {ws4.x, ws4.y, ws4.z, ws4.w};
        const float de = de_s[j];
        float du[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          du[k] = de * w4[k] * (1.f - th[k] * th[k]);
          dq[k] += du[k];
          acc_ws[k] = fmaf(de, th[k], acc_ws[k]);
        }
        float4* dkp4 = reinterpret_cast<float4*>(dkp + (size_t)j * H + n0);
        float4 v = *dkp4;
        v.x += du[0]; v.y += du[1]; v.z += du[2]; v.w += du[3];
        *dkp4 = v;
        for (int a = 0; a < p.taps; ++a) {
          const int tt = j + a - half;
          const bool in = tt >= 0 && tt < Tk;
          const float m = in ? mem_s[tt] : 0.f;
          const float4 wl = *reinterpret_cast<const float4*>(wloc_s + a * H + n0);
          float g = du[0] * wl.x + du[1] * wl.y + du[2] * wl.z + du[3] * wl.w;
#pragma unroll
          for (int k = 0; k < 4; ++k) acc_wl[a][k] = fmaf(du[k], m, acc_wl[a][k]);
#pragma unroll
          for (int o = 32; o > 0; o >>= 1) g += __shfl_xor(g, o, 64);
          if (lane == 0 && in) atomicAdd(&dmem_s[tt], g);
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) { part_s[wave][n0 + k] = dq[k]; acc_bl[k] += dq[k]; }
    }
    __syncthreads();
    if (tid < H) {
      float a = 0.f;
#pragma unroll
      for (int w = 0; w < 16; ++w) a += part_s[w][tid];
      dhq_s[tid] = a;
      p.dhq[bt * H + tid] = a;
    }
    __syncthreads();
    // H. query projection backward into h_prev; I. coverage recursion; J. hand the state gradients to step t-1
    if (tid < H) {
      float a = dhprev_s[tid];
#pragma unroll 8
      for (int n = 0; n < H && !(p.probe & 16); ++n) a = fmaf(dhq_s[n], p.wq_raw[(size_t)n * H + tid], a);
      dh_s[tid] = a;
    }
    for (int j = tid; j < Tk; j += 1024) {
      if (p.coverage) dcov_s[j] += dmem_s[j];
      else dcov_s[j] = dmem_s[j];  // location-aware: the memory of step t is the alignment of step t-1 only
      dmem_s[j] = 0.f;
    }
    __syncthreads();
  }
  if (tid < H) { p.dh0[(size_t)b * H + tid] = dh_s[tid]; p.dc0[(size_t)b * H + tid] = dc_s[tid]; }
  // per-row partial sums of the score / location layers: reduce the 16 waves' lane accumulators through LDS
  auto reduce_store = [&](float (&v)[4], float* dst) {
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) part_s[wave][n0 + k] = v[k];
    __syncthreads();
    if (tid < H) {
      float a = 0.f;
#pragma unroll
      for (int w = 0; w < 16; ++w) a += part_s[w][tid];
      dst[tid] = a;
    }
  };
  reduce_store(acc_ws, p.dwscore + (size_t)b * H);
  reduce_store(acc_bl, p.dbloc + (size_t)b * H);
  for (int a = 0; a < p.taps; ++a) {
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) part_s[wave][n0 + k] = acc_wl[a][k];
    __syncthreads();
    if (tid < H) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < 16; ++w) v += part_s[w][tid];
      p.dwloc[((size_t)b * H + tid) * p.taps + a] = v;
    }
  }
  __syncthreads();
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc_bs += __shfl_xor(acc_bs, o, 64);
  if (lane == 0) red_s[wave] = acc_bs;
  __syncthreads();
  if (tid == 0) {
    float tot = 0.f;
    for (int w = 0; w < 16; ++w) tot += red_s[w];
    p.dbscore[b] = tot;
  }
}
hipError_t launch_attn_train_lstm_bwd(const AttnTrainBwdP& p_in, hipStream_t s) {
  AttnTrainBwdP p = p_in;
  static const int probe = D2T_PROBE_ENV("D2T_LSTM_BWD_PROBE");
  p.probe = probe;
  if (p.H != 256 || p.D != 256 || p.E != 256 || p.V > 1024 || p.T - p.key_off > AD_MAXT_TRAIN || p.T - p.key_off < 1 || p.taps > 11)
    return hipErrorInvalidValue;
  hipLaunchKernelGGL(attn_train_lstm_bwd_kernel, dim3(p.B), dim3(1024), 0, s, p);
  return hipGetLastError();
}

// gradients of loc_conv.weight [kd][taps], loc_conv.bias [kd], loc_proj.weight [H][kd], loc_proj.bias [H] from the
// folded filter's: wloc[n][a] = sum_m Wp[n][m] Wc[m][a],  bloc[n] = bp[n] + sum_m Wp[n][m] bc[m]
__global__ void loc_unfold_bwd_kernel(const float* __restrict__ dwloc, const float* __restrict__ dbloc, int B,
                                      const float* __restrict__ cw, const float* __restrict__ cb,
                                      const float* __restrict__ pw, int H, int kd, int taps, float* d_cw, float* d_cb,
                                      float* d_pw, float* d_pb) {
  extern __shared__ float sm[];  // summed dwloc [H][taps] | dbloc [H]
  float* W = sm;
  float* Bv = sm + H * taps;
  for (int i = threadIdx.x; i < H * taps; i += blockDim.x) {
    float a = 0.f;
    for (int b = 0; b < B; ++b) a += dwloc[(size_t)b * H * taps + i];
    W[i] = a;
  }
  for (int i = threadIdx.x; i < H; i += blockDim.x) {
    float a = 0.f;
    for (int b = 0; b < B; ++b) a += dbloc[(size_t)b * H + i];
    Bv[i] = a;
    d_pb[i] = a;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < H * kd; i += blockDim.x) {  // d_pw[n][m]
    const int n = i / kd, m = i % kd;
    float a = Bv[n] * cb[m];
    for (int t = 0; t < taps; ++t) a = fmaf(W[n * taps + t], cw[m * taps + t], a);
    d_pw[i] = a;
  }
  for (int i = threadIdx.x; i < kd * taps; i += blockDim.x) {  // d_cw[m][t]
    const int m = i / taps, t = i % taps;
    float a = 0.f;
    for (int n = 0; n < H; ++n) a = fmaf(W[n * taps + t], pw[n * kd + m], a);
    d_cw[i] = a;
  }
  for (int m = threadIdx.x; m < kd; m += blockDim.x) {
    float a = 0.f;
    for (int n = 0; n < H; ++n) a = fmaf(Bv[n], pw[n * kd + m], a);
    d_cb[m] = a;
  }
}
hipError_t launch_loc_unfold_bwd(const float* dwloc, const float* dbloc, int B, const float* conv_w, const float* conv_b,
                                 const float* proj_w, int H, int kd, int taps, float* d_conv_w, float* d_conv_b,
                                 float* d_proj_w, float* d_proj_b, hipStream_t s) {
  hipLaunchKernelGGL(loc_unfold_bwd_kernel, dim3(1), dim3(1024), (size_t)(H * taps + H) * 4, s, dwloc, dbloc, B, conv_w, conv_b,
                     proj_w, H, kd, taps, d_conv_w, d_conv_b, d_proj_w, d_proj_b);
  return hipGetLastError();
}
__global__ void sum_over_rows_kernel(const float* __restrict__ part, float* __restrict__ out, int B, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float a = 0.f;
  for (int b = 0; b < B; ++b) a += part[(size_t)b * C + c];
  out[c] = a;
}
hipError_t launch_sum_over_rows(const float* part, float* out, int B, int C, hipStream_t s) {
  hipLaunchKernelGGL(sum_over_rows_kernel, dim3((C + 127) / 128), dim3(128), 0, s, part, out, B, C);
  return hipGetLastError();
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
