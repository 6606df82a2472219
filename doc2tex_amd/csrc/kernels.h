// Internal launch interface between the kernel translation units and the engine.
// gfx950 only; all tensors fp32, activations NHWC / row-major.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Timing-probe / ablation switches (D2T_CONV_ABL, D2T_WGRAD_ABL, D2T_LSTM_BWD_PROBE, D2T_ATTN_BWD_PROBE) skip kernel phases
// or launch kernels whose results are garbage by construction.  They exist only in a probe build (build.sh with
// D2T_PROBES=1 -> -DD2T_PROBES); the shipped library never reads them, so a stray variable in a serving or training
// environment cannot change any result.
#include <cstdlib>
// The A/B switches of the kernels' rejected variants (round 4: every former getenv of the library) follow the same rule:
// D2T_PROBE_ENV_STR is getenv in a probe build and nullptr in the shipped library.
#ifdef D2T_PROBES
#define D2T_PROBE_ENV(name) (getenv(name) ? atoi(getenv(name)) : 0)
#define D2T_PROBE_ENV_STR(name) getenv(name)
#else
#define D2T_PROBE_ENV(name) 0
#define D2T_PROBE_ENV_STR(name) (static_cast<const char*>(nullptr))
#endif

namespace d2t {

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_GELU = 2 };
enum { STORE_ROWS = 0, STORE_KV = 1 };

// Implicit-GEMM convolution / linear layer:
//   out[row(m)][n] = act( sum_{kh,kw,c} in[b, oh*SH-PH+kh, ow*SW-PW+kw, c] * w[n][((c/32)*KH*KW + kh*KW+kw)*32 + c%32]
//                         + bias[n] + res[row(m)][n] ) + row_add[add_row(m)][n]
// with m = (b*OH + oh)*OW + ow, out-of-range taps reading zero.
struct ConvP {
  const float* in;       // [B,H,W,Cin]
  const float* w;        // [Cout][K], BN folded, K swept as (32-channel chunk, tap, channel-in-chunk)
  const uint16_t* w_hi;  // optional bf16 split of w (w ~ hi + lo), same layout; selects the bf16x3 kernel
  const uint16_t* w_lo;
  // split-bf16 activations (x ~ hi + lo): per row and 32-channel group a 128-byte record
  // [32 x hi | 32 x lo] (conv_common.h plane_idx).  When in_hi != nullptr the bf16x3 kernel reads that
  // instead of `in`; when out_hi != nullptr the epilogue writes it instead of `out`; a residual may come
  // in the same form.  zero16 points at >= 16 zero bytes (out-of-image taps).
  const uint16_t* in_hi;
  uint16_t* out_hi;
  const uint16_t* res_hi;
  const void* zero16;
  int max_blocks;        // > 0: launch at most this many (persistent) blocks of the split-bf16 kernel
  int reserved_cus;      // pipelined split-bf16 kernel (one block per CU): CUs left free for other streams' kernels
  int wave_prio;         // pipelined kernel: s_setprio of its waves (experiments; 0 = default)
  int pipelined;         // 3: take the pipelined 256x128 kernel (conv_bf16x3p.hip) when the layer qualifies (Cout >= 128); 0: the 128-row kernels
  int split_tail;        // pipelined kernel: hand the rows of a sparsely filled last round of tiles to its 64 x 128 build
  // != 0: a 2x2 / stride 2 max-pool (resnet.py:94,106) fused into the epilogue of the split-record LDS-DMA kernels
  // (conv_bf16x3g_body, conv_bf16x3p16_body): the GEMM's rows are the convolution's output pixels in POOLED ORDER --
  // row m = 4 * pooled pixel + (oh & 1) * 2 + (ow & 1), M = 4 * B * (OH / 2) * (OW / 2) -- and the wide epilogue writes one
  // record row per four tile rows (their maximum) at the pooled pixel.  Needs the wide epilogue (no remap, no residual).
  int pool2;
  // second input of the pipelined split-record kernels (conv_bf16x3p.hip DmaIssuer): a 1x1 / stride 1 convolution over
  // `in2_hi` (split records [M][Cin2], the same pixels as the output) appended along K -- K = KH*KW*Cin + Cin2, the
  // weights' K rows concatenated.  A BasicBlock's shortcut (resnet.py:181-192) computed inside its conv2's launch:
  // conv2(t) + downsample(x) in ONE accumulator, no shortcut kernel, no residual round trip.
  const uint16_t* in2_hi;
  int Cin2;
  // != 0: fp16x2 arithmetic (round 3).  Records (in_hi, in2_hi, res_hi, out_hi) hold the activation once, as fp16, in their hi
  // half; w_hi / w_lo are fp16 hi / lo planes (launch_split_f16); a product is x16 * w_lo + x16 * w_hi -- two MFMAs instead of
  // three (conv_common.h split_rec / join_rec).  Split-record kernels only: conv_bf16x3g_body and conv_bf16x3p16_body.
  int f16;
  // record formats of out_hi / res_hi when they differ from the input's (round 4, "mixed" precision: a few layers on the
  // two-MFMA arithmetic inside a split-bf16 backbone).  0 = as `f16` says (bf16 split, or one fp16); otherwise 1 + format:
  // 1 = bf16 hi | lo, 2 = fp16 in the hi half, 3 = fp16 hi | fp16 lo (x ~ hi + lo to 22 bits: what a residual path reads back
  // in full precision while the next convolution's MFMAs take the hi half alone).  conv_common.h out_fmt / res_fmt.
  int out_fmt, res_fmt;
  const float* bias;     // [Cout] or nullptr
  const float* res;      // [rows][Cout] or nullptr, indexed like out
  const float* row_add;  // [*][Cout] or nullptr (positional tables), added after the activation
  float* out;
  int B, H, W, Cin, OH, OW, Cout;
  int KH, KW, SH, SW, PH, PW;
  int M;    // B*OH*OW
  int m_base;  // LDS-DMA split-bf16 kernel only: this launch covers output rows [m_base, M) (tiles count from m_base)
  int K;    // KH*KW*Cin
  int act;
  // output row remap: row(m) = (m / rows_per_img) * img_stride + row_off + m % rows_per_img
  // (rows_per_img == 0 -> identity).  add_row(m) = row_add_off + m % rows_per_img.
  int rows_per_img, img_stride, row_off, row_add_off;
  // STORE_KV: n -> (slab = n / (heads*hd), head, e), m -> (b = m / kv_T, j):
  //   out[(((slab*kv_B + b)*kv_heads + head)*kv_T + j)*kv_hd + e]
  int store_mode, kv_T, kv_heads, kv_hd, kv_B;
};
hipError_t launch_conv(const ConvP& p, hipStream_t s);         // fp32 MFMA, or bf16x3 when p.w_hi != nullptr
hipError_t launch_conv_bf16x3(const ConvP& p, hipStream_t s);
hipError_t launch_conv_bf16x3p(const ConvP& p, hipStream_t s);  // 256x128 tile, 3 LDS stages, one block per CU
hipError_t launch_conv_bf16x3g_rows(const ConvP& p, int bn, hipStream_t s);  // 128-row LDS-DMA kernel over rows [m_base, M)
// hi = bf16(w) (round-to-nearest-even), lo = bf16(w - hi)
hipError_t launch_split_bf16(const float* w, uint16_t* hi, uint16_t* lo, size_t n, hipStream_t s);
// hi = fp16(w), lo = fp16(w - hi) (fp16x2 mode, ConvP::f16)
hipError_t launch_split_f16(const float* w, uint16_t* hi, uint16_t* lo, size_t n, hipStream_t s);

// Skinny GEMM for decode steps: y[M,N] = act(x[M,K] @ w[N,K]^T + bias + res), K % 16 == 0.
// If step_ptr != nullptr the output base is advanced by (*step_ptr) * out_step_stride floats.
struct SkinnyP {
  const float* x; const float* w; const float* bias; const float* res; float* y;
  int M, K, N;
  int ldx, ldy, ldres;  // row strides (floats)
  int wld;              // row stride of w (floats); 0 = K
  unsigned long long* trace;  // debug (D2T_DECODE_TRACE): [2] = min block start / max block end, s_memrealtime ticks
  // early exit of a captured whole-loop graph: the kernel returns at once when *stop_at != 0 and *cur_step >= *stop_at
  const int* stop_at; const int* cur_step;
  int act;
  const int* step_ptr;
  long long out_step_stride;
  // optional LayerNorm prologue: the A operand is LN(x)*ln_g + ln_b over the K features of each row
  // (K = 256 or 512 only).  ln_out, if set, receives the normalised rows [M][K] as a side output.
  const float* ln_g; const float* ln_b; float ln_eps; float* ln_out;
};
hipError_t launch_skinny(const SkinnyP& p, hipStream_t s);

// conv0_1: Cin = 1, 3x3, stride 1, pad 1, Cout % 4 == 0, fused bias + ReLU.  w [Cout][9].
hipError_t launch_stem(const float* img, const float* w, const float* bias, float* out, int B, int H, int W, int Cout,
                       int act, hipStream_t s);
// same, writing split-bf16 planes
hipError_t launch_stem_split(const float* img, const float* w, const float* bias, uint16_t* out, int B, int H, int W,
                             int Cout, int act, hipStream_t s, int f16 = 0);
// fp32 [rows][C] <-> split-activation records (C % 32 == 0); used by the test entry point
hipError_t launch_split_act(const float* x, uint16_t* planes, size_t rows, int C, hipStream_t s, int f16 = 0);
hipError_t launch_merge_act(const uint16_t* planes, float* x, size_t rows, int C, hipStream_t s, int f16 = 0);
hipError_t launch_maxpool(const float* x, float* y, int B, int H, int W, int C, int SH, int SW, int PH, int PW,
                          hipStream_t s);  // 2x2 window
// 2x2 max-pool on split-bf16 planes (reconstruct, max, re-split)
hipError_t launch_maxpool_split(const uint16_t* x, uint16_t* y, int B, int H, int W, int C, int SH, int SW, int PH,
                                int PW, hipStream_t s, int f16 = 0);
hipError_t launch_maxpool_k(const float* x, float* y, int B, int H, int W, int C, int KH, int KW, int SH, int SW,
                            int PH, int PW, hipStream_t s);
// y[b][w][c] = mean_h x[b][h][w][c]   (AdaptiveAvgPool2d((None,1)) on the permuted map, build_feat.py:50-55)
hipError_t launch_mean_h(const float* x, float* y, int B, int H, int W, int C, hipStream_t s);

// One bidirectional nn.LSTM layer given the input projection of both directions:
//   g [B*T][2*4H] = x @ [W_ih ; W_ih_reverse]^T + (b_ih + b_hh) ; whh_t [2][H][4H] (transposed W_hh)
//   out [B][T][2H] = [h_fwd | h_bwd].  H = 256, gate order i,f,g,o.
hipError_t launch_bilstm(const float* g, const float* whh_t, float* out, int B, int T, int H, hipStream_t s);

// LSTM-attention greedy decoder, all steps in one launch (one block per row).
struct AttnDecP {
  const float* mem; int T; int D;      // encoder output [B][T][D] (D = 256)
  int key_off;                         // 0, or 1 = drop the cls token from the keys (AttentionV2 + TFM)
  int init_mode;                       // 0 zeros, 1 mean over all T tokens, 2 token 0
  const float* kp;                     // key_proj(mem) [B][T][H] (bias included)
  const float* wq_t; const float* bq;  // query_proj^T [H][H], bias
  const float* wloc; const float* bloc; int taps;  // folded loc_proj o loc_conv: [H][taps], [H]
  const float* wscore; float bscore;   // score [H]
  const float* wx_t; const float* bx;  // LSTMCell: [ctx(D) ; emb(E) ; h(H)] -> 4H, transposed [(D+E+H)][4H]; bias b_ih+b_hh
  const float* wg_t; const float* bg;  // generator^T [H][V], bias
  const float* wih_t; const float* bih; const float* wic_t; const float* bic;  // proj_init_h/c^T [D][H]
  const float* emb;                    // [V][E], or nullptr with one-hot targets: then ...
  const float* tokgate;                // ... [V][4H] rows of W_ih^T behind the context columns are added to the gates
  float* probs; int64_t* tokens; int* end_step;  // [B][S][V], [B][S], [B] (-1 = never ended)
  int B, S, V, H, E, coverage, end_token;
  // Beam-search step mode (S == 1): every block is one hypothesis of sample 0 (shared keys); the recurrent state is
  // read from st_*_in (unless `first`: initial state as in the greedy loop) and left in st_*_out.
  int step_mode, first;
  const float *st_h_in, *st_c_in, *st_mem_in;  // [B][H], [B][H], [B][T - key_off]
  float *st_h_out, *st_c_out, *st_mem_out;
  const int64_t* tok_in;                        // [B] input token of this step
  const int* row_sample;                        // step mode, optional [B]: the sample whose keys row b attends over (default 0)
  // Training forward (teacher forcing, seq2seq.py:311-316): the input token of step t is teacher[b*S + t]; the
  // recurrent state of every step is saved for the backward pass (all optional).
  const int64_t* teacher;
  const uint8_t* use_teacher;   // optional [S]: 0 = feed the model's own argmax at that step (scheduled sampling)
  const uint8_t* out_dropmask;  // optional [B][S][V] keep mask of the dropout on the generator output (seq2seq.py:298)
  float out_dropscale;
  int64_t* sv_tok;              // [B][S] the input token each step actually used
  float *sv_hprev, *sv_cprev, *sv_hafter, *sv_cafter;  // [B][S][H]: LSTM state before / after step t
  float* sv_gates;         // [B][S][4H]: i, f, g, o after their nonlinearities
  float* sv_alpha;         // [B][S][T - key_off]
  float* sv_hq;            // [B][S][H]   query projection
  float* sv_x;             // [B][S][D + E]: LSTMCell input [context | embedding]
};
// Backward of the teacher-forced loop above (one block per batch row, steps in reverse).  Gradients that are sums
// over (row, step) of outer products are left as per-(row, step) factors for GEMMs: dgates, dhq, demb, dh0 / dc0.
struct AttnTrainBwdP {
  const float* dlogits;    // [B][S][V]
  const float* mem; int T, D, key_off;        // keys = mem[b][key_off:]
  const float* kp;         // [B][T][H] key projection (forward value)
  const float *wg_t, *wih_raw, *whh_raw, *wq_raw, *wloc, *bloc, *wscore;  // wg_t [H][V]; raw [4H][D+E], [4H][H], [H][H]
  int taps;
  const float *sv_cprev, *sv_cafter, *sv_gates, *sv_alpha, *sv_hq;
  float* dmem;             // [B][T][D]  += context path (zero-initialised by the caller)
  float* dkp;              // [B][T][H]  += score path   (zero-initialised by the caller)
  float *dgates, *dhq, *demb;   // [B][S][4H], [B][S][H], [B][S][E] (demb null: the caller forms it from dgates afterwards)
  const float* dhl = nullptr;   // optional [B][S][H]: dlogits . generator.weight for every (row, step), formed beforehand
  float *dh0, *dc0;        // [B][H]
  float *dwloc, *dbloc, *dwscore, *dbscore;   // per-row partials [B][H][taps], [B][H], [B][H], [B]
  int B, S, V, H, E, coverage;
  int probe = 0;           // timing probe (D2T_LSTM_BWD_PROBE): bit mask of phases to skip -- 1 B, 2 D, 4 E, 8 G, 16 H
};
hipError_t launch_attn_train_lstm_bwd(const AttnTrainBwdP& p, hipStream_t s);
// fold / unfold of loc_proj o loc_conv: gradients of the four location-layer tensors from the per-row partials
hipError_t launch_loc_unfold_bwd(const float* dwloc, const float* dbloc, int B, const float* conv_w, const float* conv_b,
                                 const float* proj_w, int H, int kd, int taps, float* d_conv_w, float* d_conv_b,
                                 float* d_proj_w, float* d_proj_b, hipStream_t s);
// out[c] = sum_b part[b][c]
hipError_t launch_sum_over_rows(const float* part, float* out, int B, int C, hipStream_t s);
// dst[i][0..width) = src[idx[i]][0..width)
hipError_t launch_gather_rows(const float* src, float* dst, const int* idx, int rows, int width, hipStream_t s);
hipError_t launch_attn_decode(const AttnDecP& p, hipStream_t s);
// dst[(row_off + c) * ld + r] = src[r * cols + c]
hipError_t launch_transpose_into(const float* src, int rows, int cols, float* dst, int ld, int row_off, hipStream_t s);
hipError_t launch_layernorm(const float* x, const float* g, const float* b, float* y, int rows, int D, float eps,
                            hipStream_t s);
hipError_t launch_vit_attention(const float* qkv, float* y, int B, int N, int heads, hipStream_t s);

// Single-query attention, one wave per (b, head).
struct DecAttnP {
  const float* q; int q_stride;      // q[b*q_stride + head*hd + e]
  float* k; float* v;                // [B][heads][Lmax][hd]; row b starts at b*kv_batch_stride (0 = shared)
  long long kv_batch_stride;
  const float* cur_k; const float* cur_v; int cur_stride;  // this step's k,v rows (self-attn) or nullptr
  float* y; int y_stride;
  int B, heads, hd, Lmax;
  int L;                 // number of keys when step_ptr == nullptr
  const int* step_ptr;   // self-attn: t = *step_ptr, L = t + 1, cur_k/v are written to the cache at t
};
hipError_t launch_decode_attention(const DecAttnP& p, hipStream_t s);

// Fused per-row middle of a post-norm decoder layer (one block per row, one wave per head):
//   a  = SelfAttn(q,k,v of this step + cache)          (appends k,v to the cache)
//   y1 = a @ Wo^T + bo + xres ;  x1 = LN1(y1)
//   q2 = x1 @ Wq^T + bq ;  a2 = CrossAttn(q2, memory K/V)
//   y2 = a2 @ Wco^T + bco + x1                          (written out, pre-LN2)
// Weights are passed TRANSPOSED ([k][n], k-major) so GEMV reads are coalesced.
struct DecRowP {
  const float* qkv; int qkv_stride;   // [M][3D]
  const float* xres;                  // [M][D]
  float* sk; float* sv; long long s_batch_stride; int s_Lmax;
  const float* ck; const float* cv; long long c_batch_stride; int T;
  const int* c_row_map;               // optional: row b reads the cross K/V of sample c_row_map[b] (batched beam search)
  const float* wo_t; const float* bo;
  const float* ln1_g; const float* ln1_b; float eps;
  const float* wq_t; const float* bq;
  const float* wco_t; const float* bco;
  float* y2;                          // [M][D]
  const int* step_ptr;
  int M, D, heads;
  unsigned long long* trace;          // debug (D2T_DECODE_TRACE), as SkinnyP::trace
  const int* stop_at;                 // early exit, as SkinnyP::stop_at (the step counter is step_ptr)
  int probe;                          // probe builds only (D2T_ROW_PROBE bit mask: 1 no self-attention, 2 no GEMVs, 4 no cross-attention)
  // beam search (absorbed form only): anc[b * anc_stride + j] = cache row holding position j < step of hypothesis b
  // (launch_beam_ancestry); nullptr = every hypothesis owns its cache row.  one_row: one row per block (the faster form at
  // 5 x 128 hypothesis rows; also what keeps batched and per-sample beam search bit-identical)
  const int* anc; int anc_stride; int one_row;
  const int* rows_ptr;                // optional (device-side beam search): blocks of rows >= *rows_ptr return at once
};
hipError_t launch_decoder_row(const DecRowP& p, hipStream_t s);
// the same step with the cross-attention taken over the encoder memory itself (absorbed K / V projections, decode.hip):
// mem [samples][T][256] (row b attends over sample c_row_map[b] or b), wk [256][256] as stored, wv_t [c][o], bv [256]
// mem_hi / mem_lo (optional, greedy two-row kernel only): the same rows as bf16 hi / lo planes (launch_split_bf16) -> the
// cross-attention runs on split-bf16 MFMAs
hipError_t launch_decoder_row_absorbed(const DecRowP& p, const float* mem, long long mem_stride, const float* wk,
                                       const float* wv_t, const float* bv, hipStream_t s, const uint16_t* mem_hi = nullptr,
                                       const uint16_t* mem_lo = nullptr);
// beam search (at most 6 live hypotheses per sample): the row step split around ONE cross-attention block per sample that
// stages the sample's memory tiles once for all its hypotheses; qp [rows][8][256] and x1 [rows][256] are scratch;
// seg [nsamples][3] = (first row, live hypotheses, -) per sample, or nullptr / 1 for a single sample (rows [0, p.M))
hipError_t launch_decoder_row_beam(const DecRowP& p, const float* mem, long long mem_stride, const float* wk, const float* wv_t,
                                   const float* bv, float* qp, float* x1, const int* seg, int nsamples, hipStream_t s);
hipError_t launch_transpose(const float* src, float* dst, int rows, int cols, hipStream_t s);  // dst[c][r] = src[r][c]

// x[b] = emb[tok]*sqrt(d) + pe[t];  tok = (t == 0) ? start[b] : tokens[b*tok_stride + t - 1]
hipError_t launch_embed(const float* emb, const float* pe, const int64_t* start, const int64_t* tokens,
                        int tok_stride, const int* step_ptr, float* x, int B, int d, hipStream_t s);
// greedy argmax over logits[b][t][:V] (first maximum), writes tokens[b][t], updates end bookkeeping.
struct ArgmaxP {
  const float* logits; long long row_stride; long long step_stride;
  int64_t* tokens; int tok_stride;
  int* ended;       // [B]
  int* end_count;   // [1]
  int* steps_done;  // [1]: first t+1 at which all rows have ended (0 = not yet)
  int* step_ptr;    // read at entry, incremented by the last block to finish
  int* done_count;  // [1] zero-initialised arrival counter (reset by the kernel)
  int B, V, end_token;
  // next-step embedding written by the same kernel: x[b] = emb[token]*sqrt(d) + pe[t+1]
  const float* emb; const float* pe; float* x; int d;
  unsigned long long* trace;  // debug (D2T_DECODE_TRACE), as SkinnyP::trace
  // per-batch bookkeeping of a decode group (rows of several encoder batches in one loop): batch k = rows [k*rows_per_batch,
  // (k+1)*rows_per_batch); batch_end_count[k] / batch_steps_done[k] as end_count / steps_done; when every batch is done and
  // stop_at != nullptr, *stop_at = t + 1 makes the remaining kernels of a captured loop return immediately
  int rows_per_batch, n_batches;
  int* batch_end_count; int* batch_steps_done; int* batches_done; int* stop_at;
};
hipError_t launch_argmax_embed(const ArgmaxP& p, hipStream_t s);

// ---- beam search helpers (tools/beam.py semantics, one sample) ----
// x[i] = emb[tok[i]]*sqrt(d) + pe[*step]
hipError_t launch_embed_tokens(const float* emb, const float* pe, const int64_t* tok, const int* step_ptr, float* x,
                               int M, int d, hipStream_t s, const int* rows_ptr = nullptr, const int* stop = nullptr);
// cand[i*V+v] = score[i] + log_softmax(logits[i])[v]; the k best (value desc, flat index asc on ties)
// are written to topv[k], topi[k].  One block; M*V <= 16 * 4096.
hipError_t launch_beam_topk(const float* logits, const float* scores, int M, int V, int k, float* topv, int* topi,
                            hipStream_t s);
// the same for N independent segments of rows: seg[i] = {first row, rows, k}; results at topv/topi[i * kmax ...]
hipError_t launch_beam_topk_batch(const float* logits, const float* scores, const int* seg, int N, int V, int kmax,
                                  float* topv, int* topi, hipStream_t s, const int* step = nullptr, const int* stop = nullptr);
// Device-side beam bookkeeping (round 4; tools/beam.py:68-105 without a host round trip).  State block `BeamDev`: every array lives
// in engine memory; the step loop's kernels are launched for `cap` rows and read the live row count / the stop step from it.
struct BeamDev {
  int* ctrl;            // [0] step, [1] live rows, [2] stop-at step (0 = none), [3] steps run
  int64_t* tok;         // [cap] last token of every live hypothesis row
  float* scores;        // [cap]
  int* map;             // [cap] sample of a row
  int* prev;            // [cap] parent row (previous step's row order)
  int* seg;             // [N][3] first row, live rows, candidates wanted (beam - completed)
  int* comp_n;          // [N] completed hypotheses
  int* fin;             // [N] sample finished (beam completed hypotheses)
  int* comp_t;          // [N][beam] step at which a hypothesis completed
  int* comp_par;        // [N][beam] its parent row (that step's row order)
  float* comp_score;    // [N][beam]
  int* hist_par;        // [S][cap] parent row of the hypothesis a step created at row r (the next step's row order)
  int* hist_tok;        // [S][cap] its token
  const float* topv; const int* topi;   // [N][beam] candidates of the step (launch_beam_topk_batch)
  int N, beam, cap, V, S, end_token;
};
hipError_t launch_beam_dev_init(const BeamDev& b, int64_t go_token, hipStream_t s);
hipError_t launch_beam_dev_advance(const BeamDev& b, hipStream_t s);
// anc_new[row][0 .. t-2] = anc_old[prev[row]][...], anc_new[row][t-1] = prev[row] (t = *step_in, published to *step_out):
// which cache row holds each earlier position of hypothesis `row`.  rows_ptr / stop (optional): the device-side beam loop's
// guards -- rows >= *rows_ptr are skipped, and nothing happens once *stop != 0 and t >= *stop.
hipError_t launch_beam_ancestry(const int* anc_old, int* anc_new, const int* prev, int rows, int stride, const int* step_in,
                                int* step_out, hipStream_t s, const int* rows_ptr = nullptr, const int* stop = nullptr);
// dst[slab][i][...] = src[slab][prev[i]][...] for the first `rows` positions of every head
// (self-attention KV cache reorder after a beam step); caches are [slabs][cap][heads][Lmax][hd].
hipError_t launch_cache_gather(const float* src, float* dst, const int* prev, int slabs, int cap, int M, int heads,
                               int Lmax, int hd, int rows, hipStream_t s);

// weight packing
// OIHW (+ optional eval-BN) -> OHWI with the BN scale folded; bias_out = bn_b - mean*scale (+ conv bias*scale)
hipError_t launch_pack_conv(const float* w_oihw, const float* conv_bias, const float* bn_w, const float* bn_b,
                            const float* bn_mean, const float* bn_var, float eps, float* w_out, float* bias_out,
                            int Cout, int Cin, int KH, int KW, hipStream_t s);
// plain OHWI [Cout][KH][KW][Cin] -> the kernels' K order [Cout][Cin/32][KH*KW][32] (Cin % 32 == 0)
hipError_t launch_repack_ohwi(const float* w_ohwi, float* w_out, int Cout, int KH, int KW, int Cin, hipStream_t s);
hipError_t launch_copy(const float* src, float* dst, size_t n, hipStream_t s);
hipError_t launch_add_rows(const float* a, const float* b, float* out, int n, hipStream_t s);  // out = a + b
// out[b*img_stride*D + i] = row[i] for i < D  (cls-token row of every image)
hipError_t launch_fill_cls(const float* row, float* out, int B, long long img_stride_floats, int D, hipStream_t s);

// GlobalContext block (gcb): x [B][HW][C] += fc2(relu(LN(fc1(attention-pooled x))))
struct GCParams { const float *wg, *bg, *w1, *b1, *ln_g, *ln_b, *w2, *b2; };
hipError_t launch_global_context(float* x, const GCParams& w, float* logits, float* ctx, float* y, int B, int HW, int C,
                                 hipStream_t s);

// GlobalContext in the training step (the forward uses the two kernels of the inference path, out of place):
hipError_t launch_gc_logits(const float* x, const float* wg, const float* bg, float* logits, long long rows, int C, hipStream_t s);
hipError_t launch_gc_pool(const float* x, const float* logits, float* ctx, int B, int HW, int C, hipStream_t s);
// out[b][c] = sum_p (wts ? wts[b][p] : 1) * x[b][p][c]: per-image (weighted) column sums; part >= B * GC_CHUNKS * C floats
constexpr int GC_CHUNKS = 16;
hipError_t launch_gc_wpool(const float* x, const float* wts, float* part, float* out, int B, int HW, int C, hipStream_t s);
// out[b][p][c] = x[b][p][c] + y[b][c]
hipError_t launch_gc_bcast_add(const float* x, const float* y, float* out, int B, int HW, int C, hipStream_t s);
// backward of ctx[b] = sum_p softmax_p(logits[b]) x[b][p]: a[b][p] (the softmax), da[b][p] = dctx[b] . (x[b][p] - ctx[b]), then
// dl = a (da - sum_p a da) in place of da, sum_dl[b] = sum_p dl
hipError_t launch_gc_pool_bwd_weights(const float* x, const float* logits, const float* dctx, const float* ctx, float* a,
                                      float* dl, float* sum_dl, int B, int HW, int C, hipStream_t s);
// dx[b][p][c] = a[b][p] dctx[b][c] + dl[b][p] wg[c]
hipError_t launch_gc_pool_bwd_dx(const float* a, const float* dl, const float* dctx, const float* wg, float* dx, int B, int HW,
                                 int C, hipStream_t s);

hipError_t launch_sum_small(const float* a, int n, float* dst, hipStream_t s);  // dst[0] = sum of n (few) floats
hipError_t launch_bias_add(float* z, const float* bias, long long rows, int C, hipStream_t s);  // z[r][c] += bias[c]
hipError_t launch_scale(const float* a, float* out, size_t n, float alpha, hipStream_t s);      // out = alpha * a
// dx[b][h][w][c] = dy[b][w][c] / H  (backward of the mean over the height, build_feat.py:50-55)
hipError_t launch_mean_h_bwd(const float* dy, float* dx, int B, int H, int W, int C, hipStream_t s);
// Bidirectional LSTM recurrence of the training step (seq_modeling/bilstm.py:14-24).  gates [B*T][8H]: pre-activations
// x W_ih^T + b_ih + b_hh, direction d in columns [4H d, 4H d + 4H); whh_t [2][H][4H]; out [B*T][2H] (forward | reverse);
// saved for the backward pass: sv_gates [B*T][8H] (i, f, g, o after their nonlinearities), sv_c [B*T][2H].
hipError_t launch_bilstm_train_fwd(const float* gates, const float* whh_t, float* out, float* sv_gates, float* sv_c, int B, int T,
                                   int H, hipStream_t s);
// Backward through time: dout [B*T][2H] -> dgates [B*T][8H] (w.r.t. the pre-activations); whh [2][4H][H] raw weights
hipError_t launch_bilstm_train_bwd(const float* dout, const float* sv_gates, const float* sv_c, const float* whh_fwd,
                                   const float* whh_rev, float* dgates, int B, int T, int H, hipStream_t s);
// hprev[d][b*T + t][H] = h of direction d at the step processed before t (zero at its first step), from out [B*T][2H]
hipError_t launch_bilstm_hprev(const float* out, float* hprev_fwd, float* hprev_rev, int B, int T, int H, hipStream_t s);

// ---- training step (train_kernels.hip) --------------------------------------
// Weight gradient ("TN" GEMM, optional filter taps): part[z][tap][m][n] = sum_{p in chunk z} a[p][m] * x[src(p,tap)][n]
struct WgradP {
  const float* a;   // [P][lda]: upstream gradient rows (dz of a convolution / dy of a Linear)
  const float* b;   // geom: NHWC input [B,H,W,ldb];  else [P][ldb]
  float* part;      // [S][taps][M][N]
  long long P;      // rows of a (= B*OH*OW)
  int M, N, lda, ldb;
  int taps, geom;   // geom != 0: rows are output pixels of a convolution with the geometry below
  int H, W, OH, OW, KW, SH, SW, PH, PW;
  int S, chunk;     // split over row chunks of `chunk` rows
  int bf16x3;       // 128 x 128 tiles only: split-bf16 arithmetic (three bf16 MFMAs per product)
  // both set: the operands as split-bf16 records ([row][32 x hi | 32 x lo] per 32-column group, conv_common.h) of a / b --
  // geom convolutions with wgrad_rec_shape(M, N) >= 0 and chunk % 16 == 0 in bf16x3 mode; `zero`: 256 zero bytes
  const uint16_t* a_rec = nullptr;
  const uint16_t* b_rec = nullptr;
  const void* zero = nullptr;
};
hipError_t launch_wgrad(const WgradP& p, hipStream_t s);
// record kernel's block tile: 0 = 128 x 128 (three blocks per CU), 1 = 256 x 128 (two), 2 = 256 x 256 (one), 3 = 128 x 64 (N = 64),
// 4 = 64 x 32 (M = 64, N = 32); -1 = not served
int wgrad_rec_shape(int M, int N);
hipError_t launch_wgrad_reduce(const float* part, float* dst, int S, int taps, int M, int N, int layout, int accumulate,
                               hipStream_t s);
enum { CR_SUM = 0, CR_SUM_SQ = 1, CR_BN_BWD = 2, CR_LN_BWD = 3 };
struct ColRedP {
  const float* a;     // [R][C]
  const float* y;     // CR_BN_BWD: post-ReLU output (nullable = no ReLU)
  const float* z;     // CR_BN_BWD: pre-BN values;  CR_LN_BWD: LayerNorm input
  const float* mean;  // per channel (BN) / per row (LN)
  const float* rstd;
  float* part;        // [chunks][2][C]
  long long R;
  int C, mode;
};
int colreduce_chunks(long long R, int C);
hipError_t launch_colreduce(const ColRedP& p, hipStream_t s);
hipError_t launch_colreduce_final(const float* part, int chunks, int C, float* out0, float* out1, int accumulate,
                                  hipStream_t s);
hipError_t launch_bn_finalize(const float* part, int chunks, int C, long long R, float eps, float momentum, float* mean,
                              float* rstd, float* run_mean, float* run_var, hipStream_t s);
hipError_t launch_bn_apply(const float* z, const float* mean, const float* rstd, const float* g, const float* b,
                           const float* res, float* y, long long R, int C, int relu, hipStream_t s, uint16_t* planes = nullptr);
hipError_t launch_bn_bwd_apply(const float* dy, const float* y, const float* z, const float* mean, const float* rstd,
                               const float* gamma, const float* s0, const float* s1, float* dz, float* gout, long long R,
                               int C, hipStream_t s, uint16_t* planes = nullptr);
enum { EW_COPY = 0, EW_ADD = 1, EW_RELU_BWD = 2, EW_GELU = 3, EW_GELU_BWD = 4, EW_RELU = 5 };
hipError_t launch_ew(const float* a, const float* b, float* out, size_t n, int op, hipStream_t s);
hipError_t launch_maxpool_bwd(const float* x, const float* dy, float* dx, int B, int H, int W, int C, int SH, int SW, int PH,
                              int PW, hipStream_t s, int KW = 2);
hipError_t launch_ln_train(const float* x, const float* g, const float* b, float* y, float* mean, float* rstd, int rows,
                           int D, float eps, hipStream_t s);
hipError_t launch_ln_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* g,
                         const float* add, float* dx, int rows, int D, hipStream_t s);
struct AttnTrainP {
  const float *q, *k, *v;  // row (b*L + i), stride ld*, head offset head*hd
  float* o;                // forward: output;  backward: dO (read)
  float* probs;            // [B][heads][Lq][Lk]; the backward pass overwrites it with dS
  float *dq, *dk, *dv;     // backward outputs (q / k / v strides)
  const int64_t* keytok;   // optional [B][Lk]: keys whose token == pad_id are masked
  int B, heads, hd, Lq, Lk, ldq, ldk, ldv, ldo, causal, pad_id;
  const uint8_t* dropmask; // optional [B][heads][Lq][Lk] keep mask of the attention-probability dropout
  float dropscale;         // 1 / (1 - p)
  int probe = 0;           // timing probe of the backward kernel (D2T_ATTN_BWD_PROBE): bit k set = phase k+1 skipped
};
hipError_t launch_attn_train_fwd(const AttnTrainP& p, hipStream_t s);
hipError_t launch_attn_train_bwd(const AttnTrainP& p, hipStream_t s);
hipError_t launch_embed_train(const float* E, const float* pe, const int64_t* tok, float* x, int rows, int L, int D,
                              float scale, hipStream_t s);
hipError_t launch_embed_bwd(const float* dx, const int64_t* tok, float* dE, int rows, int V, int D, float scale, int pad_id,
                            hipStream_t s);
// mask[i] = 1 with probability 1 - p (Philox4x32-10 keyed by (seed, stream), counter = i / 8), else 0
hipError_t launch_dropout_mask(uint8_t* mask, size_t n, float p, unsigned long long seed, unsigned long long stream_id,
                               hipStream_t s);
// out = a * mask * scale
hipError_t launch_apply_mask(const float* a, const uint8_t* mask, float scale, float* out, size_t n, hipStream_t s);
hipError_t launch_ce_fwd(const float* x, const int64_t* tgt, float* loss, float* lse, int rows, int V, long long ignore,
                         hipStream_t s);
hipError_t launch_ce_bwd(const float* x, const int64_t* tgt, const float* lse, const float* dloss, float* dx, int rows, int V,
                         long long ignore, hipStream_t s);
hipError_t launch_relu_mask(const float* y, uint8_t* m, size_t n, hipStream_t s);
hipError_t launch_pool_argmax(const float* x, uint8_t* k, int B, int H, int W, int C, int SH, int SW, int PH, int PW,
                              hipStream_t s, int KW = 2);
hipError_t launch_pad_cols(const float* src, float* dst, size_t rows, int Cs, int Cd, hipStream_t s);
hipError_t launch_copy2d(const float* src, int lds, float* dst, int ldd, size_t rows, int cols, hipStream_t s);
hipError_t launch_dilate(const float* src, float* dst, int B, int OH, int OW, int C, int DH, int DW, int SH, int SW, int offh,
                         int offw, hipStream_t s);
hipError_t launch_flip_oihw(const float* w, float* out, int Cout, int Cin, int KH, int KW, hipStream_t s);
hipError_t launch_token_rows(const float* src, float* dst, int B, int n, int skip, int D, int scatter, hipStream_t s);
hipError_t launch_sum_rows_strided(const float* x, float* out, int B, long long stride_rows, int row, int D, hipStream_t s);
// posembed.hip -- ViTEncoder's learned table resized to a crop's patch grid (vit_encoder.py:58-95): [GH*GW][D] -> [gh*gw][D]
// by ATen's bicubic rule with scale_h / scale_w = 1 / scale_factor; its transpose (gradient of the table); and the sum of a
// [B][n] gradient over the batch
hipError_t launch_bicubic_table(const float* src, float* dst, int GH, int GW, int gh, int gw, int D, float scale_h,
                                float scale_w, hipStream_t s);
hipError_t launch_bicubic_table_bwd(const float* ddst, float* dsrc, int GH, int GW, int gh, int gw, int D, float scale_h,
                                    float scale_w, hipStream_t s);
hipError_t launch_sum_over_batch(const float* x, float* out, int B, long long stride, long long n, hipStream_t s);
hipError_t launch_stem_raw(const float* img, const float* w, float* z, int B, int H, int W, int Cout, hipStream_t s);
hipError_t launch_stem_wgrad(const float* img, const float* dz, float* part, int B, int H, int W, int Cout, int chunk,
                             int nchunks, hipStream_t s);

// One launch for many device-to-device copies: block b copies chunk b = {src, dst, n floats} of the table.
struct CopyChunk { const float* src; float* dst; long long n; };
hipError_t launch_multi_copy(const CopyChunk* table, int chunks, hipStream_t s);

}  // namespace d2t
