// Shared between engine.hip (inference) and train.hip (training step): the context struct, packed-weight
// records and small host helpers.  Internal to libd2t; the public surface is include/d2t.h.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/d2t.h"
#include "kernels.h"

using namespace d2t;

struct d2t_train_state;  // train.hip

namespace {

constexpr int TOK_PAD = 0, TOK_GO = 1, TOK_END = 2;  // converter/tfm_converter.py:8
const int RESNET_LAYERS[4] = {1, 2, 5, 3};           // feature_extractor/resnet.py:262

struct RawW {
  float* p = nullptr;
  std::vector<int64_t> shape;
  size_t numel = 0;
};
struct ConvW {
  float* w = nullptr;
  uint16_t *w_hi = nullptr, *w_lo = nullptr;  // bf16 split of w for the bf16x3 kernel
  uint16_t *w_h16 = nullptr, *w_l16 = nullptr;  // fp16 hi / lo of w for the fp16x2 kernels (ConvP::f16)
  float* bias = nullptr;
  int Cout = 0, Cin = 0, KH = 0, KW = 0;
  int K2 = 0;  // K rows of a second (1x1) input concatenated behind the taps (Block::c2cat), else 0
};
struct LinW {
  const float* w = nullptr;
  const float* b = nullptr;
  int N = 0, K = 0;
  const uint16_t *w_hi = nullptr, *w_lo = nullptr;  // optional bf16 split of w (large-M linears on the bf16x3 kernel)
};
struct LNW {
  const float* g = nullptr;
  const float* b = nullptr;
};
struct VitBlock {
  LNW n1, n2;
  LinW qkv, proj, fc1, fc2;
};
struct DecLayer {
  LinW sa_in, sa_out, ca_q, ca_out, l1, l2;
  LNW n1, n2, n3;
  float *sa_out_t = nullptr, *ca_q_t = nullptr, *ca_out_t = nullptr;  // [k][n] copies for the row kernel
  // absorbed cross-attention (decode.hip decoder_row_absorbed_kernel): the key projection as stored ([o][c], in ckv_w), the
  // value projection transposed ([c][o]) and its bias
  const float* ca_wk = nullptr;
  float* ca_v_t = nullptr;
  const float* ca_bv = nullptr;
};
struct Block {
  ConvW c1, c2, down;
  bool has_down = false;
  // blocks with a 1x1 shortcut: conv2 and the shortcut as ONE GEMM -- weights [Cout][9 Cmid + Cin] (K rows concatenated), bias
  // b2 + b_down -- for the pipelined split-record kernel (ConvP::in2_hi); w == nullptr when the block has no shortcut
  ConvW c2cat;
};
struct BiLstmW {
  float* wih_cat = nullptr;   // [2*4H][in]  forward rows then reverse rows
  float* bias_cat = nullptr;  // [2*4H]      b_ih + b_hh
  float* whh_t = nullptr;     // [2][H][4H]
  LinW lin;                   // Linear(2H -> out)
  int in = 0;
};
struct AttnW {
  const float* emb = nullptr;     // embedding table [V][H], or nullptr with one-hot targets ...
  const float* tokgate = nullptr; // ... whose gate contribution is row `token` of this [V][4H] table (W_ih[:, H:]^T)
  LinW key;                    // key_proj (Bahdanau: i2h, no bias)
  float *wq_t = nullptr, *wloc = nullptr, *bloc = nullptr, *wx_t = nullptr, *bx = nullptr, *wg_t = nullptr;
  float *wih_t = nullptr, *wic_t = nullptr;
  const float *bq = nullptr, *wscore = nullptr, *bg = nullptr, *bih = nullptr, *bic = nullptr;
  float bscore = 0.f;
  int taps = 0;
};
struct Act {
  float* p;
  int B, H, W, C;
  bool split = false;  // two bf16 planes (hi | lo) in the same buffer instead of fp32
  int fmt = 0;         // split records: 0 = bf16 hi | lo, 1 = one fp16 in the hi half, 2 = fp16 hi | fp16 lo (conv_common.h REC_*)
  size_t numel() const { return (size_t)B * H * W * C; }
  uint16_t* planes() const { return reinterpret_cast<uint16_t*>(p); }
};

}  // namespace

struct d2t_ctx {
  d2t_config cfg;
  std::string err;
  std::map<std::string, RawW> raw;
  std::vector<void*> owned;  // packed buffers (freed on destroy / re-finalize)
  bool finalized = false;
  bool decode_in_flight = false;  // a submitted decode loop may still be running (cleared by a host-synchronising wait)
  bool conv_bf16x3 = true;   // d2t_set_conv_precision: backbone / patch convolutions on the bf16x3 kernel
  // D2T_CONV_FP16X2 (on top of conv_bf16x3): the backbone's feature maps are fp16 records and its split-record convolutions run
  // x16 * w_lo + x16 * w_hi (two MFMAs per product); everything that takes fp32 input stays on the bf16x3 kernels
  bool conv_f16 = false;
  // D2T_CONV_MIXED (round 4; on top of conv_bf16x3, not with conv_f16): the two-MFMA fp16 arithmetic for the first
  // `mixed_units` of the backbone's plain 512 -> 512 units [layer3.1, layer3.2, layer3.3, layer3.4, conv3, layer4.0, layer4.1,
  // layer4.2] only (the K = 4608 layers: 64 % of the step); every other layer stays split-bf16.  The tensors entering and
  // leaving those units travel as fp16 hi | lo pairs (22 bits), so only the MFMA operand is rounded to 11 bits.
  int mixed_units = 0;
  int stream_prio = 0;  // priority the decode streams were acquired with (they return to the process-wide pool: engine.hip)
  int conv_max_blocks = 0;   // d2t_set_reserved_blocks: grid cap of the persistent split-bf16 convolution (0 = none)
  int num_cus = 0;
  int conv_pipelined = 3;    // d2t_set_conv_kernel: 3 = pipelined 256x128 split-bf16 kernel on 16x16x32 MFMAs (default), 0 = 128x128 on 32x32x16 (two per CU)
  int reserved_cus = 0;      // d2t_set_reserved_cus: CUs the pipelined kernel's grid leaves to other streams (decode)
  int device = 0;            // HIP device the context was created on: every stream, event and buffer lives there

  // packed weights
  std::string bb;  // backbone key prefix ("...ConvNet.")
  ConvW stem, conv0_2, conv1, conv2, conv3, conv4_1, conv4_2, patch;
  ConvW vgg[7];      // VGG convs 0,3,6,8,11(+bn12),14(+bn15),18 (feature_extractor/vgg.py:16-41)
  BiLstmW lstm[2];   // seq_modeling/bilstm.py
  AttnW attn;        // prediction_head/seq2seq.py
  std::vector<Block> layers[4];
  GCParams gc[4] = {};  // GlobalContext block after each stage (cfg.gcb)
  float* gc_ws = nullptr; size_t gc_ws_cap = 0;
  const float* pos_embed = nullptr;  // [1+gh*gw][dim]
  int pos_rows = 0;
  float* cls_row = nullptr;  // cls_token + pos_embed[0]
  int pos_GH = 0, pos_GW = 0;  // patch grid of max_dimension (the learned table's own grid)
  struct PosTab { float* p = nullptr; bool valid = false; };
  std::map<std::pair<int, int>, PosTab> pos_interp;  // D2T_VIT_POS_LEARNED_INTERP: the table resized to a crop's grid
  std::vector<VitBlock> vit;
  LNW vit_norm;
  const float* word_embed = nullptr;
  const float* word_pe = nullptr;
  int word_pe_rows = 0;
  std::vector<DecLayer> dec;
  float* ckv_w = nullptr;  // [layers*2*d][d] cross-attention K,V projections of every layer
  uint16_t *ckv_hi = nullptr, *ckv_lo = nullptr;  // its bf16 split
  float* ckv_b = nullptr;
  LinW out_proj;

  // workspace
  float* act[4] = {nullptr, nullptr, nullptr, nullptr};
  size_t act_cap = 0;
  std::map<std::pair<int, int>, float*> pe2d;  // PositionalEncoding2D crops [h*w][C]
  // decoder state
  static constexpr int MAXC = 4;                                     // decode chains / memory slots at most
  float* ckv2[MAXC] = {}; size_t ckv2_cap[MAXC] = {};                // memory copy (or projected cross K/V) per slot: a decode reads one
  float* ckv = nullptr;                                              // the slot the current decode reads
  // d_model 256 / 8 heads: the decode attends over the encoder memory itself (absorbed K / V projections); the slots then
  // hold a COPY OF THE MEMORY [B][T][d] instead of the projected K / V of every layer [layers*2][B][heads][T][hd]
  bool dec_absorbed = false;
  hipEvent_t ev_done[MAXC] = {};                                     // decode that used slot i has finished
  bool ev_done_valid[MAXC] = {};
  unsigned decode_seq = 0;
  // serving tickets: every asynchronous decode gets the next ticket and records ticket_ev[ticket % N] on its decode stream
  // when its outputs are complete (d2t_decode_last_ticket / d2t_decode_wait_ticket / d2t_decode_query)
  static constexpr int TICKET_RING = 64;
  int64_t last_ticket = 0;
  hipEvent_t ticket_ev[TICKET_RING] = {};
  int* h_steps = nullptr;               // pinned [TICKET_RING][64]: per-batch step counts of early-exit decodes
  int ticket_batches[TICKET_RING] = {};  // batches of that decode's group (< 0: not an early-exit decode)
  float* skv = nullptr; size_t skv_cap = 0;
  float* skv_alt = nullptr; size_t skv_alt_cap = 0;  // beam: reorder target (ping-pong with skv_cur)
  float* skv_cur = nullptr;                          // cache decode_step reads / appends
  float* beam_ws = nullptr; size_t beam_ws_cap = 0;  // beam logits / scores / tokens / top-k
  char* h_beam = nullptr; size_t h_beam_cap = 0;     // pinned host mirror of the device-side beam search's result block
  bool cross_fp32 = false;  // probe builds: the greedy decode's cross-attention on the fp32 MFMA instead of split-bf16
  // beam search: 1 = one cross-attention block per SAMPLE serving all its hypotheses from one staged memory tile
  // (d2t_set_beam_shared_tile; measured slower than one block per hypothesis row at 128 samples x 5: DESIGN.md 5.4), 0 = per row
  int beam_shared_tile = 0;
  int no_shortcut_fusion = 0;  // debug / A-B: 1 = the 1x1 shortcuts as their own kernels (d2t_set_conv_fusion)
  int no_pool_fusion = 0;  // debug / A-B: 1 = the two 2x2 max-pools as their own kernels (d2t_set_conv_fusion)
  float* beam_qp = nullptr; size_t beam_qp_cap = 0;  // beam, absorbed cross-attention: q' / context rows + LN1 rows (decode.hip)
  float* dws = nullptr; size_t dws_cap = 0;
  int* dstate = nullptr;   // [0]=step [1]=end_count [2]=steps_done [3..]=ended[B]
  size_t dstate_cap = 0;
  int* h_pinned = nullptr;
  void* zero_page = nullptr;  // 256 zero bytes: out-of-image taps of the split-bf16 convolution
  hipStream_t dstream = nullptr;
  // Second decode chain (own stream, self-attention cache, workspace, state): with two chains the decode
  // loops of consecutive async batches run side by side.  The members above are the ACTIVE chain; the
  // inactive ones are parked here (select_chain swaps).
  struct Chain { hipStream_t stream = nullptr; float* skv = nullptr; size_t skv_cap = 0; float* dws = nullptr;
                 size_t dws_cap = 0; int* dstate = nullptr; size_t dstate_cap = 0;
                 float* out = nullptr; size_t out_cap = 0; } chains[MAXC];  // chains[active_chain] is stale: its state lives in the members above
  // async decodes write tokens / logits here (fixed addresses, so one captured graph per chain serves every
  // caller buffer) and copy them out afterwards
  float* dout = nullptr; size_t dout_cap = 0;
  int active_chain = 0, n_chains = 1;
  hipEvent_t ev_in = nullptr;
  struct GraphKey { int B, T, steps; const void* tok; const void* logits; const void* ckv; const void* dws; const void* skv; const void* dstate; long long variant; };
  struct GraphEnt { GraphKey key; hipGraphExec_t exec; };
  std::vector<GraphEnt> graphs;  // small cache of captured decode steps (most recent last)
  // kernel timing log (d2t_profile_*)
  bool profiling = false;
  struct ProfRec { int M, N, K; hipEvent_t a, b; };
  std::vector<ProfRec> prof;
  // debug timeline of the decode kernels (env D2T_DECODE_TRACE): [slot][2] ticks, one slot per kernel node of a captured loop
  unsigned long long* dtrace = nullptr;
  int dtrace_next = 0;
  static constexpr int DTRACE_SLOTS = 8192;
  d2t_train_state* train = nullptr;  // lazily created by d2t_train_* (train.hip)
};

namespace {

int fail(d2t_ctx* c, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (c) c->err = buf;
  return code;
}
#define HIPCHK(c, expr)                                                                        \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess) return fail(c, D2T_EHIP, "%s: %s", #expr, hipGetErrorString(e_));   \
  } while (0)

int dev_alloc(d2t_ctx* c, void** p, size_t bytes) {
  if (hipMalloc(p, bytes ? bytes : 16) != hipSuccess) return fail(c, D2T_ENOMEM, "hipMalloc(%zu) failed", bytes);
  return D2T_OK;
}
// grow-only buffer; growing synchronises the device (never during graph capture)
template <typename T>
int ensure(d2t_ctx* c, T** p, size_t* cap, size_t bytes) {
  if (*cap >= bytes && *p) return D2T_OK;
  if (*p) { hipDeviceSynchronize(); hipFree(*p); *p = nullptr; }
  int rc = dev_alloc(c, reinterpret_cast<void**>(p), bytes);
  if (rc) return rc;
  *cap = bytes;
  return D2T_OK;
}

// Every entry point that takes a context runs on the context's device, whatever device is current for the calling
// thread (two Models on two GPUs in one process; a Model on cuda:1 without torch.cuda.set_device): switch on entry,
// restore on return.
struct DevGuard {
  int prev = -1;
  explicit DevGuard(const d2t_ctx* c) {
    if (!c) return;
    int cur = -1;
    if (hipGetDevice(&cur) == hipSuccess && cur != c->device && hipSetDevice(c->device) == hipSuccess) prev = cur;
  }
  ~DevGuard() { if (prev >= 0) hipSetDevice(prev); }
  DevGuard(const DevGuard&) = delete;
  DevGuard& operator=(const DevGuard&) = delete;
};
// a caller buffer must live on the context's device (raw pointers carry no device tag of their own)
int check_dev_ptr(d2t_ctx* c, const void* p, const char* what) {
  hipPointerAttribute_t a;
  if (!p || hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return D2T_OK; }  // unregistered host memory etc.
  if (a.type == hipMemoryTypeDevice && a.device != c->device)
    return fail(c, D2T_EINVAL, "%s is on HIP device %d but this context lives on device %d", what, a.device, c->device);
  return D2T_OK;
}

const RawW* find(d2t_ctx* c, const std::string& k) {
  auto it = c->raw.find(k);
  return it == c->raw.end() ? nullptr : &it->second;
}
int need(d2t_ctx* c, const std::string& k, const RawW** out, std::vector<int64_t> shape = {}) {
  const RawW* r = find(c, k);
  if (!r) return fail(c, D2T_ESTATE, "missing weight '%s'", k.c_str());
  if (!shape.empty() && r->shape != shape) {
    std::string s;
    for (auto v : r->shape) s += std::to_string(v) + ",";
    return fail(c, D2T_EINVAL, "weight '%s' has shape [%s]", k.c_str(), s.c_str());
  }
  *out = r;
  return D2T_OK;
}

}  // namespace

// engine.hip: PositionalEncoding2D crop [h][w][C] (cached per (h, w) in the context)
int d2t_internal_pe2d(d2t_ctx* c, int h, int w, int C, hipStream_t s, const float** out);
// engine.hip: launch_conv bracketed by HIP events while d2t_profile_enable is on (training GEMMs report through it too)
hipError_t d2t_internal_conv_timed(d2t_ctx* c, const ConvP& p, hipStream_t s);

// ViTEncoder.interpolating_pos_embedding (seq_modeling/vit_encoder.py:58-95): how a crop with patch grid gh x gw reads
// the learned [1 + GH*GW][D] table.  interp == false: the table as it is (the grids agree, or -- :66-67 -- the token
// counts agree and the padded feature map is square).  Otherwise F.interpolate's scale factors are (gh + 0.1) / GH and
// (gw + 0.1) / GW, computed in double as Python does, and ATen maps coordinates with float(1 / scale_factor).
struct PosGrid { bool interp; float sh, sw; };
inline PosGrid pos_grid(const d2t_config& g, int GH, int GW, int gh, int gw) {
  PosGrid r{false, 1.f, 1.f};
  if (g.vit_pos != D2T_VIT_POS_LEARNED_INTERP || (gh == GH && gw == GW)) return r;  // HybridEmbed's flag, patchembed.py:140
  if (gh * gw == GH * GW && gh * g.patch_h == gw * g.patch_w) return r;
  r.interp = true;
  r.sh = (float)(1.0 / (((double)gh + 0.1) / (double)GH));
  r.sw = (float)(1.0 / (((double)gw + 0.1) / (double)GW));
  return r;
}
