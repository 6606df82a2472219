// Training step of the recognizer: Model.forward under module.train() and its backward.
//
//   d2t_train_forward   image, teacher tokens -> logits [B][L][V]; BatchNorm on batch statistics (running statistics
//                       updated in the engine's copies), teacher-forced decoder pass with causal + PAD key-padding
//                       masks (prediction_head/tfm.py:103-118); every intermediate the backward needs stays on a tape
//   d2t_train_backward  dlogits -> gradient of every trainable parameter (what loss.backward() leaves in .grad,
//                       engine/training.py:137); read back with d2t_train_grad
//
// Supported stack: HybridViT (ResNet backbone + ViTEncoderV3) or ResNet+None encoders with the TFM head; dropout 0.
// All arithmetic fp32: GEMMs and convolutions (forward and data gradients) on the fp32-MFMA implicit-GEMM kernel
// (conv_mfma.hip), weight gradients on the fp32-MFMA TN kernel (train_kernels.hip).  The network is recorded as a
// tape of nodes over row-major [rows][cols] tensors (NHWC maps are [B*H*W][C]); the backward walks it in reverse and
// accumulates into tensor gradients, so residual fan-out needs no special cases.
#include "ctx.h"

namespace {

const int RESNET_BLOCKS[4] = {1, 2, 5, 3};  // feature_extractor/resnet.py:262

struct Arena {
  struct Blk { char* p; size_t cap; };
  std::vector<Blk> blks;
  size_t cur = 0, off = 0;
  void reset() { cur = 0; off = 0; }
  float* alloc(size_t floats) {
    const size_t bytes = (floats * 4 + 255) & ~(size_t)255;
    while (cur < blks.size() && off + bytes > blks[cur].cap) { ++cur; off = 0; }
    if (cur == blks.size()) {
      const size_t cap = std::max(bytes, (size_t)256 << 20);
      void* q = nullptr;
      if (hipMalloc(&q, cap) != hipSuccess) return nullptr;
      blks.push_back({(char*)q, cap});
      off = 0;
    }
    float* r = reinterpret_cast<float*>(blks[cur].p + off);
    off += bytes;
    return r;
  }
  void release() {
    for (auto& b : blks) hipFree(b.p);
    blks.clear();
    reset();
  }
};

struct TT {  // tensor on the tape
  float* p = nullptr;
  long long rows = 0;
  int cols = 0;
  int B = 0, H = 0, W = 0;  // NHWC maps: rows = B*H*W
  float* grad = nullptr;
  const uint16_t* planes = nullptr;  // split-bf16 records of p, when the producing kernel wrote them alongside (bf16x3 mode)
};

enum Kind { N_CONV, N_POOL, N_LINEAR, N_LN, N_ATTN, N_GELU, N_EMBED, N_TOKENS, N_ADDCONST, N_DROPOUT, N_ADD, N_LSTM, N_RELU,
            N_GCPOOL, N_BCAST, N_MEANH, N_BILSTM };

struct Node {
  Kind kind;
  int in = -1, in2 = -1, out = -1;  // tensor ids (in2: residual / second operand)
  // N_CONV: conv (+ BatchNorm) (+ residual in2) (+ ReLU)
  std::string wkey, bnkey;
  int KH = 0, KW = 0, SH = 1, SW = 1, PH = 0, PW = 0;
  float *z = nullptr, *mean = nullptr, *rstd = nullptr;
  bool relu = false, stem = false;
  // N_LINEAR: out = act(in @ W[woff : woff+N]^T + b) (+ in2)
  std::string bkey;
  int N = 0, K = 0, woff = 0;
  bool nobias = false;  // nn.Linear(..., bias=False)
  // N_LN
  std::string gkey;
  float eps = 0.f;
  // N_ATTN: q / k / v column offsets inside tensors in (q) and in2 (k, v)
  int qoff = 0, koff = 0, voff = 0, heads = 0, hd = 0, Lq = 0, Lk = 0, nb = 0, causal = 0;
  const int64_t* keytok = nullptr;
  float* probs = nullptr;
  // N_TOKENS: out[b][0] = cls + pos[0], out[b][1+i] = in[b][i] + pos[1+i]   (in = patch tokens)
  int ntok = 0;
  // N_DROPOUT / N_ATTN: keep mask (bytes) and 1 / (1 - p)
  const uint8_t* mask = nullptr;
  float mscale = 1.f;
  // N_LSTM (teacher-forced LSTM-attention decoder): in = memory, in2 = key projection, out = logits; saved state
  float* aux[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
};

}  // namespace

struct d2t_train_state {
  // d2t_train_gather: copy-table staging (pinned + device mirror), four sets in rotation so that the host waits for an
  // upload queued four calls ago, never for the work just ahead of it
  struct GatherStage { void *h = nullptr, *d = nullptr; size_t cap = 0; hipEvent_t ev = nullptr; bool pending = false; } gather[4];
  unsigned gather_calls = 0;
  Arena tape;
  std::vector<TT> t;
  std::vector<Node> nodes;
  std::map<std::string, float*> grads;  // persistent, one buffer per trainable parameter
  std::map<std::string, hipEvent_t> ready;  // recorded on the compute stream after the last kernel that writes a gradient
  std::vector<std::string> touched;         // gradients written by the node being processed
  float* part = nullptr; size_t part_cap = 0;   // wgrad / column-reduction partial sums
  float* scratch = nullptr; size_t scratch_cap = 0;
  const int64_t* tgt = nullptr;
  const float* image = nullptr;
  // dropout of nn.TransformerDecoderLayer (d2t_train_set_dropout): Philox masks keyed by (seed, forward call, site)
  float drop_p = 0.f;
  unsigned long long drop_seed = 0, drop_calls = 0, drop_site = 0;
  std::vector<uint8_t> teacher_flags;  // d2t_train_set_teacher_flags (scheduled sampling of the LSTM head); empty = all teacher
  std::vector<std::pair<const uint8_t*, size_t>> masks;  // in creation order (d2t_train_read_mask)
  int B = 0, H = 0, W = 0, L = 0;
  int logits_id = -1;
  bool have_forward = false;
  ~d2t_train_state() {
    tape.release();
    for (auto& kv : grads) hipFree(kv.second);
    for (auto& kv : ready) hipEventDestroy(kv.second);
    for (auto& g : gather) {
      if (g.h) hipHostFree(g.h);
      if (g.d) hipFree(g.d);
      if (g.ev) hipEventDestroy(g.ev);
    }
    if (part) hipFree(part);
    if (scratch) hipFree(scratch);
  }
};

namespace {

#define TCHK(expr)                                                                               \
  do {                                                                                           \
    hipError_t e_ = (expr);                                                                      \
    if (e_ != hipSuccess) return fail(c, D2T_EHIP, "%s: %s", #expr, hipGetErrorString(e_));      \
  } while (0)
#define RC(expr)            \
  do {                      \
    int rc_ = (expr);       \
    if (rc_) return rc_;    \
  } while (0)

struct Tr {  // builder / runner bound to one context and stream
  d2t_ctx* c;
  d2t_train_state* st;
  hipStream_t s;

  int alloc(float** p, size_t floats) {
    *p = st->tape.alloc(floats);
    return *p ? D2T_OK : fail(c, D2T_ENOMEM, "training tape: hipMalloc failed");
  }
  int new_tensor(long long rows, int cols, int* id, int B = 0, int H = 0, int W = 0, float* into = nullptr) {
    TT x;
    x.rows = rows; x.cols = cols; x.B = B; x.H = H; x.W = W;
    if (into) x.p = into; else RC(alloc(&x.p, (size_t)rows * cols));
    st->t.push_back(x);
    *id = (int)st->t.size() - 1;
    return D2T_OK;
  }
  int raw(const std::string& k, const float** p, size_t numel = 0) {
    const RawW* r;
    RC(need(c, k, &r));
    if (numel && r->numel != numel) return fail(c, D2T_EINVAL, "weight '%s' has %zu elements, expected %zu", k.c_str(), r->numel, numel);
    *p = r->p;
    return D2T_OK;
  }
  int grad_buf(const std::string& k, float** p) {
    st->touched.push_back(k);
    auto it = st->grads.find(k);
    if (it != st->grads.end()) { *p = it->second; return D2T_OK; }
    const RawW* r;
    RC(need(c, k, &r));
    void* q;
    RC(dev_alloc(c, &q, r->numel * 4));
    TCHK(hipMemsetAsync(q, 0, r->numel * 4, s));
    st->grads[k] = (float*)q;
    hipEvent_t ev;
    TCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    st->ready[k] = ev;
    *p = (float*)q;
    return D2T_OK;
  }
  int ensure_part(size_t floats) { return ensure(c, &st->part, &st->part_cap, floats * 4); }
  int ensure_scratch(size_t floats) { return ensure(c, &st->scratch, &st->scratch_cap, floats * 4); }

  // out[M][N] = act(a[M][K] @ w[N][K]^T + bias) + res      (fp32 MFMA GEMM; K % 32 == 0)
  int gemm_nt(const float* a, const float* w, const float* bias, const float* res, float* out, long long M, int N, int K,
              int act) {
    ConvP p{};
    p.in = a; p.w = w; p.bias = bias; p.res = res; p.out = out;
    p.B = 1; p.H = 1; p.W = (int)M; p.Cin = K; p.OH = 1; p.OW = (int)M; p.Cout = N;
    p.KH = p.KW = p.SH = p.SW = 1; p.M = (int)M; p.K = K; p.act = act;
    TCHK(d2t_internal_conv_timed(c, p, s));
    return D2T_OK;
  }
  // split-bf16 mode: hand the convolution its input as hi / lo records (a copy in the step's arena), which moves it from
  // the kernel that splits fp32 activations inside its K loop to the LDS-DMA kernels (conv_bf16x3p.hip: ~25 % faster; the
  // copy costs one pass over the input).  Same split, same three-MFMA products.
  bool want_planes(long long rows, int C) const {
    static const bool off = D2T_PROBE_ENV_STR("D2T_TRAIN_SPLIT_INPUT") && D2T_PROBE_ENV("D2T_TRAIN_SPLIT_INPUT") == 0;
    return !off && c->conv_bf16x3 && c->zero_page && C % 32 == 0 && rows * C * 2 <= 0x7fffffffLL;
  }
  // records written by the producer of a tensor (BatchNorm apply kernels), or nullptr
  int new_planes(long long rows, int C, uint16_t** out) {
    *out = nullptr;
    if (!want_planes(rows, C)) return D2T_OK;
    float* pl;
    RC(alloc(&pl, (size_t)rows * C));
    *out = reinterpret_cast<uint16_t*>(pl);
    return D2T_OK;
  }
  int split_input(ConvP* p, const float* x, long long rows, int C, const uint16_t* ready = nullptr) {
    if (!want_planes(rows, C) || p->KH * p->KW > 16) return D2T_OK;
    if (!ready) {
      float* planes;
      RC(alloc(&planes, (size_t)rows * C));
      TCHK(launch_split_act(x, reinterpret_cast<uint16_t*>(planes), (size_t)rows, C, s));
      ready = reinterpret_cast<const uint16_t*>(planes);
    }
    p->in = nullptr;
    p->in_hi = ready;
    p->zero16 = c->zero_page;
    p->pipelined = c->conv_pipelined;
    p->split_tail = 1;
    return D2T_OK;
  }
  // dst[c] (+)= column sums of a[R][C]
  int colsum(const float* a, long long R, int C, float* dst) {
    const int chunks = colreduce_chunks(R, C);
    RC(ensure_part((size_t)chunks * 2 * C));
    ColRedP p{};
    p.a = a; p.part = st->part; p.R = R; p.C = C; p.mode = CR_SUM;
    TCHK(launch_colreduce(p, s));
    TCHK(launch_colreduce_final(st->part, chunks, C, dst, nullptr, 0, s));
    return D2T_OK;
  }
  // dW[M][N](taps) = a^T (x) b  with the split-K two-pass reduction
  int wgrad(const float* a, int lda, const float* b, int ldb, long long P, int M, int N, int taps, const Node* geom,
            const TT* xin, const TT* yout, float* dst, int layout, const uint16_t* a_rec = nullptr,
            const uint16_t* b_rec = nullptr) {
    // both operands exist as split-bf16 records (a convolution between BatchNorm layers): the LDS-DMA kernel
    static const bool rec_off = D2T_PROBE_ENV_STR("D2T_WGRAD_REC") && D2T_PROBE_ENV("D2T_WGRAD_REC") == 0;
    const int shape = !rec_off && a_rec && b_rec && geom && c->conv_bf16x3 && c->zero_page && lda == M && ldb == N ? wgrad_rec_shape(M, N) : -1;
    const bool rec = shape >= 0;
    static const int TM[5] = {128, 256, 256, 128, 64}, TN[5] = {128, 128, 256, 64, 32}, SLOTS[5] = {768, 512, 256, 1024, 2048};
    const int tile = (M <= 64 || N <= 64) ? 64 : 128;
    const long long tiles = rec ? (long long)(M / TM[shape]) * (N / TN[shape]) * taps
                                : (long long)((M + tile - 1) / tile) * ((N + tile - 1) / tile) * taps;
    // split the rows into S chunks so that tiles * S blocks fill whole rounds of the block slots (two 64 KB-LDS blocks per
    // CU on 256 CUs; the record kernel: one to eight per CU by tile shape): among the S that give >= ~2 rounds pick the
    // one wasting least of its last round
    const long long slots = rec ? SLOTS[shape] : 512;
    const long long smax = std::max<long long>(1, (P + 511) / 512);
    long long S = 1;
    double best = -1.0;
    for (long long cand = 1; cand <= std::min<long long>(smax, 64); ++cand) {
      const long long blocks = tiles * cand, rounds = (blocks + slots - 1) / slots;
      double eff = (double)blocks / (double)(rounds * slots);
      if (blocks < 2 * slots && cand < smax) eff *= 0.5;  // prefer enough blocks to hide the tile prologue / epilogue
      if (eff > best + 1e-9) { best = eff; S = cand; }
    }
    // the record kernel's K loops are cheap to keep long and every extra split is another [taps][M][N] partial to write
    // and sum: one round of blocks (measured against 2 ... 4 rounds and the rule above: 113.8 vs 114.2 ... 117.8 ms per step)
    static const int rec_rounds = D2T_PROBE_ENV_STR("D2T_WGRAD_ROUNDS") ? std::max(1, D2T_PROBE_ENV("D2T_WGRAD_ROUNDS")) : 1;
    if (rec) S = std::max<long long>(1, std::min<long long>(smax, rec_rounds * slots / tiles));
    long long chunk = ((P + S - 1) / S + 31) / 32 * 32;
    S = (P + chunk - 1) / chunk;
    RC(ensure_part((size_t)S * taps * M * N));
    WgradP p{};
    p.a = a; p.b = b; p.part = st->part; p.P = P; p.M = M; p.N = N; p.lda = lda; p.ldb = ldb; p.taps = taps;
    p.KW = 1; p.S = (int)S; p.chunk = (int)chunk;
    p.bf16x3 = c->conv_bf16x3 ? 1 : 0;
    if (rec) { p.a_rec = a_rec; p.b_rec = b_rec; p.zero = c->zero_page; }
    if (geom) {
      p.geom = 1; p.H = xin->H; p.W = xin->W; p.OH = yout->H; p.OW = yout->W;
      p.KW = geom->KW; p.SH = geom->SH; p.SW = geom->SW; p.PH = geom->PH; p.PW = geom->PW;
    }
    TCHK(launch_wgrad(p, s));
    TCHK(launch_wgrad_reduce(st->part, dst, (int)S, taps, M, N, layout, 0, s));
    return D2T_OK;
  }
  // tensor gradient accumulation: first contribution takes the buffer, later ones add
  int add_grad(int id, float* g) {
    TT& x = st->t[id];
    if (!x.grad) { x.grad = g; return D2T_OK; }
    TCHK(launch_ew(x.grad, g, x.grad, (size_t)x.rows * x.cols, EW_ADD, s));
    return D2T_OK;
  }
  int own_grad(int id) {  // make sure tensor `id` has a gradient buffer (contents unspecified)
    TT& x = st->t[id];
    if (!x.grad) RC(alloc(&x.grad, (size_t)x.rows * x.cols));
    return D2T_OK;
  }

  // ---------------- forward builders ----------------
  // conv (+BN batch stats) (+res) (+relu).  bnkey empty: bias from wkey + ".bias", no normalisation.
  int conv_bn(int in, const std::string& wkey, const std::string& bnkey, int Cout, int KH, int KW, int SH, int SW, int PH,
              int PW, bool relu, int res, int* out, int OH = -1, int OW = -1) {
    const TT x = st->t[in];
    if (OH < 0) { OH = (x.H + 2 * PH - KH) / SH + 1; OW = (x.W + 2 * PW - KW) / SW + 1; }
    const long long P = (long long)x.B * OH * OW;
    Node n;
    n.kind = N_CONV; n.in = in; n.in2 = res; n.wkey = wkey; n.bnkey = bnkey;
    n.KH = KH; n.KW = KW; n.SH = SH; n.SW = SW; n.PH = PH; n.PW = PW; n.relu = relu; n.N = Cout; n.K = KH * KW * x.cols;
    const float* w;
    RC(raw(wkey + ".weight", &w, (size_t)Cout * x.cols * KH * KW));
    // raw OIHW -> the kernel's packed OHWI order (no BN folding in training mode); in bf16x3 mode also its bf16
    // hi / lo planes, which make launch_conv take the split-bf16 kernel (fp32 activations split on the fly)
    const size_t wn = (size_t)Cout * n.K;
    RC(ensure_scratch(2 * wn + Cout + 64));
    float* wp = st->scratch;
    const float* bias = nullptr;
    if (bnkey.empty()) RC(raw(wkey + ".bias", &bias, Cout));
    TCHK(launch_pack_conv(w, nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, wp, wp + wn, Cout, x.cols, KH, KW, s));
    RC(alloc(&n.z, (size_t)P * Cout));
    ConvP p{};
    if (c->conv_bf16x3) {
      uint16_t* hi = reinterpret_cast<uint16_t*>(wp + wn + Cout + 16);
      TCHK(launch_split_bf16(wp, hi, hi + wn, wn, s));
      p.w_hi = hi; p.w_lo = hi + wn;
    }
    p.in = x.p; p.w = wp; p.bias = bias; p.out = n.z;
    p.B = x.B; p.H = x.H; p.W = x.W; p.Cin = x.cols; p.OH = OH; p.OW = OW; p.Cout = Cout;
    p.KH = KH; p.KW = KW; p.SH = SH; p.SW = SW; p.PH = PH; p.PW = PW; p.M = (int)P; p.K = n.K; p.act = ACT_NONE;
    RC(split_input(&p, x.p, x.rows, x.cols, x.planes));
    if (p.in_hi && !x.planes) st->t[in].planes = p.in_hi;  // records made here: the weight gradient reads them again
    TCHK(d2t_internal_conv_timed(c, p, s));
    if (bnkey.empty()) {
      RC(new_tensor(P, Cout, out, x.B, OH, OW, n.z));
    } else {
      RC(bn_forward(n, P, Cout, res));
      RC(new_tensor(P, Cout, out, x.B, OH, OW));
      const float *g, *b;
      RC(raw(bnkey + ".weight", &g, Cout));
      RC(raw(bnkey + ".bias", &b, Cout));
      uint16_t* pl;  // the next convolution's operand records, written by the same pass
      RC(new_planes(P, Cout, &pl));
      TCHK(launch_bn_apply(n.z, n.mean, n.rstd, g, b, res >= 0 ? st->t[res].p : nullptr, st->t[*out].p, P, Cout, relu, s, pl));
      st->t[*out].planes = pl;
    }
    n.out = *out;
    st->nodes.push_back(n);
    return D2T_OK;
  }
  int bn_forward(Node& n, long long P, int C, int /*res*/) {
    RC(alloc(&n.mean, C));
    RC(alloc(&n.rstd, C));
    const int chunks = colreduce_chunks(P, C);
    RC(ensure_part((size_t)chunks * 2 * C));
    ColRedP p{};
    p.a = n.z; p.part = st->part; p.R = P; p.C = C; p.mode = CR_SUM_SQ;
    TCHK(launch_colreduce(p, s));
    const RawW *rm, *rv;
    RC(need(c, n.bnkey + ".running_mean", &rm));
    RC(need(c, n.bnkey + ".running_var", &rv));
    TCHK(launch_bn_finalize(st->part, chunks, C, P, 1e-5f, 0.1f, n.mean, n.rstd, rm->p, rv->p, s));
    return D2T_OK;
  }
  int stem(const float* img, int B, int H, int W, const std::string& wkey, const std::string& bnkey, int* out) {
    const int Cout = 32;
    const long long P = (long long)B * H * W;
    Node n;
    n.kind = N_CONV; n.stem = true; n.wkey = wkey; n.bnkey = bnkey; n.relu = true; n.N = Cout; n.K = 9;
    n.KH = n.KW = 3; n.PH = n.PW = 1;
    const float* w;
    RC(raw(wkey + ".weight", &w, (size_t)Cout * 9));
    RC(alloc(&n.z, (size_t)P * Cout));
    TCHK(launch_stem_raw(img, w, n.z, B, H, W, Cout, s));
    RC(bn_forward(n, P, Cout, -1));
    RC(new_tensor(P, Cout, out, B, H, W));
    const float *g, *b;
    RC(raw(bnkey + ".weight", &g, Cout));
    RC(raw(bnkey + ".bias", &b, Cout));
    TCHK(launch_bn_apply(n.z, n.mean, n.rstd, g, b, nullptr, st->t[*out].p, P, Cout, 1, s));
    n.out = *out;
    st->nodes.push_back(n);
    return D2T_OK;
  }
  int pool(int in, int SH, int SW, int PH, int PW, int* out, int KW = 2) {  // window 2 x KW
    const TT x = st->t[in];
    const int OH = (x.H + 2 * PH - 2) / SH + 1, OW = (x.W + 2 * PW - KW) / SW + 1;
    RC(new_tensor((long long)x.B * OH * OW, x.cols, out, x.B, OH, OW));
    if (KW == 2) TCHK(launch_maxpool(x.p, st->t[*out].p, x.B, x.H, x.W, x.cols, SH, SW, PH, PW, s));
    else TCHK(launch_maxpool_k(x.p, st->t[*out].p, x.B, x.H, x.W, x.cols, 2, KW, SH, SW, PH, PW, s));
    Node n;
    n.kind = N_POOL; n.in = in; n.out = *out; n.SH = SH; n.SW = SW; n.PH = PH; n.PW = PW; n.KW = KW;
    st->nodes.push_back(n);
    return D2T_OK;
  }
  int backbone(const float* img, int B, int H, int W, const std::string& p, int* out) {
    int x;
    RC(stem(img, B, H, W, p + "conv0_1", p + "bn0_1", &x));
    RC(conv_bn(x, p + "conv0_2", p + "bn0_2", 64, 3, 3, 1, 1, 1, 1, true, -1, &x));
    RC(pool(x, 2, 2, 0, 0, &x));
    const int chans[4] = {128, 256, 512, 512};
    auto stage = [&](int li) -> int {
      for (int i = 0; i < RESNET_BLOCKS[li]; ++i) {
        const std::string bp = p + "layer" + std::to_string(li + 1) + "." + std::to_string(i);
        int t, r = x;
        RC(conv_bn(x, bp + ".conv1", bp + ".bn1", chans[li], 3, 3, 1, 1, 1, 1, true, -1, &t));
        if (find(c, bp + ".downsample.0.weight"))
          RC(conv_bn(x, bp + ".downsample.0", bp + ".downsample.1", chans[li], 1, 1, 1, 1, 0, 0, false, -1, &r));
        RC(conv_bn(t, bp + ".conv2", bp + ".bn2", chans[li], 3, 3, 1, 1, 1, 1, true, r, &x));
      }
      if (c->cfg.gcb)  // a GlobalContext block closes the stage (resnet.py:200-201): Sequential index = number of blocks
        RC(global_context(x, p + "layer" + std::to_string(li + 1) + "." + std::to_string(RESNET_BLOCKS[li]), &x));
      return D2T_OK;
    };
    RC(stage(0));
    RC(conv_bn(x, p + "conv1", p + "bn1", 128, 3, 3, 1, 1, 1, 1, true, -1, &x));
    RC(pool(x, 2, 2, 0, 0, &x));
    RC(stage(1));
    RC(conv_bn(x, p + "conv2", p + "bn2", 256, 3, 3, 1, 1, 1, 1, true, -1, &x));
    RC(pool(x, 2, 1, 0, 1, &x));
    RC(stage(2));
    RC(conv_bn(x, p + "conv3", p + "bn3", 512, 3, 3, 1, 1, 1, 1, true, -1, &x));
    RC(stage(3));
    RC(conv_bn(x, p + "conv4_1", p + "bn4_1", 512, 2, 2, 2, 1, 0, 1, true, -1, &x));
    RC(conv_bn(x, p + "conv4_2", p + "bn4_2", 512, 2, 2, 1, 1, 0, 0, true, -1, &x));
    *out = x;
    return D2T_OK;
  }

  // VGG_FeatureExtractor.forward (feature_extractor/vgg.py:16-44): Conv2d(1, 64) + ReLU, pool, Conv2d(64, 128) + ReLU, pool,
  // two convolutions + ReLU, (2,1) pool, two bias-free convolutions with BatchNorm + ReLU, (2,1) pool, a 2x2 convolution + ReLU
  int vgg(const float* img, int B, int H, int W, const std::string& p, int* out) {
    int x;
    {  // first layer: one input channel, the stem kernels (weight gradient included) with a bias instead of a BatchNorm
      const float *w, *b;
      RC(raw(p + "0.weight", &w, (size_t)64 * 9));
      RC(raw(p + "0.bias", &b, 64));
      const long long P = (long long)B * H * W;
      Node n;
      n.kind = N_CONV; n.wkey = p + "0"; n.KH = n.KW = 3; n.PH = n.PW = 1; n.N = 64; n.K = 9; n.stem = true;
      RC(new_tensor(P, 64, &x, B, H, W));
      n.z = st->t[x].p;
      TCHK(launch_stem_raw(img, w, n.z, B, H, W, 64, s));
      TCHK(launch_bias_add(n.z, b, P, 64, s));
      n.out = x;
      st->nodes.push_back(n);
      RC(relu(x, &x));
    }
    auto cr = [&](const char* key, const char* bn, int Cout, int k, int pad) -> int {
      RC(conv_bn(x, p + key, bn ? p + bn : std::string(), Cout, k, k, 1, 1, pad, pad, bn != nullptr, -1, &x));
      if (!bn) RC(relu(x, &x));
      return D2T_OK;
    };
    RC(pool(x, 2, 2, 0, 0, &x));
    RC(cr("3", nullptr, 128, 3, 1));
    RC(pool(x, 2, 2, 0, 0, &x));
    RC(cr("6", nullptr, 256, 3, 1));
    RC(cr("8", nullptr, 256, 3, 1));
    RC(pool(x, 2, 1, 0, 0, &x, 1));
    RC(cr("11", "12", 512, 3, 1));
    RC(cr("14", "15", 512, 3, 1));
    RC(pool(x, 2, 1, 0, 0, &x, 1));
    RC(cr("18", nullptr, 512, 2, 0));
    *out = x;
    return D2T_OK;
  }
  // AdaptiveAvgPool2d((None, 1)) over the height of the permuted map (recognizers/build_feat.py:50-55): [B,H,W,C] -> [B,W,C]
  int mean_h(int in, int* out) {
    const TT x = st->t[in];
    RC(new_tensor((long long)x.B * x.W, x.cols, out, x.B, 1, x.W));
    TCHK(launch_mean_h(x.p, st->t[*out].p, x.B, x.H, x.W, x.cols, s));
    Node n;
    n.kind = N_MEANH; n.in = in; n.out = *out;
    st->nodes.push_back(n);
    return D2T_OK;
  }
  // BidirectionalLSTM (seq_modeling/bilstm.py:14-24): nn.LSTM(bidirectional, batch_first) + Linear(2H -> H).  The input
  // projection of both directions is one GEMM on the engine's packed copies (c->lstm[i]: finalized weights).
  int bilstm(int in, int layer, const std::string& key, int B, int T, int* out) {
    const TT x = st->t[in];
    const int Hh = c->cfg.bilstm_hidden;
    const BiLstmW& L = c->lstm[layer];
    if (!c->finalized || !L.wih_cat || x.cols != L.in) return fail(c, D2T_ESTATE, "BiLSTM weights are not finalized");
    Node n;
    n.kind = N_BILSTM; n.in = in; n.wkey = key + "rnn."; n.nb = B; n.Lq = T; n.K = L.in; n.woff = layer;
    float* gates;
    RC(alloc(&gates, (size_t)B * T * 8 * Hh));
    RC(gemm_nt(x.p, L.wih_cat, L.bias_cat, nullptr, gates, (long long)B * T, 8 * Hh, L.in, ACT_NONE));
    RC(alloc(&n.aux[0], (size_t)B * T * 8 * Hh));  // gates after their nonlinearities
    RC(alloc(&n.aux[1], (size_t)B * T * 2 * Hh));  // cell states
    int rec;
    RC(new_tensor((long long)B * T, 2 * Hh, &rec));
    TCHK(launch_bilstm_train_fwd(gates, L.whh_t, st->t[rec].p, n.aux[0], n.aux[1], B, T, Hh, s));
    n.out = rec;
    st->nodes.push_back(n);
    return linear(rec, key + "linear", Hh, 2 * Hh, 0, ACT_NONE, -1, out);
  }
  int bwd_bilstm(const Node& n) {
    const int B = n.nb, T = n.Lq, Hh = c->cfg.bilstm_hidden, in = n.K;
    const long long R = (long long)B * T;
    const TT& x = st->t[n.in];
    const TT& rec = st->t[n.out];
    const float *whh_f, *whh_r;
    RC(raw(n.wkey + "weight_hh_l0", &whh_f, (size_t)4 * Hh * Hh));
    RC(raw(n.wkey + "weight_hh_l0_reverse", &whh_r, (size_t)4 * Hh * Hh));
    float *dgates, *hpf, *hpr;
    RC(alloc(&dgates, (size_t)R * 8 * Hh));
    TCHK(launch_bilstm_train_bwd(rec.grad, n.aux[0], n.aux[1], whh_f, whh_r, dgates, B, T, Hh, s));
    RC(alloc(&hpf, (size_t)R * Hh));
    RC(alloc(&hpr, (size_t)R * Hh));
    TCHK(launch_bilstm_hprev(rec.p, hpf, hpr, B, T, Hh, s));
    float *bsum, *g;
    RC(alloc(&bsum, (size_t)8 * Hh));
    RC(colsum(dgates, R, 8 * Hh, bsum));
    const char* sfx[2] = {"", "_reverse"};
    for (int d = 0; d < 2; ++d) {
      const float* dg = dgates + (size_t)d * 4 * Hh;
      RC(grad_buf(n.wkey + "weight_ih_l0" + sfx[d], &g));
      RC(wgrad(dg, 8 * Hh, x.p, in, R, 4 * Hh, in, 1, nullptr, nullptr, nullptr, g, 0));
      RC(grad_buf(n.wkey + "weight_hh_l0" + sfx[d], &g));
      RC(wgrad(dg, 8 * Hh, d == 0 ? hpf : hpr, Hh, R, 4 * Hh, Hh, 1, nullptr, nullptr, nullptr, g, 0));
      RC(grad_buf(n.wkey + "bias_ih_l0" + sfx[d], &g));
      TCHK(launch_copy(bsum + (size_t)d * 4 * Hh, g, (size_t)4 * Hh, s));
      RC(grad_buf(n.wkey + "bias_hh_l0" + sfx[d], &g));
      TCHK(launch_copy(bsum + (size_t)d * 4 * Hh, g, (size_t)4 * Hh, s));
    }
    // dx = dgates @ [W_ih ; W_ih_reverse]  ([R][8H] x [8H][in]): the packed matrix transposed is the "weight" of an NT GEMM
    float *wt, *dx;
    RC(alloc(&wt, (size_t)8 * Hh * in));
    TCHK(launch_transpose(c->lstm[n.woff].wih_cat, wt, 8 * Hh, in, s));
    RC(alloc(&dx, (size_t)R * in));
    RC(gemm_nt(dgates, wt, nullptr, nullptr, dx, R, in, 8 * Hh, ACT_NONE));
    return add_grad(n.in, dx);
  }

  // out = act(in @ W[woff:woff+N]^T + b[woff:woff+N]) + res;  `into`: write the result into caller memory
  int linear(int in, const std::string& key, int N, int K, int woff, int act, int res, int* out, float* into = nullptr,
             bool bias = true) {
    const TT x = st->t[in];
    if (x.cols != K) return fail(c, D2T_EINVAL, "linear '%s': input has %d columns, expected %d", key.c_str(), x.cols, K);
    const float *w, *b = nullptr;
    RC(raw(key + (key.find("in_proj") != std::string::npos ? "_weight" : ".weight"), &w));
    if (bias) RC(raw(key + (key.find("in_proj") != std::string::npos ? "_bias" : ".bias"), &b));
    RC(new_tensor(x.rows, N, out, 0, 0, 0, into));
    RC(gemm_nt(x.p, w + (size_t)woff * K, b ? b + woff : nullptr, res >= 0 ? st->t[res].p : nullptr, st->t[*out].p, x.rows, N, K, act));
    Node n;
    n.kind = N_LINEAR; n.in = in; n.in2 = res; n.out = *out; n.wkey = key; n.N = N; n.K = K; n.woff = woff; n.relu = act == ACT_RELU;
    n.nobias = !bias;
    st->nodes.push_back(n);
    return D2T_OK;
  }
  int layernorm(int in, const std::string& key, float eps, int* out, float* into = nullptr) {
    const TT x = st->t[in];
    const float *g, *b;
    RC(raw(key + ".weight", &g, x.cols));
    RC(raw(key + ".bias", &b, x.cols));
    Node n;
    n.kind = N_LN; n.in = in; n.gkey = key; n.eps = eps;
    RC(alloc(&n.mean, x.rows));
    RC(alloc(&n.rstd, x.rows));
    RC(new_tensor(x.rows, x.cols, out, 0, 0, 0, into));
    TCHK(launch_ln_train(x.p, g, b, st->t[*out].p, n.mean, n.rstd, (int)x.rows, x.cols, eps, s));
    n.out = *out;
    st->nodes.push_back(n);
    return D2T_OK;
  }
  int attention(int qt, int qoff, int kvt, int koff, int voff, int nb, int Lq, int Lk, int heads, int hd, int causal,
                const int64_t* keytok, int* out, bool attn_dropout = false) {
    Node n;
    n.kind = N_ATTN; n.in = qt; n.in2 = kvt; n.qoff = qoff; n.koff = koff; n.voff = voff; n.nb = nb; n.Lq = Lq; n.Lk = Lk;
    n.heads = heads; n.hd = hd; n.causal = causal; n.keytok = keytok;
    RC(alloc(&n.probs, (size_t)nb * heads * Lq * Lk));
    if (attn_dropout) {  // nn.MultiheadAttention(dropout=p): on the softmax probabilities
      RC(new_mask((size_t)nb * heads * Lq * Lk, &n.mask));
      n.mscale = 1.f / (1.f - st->drop_p);
    }
    RC(new_tensor((long long)nb * Lq, heads * hd, out));
    AttnTrainP p{};
    p.dropmask = n.mask; p.dropscale = n.mscale;
    p.q = st->t[qt].p + qoff; p.k = st->t[kvt].p + koff; p.v = st->t[kvt].p + voff; p.o = st->t[*out].p; p.probs = n.probs;
    p.keytok = keytok; p.B = nb; p.heads = heads; p.hd = hd; p.Lq = Lq; p.Lk = Lk;
    p.ldq = st->t[qt].cols; p.ldk = p.ldv = st->t[kvt].cols; p.ldo = heads * hd; p.causal = causal; p.pad_id = 0;
    TCHK(launch_attn_train_fwd(p, s));
    n.out = *out;
    st->nodes.push_back(n);
    return D2T_OK;
  }
  // a fresh keep mask of n elements (nullptr when dropout is off)
  int new_mask(size_t n, const uint8_t** m, float p = -1.f) {
    *m = nullptr;
    if (p < 0.f) p = st->drop_p;
    if (p <= 0.f) return D2T_OK;
    float* raw;
    RC(alloc(&raw, n / 4 + 2));
    uint8_t* mk = reinterpret_cast<uint8_t*>(raw);
    TCHK(launch_dropout_mask(mk, n, p, st->drop_seed, (st->drop_calls << 16) | st->drop_site, s));
    ++st->drop_site;
    st->masks.push_back({mk, n});
    *m = mk;
    return D2T_OK;
  }
  // nn.Dropout(p): out = in * mask / (1 - p); identity (no node) when p == 0
  int dropout(int in, int* out, float p = -1.f) {  // p < 0: the decoder's configured rate (d2t_train_set_dropout)
    if (p < 0.f) p = st->drop_p;
    if (p <= 0.f) { *out = in; return D2T_OK; }
    const TT x = st->t[in];
    Node n;
    n.kind = N_DROPOUT; n.in = in; n.mscale = 1.f / (1.f - p);
    RC(new_mask((size_t)x.rows * x.cols, &n.mask, p));
    RC(new_tensor(x.rows, x.cols, out));
    TCHK(launch_apply_mask(x.p, n.mask, n.mscale, st->t[*out].p, (size_t)x.rows * x.cols, s));
    n.out = *out;
    st->nodes.push_back(n);
    return D2T_OK;
  }
  int add(int a, int b, int* out) {
    const TT x = st->t[a];
    RC(new_tensor(x.rows, x.cols, out));
    TCHK(launch_ew(x.p, st->t[b].p, st->t[*out].p, (size_t)x.rows * x.cols, EW_ADD, s));
    Node n;
    n.kind = N_ADD; n.in = a; n.in2 = b; n.out = *out;
    st->nodes.push_back(n);
    return D2T_OK;
  }
  int gelu(int in, int* out) {
    const TT x = st->t[in];
    RC(new_tensor(x.rows, x.cols, out));
    TCHK(launch_ew(x.p, nullptr, st->t[*out].p, (size_t)x.rows * x.cols, EW_GELU, s));
    Node n;
    n.kind = N_GELU; n.in = in; n.out = *out;
    st->nodes.push_back(n);
    return D2T_OK;
  }

  int relu(int in, int* out) {
    const TT x = st->t[in];
    RC(new_tensor(x.rows, x.cols, out, x.B, x.H, x.W));
    TCHK(launch_ew(x.p, nullptr, st->t[*out].p, (size_t)x.rows * x.cols, EW_RELU, s));
    Node n;
    n.kind = N_RELU; n.in = in; n.out = *out; n.relu = true;
    st->nodes.push_back(n);
    return D2T_OK;
  }
  // GlobalContext.forward (addon_module/visual_attention.py:147-165, use_attn + fuse_add): a 1x1-conv attention map over the
  // H*W positions, softmax, the attention-pooled channel vector, ConvMLP (fc1 -> LayerNorm2d -> ReLU -> fc2) on it, added to
  // every position.  Three tape nodes of its own (pooling, ReLU, broadcast add) around the ordinary linear / LayerNorm ones.
  int global_context(int in, const std::string& key, int* out) {
    const TT x = st->t[in];
    const int B = x.B, HW = x.H * x.W, C = x.cols;
    const float *wg, *bg;
    RC(raw(key + ".global_cxt.weight", &wg, C));
    RC(raw(key + ".global_cxt.bias", &bg, 1));
    Node n;
    n.kind = N_GCPOOL; n.in = in; n.wkey = key + ".global_cxt";
    RC(alloc(&n.z, (size_t)B * HW));  // the attention logits, kept for the backward pass
    TCHK(launch_gc_logits(x.p, wg, bg, n.z, (long long)B * HW, C, s));
    int ctx;
    RC(new_tensor(B, C, &ctx));
    TCHK(launch_gc_pool(x.p, n.z, st->t[ctx].p, B, HW, C, s));
    n.out = ctx;
    st->nodes.push_back(n);
    int h, y;
    const std::string m = key + ".bottleneck_add.";
    RC(linear(ctx, m + "fc1", C, C, 0, ACT_NONE, -1, &h));
    RC(layernorm(h, m + "norm", 1e-5f, &h));
    RC(relu(h, &h));
    RC(dropout(h, &h, 0.25f));  // ConvMLP's nn.Dropout(drop=0.25), hard-wired (visual_attention.py:86,94,100)
    RC(linear(h, m + "fc2", C, C, 0, ACT_NONE, -1, &y));
    RC(new_tensor(x.rows, C, out, x.B, x.H, x.W));
    TCHK(launch_gc_bcast_add(x.p, st->t[y].p, st->t[*out].p, B, HW, C, s));
    Node a;
    a.kind = N_BCAST; a.in = in; a.in2 = y; a.out = *out;
    st->nodes.push_back(a);
    return D2T_OK;
  }

  // out[b] = in[b] + table  (PositionalEncoding2D of the ResNet+None encoder, recognizers/build_seq.py:69-76)
  int add_const(int in, const float* table, int* out) {
    const TT x = st->t[in];
    RC(new_tensor(x.rows, x.cols, out, x.B, x.H, x.W));
    const int per = (int)(x.rows / x.B) * x.cols;
    for (int b = 0; b < x.B; ++b)
      TCHK(launch_add_rows(x.p + (size_t)b * per, table, st->t[*out].p + (size_t)b * per, per, s));
    Node n;
    n.kind = N_ADDCONST; n.in = in; n.out = *out;
    st->nodes.push_back(n);
    return D2T_OK;
  }

  // ViTEncoderV3 on top of the backbone (vit_encoder.py:249-268, patchembed.py:115-141)
  int vit(int feat, const std::string& p, int* out, float* into) {
    const d2t_config& g = c->cfg;
    const TT f = st->t[feat];
    const int D = g.vit_dim, gh = (f.H + g.patch_h - 1) / g.patch_h, gw = (f.W + g.patch_w - 1) / g.patch_w, n = gh * gw;
    // (memories beyond about 600 tokens: the training attention kernels read K / V from global memory, train_kernels.hip GKV)
    if (n + 1 > 4096) return fail(c, D2T_EINVAL, "training step: memory length %d > 4096 unsupported", n + 1);
    int patch;
    // zero padding right / bottom = out-of-image taps of the strided convolution
    RC(conv_bn(feat, p + "patch_embed.proj", "", D, g.patch_h, g.patch_w, g.patch_h, g.patch_w, 0, 0, false, -1, &patch, gh, gw));
    // tokens: [cls + pos[0] | patch + pos[1:]]
    const float *cls, *pos;
    RC(raw(p + "cls_token", &cls, D));
    const RawW* pr;
    RC(need(c, p + "pos_embed", &pr));
    pos = pr->p;
    // ViTEncoder (learned table, vit_encoder.py:58-95): a crop whose patch grid is not max_dimension's reads the table through a
    // bicubic resize, rebuilt every step (the table is being trained)
    int GH = 0, GW = 0;
    if (d2t_encoder_shape(c, g.max_h, g.max_w, nullptr, nullptr, &GH, &GW, nullptr, nullptr) || (long long)pr->numel != (long long)(GH * GW + 1) * D)
      return fail(c, D2T_EINVAL, "pos_embed does not match max_dimension's patch grid");
    const PosGrid pg = pos_grid(g, GH, GW, gh, gw);
    if (pg.interp) {
      float* tbl;
      RC(alloc(&tbl, (size_t)(n + 1) * D));
      TCHK(launch_copy(pos, tbl, (size_t)D, s));
      TCHK(launch_bicubic_table(pos + D, tbl + D, GH, GW, gh, gw, D, pg.sh, pg.sw, s));
      pos = tbl;
    } else if ((long long)pr->numel < (long long)(n + 1) * D) {
      return fail(c, D2T_EINVAL, "pos_embed too small for %d tokens", n + 1);
    }
    int x;
    RC(new_tensor((long long)f.B * (n + 1), D, &x));
    TCHK(launch_token_rows(st->t[patch].p, st->t[x].p, f.B, n, 1, D, 1, s));          // scatter, cls rows zero
    TCHK(launch_fill_cls(cls, st->t[x].p, f.B, (long long)(n + 1) * D, D, s));
    {  // + pos_embed[:, :n+1] (flat prefix slice, vit_encoder.py:260), broadcast over the batch
      for (int b = 0; b < f.B; ++b)
        TCHK(launch_add_rows(st->t[x].p + (size_t)b * (n + 1) * D, pos, st->t[x].p + (size_t)b * (n + 1) * D, (n + 1) * D, s));
    }
    Node tn;
    tn.kind = N_TOKENS; tn.in = patch; tn.out = x; tn.ntok = n; tn.wkey = p + "cls_token";
    if (g.vit_pos != D2T_VIT_POS_SINCOS_PREFIX) {  // learned table: it receives a gradient (ViTEncoderV3's is frozen, :235-237)
      tn.bkey = p + "pos_embed";
      tn.KH = GH; tn.KW = GW; tn.SH = gh; tn.SW = gw;
    }
    st->nodes.push_back(tn);
    const int rows_b = n + 1;
    for (int i = 0; i < g.vit_depth; ++i) {
      const std::string bp = p + "blocks." + std::to_string(i) + ".";
      int h, qkv, a, x1, h2, u, ge, x2;
      RC(layernorm(x, bp + "norm1", 1e-6f, &h));
      RC(linear(h, bp + "attn.qkv", 3 * D, D, 0, ACT_NONE, -1, &qkv));
      RC(attention(qkv, 0, qkv, D, 2 * D, f.B, rows_b, rows_b, g.vit_heads, D / g.vit_heads, 0, nullptr, &a));
      RC(linear(a, bp + "attn.proj", D, D, 0, ACT_NONE, x, &x1));
      RC(layernorm(x1, bp + "norm2", 1e-6f, &h2));
      RC(linear(h2, bp + "mlp.fc1", 4 * D, D, 0, ACT_NONE, -1, &u));
      RC(gelu(u, &ge));
      RC(linear(ge, bp + "mlp.fc2", D, 4 * D, 0, ACT_NONE, x1, &x2));
      x = x2;
    }
    RC(layernorm(x, p + "norm", 1e-6f, out, into));
    return D2T_OK;
  }

  // TransformerPrediction teacher-forced pass (tfm.py:103-118), post-norm nn.TransformerDecoderLayer
  int decoder(int mem, const int64_t* tgt, int B, int L, const std::string& p, float* logits, int* out) {
    const d2t_config& g = c->cfg;
    const int D = g.dec_dim, heads = g.dec_heads, hd = D / heads, T = (int)(st->t[mem].rows / B);
    const float *E, *pe;
    RC(raw(p + "word_embed.weight", &E, (size_t)g.vocab * D));
    RC(raw(p + "pos_enc.pe", &pe));
    int x;
    RC(new_tensor((long long)B * L, D, &x));
    TCHK(launch_embed_train(E, pe, tgt, st->t[x].p, B * L, L, D, sqrtf((float)D), s));
    Node en;
    en.kind = N_EMBED; en.out = x; en.wkey = p + "word_embed.weight";
    st->nodes.push_back(en);
    const bool dp = st->drop_p > 0.f;
    for (int l = 0; l < g.dec_layers; ++l) {
      const std::string lp = p + "model.layers." + std::to_string(l) + ".";
      int qkv, a, y1, x1, q2, kv, a2, y2, x2, f, y3, x3;
      RC(linear(x, lp + "self_attn.in_proj", 3 * D, D, 0, ACT_NONE, -1, &qkv));
      RC(attention(qkv, 0, qkv, D, 2 * D, B, L, L, heads, hd, 1, tgt, &a, dp));
      if (!dp) {
        RC(linear(a, lp + "self_attn.out_proj", D, D, 0, ACT_NONE, x, &y1));
      } else {  // x + dropout1(self_attn(x)): the residual add can no longer ride in the GEMM epilogue
        int z, d;
        RC(linear(a, lp + "self_attn.out_proj", D, D, 0, ACT_NONE, -1, &z));
        RC(dropout(z, &d));
        RC(add(x, d, &y1));
      }
      RC(layernorm(y1, lp + "norm1", 1e-5f, &x1));
      RC(linear(x1, lp + "multihead_attn.in_proj", D, D, 0, ACT_NONE, -1, &q2));
      RC(linear(mem, lp + "multihead_attn.in_proj", 2 * D, D, D, ACT_NONE, -1, &kv));
      RC(attention(q2, 0, kv, 0, D, B, L, T, heads, hd, 0, nullptr, &a2, dp));
      if (!dp) {
        RC(linear(a2, lp + "multihead_attn.out_proj", D, D, 0, ACT_NONE, x1, &y2));
      } else {
        int z, d;
        RC(linear(a2, lp + "multihead_attn.out_proj", D, D, 0, ACT_NONE, -1, &z));
        RC(dropout(z, &d));
        RC(add(x1, d, &y2));
      }
      RC(layernorm(y2, lp + "norm2", 1e-5f, &x2));
      RC(linear(x2, lp + "linear1", g.dec_ff, D, 0, ACT_RELU, -1, &f));
      if (!dp) {
        RC(linear(f, lp + "linear2", D, g.dec_ff, 0, ACT_NONE, x2, &y3));
      } else {  // x + dropout3(linear2(dropout(relu(linear1(x)))))
        int fd, z, d;
        RC(dropout(f, &fd));
        RC(linear(fd, lp + "linear2", D, g.dec_ff, 0, ACT_NONE, -1, &z));
        RC(dropout(z, &d));
        RC(add(x2, d, &y3));
      }
      RC(layernorm(y3, lp + "norm3", 1e-5f, &x3));
      x = x3;
    }
    RC(linear(x, p + "proj", g.vocab, D, 0, ACT_NONE, -1, out, logits));
    return D2T_OK;
  }

  // Attention / AttentionV2.forward_greedy under is_train with teacher_forcing = 1 (seq2seq.py:224-331): the whole loop
  // is one launch of the greedy kernel fed with the label tokens; its per-step state is kept for bwd_lstm.
  int lstm_decoder(int mem, const int64_t* tgt, int B, int S, float* logits, int* out) {
    const d2t_config& g = c->cfg;
    const int Hh = g.attn_hidden, V = g.vocab, T = (int)(st->t[mem].rows / B);
    const int key_off = g.attn_keys == D2T_ATTN_KEYS_NOCLS_INIT_CLS ? 1 : 0, Tk = T - key_off;
    if (!c->finalized) return fail(c, D2T_ESTATE, "the Attn training step needs finalized weights");
    // key projection: key_proj of the location-aware cell, i2h (no bias) of the Bahdanau cell (attention1D.py:77-82)
    const bool bahdanau = g.attn_cell == D2T_ATTN_CELL_BAHDANAU;
    int kp;
    RC(linear(mem, std::string("predicter.Prediction.attention_cell.attn.") + (bahdanau ? "i2h" : "key_proj"), Hh, Hh, 0, ACT_NONE, -1,
              &kp, nullptr, !bahdanau));
    Node n;
    n.kind = N_LSTM; n.in = mem; n.in2 = kp; n.nb = B; n.Lq = S; n.Lk = T; n.koff = key_off;
    const size_t BS = (size_t)B * S;
    RC(alloc(&n.aux[0], BS * Hh));      // h_prev
    RC(alloc(&n.aux[1], BS * Hh));      // c_prev
    RC(alloc(&n.aux[2], BS * Hh));      // h_after
    RC(alloc(&n.aux[3], BS * Hh));      // c_after
    RC(alloc(&n.aux[4], BS * 4 * Hh));  // gates
    RC(alloc(&n.aux[5], BS * Tk));      // alpha
    RC(alloc(&n.aux[6], BS * Hh));      // hq
    RC(alloc(&n.aux[7], BS * 2 * Hh));  // [ctx | emb]
    float *dummy, *tokbuf;
    RC(alloc(&dummy, BS * 2 + B + 16));  // tokens (int64) and end_step scratch
    RC(alloc(&tokbuf, BS * 2 + 2));      // input token per (row, step), int64
    n.keytok = reinterpret_cast<const int64_t*>(tokbuf);
    RC(new_tensor((long long)BS, V, out, 0, 0, 0, logits));
    if (st->drop_p > 0.f) {  // nn.Dropout(droprate) on the generator output of every step (seq2seq.py:298)
      RC(new_mask(BS * V, &n.mask));
      n.mscale = 1.f / (1.f - st->drop_p);
    }
    const uint8_t* d_flags = nullptr;
    if (!st->teacher_flags.empty()) {
      if ((int)st->teacher_flags.size() != S) return fail(c, D2T_EINVAL, "teacher flags: %zu entries for %d steps", st->teacher_flags.size(), S);
      float* fb;
      RC(alloc(&fb, S / 4 + 2));
      TCHK(hipMemcpyAsync(fb, st->teacher_flags.data(), S, hipMemcpyHostToDevice, s));
      TCHK(hipStreamSynchronize(s));  // the host vector may change before an asynchronous copy would run
      d_flags = reinterpret_cast<const uint8_t*>(fb);
    }
    AttnDecP p{};
    p.mem = st->t[mem].p; p.T = T; p.D = Hh; p.key_off = key_off;
    p.init_mode = !g.attn_enc_init ? 0 : (g.attn_keys == D2T_ATTN_KEYS_ALL_INIT_MEAN ? 1 : 2);
    n.causal = p.init_mode;  // (the field is free in this node kind) how the initial state was formed, for the backward pass
    p.kp = st->t[kp].p; p.wq_t = c->attn.wq_t; p.bq = c->attn.bq; p.wloc = c->attn.wloc; p.bloc = c->attn.bloc;
    p.taps = c->attn.taps; p.wscore = c->attn.wscore; p.bscore = c->attn.bscore;
    p.wx_t = c->attn.wx_t; p.bx = c->attn.bx; p.wg_t = c->attn.wg_t; p.bg = c->attn.bg;
    p.wih_t = c->attn.wih_t; p.bih = c->attn.bih; p.wic_t = c->attn.wic_t; p.bic = c->attn.bic;
    p.emb = c->attn.emb; p.tokgate = c->attn.tokgate; p.probs = logits; p.tokens = reinterpret_cast<int64_t*>(dummy);
    p.end_step = reinterpret_cast<int*>(dummy + BS * 2);
    p.B = B; p.S = S; p.V = V; p.H = Hh; p.E = Hh; p.coverage = g.attn_coverage; p.end_token = 1;
    p.teacher = tgt; p.use_teacher = d_flags; p.out_dropmask = n.mask; p.out_dropscale = n.mscale;
    p.sv_tok = reinterpret_cast<int64_t*>(tokbuf);
    p.sv_hprev = n.aux[0]; p.sv_cprev = n.aux[1]; p.sv_hafter = n.aux[2]; p.sv_cafter = n.aux[3];
    p.sv_gates = n.aux[4]; p.sv_alpha = n.aux[5]; p.sv_hq = n.aux[6]; p.sv_x = n.aux[7];
    TCHK(launch_attn_decode(p, s));
    n.out = *out;
    st->nodes.push_back(n);
    return D2T_OK;
  }
  int zeros(float** p, size_t floats) {
    RC(alloc(p, floats));
    TCHK(hipMemsetAsync(*p, 0, floats * 4, s));
    return D2T_OK;
  }
  int bwd_lstm(const Node& n) {
    const d2t_config& g = c->cfg;
    const int B = n.nb, S = n.Lq, T = n.Lk, Hh = g.attn_hidden, V = g.vocab, key_off = n.koff, Tk = T - key_off;
    const int taps = c->attn.taps, kd = g.attn_kernel_dim;
    const long long BS = (long long)B * S;
    const std::string pp = "predicter.Prediction.", ac = pp + "attention_cell.";
    const TT& mem = st->t[n.in];
    const TT& kp = st->t[n.in2];
    const float* dl = st->t[n.out].grad;
    if (n.mask) {  // output dropout: everything upstream sees the masked, rescaled gradient
      float* dlm;
      RC(alloc(&dlm, (size_t)BS * V));
      TCHK(launch_apply_mask(dl, n.mask, n.mscale, dlm, (size_t)BS * V, s));
      dl = dlm;
    }
    // Bahdanau cell: h2h is the query projection, no location layers, no score bias (the kernel's location filter is the
    // zero filter the forward pass used).  One-hot targets: rnn.weight_ih is [4H][H + V]; the kernel gets its context
    // columns beside a zero "embedding" block, and the token columns' gradients are gathered per token below.
    const bool bahdanau = g.attn_cell == D2T_ATTN_CELL_BAHDANAU, onehot = g.attn_onehot != 0;
    const std::string qk = ac + (bahdanau ? "attn.h2h" : "attn.query_proj");
    const float *wih, *whh, *wq, *cw = nullptr, *cb = nullptr, *pw = nullptr;
    RC(raw(ac + "rnn.weight_ih", &wih));
    RC(raw(ac + "rnn.weight_hh", &whh));
    RC(raw(qk + ".weight", &wq));
    if (!bahdanau) {
      RC(raw(ac + "attn.loc_conv.weight", &cw));
      RC(raw(ac + "attn.loc_conv.bias", &cb));
      RC(raw(ac + "attn.loc_proj.weight", &pw));
    }
    if (onehot) {
      float* wc;
      RC(zeros(&wc, (size_t)4 * Hh * 2 * Hh));
      TCHK(launch_copy2d(wih, Hh + V, wc, 2 * Hh, (size_t)4 * Hh, Hh, s));
      wih = wc;
    }
    AttnTrainBwdP p{};
    p.dlogits = dl; p.mem = mem.p; p.T = T; p.D = Hh; p.key_off = key_off; p.kp = kp.p;
    p.wg_t = c->attn.wg_t; p.wih_raw = wih; p.whh_raw = whh; p.wq_raw = wq; p.wloc = c->attn.wloc; p.bloc = c->attn.bloc;
    p.wscore = c->attn.wscore; p.taps = taps;
    p.sv_cprev = n.aux[1]; p.sv_cafter = n.aux[3]; p.sv_gates = n.aux[4]; p.sv_alpha = n.aux[5]; p.sv_hq = n.aux[6];
    float *dmem, *dkp, *dgates, *dhq, *demb, *dh0, *dc0, *dwloc, *dbloc, *dwscore, *dbscore;
    RC(zeros(&dmem, (size_t)mem.rows * mem.cols));
    RC(zeros(&dkp, (size_t)kp.rows * kp.cols));
    RC(alloc(&dgates, (size_t)BS * 4 * Hh));
    RC(alloc(&dhq, (size_t)BS * Hh));
    RC(alloc(&demb, (size_t)BS * Hh));
    RC(alloc(&dh0, (size_t)B * Hh));
    RC(alloc(&dc0, (size_t)B * Hh));
    RC(alloc(&dwloc, (size_t)B * Hh * taps));
    RC(alloc(&dbloc, (size_t)B * Hh));
    RC(alloc(&dwscore, (size_t)B * Hh));
    RC(alloc(&dbscore, (size_t)B + 16));
    p.dmem = dmem; p.dkp = dkp; p.dgates = dgates; p.dhq = dhq; p.dh0 = dh0; p.dc0 = dc0;
    p.dwloc = dwloc; p.dbloc = dbloc; p.dwscore = dwscore; p.dbscore = dbscore;
    p.B = B; p.S = S; p.V = V; p.H = Hh; p.E = Hh; p.coverage = g.attn_coverage;
    // The two products that do not take part in the recurrence leave the sequential kernel (each cost its weight matrix
    // per row and step from L2): dlogits . generator.weight for all (row, step) as one GEMM before it ...
    static const bool hoist = !(D2T_PROBE_ENV_STR("D2T_LSTM_BWD_HOIST") && D2T_PROBE_ENV("D2T_LSTM_BWD_HOIST") == 0);
    if (hoist) {
      const int Vp = (V + 31) / 32 * 32;
      float *gpad, *wgp, *dhl;
      RC(alloc(&gpad, (size_t)BS * Vp));
      RC(alloc(&wgp, (size_t)Hh * Vp));
      RC(alloc(&dhl, (size_t)BS * Hh));
      TCHK(launch_pad_cols(dl, gpad, (size_t)BS, V, Vp, s));
      TCHK(launch_pad_cols(c->attn.wg_t, wgp, (size_t)Hh, V, Vp, s));
      RC(gemm_nt(gpad, wgp, nullptr, nullptr, dhl, BS, Hh, Vp, ACT_NONE));
      p.dhl = dhl;
      p.demb = nullptr;
    } else {
      p.demb = demb;
    }
    TCHK(launch_attn_train_lstm_bwd(p, s));
    // ... and the gradient of the embedded target, dgates . W_ih[:, H:], as one GEMM after it (not needed with one-hot targets)
    if (hoist && !onehot) RC(gemm_nt(dgates, c->attn.wx_t + (size_t)Hh * 4 * Hh, nullptr, nullptr, demb, BS, Hh, 4 * Hh, ACT_NONE));
    // weight gradients = sums over (row, step) of outer products -> TN GEMMs on the saved factors
    float *gW, *gB, *gB2;
    RC(grad_buf(ac + "generator.weight", &gW));
    RC(grad_buf(ac + "generator.bias", &gB));
    RC(wgrad(dl, V, n.aux[2], Hh, BS, V, Hh, 1, nullptr, nullptr, nullptr, gW, 0));
    RC(colsum(dl, BS, V, gB));
    RC(grad_buf(ac + "rnn.weight_ih", &gW));
    if (!onehot) {
      RC(wgrad(dgates, 4 * Hh, n.aux[7], 2 * Hh, BS, 4 * Hh, 2 * Hh, 1, nullptr, nullptr, nullptr, gW, 0));
    } else {  // columns [0, H): the context part; column H + v: the sum of the gate gradients of the steps fed token v
      float *gc, *gt, *gtt;
      RC(alloc(&gc, (size_t)4 * Hh * Hh));
      RC(alloc(&gt, (size_t)V * 4 * Hh));
      RC(alloc(&gtt, (size_t)4 * Hh * V));
      RC(wgrad(dgates, 4 * Hh, n.aux[7], 2 * Hh, BS, 4 * Hh, Hh, 1, nullptr, nullptr, nullptr, gc, 0));
      TCHK(launch_copy2d(gc, Hh, gW, Hh + V, (size_t)4 * Hh, Hh, s));
      TCHK(launch_embed_bwd(dgates, n.keytok, gt, (int)BS, V, 4 * Hh, 1.f, -1, s));
      TCHK(launch_transpose(gt, gtt, V, 4 * Hh, s));
      TCHK(launch_copy2d(gtt, V, gW + Hh, Hh + V, (size_t)4 * Hh, V, s));
    }
    RC(grad_buf(ac + "rnn.weight_hh", &gW));
    RC(wgrad(dgates, 4 * Hh, n.aux[0], Hh, BS, 4 * Hh, Hh, 1, nullptr, nullptr, nullptr, gW, 0));
    RC(grad_buf(ac + "rnn.bias_ih", &gB));
    RC(grad_buf(ac + "rnn.bias_hh", &gB2));
    RC(colsum(dgates, BS, 4 * Hh, gB));
    TCHK(launch_copy(gB, gB2, (size_t)4 * Hh, s));
    RC(grad_buf(qk + ".weight", &gW));
    RC(grad_buf(qk + ".bias", &gB));
    RC(wgrad(dhq, Hh, n.aux[0], Hh, BS, Hh, Hh, 1, nullptr, nullptr, nullptr, gW, 0));
    RC(colsum(dhq, BS, Hh, gB));
    RC(grad_buf(ac + "attn.score.weight", &gW));
    TCHK(launch_sum_over_rows(dwscore, gW, B, Hh, s));
    if (!bahdanau) {
      RC(grad_buf(ac + "attn.score.bias", &gB));
      TCHK(launch_sum_over_rows(dbscore, gB, B, 1, s));
      float *gcw, *gcb, *gpw, *gpb;
      RC(grad_buf(ac + "attn.loc_conv.weight", &gcw));
      RC(grad_buf(ac + "attn.loc_conv.bias", &gcb));
      RC(grad_buf(ac + "attn.loc_proj.weight", &gpw));
      RC(grad_buf(ac + "attn.loc_proj.bias", &gpb));
      TCHK(launch_loc_unfold_bwd(dwloc, dbloc, B, cw, cb, pw, Hh, kd, taps, gcw, gcb, gpw, gpb, s));
    }
    if (!onehot) {
      RC(grad_buf(pp + "embedding.weight", &gW));
      TCHK(launch_embed_bwd(demb, n.keytok, gW, (int)BS, V, Hh, 1.f, 0, s));  // padding_idx = [GO] = 0 (seq2seq.py:33-35)
    }
    if (g.attn_enc_init) {  // h0 / c0 = proj_init_{h,c}(memory[:, 0]) or, on a BiLSTM encoder, of the mean over the tokens
      const bool mean = n.causal == 1;
      const float* initv = mem.p;  // row b = memory[b][0] with a row stride of T * Hh
      int ldi = T * Hh;
      if (mean) {
        float *sums, *mv;
        RC(alloc(&sums, (size_t)B * Hh));
        RC(alloc(&mv, (size_t)B * Hh));
        RC(ensure_part((size_t)B * GC_CHUNKS * Hh));
        TCHK(launch_gc_wpool(mem.p, nullptr, st->part, sums, B, T, Hh, s));
        TCHK(launch_scale(sums, mv, (size_t)B * Hh, 1.f / T, s));
        initv = mv;
        ldi = Hh;
      }
      RC(grad_buf(pp + "proj_init_h.weight", &gW));
      RC(grad_buf(pp + "proj_init_h.bias", &gB));
      RC(wgrad(dh0, Hh, initv, ldi, B, Hh, Hh, 1, nullptr, nullptr, nullptr, gW, 0));
      RC(colsum(dh0, B, Hh, gB));
      RC(grad_buf(pp + "proj_init_c.weight", &gW));
      RC(grad_buf(pp + "proj_init_c.bias", &gB));
      RC(wgrad(dc0, Hh, initv, ldi, B, Hh, Hh, 1, nullptr, nullptr, nullptr, gW, 0));
      RC(colsum(dc0, B, Hh, gB));
      float *t1, *dinit;
      RC(alloc(&t1, (size_t)B * Hh));
      RC(alloc(&dinit, (size_t)B * Hh));
      RC(gemm_nt(dh0, c->attn.wih_t, nullptr, nullptr, t1, B, Hh, Hh, ACT_NONE));
      RC(gemm_nt(dc0, c->attn.wic_t, nullptr, t1, dinit, B, Hh, Hh, ACT_NONE));
      if (mean) {  // every token carries 1 / T of the initial state's gradient
        float *dsc, *dmem2;
        RC(alloc(&dsc, (size_t)B * Hh));
        TCHK(launch_scale(dinit, dsc, (size_t)B * Hh, 1.f / T, s));
        RC(alloc(&dmem2, (size_t)mem.rows * mem.cols));
        TCHK(launch_gc_bcast_add(dmem, dsc, dmem2, B, T, Hh, s));
        dmem = dmem2;
      } else {
        for (int b = 0; b < B; ++b)
          TCHK(launch_add_rows(dmem + (size_t)b * T * Hh, dinit + (size_t)b * Hh, dmem + (size_t)b * T * Hh, Hh, s));
      }
    }
    RC(add_grad(n.in, dmem));
    RC(add_grad(n.in2, dkp));
    return D2T_OK;
  }

  // ---------------- backward ----------------
  int bwd_linear(const Node& n) {
    const TT& y = st->t[n.out];
    const TT& x = st->t[n.in];
    const float* g = y.grad;
    if (n.in2 >= 0) RC(add_grad(n.in2, y.grad));  // the residual branch takes the incoming gradient as is
    if (n.relu) {
      float* m;
      RC(alloc(&m, (size_t)y.rows * y.cols));
      TCHK(launch_ew(y.grad, y.p, m, (size_t)y.rows * y.cols, EW_RELU_BWD, s));
      g = m;
    }
    const bool inproj = n.wkey.find("in_proj") != std::string::npos;
    const std::string wk = n.wkey + (inproj ? "_weight" : ".weight"), bk = n.wkey + (inproj ? "_bias" : ".bias");
    float *dW, *db;
    RC(grad_buf(wk, &dW));
    if (!n.nobias) {
      RC(grad_buf(bk, &db));
      RC(colsum(g, y.rows, n.N, db + n.woff));
    }
    RC(wgrad(g, n.N, x.p, n.K, y.rows, n.N, n.K, 1, nullptr, nullptr, nullptr, dW + (size_t)n.woff * n.K, 0));
    // dx = g @ W  ->  NT GEMM against W^T [K][N]; the reduction dimension N must be a multiple of 32
    const float* w;
    RC(raw(wk, &w));
    w += (size_t)n.woff * n.K;
    const int Np = (n.N + 31) / 32 * 32;
    float *wt, *dx;
    RC(alloc(&wt, (size_t)n.K * Np));
    const float* gp = g;
    if (Np != n.N) {  // vocabulary projection: pad the reduction dimension with zeros
      float *wpad, *gpad;
      RC(alloc(&wpad, (size_t)Np * n.K));
      TCHK(hipMemsetAsync(wpad, 0, (size_t)Np * n.K * 4, s));
      TCHK(launch_copy(w, wpad, (size_t)n.N * n.K, s));
      TCHK(launch_transpose(wpad, wt, Np, n.K, s));
      RC(alloc(&gpad, (size_t)y.rows * Np));
      TCHK(launch_pad_cols(g, gpad, (size_t)y.rows, n.N, Np, s));
      gp = gpad;
    } else {
      TCHK(launch_transpose(w, wt, n.N, n.K, s));
    }
    RC(alloc(&dx, (size_t)x.rows * x.cols));
    RC(gemm_nt(gp, wt, nullptr, nullptr, dx, y.rows, n.K, Np, ACT_NONE));
    return add_grad(n.in, dx);
  }
  int bwd_ln(const Node& n) {
    const TT& y = st->t[n.out];
    const TT& x = st->t[n.in];
    const float* g;
    RC(raw(n.gkey + ".weight", &g));
    float *dg, *db;
    RC(grad_buf(n.gkey + ".weight", &dg));
    RC(grad_buf(n.gkey + ".bias", &db));
    const int chunks = colreduce_chunks(x.rows, x.cols);
    RC(ensure_part((size_t)chunks * 2 * x.cols));
    ColRedP p{};
    p.a = y.grad; p.z = x.p; p.mean = n.mean; p.rstd = n.rstd; p.part = st->part; p.R = x.rows; p.C = x.cols; p.mode = CR_LN_BWD;
    TCHK(launch_colreduce(p, s));
    TCHK(launch_colreduce_final(st->part, chunks, x.cols, db, dg, 0, s));
    float* dx;
    RC(alloc(&dx, (size_t)x.rows * x.cols));
    TCHK(launch_ln_bwd(y.grad, x.p, n.mean, n.rstd, g, nullptr, dx, (int)x.rows, x.cols, s));
    return add_grad(n.in, dx);
  }
  int bwd_attn(const Node& n) {
    RC(own_grad(n.in));
    RC(own_grad(n.in2));
    const TT& y = st->t[n.out];
    AttnTrainP p{};
    p.q = st->t[n.in].p + n.qoff; p.k = st->t[n.in2].p + n.koff; p.v = st->t[n.in2].p + n.voff;
    p.o = y.grad; p.probs = n.probs;
    p.dq = st->t[n.in].grad + n.qoff; p.dk = st->t[n.in2].grad + n.koff; p.dv = st->t[n.in2].grad + n.voff;
    p.B = n.nb; p.heads = n.heads; p.hd = n.hd; p.Lq = n.Lq; p.Lk = n.Lk;
    p.ldq = st->t[n.in].cols; p.ldk = p.ldv = st->t[n.in2].cols; p.ldo = n.heads * n.hd;
    p.dropmask = n.mask; p.dropscale = n.mscale;
    TCHK(launch_attn_train_bwd(p, s));
    return D2T_OK;
  }
  int bwd_conv(const Node& n) {
    const TT& y = st->t[n.out];
    const long long P = y.rows;
    const int Cout = n.N;
    const float* dz = y.grad;
    uint16_t* dz_planes = nullptr;
    float *dW;
    RC(grad_buf(n.wkey + ".weight", &dW));
    if (!n.bnkey.empty()) {
      const float* gamma;
      RC(raw(n.bnkey + ".weight", &gamma));
      float *dgam, *dbet, *s0, *s1, *dzb, *gres = nullptr;
      RC(grad_buf(n.bnkey + ".weight", &dgam));
      RC(grad_buf(n.bnkey + ".bias", &dbet));
      const int chunks = colreduce_chunks(P, Cout);
      RC(ensure_part((size_t)chunks * 2 * Cout));
      ColRedP p{};
      p.a = y.grad; p.y = n.relu ? y.p : nullptr; p.z = n.z; p.mean = n.mean; p.rstd = n.rstd; p.part = st->part;
      p.R = P; p.C = Cout; p.mode = CR_BN_BWD;
      TCHK(launch_colreduce(p, s));
      TCHK(launch_colreduce_final(st->part, chunks, Cout, dbet, dgam, 0, s));
      s0 = dbet; s1 = dgam;
      RC(alloc(&dzb, (size_t)P * Cout));
      if (n.in2 >= 0) RC(alloc(&gres, (size_t)P * Cout));
      if (!n.stem) RC(new_planes(P, Cout, &dz_planes));  // the data-gradient convolution's operand records
      TCHK(launch_bn_bwd_apply(y.grad, n.relu ? y.p : nullptr, n.z, n.mean, n.rstd, gamma, s0, s1, dzb, gres, P, Cout, s, dz_planes));
      if (n.in2 >= 0) RC(add_grad(n.in2, gres));
      dz = dzb;
    } else {
      float* db;
      RC(grad_buf(n.wkey + ".bias", &db));
      RC(colsum(dz, P, Cout, db));
    }
    if (n.stem) {
      const int chunk = 1024, nch = (int)((P + chunk - 1) / chunk);
      RC(ensure_part((size_t)nch * 9 * Cout));
      TCHK(launch_stem_wgrad(st->image, dz, st->part, st->B, st->H, st->W, Cout, chunk, nch, s));
      TCHK(launch_wgrad_reduce(st->part, dW, nch, 9, Cout, 1, 1, 0, s));
      return D2T_OK;
    }
    const TT& x = st->t[n.in];
    RC(wgrad(dz, Cout, x.p, x.cols, P, Cout, x.cols, n.KH * n.KW, &n, &x, &y, dW, 1, dz_planes, x.planes));
    // data gradient: stride-1 convolution of the (zero-dilated) dz with the flipped, transposed filter
    const float* w;
    RC(raw(n.wkey + ".weight", &w));
    const int Cin = x.cols, Kd = n.KH * n.KW * Cout;
    float *wf, *wp, *dx;
    RC(alloc(&wf, (size_t)Cin * Kd));
    RC(alloc(&wp, (size_t)Cin * Kd + Cin));
    TCHK(launch_flip_oihw(w, wf, Cout, Cin, n.KH, n.KW, s));                      // [Cin][Cout][KH][KW]
    TCHK(launch_pack_conv(wf, nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, wp, wp + (size_t)Cin * Kd, Cin, Cout, n.KH, n.KW, s));
    const float* src = dz;
    int DH = y.H, DW = y.W;
    if (n.SH != 1 || n.SW != 1 || DH != x.H + 2 * n.PH - n.KH + 1 || DW != x.W + 2 * n.PW - n.KW + 1) {
      DH = std::max(x.H + 2 * n.PH - n.KH + 1, (y.H - 1) * n.SH + 1);
      DW = std::max(x.W + 2 * n.PW - n.KW + 1, (y.W - 1) * n.SW + 1);
      float* dil;
      RC(alloc(&dil, (size_t)x.B * DH * DW * Cout));
      TCHK(launch_dilate(dz, dil, x.B, y.H, y.W, Cout, DH, DW, n.SH, n.SW, 0, 0, s));
      src = dil;
    }
    RC(alloc(&dx, (size_t)x.rows * Cin));
    ConvP p{};
    if (c->conv_bf16x3) {
      float* planes;
      RC(alloc(&planes, (size_t)Cin * Kd));  // two bf16 planes = Cin*Kd floats
      uint16_t* hi = reinterpret_cast<uint16_t*>(planes);
      TCHK(launch_split_bf16(wp, hi, hi + (size_t)Cin * Kd, (size_t)Cin * Kd, s));
      p.w_hi = hi; p.w_lo = hi + (size_t)Cin * Kd;
    }
    p.in = src; p.w = wp; p.out = dx;
    p.B = x.B; p.H = DH; p.W = DW; p.Cin = Cout; p.OH = x.H; p.OW = x.W; p.Cout = Cin;
    p.KH = n.KH; p.KW = n.KW; p.SH = p.SW = 1; p.PH = n.KH - 1 - n.PH; p.PW = n.KW - 1 - n.PW;
    p.M = (int)x.rows; p.K = Kd; p.act = ACT_NONE;
    RC(split_input(&p, src, (long long)x.B * DH * DW, Cout, src == dz ? dz_planes : nullptr));
    TCHK(d2t_internal_conv_timed(c, p, s));
    return add_grad(n.in, dx);
  }
  int bwd_pool(const Node& n) {
    const TT& x = st->t[n.in];
    float* dx;
    RC(alloc(&dx, (size_t)x.rows * x.cols));
    TCHK(launch_maxpool_bwd(x.p, st->t[n.out].grad, dx, x.B, x.H, x.W, x.cols, n.SH, n.SW, n.PH, n.PW, s, n.KW));
    return add_grad(n.in, dx);
  }
  int bwd_tokens(const Node& n) {
    const TT& y = st->t[n.out];
    const TT& x = st->t[n.in];
    const int D = y.cols, nb = (int)(y.rows / (n.ntok + 1));
    float *dcls, *dx;
    RC(grad_buf(n.wkey, &dcls));
    TCHK(launch_sum_rows_strided(y.grad, dcls, nb, n.ntok + 1, 0, D, s));
    RC(alloc(&dx, (size_t)x.rows * D));
    TCHK(launch_token_rows(y.grad, dx, nb, n.ntok, 1, D, 0, s));
    if (!n.bkey.empty()) {
      // d pos_embed: the table is broadcast over the batch, so its gradient is the sum of the token gradients over the batch --
      // written to the first ntok + 1 rows (prefix slice / table as is) or, for a resized table, pulled back through the
      // transpose of the bicubic resize; rows the crop did not read get zero
      const int GH = n.KH, GW = n.KW, gh = n.SH, gw = n.SW;
      const PosGrid pg = pos_grid(c->cfg, GH, GW, gh, gw);
      float* dpos;
      RC(grad_buf(n.bkey, &dpos));
      const size_t rows = (size_t)n.ntok + 1, all = (size_t)GH * GW + 1;
      if (!pg.interp) {
        TCHK(launch_sum_over_batch(y.grad, dpos, nb, (long long)(rows * D), (long long)(rows * D), s));
        if (all > rows) TCHK(hipMemsetAsync(dpos + rows * D, 0, (all - rows) * D * 4, s));
      } else {
        float* dt;
        RC(alloc(&dt, rows * D));
        TCHK(launch_sum_over_batch(y.grad, dt, nb, (long long)(rows * D), (long long)(rows * D), s));
        TCHK(launch_copy(dt, dpos, (size_t)D, s));
        TCHK(launch_bicubic_table_bwd(dt + D, dpos + D, GH, GW, gh, gw, D, pg.sh, pg.sw, s));
      }
    }
    return add_grad(n.in, dx);
  }
  int backward() {
    for (int i = (int)st->nodes.size() - 1; i >= 0; --i) {
      const Node& n = st->nodes[i];
      if (!st->t[n.out].grad) continue;
      switch (n.kind) {
        case N_LINEAR: RC(bwd_linear(n)); break;
        case N_LN: RC(bwd_ln(n)); break;
        case N_ATTN: RC(bwd_attn(n)); break;
        case N_CONV: RC(bwd_conv(n)); break;
        case N_POOL: RC(bwd_pool(n)); break;
        case N_TOKENS: RC(bwd_tokens(n)); break;
        case N_ADDCONST: RC(add_grad(n.in, st->t[n.out].grad)); break;
        case N_LSTM: RC(bwd_lstm(n)); break;
        case N_DROPOUT: {
          const TT& x = st->t[n.in];
          float* dx;
          RC(alloc(&dx, (size_t)x.rows * x.cols));
          TCHK(launch_apply_mask(st->t[n.out].grad, n.mask, n.mscale, dx, (size_t)x.rows * x.cols, s));
          RC(add_grad(n.in, dx));
          break;
        }
        case N_ADD: {  // both operands receive the gradient; the second gets its own copy unless it only accumulates
          float* g = st->t[n.out].grad;
          RC(add_grad(n.in, g));
          if (!st->t[n.in2].grad) {
            const TT& y = st->t[n.out];
            float* cp;
            RC(alloc(&cp, (size_t)y.rows * y.cols));
            TCHK(launch_copy(g, cp, (size_t)y.rows * y.cols, s));
            g = cp;
          }
          RC(add_grad(n.in2, g));
          break;
        }
        case N_BILSTM: RC(bwd_bilstm(n)); break;
        case N_MEANH: {
          const TT& x = st->t[n.in];
          float* dx;
          RC(alloc(&dx, (size_t)x.rows * x.cols));
          TCHK(launch_mean_h_bwd(st->t[n.out].grad, dx, x.B, x.H, x.W, x.cols, s));
          RC(add_grad(n.in, dx));
          break;
        }
        case N_RELU: {
          const TT& x = st->t[n.in];
          float* dx;
          RC(alloc(&dx, (size_t)x.rows * x.cols));
          TCHK(launch_ew(st->t[n.out].grad, st->t[n.out].p, dx, (size_t)x.rows * x.cols, EW_RELU_BWD, s));
          RC(add_grad(n.in, dx));
          break;
        }
        case N_BCAST: {  // out = x + y[b]: dx = dout; dy[b][c] = sum over the image's positions
          const TT& x = st->t[n.in];
          const TT& o = st->t[n.out];
          const int B = x.B, HW = x.H * x.W, C = x.cols;
          float *dy, *dx;
          RC(alloc(&dy, (size_t)B * C));
          RC(ensure_part((size_t)B * GC_CHUNKS * C));
          TCHK(launch_gc_wpool(o.grad, nullptr, st->part, dy, B, HW, C, s));
          RC(add_grad(n.in2, dy));
          RC(alloc(&dx, (size_t)x.rows * C));
          TCHK(hipMemcpyAsync(dx, o.grad, (size_t)x.rows * C * 4, hipMemcpyDeviceToDevice, s));
          RC(add_grad(n.in, dx));
          break;
        }
        case N_GCPOOL: {  // ctx[b] = sum_p a[b][p] x[b][p], a = softmax_p(x . wg + bg)
          const TT& x = st->t[n.in];
          const int B = x.B, HW = x.H * x.W, C = x.cols;
          const float* dctx = st->t[n.out].grad;
          const float* wg;
          RC(raw(n.wkey + ".weight", &wg, C));
          float *a, *dl, *sdl, *dx, *dwg_b, *dwg, *dbg;
          RC(alloc(&a, (size_t)B * HW));
          RC(alloc(&dl, (size_t)B * HW));
          RC(alloc(&sdl, B));
          TCHK(launch_gc_pool_bwd_weights(x.p, n.z, dctx, st->t[n.out].p, a, dl, sdl, B, HW, C, s));
          RC(alloc(&dx, (size_t)x.rows * C));
          TCHK(launch_gc_pool_bwd_dx(a, dl, dctx, wg, dx, B, HW, C, s));
          RC(add_grad(n.in, dx));
          // d wg[c] = sum_b sum_p dl[b][p] x[b][p][c]; d bg = sum dl (zero up to rounding: a softmax ignores a shift)
          RC(alloc(&dwg_b, (size_t)B * C));
          RC(ensure_part((size_t)B * GC_CHUNKS * C));
          TCHK(launch_gc_wpool(x.p, dl, st->part, dwg_b, B, HW, C, s));
          RC(grad_buf(n.wkey + ".weight", &dwg));
          RC(colsum(dwg_b, B, C, dwg));
          RC(grad_buf(n.wkey + ".bias", &dbg));
          TCHK(launch_sum_small(sdl, B, dbg, s));
          break;
        }
        case N_GELU: {
          const TT& x = st->t[n.in];
          float* dx;
          RC(alloc(&dx, (size_t)x.rows * x.cols));
          TCHK(launch_ew(st->t[n.out].grad, x.p, dx, (size_t)x.rows * x.cols, EW_GELU_BWD, s));
          RC(add_grad(n.in, dx));
          break;
        }
        case N_EMBED: {
          float* dE;
          RC(grad_buf(n.wkey, &dE));
          const d2t_config& g = c->cfg;
          TCHK(launch_embed_bwd(st->t[n.out].grad, st->tgt, dE, st->B * st->L, g.vocab, g.dec_dim, sqrtf((float)g.dec_dim), 0, s));
          break;
        }
      }
      // gradients this node wrote are final once the stream reaches this point (a parameter written by several
      // nodes is re-recorded by each): d2t_train_grad on another stream waits for exactly this
      for (const std::string& k : st->touched) TCHK(hipEventRecord(st->ready[k], s));
      st->touched.clear();
    }
    return D2T_OK;
  }
};

}  // namespace

extern "C" {

int d2t_train_forward(d2t_ctx* c, const float* image, int32_t B, int32_t H, int32_t W, const int64_t* tgt, int32_t L,
                      float* logits, d2t_stream stream) {
  DevGuard dg_(c);
  if (!c || !image || !tgt || !logits || B < 1 || H < 1 || W < 1 || L < 1) return fail(c, D2T_EINVAL, "bad argument");
  if (int rc = check_dev_ptr(c, image, "image")) return rc;
  if (int rc = check_dev_ptr(c, logits, "logits")) return rc;
  const d2t_config& g = c->cfg;
  const bool lstm = g.decoder == D2T_DEC_ATTN;
  const bool lstm_enc = g.encoder == D2T_ENC_VGG_BILSTM || g.encoder == D2T_ENC_RESNET_BILSTM;
  if (lstm && g.encoder == D2T_ENC_RESNET) return fail(c, D2T_ESTATE, "the Attn training step needs the HybridViT or a BiLSTM encoder");
  if (!lstm && lstm_enc) return fail(c, D2T_ESTATE, "the BiLSTM encoders train with the Attn / Attnv2 heads");
  if (lstm && L != g.batch_max_length + 1) return fail(c, D2T_EINVAL, "the Attn head trains on batch_max_length + 1 = %d steps", g.batch_max_length + 1);
  if (!lstm && L > g.max_seq_len + 1) return fail(c, D2T_EINVAL, "teacher sequence longer than max_seq_len + 1");
  if (!c->train) c->train = new d2t_train_state();
  d2t_train_state* st = c->train;
  st->tape.reset();
  st->t.clear();
  st->nodes.clear();
  st->have_forward = false;
  st->masks.clear();
  st->drop_site = 0;
  ++st->drop_calls;
  st->tgt = tgt; st->image = image; st->B = B; st->H = H; st->W = W; st->L = L;
  Tr tr{c, st, (hipStream_t)stream};
  int feat, mem, out;
  if (g.encoder == D2T_ENC_HYBRID_VIT) {
    const std::string sp = "seqmodeler.SequenceModeling.";
    RC(tr.backbone(image, B, H, W, sp + "patch_embed.backbone.ConvNet.", &feat));
    RC(tr.vit(feat, sp, &mem, nullptr));
  } else if (lstm_enc) {  // VGG | ResNet -> mean over the height -> 2 x BidirectionalLSTM (build_feat.py:50-55, build_seq.py)
    const std::string fp = "featextractor.FeatureExtraction.ConvNet.";
    if (g.encoder == D2T_ENC_VGG_BILSTM) RC(tr.vgg(image, B, H, W, fp, &feat));
    else RC(tr.backbone(image, B, H, W, fp, &feat));
    int seq;
    RC(tr.mean_h(feat, &seq));
    const int T = st->t[feat].W;
    RC(tr.bilstm(seq, 0, "seqmodeler.SequenceModeling.0.", B, T, &seq));
    RC(tr.bilstm(seq, 1, "seqmodeler.SequenceModeling.1.", B, T, &mem));
  } else {  // Feat=ResNet, Seq=None: + PositionalEncoding2D, [B,C,H,W] -> [B,HW,C] (already the NHWC row layout)
    RC(tr.backbone(image, B, H, W, "featextractor.FeatureExtraction.ConvNet.", &feat));
    const TT f = st->t[feat];
    const float* pe;
    RC(d2t_internal_pe2d(c, f.H, f.W, f.cols, (hipStream_t)stream, &pe));
    RC(tr.add_const(feat, pe, &mem));
  }
  if (lstm) RC(tr.lstm_decoder(mem, tgt, B, L, logits, &out));
  else RC(tr.decoder(mem, tgt, B, L, "predicter.Prediction.", logits, &out));
  st->logits_id = out;
  st->have_forward = true;
  return D2T_OK;
}

int d2t_train_backward(d2t_ctx* c, const float* dlogits, d2t_stream stream) {
  DevGuard dg_(c);
  if (!c || !dlogits) return fail(c, D2T_EINVAL, "bad argument");
  d2t_train_state* st = c->train;
  if (!st || !st->have_forward) return fail(c, D2T_ESTATE, "d2t_train_backward without a preceding d2t_train_forward");
  Tr tr{c, st, (hipStream_t)stream};
  st->t[st->logits_id].grad = const_cast<float*>(dlogits);
  st->have_forward = false;  // the tape is consumed (the attention probabilities are overwritten)
  st->touched.clear();
  return tr.backward();
}

int d2t_train_grad(d2t_ctx* c, const char* name, float* dst, int64_t numel, d2t_stream stream) {
  DevGuard dg_(c);
  if (!c || !name || !dst) return fail(c, D2T_EINVAL, "bad argument");
  d2t_train_state* st = c->train;
  if (!st) return fail(c, D2T_ESTATE, "no training step has run");
  auto it = st->grads.find(name);
  if (it == st->grads.end()) return fail(c, D2T_ESTATE, "no gradient for '%s'", name);
  const RawW* r = find(c, name);
  if (!r || (int64_t)r->numel != numel) return fail(c, D2T_EINVAL, "gradient '%s': size mismatch", name);
  // ordered after the backward kernels that produce this gradient, whichever stream `stream` is
  HIPCHK(c, hipStreamWaitEvent((hipStream_t)stream, st->ready[name], 0));
  HIPCHK(c, hipMemcpyAsync(dst, it->second, (size_t)numel * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return D2T_OK;
}

// launch one multi-copy kernel over `table` on stream s; the table travels through a pinned block of the gather ring
static int run_copy_table(d2t_ctx* c, d2t_train_state* st, const std::vector<CopyChunk>& table, hipStream_t s) {
  if (table.empty()) return D2T_OK;
  const size_t bytes = table.size() * sizeof(CopyChunk);
  d2t_train_state::GatherStage& g = st->gather[st->gather_calls++ & 3];
  if (g.pending) {
    HIPCHK(c, hipEventSynchronize(g.ev));
    g.pending = false;
  }
  if (bytes > g.cap) {
    if (g.h) { HIPCHK(c, hipHostFree(g.h)); HIPCHK(c, hipFree(g.d)); }
    g.h = g.d = nullptr;
    g.cap = 0;
    HIPCHK(c, hipHostMalloc(&g.h, bytes * 2, hipHostMallocDefault));
    HIPCHK(c, hipMalloc(&g.d, bytes * 2));
    g.cap = bytes * 2;
    if (!g.ev) HIPCHK(c, hipEventCreateWithFlags(&g.ev, hipEventDisableTiming));
  }
  memcpy(g.h, table.data(), bytes);
  HIPCHK(c, hipMemcpyAsync(g.d, g.h, bytes, hipMemcpyHostToDevice, s));
  HIPCHK(c, hipEventRecord(g.ev, s));
  g.pending = true;
  HIPCHK(c, launch_multi_copy(reinterpret_cast<const CopyChunk*>(g.d), (int)table.size(), s));
  return D2T_OK;
}

int d2t_train_gather(d2t_ctx* c, int32_t source, int32_t n, const char* const* names, const int64_t* offsets,
                     const int64_t* numels, float* flat, d2t_stream stream) {
  DevGuard dg_(c);
  if (!c || n < 0 || (n && (!names || !offsets || !numels || !flat)) || (source != 0 && source != 1)) return fail(c, D2T_EINVAL, "bad argument");
  d2t_train_state* st = c->train;
  if (source == 0 && !st) return fail(c, D2T_ESTATE, "no training step has run");
  if (!st) st = c->train = new d2t_train_state();
  if (int rc = check_dev_ptr(c, flat, "flat")) return rc;
  hipStream_t s = (hipStream_t)stream;
  constexpr long long CHUNK = 1 << 16;  // floats per block
  std::vector<CopyChunk> table;
  for (int i = 0; i < n; ++i) {
    const RawW* r = find(c, names[i]);
    if (!r || (int64_t)r->numel != numels[i]) return fail(c, D2T_EINVAL, "tensor '%s': unknown or size mismatch", names[i]);
    const float* src = r->p;
    if (source == 0) {
      auto it = st->grads.find(names[i]);
      if (it == st->grads.end()) return fail(c, D2T_ESTATE, "no gradient for '%s'", names[i]);
      src = it->second;
      HIPCHK(c, hipStreamWaitEvent(s, st->ready[names[i]], 0));  // (the same stream as the backward: no-ops)
    }
    for (long long o = 0; o < numels[i]; o += CHUNK)
      table.push_back(CopyChunk{src + o, flat + offsets[i] + o, std::min<long long>(CHUNK, numels[i] - o)});
  }
  return run_copy_table(c, st, table, s);
}

/* Refresh the engine's copies of tensors that are already loaded (same names, same sizes) from device memory, all in one
 * kernel: what the step after optimizer.step() needs -- every parameter changed, and one d2t_load_weight per tensor
 * is ~400 four-microsecond device copies.  Anything not loaded yet (or resized) is an error: use d2t_load_weight. */
int d2t_reload_weights(d2t_ctx* c, int32_t n, const char* const* names, const float* const* srcs, const int64_t* numels,
                       d2t_stream stream) {
  DevGuard dg_(c);
  if (!c || n < 0 || (n && (!names || !srcs || !numels))) return fail(c, D2T_EINVAL, "bad argument");
  if (!c->train) c->train = new d2t_train_state();
  hipStream_t s = (hipStream_t)stream;
  constexpr long long CHUNK = 1 << 16;
  std::vector<CopyChunk> table;
  for (int i = 0; i < n; ++i) {
    auto it = c->raw.find(names[i]);
    if (it == c->raw.end() || !it->second.p || (int64_t)it->second.numel != numels[i])
      return fail(c, D2T_EINVAL, "tensor '%s': not loaded or size mismatch", names[i]);
    if (int rc = check_dev_ptr(c, srcs[i], names[i])) return rc;
    for (long long o = 0; o < numels[i]; o += CHUNK)
      table.push_back(CopyChunk{srcs[i] + o, it->second.p + o, std::min<long long>(CHUNK, numels[i] - o)});
  }
  return run_copy_table(c, c->train, table, s);
}

int d2t_read_weight(d2t_ctx* c, const char* name, float* dst, int64_t numel, d2t_stream stream) {
  DevGuard dg_(c);
  if (!c || !name || !dst) return fail(c, D2T_EINVAL, "bad argument");
  const RawW* r = find(c, name);
  if (!r) return fail(c, D2T_ESTATE, "unknown tensor '%s'", name);
  if ((int64_t)r->numel != numel) return fail(c, D2T_EINVAL, "tensor '%s': size mismatch", name);
  HIPCHK(c, hipMemcpyAsync(dst, r->p, (size_t)numel * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return D2T_OK;
}

int d2t_train_set_dropout(d2t_ctx* c, float p, uint64_t seed) {
  DevGuard dg_(c);
  if (!c || !(p >= 0.f) || p >= 1.f) return fail(c, D2T_EINVAL, "dropout probability must be in [0, 1)");
  if (!c->train) c->train = new d2t_train_state();
  if (c->train->drop_seed != seed) c->train->drop_calls = 0;
  c->train->drop_p = p;
  c->train->drop_seed = seed;
  return D2T_OK;
}

int d2t_train_set_teacher_flags(d2t_ctx* c, const uint8_t* flags, int32_t n) {
  DevGuard dg_(c);
  if (!c || n < 0 || (n > 0 && !flags)) return fail(c, D2T_EINVAL, "bad argument");
  if (!c->train) c->train = new d2t_train_state();
  c->train->teacher_flags.assign(flags, flags + n);
  return D2T_OK;
}

int d2t_train_read_mask(d2t_ctx* c, int32_t index, uint8_t* dst, int64_t numel, d2t_stream stream) {
  DevGuard dg_(c);
  if (!c || !dst) return fail(c, D2T_EINVAL, "bad argument");
  d2t_train_state* st = c->train;
  if (!st || index < 0 || (size_t)index >= st->masks.size()) return fail(c, D2T_EINVAL, "no dropout mask %d", index);
  if ((int64_t)st->masks[index].second != numel) return fail(c, D2T_EINVAL, "mask %d has %zu elements", index, st->masks[index].second);
  HIPCHK(c, hipMemcpyAsync(dst, st->masks[index].first, (size_t)numel, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return D2T_OK;
}

int d2t_train_mask_count(d2t_ctx* c) { return c && c->train ? (int)c->train->masks.size() : 0; }

// The discrete decisions of the last training forward, in network order: one entry per ReLU (convolution + BatchNorm
// [+ residual] + ReLU nodes, decoder linear1) and per max-pool.  A test replays them in the oracle (ReLU -> multiply by the
// keep mask, max-pool -> gather of the recorded window element), which makes oracle and engine evaluate the SAME smooth
// function: gradients then agree to rounding everywhere, with no "a tie may have flipped" allowance.
static bool is_decision(const Node& n) {
  return n.kind == N_POOL || n.kind == N_RELU || ((n.kind == N_CONV || n.kind == N_LINEAR) && n.relu);
}

int d2t_train_decision_count(d2t_ctx* c) {
  if (!c || !c->train) return 0;
  int k = 0;
  for (const Node& n : c->train->nodes) k += is_decision(n);
  return k;
}

// index-th decision: is_pool_out = 1 for a max-pool (bytes = window element kh*2+kw per output element, NHWC), 0 for a
// ReLU keep mask (bytes = y > 0 per element, rows x cols); numel_out = its element count.  dst may be NULL (query only).
int d2t_train_read_decision(d2t_ctx* c, int32_t index, uint8_t* dst, int64_t numel, int32_t* is_pool_out, int64_t* numel_out,
                            d2t_stream stream) {
  DevGuard dg_(c);
  if (!c || !c->train) return fail(c, D2T_ESTATE, "no training forward has run");
  d2t_train_state* st = c->train;
  int k = 0;
  for (const Node& n : st->nodes) {
    if (!is_decision(n)) continue;
    if (k++ != index) continue;
    const TT& y = st->t[n.out];
    const int64_t ne = (int64_t)y.rows * y.cols;
    if (is_pool_out) *is_pool_out = n.kind == N_POOL;
    if (numel_out) *numel_out = ne;
    if (!dst) return D2T_OK;
    if (numel != ne) return fail(c, D2T_EINVAL, "decision %d has %lld elements", index, (long long)ne);
    if (n.kind == N_POOL) {
      const TT& x = st->t[n.in];
      HIPCHK(c, launch_pool_argmax(x.p, dst, x.B, x.H, x.W, x.cols, n.SH, n.SW, n.PH, n.PW, (hipStream_t)stream, n.KW));
    } else {
      HIPCHK(c, launch_relu_mask(y.p, dst, (size_t)ne, (hipStream_t)stream));
    }
    return D2T_OK;
  }
  return fail(c, D2T_EINVAL, "no decision %d", index);
}

void d2t_train_release(d2t_ctx* c) {
  DevGuard dg_(c);
  if (c && c->train) {
    hipDeviceSynchronize();
    delete c->train;
    c->train = nullptr;
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// Op-level test entry points: ONE node of the training tape, forward + backward, on caller tensors (row-major
// [rows][cols], NHWC maps).  They drive exactly the builders / backward functions the full step uses (a throw-away
// context holds the operands under fake keys), so tests/test_ops_gpu.py can hold every backward kernel -- data gradients,
// weight gradients in both arithmetic modes, BatchNorm / LayerNorm / attention / max-pool backward -- against float64
// torch autograd one op at a time.  Synchronous; not on any hot path.
// ---------------------------------------------------------------------------------------------------------------------
namespace {
struct OpCtx {
  d2t_ctx c;
  d2t_train_state st;
  Tr tr;
  OpCtx(int bf16x3, hipStream_t s) : tr{&c, &st, s} {
    c.conv_bf16x3 = bf16x3 != 0;
    hipGetDevice(&c.device);
    if (hipMalloc(&c.zero_page, 256) == hipSuccess) hipMemset(c.zero_page, 0, 256);  // out-of-image taps of the LDS-DMA kernels
  }
  ~OpCtx() {
    if (c.zero_page) hipFree(c.zero_page);
    c.zero_page = nullptr;
  }
  void put(const char* key, const float* p, std::vector<int64_t> shape) {
    RawW r;
    r.p = const_cast<float*>(p);
    r.shape = shape;
    r.numel = 1;
    for (auto v : shape) r.numel *= (size_t)v;
    c.raw[key] = r;
  }
  int out(float* dst, const float* src, size_t n) {
    if (!dst) return D2T_OK;
    if (!src) return hipMemsetAsync(dst, 0, n * 4, tr.s) == hipSuccess ? D2T_OK : D2T_EHIP;
    return hipMemcpyAsync(dst, src, n * 4, hipMemcpyDeviceToDevice, tr.s) == hipSuccess ? D2T_OK : D2T_EHIP;
  }
  int grad(float* dst, const char* key, size_t n) {
    auto it = st.grads.find(key);
    return out(dst, it == st.grads.end() ? nullptr : it->second, n);
  }
  int finish(int rc) {
    hipStreamSynchronize(tr.s);
    c.raw.clear();  // caller memory: nothing to free
    return rc;
  }
};
}  // namespace

// Fused cross-entropy, reduction 'none' (the criterion of engine/training.py:50-53): no context needed.
int d2t_ce_forward(const float* logits, const int64_t* target, float* loss, float* lse, int32_t rows, int32_t V,
                   int64_t ignore_index, d2t_stream stream) {
  if (!logits || !target || !loss || !lse || rows < 0 || V < 1) return D2T_EINVAL;
  return launch_ce_fwd(logits, target, loss, lse, rows, V, ignore_index, (hipStream_t)stream) == hipSuccess ? D2T_OK : D2T_EHIP;
}
int d2t_ce_backward(const float* logits, const int64_t* target, const float* lse, const float* dloss, float* dlogits,
                    int32_t rows, int32_t V, int64_t ignore_index, d2t_stream stream) {
  if (!logits || !target || !lse || !dloss || !dlogits || rows < 0 || V < 1) return D2T_EINVAL;
  return launch_ce_bwd(logits, target, lse, dloss, dlogits, rows, V, ignore_index, (hipStream_t)stream) == hipSuccess ? D2T_OK : D2T_EHIP;
}

int d2t_op_train_conv(const float* x, const float* w, const float* bias, const float* gamma, const float* beta,
                      const float* residual, const float* dy, float* y, float* dx, float* dw, float* dbias,
                      float* dgamma, float* dbeta, float* dres, int32_t B, int32_t H, int32_t W, int32_t Cin,
                      int32_t Cout, int32_t KH, int32_t KW, int32_t SH, int32_t SW, int32_t PH, int32_t PW, int32_t relu,
                      int32_t bf16x3, d2t_stream stream) {
  if (!x || !w || !dy || (gamma && !beta) || (!gamma && !bias && Cin != 1) || SH < 1 || SW < 1) return D2T_EINVAL;
  if (Cin != 1 && Cin % 32) return D2T_EINVAL;
  OpCtx o(bf16x3, (hipStream_t)stream);
  Tr& tr = o.tr;
  o.put("w.weight", w, {Cout, Cin, KH, KW});
  if (bias) o.put("w.bias", bias, {Cout});
  float *rm = nullptr, *rv = nullptr;
  if (gamma) {
    o.put("bn.weight", gamma, {Cout});
    o.put("bn.bias", beta, {Cout});
    RC(tr.zeros(&rm, Cout));
    RC(tr.zeros(&rv, Cout));
    o.put("bn.running_mean", rm, {Cout});
    o.put("bn.running_var", rv, {Cout});
  }
  int xin = -1, res = -1, out = -1;
  auto run = [&]() -> int {
    if (Cin == 1) {  // the stem: 3x3, stride 1, pad 1, BatchNorm + ReLU (resnet.py:205-207)
      if (KH != 3 || KW != 3 || SH != 1 || SW != 1 || PH != 1 || PW != 1 || !gamma || residual || !relu || Cout != 32)
        return fail(&o.c, D2T_EINVAL, "stem geometry");
      o.st.image = x; o.st.B = B; o.st.H = H; o.st.W = W;
      RC(tr.stem(x, B, H, W, "w", "bn", &out));
    } else {
      RC(tr.new_tensor((long long)B * H * W, Cin, &xin, B, H, W, const_cast<float*>(x)));
      const int OH = (H + 2 * PH - KH) / SH + 1, OW = (W + 2 * PW - KW) / SW + 1;
      if (residual) RC(tr.new_tensor((long long)B * OH * OW, Cout, &res, B, OH, OW, const_cast<float*>(residual)));
      RC(tr.conv_bn(xin, "w", gamma ? "bn" : "", Cout, KH, KW, SH, SW, PH, PW, relu != 0, res, &out));
    }
    const TT yo = o.st.t[out];
    RC(o.out(y, yo.p, (size_t)yo.rows * yo.cols));
    o.st.t[out].grad = const_cast<float*>(dy);
    RC(tr.backward());
    if (xin >= 0) RC(o.out(dx, o.st.t[xin].grad, (size_t)B * H * W * Cin));
    if (res >= 0) RC(o.out(dres, o.st.t[res].grad, (size_t)yo.rows * yo.cols));
    RC(o.grad(dw, "w.weight", (size_t)Cout * Cin * KH * KW));
    if (bias && !gamma) RC(o.grad(dbias, "w.bias", Cout));
    if (gamma) { RC(o.grad(dgamma, "bn.weight", Cout)); RC(o.grad(dbeta, "bn.bias", Cout)); }
    return D2T_OK;
  };
  return o.finish(run());
}

int d2t_op_train_linear(const float* x, const float* w, const float* bias, const float* residual, const float* dy, float* y,
                        float* dx, float* dw, float* dbias, float* dres, int32_t M, int32_t K, int32_t N, int32_t relu,
                        int32_t bf16x3, d2t_stream stream) {
  if (!x || !w || !bias || !dy || K % 32) return D2T_EINVAL;
  OpCtx o(bf16x3, (hipStream_t)stream);
  Tr& tr = o.tr;
  o.put("l.weight", w, {N, K});
  o.put("l.bias", bias, {N});
  auto run = [&]() -> int {
    int xin, res = -1, out;
    RC(tr.new_tensor(M, K, &xin, 0, 0, 0, const_cast<float*>(x)));
    if (residual) RC(tr.new_tensor(M, N, &res, 0, 0, 0, const_cast<float*>(residual)));
    RC(tr.linear(xin, "l", N, K, 0, relu ? ACT_RELU : ACT_NONE, res, &out));
    RC(o.out(y, o.st.t[out].p, (size_t)M * N));
    o.st.t[out].grad = const_cast<float*>(dy);
    RC(tr.backward());
    RC(o.out(dx, o.st.t[xin].grad, (size_t)M * K));
    if (res >= 0) RC(o.out(dres, o.st.t[res].grad, (size_t)M * N));
    RC(o.grad(dw, "l.weight", (size_t)N * K));
    RC(o.grad(dbias, "l.bias", N));
    return D2T_OK;
  };
  return o.finish(run());
}

int d2t_op_train_layernorm(const float* x, const float* gamma, const float* beta, const float* dy, float* y, float* dx,
                           float* dgamma, float* dbeta, int32_t rows, int32_t D, float eps, d2t_stream stream) {
  if (!x || !gamma || !beta || !dy) return D2T_EINVAL;
  OpCtx o(0, (hipStream_t)stream);
  Tr& tr = o.tr;
  o.put("n.weight", gamma, {D});
  o.put("n.bias", beta, {D});
  auto run = [&]() -> int {
    int xin, out;
    RC(tr.new_tensor(rows, D, &xin, 0, 0, 0, const_cast<float*>(x)));
    RC(tr.layernorm(xin, "n", eps, &out));
    RC(o.out(y, o.st.t[out].p, (size_t)rows * D));
    o.st.t[out].grad = const_cast<float*>(dy);
    RC(tr.backward());
    RC(o.out(dx, o.st.t[xin].grad, (size_t)rows * D));
    RC(o.grad(dgamma, "n.weight", D));
    RC(o.grad(dbeta, "n.bias", D));
    return D2T_OK;
  };
  return o.finish(run());
}

// q [nb*Lq][heads*hd], kv [nb*Lk][2*heads*hd] (keys in the first half of a row, values in the second); keytok: optional
// [nb][Lk] token ids whose PAD (0) entries are masked out (tgt_key_padding_mask, tfm.py:107-110); causal: additive -inf mask
int d2t_op_train_attention(const float* q, const float* kv, const int64_t* keytok, const float* dy, float* y, float* dq,
                           float* dkv, int32_t nb, int32_t Lq, int32_t Lk, int32_t heads, int32_t hd, int32_t causal,
                           d2t_stream stream) {
  if (!q || !kv || !dy) return D2T_EINVAL;
  OpCtx o(0, (hipStream_t)stream);
  Tr& tr = o.tr;
  auto run = [&]() -> int {
    const int D = heads * hd;
    int qt, kt, out;
    RC(tr.new_tensor((long long)nb * Lq, D, &qt, 0, 0, 0, const_cast<float*>(q)));
    RC(tr.new_tensor((long long)nb * Lk, 2 * D, &kt, 0, 0, 0, const_cast<float*>(kv)));
    RC(tr.attention(qt, 0, kt, 0, D, nb, Lq, Lk, heads, hd, causal, keytok, &out));
    RC(o.out(y, o.st.t[out].p, (size_t)nb * Lq * D));
    float* g;  // the backward overwrites the incoming gradient's buffer in places: give it a private copy
    RC(tr.alloc(&g, (size_t)nb * Lq * D));
    RC(o.out(g, dy, (size_t)nb * Lq * D));
    o.st.t[out].grad = g;
    RC(tr.zeros(&o.st.t[qt].grad, (size_t)nb * Lq * D));
    RC(tr.zeros(&o.st.t[kt].grad, (size_t)nb * Lk * 2 * D));
    RC(tr.backward());
    RC(o.out(dq, o.st.t[qt].grad, (size_t)nb * Lq * D));
    RC(o.out(dkv, o.st.t[kt].grad, (size_t)nb * Lk * 2 * D));
    return D2T_OK;
  };
  return o.finish(run());
}

int d2t_op_train_maxpool(const float* x, const float* dy, float* y, float* dx, int32_t B, int32_t H, int32_t W, int32_t C,
                         int32_t SH, int32_t SW, int32_t PH, int32_t PW, d2t_stream stream) {
  if (!x || !dy) return D2T_EINVAL;
  OpCtx o(0, (hipStream_t)stream);
  Tr& tr = o.tr;
  auto run = [&]() -> int {
    int xin, out;
    RC(tr.new_tensor((long long)B * H * W, C, &xin, B, H, W, const_cast<float*>(x)));
    RC(tr.pool(xin, SH, SW, PH, PW, &out));
    const TT yo = o.st.t[out];
    RC(o.out(y, yo.p, (size_t)yo.rows * C));
    o.st.t[out].grad = const_cast<float*>(dy);
    RC(tr.backward());
    RC(o.out(dx, o.st.t[xin].grad, (size_t)B * H * W * C));
    return D2T_OK;
  };
  return o.finish(run());
}

}  // extern "C"
