// libd2t engine: context, weight packing, encoder / decoder orchestration and
// the C-ABI declared in include/d2t.h.  Host code only launches the kernels of
// conv_mfma.hip / ops.hip on the caller's HIP stream; there is no CPU compute
// path and no fallback.
#include "ctx.h"
#include <mutex>

namespace {

int pack_conv(d2t_ctx* c, const std::string& conv, const std::string& bn, ConvW* out, hipStream_t s) {
  const RawW *w, *g = nullptr, *b = nullptr, *mu = nullptr, *var = nullptr;
  int rc = need(c, conv + ".weight", &w);
  if (rc) return rc;
  if (w->shape.size() != 4) return fail(c, D2T_EINVAL, "conv weight '%s' is not 4-D", conv.c_str());
  const RawW* cb = find(c, conv + ".bias");
  if (!bn.empty()) {
    if ((rc = need(c, bn + ".weight", &g)) || (rc = need(c, bn + ".bias", &b)) ||
        (rc = need(c, bn + ".running_mean", &mu)) || (rc = need(c, bn + ".running_var", &var)))
      return rc;
  }
  out->Cout = (int)w->shape[0]; out->Cin = (int)w->shape[1]; out->KH = (int)w->shape[2]; out->KW = (int)w->shape[3];
  void *pw, *pb;
  if ((rc = dev_alloc(c, &pw, w->numel * 4)) || (rc = dev_alloc(c, &pb, (size_t)out->Cout * 4))) return rc;
  c->owned.push_back(pw);
  c->owned.push_back(pb);
  out->w = (float*)pw;
  out->bias = (float*)pb;
  HIPCHK(c, launch_pack_conv(w->p, cb ? cb->p : nullptr, g ? g->p : nullptr, b ? b->p : nullptr, mu ? mu->p : nullptr,
                             var ? var->p : nullptr, 1e-5f, out->w, out->bias, out->Cout, out->Cin, out->KH, out->KW,
                             s));
  out->w_h16 = out->w_l16 = nullptr;  // fp16 hi / lo planes: made on first use (f16_planes), only the two-MFMA modes read them
  if (out->Cin % 32 == 0) {  // bf16 hi/lo planes of the folded weights (bf16x3 kernel)
    void *ph = nullptr, *pl = nullptr;
    if ((rc = dev_alloc(c, &ph, w->numel * 2))) return rc;
    c->owned.push_back(ph);
    if ((rc = dev_alloc(c, &pl, w->numel * 2))) return rc;
    c->owned.push_back(pl);
    out->w_hi = (uint16_t*)ph;
    out->w_lo = (uint16_t*)pl;
    HIPCHK(c, launch_split_bf16(out->w, out->w_hi, out->w_lo, w->numel, s));
  }
  return D2T_OK;
}
// fp16 hi / lo planes of a packed convolution weight (fp16x2 / mixed precision), created the first time a launch needs them:
// contexts that never leave split-bf16 (the default; training; fp32) do not pay their memory
int f16_planes(d2t_ctx* c, ConvW& w, hipStream_t s) {
  if (w.w_h16) return D2T_OK;
  if (!w.w || w.Cin % 32) return fail(c, D2T_ESTATE, "no fp16 planes for this layer");
  const size_t n = (size_t)w.Cout * w.KH * w.KW * w.Cin + (size_t)w.Cout * w.K2;
  void *qh = nullptr, *ql = nullptr;
  int rc;
  if ((rc = dev_alloc(c, &qh, n * 2))) return rc;
  c->owned.push_back(qh);
  if ((rc = dev_alloc(c, &ql, n * 2))) return rc;
  c->owned.push_back(ql);
  HIPCHK(c, launch_split_f16(w.w, (uint16_t*)qh, (uint16_t*)ql, n, s));
  w.w_h16 = (uint16_t*)qh;
  w.w_l16 = (uint16_t*)ql;
  return D2T_OK;
}
int get_lin(d2t_ctx* c, const std::string& k, LinW* out, int N, int K, bool bias = true) {
  const RawW *w, *b = nullptr;
  int rc;
  if ((rc = need(c, k + ".weight", &w, {N, K})) || (bias && (rc = need(c, k + ".bias", &b, {N})))) return rc;
  *out = LinW{w->p, b ? b->p : nullptr, N, K};
  return D2T_OK;
}
// bf16 hi/lo planes of a Linear's weight for the bf16x3 GEMM kernel (K % 32 == 0)
int split_planes(d2t_ctx* c, const float* w, size_t n, const uint16_t** hi, const uint16_t** lo, hipStream_t s) {
  void *ph, *pl;
  int rc;
  if ((rc = dev_alloc(c, &ph, n * 2)) || (rc = dev_alloc(c, &pl, n * 2))) return rc;
  c->owned.push_back(ph);
  c->owned.push_back(pl);
  HIPCHK(c, launch_split_bf16(w, (uint16_t*)ph, (uint16_t*)pl, n, s));
  *hi = (const uint16_t*)ph;
  *lo = (const uint16_t*)pl;
  return D2T_OK;
}
int split_lin(d2t_ctx* c, LinW* l, hipStream_t s) {
  if (l->K % 32) return D2T_OK;
  return split_planes(c, l->w, (size_t)l->N * l->K, &l->w_hi, &l->w_lo, s);
}

int get_ln(d2t_ctx* c, const std::string& k, LNW* out, int D) {
  const RawW *g, *b;
  int rc;
  if ((rc = need(c, k + ".weight", &g, {D})) || (rc = need(c, k + ".bias", &b, {D}))) return rc;
  *out = LNW{g->p, b->p};
  return D2T_OK;
}

void free_packed(d2t_ctx* c) {
  for (void* p : c->owned) hipFree(p);
  c->owned.clear();
  c->pos_interp.clear();  // resized copies of the previous position table (owned buffers)
  for (int i = 0; i < 4; ++i) c->layers[i].clear();
  c->vit.clear();
  c->dec.clear();
  c->finalized = false;
}

// spatial size after the backbone (resnet.py:205-245) for an H x W crop
void backbone_hw(int H, int W, int* oh, int* ow) {
  int h = H / 2, w = W / 2;          // maxpool1
  h /= 2; w /= 2;                    // maxpool2
  h = (h - 2) / 2 + 1; w = w + 1;    // maxpool3 k2 s(2,1) p(0,1)
  h = (h - 2) / 2 + 1; w = w + 1;    // conv4_1  k2 s(2,1) p(0,1)
  *oh = h - 1; *ow = w - 1;          // conv4_2  k2 s1 p0
}

// launch_conv, bracketed by HIP events on `s` while profiling is enabled
hipError_t conv_timed(d2t_ctx* c, const ConvP& p, hipStream_t s) {
  if (!c->profiling) return launch_conv(p, s);
  d2t_ctx::ProfRec r{p.M, p.Cout, p.K, nullptr, nullptr};
  hipError_t e;
  if ((e = hipEventCreate(&r.a)) != hipSuccess || (e = hipEventCreate(&r.b)) != hipSuccess) return e;
  if ((e = hipEventRecord(r.a, s)) != hipSuccess) return e;
  e = launch_conv(p, s);
  hipError_t e2 = hipEventRecord(r.b, s);
  c->prof.push_back(r);
  return e != hipSuccess ? e : e2;
}

// One convolution of the backbone.  `res` may be nullptr; `out_split` asks for split-bf16 output planes
// (only meaningful on the bf16x3 path; the consumer must be another bf16x3 convolution or the split pool).
// pool2: fuse the 2x2 / stride 2 max-pool that follows (split-record path only: see ConvP::pool2); y is then the POOLED map.
Act conv(d2t_ctx* c, hipStream_t s, hipError_t* err, const Act& x, const ConvW& w, int sh, int sw, int ph, int pw,
         int act, const Act* res, float* outbuf, const ConvP* extra = nullptr, bool out_split = false, bool pool2 = false,
         int out_fmt = -1) {
  // out_fmt: record format of a split output (conv_common.h REC_*); -1 = the input's (fp16x2 mode: one fp16; else bf16 hi | lo)
  Act y{outbuf, x.B, (x.H + 2 * ph - w.KH) / sh + 1, (x.W + 2 * pw - w.KW) / sw + 1, w.Cout};
  y.split = out_split;
  ConvP p{};
  if (extra) p = *extra;
  p.w = w.w; p.bias = w.bias;
  if (c->conv_bf16x3) { p.w_hi = w.w_hi; p.w_lo = w.w_lo; }
  if (x.split) { p.in_hi = x.planes(); p.zero16 = c->zero_page; p.max_blocks = c->conv_max_blocks; } else { p.in = x.p; }
  if (x.split && x.fmt != 0) {  // fp16 records in: the two-MFMA kernels (x16 * w_lo + x16 * w_hi), fp16 hi / lo weight planes
    if (f16_planes(c, const_cast<ConvW&>(w), s) != D2T_OK) { if (*err == hipSuccess) *err = hipErrorOutOfMemory; return y; }  // (w lives in *c)
    p.f16 = 1;
    p.w_hi = w.w_h16; p.w_lo = w.w_l16;
  }
  if (out_split) {
    y.fmt = out_fmt >= 0 ? out_fmt : (x.split ? (x.fmt ? 1 : 0) : (c->conv_f16 ? 1 : 0));
    p.out_fmt = 1 + y.fmt;
  }
  if (res && res->split) p.res_fmt = 1 + res->fmt;
  p.pipelined = c->conv_pipelined; p.reserved_cus = c->reserved_cus; p.split_tail = !c->decode_in_flight || D2T_PROBE_ENV("D2T_CONV_TAIL_ALWAYS");
  if (out_split) { p.out_hi = y.planes(); } else { p.out = outbuf; }
  if (res) {
    if (res->split) { p.res_hi = res->planes(); } else { p.res = res->p; }
  }
  p.B = x.B; p.H = x.H; p.W = x.W; p.Cin = x.C; p.OH = y.H; p.OW = y.W; p.Cout = w.Cout;
  p.KH = w.KH; p.KW = w.KW; p.SH = sh; p.SW = sw; p.PH = ph; p.PW = pw;
  p.M = y.B * y.H * y.W; p.K = w.KH * w.KW * x.C + p.Cin2; p.act = act;  // (Cin2: a second 1x1 input from `extra`)
  if (pool2) {  // rows in pooled order (floor: a last odd row / column belongs to no window and is never computed)
    p.pool2 = 1;
    p.M = 4 * y.B * (y.H / 2) * (y.W / 2);
    y.H /= 2;
    y.W /= 2;
  }
  hipError_t e = conv_timed(c, p, s);
  if (e != hipSuccess && *err == hipSuccess) *err = e;
  return y;
}

hipError_t linear_big(d2t_ctx* c, hipStream_t s, const float* x, const LinW& w, const float* res, float* y, int M,
                      int act) {
  ConvP p{};
  p.in = x; p.w = w.w; p.bias = w.b; p.res = res; p.out = y;
  if (c && c->conv_bf16x3 && w.w_hi) { p.w_hi = w.w_hi; p.w_lo = w.w_lo; }  // launch_conv picks the bf16x3 GEMM
  p.B = 1; p.H = 1; p.W = M; p.Cin = w.K; p.OH = 1; p.OW = M; p.Cout = w.N;
  p.KH = p.KW = p.SH = p.SW = 1; p.PH = p.PW = 0; p.M = M; p.K = w.K; p.act = act;
  return c ? conv_timed(c, p, s) : launch_conv(p, s);
}

hipError_t linear_any(d2t_ctx* c, hipStream_t s, const float* x, const LinW& w, const float* res, float* y, int M,
                      int act) {
  // always the MFMA GEMM when the shape allows it (not only for M > 64): a row's result must not depend on how many
  // rows share the launch, or a sample would decode differently alone and inside a batch
  if (w.K % 32 == 0) return linear_big(c, s, x, w, res, y, M, act);
  SkinnyP p{};
  p.x = x; p.w = w.w; p.bias = w.b; p.res = res; p.y = y;
  p.M = M; p.K = w.K; p.N = w.N; p.ldx = w.K; p.ldy = w.N; p.ldres = w.N; p.act = act;
  return launch_skinny(p, s);
}

// pick a rotating activation buffer that is not in `live`
float* pick(d2t_ctx* c, std::initializer_list<const float*> live) {
  for (int i = 0; i < 4; ++i) {
    bool used = false;
    for (const float* l : live) used |= (l == c->act[i]);
    if (!used) return c->act[i];
  }
  return nullptr;
}

// ResNet.forward (feature_extractor/resnet.py:205-245).  On the bf16x3 path every activation between the
// stem and the last convolution lives as split-bf16 planes; `final_split` says whether the returned map
// does too (a bf16x3 consumer follows) or is fp32 (final_out / any other consumer).
int run_backbone(d2t_ctx* c, hipStream_t s, const float* img, int B, int H, int W, Act* out, float* final_out,
                 const ConvP* final_extra, bool final_split) {
  hipError_t err = hipSuccess;
  // GlobalContext blocks read and update whole fp32 maps, so with gcb the activations stay fp32 (the convolutions still
  // take the split-bf16 kernel in bf16x3 mode, splitting their input on the fly)
  const bool sp = c->conv_bf16x3 && !c->cfg.gcb;
  Act x{pick(c, {}), B, H, W, c->stem.Cout};
  x.split = sp;
  const int f16 = sp && c->conv_f16;  // fp16 records between the stem and the last convolution
  x.fmt = f16;
  // mixed precision (ctx.h mixed_units): unit u of [layer3.1 .. layer3.4, conv3, layer4.0 .. layer4.2] runs the two-MFMA
  // arithmetic; a tensor that a mixed unit consumes is written as fp16 hi | lo pairs (its MFMAs read the hi half, the
  // residual add both), the map between a mixed block's two convolutions as one fp16
  const int nmix = (sp && !f16 && c->conv_pipelined == 3) ? c->mixed_units : 0;
  auto mixed = [&](int u) { return u >= 0 && u < nmix; };
  auto fmt_for = [&](int consumer_unit) { return mixed(consumer_unit) ? 2 : -1; };
  if (sp) HIPCHK(c, launch_stem_split(img, c->stem.w, c->stem.bias, x.planes(), B, H, W, c->stem.Cout, ACT_RELU, s, f16));
  else HIPCHK(c, launch_stem(img, c->stem.w, c->stem.bias, x.p, B, H, W, c->stem.Cout, ACT_RELU, s));
  // the two 2x2 / stride 2 max-pools (resnet.py:94,106) run inside the epilogue of the convolution in front of them on the
  // split-record path with the 16x16x32 kernels: conv0_2 writes 268 MB instead of 1.07 GB and no pool kernel re-reads it
  const bool fuse_pool = sp && c->conv_pipelined == 3 && !c->no_pool_fusion;
  x = conv(c, s, &err, x, c->conv0_2, 1, 1, 1, 1, ACT_RELU, nullptr, pick(c, {x.p}), nullptr, sp, fuse_pool);
  auto pool = [&](const Act& a, int sh, int sw, int ph, int pw) {
    Act y{pick(c, {a.p}), a.B, (a.H + 2 * ph - 2) / sh + 1, (a.W + 2 * pw - 2) / sw + 1, a.C};
    y.split = a.split;
    y.fmt = a.fmt;
    hipError_t e = a.split ? launch_maxpool_split(a.planes(), y.planes(), a.B, a.H, a.W, a.C, sh, sw, ph, pw, s, f16)
                           : launch_maxpool(a.p, y.p, a.B, a.H, a.W, a.C, sh, sw, ph, pw, s);
    if (e != hipSuccess && err == hipSuccess) err = e;
    return y;
  };
  auto stage = [&](int li) {
    int bi = -1;
    for (const Block& b : c->layers[li]) {
      ++bi;
      // this block's unit and the unit that consumes its output (layer3.4 -> conv3 = unit 4; layer4.2 -> conv4_1: never mixed)
      const int unit = li == 2 ? bi - 1 : li == 3 ? 5 + bi : -1;
      const int next = li == 2 ? bi : li == 3 ? (bi < 2 ? 6 + bi : -1) : -1;
      if (mixed(unit) && !b.has_down) {
        Act t = conv(c, s, &err, x, b.c1, 1, 1, 1, 1, ACT_RELU, nullptr, pick(c, {x.p}), nullptr, sp, false, 1);
        x = conv(c, s, &err, t, b.c2, 1, 1, 1, 1, ACT_RELU, &x, pick(c, {x.p, t.p}), nullptr, sp, false, mixed(next) ? 2 : 0);
        continue;
      }
      Act t = conv(c, s, &err, x, b.c1, 1, 1, 1, 1, ACT_RELU, nullptr, pick(c, {x.p}), nullptr, sp);
      if (b.has_down && b.c2cat.w && sp && c->conv_pipelined == 3 && b.c2.Cout >= 128 && !c->no_shortcut_fusion) {
        // the 1x1 shortcut inside conv2's launch: K-steps over x appended behind the taps over t, one accumulator, no residual
        ConvP ex{};
        ex.in2_hi = x.planes();
        ex.Cin2 = x.C;
        x = conv(c, s, &err, t, b.c2cat, 1, 1, 1, 1, ACT_RELU, nullptr, pick(c, {x.p, t.p}), &ex, sp, false, fmt_for(next));
        continue;
      }
      Act r = x;
      if (b.has_down) r = conv(c, s, &err, x, b.down, 1, 1, 0, 0, ACT_NONE, nullptr, pick(c, {x.p, t.p}), nullptr, sp);
      x = conv(c, s, &err, t, b.c2, 1, 1, 1, 1, ACT_RELU, &r, pick(c, {x.p, t.p, r.p}), nullptr, sp, false, fmt_for(next));
    }
    if (c->cfg.gcb) {  // resnet.py:200-201: GlobalContext closes the stage
      const int HW = x.H * x.W;
      float* ws = c->gc_ws;  // logits [B*HW] | ctx [B][C] | y [B][C]
      hipError_t e = launch_global_context(x.p, c->gc[li], ws, ws + (size_t)x.B * HW, ws + (size_t)x.B * HW + (size_t)x.B * x.C,
                                           x.B, HW, x.C, s);
      if (e != hipSuccess && err == hipSuccess) err = e;
    }
  };
  if (c->cfg.gcb) {
    const int rc = ensure(c, &c->gc_ws, &c->gc_ws_cap, ((size_t)B * (H / 2) * (W / 2) + 2 * (size_t)B * 512 + 64) * 4);
    if (rc) return rc;
  }
  if (!fuse_pool) x = pool(x, 2, 2, 0, 0);
  stage(0);
  const bool fuse_pool2 = fuse_pool && c->conv1.Cout >= 128;  // (the pipelined 16x16x32 kernel takes layers with >= 128 channels)
  x = conv(c, s, &err, x, c->conv1, 1, 1, 1, 1, ACT_RELU, nullptr, pick(c, {x.p}), nullptr, sp, fuse_pool2);
  if (!fuse_pool2) x = pool(x, 2, 2, 0, 0);
  stage(1);
  x = conv(c, s, &err, x, c->conv2, 1, 1, 1, 1, ACT_RELU, nullptr, pick(c, {x.p}), nullptr, sp);
  x = pool(x, 2, 1, 0, 1);
  stage(2);
  x = conv(c, s, &err, x, c->conv3, 1, 1, 1, 1, ACT_RELU, nullptr, pick(c, {x.p}), nullptr, sp, false, mixed(5) ? 2 : x.fmt == 2 ? 0 : -1);
  stage(3);
  x = conv(c, s, &err, x, c->conv4_1, 2, 1, 0, 1, ACT_RELU, nullptr, pick(c, {x.p}), nullptr, sp);
  x = conv(c, s, &err, x, c->conv4_2, 1, 1, 0, 0, ACT_RELU, nullptr, final_out ? final_out : pick(c, {x.p}),
           final_extra, sp && final_split && !final_out);
  if (err != hipSuccess) return fail(c, D2T_EHIP, "backbone launch: %s", hipGetErrorString(err));
  *out = x;
  return D2T_OK;
}

// Position rows a crop with patch grid gh x gw adds to its tokens ([1 + gh*gw][D], row 0 = the cls row): the loaded table,
// or -- ViTEncoder, D2T_VIT_POS_LEARNED_INTERP -- its bicubic resize (vit_encoder.py:58-95), built once per grid
int pos_table_for(d2t_ctx* c, int gh, int gw, hipStream_t s, const float** out) {
  const PosGrid pg = pos_grid(c->cfg, c->pos_GH, c->pos_GW, gh, gw);
  if (!pg.interp) { *out = c->pos_embed; return D2T_OK; }
  const int D = c->cfg.vit_dim;
  d2t_ctx::PosTab& t = c->pos_interp[std::make_pair(gh, gw)];
  if (!t.p) {
    void* p;
    if (int rc = dev_alloc(c, &p, (size_t)(gh * gw + 1) * D * 4)) return rc;
    c->owned.push_back(p);
    t.p = (float*)p;
    t.valid = false;
  }
  if (!t.valid) {
    HIPCHK(c, launch_copy(c->pos_embed, t.p, (size_t)D, s));  // class_pos_embedding is kept (:69,:93-95)
    HIPCHK(c, launch_bicubic_table(c->pos_embed + D, t.p + D, c->pos_GH, c->pos_GW, gh, gw, D, pg.sh, pg.sw, s));
    t.valid = true;
  }
  *out = t.p;
  return D2T_OK;
}

// PositionalEncoding2D crop [h][w][C] (common/postional_encoding.py:105-134,146-157)
int get_pe2d(d2t_ctx* c, int h, int w, int C, hipStream_t s, const float** out) {
  auto key = std::make_pair(h, w);
  auto it = c->pe2d.find(key);
  if (it != c->pe2d.end()) { *out = it->second; return D2T_OK; }
  const int half = C / 2;
  std::vector<float> div(half / 2);
  for (int i = 0; i < half / 2; ++i) div[i] = expf((float)(2 * i) * (float)(-std::log(10000.0) / (double)half));
  std::vector<float> host((size_t)h * w * C);
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      float* o = host.data() + ((size_t)y * w + x) * C;
      for (int i = 0; i < half / 2; ++i) {
        o[2 * i] = sinf((float)y * div[i]);
        o[2 * i + 1] = cosf((float)y * div[i]);
        o[half + 2 * i] = sinf((float)x * div[i]);
        o[half + 2 * i + 1] = cosf((float)x * div[i]);
      }
    }
  void* d;
  int rc = dev_alloc(c, &d, host.size() * 4);
  if (rc) return rc;
  HIPCHK(c, hipMemcpy(d, host.data(), host.size() * 4, hipMemcpyHostToDevice));
  c->pe2d[key] = (float*)d;
  *out = (float*)d;
  return D2T_OK;
}

}  // namespace

int d2t_internal_pe2d(d2t_ctx* c, int h, int w, int C, hipStream_t s, const float** out) { return get_pe2d(c, h, w, C, s, out); }
hipError_t d2t_internal_conv_timed(d2t_ctx* c, const ConvP& p, hipStream_t s) { return conv_timed(c, p, s); }

// Process-wide pool of the engine's decode streams.  The first streams a process creates get hardware queues of their own;
// streams created after others were destroyed can end up sharing one (measured: the SECOND context of a process decoded its
// three chains at the rate of ~2.4: 2830 instead of 3810 formulas/s on config C1).  A destroyed context's streams are
// therefore kept and handed, in the same roles, to the next context of that device and priority.
namespace {
std::mutex g_stream_mu;
std::map<std::pair<int, int>, std::vector<hipStream_t>> g_stream_pool;
hipError_t acquire_stream(int device, int prio, hipStream_t* out) {
  {
    std::lock_guard<std::mutex> lk(g_stream_mu);
    auto& v = g_stream_pool[std::make_pair(device, prio)];
    if (!v.empty()) {
      *out = v.back();
      v.pop_back();
      return hipSuccess;
    }
  }
  return hipStreamCreateWithPriority(out, hipStreamNonBlocking, prio);
}
void release_stream(int device, int prio, hipStream_t st) {
  hipStreamSynchronize(st);
  std::lock_guard<std::mutex> lk(g_stream_mu);
  g_stream_pool[std::make_pair(device, prio)].push_back(st);
}
}  // namespace

// ===========================================================================
// C-ABI
// ===========================================================================
extern "C" {

int d2t_device_available(void) {
  int n = 0;
  return hipGetDeviceCount(&n) == hipSuccess && n > 0;
}

int d2t_create(const d2t_config* cfg, d2t_ctx** out) {
  if (!cfg || !out) return D2T_EINVAL;
  *out = nullptr;
  d2t_ctx* c = new d2t_ctx();
  c->cfg = *cfg;
  *out = c;  // returned even on error so that d2t_last_error works; caller destroys it
  if (cfg->in_channels != 1) return fail(c, D2T_EINVAL, "in_channels must be 1 (grey crops)");
  if (cfg->backbone_out != 512) return fail(c, D2T_EINVAL, "backbone output_channel must be 512");
  if (cfg->encoder == D2T_ENC_HYBRID_VIT) {
    if (cfg->vit_dim != 256 && cfg->vit_dim != 512) return fail(c, D2T_EINVAL, "ViT hidden_size must be 256 or 512");
    if (cfg->vit_dim / cfg->vit_heads != 32) return fail(c, D2T_EINVAL, "ViT head_dim must be 32");
    if (cfg->patch_h < 1 || cfg->patch_w < 1) return fail(c, D2T_EINVAL, "bad patch size");
  } else if (cfg->encoder != D2T_ENC_RESNET && cfg->encoder != D2T_ENC_VGG_BILSTM &&
             cfg->encoder != D2T_ENC_RESNET_BILSTM) {
    return fail(c, D2T_EINVAL, "unknown encoder %d", cfg->encoder);
  }
  if (cfg->decoder == D2T_DEC_TFM) {
    const int hd = cfg->dec_heads > 0 ? cfg->dec_dim / cfg->dec_heads : 0;
    if (cfg->dec_dim != 256 && cfg->dec_dim != 512) return fail(c, D2T_EINVAL, "decoder d_model must be 256 or 512");
    if (hd != 32 && hd != 64) return fail(c, D2T_EINVAL, "decoder head_dim must be 32 or 64");
    if (cfg->dec_heads != 8) return fail(c, D2T_EINVAL, "decoder nhead must be 8");
    if (cfg->dec_ff % 64) return fail(c, D2T_EINVAL, "dim_feedforward must be a multiple of 64");
    if (cfg->max_seq_len + 2 > 512) return fail(c, D2T_EINVAL, "max_seq_len must be <= 510");
    if (cfg->encoder == D2T_ENC_VGG_BILSTM || cfg->encoder == D2T_ENC_RESNET_BILSTM)
      return fail(c, D2T_EINVAL, "BiLSTM encoders are paired with the Attn decoder only");
  } else if (cfg->decoder == D2T_DEC_ATTN) {
    if (cfg->attn_hidden != 256) return fail(c, D2T_EINVAL, "Attn hidden_size / input_size must be 256");
    if (cfg->attn_kernel_size < 0 || cfg->attn_kernel_size > 5) return fail(c, D2T_EINVAL, "Attn kernel_size must be <= 5");
    if (cfg->vocab > 1024) return fail(c, D2T_EINVAL, "Attn decoder supports num_class <= 1024");
    if (cfg->batch_max_length < 1) return fail(c, D2T_EINVAL, "batch_max_length must be >= 1");
    if (cfg->encoder == D2T_ENC_RESNET) return fail(c, D2T_EINVAL, "Feat=ResNet+Seq=None is paired with the TFM decoder only");
    if (cfg->encoder == D2T_ENC_HYBRID_VIT && cfg->vit_dim != 256) return fail(c, D2T_EINVAL, "Attn over ViT needs hidden_size 256");
    if ((cfg->encoder == D2T_ENC_VGG_BILSTM || cfg->encoder == D2T_ENC_RESNET_BILSTM) && cfg->bilstm_hidden != 256)
      return fail(c, D2T_EINVAL, "BiLSTM hidden_size must be 256");
  } else {
    return fail(c, D2T_EINVAL, "unknown decoder %d", cfg->decoder);
  }
  if (!d2t_device_available()) return fail(c, D2T_EHIP, "no HIP device visible");
  HIPCHK(c, hipGetDevice(&c->device));  // the calling thread's current device becomes the context's device
  c->cross_fp32 = D2T_PROBE_ENV("D2T_DECODE_CROSS_FP32") != 0;  // probe builds, A/B: the greedy cross-attention on the fp32 MFMA
  {  // the decode stream carries a latency-bound chain of small kernels: give it the highest priority so its
     // workgroups are placed first whenever the encoder of the next batch is filling the chip
    int lo = 0, hi = 0;
    HIPCHK(c, hipDeviceGetStreamPriorityRange(&lo, &hi));
    c->stream_prio = D2T_PROBE_ENV_STR("D2T_NO_PRIO") ? lo : hi;
    HIPCHK(c, acquire_stream(c->device, c->stream_prio, &c->dstream));
    for (int i = 1; i < d2t_ctx::MAXC; ++i) HIPCHK(c, acquire_stream(c->device, c->stream_prio, &c->chains[i].stream));
  }
  HIPCHK(c, hipEventCreateWithFlags(&c->ev_in, hipEventDisableTiming));
  for (int i = 0; i < d2t_ctx::MAXC; ++i) HIPCHK(c, hipEventCreateWithFlags(&c->ev_done[i], hipEventDisableTiming));
  HIPCHK(c, hipHostMalloc(reinterpret_cast<void**>(&c->h_pinned), 64, hipHostMallocDefault));
  HIPCHK(c, hipMalloc(&c->zero_page, 256));
  HIPCHK(c, hipMemset(c->zero_page, 0, 256));
  return D2T_OK;
}

void d2t_destroy(d2t_ctx* c) {
  DevGuard dg_(c);
  if (!c) return;
  hipDeviceSynchronize();
  d2t_train_release(c);
  for (auto& ge : c->graphs) hipGraphExecDestroy(ge.exec);
  c->graphs.clear();
  free_packed(c);
  for (auto& kv : c->raw) hipFree(kv.second.p);
  for (auto& kv : c->pe2d) hipFree(kv.second);
  for (int i = 0; i < 4; ++i) if (c->act[i]) hipFree(c->act[i]);
  for (int i = 0; i < d2t_ctx::MAXC; ++i) {
    if (c->ckv2[i]) hipFree(c->ckv2[i]);
    if (c->ev_done[i]) hipEventDestroy(c->ev_done[i]);
  }
  if (c->skv) hipFree(c->skv);
  if (c->skv_alt) hipFree(c->skv_alt);
  if (c->beam_ws) hipFree(c->beam_ws);
  if (c->h_beam) hipHostFree(c->h_beam);
  if (c->beam_qp) hipFree(c->beam_qp);
  if (c->dws) hipFree(c->dws);
  if (c->dstate) hipFree(c->dstate);
  if (c->h_pinned) hipHostFree(c->h_pinned);
  if (c->zero_page) hipFree(c->zero_page);
  if (c->gc_ws) hipFree(c->gc_ws);
  if (c->ev_in) hipEventDestroy(c->ev_in);
  if (c->h_steps) hipHostFree(c->h_steps);
  for (hipEvent_t ev : c->ticket_ev) if (ev) hipEventDestroy(ev);
  if (c->dout) hipFree(c->dout);
  for (int i = 0; i < d2t_ctx::MAXC; ++i) {
    if (i == c->active_chain) continue;  // (the active chain's buffers are the members freed around here)
    d2t_ctx::Chain& o = c->chains[i];
    if (o.skv) hipFree(o.skv);
    if (o.dws) hipFree(o.dws);
    if (o.dstate) hipFree(o.dstate);
    if (o.out) hipFree(o.out);
  }
  // the streams go back to the process-wide pool, last acquired first (the next context takes them in the same roles)
  for (int i = d2t_ctx::MAXC - 1; i >= 0; --i) {
    hipStream_t st = i == c->active_chain ? c->dstream : c->chains[i].stream;
    if (st) release_stream(c->device, c->stream_prio, st);
  }
  delete c;
}

const char* d2t_last_error(const d2t_ctx* c) { return c ? c->err.c_str() : "null ctx"; }

int d2t_device_of(const d2t_ctx* c) { return c ? c->device : -1; }

int d2t_load_weight(d2t_ctx* c, const char* name, const float* dev, const int64_t* shape, int32_t ndim,
                    d2t_stream stream) {
  DevGuard dg_(c);
  if (!c || !name || !dev || ndim < 0 || (ndim > 0 && !shape)) return fail(c, D2T_EINVAL, "bad argument");
  if (int rc = check_dev_ptr(c, dev, name)) return rc;
  size_t n = 1;
  std::vector<int64_t> shp(shape, shape + ndim);
  for (auto v : shp) n *= (size_t)v;
  RawW& r = c->raw[name];
  if (r.p && r.numel != n) { hipDeviceSynchronize(); hipFree(r.p); r.p = nullptr; }
  if (!r.p) {
    void* p;
    int rc = dev_alloc(c, &p, n * 4);
    if (rc) return rc;
    r.p = (float*)p;
  }
  r.shape = shp;
  r.numel = n;
  HIPCHK(c, hipMemcpyAsync(r.p, dev, n * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return D2T_OK;
}

int d2t_finalize_weights(d2t_ctx* c, d2t_stream stream) {
  DevGuard dg_(c);
  if (!c) return D2T_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  hipDeviceSynchronize();
  for (auto& ge : c->graphs) hipGraphExecDestroy(ge.exec);
  c->graphs.clear();
  free_packed(c);
  const d2t_config& g = c->cfg;
  int rc;
  const bool vit = g.encoder == D2T_ENC_HYBRID_VIT;
  const std::string sm = "seqmodeler.SequenceModeling.";
  c->bb = vit ? sm + "patch_embed.backbone.ConvNet." : "featextractor.FeatureExtraction.ConvNet.";
  const std::string& bb = c->bb;
  const bool is_vgg = g.encoder == D2T_ENC_VGG_BILSTM;
  const bool has_lstm = is_vgg || g.encoder == D2T_ENC_RESNET_BILSTM;
  if (is_vgg) {
    // VGG_FeatureExtractor (feature_extractor/vgg.py:16-41): nn.Sequential indices of the convs / BNs
    const char* convs[7] = {"0", "3", "6", "8", "11", "14", "18"};
    const char* bns[7] = {"", "", "", "", "12", "15", ""};
    for (int i = 0; i < 7; ++i)
      if ((rc = pack_conv(c, bb + convs[i], bns[i][0] ? bb + bns[i] : std::string(), &c->vgg[i], s))) return rc;
    if (c->vgg[0].Cin != 1 || c->vgg[0].KH != 3 || c->vgg[6].KH != 2 || c->vgg[6].Cout != 512)
      return fail(c, D2T_EINVAL, "unexpected VGG_FeatureExtractor shapes");
  }
  if (!is_vgg) {
  if ((rc = pack_conv(c, bb + "conv0_1", bb + "bn0_1", &c->stem, s))) return rc;
  if (c->stem.Cin != 1 || c->stem.KH != 3 || c->stem.KW != 3)
    return fail(c, D2T_EINVAL, "conv0_1 must be 1-channel 3x3");
  if ((rc = pack_conv(c, bb + "conv0_2", bb + "bn0_2", &c->conv0_2, s))) return rc;
  for (int li = 0; li < 4; ++li) {
    for (int bi = 0; bi < RESNET_LAYERS[li]; ++bi) {
      const std::string p = bb + "layer" + std::to_string(li + 1) + "." + std::to_string(bi);
      Block b;
      if ((rc = pack_conv(c, p + ".conv1", p + ".bn1", &b.c1, s))) return rc;
      if ((rc = pack_conv(c, p + ".conv2", p + ".bn2", &b.c2, s))) return rc;
      if (find(c, p + ".downsample.0.weight")) {
        b.has_down = true;
        if ((rc = pack_conv(c, p + ".downsample.0", p + ".downsample.1", &b.down, s))) return rc;
        if (b.c2.w_hi && b.down.w_hi && b.down.KH == 1 && b.down.KW == 1 && b.down.Cout == b.c2.Cout) {
          // conv2 | shortcut concatenated along K (both already folded and in the kernels' K order: a 1x1 layer's is plain)
          const int Co = b.c2.Cout, K2 = b.c2.KH * b.c2.KW * b.c2.Cin, Kd = b.down.Cin;
          void* bufs[4] = {nullptr, nullptr, nullptr, nullptr};
          const size_t n = (size_t)Co * (K2 + Kd);
          const size_t bytes[4] = {n * 4, (size_t)Co * 4, n * 2, n * 2};
          for (int q = 0; q < 4; ++q) {  // (each buffer is owned as soon as it exists: nothing leaks when a later one fails)
            if ((rc = dev_alloc(c, &bufs[q], bytes[q]))) return rc;
            c->owned.push_back(bufs[q]);
          }
          void *pw = bufs[0], *pb = bufs[1], *ph = bufs[2], *pl = bufs[3];
          b.c2cat = b.c2;
          b.c2cat.w = (float*)pw; b.c2cat.bias = (float*)pb; b.c2cat.w_hi = (uint16_t*)ph; b.c2cat.w_lo = (uint16_t*)pl;
          b.c2cat.w_h16 = b.c2cat.w_l16 = nullptr; b.c2cat.K2 = Kd;  // (K2: the K rows of the second, 1x1 input)
          HIPCHK(c, hipMemcpy2DAsync(pw, (size_t)(K2 + Kd) * 4, b.c2.w, (size_t)K2 * 4, (size_t)K2 * 4, Co, hipMemcpyDeviceToDevice, s));
          HIPCHK(c, hipMemcpy2DAsync((float*)pw + K2, (size_t)(K2 + Kd) * 4, b.down.w, (size_t)Kd * 4, (size_t)Kd * 4, Co,
                                     hipMemcpyDeviceToDevice, s));
          HIPCHK(c, launch_add_rows(b.c2.bias, b.down.bias, b.c2cat.bias, Co, s));
          HIPCHK(c, launch_split_bf16(b.c2cat.w, b.c2cat.w_hi, b.c2cat.w_lo, n, s));
        } else {
          b.c2cat.w = nullptr;
        }
      }
      c->layers[li].push_back(b);
    }
    if (c->cfg.gcb) {  // GlobalContext(planes) appended to the stage (visual_attention.py:105-165)
      const std::string p = bb + "layer" + std::to_string(li + 1) + "." + std::to_string(RESNET_LAYERS[li]) + ".";
      const int C = c->layers[li].back().c2.Cout;
      const RawW *wg, *bg, *w1, *b1, *lg, *lb, *w2, *b2;
      if ((rc = need(c, p + "global_cxt.weight", &wg, {1, C, 1, 1})) || (rc = need(c, p + "global_cxt.bias", &bg, {1})) ||
          (rc = need(c, p + "bottleneck_add.fc1.weight", &w1, {C, C, 1, 1})) ||
          (rc = need(c, p + "bottleneck_add.fc1.bias", &b1, {C})) ||
          (rc = need(c, p + "bottleneck_add.norm.weight", &lg, {C})) || (rc = need(c, p + "bottleneck_add.norm.bias", &lb, {C})) ||
          (rc = need(c, p + "bottleneck_add.fc2.weight", &w2, {C, C, 1, 1})) ||
          (rc = need(c, p + "bottleneck_add.fc2.bias", &b2, {C})))
        return rc;
      c->gc[li] = GCParams{wg->p, bg->p, w1->p, b1->p, lg->p, lb->p, w2->p, b2->p};
    }
  }
  if ((rc = pack_conv(c, bb + "conv1", bb + "bn1", &c->conv1, s))) return rc;
  if ((rc = pack_conv(c, bb + "conv2", bb + "bn2", &c->conv2, s))) return rc;
  if ((rc = pack_conv(c, bb + "conv3", bb + "bn3", &c->conv3, s))) return rc;
  if ((rc = pack_conv(c, bb + "conv4_1", bb + "bn4_1", &c->conv4_1, s))) return rc;
  if ((rc = pack_conv(c, bb + "conv4_2", bb + "bn4_2", &c->conv4_2, s))) return rc;
  }  // !is_vgg

  if (has_lstm) {
    // 2x BidirectionalLSTM (seq_modeling/bilstm.py:6-24, build_seq.py:20-23): nn.LSTM(bidirectional) + Linear
    const int Hh = g.bilstm_hidden, G4 = 4 * Hh;
    int in = 512;
    for (int i = 0; i < 2; ++i) {
      const std::string lp = sm + std::to_string(i) + ".";
      BiLstmW& L = c->lstm[i];
      L.in = in;
      void *a, *b2, *t;
      if ((rc = dev_alloc(c, &a, (size_t)2 * G4 * in * 4)) || (rc = dev_alloc(c, &b2, (size_t)2 * G4 * 4)) ||
          (rc = dev_alloc(c, &t, (size_t)2 * Hh * G4 * 4)))
        return rc;
      c->owned.push_back(a); c->owned.push_back(b2); c->owned.push_back(t);
      L.wih_cat = (float*)a; L.bias_cat = (float*)b2; L.whh_t = (float*)t;
      const char* sfx[2] = {"", "_reverse"};
      for (int dirn = 0; dirn < 2; ++dirn) {
        const RawW *wi, *wh, *bi, *bh;
        if ((rc = need(c, lp + "rnn.weight_ih_l0" + sfx[dirn], &wi, {G4, in})) ||
            (rc = need(c, lp + "rnn.weight_hh_l0" + sfx[dirn], &wh, {G4, Hh})) ||
            (rc = need(c, lp + "rnn.bias_ih_l0" + sfx[dirn], &bi, {G4})) ||
            (rc = need(c, lp + "rnn.bias_hh_l0" + sfx[dirn], &bh, {G4})))
          return rc;
        HIPCHK(c, launch_copy(wi->p, L.wih_cat + (size_t)dirn * G4 * in, (size_t)G4 * in, s));
        HIPCHK(c, launch_add_rows(bi->p, bh->p, L.bias_cat + (size_t)dirn * G4, G4, s));
        HIPCHK(c, launch_transpose_into(wh->p, G4, Hh, L.whh_t + (size_t)dirn * Hh * G4, G4, 0, s));
      }
      if ((rc = get_lin(c, lp + "linear", &L.lin, Hh, 2 * Hh))) return rc;
      in = Hh;
    }
  }

  if (vit) {
    const int D = g.vit_dim;
    if ((rc = pack_conv(c, sm + "patch_embed.proj", "", &c->patch, s))) return rc;
    if (c->patch.Cout != D || c->patch.KH != g.patch_h || c->patch.KW != g.patch_w)
      return fail(c, D2T_EINVAL, "patch_embed.proj shape does not match the config");
    const RawW *pos, *cls;
    if ((rc = need(c, sm + "pos_embed", &pos)) || (rc = need(c, sm + "cls_token", &cls, {1, 1, D}))) return rc;
    if (pos->shape.size() != 3 || pos->shape[2] != D) return fail(c, D2T_EINVAL, "pos_embed must be [1,N,%d]", D);
    c->pos_embed = pos->p;
    c->pos_rows = (int)pos->shape[1];
    if (g.vit_pos < D2T_VIT_POS_SINCOS_PREFIX || g.vit_pos > D2T_VIT_POS_LEARNED_PREFIX) return fail(c, D2T_EINVAL, "vit_pos %d", g.vit_pos);
    if (d2t_encoder_shape(c, g.max_h, g.max_w, nullptr, nullptr, &c->pos_GH, &c->pos_GW, nullptr, nullptr))
      return fail(c, D2T_EINVAL, "max_dimension %dx%d leaves no backbone output", g.max_h, g.max_w);
    if (c->pos_rows != c->pos_GH * c->pos_GW + 1)
      return fail(c, D2T_EINVAL, "pos_embed has %d rows, max_dimension %dx%d gives a %dx%d patch grid", c->pos_rows, g.max_h,
                  g.max_w, c->pos_GH, c->pos_GW);
    void* p;
    if ((rc = dev_alloc(c, &p, (size_t)D * 4))) return rc;
    c->owned.push_back(p);
    c->cls_row = (float*)p;
    HIPCHK(c, launch_add_rows(cls->p, pos->p, c->cls_row, D, s));
    for (int i = 0; i < g.vit_depth; ++i) {
      const std::string b = sm + "blocks." + std::to_string(i) + ".";
      VitBlock vb;
      if ((rc = get_ln(c, b + "norm1", &vb.n1, D)) || (rc = get_ln(c, b + "norm2", &vb.n2, D)) ||
          (rc = get_lin(c, b + "attn.qkv", &vb.qkv, 3 * D, D)) || (rc = get_lin(c, b + "attn.proj", &vb.proj, D, D)))
        return rc;
      const RawW* f1;
      if ((rc = need(c, b + "mlp.fc1.weight", &f1))) return rc;
      const int hid = (int)f1->shape[0];
      if ((rc = get_lin(c, b + "mlp.fc1", &vb.fc1, hid, D)) || (rc = get_lin(c, b + "mlp.fc2", &vb.fc2, D, hid)))
        return rc;
      if ((rc = split_lin(c, &vb.qkv, s)) || (rc = split_lin(c, &vb.proj, s)) || (rc = split_lin(c, &vb.fc1, s)) ||
          (rc = split_lin(c, &vb.fc2, s)))
        return rc;
      c->vit.push_back(vb);
    }
    if ((rc = get_ln(c, sm + "norm", &c->vit_norm, D))) return rc;
    if (g.decoder == D2T_DEC_TFM && g.dec_dim != D)
      return fail(c, D2T_EINVAL, "decoder d_model must equal the ViT hidden_size");
  } else if (g.decoder == D2T_DEC_TFM && g.dec_dim != c->conv4_2.Cout) {
    return fail(c, D2T_EINVAL, "decoder d_model must equal the backbone output_channel");
  }

  const std::string pp = "predicter.Prediction.";
  if (g.decoder == D2T_DEC_ATTN) {
    // Attention.__init__ (prediction_head/seq2seq.py:11-68) + LocationAwareAttention (attention1D.py:121-133,203-214)
    const int Hh = g.attn_hidden, V = g.vocab, taps = 2 * g.attn_kernel_size + 1, kd = g.attn_kernel_dim;
    const std::string ac = pp + "attention_cell.";
    AttnW& A = c->attn;
    A = AttnW{};
    A.taps = taps;
    const RawW *emb = nullptr, *lcw = nullptr, *lcb = nullptr, *lpw = nullptr, *lpb = nullptr, *qw, *qb, *sw, *sb = nullptr,
               *wih, *whh, *bih, *bhh, *gw, *gb;
    const bool bahdanau = g.attn_cell == D2T_ATTN_CELL_BAHDANAU, onehot = g.attn_onehot != 0;
    const int Ein = onehot ? V : Hh;  // width of the decoder-input part of rnn.weight_ih
    if (!onehot && (rc = need(c, pp + "embedding.weight", &emb, {V, Hh}))) return rc;
    if (bahdanau) {  // BahdanauAttentionCell (attention1D.py:71-85): i2h without bias, h2h, score without bias
      if ((rc = need(c, ac + "attn.h2h.weight", &qw, {Hh, Hh})) || (rc = need(c, ac + "attn.h2h.bias", &qb, {Hh})) ||
          (rc = get_lin(c, ac + "attn.i2h", &A.key, Hh, Hh, /*bias=*/false)) ||
          (rc = need(c, ac + "attn.score.weight", &sw, {1, Hh})))
        return rc;
    } else if ((rc = need(c, ac + "attn.loc_conv.weight", &lcw, {kd, 1, taps})) ||
               (rc = need(c, ac + "attn.loc_conv.bias", &lcb, {kd})) ||
               (rc = need(c, ac + "attn.loc_proj.weight", &lpw, {Hh, kd})) ||
               (rc = need(c, ac + "attn.loc_proj.bias", &lpb, {Hh})) ||
               (rc = need(c, ac + "attn.query_proj.weight", &qw, {Hh, Hh})) ||
               (rc = need(c, ac + "attn.query_proj.bias", &qb, {Hh})) ||
               (rc = get_lin(c, ac + "attn.key_proj", &A.key, Hh, Hh)) ||
               (rc = need(c, ac + "attn.score.weight", &sw, {1, Hh})) || (rc = need(c, ac + "attn.score.bias", &sb, {1}))) {
      return rc;
    }
    if ((rc = need(c, ac + "rnn.weight_ih", &wih, {4 * Hh, Hh + Ein})) ||
        (rc = need(c, ac + "rnn.weight_hh", &whh, {4 * Hh, Hh})) || (rc = need(c, ac + "rnn.bias_ih", &bih, {4 * Hh})) ||
        (rc = need(c, ac + "rnn.bias_hh", &bhh, {4 * Hh})) || (rc = need(c, ac + "generator.weight", &gw, {V, Hh})) ||
        (rc = need(c, ac + "generator.bias", &gb, {V})))
      return rc;
    A.emb = emb ? emb->p : nullptr; A.bq = qb->p; A.wscore = sw->p; A.bg = gb->p;
    auto alloc = [&](float** dst, size_t n) -> int {
      void* q;
      int r2 = dev_alloc(c, &q, n * 4);
      if (r2) return r2;
      c->owned.push_back(q);
      *dst = (float*)q;
      return D2T_OK;
    };
    if ((rc = alloc(&A.wq_t, (size_t)Hh * Hh)) || (rc = alloc(&A.wloc, (size_t)Hh * taps)) ||
        (rc = alloc(&A.bloc, Hh)) || (rc = alloc(&A.wx_t, (size_t)3 * Hh * 4 * Hh)) || (rc = alloc(&A.bx, 4 * Hh)) ||
        (rc = alloc(&A.wg_t, (size_t)Hh * V)))
      return rc;
    HIPCHK(c, launch_transpose_into(qw->p, Hh, Hh, A.wq_t, Hh, 0, s));
    if (!onehot) {
      HIPCHK(c, launch_transpose_into(wih->p, 4 * Hh, 2 * Hh, A.wx_t, 4 * Hh, 0, s));       // rows [ctx ; emb]
    } else {
      // one-hot decoder input (seq2seq.py:72-78): W_ih . [ctx ; onehot(tok)] = W_ih[:, :H] . ctx + W_ih[:, H + tok]; the
      // kernel keeps its [ctx ; emb ; h] layout with zero "emb" rows and adds row `tok` of the transposed tail of W_ih
      float* full;  // W_ih^T [H + V][4H]
      if ((rc = alloc(&full, (size_t)(Hh + V) * 4 * Hh))) return rc;
      HIPCHK(c, launch_transpose_into(wih->p, 4 * Hh, Hh + V, full, 4 * Hh, 0, s));
      HIPCHK(c, hipMemcpyAsync(A.wx_t, full, (size_t)Hh * 4 * Hh * 4, hipMemcpyDeviceToDevice, s));
      HIPCHK(c, hipMemsetAsync(A.wx_t + (size_t)Hh * 4 * Hh, 0, (size_t)Hh * 4 * Hh * 4, s));
      A.tokgate = full + (size_t)Hh * 4 * Hh;
    }
    HIPCHK(c, launch_transpose_into(whh->p, 4 * Hh, Hh, A.wx_t, 4 * Hh, 2 * Hh, s));      // rows h
    HIPCHK(c, launch_add_rows(bih->p, bhh->p, A.bx, 4 * Hh, s));
    HIPCHK(c, launch_transpose_into(gw->p, V, Hh, A.wg_t, V, 0, s));
    if (bahdanau) {  // no location term, no score bias: a zero one-tap filter
      HIPCHK(c, hipMemsetAsync(A.wloc, 0, (size_t)Hh * taps * 4, s));
      HIPCHK(c, hipMemsetAsync(A.bloc, 0, (size_t)Hh * 4, s));
      A.bscore = 0.f;
    } else {  // fold loc_proj o loc_conv (attention1D.py:150-152) into one [H][taps] filter on the host
      std::vector<float> hcw((size_t)kd * taps), hcb(kd), hpw((size_t)Hh * kd), hpb(Hh), hsb(1);
      HIPCHK(c, hipStreamSynchronize(s));
      HIPCHK(c, hipMemcpy(hcw.data(), lcw->p, hcw.size() * 4, hipMemcpyDeviceToHost));
      HIPCHK(c, hipMemcpy(hcb.data(), lcb->p, hcb.size() * 4, hipMemcpyDeviceToHost));
      HIPCHK(c, hipMemcpy(hpw.data(), lpw->p, hpw.size() * 4, hipMemcpyDeviceToHost));
      HIPCHK(c, hipMemcpy(hpb.data(), lpb->p, hpb.size() * 4, hipMemcpyDeviceToHost));
      HIPCHK(c, hipMemcpy(hsb.data(), sb->p, 4, hipMemcpyDeviceToHost));
      std::vector<float> w((size_t)Hh * taps), b(Hh);
      for (int n = 0; n < Hh; ++n) {
        double bb2 = hpb[n];
        for (int m = 0; m < kd; ++m) bb2 += (double)hpw[(size_t)n * kd + m] * hcb[m];
        b[n] = (float)bb2;
        for (int j = 0; j < taps; ++j) {
          double a = 0.0;
          for (int m = 0; m < kd; ++m) a += (double)hpw[(size_t)n * kd + m] * hcw[(size_t)m * taps + j];
          w[(size_t)n * taps + j] = (float)a;
        }
      }
      HIPCHK(c, hipMemcpy(A.wloc, w.data(), w.size() * 4, hipMemcpyHostToDevice));
      HIPCHK(c, hipMemcpy(A.bloc, b.data(), b.size() * 4, hipMemcpyHostToDevice));
      A.bscore = hsb[0];
    }
    if (g.attn_enc_init) {
      const RawW *hw, *hb, *cw, *cb;
      if ((rc = need(c, pp + "proj_init_h.weight", &hw, {Hh, Hh})) || (rc = need(c, pp + "proj_init_h.bias", &hb, {Hh})) ||
          (rc = need(c, pp + "proj_init_c.weight", &cw, {Hh, Hh})) || (rc = need(c, pp + "proj_init_c.bias", &cb, {Hh})))
        return rc;
      if ((rc = alloc(&A.wih_t, (size_t)Hh * Hh)) || (rc = alloc(&A.wic_t, (size_t)Hh * Hh))) return rc;
      HIPCHK(c, launch_transpose_into(hw->p, Hh, Hh, A.wih_t, Hh, 0, s));
      HIPCHK(c, launch_transpose_into(cw->p, Hh, Hh, A.wic_t, Hh, 0, s));
      A.bih = hb->p; A.bic = cb->p;
    }
    HIPCHK(c, hipStreamSynchronize(s));
    c->finalized = true;
    return D2T_OK;
  }

  // decoder (prediction_head/tfm.py:36-72)
  const int d = g.dec_dim, V = g.vocab;
  const RawW *we, *pe;
  if ((rc = need(c, pp + "word_embed.weight", &we, {V, d})) || (rc = need(c, pp + "pos_enc.pe", &pe))) return rc;
  if (pe->shape.size() != 2 || pe->shape[1] != d || pe->shape[0] < g.max_seq_len + 2)
    return fail(c, D2T_EINVAL, "pos_enc.pe must be [>=%d,%d]", g.max_seq_len + 2, d);
  c->word_embed = we->p;
  c->word_pe = pe->p;
  c->word_pe_rows = (int)pe->shape[0];
  void *kw, *kb;
  if ((rc = dev_alloc(c, &kw, (size_t)g.dec_layers * 2 * d * d * 4)) ||
      (rc = dev_alloc(c, &kb, (size_t)g.dec_layers * 2 * d * 4)))
    return rc;
  c->owned.push_back(kw);
  c->owned.push_back(kb);
  c->ckv_w = (float*)kw;
  c->ckv_b = (float*)kb;
  for (int i = 0; i < g.dec_layers; ++i) {
    const std::string l = pp + "model.layers." + std::to_string(i) + ".";
    DecLayer dl;
    const RawW *siw, *sib, *ciw, *cib;
    if ((rc = need(c, l + "self_attn.in_proj_weight", &siw, {3 * d, d})) ||
        (rc = need(c, l + "self_attn.in_proj_bias", &sib, {3 * d})) ||
        (rc = need(c, l + "multihead_attn.in_proj_weight", &ciw, {3 * d, d})) ||
        (rc = need(c, l + "multihead_attn.in_proj_bias", &cib, {3 * d})))
      return rc;
    dl.sa_in = LinW{siw->p, sib->p, 3 * d, d};
    dl.ca_q = LinW{ciw->p, cib->p, d, d};
    HIPCHK(c, launch_copy(ciw->p + (size_t)d * d, c->ckv_w + (size_t)i * 2 * d * d, (size_t)2 * d * d, s));
    HIPCHK(c, launch_copy(cib->p + d, c->ckv_b + (size_t)i * 2 * d, (size_t)2 * d, s));
    if ((rc = get_lin(c, l + "self_attn.out_proj", &dl.sa_out, d, d)) ||
        (rc = get_lin(c, l + "multihead_attn.out_proj", &dl.ca_out, d, d)) ||
        (rc = get_lin(c, l + "linear1", &dl.l1, g.dec_ff, d)) || (rc = get_lin(c, l + "linear2", &dl.l2, d, g.dec_ff)) ||
        (rc = get_ln(c, l + "norm1", &dl.n1, d)) || (rc = get_ln(c, l + "norm2", &dl.n2, d)) ||
        (rc = get_ln(c, l + "norm3", &dl.n3, d)))
      return rc;
    for (auto pr : {std::make_pair(&dl.sa_out_t, dl.sa_out.w), std::make_pair(&dl.ca_q_t, dl.ca_q.w),
                    std::make_pair(&dl.ca_out_t, dl.ca_out.w)}) {
      void* tp;
      if ((rc = dev_alloc(c, &tp, (size_t)d * d * 4))) return rc;
      c->owned.push_back(tp);
      *pr.first = (float*)tp;
      HIPCHK(c, launch_transpose(pr.second, *pr.first, d, d, s));
    }
    {  // absorbed cross-attention: W_k as stored, W_v transposed (read from the engine's stacked copy), b_v
      void* tp;
      if ((rc = dev_alloc(c, &tp, (size_t)d * d * 4))) return rc;
      c->owned.push_back(tp);
      dl.ca_v_t = (float*)tp;
      dl.ca_wk = c->ckv_w + (size_t)i * 2 * d * d;
      dl.ca_bv = c->ckv_b + (size_t)i * 2 * d + d;
      HIPCHK(c, launch_transpose(c->ckv_w + (size_t)i * 2 * d * d + (size_t)d * d, dl.ca_v_t, d, d, s));
    }
    c->dec.push_back(dl);
  }
  if ((rc = get_lin(c, pp + "proj", &c->out_proj, V, d))) return rc;
  {  // bf16 split of the stacked cross-attention K/V projection (one large-M GEMM per batch)
    const uint16_t *hi = nullptr, *lo = nullptr;
    if (d % 32 == 0 && (rc = split_planes(c, c->ckv_w, (size_t)g.dec_layers * 2 * d * d, &hi, &lo, s))) return rc;
    c->ckv_hi = const_cast<uint16_t*>(hi);
    c->ckv_lo = const_cast<uint16_t*>(lo);
  }
  c->dec_absorbed = d == 256 && g.dec_heads == 8 && !D2T_PROBE_ENV_STR("D2T_DECODE_PROJECTED_KV");
  HIPCHK(c, hipStreamSynchronize(s));
  c->finalized = true;
  return D2T_OK;
}

int d2t_encoder_shape(const d2t_ctx* c, int32_t H, int32_t W, int32_t* T, int32_t* d, int32_t* grid_h,
                      int32_t* grid_w, int32_t* pad_w, int32_t* pad_h) {
  if (!c) return D2T_EINVAL;
  if (H < 4 || W < 4) return D2T_EINVAL;
  int fh, fw;
  if (c->cfg.encoder == D2T_ENC_VGG_BILSTM) {  // vgg.py:16-41: pools (2,2),(2,2),(2,1),(2,1), final 2x2 conv
    fh = H / 2 / 2 / 2 / 2 - 1;
    fw = W / 2 / 2 - 1;
  } else {
    backbone_hw(H, W, &fh, &fw);
  }
  if (fh < 1 || fw < 1) return D2T_EINVAL;
  int gh = fh, gw = fw, pw = 0, ph = 0, t, dim = c->cfg.backbone_out;
  if (c->cfg.encoder == D2T_ENC_VGG_BILSTM || c->cfg.encoder == D2T_ENC_RESNET_BILSTM) {
    t = fw;  // the height is averaged away (build_feat.py:50-55)
    dim = c->cfg.bilstm_hidden;
  } else if (c->cfg.encoder == D2T_ENC_HYBRID_VIT) {
    ph = (c->cfg.patch_h - fh % c->cfg.patch_h) % c->cfg.patch_h;
    pw = (c->cfg.patch_w - fw % c->cfg.patch_w) % c->cfg.patch_w;
    gh = (fh + ph) / c->cfg.patch_h;
    gw = (fw + pw) / c->cfg.patch_w;
    t = gh * gw + 1;
    dim = c->cfg.vit_dim;
  } else {
    t = fh * fw;
  }
  if (T) *T = t;
  if (d) *d = dim;
  if (grid_h) *grid_h = gh;
  if (grid_w) *grid_w = gw;
  if (pad_w) *pad_w = pw;
  if (pad_h) *pad_h = ph;
  return D2T_OK;
}

// Longest encoder memory a decode can attend over: 4096 tokens (the shipped max_dimension [800, 800] gives 2526).  The TFM row
// kernels walk the keys with a running softmax -- the absorbed form (d_model 256) in 16-key tiles, the projected-K/V form
// (d_model 512) in groups per lane -- and the LSTM-attention decode kernel keeps two alignment rows of that length in LDS
// (recurrent.hip AD_MAXT).
static int memory_cap(const d2t_ctx*) { return 4096; }

int d2t_encode(d2t_ctx* c, const float* image, int32_t B, int32_t H, int32_t W, float* memory, d2t_stream stream) {
  DevGuard dg_(c);
  if (!c || !image || !memory || B < 1) return fail(c, D2T_EINVAL, "bad argument");
  if (!c->finalized) return fail(c, D2T_ESTATE, "weights not finalized");
  if (int rc = check_dev_ptr(c, image, "image")) return rc;
  if (int rc = check_dev_ptr(c, memory, "memory")) return rc;
  hipStream_t s = (hipStream_t)stream;
  const d2t_config& g = c->cfg;
  int T, dim, gh, gw, pw, ph;
  if (d2t_encoder_shape(c, H, W, &T, &dim, &gh, &gw, &pw, &ph))
    return fail(c, D2T_EINVAL, "unsupported crop %dx%d (backbone output would be empty)", H, W);
  const bool vit = g.encoder == D2T_ENC_HYBRID_VIT;
  if (vit && g.vit_pos != D2T_VIT_POS_LEARNED_INTERP && T > c->pos_rows)  // the prefix slice would run off the table
    return fail(c, D2T_EINVAL, "crop %dx%d exceeds max_dimension (pos_embed rows %d)", H, W, c->pos_rows);
  if (T > memory_cap(c)) return fail(c, D2T_EINVAL, "memory length %d > %d unsupported", T, memory_cap(c));
  // largest activation: conv0_2 output B*H*W*64 floats; ViT needs B*T*max(3*dim, hidden)
  size_t need_floats = (size_t)B * H * W * 64;
  if (vit) {
    size_t hid = c->vit.empty() ? 0 : (size_t)c->vit[0].fc1.N;
    size_t v = (size_t)B * T * (hid > 3 * (size_t)dim ? hid : 3 * (size_t)dim);
    if (v > need_floats) need_floats = v;
  }
  const bool lstm_enc = g.encoder == D2T_ENC_VGG_BILSTM || g.encoder == D2T_ENC_RESNET_BILSTM;
  if (lstm_enc && (size_t)B * T * 8 * g.bilstm_hidden > need_floats) need_floats = (size_t)B * T * 8 * g.bilstm_hidden;
  if (c->act_cap < need_floats * 4) {
    hipDeviceSynchronize();
    for (int i = 0; i < 4; ++i) {
      if (c->act[i]) hipFree(c->act[i]);
      c->act[i] = nullptr;
      void* p;
      int rc = dev_alloc(c, &p, need_floats * 4);
      if (rc) { c->act_cap = 0; return rc; }
      c->act[i] = (float*)p;
    }
    c->act_cap = need_floats * 4;
  }
  Act f{};
  int rc;
  if (lstm_enc) {
    hipError_t err = hipSuccess;
    if (g.encoder == D2T_ENC_VGG_BILSTM) {
      // VGG_FeatureExtractor.forward (feature_extractor/vgg.py:16-44)
      Act x{pick(c, {}), B, H, W, c->vgg[0].Cout};
      HIPCHK(c, launch_stem(image, c->vgg[0].w, c->vgg[0].bias, x.p, B, H, W, x.C, ACT_RELU, s));
      auto pool = [&](const Act& a, int kh, int kw) {
        Act y{pick(c, {a.p}), a.B, (a.H - kh) / kh + 1, (a.W - kw) / kw + 1, a.C};
        hipError_t e = launch_maxpool_k(a.p, y.p, a.B, a.H, a.W, a.C, kh, kw, kh, kw, 0, 0, s);
        if (e != hipSuccess && err == hipSuccess) err = e;
        return y;
      };
      x = pool(x, 2, 2);
      x = conv(c, s, &err, x, c->vgg[1], 1, 1, 1, 1, ACT_RELU, nullptr, pick(c, {x.p}));
      x = pool(x, 2, 2);
      x = conv(c, s, &err, x, c->vgg[2], 1, 1, 1, 1, ACT_RELU, nullptr, pick(c, {x.p}));
      x = conv(c, s, &err, x, c->vgg[3], 1, 1, 1, 1, ACT_RELU, nullptr, pick(c, {x.p}));
      x = pool(x, 2, 1);
      x = conv(c, s, &err, x, c->vgg[4], 1, 1, 1, 1, ACT_RELU, nullptr, pick(c, {x.p}));
      x = conv(c, s, &err, x, c->vgg[5], 1, 1, 1, 1, ACT_RELU, nullptr, pick(c, {x.p}));
      x = pool(x, 2, 1);
      f = conv(c, s, &err, x, c->vgg[6], 1, 1, 0, 0, ACT_RELU, nullptr, pick(c, {x.p}));
      if (err != hipSuccess) return fail(c, D2T_EHIP, "VGG launch: %s", hipGetErrorString(err));
    } else if ((rc = run_backbone(c, s, image, B, H, W, &f, nullptr, nullptr, false))) {
      return rc;
    }
    if (f.W != T || f.C != 512) return fail(c, D2T_EINVAL, "unexpected feature map %dx%dx%d", f.H, f.W, f.C);
    // AdaptiveAvgPool2d((None,1)) over the height (build_feat.py:50-55), then 2x BidirectionalLSTM
    float* seq = pick(c, {f.p});
    HIPCHK(c, launch_mean_h(f.p, seq, B, f.H, f.W, f.C, s));
    const int Hh = g.bilstm_hidden;
    for (int i = 0; i < 2; ++i) {
      const BiLstmW& L = c->lstm[i];
      float* gates = pick(c, {seq});
      float* rec = pick(c, {seq, gates});
      LinW ih{L.wih_cat, L.bias_cat, 8 * Hh, L.in};
      HIPCHK(c, linear_any(c, s, seq, ih, nullptr, gates, B * T, ACT_NONE));
      HIPCHK(c, launch_bilstm(gates, L.whh_t, rec, B, T, Hh, s));
      float* out = i == 1 ? memory : pick(c, {rec});
      HIPCHK(c, linear_any(c, s, rec, L.lin, nullptr, out, B * T, ACT_NONE));
      seq = out;
    }
    return D2T_OK;
  }
  if (!vit) {
    // Feat=ResNet, Seq=None: PositionalEncoding2D add, [B,C,H,W] -> [B,HW,C] (build_seq.py:69-76)
    int fh, fw;
    backbone_hw(H, W, &fh, &fw);
    const float* pe;
    if ((rc = get_pe2d(c, fh, fw, g.backbone_out, s, &pe))) return rc;
    ConvP ex{};
    ex.row_add = pe; ex.rows_per_img = fh * fw; ex.img_stride = fh * fw; ex.row_off = 0; ex.row_add_off = 0;
    return run_backbone(c, s, image, B, H, W, &f, memory, &ex, false);
  }
  if ((rc = run_backbone(c, s, image, B, H, W, &f, nullptr, nullptr, true))) return rc;
  // HybridEmbed.forward (patchembed.py:115-141): zero-pad right/bottom + Conv2d(k=s=patch) as one
  // implicit GEMM whose out-of-range taps read zero; epilogue adds pos_embed[1+i] (flat prefix
  // slice, vit_encoder.py:260) and leaves row 0 of every image for the cls token.
  hipError_t err = hipSuccess;
  float* X = pick(c, {f.p});
  {
    Act y{X, f.B, gh, gw, dim};
    ConvP p{};
    p.w = c->patch.w; p.bias = c->patch.bias; p.out = X;
    if (c->conv_bf16x3) { p.w_hi = c->patch.w_hi; p.w_lo = c->patch.w_lo; }
    if (f.split) { p.in_hi = f.planes(); p.zero16 = c->zero_page; p.max_blocks = c->conv_max_blocks; } else { p.in = f.p; }
    if (f.split && f.fmt != 0) {  // fp16 records from the backbone
      if ((rc = f16_planes(c, c->patch, s))) return rc;
      p.f16 = 1; p.w_hi = c->patch.w_h16; p.w_lo = c->patch.w_l16;
    }
    p.pipelined = c->conv_pipelined; p.reserved_cus = c->reserved_cus; p.split_tail = !c->decode_in_flight || D2T_PROBE_ENV("D2T_CONV_TAIL_ALWAYS");
    p.B = f.B; p.H = f.H; p.W = f.W; p.Cin = f.C; p.OH = gh; p.OW = gw; p.Cout = dim;
    p.KH = g.patch_h; p.KW = g.patch_w; p.SH = g.patch_h; p.SW = g.patch_w; p.PH = 0; p.PW = 0;
    p.M = B * gh * gw; p.K = p.KH * p.KW * f.C; p.act = ACT_NONE;
    const float* pos;
    if ((rc = pos_table_for(c, gh, gw, s, &pos))) return rc;
    p.row_add = pos; p.rows_per_img = gh * gw; p.img_stride = T; p.row_off = 1; p.row_add_off = 1;
    HIPCHK(c, conv_timed(c, p, s));
    HIPCHK(c, launch_fill_cls(c->cls_row, X, B, (long long)T * dim, dim, s));
  }
  const int M = B * T;
  float* Hn = pick(c, {X});
  float* Q = pick(c, {X, Hn});
  float* X2 = pick(c, {X, Hn, Q});
  for (const VitBlock& vb : c->vit) {
    // Block.forward (vision_transformer.py:119-122), LayerNorm eps 1e-6 (:175)
    HIPCHK(c, launch_layernorm(X, vb.n1.g, vb.n1.b, Hn, M, dim, 1e-6f, s));
    HIPCHK(c, linear_any(c, s, Hn, vb.qkv, nullptr, Q, M, ACT_NONE));
    HIPCHK(c, launch_vit_attention(Q, Hn, B, T, g.vit_heads, s));
    HIPCHK(c, linear_any(c, s, Hn, vb.proj, X, X2, M, ACT_NONE));
    HIPCHK(c, launch_layernorm(X2, vb.n2.g, vb.n2.b, Hn, M, dim, 1e-6f, s));
    HIPCHK(c, linear_any(c, s, Hn, vb.fc1, nullptr, Q, M, ACT_GELU));
    HIPCHK(c, linear_any(c, s, Q, vb.fc2, X2, X, M, ACT_NONE));
  }
  HIPCHK(c, launch_layernorm(X, c->vit_norm.g, c->vit_norm.b, memory, M, dim, 1e-6f, s));
  (void)err;
  return D2T_OK;
}

// ---------------------------------------------------------------------------
// decoder
// ---------------------------------------------------------------------------
namespace {
// decode-group bookkeeping words behind the per-row `ended` flags of the device state array:
// [0, MAXB) per-batch end counters, [MAXB, 2 MAXB) per-batch "steps done", [2 MAXB] batches done, [2 MAXB + 1] stop-at step
constexpr int GRP_MAXB = 64, GRP_WORDS = 2 * GRP_MAXB + 2;

struct DecBufs {
  float *x;    // normalised layer input (x0 embedding, or LN3 of the previous layer)
  float *y1, *x1, *y2, *x2, *y3;  // pre-LayerNorm sums y* and their normalised forms x*
  float *qkv, *q2, *a, *f;
};

// make chain i the active one (c->dstream / skv / dws / dstate)
void select_chain(d2t_ctx* c, int i) {
  if (c->active_chain == i) return;
  c->chains[c->active_chain] = d2t_ctx::Chain{c->dstream, c->skv, c->skv_cap, c->dws, c->dws_cap, c->dstate, c->dstate_cap, c->dout, c->dout_cap};
  const d2t_ctx::Chain& o = c->chains[i];
  c->dstream = o.stream; c->skv = o.skv; c->skv_cap = o.skv_cap; c->dws = o.dws; c->dws_cap = o.dws_cap;
  c->dstate = o.dstate; c->dstate_cap = o.dstate_cap; c->dout = o.out; c->dout_cap = o.out_cap;
  c->active_chain = i;
}

// every decode stream has drained (the active chain's stream lives in c->dstream)
hipError_t sync_chains(d2t_ctx* c) {
  hipError_t e = hipStreamSynchronize(c->dstream);
  for (int i = 0; i < d2t_ctx::MAXC && e == hipSuccess; ++i)
    if (i != c->active_chain && c->chains[i].stream) e = hipStreamSynchronize(c->chains[i].stream);
  return e;
}

int dec_prepare(d2t_ctx* c, int B, int T, DecBufs* bufs) {
  const d2t_config& g = c->cfg;
  const int d = g.dec_dim, Lmax = g.max_seq_len + 2;
  int rc;
  // a slot holds the encoder memory copy [B][T][d] (absorbed cross-attention) or the projected K/V of every layer
  // (absorbed form: the fp32 rows, and behind them the same rows as bf16 hi / lo planes for the greedy two-row kernel)
  const size_t slot_bytes = c->dec_absorbed ? (size_t)B * T * d * 8 + 64 : (size_t)g.dec_layers * 2 * B * T * d * 4;
  for (int i = 0; i < (c->n_chains > 2 ? c->n_chains : 2); ++i)
    if ((rc = ensure(c, &c->ckv2[i], &c->ckv2_cap[i], slot_bytes))) return rc;
  if (!c->ckv) c->ckv = c->ckv2[0];
  if ((rc = ensure(c, &c->skv, &c->skv_cap, (size_t)g.dec_layers * 2 * B * Lmax * d * 4))) return rc;
  const size_t per = (size_t)B * (8 * d + 3 * d + g.dec_ff);
  if ((rc = ensure(c, &c->dws, &c->dws_cap, per * 4))) return rc;
  if ((rc = ensure(c, &c->dstate, &c->dstate_cap, (size_t)(4 + B + GRP_WORDS) * 4))) return rc;
  // beam search, absorbed form: absorbed queries / context rows [B][8][d] and LN1 outputs [B][d] between the row kernel's halves
  if (c->dec_absorbed && (rc = ensure(c, &c->beam_qp, &c->beam_qp_cap, (size_t)B * 9 * d * 4))) return rc;
  float* p = c->dws;
  float** six[] = {&bufs->x, &bufs->y1, &bufs->x1, &bufs->y2, &bufs->x2, &bufs->y3, &bufs->q2, &bufs->a};
  for (float** q : six) { *q = p; p += (size_t)B * d; }
  bufs->qkv = p; p += (size_t)B * 3 * d;
  bufs->f = p;
  return D2T_OK;
}

struct Lin { const float* x; int ldx; const LinW* w; const float* res; float* y; int ldy; int act; };


unsigned long long* trace_slot(d2t_ctx* c) {
  if (!c->dtrace || c->dtrace_next >= d2t_ctx::DTRACE_SLOTS) return nullptr;
  return c->dtrace + 2 * (size_t)(c->dtrace_next++);
}

hipError_t skinny(hipStream_t s, const Lin& l, int M, const LNW* ln = nullptr, float* ln_out = nullptr,
                  const int* step_ptr = nullptr, long long step_stride = 0, unsigned long long* trace = nullptr,
                  const int* stop_at = nullptr, const int* cur_step = nullptr) {
  SkinnyP p{};
  p.trace = trace;
  p.stop_at = stop_at; p.cur_step = cur_step;
  p.x = l.x; p.w = l.w->w; p.bias = l.w->b; p.res = l.res; p.y = l.y;
  p.M = M; p.K = l.w->K; p.N = l.w->N; p.ldx = l.ldx; p.ldy = l.ldy; p.ldres = l.w->N; p.act = l.act;
  p.step_ptr = step_ptr; p.out_step_stride = step_stride;
  if (ln) { p.ln_g = ln->g; p.ln_b = ln->b; p.ln_eps = 1e-5f; p.ln_out = ln_out; }
  return launch_skinny(p, s);
}

// cross-attention K,V of every layer, once per batch: [layers*2][B][heads][T][hd]
hipError_t cross_kv(d2t_ctx* c, hipStream_t s, const float* memory, int B, int T) {
  const d2t_config& g = c->cfg;
  const int d = g.dec_dim;
  // absorbed form: no projection at all -- the step loop reads the memory rows; the slot keeps a copy so that the captured
  // loop holds an engine address and the caller's tensor is free again as soon as this copy has run
  if (c->dec_absorbed) {
    const size_t n = (size_t)B * T * d;
    hipError_t e = hipMemcpyAsync(c->ckv, memory, n * sizeof(float), hipMemcpyDeviceToDevice, s);
    if (e != hipSuccess) return e;
    uint16_t* hi = reinterpret_cast<uint16_t*>(c->ckv + n);
    return launch_split_bf16(memory, hi, hi + n, n, s);  // the planes of the split-bf16 cross-attention (decode.hip)
  }
  ConvP p{};
  p.in = memory; p.w = c->ckv_w; p.bias = c->ckv_b; p.out = c->ckv;
  if (c->conv_bf16x3 && c->ckv_hi) { p.w_hi = c->ckv_hi; p.w_lo = c->ckv_lo; }
  p.B = 1; p.H = 1; p.W = B * T; p.Cin = d; p.OH = 1; p.OW = B * T; p.Cout = g.dec_layers * 2 * d;
  p.KH = p.KW = p.SH = p.SW = 1; p.M = B * T; p.K = d; p.act = ACT_NONE;
  p.store_mode = STORE_KV; p.kv_T = T; p.kv_heads = g.dec_heads; p.kv_hd = d / g.dec_heads; p.kv_B = B;
  return launch_conv(p, s);
}

// One decode step for M rows up to the vocabulary logits (greedy: M = B rows).
// bf.x holds the embedded input of this step.  Post-norm decoder layer
// (nn.TransformerDecoderLayer, norm_first=False): every LayerNorm is evaluated as
// the prologue of the GEMM that consumes it (which also writes the normalised rows
// needed later as the residual), so a layer is 4 launches:
//   [LN3 prev] qkv GEMM | fused row kernel (self-attn, out-proj+res, LN1, q-proj, cross-attn, out-proj+res)
//   | [LN2] ff1+ReLU GEMM | ff2+res GEMM
// All position-dependent values come from the device step counter (graph-replayable).
// shared_mem: cross K/V of sample 0 shared by every row (beam search over one sample).
// beam > 0 (beam search with at most 6 hypotheses per sample, absorbed form): the row work runs as pre / per-SAMPLE cross /
// post (launch_decoder_row_beam) with the samples' row segments in `seg` (nullptr: one sample, rows [0, M)).
hipError_t decode_step(d2t_ctx* c, hipStream_t s, const DecBufs& bf, int M, int T, int kvB, bool shared_mem,
                       float* logits, long long logit_row_stride, long long logit_step_stride, int ckvB = -1,
                       const int* row_map = nullptr, const int* stop = nullptr, int beam = 0, const int* seg = nullptr,
                       const int* anc = nullptr, const int* rows_ptr = nullptr) {
  const d2t_config& g = c->cfg;
  const int d = g.dec_dim, heads = g.dec_heads, hd = d / heads, Lmax = g.max_seq_len + 2;
  const int* step = c->dstate;
  hipError_t e;
#define TRY(x) do { if ((e = (x)) != hipSuccess) return e; } while (0)
  const size_t skv_layer = (size_t)kvB * heads * Lmax * hd;
  // batched beam search: ckvB samples' cross K/V, row b attends over sample row_map[b]
  const size_t ckv_slab = (size_t)(shared_mem ? 1 : (ckvB > 0 ? ckvB : kvB)) * heads * T * hd;
  for (int l = 0; l < g.dec_layers; ++l) {
    const DecLayer& L = c->dec[l];
    if (l == 0) {
      TRY(skinny(s, Lin{bf.x, d, &L.sa_in, nullptr, bf.qkv, 3 * d, ACT_NONE}, M, nullptr, nullptr, nullptr, 0, trace_slot(c), stop, step));
    } else {
      TRY(skinny(s, Lin{bf.y3, d, &L.sa_in, nullptr, bf.qkv, 3 * d, ACT_NONE}, M, &c->dec[l - 1].n3, bf.x, nullptr, 0, trace_slot(c), stop, step));
    }
    DecRowP r{};
    r.qkv = bf.qkv; r.qkv_stride = 3 * d; r.xres = bf.x;
    r.sk = c->skv_cur + (size_t)(2 * l) * skv_layer; r.sv = c->skv_cur + (size_t)(2 * l + 1) * skv_layer;
    r.s_batch_stride = (long long)heads * Lmax * hd; r.s_Lmax = Lmax;
    r.ck = c->ckv + (size_t)(2 * l) * ckv_slab; r.cv = c->ckv + (size_t)(2 * l + 1) * ckv_slab;
    r.c_batch_stride = shared_mem ? 0 : (long long)heads * T * hd;  // beam: every hypothesis reads sample 0
    r.c_row_map = row_map;
    r.T = T;
    r.wo_t = L.sa_out_t; r.bo = L.sa_out.b; r.ln1_g = L.n1.g; r.ln1_b = L.n1.b; r.eps = 1e-5f;
    r.wq_t = L.ca_q_t; r.bq = L.ca_q.b; r.wco_t = L.ca_out_t; r.bco = L.ca_out.b;
    r.y2 = bf.y2; r.step_ptr = step; r.M = M; r.D = d; r.heads = heads;
    r.trace = trace_slot(c);
    r.stop_at = stop;
    r.anc = anc; r.anc_stride = Lmax; r.one_row = beam > 0;
    r.rows_ptr = rows_ptr;
    if (c->dec_absorbed && c->beam_shared_tile && beam > 0 && beam <= 6 && c->beam_qp && (shared_mem || row_map))
      TRY(launch_decoder_row_beam(r, c->ckv, shared_mem ? 0 : (long long)T * d, L.ca_wk, L.ca_v_t, L.ca_bv, c->beam_qp,
                                  c->beam_qp + (size_t)kvB * 8 * d, seg, shared_mem ? 1 : ckvB, s));
    else if (c->dec_absorbed) {
      // the split-bf16 cross-attention reads the planes behind the slot's fp32 rows (cross_kv): greedy rows (two per block) and beam
      // rows (one per block, ancestry) alike
      const size_t memn = (size_t)(shared_mem ? 1 : (ckvB > 0 ? ckvB : kvB)) * T * d;
      const uint16_t* mhi = c->cross_fp32 ? nullptr : reinterpret_cast<const uint16_t*>(c->ckv + memn);
      TRY(launch_decoder_row_absorbed(r, c->ckv, shared_mem ? 0 : (long long)T * d, L.ca_wk, L.ca_v_t, L.ca_bv, s, mhi, mhi ? mhi + memn : nullptr));
    }
    else TRY(launch_decoder_row(r, s));
    TRY(skinny(s, Lin{bf.y2, d, &L.l1, nullptr, bf.f, g.dec_ff, ACT_RELU}, M, &L.n2, bf.x2, nullptr, 0, trace_slot(c), stop, step));
    TRY(skinny(s, Lin{bf.f, g.dec_ff, &L.l2, bf.x2, bf.y3, d, ACT_NONE}, M, nullptr, nullptr, nullptr, 0, trace_slot(c), stop, step));
  }
  TRY(skinny(s, Lin{bf.y3, d, &c->out_proj, nullptr, logits, (int)logit_row_stride, ACT_NONE}, M,
             &c->dec[g.dec_layers - 1].n3, nullptr, step, logit_step_stride, trace_slot(c), stop, step));
#undef TRY
  return hipSuccess;
}
}  // namespace

namespace {
// Greedy decode.  The cross-attention K/V projection runs on the caller's stream into one of two slots;
// the step loop runs on the internal stream, ordered after it.  async != 0: return right after
// enqueueing (always max_seq_len+1 steps); the caller orders later work with d2t_decode_wait.
// rows_per_batch > 0 (async only): the B rows are rows_per_batch-row encoder batches decoded by one loop (a decode group);
// with is_test every batch gets its own "first step at which all ITS rows had ended", and the captured loop stops working
// once every batch has one (device-side early exit: the remaining kernels of the graph return at their first instruction).
int greedy_impl(d2t_ctx* c, const float* memory, int B, int T, const int64_t* start_tokens, int is_test,
                int64_t* tokens, float* logits, int* steps_out, hipStream_t user, bool async, int rows_per_batch = 0) {
  const d2t_config& g = c->cfg;
  const int S = g.max_seq_len + 1, V = g.vocab;
  // memory slots rotate (at least two: the next batch's copy is written while the previous decode still reads its own);
  // async decodes rotate over the chains (chain == slot); everything else runs on chain 0
  const int nslots = c->n_chains > 2 ? c->n_chains : 2;
  const int slot = (int)(c->decode_seq++ % (unsigned)nslots);
  select_chain(c, (async && c->n_chains > 1) ? slot % c->n_chains : 0);
  hipStream_t s = c->dstream;
  DecBufs bf;
  int rc = dec_prepare(c, B, T, &bf);
  if (rc) return rc;
  c->skv_cur = c->skv;
  c->ckv = c->ckv2[slot];
  const bool use_graph = D2T_PROBE_ENV_STR("D2T_NO_GRAPH") == nullptr;
  int64_t* const user_tokens = tokens;
  float* const user_logits = logits;
  const size_t tok_bytes = (size_t)B * S * sizeof(int64_t), log_bytes = (size_t)B * S * V * sizeof(float);
  if (use_graph) {  // engine-owned staging [logits | tokens] on both paths: the graph key holds engine addresses only, so a
                    // caller that allocates fresh output tensors per call (Model.forward does) never forces a re-capture
    if ((rc = ensure(c, &c->dout, &c->dout_cap, log_bytes + tok_bytes))) return rc;
    logits = c->dout;
    tokens = reinterpret_cast<int64_t*>(reinterpret_cast<char*>(c->dout) + log_bytes);
  }
  // the decode that last read this K/V slot must be finished before it is overwritten
  if (c->ev_done_valid[slot]) HIPCHK(c, hipStreamWaitEvent(user, c->ev_done[slot], 0));
  HIPCHK(c, cross_kv(c, user, memory, B, T));
  // order the internal stream after the caller's work (K/V slot, start tokens)
  HIPCHK(c, hipEventRecord(c->ev_in, user));
  HIPCHK(c, hipStreamWaitEvent(s, c->ev_in, 0));
  HIPCHK(c, hipMemsetAsync(c->dstate, 0, (size_t)(4 + B + GRP_WORDS) * 4, s));
  const bool dev_exit = async && is_test;  // early exit decided on the device inside the whole-loop graph
  if (rows_per_batch <= 0 || B % rows_per_batch) rows_per_batch = B;
  const int n_batches = B / rows_per_batch;
  if (n_batches > GRP_MAXB) return fail(c, D2T_EINVAL, "a decode group holds at most %d batches", GRP_MAXB);
  int* grp = c->dstate + 4 + B;
  const int* stop = dev_exit ? grp + 2 * GRP_MAXB + 1 : nullptr;
  // step 0 input: Embedding([GO]) * sqrt(d) + pe[0]; later inputs are written by argmax_embed
  HIPCHK(c, launch_embed(c->word_embed, c->word_pe, start_tokens, tokens, S, c->dstate, bf.x, B, g.dec_dim, s));

  ArgmaxP am{};
  am.logits = logits; am.row_stride = (long long)S * V; am.step_stride = V;
  am.tokens = tokens; am.tok_stride = S;
  am.ended = c->dstate + 4; am.end_count = c->dstate + 1; am.steps_done = c->dstate + 2; am.step_ptr = c->dstate;
  am.B = B; am.V = V; am.end_token = TOK_END;
  am.emb = c->word_embed; am.pe = c->word_pe; am.x = bf.x; am.d = g.dec_dim;
  am.done_count = c->dstate + 3;
  am.rows_per_batch = rows_per_batch; am.n_batches = n_batches;
  am.batch_end_count = grp; am.batch_steps_done = grp + GRP_MAXB; am.batches_done = grp + 2 * GRP_MAXB;
  am.stop_at = dev_exit ? grp + 2 * GRP_MAXB + 1 : nullptr;
  auto one_step = [&](hipStream_t st) -> hipError_t {
    hipError_t e = decode_step(c, st, bf, B, T, B, false, logits, (long long)S * V, V, -1, nullptr, stop);
    if (e != hipSuccess) return e;
    am.trace = trace_slot(c);
    return launch_argmax_embed(am, st);
  };

  // With early exit the host polls between steps, so one captured graph = one step, replayed.  Without it
  // (async, or is_test == 0) the whole max_seq_len+1 step loop is ONE graph: a single launch per batch keeps
  // the host free to enqueue the next batch's encoder while this one decodes.
  const int steps_per_graph = (!is_test || dev_exit) ? S : 1;
  hipGraphExec_t exec = nullptr;
  if (use_graph) {
    d2t_ctx::GraphKey k;
    memset(&k, 0, sizeof k);  // compared with memcmp: the padding must be defined
    k.B = B; k.T = T; k.steps = steps_per_graph; k.tok = tokens; k.logits = logits; k.ckv = c->ckv; k.dws = c->dws;
    k.skv = c->skv; k.dstate = c->dstate;
    k.variant = (dev_exit ? 1 : 0) | ((long long)rows_per_batch << 1);
    for (size_t i = 0; i < c->graphs.size(); ++i)
      if (memcmp(&k, &c->graphs[i].key, sizeof k) == 0) {
        exec = c->graphs[i].exec;
        if (i + 1 != c->graphs.size()) std::swap(c->graphs[i], c->graphs.back());
        break;
      }
    if (!exec) {
      if (D2T_PROBE_ENV_STR("D2T_DECODE_TRACE")) {  // debug timeline: the kernel nodes of THIS captured loop get slots 0 .. n-1
        if (!c->dtrace) HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->dtrace), (size_t)d2t_ctx::DTRACE_SLOTS * 16));
        c->dtrace_next = 0;
      }
      hipGraph_t gr = nullptr;
      HIPCHK(c, hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
      hipError_t e = hipSuccess;
      for (int t = 0; t < steps_per_graph && e == hipSuccess; ++t) e = one_step(s);
      hipError_t e2 = hipStreamEndCapture(s, &gr);
      if (e != hipSuccess || e2 != hipSuccess) {
        if (gr) hipGraphDestroy(gr);
        return fail(c, D2T_EHIP, "decode graph capture: %s", hipGetErrorString(e != hipSuccess ? e : e2));
      }
      e = hipGraphInstantiate(&exec, gr, nullptr, nullptr, 0);
      hipGraphDestroy(gr);
      if (e != hipSuccess) return fail(c, D2T_EHIP, "hipGraphInstantiate: %s", hipGetErrorString(e));
      if (c->graphs.size() >= 40) {  // evict the least recently used; it may still be queued on a decode stream
        HIPCHK(c, sync_chains(c));
        hipGraphExecDestroy(c->graphs.front().exec);
        c->graphs.erase(c->graphs.begin());
      }
      c->graphs.push_back({k, exec});
    }
  }
  int steps = S;
  // device-side early exit: the loop stops writing at the group's stop step, so define everything past it (PAD ids, zero
  // logits) instead of handing the caller whatever an earlier decode left in the staging buffer
  if (dev_exit) {
    HIPCHK(c, hipMemsetAsync(logits, 0, log_bytes, s));
    HIPCHK(c, hipMemsetAsync(tokens, 0, tok_bytes, s));
  }
  d2t_ctx::ProfRec drec{-1, B, S, nullptr, nullptr};  // profiling: the decode loop as ONE record (M = -1, N = rows, K = steps)
  if (c->profiling && hipEventCreate(&drec.a) == hipSuccess && hipEventCreate(&drec.b) == hipSuccess) HIPCHK(c, hipEventRecord(drec.a, s));
  for (int t = 0; t < S; t += (use_graph ? steps_per_graph : 1)) {
    if (use_graph) HIPCHK(c, hipGraphLaunch(exec, s));
    else HIPCHK(c, one_step(s));
    if (!async && is_test && ((t & 7) == 7 || t == S - 1)) {
      HIPCHK(c, hipMemcpyAsync(c->h_pinned, c->dstate + 2, 4, hipMemcpyDeviceToHost, s));
      HIPCHK(c, hipStreamSynchronize(s));
      if (c->h_pinned[0] > 0) { steps = c->h_pinned[0]; break; }
    }
  }
  if (drec.b) {
    HIPCHK(c, hipEventRecord(drec.b, s));
    c->prof.push_back(drec);
  }
  if (tokens != user_tokens) {
    HIPCHK(c, hipMemcpyAsync(user_logits, logits, log_bytes, hipMemcpyDeviceToDevice, s));
    HIPCHK(c, hipMemcpyAsync(user_tokens, tokens, tok_bytes, hipMemcpyDeviceToDevice, s));
  }
  HIPCHK(c, hipEventRecord(c->ev_done[slot], s));
  c->ev_done_valid[slot] = true;
  if (async) {  // serving ticket: this decode's outputs are complete once its event has fired
    const int64_t t = ++c->last_ticket;
    c->decode_in_flight = true;
    {  // per-batch step counts of this decode, readable through d2t_decode_steps once the ticket is complete
      if (!c->h_steps) HIPCHK(c, hipHostMalloc(reinterpret_cast<void**>(&c->h_steps), (size_t)d2t_ctx::TICKET_RING * GRP_MAXB * 4, hipHostMallocDefault));
      int* slot_steps = c->h_steps + (size_t)(t % d2t_ctx::TICKET_RING) * GRP_MAXB;
      c->ticket_batches[t % d2t_ctx::TICKET_RING] = dev_exit ? n_batches : -S;
      if (dev_exit) HIPCHK(c, hipMemcpyAsync(slot_steps, grp + GRP_MAXB, (size_t)n_batches * 4, hipMemcpyDeviceToHost, s));
    }
    hipEvent_t& ev = c->ticket_ev[t % d2t_ctx::TICKET_RING];
    if (!ev) HIPCHK(c, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    HIPCHK(c, hipEventRecord(ev, s));
  }
  if (!async) HIPCHK(c, hipStreamSynchronize(s));
  if (steps_out) *steps_out = steps;
  return D2T_OK;
}
}  // namespace

int d2t_decode_greedy(d2t_ctx* c, const float* memory, int32_t B, int32_t T, const int64_t* start_tokens,
                      int32_t is_test, int64_t* tokens, float* logits, int32_t* steps_out, d2t_stream stream) {
  DevGuard dg_(c);
  if (!c || !memory || !start_tokens || !tokens || !logits || !steps_out || B < 1 || T < 1)
    return fail(c, D2T_EINVAL, "bad argument");
  if (!c->finalized) return fail(c, D2T_ESTATE, "weights not finalized");
  if (T > memory_cap(c)) return fail(c, D2T_EINVAL, "memory length %d > %d unsupported", T, memory_cap(c));
  if (c->cfg.decoder != D2T_DEC_TFM) return fail(c, D2T_ESTATE, "context was not created with the TFM decoder");
  if (int rc = check_dev_ptr(c, memory, "memory")) return rc;
  if (int rc = check_dev_ptr(c, logits, "logits")) return rc;
  return greedy_impl(c, memory, B, T, start_tokens, is_test, tokens, logits, steps_out, (hipStream_t)stream, false);
}

int d2t_decode_attn_greedy(d2t_ctx* c, const float* memory, int32_t B, int32_t T, int32_t is_test, int64_t* tokens,
                           float* probs, int32_t* steps_out, d2t_stream stream) {
  DevGuard dg_(c);
  if (!c || !memory || !tokens || !probs || !steps_out || B < 1 || T < 1) return fail(c, D2T_EINVAL, "bad argument");
  if (!c->finalized) return fail(c, D2T_ESTATE, "weights not finalized");
  const d2t_config& g = c->cfg;
  if (g.decoder != D2T_DEC_ATTN) return fail(c, D2T_ESTATE, "context was not created with the Attn decoder");
  const int Hh = g.attn_hidden, S = g.batch_max_length + 1, V = g.vocab;
  const int key_off = g.attn_keys == D2T_ATTN_KEYS_NOCLS_INIT_CLS ? 1 : 0;
  if (T - key_off < 1 || T - key_off > memory_cap(c)) return fail(c, D2T_EINVAL, "memory length %d unsupported", T);
  hipStream_t s = (hipStream_t)stream;
  int rc;
  // workspace: key_proj(memory) [B*T][H] | end_step [B]
  if ((rc = ensure(c, &c->dws, &c->dws_cap, ((size_t)B * T * Hh + B + 16) * 4))) return rc;
  float* kp = c->dws;
  int* end_step = reinterpret_cast<int*>(c->dws + (size_t)B * T * Hh);
  HIPCHK(c, linear_any(nullptr, s, memory, c->attn.key, nullptr, kp, B * T, ACT_NONE));
  HIPCHK(c, hipMemsetAsync(end_step, 0xFF, (size_t)B * 4, s));  // -1 = never emitted [s]
  AttnDecP p{};
  p.mem = memory; p.T = T; p.D = Hh; p.key_off = key_off;
  p.init_mode = !g.attn_enc_init ? 0 : (g.attn_keys == D2T_ATTN_KEYS_ALL_INIT_MEAN ? 1 : 2);
  p.kp = kp; p.wq_t = c->attn.wq_t; p.bq = c->attn.bq; p.wloc = c->attn.wloc; p.bloc = c->attn.bloc;
  p.taps = c->attn.taps; p.wscore = c->attn.wscore; p.bscore = c->attn.bscore;
  p.wx_t = c->attn.wx_t; p.bx = c->attn.bx; p.wg_t = c->attn.wg_t; p.bg = c->attn.bg;
  p.wih_t = c->attn.wih_t; p.bih = c->attn.bih; p.wic_t = c->attn.wic_t; p.bic = c->attn.bic;
  p.emb = c->attn.emb; p.tokgate = c->attn.tokgate; p.probs = probs; p.tokens = tokens; p.end_step = end_step;
  p.B = B; p.S = S; p.V = V; p.H = Hh; p.E = Hh; p.coverage = g.attn_coverage; p.end_token = 1;  // attn_converter.py:8
  HIPCHK(c, launch_attn_decode(p, s));
  int steps = S;
  if (is_test) {
    // reference: break as soon as every row has emitted [s]; the pre-zeroed probs keep zeros afterwards
    std::vector<int> h(B);
    HIPCHK(c, hipMemcpyAsync(h.data(), end_step, (size_t)B * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    int last = -1;
    bool all = true;
    for (int b = 0; b < B; ++b) { all = all && h[b] >= 0; last = std::max(last, h[b]); }
    if (all && last + 1 < S) {
      steps = last + 1;
      HIPCHK(c, hipMemset2DAsync(probs + (size_t)steps * V, (size_t)S * V * 4, 0, (size_t)(S - steps) * V * 4, B, s));
      HIPCHK(c, hipMemset2DAsync(tokens + steps, (size_t)S * 8, 0, (size_t)(S - steps) * 8, B, s));
    }
  }
  *steps_out = steps;
  return D2T_OK;
}

int d2t_decode_attn_beam(d2t_ctx* c, const float* memory, int32_t T, int32_t beam_size, int64_t* seq_out, int32_t* len_out,
                         float* score_out, d2t_stream stream) {
  DevGuard dg_(c);
  // Attention.forward_beam (prediction_head/seq2seq.py:83-222) / AttentionV2.forward_beam (seq2seq_v2.py:12-174) for
  // one sample: the attention cell + LSTMCell + generator of every live hypothesis run as ONE launch per step (the
  // greedy kernel in step mode, one block per hypothesis, keys shared), log_softmax + flat top-k on the device, the
  // reference's bookkeeping on the host -- including its quirks: step 0 ranks row 0 only; the LSTM state follows
  // prev_word_inds[incomplete] but the coverage memory only `incomplete`; if the last executed step completed nothing
  // the first live sequence is returned; otherwise the best score/len sequence with the MAXIMUM raw score.
  if (!c || !memory || !seq_out || !len_out || !score_out || T < 1) return fail(c, D2T_EINVAL, "bad argument");
  if (!c->finalized) return fail(c, D2T_ESTATE, "weights not finalized");
  const d2t_config& g = c->cfg;
  if (g.decoder != D2T_DEC_ATTN) return fail(c, D2T_ESTATE, "context was not created with the Attn decoder");
  if (!g.attn_coverage && g.attn_cell != D2T_ATTN_CELL_BAHDANAU)
    return fail(c, D2T_ESTATE, "LSTM beam search is implemented for the coverage and Bahdanau cells (the reference's 'loc_aware' beam "
                "hands the previous beam's un-reordered alignment to the next step, seq2seq.py:207)");
  if (beam_size < 1 || beam_size > 16) return fail(c, D2T_EINVAL, "beam_size must be in [1,16]");
  const int Hh = g.attn_hidden, S = g.batch_max_length + 1, V = g.vocab, cap = beam_size;
  const int key_off = g.attn_keys == D2T_ATTN_KEYS_NOCLS_INIT_CLS ? 1 : 0;
  const int Tk = T - key_off;
  if (Tk < 1 || Tk > memory_cap(c)) return fail(c, D2T_EINVAL, "memory length %d unsupported", T);
  hipStream_t s = (hipStream_t)stream;
  int rc;
  if ((rc = ensure(c, &c->dws, &c->dws_cap, ((size_t)T * Hh + 16) * 4))) return rc;
  float* kp = c->dws;
  // workspace (floats): logits [cap][V] | scores | topv | h_in c_in h_out c_out [cap][H] | mem_in mem_out [cap][Tk]
  //                     | tok i64 [cap] | dummy tokens i64 [cap] | topi | idx_h | idx_m | end_step  (ints [cap])
  const size_t nf = (size_t)cap * V + 2 * cap + 4 * (size_t)cap * Hh + 2 * (size_t)cap * Tk;
  const size_t tok_off = (nf + 1) & ~(size_t)1;
  const size_t ws_bytes = tok_off * 4 + 2 * (size_t)cap * 8 + 4 * (size_t)cap * 4 + 64;
  if ((rc = ensure(c, &c->beam_ws, &c->beam_ws_cap, ws_bytes))) return rc;
  float* d_logits = c->beam_ws;
  float* d_scores = d_logits + (size_t)cap * V;
  float* d_topv = d_scores + cap;
  float* st[6];
  st[0] = d_topv + cap;                          // h_in
  st[1] = st[0] + (size_t)cap * Hh;              // c_in
  st[2] = st[1] + (size_t)cap * Hh;              // h_out
  st[3] = st[2] + (size_t)cap * Hh;              // c_out
  st[4] = st[3] + (size_t)cap * Hh;              // mem_in
  st[5] = st[4] + (size_t)cap * Tk;              // mem_out
  int64_t* d_tok = reinterpret_cast<int64_t*>(d_logits + tok_off);
  int64_t* d_dummy = d_tok + cap;
  int* d_topi = reinterpret_cast<int*>(d_dummy + cap);
  int* d_idxh = d_topi + cap;
  int* d_idxm = d_idxh + cap;
  int* d_end = d_idxm + cap;
  char* hp = nullptr;
  const size_t hbytes = (size_t)cap * (8 + 4 * 5) + 64;
  if (hipHostMalloc(reinterpret_cast<void**>(&hp), hbytes, hipHostMallocDefault) != hipSuccess)
    return fail(c, D2T_ENOMEM, "hipHostMalloc failed");
  int64_t* h_tok = reinterpret_cast<int64_t*>(hp);
  float* h_scores = reinterpret_cast<float*>(hp + (size_t)cap * 8);
  float* h_topv = h_scores + cap;
  int* h_topi = reinterpret_cast<int*>(h_topv + cap);
  int* h_idxh = h_topi + cap;
  int* h_idxm = h_idxh + cap;
  auto done = [&](int code) { hipHostFree(hp); return code; };
#define BCHK(expr)                                                                              \
  do {                                                                                          \
    hipError_t e_ = (expr);                                                                     \
    if (e_ != hipSuccess) return done(fail(c, D2T_EHIP, "%s: %s", #expr, hipGetErrorString(e_))); \
  } while (0)
  BCHK(linear_any(nullptr, s, memory, c->attn.key, nullptr, kp, T, ACT_NONE));
  AttnDecP p{};
  p.mem = memory; p.T = T; p.D = Hh; p.key_off = key_off;
  p.init_mode = !g.attn_enc_init ? 0 : (g.attn_keys == D2T_ATTN_KEYS_ALL_INIT_MEAN ? 1 : 2);
  p.kp = kp; p.wq_t = c->attn.wq_t; p.bq = c->attn.bq; p.wloc = c->attn.wloc; p.bloc = c->attn.bloc;
  p.taps = c->attn.taps; p.wscore = c->attn.wscore; p.bscore = c->attn.bscore;
  p.wx_t = c->attn.wx_t; p.bx = c->attn.bx; p.wg_t = c->attn.wg_t; p.bg = c->attn.bg;
  p.wih_t = c->attn.wih_t; p.bih = c->attn.bih; p.wic_t = c->attn.wic_t; p.bic = c->attn.bic;
  p.emb = c->attn.emb; p.tokgate = c->attn.tokgate; p.probs = d_logits; p.tokens = d_dummy; p.end_step = d_end;
  p.S = 1; p.V = V; p.H = Hh; p.E = Hh; p.coverage = 1; p.end_token = 1;
  p.step_mode = 1;
  p.st_h_in = st[0]; p.st_c_in = st[1]; p.st_mem_in = st[4];
  p.st_h_out = st[2]; p.st_c_out = st[3]; p.st_mem_out = st[5];
  p.tok_in = d_tok;

  std::vector<std::vector<int64_t>> seqs((size_t)beam_size, std::vector<int64_t>{0});  // each starts with [GO] = 0
  std::vector<float> live_scores((size_t)beam_size, 0.f);
  std::vector<std::vector<int64_t>> complete;
  std::vector<float> complete_scores;
  int k = beam_size;
  bool last_completed_any = false;
  for (int step = 0; step < S; ++step) {
    const int M = (int)seqs.size();
    for (int i = 0; i < M; ++i) h_scores[i] = live_scores[i];
    BCHK(hipMemcpyAsync(d_scores, h_scores, (size_t)M * 4, hipMemcpyHostToDevice, s));
    p.B = M; p.first = step == 0;
    BCHK(launch_attn_decode(p, s));
    // step 0: the rows are identical and the reference ranks row 0 only (seq2seq.py:145-146)
    BCHK(launch_beam_topk(d_logits, d_scores, step == 0 ? 1 : M, V, k, d_topv, d_topi, s));
    BCHK(hipMemcpyAsync(h_topv, d_topv, (size_t)k * 4, hipMemcpyDeviceToHost, s));
    BCHK(hipMemcpyAsync(h_topi, d_topi, (size_t)k * 4, hipMemcpyDeviceToHost, s));
    BCHK(hipStreamSynchronize(s));
    std::vector<std::vector<int64_t>> nseqs;
    std::vector<float> nscores;
    int ninc = 0;
    last_completed_any = false;
    for (int r = 0; r < k; ++r) {
      const int prev = h_topi[r] / V, word = h_topi[r] % V;
      std::vector<int64_t> sq = seqs[prev];
      sq.push_back(word);
      if (word == 1) {  // [s] (attn_converter.py:8)
        complete.push_back(std::move(sq));
        complete_scores.push_back(h_topv[r]);
        last_completed_any = true;
      } else {
        h_idxh[ninc] = prev;  // LSTM state: hidden[prev_word_inds[incomplete]]
        h_idxm[ninc] = r;     // coverage memory: (alpha_cum + alpha)[incomplete]
        h_tok[ninc] = word;
        nseqs.push_back(std::move(sq));
        nscores.push_back(h_topv[r]);
        ++ninc;
      }
    }
    seqs.swap(nseqs);
    live_scores.swap(nscores);
    k = ninc;
    if (k == 0) break;
    if (step + 1 < S) {
      BCHK(hipMemcpyAsync(d_idxh, h_idxh, (size_t)k * 4, hipMemcpyHostToDevice, s));
      BCHK(hipMemcpyAsync(d_idxm, h_idxm, (size_t)k * 4, hipMemcpyHostToDevice, s));
      BCHK(hipMemcpyAsync(d_tok, h_tok, (size_t)k * 8, hipMemcpyHostToDevice, s));
      BCHK(launch_gather_rows(st[2], st[0], d_idxh, k, Hh, s));
      BCHK(launch_gather_rows(st[3], st[1], d_idxh, k, Hh, s));
      BCHK(launch_gather_rows(st[5], st[4], d_idxm, k, Tk, s));
    }
  }
  BCHK(hipStreamSynchronize(s));
#undef BCHK
  std::vector<int64_t> out;
  float score;
  if (!last_completed_any) {  // seq2seq.py:209-216
    out.assign(seqs[0].begin() + 1, seqs[0].end());
    score = live_scores[0];
  } else {
    size_t best = 0;
    for (size_t i = 1; i < complete.size(); ++i)
      if ((double)complete_scores[i] / (double)complete[i].size() > (double)complete_scores[best] / (double)complete[best].size())
        best = i;
    out.assign(complete[best].begin() + 1, complete[best].end());
    score = *std::max_element(complete_scores.begin(), complete_scores.end());
  }
  const int n = (int)std::min<size_t>(out.size(), (size_t)S);
  for (int i = 0; i < n; ++i) seq_out[i] = out[i];
  *len_out = n;
  *score_out = score;
  return done(D2T_OK);
}

int d2t_decode_attn_beam_batch(d2t_ctx* c, const float* memory, int32_t N, int32_t T, int32_t beam_size, int64_t* seq_out,
                               int32_t* len_out, float* score_out, d2t_stream stream) {
  DevGuard dg_(c);
  // Attention / AttentionV2.forward_beam for N samples in one step loop: rows = live hypotheses of all samples, each
  // attending over its own sample's keys (row map); log_softmax + top-k per sample segment; per-sample bookkeeping
  // exactly as in d2t_decode_attn_beam (whose results this reproduces sample by sample).
  if (!c || !memory || !seq_out || !len_out || !score_out || N < 1 || T < 1) return fail(c, D2T_EINVAL, "bad argument");
  if (!c->finalized) return fail(c, D2T_ESTATE, "weights not finalized");
  const d2t_config& g = c->cfg;
  if (g.decoder != D2T_DEC_ATTN) return fail(c, D2T_ESTATE, "context was not created with the Attn decoder");
  if (!g.attn_coverage && g.attn_cell != D2T_ATTN_CELL_BAHDANAU)
    return fail(c, D2T_ESTATE, "LSTM beam search is implemented for the coverage and Bahdanau cells (the reference's 'loc_aware' beam "
                "hands the previous beam's un-reordered alignment to the next step, seq2seq.py:207)");
  if (beam_size < 1 || beam_size > 16) return fail(c, D2T_EINVAL, "beam_size must be in [1,16]");
  const int Hh = g.attn_hidden, S = g.batch_max_length + 1, V = g.vocab, cap = N * beam_size;
  const int key_off = g.attn_keys == D2T_ATTN_KEYS_NOCLS_INIT_CLS ? 1 : 0;
  const int Tk = T - key_off;
  if (Tk < 1 || Tk > memory_cap(c)) return fail(c, D2T_EINVAL, "memory length %d unsupported", T);
  hipStream_t s = (hipStream_t)stream;
  int rc;
  if ((rc = ensure(c, &c->dws, &c->dws_cap, ((size_t)N * T * Hh + 16) * 4))) return rc;
  float* kp = c->dws;
  const size_t nf = (size_t)cap * V + 2 * (size_t)cap + 4 * (size_t)cap * Hh + 2 * (size_t)cap * Tk;
  const size_t tok_off = (nf + 1) & ~(size_t)1;
  const size_t ws_bytes = tok_off * 4 + 2 * (size_t)cap * 8 + (5 * (size_t)cap + 3 * (size_t)N) * 4 + 64;
  if ((rc = ensure(c, &c->beam_ws, &c->beam_ws_cap, ws_bytes))) return rc;
  float* d_logits = c->beam_ws;
  float* d_scores = d_logits + (size_t)cap * V;
  float* d_topv = d_scores + cap;
  float* st[6];
  st[0] = d_topv + cap;
  st[1] = st[0] + (size_t)cap * Hh;
  st[2] = st[1] + (size_t)cap * Hh;
  st[3] = st[2] + (size_t)cap * Hh;
  st[4] = st[3] + (size_t)cap * Hh;
  st[5] = st[4] + (size_t)cap * Tk;
  int64_t* d_tok = reinterpret_cast<int64_t*>(d_logits + tok_off);
  int64_t* d_dummy = d_tok + cap;
  int* d_topi = reinterpret_cast<int*>(d_dummy + cap);
  int* d_idxh = d_topi + cap;
  int* d_idxm = d_idxh + cap;
  int* d_end = d_idxm + cap;
  int* d_map = d_end + cap;
  int* d_seg = d_map + cap;
  char* hp = nullptr;
  const size_t hbytes = (size_t)cap * (8 + 4 * 6) + (size_t)N * 12 + 64;
  if (hipHostMalloc(reinterpret_cast<void**>(&hp), hbytes, hipHostMallocDefault) != hipSuccess)
    return fail(c, D2T_ENOMEM, "hipHostMalloc failed");
  int64_t* h_tok = reinterpret_cast<int64_t*>(hp);
  float* h_scores = reinterpret_cast<float*>(hp + (size_t)cap * 8);
  float* h_topv = h_scores + cap;
  int* h_topi = reinterpret_cast<int*>(h_topv + cap);
  int* h_idxh = h_topi + cap;
  int* h_idxm = h_idxh + cap;
  int* h_map = h_idxm + cap;
  int* h_seg = h_map + cap;
  auto done = [&](int code) { hipHostFree(hp); return code; };
#define BCHK(expr)                                                                              \
  do {                                                                                          \
    hipError_t e_ = (expr);                                                                     \
    if (e_ != hipSuccess) return done(fail(c, D2T_EHIP, "%s: %s", #expr, hipGetErrorString(e_))); \
  } while (0)
  BCHK(linear_any(nullptr, s, memory, c->attn.key, nullptr, kp, N * T, ACT_NONE));
  AttnDecP p{};
  p.mem = memory; p.T = T; p.D = Hh; p.key_off = key_off;
  p.init_mode = !g.attn_enc_init ? 0 : (g.attn_keys == D2T_ATTN_KEYS_ALL_INIT_MEAN ? 1 : 2);
  p.kp = kp; p.wq_t = c->attn.wq_t; p.bq = c->attn.bq; p.wloc = c->attn.wloc; p.bloc = c->attn.bloc;
  p.taps = c->attn.taps; p.wscore = c->attn.wscore; p.bscore = c->attn.bscore;
  p.wx_t = c->attn.wx_t; p.bx = c->attn.bx; p.wg_t = c->attn.wg_t; p.bg = c->attn.bg;
  p.wih_t = c->attn.wih_t; p.bih = c->attn.bih; p.wic_t = c->attn.wic_t; p.bic = c->attn.bic;
  p.emb = c->attn.emb; p.tokgate = c->attn.tokgate; p.probs = d_logits; p.tokens = d_dummy; p.end_step = d_end;
  p.S = 1; p.V = V; p.H = Hh; p.E = Hh; p.coverage = 1; p.end_token = 1;
  p.step_mode = 1;
  p.st_h_in = st[0]; p.st_c_in = st[1]; p.st_mem_in = st[4];
  p.st_h_out = st[2]; p.st_c_out = st[3]; p.st_mem_out = st[5];
  p.tok_in = d_tok; p.row_sample = d_map;

  struct Smp {
    std::vector<std::vector<int64_t>> seqs, complete;
    std::vector<float> live, cscores;
    int k;
    bool last_completed = false, finished = false;
  };
  std::vector<Smp> sm((size_t)N);
  for (auto& x : sm) {
    x.seqs.assign((size_t)beam_size, std::vector<int64_t>{0});
    x.live.assign((size_t)beam_size, 0.f);
    x.k = beam_size;
  }
  for (int step = 0; step < S; ++step) {
    int rows = 0;
    for (int i = 0; i < N; ++i) {
      Smp& x = sm[i];
      const int M = x.finished ? 0 : (int)x.seqs.size();
      // step 0: all rows of a sample are identical and the reference ranks its row 0 only
      h_seg[3 * i] = rows; h_seg[3 * i + 1] = M ? (step == 0 ? 1 : M) : 0; h_seg[3 * i + 2] = x.finished ? 0 : x.k;
      for (int j = 0; j < M; ++j) { h_scores[rows + j] = x.live[j]; h_map[rows + j] = i; }
      rows += M;
    }
    if (!rows) break;
    BCHK(hipMemcpyAsync(d_scores, h_scores, (size_t)rows * 4, hipMemcpyHostToDevice, s));
    BCHK(hipMemcpyAsync(d_map, h_map, (size_t)rows * 4, hipMemcpyHostToDevice, s));
    BCHK(hipMemcpyAsync(d_seg, h_seg, (size_t)N * 12, hipMemcpyHostToDevice, s));
    p.B = rows; p.first = step == 0;
    BCHK(launch_attn_decode(p, s));
    BCHK(launch_beam_topk_batch(d_logits, d_scores, d_seg, N, V, beam_size, d_topv, d_topi, s));
    BCHK(hipMemcpyAsync(h_topv, d_topv, (size_t)cap * 4, hipMemcpyDeviceToHost, s));
    BCHK(hipMemcpyAsync(h_topi, d_topi, (size_t)cap * 4, hipMemcpyDeviceToHost, s));
    BCHK(hipStreamSynchronize(s));
    int nrows = 0;
    for (int i = 0; i < N; ++i) {
      Smp& x = sm[i];
      if (x.finished) continue;
      const int off = h_seg[3 * i];
      std::vector<std::vector<int64_t>> nseqs;
      std::vector<float> nscores;
      std::vector<int> ih, im;
      std::vector<int64_t> nt;
      x.last_completed = false;
      for (int r = 0; r < x.k; ++r) {
        const int idx = h_topi[(size_t)i * beam_size + r], prev = idx / V, word = idx % V;
        std::vector<int64_t> sq = x.seqs[prev];
        sq.push_back(word);
        if (word == 1) {
          x.complete.push_back(std::move(sq));
          x.cscores.push_back(h_topv[(size_t)i * beam_size + r]);
          x.last_completed = true;
        } else {
          ih.push_back(off + prev);  // LSTM state: hidden[prev_word_inds[incomplete]]
          im.push_back(off + r);     // coverage memory: (alpha_cum + alpha)[incomplete]
          nt.push_back(word);
          nseqs.push_back(std::move(sq));
          nscores.push_back(h_topv[(size_t)i * beam_size + r]);
        }
      }
      x.seqs.swap(nseqs);
      x.live.swap(nscores);
      x.k = (int)x.seqs.size();
      if (x.k == 0) { x.finished = true; continue; }
      for (size_t j = 0; j < ih.size(); ++j) { h_idxh[nrows] = ih[j]; h_idxm[nrows] = im[j]; h_tok[nrows] = nt[j]; ++nrows; }
    }
    if (nrows && step + 1 < S) {
      BCHK(hipMemcpyAsync(d_idxh, h_idxh, (size_t)nrows * 4, hipMemcpyHostToDevice, s));
      BCHK(hipMemcpyAsync(d_idxm, h_idxm, (size_t)nrows * 4, hipMemcpyHostToDevice, s));
      BCHK(hipMemcpyAsync(d_tok, h_tok, (size_t)nrows * 8, hipMemcpyHostToDevice, s));
      BCHK(launch_gather_rows(st[2], st[0], d_idxh, nrows, Hh, s));
      BCHK(launch_gather_rows(st[3], st[1], d_idxh, nrows, Hh, s));
      BCHK(launch_gather_rows(st[5], st[4], d_idxm, nrows, Tk, s));
    }
  }
  BCHK(hipStreamSynchronize(s));
#undef BCHK
  for (int i = 0; i < N; ++i) {
    Smp& x = sm[i];
    std::vector<int64_t> out;
    float score;
    if (!x.last_completed) {  // seq2seq.py:209-216
      out.assign(x.seqs[0].begin() + 1, x.seqs[0].end());
      score = x.live[0];
    } else {
      size_t best = 0;
      for (size_t j = 1; j < x.complete.size(); ++j)
        if ((double)x.cscores[j] / (double)x.complete[j].size() > (double)x.cscores[best] / (double)x.complete[best].size()) best = j;
      out.assign(x.complete[best].begin() + 1, x.complete[best].end());
      score = *std::max_element(x.cscores.begin(), x.cscores.end());
    }
    const int n = (int)std::min<size_t>(out.size(), (size_t)S);
    for (int j = 0; j < n; ++j) seq_out[(size_t)i * S + j] = out[j];
    len_out[i] = n;
    score_out[i] = score;
  }
  return done(D2T_OK);
}

int d2t_decode_greedy_async(d2t_ctx* c, const float* memory, int32_t B, int32_t T, const int64_t* start_tokens,
                            int64_t* tokens, float* logits, d2t_stream stream) {
  DevGuard dg_(c);
  if (!c || !memory || !start_tokens || !tokens || !logits || B < 1 || T < 1) return fail(c, D2T_EINVAL, "bad argument");
  if (!c->finalized) return fail(c, D2T_ESTATE, "weights not finalized");
  if (T > memory_cap(c)) return fail(c, D2T_EINVAL, "memory length %d > %d unsupported", T, memory_cap(c));
  if (c->cfg.decoder != D2T_DEC_TFM) return fail(c, D2T_ESTATE, "context was not created with the TFM decoder");
  if (int rc = check_dev_ptr(c, memory, "memory")) return rc;
  if (int rc = check_dev_ptr(c, logits, "logits")) return rc;
  return greedy_impl(c, memory, B, T, start_tokens, 0, tokens, logits, nullptr, (hipStream_t)stream, true);
}

int d2t_decode_greedy_submit(d2t_ctx* c, const float* memory, int32_t B, int32_t T, const int64_t* start_tokens,
                             int32_t is_test, int32_t rows_per_batch, int64_t* tokens, float* logits, d2t_stream stream,
                             int64_t* ticket_out) {
  DevGuard dg_(c);
  if (!c || !memory || !start_tokens || !tokens || !logits || B < 1 || T < 1) return fail(c, D2T_EINVAL, "bad argument");
  if (!c->finalized) return fail(c, D2T_ESTATE, "weights not finalized");
  if (T > memory_cap(c)) return fail(c, D2T_EINVAL, "memory length %d > %d unsupported", T, memory_cap(c));
  if (c->cfg.decoder != D2T_DEC_TFM) return fail(c, D2T_ESTATE, "context was not created with the TFM decoder");
  if (rows_per_batch < 0 || (rows_per_batch > 0 && B % rows_per_batch)) return fail(c, D2T_EINVAL, "rows_per_batch must divide the row count");
  if (int rc = check_dev_ptr(c, memory, "memory")) return rc;
  if (int rc = check_dev_ptr(c, logits, "logits")) return rc;
  const int rc = greedy_impl(c, memory, B, T, start_tokens, is_test, tokens, logits, nullptr, (hipStream_t)stream, true, rows_per_batch);
  if (rc == D2T_OK && ticket_out) *ticket_out = c->last_ticket;
  return rc;
}

int d2t_decode_wait(d2t_ctx* c, d2t_stream stream, int32_t host_sync) {
  DevGuard dg_(c);
  if (!c) return D2T_EINVAL;
  for (int i = 0; i < d2t_ctx::MAXC; ++i)
    if (c->ev_done_valid[i]) HIPCHK(c, hipStreamWaitEvent((hipStream_t)stream, c->ev_done[i], 0));
  if (host_sync) {
    HIPCHK(c, sync_chains(c));
    c->decode_in_flight = false;
  }
  return D2T_OK;
}

int64_t d2t_decode_last_ticket(const d2t_ctx* c) { return c ? c->last_ticket : 0; }

// debug (undocumented, env D2T_DECODE_TRACE): reset / read the per-kernel-node timeline of the most recently captured loop
int d2t_debug_trace(d2t_ctx* c, unsigned long long* out, int32_t max_slots, int32_t reset) {
  DevGuard dg_(c);
  if (!c || !c->dtrace) return 0;
  hipDeviceSynchronize();
  const int n = std::min<int>(max_slots, c->dtrace_next);
  if (out && n > 0) hipMemcpy(out, c->dtrace, (size_t)n * 16, hipMemcpyDeviceToHost);
  if (reset) {
    std::vector<unsigned long long> init((size_t)d2t_ctx::DTRACE_SLOTS * 2);
    for (size_t i = 0; i < init.size(); i += 2) { init[i] = ~0ull; init[i + 1] = 0; }
    hipMemcpy(c->dtrace, init.data(), init.size() * 8, hipMemcpyHostToDevice);
  }
  return n;
}

// 1: the decode with this ticket has completed; 0: still running; < 0: error.  Tickets older than the event ring are
// complete by construction: a chain is an in-order stream and the ring holds TICKET_RING >> 2 chains' worth of decodes.
int d2t_decode_query(d2t_ctx* c, int64_t ticket) {
  DevGuard dg_(c);
  if (!c || ticket < 1 || ticket > c->last_ticket) return -D2T_EINVAL;
  if (ticket + d2t_ctx::TICKET_RING <= c->last_ticket) return 1;
  const hipError_t e = hipEventQuery(c->ticket_ev[ticket % d2t_ctx::TICKET_RING]);
  if (e == hipSuccess) return 1;
  if (e == hipErrorNotReady) { (void)hipGetLastError(); return 0; }
  fail(c, D2T_EHIP, "hipEventQuery: %s", hipGetErrorString(e));
  return -D2T_EHIP;
}

// Decode steps of the batches of one asynchronous decode (blocks until that decode is complete): for an is_test decode the
// first step at which all rows of batch k had emitted [s] (max_seq_len + 1 if that never happened); otherwise max_seq_len + 1.
// n_out receives the number of batches in the decode's group.
int d2t_decode_steps(d2t_ctx* c, int64_t ticket, int32_t* steps_out, int32_t max_batches, int32_t* n_out) {
  DevGuard dg_(c);
  if (!c || !steps_out || ticket < 1 || ticket > c->last_ticket) return fail(c, D2T_EINVAL, "unknown decode ticket %lld", (long long)ticket);
  if (ticket + d2t_ctx::TICKET_RING <= c->last_ticket) return fail(c, D2T_ESTATE, "decode ticket %lld is too old", (long long)ticket);
  const int slot = (int)(ticket % d2t_ctx::TICKET_RING);
  HIPCHK(c, hipEventSynchronize(c->ticket_ev[slot]));
  const int nb = c->ticket_batches[slot];
  const int S = c->cfg.max_seq_len + 1;
  if (nb < 0) {  // not an early-exit decode: one entry, all steps
    if (max_batches < 1) return fail(c, D2T_EINVAL, "steps_out too small");
    steps_out[0] = S;
    if (n_out) *n_out = 1;
    return D2T_OK;
  }
  if (nb > max_batches) return fail(c, D2T_EINVAL, "steps_out holds %d entries, the decode has %d batches", max_batches, nb);
  for (int k = 0; k < nb; ++k) {
    const int v = c->h_steps[(size_t)slot * GRP_MAXB + k];
    steps_out[k] = v > 0 ? v : S;
  }
  if (n_out) *n_out = nb;
  return D2T_OK;
}

int d2t_decode_wait_ticket(d2t_ctx* c, int64_t ticket, d2t_stream stream, int32_t host_sync) {
  DevGuard dg_(c);
  if (!c || ticket < 1 || ticket > c->last_ticket) return fail(c, D2T_EINVAL, "unknown decode ticket %lld", (long long)ticket);
  if (ticket + d2t_ctx::TICKET_RING <= c->last_ticket) return D2T_OK;  // long since complete
  hipEvent_t ev = c->ticket_ev[ticket % d2t_ctx::TICKET_RING];
  HIPCHK(c, hipStreamWaitEvent((hipStream_t)stream, ev, 0));
  if (host_sync) HIPCHK(c, hipEventSynchronize(ev));
  return D2T_OK;
}


namespace {
// forward_beam (tfm.py:145-186) + Beam (tools/beam.py:38-140) for N samples with the bookkeeping ON THE DEVICE (round 4):
// the hypotheses of all samples are rows of one step loop, every kernel of a step is launched for the full N x beam row slots
// and reads the live row count / the stop step from the state block (kernels.h BeamDev), beam_dev_advance_kernel does
// Beam.advance for every sample after the per-sample top-k -- no host round trip in the loop, which is therefore ONE captured
// graph per (N, T, beam).  The host walks the (parent, token) history back once at the end.  Needs the absorbed row kernel
// with ancestry rows (no cache copy).  Row results are those of the host-side loop bit for bit (same kernels per row).
int beam_device_impl(d2t_ctx* c, const float* memory, int N, int T, int beam_size, int64_t* seq_out, int32_t* len_out,
                     float* score_out, hipStream_t user) {
  const d2t_config& g = c->cfg;
  const int S = g.max_seq_len + 1, V = g.vocab, d = g.dec_dim, cap = N * beam_size, Lmax = g.max_seq_len + 2;
  if (N > 1024) return fail(c, D2T_EINVAL, "batched beam search takes at most 1024 samples per call");
  select_chain(c, 0);
  hipStream_t s = c->dstream;
  DecBufs bf;
  int rc = dec_prepare(c, cap, T, &bf);
  if (rc) return rc;
  // workspace (4-byte words unless noted): logits [cap][V] | topv [cap] | topi [cap] | tok [cap] i64 | scores | map | prev [cap]
  // | seg [N][3] | ctrl [8] | comp_n, fin [N] | comp_t, comp_par, comp_score [N][beam] | hist_par, hist_tok [S][cap] | anc [2][cap][Lmax]
  size_t w = 0;
  auto take = [&](size_t words) { const size_t at = w; w += (words + 3) & ~(size_t)3; return at; };
  const size_t o_logits = take((size_t)cap * V), o_topv = take(cap), o_topi = take(cap), o_tok = take(2 * (size_t)cap);
  const size_t o_res = w;  // ---- from here to o_anc: the block copied back to the host at the end ----
  const size_t o_scores = take(cap), o_seg = take(3 * (size_t)N), o_ctrl = take(8), o_compn = take(N), o_fin = take(N);
  const size_t o_ct = take(cap), o_cp = take(cap), o_cs = take(cap), o_hp = take((size_t)S * cap), o_ht = take((size_t)S * cap);
  const size_t o_map = take(cap), o_prev = take(cap);
  const size_t res_words = o_map - o_res;
  const size_t o_anc = take(2 * (size_t)cap * Lmax);
  if ((rc = ensure(c, &c->beam_ws, &c->beam_ws_cap, w * 4 + 64))) return rc;
  float* base = c->beam_ws;
  float* d_logits = base + o_logits;
  BeamDev b{};
  b.ctrl = reinterpret_cast<int*>(base + o_ctrl);
  b.tok = reinterpret_cast<int64_t*>(base + o_tok);
  b.scores = base + o_scores;
  b.map = reinterpret_cast<int*>(base + o_map);
  b.prev = reinterpret_cast<int*>(base + o_prev);
  b.seg = reinterpret_cast<int*>(base + o_seg);
  b.comp_n = reinterpret_cast<int*>(base + o_compn);
  b.fin = reinterpret_cast<int*>(base + o_fin);
  b.comp_t = reinterpret_cast<int*>(base + o_ct);
  b.comp_par = reinterpret_cast<int*>(base + o_cp);
  b.comp_score = base + o_cs;
  b.hist_par = reinterpret_cast<int*>(base + o_hp);
  b.hist_tok = reinterpret_cast<int*>(base + o_ht);
  b.topv = base + o_topv;
  b.topi = reinterpret_cast<const int*>(base + o_topi);
  b.N = N; b.beam = beam_size; b.cap = cap; b.V = V; b.S = S; b.end_token = TOK_END;
  int* d_anc[2] = {reinterpret_cast<int*>(base + o_anc), reinterpret_cast<int*>(base + o_anc) + (size_t)cap * Lmax};
  const int* rows_ptr = b.ctrl + 1;
  const int* stop = b.ctrl + 2;
  if (c->h_beam_cap < res_words * 4) {
    if (c->h_beam) hipHostFree(c->h_beam);
    c->h_beam = nullptr; c->h_beam_cap = 0;
    if (hipHostMalloc(reinterpret_cast<void**>(&c->h_beam), res_words * 4, hipHostMallocDefault) != hipSuccess)
      return fail(c, D2T_ENOMEM, "hipHostMalloc failed");
    c->h_beam_cap = res_words * 4;
  }
  HIPCHK(c, hipEventRecord(c->ev_in, user));
  HIPCHK(c, hipStreamWaitEvent(s, c->ev_in, 0));
  c->ckv = c->ckv2[0];  // the internal stream is in order, so earlier decodes are done with the slot
  HIPCHK(c, cross_kv(c, s, memory, N, T));
  c->skv_cur = c->skv;
  auto enqueue_loop = [&](hipStream_t st) -> hipError_t {
    hipError_t e;
#define LTRY(x) do { if ((e = (x)) != hipSuccess) return e; } while (0)
    LTRY(hipMemsetAsync(c->dstate, 0, (size_t)(4 + cap) * 4, st));
    LTRY(launch_beam_dev_init(b, TOK_GO, st));
    for (int step = 0; step < S; ++step) {
      LTRY(launch_beam_ancestry(d_anc[(step + 1) & 1], d_anc[step & 1], b.prev, cap, Lmax, b.ctrl, c->dstate, st, rows_ptr, stop));
      LTRY(launch_embed_tokens(c->word_embed, c->word_pe, b.tok, c->dstate, bf.x, cap, d, st, rows_ptr, stop));
      LTRY(decode_step(c, st, bf, cap, T, cap, false, d_logits, V, 0, N, b.map, stop, beam_size, b.seg, d_anc[step & 1], rows_ptr));
      LTRY(launch_beam_topk_batch(d_logits, b.scores, b.seg, N, V, beam_size, base + o_topv, reinterpret_cast<int*>(base + o_topi), st,
                                  c->dstate, stop));
      LTRY(launch_beam_dev_advance(b, st));
    }
#undef LTRY
    return hipSuccess;
  };
  const bool use_graph = D2T_PROBE_ENV_STR("D2T_NO_GRAPH") == nullptr;
  if (use_graph) {
    d2t_ctx::GraphKey k;
    memset(&k, 0, sizeof k);
    k.B = cap; k.T = T; k.steps = S; k.tok = nullptr; k.logits = base; k.ckv = c->ckv; k.dws = c->dws; k.skv = c->skv; k.dstate = c->dstate;
    k.variant = 3 | ((long long)N << 8) | ((long long)beam_size << 40);  // (bits 0-1 = 3: the device-side beam loop)
    hipGraphExec_t exec = nullptr;
    for (size_t i = 0; i < c->graphs.size(); ++i)
      if (memcmp(&k, &c->graphs[i].key, sizeof k) == 0) {
        exec = c->graphs[i].exec;
        if (i + 1 != c->graphs.size()) std::swap(c->graphs[i], c->graphs.back());
        break;
      }
    if (!exec) {
      hipGraph_t gr = nullptr;
      HIPCHK(c, hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
      hipError_t e = enqueue_loop(s);
      hipError_t e2 = hipStreamEndCapture(s, &gr);
      if (e != hipSuccess || e2 != hipSuccess) {
        if (gr) hipGraphDestroy(gr);
        return fail(c, D2T_EHIP, "beam graph capture: %s", hipGetErrorString(e != hipSuccess ? e : e2));
      }
      e = hipGraphInstantiate(&exec, gr, nullptr, nullptr, 0);
      hipGraphDestroy(gr);
      if (e != hipSuccess) return fail(c, D2T_EHIP, "hipGraphInstantiate: %s", hipGetErrorString(e));
      if (c->graphs.size() >= 40) {
        HIPCHK(c, sync_chains(c));
        hipGraphExecDestroy(c->graphs.front().exec);
        c->graphs.erase(c->graphs.begin());
      }
      c->graphs.push_back({k, exec});
    }
    HIPCHK(c, hipGraphLaunch(exec, s));
  } else {
    HIPCHK(c, enqueue_loop(s));
  }
  HIPCHK(c, hipMemcpyAsync(c->h_beam, base + o_res, res_words * 4, hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipStreamSynchronize(s));
  // ---- host: pick every sample's best hypothesis (beam.py:107-140) and walk its tokens back through the history ----
  const char* hb = c->h_beam;
  auto hw = [&](size_t off) { return reinterpret_cast<const int*>(hb + (off - o_res) * 4); };
  const float* h_scores = reinterpret_cast<const float*>(hw(o_scores));
  const int *h_seg = hw(o_seg), *h_ctrl = hw(o_ctrl), *h_compn = hw(o_compn), *h_ct = hw(o_ct), *h_cp = hw(o_cp);
  const float* h_cs = reinterpret_cast<const float*>(hw(o_cs));
  const int *h_hp = hw(o_hp), *h_ht = hw(o_ht);
  const int steps_run = h_ctrl[3];
  for (int i = 0; i < N; ++i) {
    int64_t* out = seq_out + (size_t)i * S;
    const int nc = h_compn[i];
    int row, last_step, n;  // the hypothesis ends with the history record (last_step, row); n tokens are returned
    float score;
    if (nc > 0) {
      int best = 0;
      for (int j = 1; j < nc; ++j)
        if ((double)h_cs[(size_t)i * beam_size + j] / (double)(h_ct[(size_t)i * beam_size + j] + 1) >
            (double)h_cs[(size_t)i * beam_size + best] / (double)(h_ct[(size_t)i * beam_size + best] + 1))
          best = j;
      const int t = h_ct[(size_t)i * beam_size + best];
      n = std::min(t + 1, S);
      for (int j = 0; j < n; ++j) out[j] = TOK_PAD;
      if (t < n) out[t] = TOK_END;
      row = h_cp[(size_t)i * beam_size + best];
      last_step = t - 1;
      score = h_cs[(size_t)i * beam_size + best];
    } else {  // Beam.set_hypothesis (beam.py:132-140): the first live hypothesis, padded to max_seq_len + 1
      n = S;
      for (int j = 0; j < n; ++j) out[j] = TOK_PAD;
      if (h_seg[3 * i + 1] > 0) { row = h_seg[3 * i]; last_step = steps_run - 1; score = h_scores[row]; }
      else { row = -1; last_step = -1; score = 0.f; }
    }
    for (int p = last_step; p >= 0 && row >= 0; --p) {
      if (p < n) out[p] = h_ht[(size_t)p * cap + row];
      row = h_hp[(size_t)p * cap + row];
    }
    len_out[i] = n;
    score_out[i] = score;
  }
  return D2T_OK;
}
}  // namespace

int d2t_decode_beam(d2t_ctx* c, const float* memory, int32_t T, int32_t beam_size, int64_t* seq_out, int32_t* len_out,
                    float* score_out, d2t_stream stream) {
  DevGuard dg_(c);
  // TransformerPrediction.forward_beam (tfm.py:145-186) with tools/beam.py:38-140 bookkeeping on the
  // host; a fresh beam per call (demo reset_beam semantics).  The model runs KV-cached on
  // the device for the live hypotheses only; log_softmax + flat top-k run on the device too, so each
  // step moves k <= beam_size (value, index) pairs to the host instead of the [hyp, V] log-prob matrix.
  if (!c || !memory || !seq_out || !len_out || !score_out || T < 1) return fail(c, D2T_EINVAL, "bad argument");
  if (!c->finalized) return fail(c, D2T_ESTATE, "weights not finalized");
  if (beam_size < 1 || beam_size > 16) return fail(c, D2T_EINVAL, "beam_size must be in [1,16]");
  if (c->cfg.decoder != D2T_DEC_TFM) return fail(c, D2T_ESTATE, "beam search is implemented for the TFM decoder only");
  if (T > memory_cap(c)) return fail(c, D2T_EINVAL, "memory length %d > %d unsupported", T, memory_cap(c));
  const d2t_config& g = c->cfg;
  const int S = g.max_seq_len + 1, V = g.vocab, d = g.dec_dim, cap = beam_size;
  const int heads = g.dec_heads, hd = d / heads, Lmax = g.max_seq_len + 2;
  if ((long long)cap * V > 16 * 4096) return fail(c, D2T_EINVAL, "beam_size * vocab too large");
  // the absorbed d_model-256 decoder: the device-side loop of the batched search with one sample (the same row kernels, so
  // batched == per-sample bit for bit), one graph launch per call instead of ~4000 kernel launches and 151 host round trips
  if (c->dec_absorbed && !c->beam_shared_tile && Lmax <= 512 && D2T_PROBE_ENV_STR("D2T_BEAM_HOST") == nullptr)
    return beam_device_impl(c, memory, 1, T, beam_size, seq_out, len_out, score_out, (hipStream_t)stream);
  select_chain(c, 0);
  hipStream_t user = (hipStream_t)stream, s = c->dstream;
  DecBufs bf;
  int rc = dec_prepare(c, cap, T, &bf);
  if (rc) return rc;
  const size_t skv_bytes = (size_t)g.dec_layers * 2 * cap * Lmax * d * 4;
  if ((rc = ensure(c, &c->skv_alt, &c->skv_alt_cap, skv_bytes))) return rc;
  // beam workspace: logits [cap][V] | scores [cap] | topv [cap] | tok [cap] i64 | topi [cap] | prev [cap]
  const size_t ws_bytes = ((size_t)cap * V + 2 * cap) * 4 + (size_t)cap * 8 + 2 * (size_t)cap * 4 + 64;
  if ((rc = ensure(c, &c->beam_ws, &c->beam_ws_cap, ws_bytes))) return rc;
  float* d_logits = c->beam_ws;
  float* d_scores = d_logits + (size_t)cap * V;
  float* d_topv = d_scores + cap;
  const size_t tok_off = ((size_t)cap * V + 2 * (size_t)cap + 1) & ~(size_t)1;  // 8-byte aligned
  int64_t* d_tok = reinterpret_cast<int64_t*>(d_logits + tok_off);
  int* d_topi = reinterpret_cast<int*>(d_tok + cap);
  int* d_prev = d_topi + cap;
  // pinned host mirror: [0]=step | tok i64[cap] | scores[cap] | topv[cap] | topi[cap] | prev[cap]
  static_assert(sizeof(int64_t) == 8, "");
  char* hp = nullptr;
  const size_t hbytes = 16 + (size_t)cap * (8 + 4 * 4);
  if (hipHostMalloc(reinterpret_cast<void**>(&hp), hbytes, hipHostMallocDefault) != hipSuccess)
    return fail(c, D2T_ENOMEM, "hipHostMalloc failed");
  int* h_step = reinterpret_cast<int*>(hp);
  int64_t* h_tok = reinterpret_cast<int64_t*>(hp + 16);
  float* h_scores = reinterpret_cast<float*>(hp + 16 + (size_t)cap * 8);
  float* h_topv = h_scores + cap;
  int* h_topi = reinterpret_cast<int*>(h_topv + cap);
  int* h_prev = h_topi + cap;
  auto done = [&](int code) { hipHostFree(hp); return code; };
#define BCHK(expr)                                                                              \
  do {                                                                                          \
    hipError_t e_ = (expr);                                                                     \
    if (e_ != hipSuccess) return done(fail(c, D2T_EHIP, "%s: %s", #expr, hipGetErrorString(e_))); \
  } while (0)

  BCHK(hipEventRecord(c->ev_in, user));
  BCHK(hipStreamWaitEvent(s, c->ev_in, 0));
  BCHK(hipMemsetAsync(c->dstate, 0, (size_t)(4 + cap) * 4, s));
  c->ckv = c->ckv2[0];  // the internal stream is in order, so earlier decodes are done with the slot
  BCHK(cross_kv(c, s, memory, 1, T));
  c->skv_cur = c->skv;
  float* skv_other = c->skv_alt;

  struct Hyp { std::vector<int64_t> seq; float score; };  // seq without the leading [GO]
  std::vector<Hyp> hyps(1), completed;
  hyps[0].score = 0.f;
  int64_t last_tok[16];
  last_tok[0] = TOK_GO;
  for (int step = 0; step < S; ++step) {
    const int M = (int)hyps.size();
    *h_step = step;
    for (int i = 0; i < M; ++i) { h_tok[i] = last_tok[i]; h_scores[i] = hyps[i].score; }
    BCHK(hipMemcpyAsync(c->dstate, h_step, 4, hipMemcpyHostToDevice, s));
    BCHK(hipMemcpyAsync(d_tok, h_tok, (size_t)M * 8, hipMemcpyHostToDevice, s));
    BCHK(hipMemcpyAsync(d_scores, h_scores, (size_t)M * 4, hipMemcpyHostToDevice, s));
    BCHK(launch_embed_tokens(c->word_embed, c->word_pe, d_tok, c->dstate, bf.x, M, d, s));
    BCHK(decode_step(c, s, bf, M, T, cap, true, d_logits, V, 0, -1, nullptr, nullptr, beam_size));
    const int live = beam_size - (int)completed.size();
    BCHK(launch_beam_topk(d_logits, d_scores, M, V, live, d_topv, d_topi, s));
    BCHK(hipMemcpyAsync(h_topv, d_topv, (size_t)live * 4, hipMemcpyDeviceToHost, s));
    BCHK(hipMemcpyAsync(h_topi, d_topi, (size_t)live * 4, hipMemcpyDeviceToHost, s));
    BCHK(hipStreamSynchronize(s));
    // Beam.advance (tools/beam.py:68-105)
    std::vector<Hyp> next;
    int nprev = 0;
    for (int r = 0; r < live; ++r) {
      const int prev = h_topi[r] / V, word = h_topi[r] % V;
      Hyp h;
      h.seq = hyps[prev].seq;
      h.seq.push_back(word);
      h.score = h_topv[r];
      if (word == TOK_END) {
        completed.push_back(std::move(h));
      } else {
        last_tok[nprev] = word;
        h_prev[nprev++] = prev;
        next.push_back(std::move(h));
      }
    }
    if ((int)completed.size() == beam_size) { hyps.swap(next); break; }  // Beam.done, tfm.py:175-176
    hyps.swap(next);
    if (step + 1 < S) {  // reorder the self-attention caches to the surviving hypotheses
      BCHK(hipMemcpyAsync(d_prev, h_prev, (size_t)nprev * 4, hipMemcpyHostToDevice, s));
      BCHK(launch_cache_gather(c->skv_cur, skv_other, d_prev, g.dec_layers * 2, cap, nprev, heads, Lmax, hd, step + 1,
                               s));
      std::swap(c->skv_cur, skv_other);
    }
  }
  BCHK(hipStreamSynchronize(s));
#undef BCHK
  if (completed.empty()) {  // Beam.set_hypothesis (beam.py:132-140): hypotheses[0, 1:] incl. trailing [PAD]s
    Hyp h = hyps.empty() ? Hyp{} : hyps[0];
    h.seq.resize((size_t)g.max_seq_len + 1, TOK_PAD);
    completed.push_back(std::move(h));
  }
  size_t best = 0;
  for (size_t i = 1; i < completed.size(); ++i)
    if ((double)completed[i].score / (double)std::max<size_t>(1, completed[i].seq.size()) >
        (double)completed[best].score / (double)std::max<size_t>(1, completed[best].seq.size()))
      best = i;
  const Hyp& bh = completed[best];
  const int n = (int)std::min<size_t>(bh.seq.size(), (size_t)S);
  for (int i = 0; i < n; ++i) seq_out[i] = bh.seq[i];
  *len_out = n;
  *score_out = bh.score;
  return done(D2T_OK);
}

int d2t_decode_beam_batch(d2t_ctx* c, const float* memory, int32_t N, int32_t T, int32_t beam_size, int64_t* seq_out,
                          int32_t* len_out, float* score_out, d2t_stream stream) {
  DevGuard dg_(c);
  // forward_beam (tfm.py:145-186) + Beam (tools/beam.py) for N samples AT ONCE: the hypotheses of all samples are rows
  // of one step loop (each row attends over its own sample's cross K/V through a row map), log_softmax + top-k run per
  // sample segment, the bookkeeping of every sample is the single-sample one.  Results equal N calls of
  // d2t_decode_beam; the point is throughput (one host round trip per step instead of N).
  if (!c || !memory || !seq_out || !len_out || !score_out || N < 1 || T < 1) return fail(c, D2T_EINVAL, "bad argument");
  if (!c->finalized) return fail(c, D2T_ESTATE, "weights not finalized");
  if (beam_size < 1 || beam_size > 16) return fail(c, D2T_EINVAL, "beam_size must be in [1,16]");
  if (c->cfg.decoder != D2T_DEC_TFM) return fail(c, D2T_ESTATE, "beam search is implemented for the TFM decoder only");
  if (T > memory_cap(c)) return fail(c, D2T_EINVAL, "memory length %d > %d unsupported", T, memory_cap(c));
  const d2t_config& g = c->cfg;
  const int S = g.max_seq_len + 1, V = g.vocab, d = g.dec_dim, cap = N * beam_size;
  const int heads = g.dec_heads, hd = d / heads, Lmax = g.max_seq_len + 2;
  if ((long long)beam_size * V > 16 * 4096) return fail(c, D2T_EINVAL, "beam_size * vocab too large");
  if (c->dec_absorbed && !c->beam_shared_tile && Lmax <= 512 && N <= 1024 && D2T_PROBE_ENV_STR("D2T_BEAM_HOST") == nullptr)
    return beam_device_impl(c, memory, N, T, beam_size, seq_out, len_out, score_out, (hipStream_t)stream);
  select_chain(c, 0);
  hipStream_t user = (hipStream_t)stream, s = c->dstream;
  DecBufs bf;
  int rc = dec_prepare(c, cap, T, &bf);
  if (rc) return rc;
  // Round 3: with the absorbed row kernel the self-attention cache is never copied -- every hypothesis keeps an ancestry row
  // (which cache row holds each of its earlier positions, launch_beam_ancestry); otherwise the survivors' caches are gathered
  // into the other buffer as before.
  const bool use_anc = c->dec_absorbed && !c->beam_shared_tile && Lmax <= 512 && D2T_PROBE_ENV_STR("D2T_BEAM_CACHE_COPY") == nullptr;
  const size_t skv_bytes = (size_t)g.dec_layers * 2 * cap * Lmax * d * 4;
  if (!use_anc && (rc = ensure(c, &c->skv_alt, &c->skv_alt_cap, skv_bytes))) return rc;
  // workspace: logits [cap][V] | topv [cap] | topi [cap] | step pack (one host -> device copy per step):
  //   tok [cap] i64 | scores [cap] | rowmap [cap] | prev [cap] | seg [N][3] | step [4] | ancestry [2][cap][Lmax]
  const size_t pack_off = (((size_t)cap * V + 2 * (size_t)cap) * 4 + 15) & ~(size_t)15;
  const size_t pack_bytes = ((size_t)cap * (8 + 3 * 4) + (size_t)N * 12 + 16 + 15) & ~(size_t)15;
  const size_t anc_words = use_anc ? 2 * (size_t)cap * Lmax : 0;
  const size_t ws_bytes = pack_off + pack_bytes + anc_words * 4 + 64;
  if ((rc = ensure(c, &c->beam_ws, &c->beam_ws_cap, ws_bytes))) return rc;
  float* d_logits = c->beam_ws;
  float* d_topv = d_logits + (size_t)cap * V;
  int* d_topi = reinterpret_cast<int*>(d_topv + cap);
  char* d_pack = reinterpret_cast<char*>(c->beam_ws) + pack_off;
  int64_t* d_tok = reinterpret_cast<int64_t*>(d_pack);
  float* d_scores = reinterpret_cast<float*>(d_tok + cap);
  int* d_map = reinterpret_cast<int*>(d_scores + cap);
  int* d_prev = d_map + cap;
  int* d_seg = d_prev + cap;
  int* d_step = d_seg + 3 * (size_t)N;
  int* d_anc[2] = {reinterpret_cast<int*>(d_pack + pack_bytes), reinterpret_cast<int*>(d_pack + pack_bytes) + (size_t)cap * Lmax};
  char* hp = nullptr;
  if (hipHostMalloc(reinterpret_cast<void**>(&hp), pack_bytes + 2 * (size_t)cap * 4 + 64, hipHostMallocDefault) != hipSuccess)
    return fail(c, D2T_ENOMEM, "hipHostMalloc failed");
  int64_t* h_tok = reinterpret_cast<int64_t*>(hp);
  float* h_scores = reinterpret_cast<float*>(h_tok + cap);
  int* h_map = reinterpret_cast<int*>(h_scores + cap);
  int* h_prev = h_map + cap;
  int* h_seg = h_prev + cap;
  int* h_step = h_seg + 3 * (size_t)N;
  float* h_topv = reinterpret_cast<float*>(hp + pack_bytes);  // [topv | topi]: one device -> host copy per step
  int* h_topi = reinterpret_cast<int*>(h_topv + cap);
  auto done = [&](int code) { hipHostFree(hp); return code; };
#define BCHK(expr)                                                                              \
  do {                                                                                          \
    hipError_t e_ = (expr);                                                                     \
    if (e_ != hipSuccess) return done(fail(c, D2T_EHIP, "%s: %s", #expr, hipGetErrorString(e_))); \
  } while (0)
  BCHK(hipEventRecord(c->ev_in, user));
  BCHK(hipStreamWaitEvent(s, c->ev_in, 0));
  BCHK(hipMemsetAsync(c->dstate, 0, (size_t)(4 + cap) * 4, s));
  c->ckv = c->ckv2[0];
  BCHK(cross_kv(c, s, memory, N, T));
  c->skv_cur = c->skv;
  float* skv_other = c->skv_alt;

  // Beam bookkeeping (tools/beam.py:68-105) on a token trie: a hypothesis is (score, node); its sequence is the path to the
  // root, written out once at the end (the reference concatenates the sequences every step)
  struct Node { int parent; int64_t tok; int len; };
  struct Hyp { int node; float score; };
  std::vector<Node> trie;
  trie.reserve((size_t)cap * S);
  auto seq_len = [&](int node) { return node < 0 ? 0 : trie[(size_t)node].len; };
  std::vector<std::vector<Hyp>> hyps((size_t)N, std::vector<Hyp>(1, Hyp{-1, 0.f})), completed((size_t)N);
  std::vector<std::vector<int64_t>> last((size_t)N, std::vector<int64_t>{TOK_GO});
  std::vector<char> finished((size_t)N, 0);
  int nprev = 0;  // survivors of the previous step, in this step's row order: h_prev[0 .. nprev)
  for (int step = 0; step < S; ++step) {
    int rows = 0;
    for (int i = 0; i < N; ++i) {
      const int M = finished[i] ? 0 : (int)hyps[i].size();
      h_seg[3 * i] = rows; h_seg[3 * i + 1] = M; h_seg[3 * i + 2] = finished[i] ? 0 : beam_size - (int)completed[i].size();
      for (int j = 0; j < M; ++j) { h_tok[rows + j] = last[i][j]; h_scores[rows + j] = hyps[i][j].score; h_map[rows + j] = i; }
      rows += M;
    }
    if (!rows) break;
    if (step > 0 && nprev != rows) return done(fail(c, D2T_ESTATE, "beam bookkeeping: %d survivors, %d rows", nprev, rows));
    *h_step = step;
    BCHK(hipMemcpyAsync(d_pack, hp, pack_bytes, hipMemcpyHostToDevice, s));
    if (use_anc) {
      BCHK(launch_beam_ancestry(d_anc[(step + 1) & 1], d_anc[step & 1], d_prev, rows, Lmax, d_step, c->dstate, s));
    } else {
      BCHK(launch_beam_ancestry(nullptr, nullptr, d_prev, 1, 0, d_step, c->dstate, s));  // publishes the step only
      if (step > 0) {  // the survivors' caches move to their new row positions
        BCHK(launch_cache_gather(c->skv_cur, skv_other, d_prev, g.dec_layers * 2, cap, rows, heads, Lmax, hd, step, s));
        std::swap(c->skv_cur, skv_other);
      }
    }
    BCHK(launch_embed_tokens(c->word_embed, c->word_pe, d_tok, c->dstate, bf.x, rows, d, s));
    BCHK(decode_step(c, s, bf, rows, T, cap, false, d_logits, V, 0, N, d_map, nullptr, beam_size, d_seg,
                     use_anc ? d_anc[step & 1] : nullptr));
    BCHK(launch_beam_topk_batch(d_logits, d_scores, d_seg, N, V, beam_size, d_topv, d_topi, s));
    BCHK(hipMemcpyAsync(h_topv, d_topv, 2 * (size_t)cap * 4, hipMemcpyDeviceToHost, s));
    BCHK(hipStreamSynchronize(s));
    nprev = 0;
    for (int i = 0; i < N; ++i) {  // Beam.advance (tools/beam.py:68-105) per sample
      if (finished[i]) continue;
      const int off = h_seg[3 * i], live = h_seg[3 * i + 2];
      std::vector<Hyp> next;
      std::vector<int64_t> nl;
      const int first_prev = nprev;
      for (int r = 0; r < live; ++r) {
        const int idx = h_topi[(size_t)i * beam_size + r], prev = idx / V, word = idx % V;
        const int parent = hyps[i][prev].node;
        trie.push_back(Node{parent, (int64_t)word, seq_len(parent) + 1});
        const Hyp h{(int)trie.size() - 1, h_topv[(size_t)i * beam_size + r]};
        if (word == TOK_END) {
          completed[i].push_back(h);
        } else {
          nl.push_back(word);
          h_prev[nprev++] = off + prev;
          next.push_back(h);
        }
      }
      hyps[i].swap(next);
      last[i].swap(nl);
      if ((int)completed[i].size() == beam_size) { finished[i] = 1; nprev = first_prev; }  // Beam.done: its rows drop out
    }
  }
  BCHK(hipStreamSynchronize(s));
#undef BCHK
  for (int i = 0; i < N; ++i) {
    std::vector<Hyp>& comp = completed[i];
    bool padded = false;
    if (comp.empty()) {  // Beam.set_hypothesis (beam.py:132-140): the first live hypothesis, padded to max_seq_len + 1
      comp.push_back(hyps[i].empty() ? Hyp{-1, 0.f} : hyps[i][0]);
      padded = true;
    }
    auto len_of = [&](const Hyp& h) { return padded ? (size_t)g.max_seq_len + 1 : (size_t)seq_len(h.node); };
    size_t best = 0;
    for (size_t j = 1; j < comp.size(); ++j)
      if ((double)comp[j].score / (double)std::max<size_t>(1, len_of(comp[j])) >
          (double)comp[best].score / (double)std::max<size_t>(1, len_of(comp[best])))
        best = j;
    const Hyp& bh = comp[best];
    const int have = seq_len(bh.node), n = (int)std::min<size_t>(len_of(bh), (size_t)S);
    for (int j = 0; j < n; ++j) seq_out[(size_t)i * S + j] = TOK_PAD;
    int node = bh.node;
    for (int j = have - 1; j >= 0; --j, node = trie[(size_t)node].parent)
      if (j < n) seq_out[(size_t)i * S + j] = trie[(size_t)node].tok;
    len_out[i] = n;
    score_out[i] = bh.score;
  }
  return done(D2T_OK);
}

int d2t_set_reserved_blocks(d2t_ctx* c, int32_t blocks) {
  DevGuard dg_(c);
  if (!c || blocks < 0) return fail(c, D2T_EINVAL, "bad argument");
  if (!c->num_cus) {
    hipDeviceProp_t prop;
    int dev = 0;
    HIPCHK(c, hipGetDevice(&dev));
    HIPCHK(c, hipGetDeviceProperties(&prop, dev));
    c->num_cus = prop.multiProcessorCount;
  }
  const int slots = 2 * c->num_cus;  // the convolution runs two blocks per CU
  if (blocks >= slots) return fail(c, D2T_EINVAL, "cannot reserve %d of %d block slots", blocks, slots);
  c->conv_max_blocks = blocks ? slots - blocks : 0;
  return D2T_OK;
}

int d2t_set_reserved_cus(d2t_ctx* c, int32_t cus) {
  DevGuard dg_(c);
  if (!c || cus < 0 || cus > 128) return fail(c, D2T_EINVAL, "reserved CUs must be in [0, 128]");
  c->reserved_cus = cus;
  return D2T_OK;
}

int d2t_set_conv_kernel(d2t_ctx* c, int32_t kind) {
  DevGuard dg_(c);
  if (!c || (kind != 0 && kind != 3))
    return fail(c, D2T_EINVAL, "conv kernel must be 3 (pipelined 256x128 on 16x16x32 MFMAs, one block per CU) or 0 (128x128 on 32x32x16 MFMAs, two blocks per CU)");
  if ((c->conv_f16 || c->mixed_units) && kind != 3) return fail(c, D2T_ESTATE, "the fp16x2 convolutions exist for conv kernel 3 only");
  c->conv_pipelined = kind;
  return D2T_OK;
}

int d2t_set_beam_shared_tile(d2t_ctx* c, int32_t on) {
  DevGuard dg_(c);
  if (!c) return D2T_EINVAL;
  c->beam_shared_tile = on != 0;
  return D2T_OK;
}

int d2t_set_conv_fusion(d2t_ctx* c, int32_t pools, int32_t shortcuts) {
  DevGuard dg_(c);
  if (!c) return D2T_EINVAL;
  c->no_pool_fusion = pools == 0;
  c->no_shortcut_fusion = shortcuts == 0;
  return D2T_OK;
}

int d2t_set_decode_chains(d2t_ctx* c, int32_t chains) {
  DevGuard dg_(c);
  if (!c || chains < 1 || chains > d2t_ctx::MAXC) return fail(c, D2T_EINVAL, "decode chains must be 1 .. %d", d2t_ctx::MAXC);
  c->n_chains = chains;
  return D2T_OK;
}

int d2t_set_conv_precision(d2t_ctx* c, int32_t mode) {
  DevGuard dg_(c);
  if (!c || (mode != D2T_CONV_FP32 && mode != D2T_CONV_BF16X3 && mode != D2T_CONV_FP16X2 && mode != D2T_CONV_MIXED))
    return fail(c, D2T_EINVAL, "unknown conv precision %d", mode);
  if ((mode == D2T_CONV_FP16X2 || mode == D2T_CONV_MIXED) && c->conv_pipelined != 3)
    return fail(c, D2T_ESTATE, "the fp16x2 convolutions exist for conv kernel 3 (pipelined 256x128 on 16x16x32 MFMAs) only");
  c->conv_bf16x3 = mode != D2T_CONV_FP32;
  c->conv_f16 = mode == D2T_CONV_FP16X2;
  c->mixed_units = mode == D2T_CONV_MIXED ? D2T_MIXED_UNITS_DEFAULT : 0;
  return D2T_OK;
}

int d2t_set_mixed_units(d2t_ctx* c, int32_t units) {
  DevGuard dg_(c);
  if (!c || units < 0 || units > 8) return fail(c, D2T_EINVAL, "mixed units must be 0 .. 8");
  if (units > 0 && (!c->conv_bf16x3 || c->conv_f16 || c->conv_pipelined != 3))
    return fail(c, D2T_ESTATE, "mixed units need the split-bf16 arithmetic on conv kernel 3");
  c->mixed_units = units;
  return D2T_OK;
}

int d2t_profile_enable(d2t_ctx* c, int32_t on) {
  DevGuard dg_(c);
  if (!c) return D2T_EINVAL;
  c->profiling = on != 0;
  return D2T_OK;
}

int d2t_profile_read(d2t_ctx* c, int32_t max_records, int32_t* n, int32_t* M, int32_t* N, int32_t* K, float* ms) {
  DevGuard dg_(c);
  if (!c || !n) return D2T_EINVAL;
  HIPCHK(c, hipDeviceSynchronize());
  int out = 0;
  for (auto& r : c->prof) {
    if (out < max_records && M && N && K && ms) {
      float t = 0.f;
      if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) t = -1.f;
      M[out] = r.M; N[out] = r.N; K[out] = r.K; ms[out] = t;
      ++out;
    }
    hipEventDestroy(r.a);
    hipEventDestroy(r.b);
  }
  c->prof.clear();
  *n = out;
  return D2T_OK;
}

// ---------------------------------------------------------------------------
// single-kernel entry points
// ---------------------------------------------------------------------------
int d2t_op_conv2d(const float* x, const float* w, const float* bias, const float* residual, float* y, int32_t B,
                  int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t KH, int32_t KW, int32_t SH, int32_t SW,
                  int32_t PH, int32_t PW, int32_t act, d2t_stream stream) {
  if (!x || !w || !y || SH < 1 || SW < 1) return D2T_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (Cin == 1) {
    if (KH != 3 || KW != 3 || SH != 1 || SW != 1 || PH != 1 || PW != 1 || residual) return D2T_EINVAL;
    return launch_stem(x, w, bias, y, B, H, W, Cout, act, s) == hipSuccess ? D2T_OK : D2T_EHIP;
  }
  if (Cin % 32) return D2T_EINVAL;
  float* wp = nullptr;  // the kernel's K order (test entry point: temporary repack, synchronous)
  if (hipMalloc(reinterpret_cast<void**>(&wp), (size_t)Cout * KH * KW * Cin * 4) != hipSuccess) return D2T_ENOMEM;
  ConvP p{};
  p.in = x; p.w = wp; p.bias = bias; p.res = residual; p.out = y;
  p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
  p.OH = (H + 2 * PH - KH) / SH + 1; p.OW = (W + 2 * PW - KW) / SW + 1;
  p.KH = KH; p.KW = KW; p.SH = SH; p.SW = SW; p.PH = PH; p.PW = PW;
  p.M = B * p.OH * p.OW; p.K = KH * KW * Cin; p.act = act;
  hipError_t e = launch_repack_ohwi(w, wp, Cout, KH, KW, Cin, s);
  if (e == hipSuccess) e = launch_conv(p, s);
  hipStreamSynchronize(s);
  hipFree(wp);
  return e == hipSuccess ? D2T_OK : D2T_EHIP;
}

int d2t_op_conv2d_bf16x3(const float* x, const float* w, const float* bias, const float* residual, float* y, int32_t B,
                         int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t KH, int32_t KW, int32_t SH, int32_t SW,
                         int32_t PH, int32_t PW, int32_t act, d2t_stream stream) {
  if (!x || !w || !y || SH < 1 || SW < 1 || Cin % 32) return D2T_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const size_t n = (size_t)Cout * KH * KW * Cin;
  uint16_t *hi = nullptr, *lo = nullptr;
  float* wp = nullptr;
  if (hipMalloc(reinterpret_cast<void**>(&hi), n * 2) != hipSuccess) return D2T_ENOMEM;
  if (hipMalloc(reinterpret_cast<void**>(&lo), n * 2) != hipSuccess) { hipFree(hi); return D2T_ENOMEM; }
  if (hipMalloc(reinterpret_cast<void**>(&wp), n * 4) != hipSuccess) { hipFree(hi); hipFree(lo); return D2T_ENOMEM; }
  ConvP p{};
  p.in = x; p.w = wp; p.w_hi = hi; p.w_lo = lo; p.bias = bias; p.res = residual; p.out = y;
  p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
  p.OH = (H + 2 * PH - KH) / SH + 1; p.OW = (W + 2 * PW - KW) / SW + 1;
  p.KH = KH; p.KW = KW; p.SH = SH; p.SW = SW; p.PH = PH; p.PW = PW;
  p.M = B * p.OH * p.OW; p.K = KH * KW * Cin; p.act = act;
  hipError_t e = launch_repack_ohwi(w, wp, Cout, KH, KW, Cin, s);
  if (e == hipSuccess) e = launch_split_bf16(wp, hi, lo, n, s);
  if (e == hipSuccess) e = launch_conv_bf16x3(p, s);
  hipStreamSynchronize(s);
  hipFree(hi);
  hipFree(lo);
  hipFree(wp);
  return e == hipSuccess ? D2T_OK : D2T_EHIP;
}

// kernel selection of d2t_op_conv2d_bf16x3_split (process-wide; op-level tests and tools/conv_bench.py only)
static int g_op_conv_kind = 3, g_op_reserved_cus = 0;
int d2t_op_set_conv_kernel(int32_t kind, int32_t reserved_cus) {
  if ((kind != 0 && kind != 3 && kind != 8) || reserved_cus < 0 || reserved_cus > 128) return D2T_EINVAL;  // 0 / 3 as d2t_set_conv_kernel; 8: kind 3 in fp16x2 arithmetic (ConvP::f16)
  g_op_conv_kind = kind;
  g_op_reserved_cus = reserved_cus;
  return D2T_OK;
}

int d2t_op_conv2d_bf16x3_split(const float* x, const float* w, const float* bias, const float* residual, float* y,
                               int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t KH, int32_t KW,
                               int32_t SH, int32_t SW, int32_t PH, int32_t PW, int32_t act, d2t_stream stream) {
  // test entry for the split-activation kernel: input, residual and output travel as bf16 hi/lo records
  if (!x || !w || !y || SH < 1 || SW < 1 || Cin % 32 || Cout % 32) return D2T_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const int OH = (H + 2 * PH - KH) / SH + 1, OW = (W + 2 * PW - KW) / SW + 1;
  const size_t nw = (size_t)Cout * KH * KW * Cin, rx = (size_t)B * H * W, ry = (size_t)B * OH * OW;
  const size_t nx = rx * Cin, ny = ry * Cout;
  void* buf = nullptr;
  const size_t bytes = nw * 4 + nw * 4 + nx * 4 + ny * 4 + (residual ? ny * 4 : 0) + 256;
  if (hipMalloc(&buf, bytes) != hipSuccess) return D2T_ENOMEM;
  char* q = (char*)buf;
  float* wp = (float*)q; q += nw * 4;
  uint16_t* whi = (uint16_t*)q; q += nw * 2;
  uint16_t* wlo = (uint16_t*)q; q += nw * 2;
  uint16_t* xs = (uint16_t*)q; q += nx * 4;
  uint16_t* ys = (uint16_t*)q; q += ny * 4;
  uint16_t* rs = nullptr;
  if (residual) { rs = (uint16_t*)q; q += ny * 4; }
  void* zero = q;
  ConvP p{};
  p.w = wp; p.w_hi = whi; p.w_lo = wlo; p.bias = bias;
  p.in_hi = xs; p.out_hi = ys; p.res_hi = rs; p.zero16 = zero;
  const int f16 = g_op_conv_kind == 8;  // fp16 records, fp16 hi / lo weights, two MFMAs per product
  p.f16 = f16;
  p.pipelined = f16 ? 3 : g_op_conv_kind; p.reserved_cus = g_op_reserved_cus; p.split_tail = 1;
  p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.OH = OH; p.OW = OW;
  p.KH = KH; p.KW = KW; p.SH = SH; p.SW = SW; p.PH = PH; p.PW = PW;
  p.M = B * OH * OW; p.K = KH * KW * Cin; p.act = act;
  hipError_t e = hipMemsetAsync(zero, 0, 256, s);
  if (e == hipSuccess) e = launch_repack_ohwi(w, wp, Cout, KH, KW, Cin, s);
  if (e == hipSuccess) e = f16 ? launch_split_f16(wp, whi, wlo, nw, s) : launch_split_bf16(wp, whi, wlo, nw, s);
  if (e == hipSuccess) e = launch_split_act(x, xs, rx, Cin, s, f16);
  if (e == hipSuccess && residual) e = launch_split_act(residual, rs, ry, Cout, s, f16);
  if (e == hipSuccess) e = launch_conv_bf16x3(p, s);
  if (e == hipSuccess) e = launch_merge_act(ys, y, ry, Cout, s, f16);
  hipStreamSynchronize(s);
  hipFree(buf);
  return e == hipSuccess ? D2T_OK : D2T_EHIP;
}

int d2t_op_conv2d_bf16x3_split_pool(const float* x, const float* w, const float* bias, float* y,
                               int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t KH, int32_t KW,
                               int32_t SH, int32_t SW, int32_t PH, int32_t PW, int32_t act, d2t_stream stream) {
  // test entry for the fused 2x2 / stride 2 max-pool (ConvP::pool2): y is the POOLED map [B][OH/2][OW/2][Cout]
  const float* residual = nullptr;
  if (!x || !w || !y || SH < 1 || SW < 1 || Cin % 32 || Cout % 32 || (Cout > 64 && Cout < 128)) return D2T_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const int OH = (H + 2 * PH - KH) / SH + 1, OW = (W + 2 * PW - KW) / SW + 1;
  const size_t nw = (size_t)Cout * KH * KW * Cin, rx = (size_t)B * H * W, ry = (size_t)B * OH * OW;
  const size_t nx = rx * Cin, ny = ry * Cout;
  void* buf = nullptr;
  const size_t bytes = nw * 4 + nw * 4 + nx * 4 + ny * 4 + (residual ? ny * 4 : 0) + 256;
  if (hipMalloc(&buf, bytes) != hipSuccess) return D2T_ENOMEM;
  char* q = (char*)buf;
  float* wp = (float*)q; q += nw * 4;
  uint16_t* whi = (uint16_t*)q; q += nw * 2;
  uint16_t* wlo = (uint16_t*)q; q += nw * 2;
  uint16_t* xs = (uint16_t*)q; q += nx * 4;
  uint16_t* ys = (uint16_t*)q; q += ny * 4;
  uint16_t* rs = nullptr;
  if (residual) { rs = (uint16_t*)q; q += ny * 4; }
  void* zero = q;
  ConvP p{};
  p.w = wp; p.w_hi = whi; p.w_lo = wlo; p.bias = bias;
  p.in_hi = xs; p.out_hi = ys; p.res_hi = rs; p.zero16 = zero;
  p.reserved_cus = g_op_reserved_cus;
  p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.OH = OH; p.OW = OW;
  p.KH = KH; p.KW = KW; p.SH = SH; p.SW = SW; p.PH = PH; p.PW = PW;
  p.M = 4 * B * (OH / 2) * (OW / 2); p.K = KH * KW * Cin; p.act = act;
  p.pool2 = 1; p.pipelined = 3;
  const int f16 = g_op_conv_kind == 8;
  p.f16 = f16;
  const size_t rp = (size_t)B * (OH / 2) * (OW / 2);
  hipError_t e = hipMemsetAsync(zero, 0, 256, s);
  if (e == hipSuccess) e = launch_repack_ohwi(w, wp, Cout, KH, KW, Cin, s);
  if (e == hipSuccess) e = f16 ? launch_split_f16(wp, whi, wlo, nw, s) : launch_split_bf16(wp, whi, wlo, nw, s);
  if (e == hipSuccess) e = launch_split_act(x, xs, rx, Cin, s, f16);
  if (e == hipSuccess && residual) e = launch_split_act(residual, rs, ry, Cout, s, f16);
  void* wbuf = nullptr;
  if (e == hipSuccess) e = launch_conv_bf16x3(p, s);
  if (e == hipSuccess) e = launch_merge_act(ys, y, rp, Cout, s, f16);
  hipStreamSynchronize(s);
  hipFree(buf);
  return e == hipSuccess ? D2T_OK : D2T_EHIP;
}

int d2t_op_linear(const float* x, const float* w, const float* bias, const float* residual, float* y, int32_t M,
                  int32_t K, int32_t N, int32_t act, d2t_stream stream) {
  if (!x || !w || !y || K % 16) return D2T_EINVAL;
  LinW lw{w, bias, N, K};
  return linear_any(nullptr, (hipStream_t)stream, x, lw, residual, y, M, act) == hipSuccess ? D2T_OK : D2T_EHIP;
}

int d2t_op_maxpool2x2(const float* x, float* y, int32_t B, int32_t H, int32_t W, int32_t C, int32_t SH, int32_t SW,
                      int32_t PH, int32_t PW, d2t_stream stream) {
  if (!x || !y) return D2T_EINVAL;
  return launch_maxpool(x, y, B, H, W, C, SH, SW, PH, PW, (hipStream_t)stream) == hipSuccess ? D2T_OK : D2T_EHIP;
}

int d2t_op_layernorm(const float* x, const float* gamma, const float* beta, float* y, int32_t rows, int32_t D,
                     float eps, d2t_stream stream) {
  if (!x || !gamma || !beta || !y) return D2T_EINVAL;
  return launch_layernorm(x, gamma, beta, y, rows, D, eps, (hipStream_t)stream) == hipSuccess ? D2T_OK : D2T_EHIP;
}

int d2t_op_vit_attention(const float* qkv, float* y, int32_t B, int32_t N, int32_t heads, d2t_stream stream) {
  if (!qkv || !y) return D2T_EINVAL;
  return launch_vit_attention(qkv, y, B, N, heads, (hipStream_t)stream) == hipSuccess ? D2T_OK : D2T_EHIP;
}

int d2t_op_decode_attention(const float* q, const float* k, const float* v, float* y, int32_t B, int32_t heads,
                            int32_t hd, int32_t L, int32_t Lmax, d2t_stream stream) {
  if (!q || !k || !v || !y || L > Lmax) return D2T_EINVAL;
  DecAttnP p{};
  p.q = q; p.q_stride = heads * hd; p.k = const_cast<float*>(k); p.v = const_cast<float*>(v);
  p.y = y; p.y_stride = heads * hd; p.B = B; p.heads = heads; p.hd = hd; p.Lmax = Lmax; p.L = L;
  p.kv_batch_stride = (long long)heads * Lmax * hd;
  return launch_decode_attention(p, (hipStream_t)stream) == hipSuccess ? D2T_OK : D2T_EHIP;
}

}  // extern "C"
