// Split-bf16 implicit-GEMM convolution, pipelined form (the roofline kernel of the recognizer's backbone).
//
// Same operand layouts and K order as conv_bf16x3g_body (conv_bf16x3.hip): split-bf16 activation records in, three
// v_mfma_f32_16x16x32_bf16 per product (lo*hi, hi*lo, hi*hi, fp32 accumulate) -- or, on fp16 records (ConvP::f16), two
// v_mfma_f32_16x16x32_f16 (x*w_lo, x*w_hi).  How a CU is kept busy:
//
//   * block tile 256 x 128 x 32, 8 compute waves as 4 (M) x 2 (N), wave tile 64 x 64 = 4 x 4 MFMA blocks: 16 ds_read_b128 and
//     48 MFMAs per wave and K-step, 48 KB of L2 -> LDS traffic per 1536 MFMA cycles of a SIMD;
//   * ONE block per CU holding THREE LDS stages (144 KB): the LDS-DMA of K-step t+2 is issued while K-step t is computed,
//     the issuer waits only for its OWN pieces of step t with a counted `s_waitcnt vmcnt(N)` (the pieces of step t+1 stay
//     in flight), and a K-step costs one raw s_barrier -- no `vmcnt(0)` drain, no second barrier
//     (cdna_hip_programming.md "Pipelining across barriers": the 3-buffer span);
//   * LOADER WAVES: the block is 8 compute waves + 4 loader waves (one per SIMD).  A K-step's 48 KB go through the CU's
//     vector-memory path at 64 B/clk = 768 cycles -- half of the 1536 MFMA cycles a SIMD needs for the step.  When the
//     compute waves issue the LDS-DMA themselves they all sit in that queue at the same time (one block per CU: nobody
//     else has MFMAs to issue) and the two phases add up: measured 3090 cycles per K-step.  Loader waves own the whole
//     vector-memory side (addresses, LDS-DMA, counted waits); compute waves only ds_read and MFMA;
//   * STAGGER (MI355X_MICROARCH.md "Two waves per SIMD", item 9): the two compute waves that share a SIMD (w and w + 4) would
//     reach their fragment reads, their MFMAs and the barrier together -- both wait for LDS, then both want the matrix pipe.
//     Waves 4-7 therefore run half a K-step behind: they load the fragments of a step's second half (two of the four A row
//     blocks) BEFORE the next barrier and issue those MFMAs right AFTER it, while the SIMD partner waits for its first
//     fragments.  Every accumulator still sees the same MFMAs in the same order;
//   * persistent over tiles with a grid of (CUs - reserved), XCD-aware tile order.
//
// Hazards (checked against the rules of the guide):
//   RAW  a stage is read only after every wave has passed the barrier that follows its own counted vmcnt wait for that
//        stage's pieces ("read a staged buffer after the wait that retires it and a barrier the reader has passed");
//   WAR  stage (t+2)%3 == (t-1)%3 is overwritten by DMAs issued after barrier t; every wave has finished the ds_reads of
//        step t-1 before it arrives there (they feed MFMAs that precede the barrier in program order, and the compiler's
//        lgkmcnt waits sit in front of those MFMAs).
//
// Rounds 1-3 built this kernel in several other shapes -- 32x32x16 MFMAs, a patch-resident and a band-resident 3x3 form, a
// 256 x 256 tile on eight waves, a loader-less 8-wave form, Winograd F(2x2,3x3) -- all bit-identical or equally accurate, none
// faster in serving (docs/experiments_r03.md).  They left the tree in round 4 (git history: round 3).

#include "conv_common.h"

namespace d2t {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

namespace {
constexpr int PBK = 32;   // K-step
constexpr int PROW = 64;  // bytes per LDS row of one plane (32 bf16)
__device__ __forceinline__ int pswz(int row, int c) { return c ^ ((row >> 2) & 3); }
// the same for the 16x16x32 fragment pattern (lane -> row lane & 15, 16-byte k-chunk lane >> 4): the 16-lane service groups
// of a ds_read_b128 then pair rows 0-3 / 12-15 at chunk c with rows 4-11 at chunk c ^ 1, and XOR-ing bit 1 of the chunk
// with bit 3 of the row spreads every group over all 64 banks (tools/probe/lds_swizzle_check.py)
__device__ __forceinline__ int pswz16(int row, int c) { return c ^ ((row >> 2) & 2); }

template <int N>
__device__ __forceinline__ void wait_vm() {
  static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit field on gfx9");
  // s_waitcnt simm16: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt_hi[15:14]; leave expcnt / lgkmcnt untouched (max)
  __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));
}
#ifdef D2T_PROBES
// probe builds: thread 0 of block 0 (a compute wave that is not staggered) adds the time between consecutive marks of a tile
// (s_memrealtime, 10 ns ticks) to d2t_conv_phase[k]; [15] counts tiles.  tools/probe/conv_phases.py
__device__ unsigned long long d2t_conv_phase[16];
#define CONV_PHASE(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) { const unsigned long long now_ = __builtin_amdgcn_s_memrealtime(); \
    atomicAdd(&d2t_conv_phase[k], now_ - cphase_t_); cphase_t_ = now_; } } while (0)
#define CONV_PHASE_INIT() unsigned long long cphase_t_ = __builtin_amdgcn_s_memrealtime()
#define CONV_PHASE_TILE() do { if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&d2t_conv_phase[15], 1ull); } while (0)
#else
#define CONV_PHASE(k) do { } while (0)
#define CONV_PHASE_INIT() do { } while (0)
#define CONV_PHASE_TILE() do { } while (0)
#endif
}  // namespace

// Second phase of the wide epilogue (conv_common.h conv_epilogue_wide), run by EVERY thread of the block, loader waves
// included: the fp32 [BM][BN] tile is in LDS (32-float column blocks XOR-ed with bit 2 of the row); a thread owns four
// consecutive channels of one output row.  Same arithmetic per element in the same order as conv_epilogue:
// v = acc + bias; v += res | (res_hi + res_lo); activation; split.
template <int BM, int BN, int NT, int ROWS = BM>  // ROWS < BM: only the first ROWS rows of the tile are real output rows
__device__ __forceinline__ void epilogue_rows(const ConvP& p, const unsigned char* smem, int m0, int n0, int tid) {
  const float* tile = reinterpret_cast<const float*>(smem);
  constexpr int OPR = BN / 8;  // 8-channel octets per tile row: 16-byte accesses to each half of a record
  const uint16_t* __restrict__ res_hi = p.res_hi;
  const float* __restrict__ res = p.res;
  const float* __restrict__ bias = p.bias;
  uint16_t* __restrict__ out_hi = p.out_hi;
  float* __restrict__ out = p.out;
#pragma unroll 2
  for (int idx = tid; idx < ROWS * OPR; idx += NT) {
    const int row = idx / OPR, o = idx % OPR;
    const int m = m0 + row, n = n0 + o * 8;
    if (m >= p.M || n >= p.Cout) continue;
    const int col = (o * 8) ^ (((row >> 2) & 1) << 5);
    const float4 a0 = *reinterpret_cast<const float4*>(tile + row * BN + col);
    const float4 a1 = *reinterpret_cast<const float4*>(tile + row * BN + col + 4);
    float v[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
    if (bias) {
      const float4 b0 = *reinterpret_cast<const float4*>(bias + n), b1 = *reinterpret_cast<const float4*>(bias + n + 4);
      v[0] += b0.x, v[1] += b0.y, v[2] += b0.z, v[3] += b0.w, v[4] += b1.x, v[5] += b1.y, v[6] += b1.z, v[7] += b1.w;
    }
    const size_t off = (size_t)m * p.Cout + n;
    const size_t pi = plane_idx((size_t)m, n, p.Cout);
    if (res) {
      const float4 r0 = *reinterpret_cast<const float4*>(res + off), r1 = *reinterpret_cast<const float4*>(res + off + 4);
      v[0] += r0.x, v[1] += r0.y, v[2] += r0.z, v[3] += r0.w, v[4] += r1.x, v[5] += r1.y, v[6] += r1.z, v[7] += r1.w;
    }
    if (res_hi) {
      const uint4 rh = *reinterpret_cast<const uint4*>(res_hi + pi);
      const unsigned h[4] = {rh.x, rh.y, rh.z, rh.w};
      const int rf = res_fmt(p);
      if (rf == REC_F16) {  // one fp16 per element: the value is the hi half
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[2 * e] += f16_bits_to_f32((uint16_t)(h[e] & 0xFFFFu));
          v[2 * e + 1] += f16_bits_to_f32((uint16_t)(h[e] >> 16));
        }
      } else {
        const uint4 rl = *reinterpret_cast<const uint4*>(res_hi + pi + 32);
        const unsigned l[4] = {rl.x, rl.y, rl.z, rl.w};
        if (rf == REC_F16_PAIR) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[2 * e] += f16_bits_to_f32((uint16_t)(h[e] & 0xFFFFu)) + f16_bits_to_f32((uint16_t)(l[e] & 0xFFFFu));
            v[2 * e + 1] += f16_bits_to_f32((uint16_t)(h[e] >> 16)) + f16_bits_to_f32((uint16_t)(l[e] >> 16));
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[2 * e] += __uint_as_float(h[e] << 16) + __uint_as_float(l[e] << 16);
            v[2 * e + 1] += __uint_as_float(h[e] & 0xFFFF0000u) + __uint_as_float(l[e] & 0xFFFF0000u);
          }
        }
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = apply_act(v[e], p.act);
    if (out_hi) {
      uint16_t hi[8], lo[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) split_rec(v[e], hi[e], lo[e], out_fmt(p));
      uint4 oh, ol;
      oh.x = (unsigned)hi[0] | ((unsigned)hi[1] << 16), oh.y = (unsigned)hi[2] | ((unsigned)hi[3] << 16);
      oh.z = (unsigned)hi[4] | ((unsigned)hi[5] << 16), oh.w = (unsigned)hi[6] | ((unsigned)hi[7] << 16);
      ol.x = (unsigned)lo[0] | ((unsigned)lo[1] << 16), ol.y = (unsigned)lo[2] | ((unsigned)lo[3] << 16);
      ol.z = (unsigned)lo[4] | ((unsigned)lo[5] << 16), ol.w = (unsigned)lo[6] | ((unsigned)lo[7] << 16);
      *reinterpret_cast<uint4*>(out_hi + pi) = oh;
      if (out_fmt(p) != REC_F16) *reinterpret_cast<uint4*>(out_hi + pi + 32) = ol;  // (one-fp16 records: the lo half is never read)
    } else {
      *reinterpret_cast<float4*>(out + off) = make_float4(v[0], v[1], v[2], v[3]);
      *reinterpret_cast<float4*>(out + off + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
  }
}

// The fused 2x2 max-pool form of epilogue_rows (ConvP::pool2): tile rows 4r .. 4r+3 are one pooling window (pooled row order of
// the GEMM's rows); a thread owns eight channels of one POOLED row.  max, then bias, then activation -- the pool of the
// activated outputs, bit for bit (both monotone); no residual (the layers in front of a pool have none).
template <int BM, int BN, int NT>
__device__ __forceinline__ void epilogue_rows_pool(const ConvP& p, const unsigned char* smem, int m0, int n0, int tid) {
  const float* tile = reinterpret_cast<const float*>(smem);
  constexpr int OPR = BN / 8;
  const float* __restrict__ bias = p.bias;
  uint16_t* __restrict__ out_hi = p.out_hi;
  float* __restrict__ out = p.out;
  for (int idx = tid; idx < (BM / 4) * OPR; idx += NT) {
    const int pr = idx / OPR, o = idx % OPR;
    const int m = m0 + 4 * pr, n = n0 + o * 8;
    if (m >= p.M || n >= p.Cout) continue;
    const int col = (o * 8) ^ ((pr & 1) << 5);  // rows 4 pr .. 4 pr + 3 share (row >> 2) & 1 = pr & 1
    float v[8];
    {
      const float4 a0 = *reinterpret_cast<const float4*>(tile + (4 * pr) * BN + col);
      const float4 a1 = *reinterpret_cast<const float4*>(tile + (4 * pr) * BN + col + 4);
      v[0] = a0.x, v[1] = a0.y, v[2] = a0.z, v[3] = a0.w, v[4] = a1.x, v[5] = a1.y, v[6] = a1.z, v[7] = a1.w;
    }
#pragma unroll
    for (int k = 1; k < 4; ++k) {
      const float4 a0 = *reinterpret_cast<const float4*>(tile + (4 * pr + k) * BN + col);
      const float4 a1 = *reinterpret_cast<const float4*>(tile + (4 * pr + k) * BN + col + 4);
      v[0] = fmaxf(v[0], a0.x), v[1] = fmaxf(v[1], a0.y), v[2] = fmaxf(v[2], a0.z), v[3] = fmaxf(v[3], a0.w);
      v[4] = fmaxf(v[4], a1.x), v[5] = fmaxf(v[5], a1.y), v[6] = fmaxf(v[6], a1.z), v[7] = fmaxf(v[7], a1.w);
    }
    if (bias) {
      const float4 b0 = *reinterpret_cast<const float4*>(bias + n), b1 = *reinterpret_cast<const float4*>(bias + n + 4);
      v[0] += b0.x, v[1] += b0.y, v[2] += b0.z, v[3] += b0.w, v[4] += b1.x, v[5] += b1.y, v[6] += b1.z, v[7] += b1.w;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = apply_act(v[e], p.act);
    const size_t mp = (size_t)(m >> 2);
    const size_t pi = plane_idx(mp, n, p.Cout);
    if (out_hi) {
      uint16_t hi[8], lo[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) split_rec(v[e], hi[e], lo[e], out_fmt(p));
      uint4 oh, ol;
      oh.x = (unsigned)hi[0] | ((unsigned)hi[1] << 16), oh.y = (unsigned)hi[2] | ((unsigned)hi[3] << 16);
      oh.z = (unsigned)hi[4] | ((unsigned)hi[5] << 16), oh.w = (unsigned)hi[6] | ((unsigned)hi[7] << 16);
      ol.x = (unsigned)lo[0] | ((unsigned)lo[1] << 16), ol.y = (unsigned)lo[2] | ((unsigned)lo[3] << 16);
      ol.z = (unsigned)lo[4] | ((unsigned)lo[5] << 16), ol.w = (unsigned)lo[6] | ((unsigned)lo[7] << 16);
      *reinterpret_cast<uint4*>(out_hi + pi) = oh;
      if (out_fmt(p) != REC_F16) *reinterpret_cast<uint4*>(out_hi + pi + 32) = ol;  // (one-fp16 records: the lo half is never read)
    } else {
      *reinterpret_cast<float4*>(out + mp * p.Cout + n) = make_float4(v[0], v[1], v[2], v[3]);
      *reinterpret_cast<float4*>(out + mp * p.Cout + n + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
  }
}

// Element-wise epilogue of one wave's MI x NJ grid of 16x16 accumulators (the layers the wide epilogue does not take: row
// remap, positional add, Cout % 32 != 0): same arithmetic per element as conv_epilogue.
template <int MI, int NJ>
__device__ __forceinline__ void conv_epilogue16(const ConvP& p, float __attribute__((ext_vector_type(4))) (&acc)[MI][NJ], int mw, int nw,
                                                int r, int q) {
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int n = nw + j * 16 + r;
    if (n >= p.Cout) continue;
    const float bias = p.bias ? p.bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int m = mw + i * 16 + 4 * q + reg;
        if (m >= p.M) continue;
        float v = acc[i][j][reg] + bias;
        size_t row = (size_t)m;
        int in_img = 0;
        if (p.rows_per_img > 0) {
          const int img = m / p.rows_per_img;
          in_img = m - img * p.rows_per_img;
          row = (size_t)img * p.img_stride + p.row_off + in_img;
        }
        const size_t off = row * p.Cout + n;
        if (p.res) v += p.res[off];
        if (p.res_hi) {
          const size_t ri = plane_idx(row, n, p.Cout);
          v += join_rec(p.res_hi[ri], p.res_hi[ri + 32], res_fmt(p));
        }
        v = apply_act(v, p.act);
        if (p.row_add) v += p.row_add[(size_t)(p.row_add_off + in_img) * p.Cout + n];
        if (p.out_hi) {
          uint16_t hi, lo;
          split_rec(v, hi, lo, out_fmt(p));
          const size_t oi = plane_idx(row, n, p.Cout);
          p.out_hi[oi] = hi;
          p.out_hi[oi + 32] = lo;
        } else {
          p.out[off] = v;
        }
      }
  }
}

// The vector-memory side of a tile for ONE issuing wave: which 16-row pieces of the A / B planes it copies, their per-lane
// source addresses (XOR-swizzled k-chunk applied on the SOURCE: the LDS destination of a wave's LDS-DMA is linear) and the
// per-row validity mask over the filter taps (out-of-image taps and rows beyond M / Cout read a zero page).
// A operand: one LDS row per tile row holding the whole 128-byte record [hi 64 | lo 64]; a 1 KiB piece = 8 rows x 128 B, i.e.
// a wave's LDS-DMA touches 8 full cache lines instead of 16 half lines (the vector-memory path's cost is per line touched:
// 32-byte segments measured half the rate of 64-byte ones).  The eight 16-byte chunks of a row are XOR-ed with (row >> 1) & 7
// -- on the source side of the DMA and on the ds_read side -- so that the 16-lane groups of a ds_read_b128 hit every bank once.
__device__ __forceinline__ int aswz(int row, int c) { return c ^ ((row >> 1) & 7); }

// F16 (ConvP::f16): only the hi half of a record (the fp16 activation) is staged -- a piece is 16 rows x 64 B, the A plane
// BM x 64 B with the B planes' swizzle (pswz16)
template <int BM, int BN, int LW, int ABL, bool S16 = false, bool F16 = false>
struct DmaIssuer {
  static constexpr int AJ = BM / ((F16 ? 16 : 8) * LW), BJ = BN / (16 * LW);
  static constexpr int PLANE_A = BM * PROW, PLANE_B = BN * PROW, STAGE = (F16 ? 1 : 2) * PLANE_A + 2 * PLANE_B;
  static constexpr int A_BYTES = (F16 ? 1 : 2) * PLANE_A;
  static constexpr int PER_STEP = AJ + 2 * BJ;  // LDS-DMA instructions per K-step
  static_assert(AJ >= 1 && BJ >= 1, "tile too small for the issuing waves");
  static_assert(PER_STEP <= 31, "two K-steps of pieces must fit the 6-bit vmcnt");
  int a_off[AJ];
  int a_off2[AJ];  // second input (ConvP::in2_hi): element offset of this lane's chunk of the row's first record, -1 = no row
  unsigned a_mask[AJ];
  const uint16_t* b_hi[BJ];
  const uint16_t* b_lo[BJ];
  int kh, kw, c0, lw;

  __device__ __forceinline__ void setup(const ConvP& p, int m0, int n0, int lw_, int lane) {
    lw = lw_;
    kh = kw = c0 = 0;
    const int lr = lane >> 2, pos = lane & 3;
    const int ohow = p.OH * p.OW;
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
      const int row = F16 ? (lw * AJ + j) * 16 + (lane >> 2) : (lw * AJ + j) * 8 + (lane >> 3);
      const int c = F16 ? pswz16(row, lane & 3) : aswz(row, lane & 7);  // the chunk of the record that belongs at this lane's LDS position
      const int m = m0 + row;
      a_off[j] = 0;
      a_off2[j] = -1;
      a_mask[j] = 0;
      if (m < p.M) {
        if (p.Cin2) a_off2[j] = m * p.Cin2 * 2 + c * 8;
        int b, oh, ow;
        conv_row_coords(p, m, b, oh, ow);
        const int ih0 = oh * p.SH - p.PH, iw0 = ow * p.SW - p.PW;
        a_off[j] = ((b * p.H + ih0) * p.W + iw0) * p.Cin * 2 + c * 8;
        for (int y = 0; y < p.KH; ++y)
          for (int x = 0; x < p.KW; ++x)
            if ((unsigned)(ih0 + y) < (unsigned)p.H && (unsigned)(iw0 + x) < (unsigned)p.W) a_mask[j] |= 1u << (y * p.KW + x);
      }
    }
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
      const int row = (lw * BJ + j) * 16 + lr;
      const int c = S16 ? pswz16(row, pos) : pswz(row, pos);
      const int n = n0 + row;
      const bool ok = n < p.Cout;
      b_hi[j] = ok ? p.w_hi + (size_t)n * p.K + c * 8 : nullptr;
      b_lo[j] = ok ? p.w_lo + (size_t)n * p.K + c * 8 : nullptr;
    }
  }
  // K-steps must be issued in order (the tap / channel-chunk cursor advances)
  __device__ __forceinline__ void issue(const ConvP& p, unsigned char* smem, int kt, int buf) {
    unsigned char* ah = smem + buf * STAGE;
    unsigned char* bh = ah + A_BYTES;
    unsigned char* bl = bh + PLANE_B;
    const uint16_t* zero = reinterpret_cast<const uint16_t*>(p.zero16);
    const bool second = c0 >= p.Cin;  // the K-steps behind the filter taps: the 1x1 second input (wave-uniform)
    const int tap = kh * p.KW + kw;
    const int tapoff = second ? (c0 - p.Cin) * 2 : ((kh * p.W + kw) * p.Cin + c0) * 2;
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
      const uint16_t* src;
      if (second) src = a_off2[j] >= 0 ? p.in2_hi + (a_off2[j] + tapoff) : zero;
      else src = ((a_mask[j] >> tap) & 1u) ? p.in_hi + (a_off[j] + tapoff) : zero;
      const int piece = (lw * AJ + j) * 1024;
      if (ABL == 1) continue;
      __builtin_amdgcn_global_load_lds(src, (lds_ptr_t)(ah + piece), 16, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
      const int piece = (lw * BJ + j) * 1024;
      if (ABL == 1) continue;
      __builtin_amdgcn_global_load_lds(b_hi[j] ? b_hi[j] + (size_t)kt * PBK : zero, (lds_ptr_t)(bh + piece), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(b_lo[j] ? b_lo[j] + (size_t)kt * PBK : zero, (lds_ptr_t)(bl + piece), 16, 0, 0);
    }
    if (second) {
      c0 += 32;
    } else if (++kw == p.KW) {
      kw = 0;
      if (++kh == p.KH) { kh = 0; c0 += 32; }
    }
  }
};

// ---------------------------------------------------------------------------------------------------------------------
// The kernel body on v_mfma_f32_16x16x32_bf16 (round 3; rounds 1-2 used 32x32x16): one MFMA consumes the WHOLE K-step of
// 32 for a 16 x 16 output block (16 cycles on a SIMD) instead of half of it for a 32 x 32 block (32 cycles).  Per K-step a
// wave reads the same 16 fragments (8 of A: 4 row blocks x hi / lo; 8 of B) and issues 48 MFMAs with the same FLOPs and
// the same matrix-pipe cycles, but the chip holds a higher clock under this shape on random data (MI355X_MICROARCH.md
// "DVFS give-back" item 7: 1.12-1.15 x the FLOP/s of the 32x32x16 loop with every operand re-read from LDS) -- and this
// kernel is power-bound, not issue-bound (1.57 ms on random operands, 1.27 ms on zeros).
// Per output element still three products per K-step in the order lo*hi, hi*lo, hi*hi, fp32 accumulate; the summation
// INSIDE an MFMA now spans 32 k instead of 16, so results are not bit-identical to the 32x32x16 kernels (they agree to fp32
// rounding; every kernel that may compute rows of the same layer must use the same shape -- see launch_conv_bf16x3p).
// The stagger splits a K-step by ROW BLOCKS (first two of A, then the other two) since it can no longer be split by k:
// waves 4-7 carry the B fragments and the second pair of A fragments across the barrier.
typedef float f32x4v __attribute__((ext_vector_type(4)));

typedef _Float16 f16x8v __attribute__((ext_vector_type(8)));

// F16 (ConvP::f16, fp16x2 mode): A = the fp16 hi halves of the records only (12 instead of 16 fragment reads, 8 instead of
// 12 LDS-DMA pieces per loader and K-step), B = fp16 hi / lo planes, 32 MFMAs (x * w_lo, x * w_hi) instead of 48
template <int BM, int BN, int WM, int WN, int NL, bool STG, int ABL = 0, bool F16 = false>  // ABL (probe builds): 1 no LDS-DMA, 2 no MFMA, 4 LDS-DMA never awaited
__device__ __forceinline__ void conv_bf16x3p16_body(const ConvP& p, unsigned char* smem) {
  constexpr int NW = WM * WN, NT = (NW + NL) * 64;
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int MI = WTM / 16, NJ = WTN / 16, MH = MI / 2;
  constexpr int NSTG = 3;  // LDS stages in the ring
  constexpr bool UPFRONT = ABL != 8;  // (8: the reads as the compiler schedules them, for A/B timing in probe builds)
  static_assert(MI >= 2 && (MI & 1) == 0 && NJ >= 1 && NL > 0, "wave tile / loader configuration");
  using Issuer = DmaIssuer<BM, BN, NL, ABL == 1 ? 1 : 0, true, F16>;
  constexpr int PLANE_B = Issuer::PLANE_B, STAGE = Issuer::STAGE, PER_STEP = Issuer::PER_STEP, A_BYTES = Issuer::A_BYTES;
  static_assert(F16 || BM * BN * 4 <= 3 * STAGE, "the fp32 epilogue tile must fit in the staging area");  // (F16: the kernel allocates the tile's 128 KB)

  const int nt = (p.Cout + BN - 1) / BN;
  const int ntiles = nt * ((p.M - p.m_base + BM - 1) / BM);  // rows [m_base, M): the rows a 256 x 256 launch left over
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const bool loader = wave >= NW;  // wave-uniform
  const int KT = p.K / PBK;
  const int G = gridDim.x, xq = G >> 3, xr = G & 7, xcd = blockIdx.x & 7;
  const int slot = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (blockIdx.x >> 3);

  CONV_PHASE_INIT();
  for (int tile = slot; tile < ntiles; tile += G) {
    const int m0 = p.m_base + (tile / nt) * BM;
    const int n0 = (tile % nt) * BN;
    CONV_PHASE(0);  // previous tile's last barrier .. here (loop overhead; the first tile: kernel entry)
    CONV_PHASE_TILE();
    if (loader) {  // (the compute waves below run the same barrier sequence)
      Issuer dma;
      dma.setup(p, m0, n0, wave - NW, lane);
      dma.issue(p, smem, 0, 0);
      if (KT > 1) dma.issue(p, smem, 1, 1);
      int nxt2 = 2;
      for (int kt = 0; kt < KT; ++kt) {
        if (ABL != 4) { if (kt + 1 < KT) wait_vm<PER_STEP>(); else wait_vm<0>(); }
        __builtin_amdgcn_s_barrier();
        if (kt + 2 < KT) dma.issue(p, smem, kt + 2, nxt2);
        nxt2 = nxt2 == 2 ? 0 : nxt2 + 1;
      }
      if (ABL == 4) wait_vm<0>();
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_s_barrier();
      if (wide_epilogue_ok(p)) { if (p.pool2) epilogue_rows_pool<BM, BN, NT>(p, smem, m0, n0, tid); else epilogue_rows<BM, BN, NT>(p, smem, m0, n0, tid); }
      __builtin_amdgcn_s_barrier();
      continue;
    }
    const int wm = wave / WN, wn = wave % WN;
    const int r = lane & 15, q = lane >> 4;
    f32x4v acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};
    int offa[MI], offal[MI], offb[NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int row = wm * WTM + i * 16 + r;
      offa[i] = F16 ? row * PROW + pswz16(row, q) * 16 : row * 128 + aswz(row, q) * 16;
      offal[i] = row * 128 + aswz(row, 4 + q) * 16;  // (unused with F16)
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int row = wn * WTN + j * 16 + r;
      offb[j] = row * PROW + pswz16(row, q) * 16;
    }
    auto read_b = [&](const unsigned char* ah, bf16x8 (&fbh)[NJ], bf16x8 (&fbl)[NJ]) {
      const unsigned char* bh = ah + A_BYTES;
      const unsigned char* bl = bh + PLANE_B;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        fbh[j] = *reinterpret_cast<const bf16x8*>(bh + offb[j]);
        fbl[j] = *reinterpret_cast<const bf16x8*>(bl + offb[j]);
      }
    };
    auto read_a = [&](const unsigned char* ah, auto half_c, bf16x8 (&fah)[MH], bf16x8 (&fal)[MH]) {
      constexpr int half = decltype(half_c)::value;
#pragma unroll
      for (int i = 0; i < MH; ++i) {
        fah[i] = *reinterpret_cast<const bf16x8*>(ah + offa[half * MH + i]);
        if (!F16) fal[i] = *reinterpret_cast<const bf16x8*>(ah + offal[half * MH + i]);
      }
    };
    auto mma = [&](auto half_c, const bf16x8 (&fah)[MH], const bf16x8 (&fal)[MH], const bf16x8 (&fbh)[NJ], const bf16x8 (&fbl)[NJ]) {
      constexpr int half = decltype(half_c)::value;
#pragma unroll
      for (int i = 0; i < MH; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          if (ABL == 2) {  // keep the reads alive, drop the matrix work
            asm volatile("" ::"v"(fal[i]), "v"(fah[i]), "v"(fbh[j]), "v"(fbl[j]));
            continue;
          }
          f32x4v c = acc[half * MH + i][j];
          if (F16) {
            const f16x8v xa = __builtin_bit_cast(f16x8v, fah[i]);
            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(xa, __builtin_bit_cast(f16x8v, fbl[j]), c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(xa, __builtin_bit_cast(f16x8v, fbh[j]), c, 0, 0, 0);
            acc[half * MH + i][j] = c;
            continue;
          }
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fal[i], fbh[j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fah[i], fbl[j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fah[i], fbh[j], c, 0, 0, 0);
          acc[half * MH + i][j] = c;
        }
    };
    using H0 = std::integral_constant<int, 0>;
    using H1 = std::integral_constant<int, 1>;
    int cur = 0;
    const bool late = STG && wave >= NW / 2;
    if (!late) {
      for (int kt = 0; kt < KT; ++kt) {
        __builtin_amdgcn_s_barrier();  // stage kt (and kt + 1) complete for everyone; nobody reads the stages before
        if (kt == 0) CONV_PHASE(1);  // setup + the first two stages' LDS-DMA landed (pipeline fill)
        const unsigned char* ah = smem + cur * STAGE;
        // all sixteen fragment reads of the K-step go out before the first MFMA (the SIMD partner's carried MFMAs cover
        // their latency); left to itself the compiler interleaves them in three groups, each with its own wait
        bf16x8 fbh[NJ], fbl[NJ], fah[MH], fal[MH], fch[MH], fcl[MH];
        read_b(ah, fbh, fbl);
        read_a(ah, H0{}, fah, fal);
        read_a(ah, H1{}, fch, fcl);
        if (UPFRONT) __builtin_amdgcn_sched_barrier(0);
        mma(H0{}, fah, fal, fbh, fbl);
        mma(H1{}, fch, fcl, fbh, fbl);
        cur = cur == NSTG - 1 ? 0 : cur + 1;
      }
    } else {
      bf16x8 gah[MH], gal[MH], gbh[NJ], gbl[NJ];  // second-half A fragments and the step's B fragments, carried across the barrier
      auto step = [&](auto carried_c, int kt) {
        __builtin_amdgcn_s_barrier();
        const unsigned char* ah = smem + cur * STAGE;
        if (decltype(carried_c)::value) mma(H1{}, gah, gal, gbh, gbl);  // second half of the previous K-step
        __builtin_amdgcn_sched_barrier(0);  // (no reads of this step hoisted above: the carried fragments die first)
        read_b(ah, gbh, gbl);
        {
          bf16x8 fah[MH], fal[MH];
          read_a(ah, H0{}, fah, fal);
          if (UPFRONT) {  // the carried half's fragments are requested before the first half's MFMAs, not after them
            read_a(ah, H1{}, gah, gal);
            __builtin_amdgcn_sched_barrier(0);
          }
          mma(H0{}, fah, fal, gbh, gbl);
        }
        if (!UPFRONT) read_a(ah, H1{}, gah, gal);
        // the reads have returned before this wave arrives at the next barrier (after it the stage may be overwritten)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        cur = cur == NSTG - 1 ? 0 : cur + 1;
      };
      step(H0{}, 0);  // peeled: nothing carried into K-step 0
      for (int kt = 1; kt < KT; ++kt) step(H1{}, kt);
      mma(H1{}, gah, gal, gbh, gbl);
    }
    CONV_PHASE(2);  // the K loop of this wave
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // every wave is done with the last stages: the staging area becomes the epilogue's fp32 tile
    CONV_PHASE(3);  // waiting for the other waves (staggered partners finish half a K-step later)
    // C/D map of v_mfma_f32_16x16x32: col = lane & 15 -> n, row = 4 * (lane >> 4) + reg -> m
    if (wide_epilogue_ok(p)) {  // block-uniform
      float* tile_f = reinterpret_cast<float*>(smem);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) {
            const int row = wm * WTM + i * 16 + 4 * q + reg;
            const int col = (wn * WTN + j * 16 + r) ^ (((row >> 2) & 1) << 5);
            tile_f[row * BN + col] = acc[i][j][reg];
          }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      CONV_PHASE(4);  // accumulators to the LDS tile + barrier
      if (p.pool2) epilogue_rows_pool<BM, BN, NT>(p, smem, m0, n0, tid);
      else epilogue_rows<BM, BN, NT>(p, smem, m0, n0, tid);
      CONV_PHASE(5);  // this thread's rows: LDS reads, bias / residual / activation / split, stores issued
    } else {
      __builtin_amdgcn_s_barrier();
      conv_epilogue16<MI, NJ>(p, acc, m0 + wm * WTM, n0 + wn * WTN, r, q);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // the tile is staging memory again (next tile's LDS-DMA)
    CONV_PHASE(6);  // waiting for the slowest thread's epilogue
  }
}


// non-template entry points (the host-side stub of a __global__ template using the LDS-DMA builtin is not emitted)
__global__ __launch_bounds__(768, 3) void conv_bf16x3p16_256x128_s(const ConvP p) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[3 * (2 * 256 * PROW + 2 * 128 * PROW)];
  conv_bf16x3p16_body<256, 128, 4, 2, 4, true>(p, smem);
}
// the dominant GEMM shape (512 -> 512 channels, 3x3: K = 4608) under its own symbol, so that rocprofv3's per-kernel rows
// separate it from the other layers
__global__ __launch_bounds__(768, 3) void conv_bf16x3p16_256x128_s_k4608(const ConvP p) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[3 * (2 * 256 * PROW + 2 * 128 * PROW)];
  conv_bf16x3p16_body<256, 128, 4, 2, 4, true>(p, smem);
}

// 64 x 128 tile on the same body (eight 64 x 16 wave tiles + four loaders, 24 KB stages): the rows of a last, sparsely
// filled round of 256-row tiles (launch_conv_bf16x3p).  Same MFMA shape, K order and products: bit-identical to the 256 x 128
// build, so which of the two computes a row never shows in the values.
__global__ __launch_bounds__(768, 3) void conv_bf16x3p16_64x128_tail(const ConvP p) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[3 * (2 * 64 * PROW + 2 * 128 * PROW)];
  conv_bf16x3p16_body<64, 128, 1, 8, 4, false>(p, smem);
}
__global__ __launch_bounds__(768, 3) void conv_f16x2p16_64x128_tail(const ConvP p) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[3 * (64 * PROW + 2 * 128 * PROW)];  // 20 KB stages (>= the 32 KB fp32 tile)
  conv_bf16x3p16_body<64, 128, 1, 8, 4, false, 0, true>(p, smem);
}

// fp16x2 builds of the pipelined kernel (ConvP::f16): 32 KB stages; the 128 KB are the epilogue's fp32 tile
__global__ __launch_bounds__(768, 3) void conv_f16x2p16_256x128_s(const ConvP p) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[256 * 128 * 4];
  conv_bf16x3p16_body<256, 128, 4, 2, 4, true, 0, true>(p, smem);
}
__global__ __launch_bounds__(768, 3) void conv_f16x2p16_256x128_s_k4608(const ConvP p) {  // the dominant shape, own symbol
  __shared__ __attribute__((aligned(1024))) unsigned char smem[256 * 128 * 4];
  conv_bf16x3p16_body<256, 128, 4, 2, 4, true, 0, true>(p, smem);
}

#ifdef D2T_PROBES  // ablation probes of the 16x16x32 kernel (D2T_CONV_ABL=1|2|4): results are garbage by construction
template <int ABL>
__global__ __launch_bounds__(768, 3) void conv_bf16x3p16_probe(const ConvP p) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[3 * (2 * 256 * PROW + 2 * 128 * PROW)];
  conv_bf16x3p16_body<256, 128, 4, 2, 4, true, ABL>(p, smem);
}
#endif


static int abl_probe() {
  static const int abl = D2T_PROBE_ENV("D2T_CONV_ABL");
  return abl;
}

// grid of the pipelined kernel: one block per CU on (CUs - reserved) CUs, never more blocks than tiles
hipError_t launch_conv_bf16x3p(const ConvP& p, hipStream_t s) {
  static int cus = 0;
  if (!cus) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1)
      return hipErrorInvalidDevice;
    cus = n;
  }
  if (p.pipelined != 3) return hipErrorInvalidValue;
  ConvP q = p;
  q.wave_prio = D2T_PROBE_ENV("D2T_CONV_PRIO");
  const int nt = (p.Cout + 127) / 128;
  int grid = cus - (p.reserved_cus > 0 ? p.reserved_cus : 0);
  if (grid < 8) grid = 8;
  int tiles = ((p.M - p.m_base + 255) / 256) * nt;
  // Whole rounds only: when the last round of 256-row tiles would be less than half full (the dominant layer at B = 64: 2064
  // tiles = eight rounds + 16 tiles; at B = 32: 4.03 rounds; config C1: 260 tiles), the rows behind the whole rounds go to the
  // 64 x 128 build of the same body, whose small tiles spread over the chip.  Same MFMA shape, K order and products per output
  // element in both kernels (tests assert bit-identity), so which of them computes a row never shows in the values.  The
  // caller switches it off (split_tail = 0) while decode loops are in flight: their kernels run in exactly that hole.
  static const float tail_frac = D2T_PROBE_ENV_STR("D2T_CONV_TAIL") ? (float)atof(D2T_PROBE_ENV_STR("D2T_CONV_TAIL")) : 0.5f;
  int tail_from = -1;
  if (p.split_tail && p.m_base == 0 && !abl_probe()) {
    const int rounds = tiles / grid, rem = tiles - rounds * grid;
    if (rounds >= 1 && rem > 0 && rem < tail_frac * grid) {
      const int main_mt = rounds * grid / nt;  // whole rows of tiles that fit into the whole rounds
      tail_from = main_mt * 256;
      q.M = tail_from;
      tiles = main_mt * nt;
    }
  }
  if (grid > tiles) grid = tiles;
  const bool dom = p.K == 4608 && p.Cout == 512;
#ifdef D2T_PROBES
  if (abl_probe() && !p.f16) {
    switch (abl_probe()) {
      case 1: hipLaunchKernelGGL(conv_bf16x3p16_probe<1>, dim3(grid), dim3(768), 0, s, q); break;
      case 2: hipLaunchKernelGGL(conv_bf16x3p16_probe<2>, dim3(grid), dim3(768), 0, s, q); break;
      case 8: hipLaunchKernelGGL(conv_bf16x3p16_probe<8>, dim3(grid), dim3(768), 0, s, q); break;
      default: hipLaunchKernelGGL(conv_bf16x3p16_probe<4>, dim3(grid), dim3(768), 0, s, q); break;
    }
    return hipGetLastError();
  }
#endif
  if (p.f16) {  // fp16 records in: two MFMAs per product
    if (dom) hipLaunchKernelGGL(conv_f16x2p16_256x128_s_k4608, dim3(grid), dim3(768), 0, s, q);
    else hipLaunchKernelGGL(conv_f16x2p16_256x128_s, dim3(grid), dim3(768), 0, s, q);
  } else {
    if (dom) hipLaunchKernelGGL(conv_bf16x3p16_256x128_s_k4608, dim3(grid), dim3(768), 0, s, q);
    else hipLaunchKernelGGL(conv_bf16x3p16_256x128_s, dim3(grid), dim3(768), 0, s, q);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess || tail_from < 0) return e;
  ConvP t = p;
  t.wave_prio = q.wave_prio;
  t.m_base = tail_from;
  const int tt = ((p.M - tail_from + 63) / 64) * nt;
  if (p.f16) hipLaunchKernelGGL(conv_f16x2p16_64x128_tail, dim3(tt), dim3(768), 0, s, t);
  else hipLaunchKernelGGL(conv_bf16x3p16_64x128_tail, dim3(tt), dim3(768), 0, s, t);
  return hipGetLastError();
}

#ifdef D2T_PROBES
}  // namespace d2t
extern "C" int d2t_debug_conv_phases(unsigned long long* out, int reset) {  // probe builds: read (and clear) d2t_conv_phase
  hipDeviceSynchronize();
  if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(d2t::d2t_conv_phase), 16 * 8) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[16] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(d2t::d2t_conv_phase), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
namespace d2t {
#endif
}  // namespace d2t
