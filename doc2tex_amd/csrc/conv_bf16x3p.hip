// Split-bf16 implicit-GEMM convolution, pipelined form (the roofline kernel of the recognizer's backbone).
//
// Same arithmetic, operand layouts and K order as conv_bf16x3g_body (conv_bf16x3.hip): split-bf16 activation records
// in, three v_mfma_f32_32x32x16_bf16 per product (lo*hi, hi*lo, hi*hi, fp32 accumulate) -- an output element goes through
// the SAME sequence of MFMA accumulations, so results are bit-identical to that kernel (tests assert it).  What differs is
// how a CU is kept busy:
//
//   * block tile 256 x 128 x 32, 8 waves as 4 (M) x 2 (N), wave tile 64 x 64: 16 ds_read_b128 per 24 MFMAs
//     (0.67 per MFMA instead of 1.0) and 48 KB of L2 -> LDS traffic per 1536 MFMA cycles of a SIMD instead of 64 KB;
//   * ONE block per CU holding THREE LDS stages (144 KB): the LDS-DMA of K-step t+2 is issued while K-step t is computed,
//     the issuer waits only for its OWN pieces of step t with a counted `s_waitcnt vmcnt(N)` (the pieces of step t+1 stay
//     in flight), and a K-step costs one raw s_barrier -- no `vmcnt(0)` drain, no second barrier
//     (cdna_hip_programming.md "Pipelining across barriers": the 3-buffer span);
//   * LOADER WAVES: the block is 8 compute waves + 4 loader waves (one per SIMD).  A K-step's 48 KB go through the CU's
//     vector-memory path at 64 B/clk = 768 cycles -- half of the 1536 MFMA cycles a SIMD needs for the step.  When the
//     compute waves issue the LDS-DMA themselves they all sit in that queue at the same time (one block per CU: nobody
//     else has MFMAs to issue) and the two phases add up: measured 3090 cycles per K-step.  Loader waves own the whole
//     vector-memory side (addresses, LDS-DMA, counted waits); compute waves only ds_read and MFMA;
//   * persistent over tiles with a grid of (CUs - reserved): whole rounds of tiles for the dominant layer
//     (2064 tiles = 9 x 229.3) and the remaining CUs are free for the latency-bound decode kernels of the previous batch.
//
// Hazards (checked against the rules of the guide):
//   RAW  a stage is read only after every wave has passed the barrier that follows its own counted vmcnt wait for that
//        stage's pieces ("read a staged buffer after the wait that retires it and a barrier the reader has passed");
//   WAR  stage (t+2)%3 == (t-1)%3 is overwritten by DMAs issued after barrier t; every wave has finished the ds_reads of
//        step t-1 before it arrives there (they feed MFMAs that precede the barrier in program order, and the compiler's
//        lgkmcnt waits sit in front of those MFMAs).
#include <cstdlib>
#include <type_traits>

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
      if (p.f16) {  // fp16 records: the value is the hi half
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[2 * e] += f16_bits_to_f32((uint16_t)(h[e] & 0xFFFFu));
          v[2 * e + 1] += f16_bits_to_f32((uint16_t)(h[e] >> 16));
        }
      } else {
        const uint4 rl = *reinterpret_cast<const uint4*>(res_hi + pi + 32);
        const unsigned l[4] = {rl.x, rl.y, rl.z, rl.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[2 * e] += __uint_as_float(h[e] << 16) + __uint_as_float(l[e] << 16);
          v[2 * e + 1] += __uint_as_float(h[e] & 0xFFFF0000u) + __uint_as_float(l[e] & 0xFFFF0000u);
        }
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = apply_act(v[e], p.act);
    if (out_hi) {
      uint16_t hi[8], lo[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) split_rec(v[e], hi[e], lo[e], p.f16);
      uint4 oh, ol;
      oh.x = (unsigned)hi[0] | ((unsigned)hi[1] << 16), oh.y = (unsigned)hi[2] | ((unsigned)hi[3] << 16);
      oh.z = (unsigned)hi[4] | ((unsigned)hi[5] << 16), oh.w = (unsigned)hi[6] | ((unsigned)hi[7] << 16);
      ol.x = (unsigned)lo[0] | ((unsigned)lo[1] << 16), ol.y = (unsigned)lo[2] | ((unsigned)lo[3] << 16);
      ol.z = (unsigned)lo[4] | ((unsigned)lo[5] << 16), ol.w = (unsigned)lo[6] | ((unsigned)lo[7] << 16);
      *reinterpret_cast<uint4*>(out_hi + pi) = oh;
      if (!p.f16) *reinterpret_cast<uint4*>(out_hi + pi + 32) = ol;  // (fp16 records: the lo half is never read)
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
      for (int e = 0; e < 8; ++e) split_rec(v[e], hi[e], lo[e], p.f16);
      uint4 oh, ol;
      oh.x = (unsigned)hi[0] | ((unsigned)hi[1] << 16), oh.y = (unsigned)hi[2] | ((unsigned)hi[3] << 16);
      oh.z = (unsigned)hi[4] | ((unsigned)hi[5] << 16), oh.w = (unsigned)hi[6] | ((unsigned)hi[7] << 16);
      ol.x = (unsigned)lo[0] | ((unsigned)lo[1] << 16), ol.y = (unsigned)lo[2] | ((unsigned)lo[3] << 16);
      ol.z = (unsigned)lo[4] | ((unsigned)lo[5] << 16), ol.w = (unsigned)lo[6] | ((unsigned)lo[7] << 16);
      *reinterpret_cast<uint4*>(out_hi + pi) = oh;
      if (!p.f16) *reinterpret_cast<uint4*>(out_hi + pi + 32) = ol;  // (fp16 records: the lo half is never read)
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
          v += join_rec(p.res_hi[ri], p.res_hi[ri + 32], p.f16);
        }
        v = apply_act(v, p.act);
        if (p.row_add) v += p.row_add[(size_t)(p.row_add_off + in_img) * p.Cout + n];
        if (p.out_hi) {
          uint16_t hi, lo;
          split_rec(v, hi, lo, p.f16);
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

// block tile BM x BN, compute-wave grid WM x WN, NL dedicated loader waves (0: the compute waves issue the LDS-DMA themselves,
// each its share, right after the K-step's barrier); ABL: ablation probes (1 no DMA, 2 no MFMA, 3 no ds_read / MFMA)
// STG (stagger, MI355X_MICROARCH.md "Two waves per SIMD", item 9): all eight compute waves run the same program with one
// barrier per K-step, so the two waves that share a SIMD (w and w + 4) reach their fragment reads, their MFMAs and the barrier
// together -- both wait for LDS, then both want the matrix pipe.  With STG the second-dispatched half (waves 4-7) runs half
// a K-step behind: it loads the fragments of a step's second half BEFORE the next barrier (the reads have returned when it
// arrives there, so the stage may be overwritten as before) and issues those MFMAs right AFTER the barrier, while its SIMD
// partner waits for its first fragments; later it reads while the partner multiplies.  Every accumulator still sees the
// same MFMAs in the same order: results are bit-identical.
template <int BM, int BN, int WM, int WN, int NL, int ABL = 0, bool STG = false>
__device__ __forceinline__ void conv_bf16x3p_body(const ConvP& p, unsigned char* smem) {
  constexpr int NW = WM * WN, NT = (NW + NL) * 64;
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int MI = WTM / 32, NJ = WTN / 32;
  static_assert(MI >= 1 && NJ >= 1, "tile too small for the wave grid");
  using Issuer = DmaIssuer<BM, BN, (NL > 0 ? NL : NW), ABL>;
  constexpr int PLANE_A = Issuer::PLANE_A, PLANE_B = Issuer::PLANE_B, STAGE = Issuer::STAGE, PER_STEP = Issuer::PER_STEP;
  static_assert(BM * BN * 4 <= 3 * STAGE, "the fp32 epilogue tile must fit in the staging area");

  const int nt = (p.Cout + BN - 1) / BN;
  const int ntiles = nt * ((p.M + BM - 1) / BM);
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const bool loader = NL > 0 && wave >= NW;  // wave-uniform
  if (p.wave_prio == 1) __builtin_amdgcn_s_setprio(1);
  else if (p.wave_prio == 2) __builtin_amdgcn_s_setprio(2);
  else if (p.wave_prio == 3) __builtin_amdgcn_s_setprio(3);
  const int KT = p.K / PBK;

  // Tile order: in every round the blocks that share an XCD (equal blockIdx % 8) take consecutive tiles = the column
  // tiles of the same pixels and the neighbouring rows, so they share that XCD's L2 (bijective for any grid size).
  const int G = gridDim.x, xq = G >> 3, xr = G & 7, xcd = blockIdx.x & 7;
  const int slot = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (blockIdx.x >> 3);

  for (int tile = slot; tile < ntiles; tile += G) {
    const int m0 = (tile / nt) * BM;
    const int n0 = (tile % nt) * BN;

    // Every wave executes exactly the same sequence of barriers per tile: one per K-step (barrier kt: "stage kt is complete,
    // stage kt-1 is free"), one before the epilogue, and the epilogue's own two.
    if (loader) {
      Issuer dma;
      dma.setup(p, m0, n0, wave - NW, lane);
      dma.issue(p, smem, 0, 0);
      if (KT > 1) dma.issue(p, smem, 1, 1);
      int nxt2 = 2;  // stage of K-step kt+2
      for (int kt = 0; kt < KT; ++kt) {
        // this wave's pieces of K-step kt have landed (those of kt+1 may still be in flight) ...
        if (kt + 1 < KT) wait_vm<PER_STEP>(); else wait_vm<0>();
        __builtin_amdgcn_s_barrier();  // ... and so have the other loaders'; the compute waves are done reading K-step kt-1
        if (kt + 2 < KT) dma.issue(p, smem, kt + 2, nxt2);
        nxt2 = nxt2 == 2 ? 0 : nxt2 + 1;
      }
      __builtin_amdgcn_s_barrier();  // (compute waves: done with the last stage)
      __builtin_amdgcn_s_barrier();  // (compute waves: accumulators are in the LDS tile)
      if (wide_epilogue_ok(p)) epilogue_rows<BM, BN, NT>(p, smem, m0, n0, tid);
      __builtin_amdgcn_s_barrier();  // the tile is staging memory again
      continue;
    }

    Issuer dma;  // NL == 0 only: this compute wave's share of the LDS-DMA
    if (NL == 0) {
      dma.setup(p, m0, n0, wave, lane);
      dma.issue(p, smem, 0, 0);
      if (KT > 1) dma.issue(p, smem, 1, 1);
    }
    const int wm = wave / WN, wn = wave % WN;
    const int r = lane & 31, h = lane >> 5;
    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // ds_read byte offsets of this lane's fragments, for the two 16-deep halves of a K-step: A rows are 128-byte records
    // (hi chunks 0..3, lo chunks 4..7, swizzled by aswz), B rows 64 bytes per plane (pswz)
    int offa[2][MI], offal[2][MI], offb[2][NJ];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int c = 2 * kk + h;
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int row = wm * WTM + i * 32 + r;
        offa[kk][i] = row * 128 + aswz(row, c) * 16;
        offal[kk][i] = row * 128 + aswz(row, 4 + c) * 16;
      }
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int row = wn * WTN + j * 32 + r;
        offb[kk][j] = row * PROW + pswz(row, c) * 16;
      }
    }
    int cur = 0, nxt2 = 2;  // stage of K-step kt / kt+2
    auto read_half = [&](const unsigned char* ah, int kk, bf16x8 (&fah)[MI], bf16x8 (&fal)[MI], bf16x8 (&fbh)[NJ], bf16x8 (&fbl)[NJ]) {
      const unsigned char* bh = ah + 2 * PLANE_A;
      const unsigned char* bl = bh + PLANE_B;
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        fah[i] = *reinterpret_cast<const bf16x8*>(ah + offa[kk][i]);
        fal[i] = *reinterpret_cast<const bf16x8*>(ah + offal[kk][i]);
      }
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        fbh[j] = *reinterpret_cast<const bf16x8*>(bh + offb[kk][j]);
        fbl[j] = *reinterpret_cast<const bf16x8*>(bl + offb[kk][j]);
      }
    };
    auto mma_half = [&](const bf16x8 (&fah)[MI], const bf16x8 (&fal)[MI], const bf16x8 (&fbh)[NJ], const bf16x8 (&fbl)[NJ]) {
      if (ABL == 2) {  // keep the reads alive, drop the matrix work
#pragma unroll
        for (int i = 0; i < MI; ++i) asm volatile("" ::"v"(fah[i]), "v"(fal[i]));
#pragma unroll
        for (int j = 0; j < NJ; ++j) asm volatile("" ::"v"(fbh[j]), "v"(fbl[j]));
        return;
      }
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fal[i], fbh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fah[i], fbl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fah[i], fbh[j], acc[i][j], 0, 0, 0);
        }
    };
    const bool late = STG && wave >= NW / 2;  // wave-uniform: the half of the block that runs half a K-step behind
    // two whole loops, not a branch inside one: an `if` around MFMAs that update the accumulators makes the compiler keep
    // two copies of them (64 VGPRs each) and move them back and forth
    if (!late) {
      for (int kt = 0; kt < KT; ++kt) {
        if (NL == 0) {  // this wave's own pieces of K-step kt have landed (those of kt+1 may still be in flight)
          if (kt + 1 < KT) wait_vm<PER_STEP>(); else wait_vm<0>();
        }
        __builtin_amdgcn_s_barrier();  // stage kt is complete for everyone; nobody reads stage kt-1 any more
        if (NL == 0 && kt + 2 < KT) dma.issue(p, smem, kt + 2, nxt2);
        __builtin_amdgcn_sched_barrier(0);  // keep the DMA issue ahead of the ds_reads / MFMAs
        const unsigned char* ah = smem + cur * STAGE;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          if (ABL == 3) continue;
          bf16x8 fah[MI], fal[MI], fbh[NJ], fbl[NJ];
          read_half(ah, kk, fah, fal, fbh, fbl);
          mma_half(fah, fal, fbh, fbl);
        }
        cur = cur == 2 ? 0 : cur + 1;
        nxt2 = nxt2 == 2 ? 0 : nxt2 + 1;
      }
    } else {
      bf16x8 gah[MI], gal[MI], gbh[NJ], gbl[NJ];  // second-half fragments carried across the barrier
      auto step = [&](bool carried) {
        __builtin_amdgcn_s_barrier();
        const unsigned char* ah = smem + cur * STAGE;
        if (carried) mma_half(gah, gal, gbh, gbl);  // second half of the previous K-step, loaded before this barrier
        __builtin_amdgcn_sched_barrier(0);          // (no reads of this step hoisted above: the carried fragments die first)
        {
          bf16x8 fah[MI], fal[MI], fbh[NJ], fbl[NJ];
          read_half(ah, 0, fah, fal, fbh, fbl);
          mma_half(fah, fal, fbh, fbl);
        }
        read_half(ah, 1, gah, gal, gbh, gbl);
        // the reads have returned before this wave arrives at the next barrier (after it the stage may be overwritten)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        cur = cur == 2 ? 0 : cur + 1;
      };
      step(false);  // peeled: nothing carried into K-step 0 (a branch around accumulator updates would duplicate them)
      for (int kt = 1; kt < KT; ++kt) step(true);
      mma_half(gah, gal, gbh, gbl);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // every wave is done with the last stages: the staging area becomes the epilogue's fp32 tile
    if (wide_epilogue_ok(p)) {  // block-uniform
      float* tile_f = reinterpret_cast<float*>(smem);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int reg = 0; reg < 16; ++reg) {
            const int row = wm * WTM + i * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
            const int col = (wn * WTN + j * 32 + r) ^ (((row >> 2) & 1) << 5);
            tile_f[row * BN + col] = acc[i][j][reg];
          }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      epilogue_rows<BM, BN, NT>(p, smem, m0, n0, tid);
    } else {
      __builtin_amdgcn_s_barrier();
      conv_epilogue<MI, NJ>(p, acc, m0 + wm * WTM, n0 + wn * WTN, r, h);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // the tile is staging memory again (next tile's LDS-DMA)
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same kernel on v_mfma_f32_16x16x32_bf16 (round 3).  Block tile, LDS stages, loader waves, counted waits, barriers and
// tile order are those of conv_bf16x3p_body; what changes is the matrix instruction: one MFMA consumes the WHOLE K-step of
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
// KPB = 2 (an experiment for the fp16x2 builds: 32 KB stages, four of them): ONE barrier per TWO K-steps, on the hypothesis
// that a barrier interval has a fixed cost of 550-650 cycles in either arithmetic (K-step 2200 cycles for 1536 of MFMAs in
// split-bf16, 1580 for 1024 in fp16x2).  Measured: correct, and slower -- dominant layer 1.12 vs 1.04 ms in situ, 1786 vs
// 1889 formulas/s (the loaders then issue sixteen pieces in one burst).  Not instantiated.
template <int BM, int BN, int WM, int WN, int NL, bool STG, int ABL = 0, bool F16 = false, int KPB = 1>  // ABL (probe builds): 1 no LDS-DMA, 2 no MFMA, 4 LDS-DMA never awaited
__device__ __forceinline__ void conv_bf16x3p16_body(const ConvP& p, unsigned char* smem) {
  constexpr int NW = WM * WN, NT = (NW + NL) * 64;
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int MI = WTM / 16, NJ = WTN / 16, MH = MI / 2;
  constexpr int NSTG = KPB == 2 ? 4 : 3;  // LDS stages in the ring
  static_assert(KPB == 1 || KPB == 2, "one or two K-steps per barrier");
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

  for (int tile = slot; tile < ntiles; tile += G) {
    const int m0 = p.m_base + (tile / nt) * BM;
    const int n0 = (tile % nt) * BN;
    if (loader) {  // identical to conv_bf16x3p_body's loader (same barrier sequence)
      Issuer dma;
      dma.setup(p, m0, n0, wave - NW, lane);
      dma.issue(p, smem, 0, 0);
      if (KT > 1) dma.issue(p, smem, 1, 1);
      int nxt2 = 2;
      if (KPB == 2) {
        for (int kt = 0; kt < KT; kt += 2) {
          wait_vm<0>();                  // the pair of stages (kt, kt + 1) has landed ...
          __builtin_amdgcn_s_barrier();  // ... for every loader; the compute waves are done with the pair before it
          if (kt + 2 < KT) dma.issue(p, smem, kt + 2, (kt + 2) & 3);
          if (kt + 3 < KT) dma.issue(p, smem, kt + 3, (kt + 3) & 3);
        }
      } else
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
        if (KPB == 1 || (kt & 1) == 0) __builtin_amdgcn_s_barrier();  // stage kt (and kt + 1) complete for everyone; nobody reads the stages before
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
        if (KPB == 1 || (kt & 1) == 0) __builtin_amdgcn_s_barrier();
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
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // every wave is done with the last stages: the staging area becomes the epilogue's fp32 tile
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
      if (p.pool2) epilogue_rows_pool<BM, BN, NT>(p, smem, m0, n0, tid);
      else epilogue_rows<BM, BN, NT>(p, smem, m0, n0, tid);
    } else {
      __builtin_amdgcn_s_barrier();
      conv_epilogue16<MI, NJ>(p, acc, m0 + wm * WTM, n0 + wn * WTN, r, q);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // the tile is staging memory again (next tile's LDS-DMA)
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// 256 x 256 block tile on eight compute waves (round 3).  Ablations of conv_bf16x3p16_body on the dominant layer (probe
// build, tools/probe/conv_abl.sh): without MFMAs it still takes 0.91 of 1.5 ms, with the input band resident (conv3x3_band16_body,
// 42 % less LDS-DMA) the same -- what runs beside the matrix pipe is the LDS itself: eight 64 x 64 wave tiles read 128 KB of
// fragments per K-step (A twice, B four times) and the LDS-DMA writes 48 KB, ~1400 cycles at 128 B/clk against 1536 cycles of
// MFMAs.  Here a wave owns 128 x 64 (accumulators: 128 VGPRs, which needs the 256-register budget of two waves per SIMD, so
// there are no loader waves: every wave issues an eighth of the LDS-DMA), the block 256 x 256: per K-step 192 KB of fragment
// reads + 64 KB of LDS-DMA for 3072 cycles of MFMAs -- two thirds of the LDS and L2 traffic per FLOP.  Two 64 KB stages; the
// next stage's LDS-DMA is issued right after the barrier that frees it and has the whole K-step (~1.5 us) to land.
// Same K order, same three MFMAs per product in the same order: bit-identical to conv_bf16x3p16_body, which takes the rows that
// do not fill whole rounds of 256-row tiles (launch_conv_bf16x3p) and every layer this tile does not fit.
// ---------------------------------------------------------------------------------------------------------------------
// F16 (ConvP::f16): fp16 hi halves of the records x fp16 hi / lo weights, two MFMAs per product; 48 KB stages, THREE of them
// (the LDS-DMA two K-steps ahead, as in conv_bf16x3p16_body).
template <bool STG, int ABL = 0, bool F16 = false>
__device__ __forceinline__ void conv_bf16x3w16_body(const ConvP& p, unsigned char* smem) {
  constexpr int BM = 256, BN = 256, WM = 2, WN = 4, NW = 8, NT = NW * 64;
  constexpr int WTM = BM / WM, WTN = BN / WN, MI = WTM / 16, NJ = WTN / 16, MH = MI / 2;
  using Issuer = DmaIssuer<BM, BN, NW, ABL == 1 ? 1 : 0, true, F16>;
  constexpr int PLANE_B = Issuer::PLANE_B, STAGE = Issuer::STAGE, A_BYTES = Issuer::A_BYTES, PER_STEP = Issuer::PER_STEP;
  constexpr int NSTG = F16 ? 3 : 2;  // LDS stages; the LDS-DMA runs NSTG - 1 K-steps ahead
  static_assert(BM * (BN / 2) * 4 <= NSTG * STAGE, "half of the fp32 epilogue tile must fit in the staging area");
  const int nt = (p.Cout + BN - 1) / BN;
  const int ntiles = nt * ((p.M + BM - 1) / BM);
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int KT = p.K / PBK;
  const int G = gridDim.x, xq = G >> 3, xr = G & 7, xcd = blockIdx.x & 7;
  const int slot = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (blockIdx.x >> 3);

  for (int tile = slot; tile < ntiles; tile += G) {
    const int m0 = (tile / nt) * BM;
    const int n0 = (tile % nt) * BN;
    Issuer dma;
    dma.setup(p, m0, n0, wave, lane);
    dma.issue(p, smem, 0, 0);
    if (NSTG == 3 && KT > 1) dma.issue(p, smem, 1, 1);
    const int wm = wave / WN, wn = wave % WN;
    const int r = lane & 15, q = lane >> 4;
    f32x4v acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};
    // fragment offsets: rows 16 apart share the swizzle (aswz: (row >> 1) & 7, pswz16: (row >> 2) & 2), lo = hi ^ 64
    const int offa0 = F16 ? (wm * WTM + r) * PROW + pswz16(wm * WTM + r, q) * 16 : (wm * WTM + r) * 128 + aswz(wm * WTM + r, q) * 16;
    constexpr int AROW = F16 ? PROW : 128;  // bytes per A row in LDS
    const int offb0 = (wn * WTN + r) * PROW + pswz16(wn * WTN + r, q) * 16;
    auto read_b = [&](const unsigned char* ah, bf16x8 (&fbh)[NJ], bf16x8 (&fbl)[NJ]) {
      const unsigned char* bh = ah + A_BYTES + offb0;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        fbh[j] = *reinterpret_cast<const bf16x8*>(bh + j * 16 * PROW);
        fbl[j] = *reinterpret_cast<const bf16x8*>(bh + PLANE_B + j * 16 * PROW);
      }
    };
    auto read_a = [&](const unsigned char* ah, auto half_c, bf16x8 (&fah)[MH], bf16x8 (&fal)[MH]) {
      constexpr int half = decltype(half_c)::value;
#pragma unroll
      for (int i = 0; i < MH; ++i) {
        fah[i] = *reinterpret_cast<const bf16x8*>(ah + offa0 + (half * MH + i) * 16 * AROW);
        if (!F16) fal[i] = *reinterpret_cast<const bf16x8*>(ah + (offa0 ^ 64) + (half * MH + i) * 16 * AROW);
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
    int cur = 0, nxt = NSTG - 1;  // stage of K-step kt / of the K-step whose LDS-DMA is issued at kt
    const bool late = STG && wave >= NW / 2;
    if (!late) {
      for (int kt = 0; kt < KT; ++kt) {
        if (ABL != 4) { if (NSTG == 3 && kt + 1 < KT) wait_vm<PER_STEP>(); else wait_vm<0>(); }  // this wave's pieces of stage kt have landed ...
        __builtin_amdgcn_s_barrier();    // ... and everyone's; nobody reads stage kt - 1 any more
        if (kt + NSTG - 1 < KT) dma.issue(p, smem, kt + NSTG - 1, nxt);
        __builtin_amdgcn_sched_barrier(0);  // keep the LDS-DMA issue ahead of the ds_reads / MFMAs
        const unsigned char* ah = smem + cur * STAGE;
        bf16x8 fbh[NJ], fbl[NJ];
        read_b(ah, fbh, fbl);
        {
          bf16x8 fah[MH], fal[MH];
          read_a(ah, H0{}, fah, fal);
          mma(H0{}, fah, fal, fbh, fbl);
        }
        {
          bf16x8 fah[MH], fal[MH];
          read_a(ah, H1{}, fah, fal);
          mma(H1{}, fah, fal, fbh, fbl);
        }
        cur = cur == NSTG - 1 ? 0 : cur + 1;
        nxt = nxt == NSTG - 1 ? 0 : nxt + 1;
      }
    } else {
      bf16x8 gah[MH], gal[MH], gbh[NJ], gbl[NJ];  // second-half A fragments and the step's B fragments, carried across the barrier
      auto step = [&](auto carried_c, int kt) {
        if (ABL != 4) { if (NSTG == 3 && kt + 1 < KT) wait_vm<PER_STEP>(); else wait_vm<0>(); }
        __builtin_amdgcn_s_barrier();
        const unsigned char* ah = smem + cur * STAGE;
        if (decltype(carried_c)::value) mma(H1{}, gah, gal, gbh, gbl);  // second half of the previous K-step (the SIMD partner issues its LDS-DMA meanwhile)
        __builtin_amdgcn_sched_barrier(0);
        if (kt + NSTG - 1 < KT) dma.issue(p, smem, kt + NSTG - 1, nxt);
        __builtin_amdgcn_sched_barrier(0);
        read_b(ah, gbh, gbl);
        {
          bf16x8 fah[MH], fal[MH];
          read_a(ah, H0{}, fah, fal);
          mma(H0{}, fah, fal, gbh, gbl);
        }
        read_a(ah, H1{}, gah, gal);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the reads have returned before the next barrier
        cur = cur == NSTG - 1 ? 0 : cur + 1;
        nxt = nxt == NSTG - 1 ? 0 : nxt + 1;
      };
      step(H0{}, 0);
      for (int kt = 1; kt < KT; ++kt) step(H1{}, kt);
      mma(H1{}, gah, gal, gbh, gbl);
    }
    if (ABL == 4) wait_vm<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // every wave is done with the stages: they become the epilogue's fp32 tile, one column half at a time
    if (wide_epilogue_ok(p)) {  // block-uniform
      float* tile_f = reinterpret_cast<float*>(smem);
#pragma unroll
      for (int hc = 0; hc < 2; ++hc) {
        if ((wn >> 1) == hc) {  // wave-uniform
#pragma unroll
          for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
              for (int reg = 0; reg < 4; ++reg) {
                const int row = wm * WTM + i * 16 + 4 * q + reg;
                const int col = ((wn & 1) * WTN + j * 16 + r) ^ (((row >> 2) & 1) << 5);
                tile_f[row * (BN / 2) + col] = acc[i][j][reg];
              }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (p.pool2) epilogue_rows_pool<BM, BN / 2, NT>(p, smem, m0, n0 + hc * (BN / 2), tid);
        else epilogue_rows<BM, BN / 2, NT>(p, smem, m0, n0 + hc * (BN / 2), tid);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
    } else {
      conv_epilogue16<MI, NJ>(p, acc, m0 + wm * WTM, n0 + wn * WTN, r, q);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1 convolutions on narrow maps (W <= 131): the INPUT PATCH of a tile stays in LDS for all nine taps.
//
// The kernels above fetch, for every K-step (tap, 32 channels), the tile's 256 input records again -- the same pixels under
// another tap: 9 x 32 KB per channel chunk, two thirds of everything a CU moves through its vector-memory path, which is
// what the loaders, the decode kernels that share a CU, and the power budget feel.  With output tiles that are linear in
// the pixel index m = (b*H + oh)*W + ow, tap (dy, dx) of output pixel m is input pixel m + dy*W + dx whenever it is inside
// the image, so all nine taps of 248 consecutive output pixels read from ONE run of 248 + 2*(W+1) <= 512 input records:
//   * LDS: two patch buffers of 512 records (64 KB each: the channel chunk in use and the next one, loaded 1/9 per K-step
//     while the taps of the current chunk run) + two 16 KB stages of weights = 163,840 B, all of a CU's LDS;
//   * a tile computes 256 rows (the wave grid of the kernels above) of which the first 248 are output rows -- the patch
//     buffer holds 512 records, and 256 + 2*130 = 516 would not fit; rows 248..255 read whatever follows and are dropped;
//   * fragment reads: record (row + (W+1) + dy*W + dx) of the patch, chunk XOR-swizzled by the record index as above; a tap
//     that falls outside the image (or into the neighbouring image row / image) is zeroed per lane from a 9-bit mask;
//   * LDS-DMA per (tap, 32 channels): 7 KB of patch (amortised) + 16 KB of weights instead of 32 + 16 KB.
// Same K order (channel chunk, tap, channel) and the same three-MFMA products: bit-identical to the other kernels.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int QROWS = 248, QREC = 512, QPATCH = QREC * 128, QBST = 2 * 128 * PROW;  // real rows, patch records, bytes
static_assert(2 * QPATCH + 2 * QBST == 160 * 1024, "the patch kernel uses all of a CU's LDS");

template <bool STG>
__device__ __forceinline__ void conv3x3_patch_body(const ConvP& p, unsigned char* smem) {
  constexpr int BM = 256, BN = 128, WM = 4, WN = 2, NW = 8, NL = 4, NT = (NW + NL) * 64, MI = 2, NJ = 2;
  const int nt = (p.Cout + BN - 1) / BN;
  const int ntiles = nt * ((p.M + QROWS - 1) / QROWS);
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const bool loader = wave >= NW;
  const int nchunks = p.Cin / 32, KT = 9 * nchunks;
  const int W1 = p.W + 1;
  unsigned char* const bst = smem + 2 * QPATCH;
  const uint16_t* zero = reinterpret_cast<const uint16_t*>(p.zero16);

  const int G = gridDim.x, xq = G >> 3, xr = G & 7, xcd = blockIdx.x & 7;
  const int slot = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (blockIdx.x >> 3);

  for (int tile = slot; tile < ntiles; tile += G) {
    const int m0 = (tile / nt) * QROWS;
    const int n0 = (tile % nt) * BN;

    if (loader) {
      const int lw = wave - NW;
      // patch pieces of this loader: q = lw, lw + 4, ... (16 of 64); a piece = 8 records, lane -> (record, 16-byte position)
      int a_off[16];
      unsigned a_ok = 0;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int rec = (lw + 4 * j) * 8 + (lane >> 3);
        const int g = m0 - W1 + rec;  // input pixel (linear index) held by patch record `rec`
        const int c = (lane & 7) ^ ((rec >> 1) & 7);
        a_off[j] = g * p.Cin * 2 + c * 8;
        if (g >= 0 && g < p.M) a_ok |= 1u << j;  // stride 1, pad 1: as many input as output pixels
      }
      const uint16_t* b_hi[2];
      const uint16_t* b_lo[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int row = (lw * 2 + j) * 16 + (lane >> 2);
        const int n = n0 + row;
        const int c = pswz(row, lane & 3);
        b_hi[j] = n < p.Cout ? p.w_hi + (size_t)n * p.K + c * 8 : nullptr;
        b_lo[j] = n < p.Cout ? p.w_lo + (size_t)n * p.K + c * 8 : nullptr;
      }
      auto issue_b = [&](int kt) {
        unsigned char* bh = bst + (kt & 1) * QBST;
        unsigned char* bl = bh + BN * PROW;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int piece = (lw * 2 + j) * 1024;
          __builtin_amdgcn_global_load_lds(b_hi[j] ? b_hi[j] + (size_t)kt * PBK : zero, (lds_ptr_t)(bh + piece), 16, 0, 0);
          __builtin_amdgcn_global_load_lds(b_lo[j] ? b_lo[j] + (size_t)kt * PBK : zero, (lds_ptr_t)(bl + piece), 16, 0, 0);
        }
      };
      auto issue_a = [&](int chunk, int j) {  // piece j (0..15) of this loader, channel chunk `chunk`
        unsigned char* dst = smem + (chunk & 1) * QPATCH + (lw + 4 * j) * 1024;
        const uint16_t* src = ((a_ok >> j) & 1u) ? p.in_hi + (a_off[j] + chunk * 64) : zero;
        __builtin_amdgcn_global_load_lds(src, (lds_ptr_t)dst, 16, 0, 0);
      };
      // prologue: the whole patch of chunk 0 and the weights of K-step 0
#pragma unroll
      for (int j = 0; j < 16; ++j) issue_a(0, j);
      issue_b(0);
      wait_vm<0>();
      int chunk = 0, tap = 0;
      for (int kt = 0; kt < KT; ++kt) {
        __builtin_amdgcn_s_barrier();  // K-step kt may start: its weights (and at tap 0 its patch) are in LDS; kt-1 is read
        int na = 0;
        if (kt + 1 < KT) issue_b(kt + 1);
        if (chunk + 1 < nchunks && tap < 8) {  // two of this loader's sixteen pieces of the next chunk's patch per K-step
#pragma unroll
          for (int j = 0; j < 16; ++j)
            if (j == 2 * tap || j == 2 * tap + 1) issue_a(chunk + 1, j);
          na = 2;
        }
        // before the next barrier: the weights of kt+1 have landed (the patch pieces issued after them may still fly);
        // before the first tap of a chunk: all of its patch
        if (tap == 7 || na == 0) wait_vm<0>(); else wait_vm<2>();
        if (++tap == 9) { tap = 0; ++chunk; }
      }
      __builtin_amdgcn_s_barrier();  // (compute waves: done with the last stage)
      __builtin_amdgcn_s_barrier();  // (compute waves: accumulators are in the LDS tile)
      if (wide_epilogue_ok(p)) epilogue_rows<BM, BN, NT, QROWS>(p, smem, m0, n0, tid);
      __builtin_amdgcn_s_barrier();  // the tile is staging memory again
      continue;
    }

    const int wm = wave / WN, wn = wave % WN;
    const int r = lane & 31, h = lane >> 5;
    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // this lane's two output rows: which of the nine taps lie inside the image
    unsigned tmask[MI];
    int rbase[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int row = wm * 64 + i * 32 + r;
      rbase[i] = row + W1;
      const int m = m0 + row;
      tmask[i] = 0;
      if (row < QROWS && m < p.M) {
        const int rem = m % (p.H * p.W), oh = rem / p.W, ow = rem - oh * p.W;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const int y = oh + t / 3 - 1, x = ow + t % 3 - 1;
          if ((unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W) tmask[i] |= 1u << t;
        }
      }
    }
    int offb[2][NJ];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int row = wn * 64 + j * 32 + r;
        offb[kk][j] = row * PROW + pswz(row, 2 * kk + h) * 16;
      }
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    auto read_half = [&](const unsigned char* patch, const unsigned char* bh, int shift, int tap, int kk, bf16x8 (&fah)[MI],
                         bf16x8 (&fal)[MI], bf16x8 (&fbh)[NJ], bf16x8 (&fbl)[NJ]) {
      const unsigned char* bl = bh + BN * PROW;
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int rec = rbase[i] + shift;
        const int off = rec * 128 + (((2 * kk + h) ^ ((rec >> 1) & 7)) << 4);
        const unsigned keep = ((tmask[i] >> tap) & 1u) ? 0xFFFFFFFFu : 0u;
        u4 a = *reinterpret_cast<const u4*>(patch + off), b = *reinterpret_cast<const u4*>(patch + (off ^ 64));
        a &= keep;
        b &= keep;
        fah[i] = __builtin_bit_cast(bf16x8, a);
        fal[i] = __builtin_bit_cast(bf16x8, b);
      }
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        fbh[j] = *reinterpret_cast<const bf16x8*>(bh + offb[kk][j]);
        fbl[j] = *reinterpret_cast<const bf16x8*>(bl + offb[kk][j]);
      }
    };
    auto mma_half = [&](const bf16x8 (&fah)[MI], const bf16x8 (&fal)[MI], const bf16x8 (&fbh)[NJ], const bf16x8 (&fbl)[NJ]) {
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fal[i], fbh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fah[i], fbl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fah[i], fbh[j], acc[i][j], 0, 0, 0);
        }
    };
    const bool late = STG && wave >= NW / 2;
    int chunk = 0, tap = 0;
    auto step_refs = [&](const unsigned char*& patch, const unsigned char*& bh, int& shift, int kt) {
      patch = smem + (chunk & 1) * QPATCH;
      bh = bst + (kt & 1) * QBST;
      shift = (tap / 3 - 1) * p.W + (tap % 3 - 1);
    };
    if (!late) {
      for (int kt = 0; kt < KT; ++kt) {
        __builtin_amdgcn_s_barrier();
        const unsigned char *patch, *bh;
        int shift;
        step_refs(patch, bh, shift, kt);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          bf16x8 fah[MI], fal[MI], fbh[NJ], fbl[NJ];
          read_half(patch, bh, shift, tap, kk, fah, fal, fbh, fbl);
          mma_half(fah, fal, fbh, fbl);
        }
        if (++tap == 9) { tap = 0; ++chunk; }
      }
    } else {
      bf16x8 gah[MI], gal[MI], gbh[NJ], gbl[NJ];  // second-half fragments carried across the barrier
      auto step = [&](bool carried, int kt) {
        __builtin_amdgcn_s_barrier();
        const unsigned char *patch, *bh;
        int shift;
        step_refs(patch, bh, shift, kt);
        if (carried) mma_half(gah, gal, gbh, gbl);
        __builtin_amdgcn_sched_barrier(0);
        {
          bf16x8 fah[MI], fal[MI], fbh[NJ], fbl[NJ];
          read_half(patch, bh, shift, tap, 0, fah, fal, fbh, fbl);
          mma_half(fah, fal, fbh, fbl);
        }
        read_half(patch, bh, shift, tap, 1, gah, gal, gbh, gbl);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the reads have returned before the next barrier
        if (++tap == 9) { tap = 0; ++chunk; }
      };
      step(false, 0);
      for (int kt = 1; kt < KT; ++kt) step(true, kt);
      mma_half(gah, gal, gbh, gbl);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // every wave is done with the patches: they become the epilogue's fp32 tile
    if (wide_epilogue_ok(p)) {  // block-uniform
      float* tile_f = reinterpret_cast<float*>(smem);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int reg = 0; reg < 16; ++reg) {
            const int row = wm * 64 + i * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
            const int col = (wn * 64 + j * 32 + r) ^ (((row >> 2) & 1) << 5);
            tile_f[row * BN + col] = acc[i][j][reg];
          }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      epilogue_rows<BM, BN, NT, QROWS>(p, smem, m0, n0, tid);
    } else {
      __builtin_amdgcn_s_barrier();
      // (the narrow epilogue writes whole wave tiles: mask the eight surplus rows by shrinking M for the last row block)
      ConvP q = p;
      if (m0 + QROWS < q.M) q.M = m0 + QROWS;
      conv_epilogue<MI, NJ>(q, acc, m0 + wm * 64, n0 + wn * 64, r, h);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // the tile is staging memory again (next tile's LDS-DMA)
  }
}

// The patch-resident 3x3 kernel on v_mfma_f32_16x16x32_bf16 (round 3).  With the 16x16x32 build the pipelined kernel above is
// no longer power-bound: its K-step takes ~2150 cycles against 1536 of MFMA work, and what it waits for are the loader
// waves -- twelve LDS-DMA pieces per loader and K-step at the 130-180 cycles an LDS-DMA instruction costs its issuing wave
// (the same limit kept the Winograd form, conv_winograd.hip, from paying off).  Keeping a tile's input patch in LDS for all
// nine taps cuts that to 23 KB (six pieces per loader) per K-step.  Same structure as conv3x3_patch_body; fragments are the
// 16 x 32 ones of conv_bf16x3p16_body, patch records are XOR-swizzled with (record & 7) -- conflict-free for the 16-row
// fragment pattern at ANY tap shift (tools/probe/lds_swizzle_check.py; the 32-row pattern wanted (record >> 1) & 7) --
// and the stagger splits a K-step by row blocks.  Per output element the same three products per K-step in the same order as
// conv_bf16x3p16_body: bit-identical to it (tests), so a layer may take either.
template <bool STG>
__device__ __forceinline__ void conv3x3_patch16_body(const ConvP& p, unsigned char* smem) {
  constexpr int BM = 256, BN = 128, WN = 2, NW = 8, NL = 4, NT = (NW + NL) * 64, MI = 4, NJ = 4, MH = 2;
  const int nt = (p.Cout + BN - 1) / BN;
  const int ntiles = nt * ((p.M + QROWS - 1) / QROWS);
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const bool loader = wave >= NW;
  const int nchunks = p.Cin / 32, KT = 9 * nchunks;
  const int W1 = p.W + 1;
  unsigned char* const bst = smem + 2 * QPATCH;
  const uint16_t* zero = reinterpret_cast<const uint16_t*>(p.zero16);
  const int G = gridDim.x, xq = G >> 3, xr = G & 7, xcd = blockIdx.x & 7;
  const int slot = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (blockIdx.x >> 3);

  for (int tile = slot; tile < ntiles; tile += G) {
    const int m0 = (tile / nt) * QROWS;
    const int n0 = (tile % nt) * BN;
    if (loader) {
      const int lw = wave - NW;
      int a_off[16];
      unsigned a_ok = 0;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int rec = (lw + 4 * j) * 8 + (lane >> 3);
        const int g = m0 - W1 + rec;  // input pixel (linear index) held by patch record `rec`
        const int c = (lane & 7) ^ (rec & 7);
        a_off[j] = g * p.Cin * 2 + c * 8;
        if (g >= 0 && g < p.M) a_ok |= 1u << j;
      }
      const uint16_t* b_hi[2];
      const uint16_t* b_lo[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int row = (lw * 2 + j) * 16 + (lane >> 2);
        const int n = n0 + row;
        const int c = pswz16(row, lane & 3);
        b_hi[j] = n < p.Cout ? p.w_hi + (size_t)n * p.K + c * 8 : nullptr;
        b_lo[j] = n < p.Cout ? p.w_lo + (size_t)n * p.K + c * 8 : nullptr;
      }
      auto issue_b = [&](int kt) {
        unsigned char* bh = bst + (kt & 1) * QBST;
        unsigned char* bl = bh + BN * PROW;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int piece = (lw * 2 + j) * 1024;
          __builtin_amdgcn_global_load_lds(b_hi[j] ? b_hi[j] + (size_t)kt * PBK : zero, (lds_ptr_t)(bh + piece), 16, 0, 0);
          __builtin_amdgcn_global_load_lds(b_lo[j] ? b_lo[j] + (size_t)kt * PBK : zero, (lds_ptr_t)(bl + piece), 16, 0, 0);
        }
      };
      auto issue_a = [&](int chunk, int j) {
        unsigned char* dst = smem + (chunk & 1) * QPATCH + (lw + 4 * j) * 1024;
        const uint16_t* src = ((a_ok >> j) & 1u) ? p.in_hi + (a_off[j] + chunk * 64) : zero;
        __builtin_amdgcn_global_load_lds(src, (lds_ptr_t)dst, 16, 0, 0);
      };
#pragma unroll
      for (int j = 0; j < 16; ++j) issue_a(0, j);
      issue_b(0);
      wait_vm<0>();
      int chunk = 0, tap = 0;
      for (int kt = 0; kt < KT; ++kt) {
        __builtin_amdgcn_s_barrier();
        int na = 0;
        if (kt + 1 < KT) issue_b(kt + 1);
        if (chunk + 1 < nchunks && tap < 8) {
#pragma unroll
          for (int j = 0; j < 16; ++j)
            if (j == 2 * tap || j == 2 * tap + 1) issue_a(chunk + 1, j);
          na = 2;
        }
        if (tap == 7 || na == 0) wait_vm<0>(); else wait_vm<2>();
        if (++tap == 9) { tap = 0; ++chunk; }
      }
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_s_barrier();
      if (wide_epilogue_ok(p)) epilogue_rows<BM, BN, NT, QROWS>(p, smem, m0, n0, tid);
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
    unsigned tmask[MI];
    int rbase[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int row = wm * 64 + i * 16 + r;
      rbase[i] = row + W1;
      const int m = m0 + row;
      tmask[i] = 0;
      if (row < QROWS && m < p.M) {
        const int rem = m % (p.H * p.W), oh = rem / p.W, ow = rem - oh * p.W;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const int y = oh + t / 3 - 1, x = ow + t % 3 - 1;
          if ((unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W) tmask[i] |= 1u << t;
        }
      }
    }
    int offb[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int row = wn * 64 + j * 16 + r;
      offb[j] = row * PROW + pswz16(row, q) * 16;
    }
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    auto read_b = [&](const unsigned char* bh, bf16x8 (&fbh)[NJ], bf16x8 (&fbl)[NJ]) {
      const unsigned char* bl = bh + BN * PROW;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        fbh[j] = *reinterpret_cast<const bf16x8*>(bh + offb[j]);
        fbl[j] = *reinterpret_cast<const bf16x8*>(bl + offb[j]);
      }
    };
    auto read_a = [&](const unsigned char* patch, int shift, int tap, auto half_c, bf16x8 (&fah)[MH], bf16x8 (&fal)[MH]) {
      constexpr int half = decltype(half_c)::value;
#pragma unroll
      for (int i = 0; i < MH; ++i) {
        const int rec = rbase[half * MH + i] + shift;
        const int off = rec * 128 + ((q ^ (rec & 7)) << 4);
        const unsigned keep = ((tmask[half * MH + i] >> tap) & 1u) ? 0xFFFFFFFFu : 0u;
        u4 a = *reinterpret_cast<const u4*>(patch + off), b = *reinterpret_cast<const u4*>(patch + (off ^ 64));
        a &= keep;
        b &= keep;
        fah[i] = __builtin_bit_cast(bf16x8, a);
        fal[i] = __builtin_bit_cast(bf16x8, b);
      }
    };
    auto mma = [&](auto half_c, const bf16x8 (&fah)[MH], const bf16x8 (&fal)[MH], const bf16x8 (&fbh)[NJ], const bf16x8 (&fbl)[NJ]) {
      constexpr int half = decltype(half_c)::value;
#pragma unroll
      for (int i = 0; i < MH; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          f32x4v c = acc[half * MH + i][j];
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fal[i], fbh[j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fah[i], fbl[j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fah[i], fbh[j], c, 0, 0, 0);
          acc[half * MH + i][j] = c;
        }
    };
    using H0 = std::integral_constant<int, 0>;
    using H1 = std::integral_constant<int, 1>;
    const bool late = STG && wave >= NW / 2;
    int chunk = 0, tap = 0;
    auto step_refs = [&](const unsigned char*& patch, const unsigned char*& bh, int& shift, int kt) {
      patch = smem + (chunk & 1) * QPATCH;
      bh = bst + (kt & 1) * QBST;
      shift = (tap / 3 - 1) * p.W + (tap % 3 - 1);
    };
    if (!late) {
      for (int kt = 0; kt < KT; ++kt) {
        __builtin_amdgcn_s_barrier();
        const unsigned char *patch, *bh;
        int shift;
        step_refs(patch, bh, shift, kt);
        bf16x8 fbh[NJ], fbl[NJ];
        read_b(bh, fbh, fbl);
        {
          bf16x8 fah[MH], fal[MH];
          read_a(patch, shift, tap, H0{}, fah, fal);
          mma(H0{}, fah, fal, fbh, fbl);
        }
        {
          bf16x8 fah[MH], fal[MH];
          read_a(patch, shift, tap, H1{}, fah, fal);
          mma(H1{}, fah, fal, fbh, fbl);
        }
        if (++tap == 9) { tap = 0; ++chunk; }
      }
    } else {
      bf16x8 gah[MH], gal[MH], gbh[NJ], gbl[NJ];  // second-half A fragments and the step's B fragments, carried across the barrier
      auto step = [&](auto carried_c, int kt) {
        __builtin_amdgcn_s_barrier();
        const unsigned char *patch, *bh;
        int shift;
        step_refs(patch, bh, shift, kt);
        if (decltype(carried_c)::value) mma(H1{}, gah, gal, gbh, gbl);
        __builtin_amdgcn_sched_barrier(0);
        read_b(bh, gbh, gbl);
        {
          bf16x8 fah[MH], fal[MH];
          read_a(patch, shift, tap, H0{}, fah, fal);
          mma(H0{}, fah, fal, gbh, gbl);
        }
        read_a(patch, shift, tap, H1{}, gah, gal);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the reads have returned before the next barrier
        if (++tap == 9) { tap = 0; ++chunk; }
      };
      step(H0{}, 0);
      for (int kt = 1; kt < KT; ++kt) step(H1{}, kt);
      mma(H1{}, gah, gal, gbh, gbl);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // every wave is done with the patches: they become the epilogue's fp32 tile
    if (wide_epilogue_ok(p)) {  // block-uniform
      float* tile_f = reinterpret_cast<float*>(smem);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) {
            const int row = wm * 64 + i * 16 + 4 * q + reg;
            const int col = (wn * 64 + j * 16 + r) ^ (((row >> 2) & 1) << 5);
            tile_f[row * BN + col] = acc[i][j][reg];
          }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      epilogue_rows<BM, BN, NT, QROWS>(p, smem, m0, n0, tid);
    } else {
      __builtin_amdgcn_s_barrier();
      ConvP qq = p;  // (the narrow epilogue writes whole wave tiles: mask the eight surplus rows by shrinking M for the last row block)
      if (m0 + QROWS < qq.M) qq.M = m0 + QROWS;
      conv_epilogue16<MI, NJ>(qq, acc, m0 + wm * 64, n0 + wn * 64, r, q);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // the tile is staging memory again (next tile's LDS-DMA)
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1 convolutions of ANY width: a BAND of input records stays in LDS for the three taps of a filter row.
//
// Ablations of conv_bf16x3p16_body on the dominant layer (probe build, tools/probe/conv_abl.sh): without LDS-DMA 1.08 ms,
// without MFMAs 0.91 ms, both 1.53 ms, LDS-DMA never awaited 1.55 ms -- the kernel is bound by how many bytes a CU can take in
// per cycle (~33: 48 KB per K-step = 1470 cycles) next to 1536 cycles of MFMA work, not by latency.  Two thirds of those bytes
// are the tile's 256 input records, fetched again for every tap.  With tiles that are linear in the pixel index, the taps
// (kh, 0..2) of 256 consecutive output pixels read 258 consecutive input records: one band of 36 KiB (288 records, 36 LDS-DMA
// pieces) per (32-channel chunk, kh) serves three K-steps -- 12 + 16 KB per K-step instead of 32 + 16.
//   * LDS: three bands (the one in use and the next two, loaded 1/3 per K-step) + three 16 KB weight stages = 156 KB;
//   * every loader wave issues 3 band pieces + 4 weight pieces per K-step (12 in conv_bf16x3p16_body);
//   * fragment reads as in conv3x3_patch16_body: record (row + 1 + kw) of the band, chunk XOR-ed with (record & 7), taps
//     outside the image zeroed per lane from the row's 9-bit mask.
// Unlike the nine-tap patch kernel this needs no narrow map, keeps all 256 tile rows, and prefetches the weights two K-steps
// ahead.  Same K order (channel chunk, kh, kw, channel) and the same three MFMAs per product: bit-identical to
// conv_bf16x3p16_body (tests), so a layer may take either.  Not for the fused forms (pool2: rows are not linear in the pixel
// index; Cin2: its K-steps are not taps).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int RBREC = 288, RBAND = RBREC * 128, RBST = 2 * 128 * PROW;  // records per band, bytes per band / weight stage
static_assert(3 * RBAND + 3 * RBST <= 160 * 1024, "three bands and three weight stages must fit a CU's LDS");

template <bool STG, int ABL = 0>
__device__ __forceinline__ void conv3x3_band16_body(const ConvP& p, unsigned char* smem) {
  constexpr int BM = 256, BN = 128, WN = 2, NW = 8, NL = 4, NT = (NW + NL) * 64, MI = 4, NJ = 4, MH = 2;
  static_assert(BM * BN * 4 <= 3 * RBAND + 3 * RBST, "the fp32 epilogue tile must fit in the staging area");
  const int nt = (p.Cout + BN - 1) / BN;
  const int ntiles = nt * ((p.M + BM - 1) / BM);
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const bool loader = wave >= NW;
  const int nchunks = p.Cin / 32, NB = 3 * nchunks, KT = 9 * nchunks;  // bands, K-steps
  unsigned char* const bst = smem + 3 * RBAND;
  const uint16_t* zero = reinterpret_cast<const uint16_t*>(p.zero16);
  const int G = gridDim.x, xq = G >> 3, xr = G & 7, xcd = blockIdx.x & 7;
  const int slot = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (blockIdx.x >> 3);

  for (int tile = slot; tile < ntiles; tile += G) {
    const int m0 = (tile / nt) * BM;
    const int n0 = (tile % nt) * BN;
    if (loader) {
      const int lw = wave - NW;
      // band pieces of this loader: lw, lw + 4, ... (9 of 36); a piece = 8 records, lane -> (record, 16-byte position)
      int a_g[9], a_off[9];
#pragma unroll
      for (int j = 0; j < 9; ++j) {
        const int rec = (lw + 4 * j) * 8 + (lane >> 3);
        const int g = m0 - 1 + rec;  // input pixel (linear index) held by record `rec` of a kh = 1 band
        const int c = (lane & 7) ^ (rec & 7);
        a_g[j] = g;
        a_off[j] = g * p.Cin * 2 + c * 8;
      }
      const uint16_t* b_hi[2];
      const uint16_t* b_lo[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int row = (lw * 2 + j) * 16 + (lane >> 2);
        const int n = n0 + row;
        const int c = pswz16(row, lane & 3);
        b_hi[j] = n < p.Cout ? p.w_hi + (size_t)n * p.K + c * 8 : nullptr;
        b_lo[j] = n < p.Cout ? p.w_lo + (size_t)n * p.K + c * 8 : nullptr;
      }
      auto issue_b = [&](int kt, int stage) {
        unsigned char* bh = bst + stage * RBST;
        unsigned char* bl = bh + BN * PROW;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int piece = (lw * 2 + j) * 1024;
          if (ABL == 1) continue;
          __builtin_amdgcn_global_load_lds(b_hi[j] ? b_hi[j] + (size_t)kt * PBK : zero, (lds_ptr_t)(bh + piece), 16, 0, 0);
          __builtin_amdgcn_global_load_lds(b_lo[j] ? b_lo[j] + (size_t)kt * PBK : zero, (lds_ptr_t)(bl + piece), 16, 0, 0);
        }
      };
      // pieces [3 * third, 3 * third + 3) of this loader's nine, band (chunk, kh) into ring slot `buf`
      auto issue_a = [&](int chunk, int kh, int buf, int third) {
        const int dg = (kh - 1) * p.W, doff = dg * p.Cin * 2 + chunk * 64;
#pragma unroll
        for (int j = 0; j < 9; ++j) {
          if (j / 3 != third) continue;
          unsigned char* dst = smem + buf * RBAND + (lw + 4 * j) * 1024;
          const uint16_t* src = (unsigned)(a_g[j] + dg) < (unsigned)p.M ? p.in_hi + (a_off[j] + doff) : zero;
          if (ABL == 1) continue;
          __builtin_amdgcn_global_load_lds(src, (lds_ptr_t)dst, 16, 0, 0);
        }
      };
      // prologue: bands 0 and 1 whole, weights of K-steps 0 and 1 (bands first: the wait before barrier 0 leaves only the
      // second band and the second weight stage in flight)
      for (int t3 = 0; t3 < 3; ++t3) issue_a(0, 0, 0, t3);
      issue_b(0, 0);
      for (int t3 = 0; t3 < 3; ++t3) issue_a(0, 1, 1, t3);  // (NB >= 3 always)
      issue_b(1, 1);
      int prev = 13;  // pieces issued after everything K-step kt needs: before barrier kt they may still be in flight
      int b2c = 0, b2k = 2, b2buf = 2, third = 0, wst = 2;  // next band to load (chunk, kh, ring slot), its third, weight stage of kt + 2
      for (int kt = 0; kt < KT; ++kt) {
        if (ABL != 4) {
          if (prev == 13) wait_vm<13>();
          else if (prev == 7) wait_vm<7>();
          else if (prev == 4) wait_vm<4>();
          else if (prev == 3) wait_vm<3>();
          else wait_vm<0>();
        }
        __builtin_amdgcn_s_barrier();
        prev = 0;
        if (b2c < nchunks) {  // (wave-uniform) a third of band kt / 3 + 2
          issue_a(b2c, b2k, b2buf, third);
          prev += 3;
          if (++third == 3) {
            third = 0;
            b2buf = b2buf == 2 ? 0 : b2buf + 1;
            if (++b2k == 3) { b2k = 0; ++b2c; }
          }
        }
        if (kt + 2 < KT) {
          issue_b(kt + 2, wst);
          prev += 4;
        }
        wst = wst == 2 ? 0 : wst + 1;
      }
      if (ABL == 4) wait_vm<0>();
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_s_barrier();
      if (wide_epilogue_ok(p)) epilogue_rows<BM, BN, NT>(p, smem, m0, n0, tid);
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
    unsigned tmask[MI];
    int rbase[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int row = wm * 64 + i * 16 + r;
      rbase[i] = row;  // band record row + kw holds input pixel m0 + row + (kh - 1) W + kw - 1
      const int m = m0 + row;
      tmask[i] = 0;
      if (m < p.M) {
        const int rem = m % (p.H * p.W), oh = rem / p.W, ow = rem - oh * p.W;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const int y = oh + t / 3 - 1, x = ow + t % 3 - 1;
          if ((unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W) tmask[i] |= 1u << t;
        }
      }
    }
    int offb[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int row = wn * 64 + j * 16 + r;
      offb[j] = row * PROW + pswz16(row, q) * 16;
    }
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    auto read_b = [&](const unsigned char* bh, bf16x8 (&fbh)[NJ], bf16x8 (&fbl)[NJ]) {
      const unsigned char* bl = bh + BN * PROW;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        fbh[j] = *reinterpret_cast<const bf16x8*>(bh + offb[j]);
        fbl[j] = *reinterpret_cast<const bf16x8*>(bl + offb[j]);
      }
    };
    auto read_a = [&](const unsigned char* band, int shift, int tap, auto half_c, bf16x8 (&fah)[MH], bf16x8 (&fal)[MH]) {
      constexpr int half = decltype(half_c)::value;
#pragma unroll
      for (int i = 0; i < MH; ++i) {
        const int rec = rbase[half * MH + i] + shift;
        const int off = rec * 128 + ((q ^ (rec & 7)) << 4);
        const unsigned keep = ((tmask[half * MH + i] >> tap) & 1u) ? 0xFFFFFFFFu : 0u;
        u4 a = *reinterpret_cast<const u4*>(band + off), b = *reinterpret_cast<const u4*>(band + (off ^ 64));
        a &= keep;
        b &= keep;
        fah[i] = __builtin_bit_cast(bf16x8, a);
        fal[i] = __builtin_bit_cast(bf16x8, b);
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
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fal[i], fbh[j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fah[i], fbl[j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fah[i], fbh[j], c, 0, 0, 0);
          acc[half * MH + i][j] = c;
        }
    };
    using H0 = std::integral_constant<int, 0>;
    using H1 = std::integral_constant<int, 1>;
    const bool late = STG && wave >= NW / 2;
    int tap = 0, kw = 0, bbuf = 0, wst = 0;  // tap = 3 kh + kw of this K-step, its band's ring slot, its weight stage
    auto advance = [&]() {
      wst = wst == 2 ? 0 : wst + 1;
      if (++kw == 3) {
        kw = 0;
        bbuf = bbuf == 2 ? 0 : bbuf + 1;
      }
      if (++tap == 9) tap = 0;
    };
    if (!late) {
      for (int kt = 0; kt < KT; ++kt) {
        __builtin_amdgcn_s_barrier();
        const unsigned char* band = smem + bbuf * RBAND;
        const unsigned char* bh = bst + wst * RBST;
        bf16x8 fbh[NJ], fbl[NJ];
        read_b(bh, fbh, fbl);
        {
          bf16x8 fah[MH], fal[MH];
          read_a(band, kw, tap, H0{}, fah, fal);
          mma(H0{}, fah, fal, fbh, fbl);
        }
        {
          bf16x8 fah[MH], fal[MH];
          read_a(band, kw, tap, H1{}, fah, fal);
          mma(H1{}, fah, fal, fbh, fbl);
        }
        advance();
      }
    } else {
      bf16x8 gah[MH], gal[MH], gbh[NJ], gbl[NJ];  // second-half A fragments and the step's B fragments, carried across the barrier
      auto step = [&](auto carried_c) {
        __builtin_amdgcn_s_barrier();
        const unsigned char* band = smem + bbuf * RBAND;
        const unsigned char* bh = bst + wst * RBST;
        if (decltype(carried_c)::value) mma(H1{}, gah, gal, gbh, gbl);
        __builtin_amdgcn_sched_barrier(0);
        read_b(bh, gbh, gbl);
        {
          bf16x8 fah[MH], fal[MH];
          read_a(band, kw, tap, H0{}, fah, fal);
          mma(H0{}, fah, fal, gbh, gbl);
        }
        read_a(band, kw, tap, H1{}, gah, gal);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the reads have returned before the next barrier
        advance();
      };
      step(H0{});
      for (int kt = 1; kt < KT; ++kt) step(H1{});
      mma(H1{}, gah, gal, gbh, gbl);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // every wave is done with the bands: they become the epilogue's fp32 tile
    if (wide_epilogue_ok(p)) {  // block-uniform
      float* tile_f = reinterpret_cast<float*>(smem);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) {
            const int row = wm * 64 + i * 16 + 4 * q + reg;
            const int col = (wn * 64 + j * 16 + r) ^ (((row >> 2) & 1) << 5);
            tile_f[row * BN + col] = acc[i][j][reg];
          }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      epilogue_rows<BM, BN, NT>(p, smem, m0, n0, tid);
    } else {
      __builtin_amdgcn_s_barrier();
      conv_epilogue16<MI, NJ>(p, acc, m0 + wm * 64, n0 + wn * 64, r, q);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // the tile is staging memory again (next tile's LDS-DMA)
  }
}

__global__ __launch_bounds__(768, 3) void conv_bf16x3b16_3x3_band(const ConvP p) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[3 * RBAND + 3 * RBST];
  conv3x3_band16_body<true>(p, smem);
}
__global__ __launch_bounds__(768, 3) void conv_bf16x3b16_3x3_band_k4608(const ConvP p) {  // the dominant shape, own symbol
  __shared__ __attribute__((aligned(1024))) unsigned char smem[3 * RBAND + 3 * RBST];
  conv3x3_band16_body<true>(p, smem);
}
#ifdef D2T_PROBES
template <int ABL>
__global__ __launch_bounds__(768, 3) void conv_bf16x3b16_probe(const ConvP p) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[3 * RBAND + 3 * RBST];
  conv3x3_band16_body<true, ABL>(p, smem);
}
#endif

__global__ __launch_bounds__(768, 3) void conv_bf16x3q16_3x3_patch(const ConvP p) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[160 * 1024];
  conv3x3_patch16_body<true>(p, smem);
}
__global__ __launch_bounds__(768, 3) void conv_bf16x3q16_3x3_patch_k4608(const ConvP p) {  // the dominant shape, own symbol
  __shared__ __attribute__((aligned(1024))) unsigned char smem[160 * 1024];
  conv3x3_patch16_body<true>(p, smem);
}

__global__ __launch_bounds__(768, 3) void conv_bf16x3q_3x3_patch(const ConvP p) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[160 * 1024];
  conv3x3_patch_body<true>(p, smem);
}
__global__ __launch_bounds__(768, 3) void conv_bf16x3q_3x3_patch_k4608(const ConvP p) {  // the dominant shape, own symbol
  __shared__ __attribute__((aligned(1024))) unsigned char smem[160 * 1024];
  conv3x3_patch_body<true>(p, smem);
}

// non-template entry points (the host-side stub of a __global__ template using the LDS-DMA builtin is not emitted)
__global__ __launch_bounds__(768, 3) __attribute__((amdgpu_num_vgpr(112))) void conv_bf16x3p_256x128(const ConvP p) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[3 * (2 * 256 * PROW + 2 * 128 * PROW)];
  conv_bf16x3p_body<256, 128, 4, 2, 4>(p, smem);
}
// the dominant GEMM shape (512 -> 512 channels, 3x3: K = 4608) under its own symbol, so that rocprofv3's per-kernel rows
// separate it from the other layers
__global__ __launch_bounds__(768, 3) __attribute__((amdgpu_num_vgpr(112))) void conv_bf16x3p_256x128_k4608(const ConvP p) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[3 * (2 * 256 * PROW + 2 * 128 * PROW)];
  conv_bf16x3p_body<256, 128, 4, 2, 4>(p, smem);
}

// waves 4-7 half a K-step behind their SIMD partners
__global__ __launch_bounds__(768, 3) void conv_bf16x3p_256x128_s(const ConvP p) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[3 * (2 * 256 * PROW + 2 * 128 * PROW)];
  conv_bf16x3p_body<256, 128, 4, 2, 4, 0, true>(p, smem);
}
__global__ __launch_bounds__(768, 3) void conv_bf16x3p_256x128_s_k4608(const ConvP p) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[3 * (2 * 256 * PROW + 2 * 128 * PROW)];
  conv_bf16x3p_body<256, 128, 4, 2, 4, 0, true>(p, smem);
}

// 16x16x32 MFMA build of the same kernel (p.pipelined == 3)
__global__ __launch_bounds__(768, 3) void conv_bf16x3p16_256x128_s(const ConvP p) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[3 * (2 * 256 * PROW + 2 * 128 * PROW)];
  conv_bf16x3p16_body<256, 128, 4, 2, 4, true>(p, smem);
}
__global__ __launch_bounds__(768, 3) void conv_bf16x3p16_256x128_s_k4608(const ConvP p) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[3 * (2 * 256 * PROW + 2 * 128 * PROW)];
  conv_bf16x3p16_body<256, 128, 4, 2, 4, true>(p, smem);
}
__global__ __launch_bounds__(768, 3) void conv_bf16x3p16_256x128(const ConvP p) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[3 * (2 * 256 * PROW + 2 * 128 * PROW)];
  conv_bf16x3p16_body<256, 128, 4, 2, 4, false>(p, smem);
}

// 256 x 256 tile, eight waves (two per SIMD, 256 VGPRs), p.pipelined == 7
__global__ __launch_bounds__(512, 2) void conv_bf16x3w16_256x256(const ConvP p) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * (2 * 256 * PROW + 2 * 256 * PROW)];
  conv_bf16x3w16_body<true>(p, smem);
}
__global__ __launch_bounds__(512, 2) void conv_bf16x3w16_256x256_k4608(const ConvP p) {  // the dominant shape, own symbol
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * (2 * 256 * PROW + 2 * 256 * PROW)];
  conv_bf16x3w16_body<true>(p, smem);
}
__global__ __launch_bounds__(512, 2) void conv_bf16x3w16_256x256_ns(const ConvP p) {  // no stagger (A/B: D2T_CONV_STAGGER=0)
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * (2 * 256 * PROW + 2 * 256 * PROW)];
  conv_bf16x3w16_body<false>(p, smem);
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

// fp16x2 build of the 256 x 256 tile (three 48 KB stages)
__global__ __launch_bounds__(512, 2) void conv_f16x2w16_256x256(const ConvP p) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[3 * (256 * PROW + 2 * 256 * PROW)];
  conv_bf16x3w16_body<true, 0, true>(p, smem);
}
__global__ __launch_bounds__(512, 2) void conv_f16x2w16_256x256_k4608(const ConvP p) {  // the dominant shape, own symbol
  __shared__ __attribute__((aligned(1024))) unsigned char smem[3 * (256 * PROW + 2 * 256 * PROW)];
  conv_bf16x3w16_body<true, 0, true>(p, smem);
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

// the same without dedicated loader waves: 8 waves (2 per SIMD), the compute waves issue the LDS-DMA themselves
__global__ __launch_bounds__(512, 2) void conv_bf16x3p_256x128_w8(const ConvP p) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[3 * (2 * 256 * PROW + 2 * 128 * PROW)];
  conv_bf16x3p_body<256, 128, 4, 2, 0>(p, smem);
}
__global__ __launch_bounds__(512, 2) void conv_bf16x3p_256x128_w8_k4608(const ConvP p) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[3 * (2 * 256 * PROW + 2 * 128 * PROW)];
  conv_bf16x3p_body<256, 128, 4, 2, 0>(p, smem);
}

// ablation probes of the dominant shape (tools/conv_bench.py, D2T_CONV_ABL=1|2|3): results are garbage by construction
__global__ __launch_bounds__(768, 3) void conv_bf16x3p_probe_no_dma(const ConvP p) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[3 * (2 * 256 * PROW + 2 * 128 * PROW)];
  conv_bf16x3p_body<256, 128, 4, 2, 4, 1>(p, smem);
}
__global__ __launch_bounds__(768, 3) void conv_bf16x3p_probe_no_mfma(const ConvP p) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[3 * (2 * 256 * PROW + 2 * 128 * PROW)];
  conv_bf16x3p_body<256, 128, 4, 2, 4, 2>(p, smem);
}
__global__ __launch_bounds__(768, 3) void conv_bf16x3p_probe_dma_only(const ConvP p) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[3 * (2 * 256 * PROW + 2 * 128 * PROW)];
  conv_bf16x3p_body<256, 128, 4, 2, 4, 3>(p, smem);
}

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
  static const int prio = getenv("D2T_CONV_PRIO") ? atoi(getenv("D2T_CONV_PRIO")) : 0;
  ConvP q = p;
  q.wave_prio = prio;
  const ConvP& p2 = q;
  const int nt = (p.Cout + 127) / 128;
  int tiles = ((p.M + 255) / 256) * nt;
  int grid = cus - (p.reserved_cus > 0 ? p.reserved_cus : 0);
  if (grid < 8) grid = 8;
  // 3x3 / stride 1 / pad 1 on a narrow map: the kernel that keeps a tile's input patch in LDS for all nine taps
  // (p.pipelined == 2 or D2T_CONV_PATCH=1; not the default: in the bench's sustained, power-bound state it runs the dominant
  // layer in 1.67 ms like the kernel below, cold it is 5 % slower -- DESIGN.md 5.1)
  static const int patch_env = getenv("D2T_CONV_PATCH") ? atoi(getenv("D2T_CONV_PATCH")) : 0;
  if ((patch_env || p.pipelined == 2) && !abl_probe() && p.KH == 3 && p.KW == 3 && p.SH == 1 && p.SW == 1 && p.PH == 1 && p.PW == 1 && p.OH == p.H &&
      p.OW == p.W && p.W + 1 <= (QREC - QROWS) / 2 && p.Cin % 32 == 0 && p.M >= QROWS) {
    const int qtiles = ((p.M + QROWS - 1) / QROWS) * nt;
    if (grid > qtiles) grid = qtiles;
    if (p.K == 4608 && p.Cout == 512) hipLaunchKernelGGL(conv_bf16x3q_3x3_patch_k4608, dim3(grid), dim3(768), 0, s, p2);
    else hipLaunchKernelGGL(conv_bf16x3q_3x3_patch, dim3(grid), dim3(768), 0, s, p2);
    return hipGetLastError();
  }
  if (p.f16) {  // fp16x2 mode: the pipelined 16x16x32 kernel only
    if (p.pipelined != 3) return hipErrorInvalidValue;
    static const int f16_wide = getenv("D2T_F16_WIDE") ? atoi(getenv("D2T_F16_WIDE")) : 0;
    if (f16_wide && p.Cout >= 256 && p.m_base == 0) {  // 256 x 256 tiles for whole rounds, the 256 x 128 kernel for the leftover tile rows
      const int ntw = (p.Cout + 255) / 256, mtw = (p.M + 255) / 256;
      int tw = mtw * ntw, gw = grid < tw ? grid : tw;
      const int rounds = tw / gw, rem = tw - rounds * gw;
      int tail_from = -1;
      if (rounds >= 1 && rem > 0 && rem < 0.5f * gw) {
        const int main_mt = rounds * gw / ntw;
        tail_from = main_mt * 256;
        q.M = tail_from;
        tw = main_mt * ntw;
        if (gw > tw) gw = tw;
      }
      if (p.K == 4608 && p.Cout == 512) hipLaunchKernelGGL(conv_f16x2w16_256x256_k4608, dim3(gw), dim3(512), 0, s, p2);
      else hipLaunchKernelGGL(conv_f16x2w16_256x256, dim3(gw), dim3(512), 0, s, p2);
      hipError_t e = hipGetLastError();
      if (e != hipSuccess || tail_from < 0) return e;
      ConvP t = p;
      t.wave_prio = prio;
      t.m_base = tail_from;
      const int tt = ((p.M - tail_from + 255) / 256) * nt;
      hipLaunchKernelGGL(conv_f16x2p16_256x128_s, dim3(tt < grid ? tt : grid), dim3(768), 0, s, t);
      return hipGetLastError();
    }
    tiles = ((p.M - p.m_base + 255) / 256) * nt;
    int tail_f = -1;  // whole rounds only, as for the split-bf16 build below
    static const float tail_frac_f = getenv("D2T_CONV_TAIL") ? (float)atof(getenv("D2T_CONV_TAIL")) : 0.5f;
    if (p.split_tail && p.m_base == 0) {
      const int rounds = tiles / grid, rem = tiles - rounds * grid;
      if (rounds >= 1 && rem > 0 && rem < tail_frac_f * grid) {
        const int main_mt = rounds * grid / nt;
        tail_f = main_mt * 256;
        q.M = tail_f;
        tiles = main_mt * nt;
      }
    }
    if (grid > tiles) grid = tiles;
    if (p.K == 4608 && p.Cout == 512) hipLaunchKernelGGL(conv_f16x2p16_256x128_s_k4608, dim3(grid), dim3(768), 0, s, p2);
    else hipLaunchKernelGGL(conv_f16x2p16_256x128_s, dim3(grid), dim3(768), 0, s, p2);
    hipError_t ef = hipGetLastError();
    if (ef != hipSuccess || tail_f < 0) return ef;
    ConvP tf = p;
    tf.wave_prio = prio;
    tf.m_base = tail_f;
    hipLaunchKernelGGL(conv_f16x2p16_64x128_tail, dim3(((p.M - tail_f + 63) / 64) * nt), dim3(768), 0, s, tf);
    return hipGetLastError();
  }
  if (p.pipelined == 7 && p.Cout >= 256 && p.m_base == 0) {
    // 256 x 256 tiles on eight waves for the rows that fill whole rounds of them; a last round that would be less than half
    // full goes, as whole 256-row tile rows, to the 256 x 128 kernel (same MFMA shape, K order and products: bit-identical)
    const int ntw = (p.Cout + 255) / 256, mtw = (p.M + 255) / 256;
    int tw = mtw * ntw, gw = grid < tw ? grid : tw;
    const int rounds = tw / gw, rem = tw - rounds * gw;
    int tail_from = -1;
    static const float tail_frac = getenv("D2T_CONV_TAIL") ? (float)atof(getenv("D2T_CONV_TAIL")) : 0.5f;
    if (rounds >= 1 && rem > 0 && rem < tail_frac * gw) {
      const int main_mt = rounds * gw / ntw;
      tail_from = main_mt * 256;
      q.M = tail_from;
      tw = main_mt * ntw;
      if (gw > tw) gw = tw;
    }
    static const int stagger_w = getenv("D2T_CONV_STAGGER") ? atoi(getenv("D2T_CONV_STAGGER")) : 1;
    if (!stagger_w) hipLaunchKernelGGL(conv_bf16x3w16_256x256_ns, dim3(gw), dim3(512), 0, s, p2);
    else if (p.K == 4608 && p.Cout == 512) hipLaunchKernelGGL(conv_bf16x3w16_256x256_k4608, dim3(gw), dim3(512), 0, s, p2);
    else hipLaunchKernelGGL(conv_bf16x3w16_256x256, dim3(gw), dim3(512), 0, s, p2);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || tail_from < 0) return e;
    ConvP t = p;
    t.wave_prio = prio;
    t.m_base = tail_from;
    const int tt = ((p.M - tail_from + 255) / 256) * nt;
    hipLaunchKernelGGL(conv_bf16x3p16_256x128_s, dim3(tt < grid ? tt : grid), dim3(768), 0, s, t);
    return hipGetLastError();
  }
  if (p.pipelined == 3 || p.pipelined == 5 || p.pipelined == 6 || p.pipelined == 7) {  // the 16x16x32 builds: all rows of a layer on ONE MFMA shape (no hand-over of tail rows to the 32x32x16 kernels)
    // 5: 3x3 / stride 1 / pad 1 layers on narrow maps take the patch-resident form (half the LDS-DMA pieces per K-step);
    // bit-identical to the plain 16x16x32 kernel, so the choice never shows in the values
    if (p.pipelined == 5 && p.KH == 3 && p.KW == 3 && p.SH == 1 && p.SW == 1 && p.PH == 1 && p.PW == 1 && p.OH == p.H && p.OW == p.W &&
        p.W + 1 <= (QREC - QROWS) / 2 && p.Cin % 32 == 0 && p.M >= QROWS) {
      const int qtiles = ((p.M + QROWS - 1) / QROWS) * nt;
      if (grid > qtiles) grid = qtiles;
      if (p.K == 4608 && p.Cout == 512) hipLaunchKernelGGL(conv_bf16x3q16_3x3_patch_k4608, dim3(grid), dim3(768), 0, s, p2);
      else hipLaunchKernelGGL(conv_bf16x3q16_3x3_patch, dim3(grid), dim3(768), 0, s, p2);
      return hipGetLastError();
    }
    tiles = ((p.M - p.m_base + 255) / 256) * nt;
    if (grid > tiles) grid = tiles;
    // 6: 3x3 / stride 1 / pad 1 layers (any width) keep a band of input records in LDS for the three taps of a filter row
    if (p.pipelined == 6 && p.m_base == 0 && p.KH == 3 && p.KW == 3 && p.SH == 1 && p.SW == 1 && p.PH == 1 && p.PW == 1 && p.OH == p.H && p.OW == p.W &&
        p.Cin % 32 == 0 && !p.pool2 && !p.Cin2) {
#ifdef D2T_PROBES
      if (abl_probe()) {
        switch (abl_probe()) {
          case 1: hipLaunchKernelGGL(conv_bf16x3b16_probe<1>, dim3(grid), dim3(768), 0, s, p2); break;
          case 2: hipLaunchKernelGGL(conv_bf16x3b16_probe<2>, dim3(grid), dim3(768), 0, s, p2); break;
          default: hipLaunchKernelGGL(conv_bf16x3b16_probe<4>, dim3(grid), dim3(768), 0, s, p2); break;
        }
        return hipGetLastError();
      }
#endif
      if (p.K == 4608 && p.Cout == 512) hipLaunchKernelGGL(conv_bf16x3b16_3x3_band_k4608, dim3(grid), dim3(768), 0, s, p2);
      else hipLaunchKernelGGL(conv_bf16x3b16_3x3_band, dim3(grid), dim3(768), 0, s, p2);
      return hipGetLastError();
    }
    static const int stagger16 = getenv("D2T_CONV_STAGGER") ? atoi(getenv("D2T_CONV_STAGGER")) : 1;
    // Whole rounds only (round 3, the 16x16x32 form of round 2's hand-over): when the last round of 256-row tiles would be
    // less than half full (the dominant layer at B = 64: 2064 tiles = eight rounds + 16 tiles; at B = 32: 4.03 rounds), the
    // rows behind the whole rounds go to the 64 x 128 build of the same body.  The caller switches it off (split_tail = 0)
    // while decode loops are in flight: their kernels run in exactly that hole.
    static const float tail16 = getenv("D2T_CONV_TAIL") ? (float)atof(getenv("D2T_CONV_TAIL")) : 0.5f;
    int tail_from16 = -1;
    if (p.pipelined == 3 && p.split_tail && p.m_base == 0 && !abl_probe()) {
      const int g0 = grid, rounds = tiles / g0, rem = tiles - rounds * g0;
      if (rounds >= 1 && rem > 0 && rem < tail16 * g0) {
        const int main_mt = rounds * g0 / nt;  // whole rows of tiles that fit into the whole rounds
        tail_from16 = main_mt * 256;
        q.M = tail_from16;
        tiles = main_mt * nt;
        if (grid > tiles) grid = tiles;
      }
    }
    auto launch_tail16 = [&]() -> hipError_t {
      hipError_t e = hipGetLastError();
      if (e != hipSuccess || tail_from16 < 0) return e;
      ConvP t = p;
      t.wave_prio = prio;
      t.m_base = tail_from16;
      const int tt = ((p.M - tail_from16 + 63) / 64) * nt;
      hipLaunchKernelGGL(conv_bf16x3p16_64x128_tail, dim3(tt), dim3(768), 0, s, t);
      return hipGetLastError();
    };
#ifdef D2T_PROBES
    if (abl_probe()) {
      switch (abl_probe()) {
        case 1: hipLaunchKernelGGL(conv_bf16x3p16_probe<1>, dim3(grid), dim3(768), 0, s, p2); break;
        case 2: hipLaunchKernelGGL(conv_bf16x3p16_probe<2>, dim3(grid), dim3(768), 0, s, p2); break;
        case 8: hipLaunchKernelGGL(conv_bf16x3p16_probe<8>, dim3(grid), dim3(768), 0, s, p2); break;
        default: hipLaunchKernelGGL(conv_bf16x3p16_probe<4>, dim3(grid), dim3(768), 0, s, p2); break;
      }
      return hipGetLastError();
    }
#endif
    if (!stagger16) hipLaunchKernelGGL(conv_bf16x3p16_256x128, dim3(grid), dim3(768), 0, s, p2);
    else if (p.K == 4608 && p.Cout == 512) hipLaunchKernelGGL(conv_bf16x3p16_256x128_s_k4608, dim3(grid), dim3(768), 0, s, p2);
    else hipLaunchKernelGGL(conv_bf16x3p16_256x128_s, dim3(grid), dim3(768), 0, s, p2);
    return launch_tail16();
  }
  // Whole rounds only.  The dominant layer has 2064 tiles: eight rounds on 256 CUs and then sixteen tiles that keep 16 CUs
  // busy for a ninth of the kernel's duration while 240 idle.  When the last round is less than `tail_frac` full, the pipelined
  // kernel stops after the whole rounds (cut back to whole rows of tiles) and the remaining rows go to the 128-row kernel,
  // whose small tiles spread over the chip.  Same arithmetic per output element in both kernels (tests assert bit-identity),
  // so which kernel computes a row never shows in the values.  Alone this is worth 2.4 % on the dominant layer (1.72 ->
  // 1.68 ms: the ninth round runs at a higher clock and without contention, so it costs 145 us, the 128-row kernel 105 us).
  // The caller switches it off (split_tail = 0) while decode loops of earlier batches are in flight: their kernels run in
  // exactly that hole, and the end-to-end rate is the same either way.
  static const float tail_frac = getenv("D2T_CONV_TAIL") ? (float)atof(getenv("D2T_CONV_TAIL")) : 0.5f;
  static const int tail_bn = getenv("D2T_CONV_TAIL_BN") ? atoi(getenv("D2T_CONV_TAIL_BN")) : 64;
  int tail_from = -1;
  const int rounds = tiles / grid, rem = tiles - rounds * grid;
  if (p.split_tail && rounds >= 1 && rem > 0 && rem < tail_frac * grid && !abl_probe()) {
    const int main_mt = rounds * grid / nt;  // whole rows of tiles that fit into the whole rounds
    tail_from = main_mt * 256;
    q.M = tail_from;
    tiles = main_mt * nt;
  }
  if (grid > tiles) grid = tiles;
  const int abl = abl_probe();
  static const int stagger = getenv("D2T_CONV_STAGGER") ? atoi(getenv("D2T_CONV_STAGGER")) : 1;
  static const int loaders = getenv("D2T_CONV_LOADERS") ? atoi(getenv("D2T_CONV_LOADERS")) : 4;
  if (abl == 1) hipLaunchKernelGGL(conv_bf16x3p_probe_no_dma, dim3(grid), dim3(768), 0, s, p2);
  else if (abl == 2) hipLaunchKernelGGL(conv_bf16x3p_probe_no_mfma, dim3(grid), dim3(768), 0, s, p2);
  else if (abl == 3) hipLaunchKernelGGL(conv_bf16x3p_probe_dma_only, dim3(grid), dim3(768), 0, s, p2);
  else if (stagger && p.K == 4608 && p.Cout == 512) hipLaunchKernelGGL(conv_bf16x3p_256x128_s_k4608, dim3(grid), dim3(768), 0, s, p2);
  else if (stagger) hipLaunchKernelGGL(conv_bf16x3p_256x128_s, dim3(grid), dim3(768), 0, s, p2);
  else if (loaders == 0 && p.K == 4608 && p.Cout == 512) hipLaunchKernelGGL(conv_bf16x3p_256x128_w8_k4608, dim3(grid), dim3(512), 0, s, p2);
  else if (loaders == 0) hipLaunchKernelGGL(conv_bf16x3p_256x128_w8, dim3(grid), dim3(512), 0, s, p2);
  else if (p.K == 4608 && p.Cout == 512) hipLaunchKernelGGL(conv_bf16x3p_256x128_k4608, dim3(grid), dim3(768), 0, s, p2);
  else hipLaunchKernelGGL(conv_bf16x3p_256x128, dim3(grid), dim3(768), 0, s, p2);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess || tail_from < 0) return e;
  ConvP t = p;
  t.m_base = tail_from;
  return launch_conv_bf16x3g_rows(t, tail_bn, s);
}

}  // namespace d2t
