// Kernels of the training step (train.hip): everything module.train() + loss.backward() needs beyond the
// forward GEMM/convolution kernel, all fp32.
//
//   wgrad_kernel        weight gradients of Conv2d / Linear: the "TN" GEMM  dW[co][tap][ci] = sum_p dz[p][co] * x[p+tap][ci]
//                       on the fp32 MFMA, split over pixel chunks (deterministic two-pass reduction)
//   colreduce_kernel    per-channel sums over rows: BatchNorm batch statistics, BatchNorm / LayerNorm parameter
//                       gradients, bias gradients
//   bn_*                BatchNorm2d in training mode (batch statistics, running-statistics update, backward)
//   ln_*                LayerNorm forward with saved statistics, backward
//   attn_*              softmax attention forward (saving the probabilities) and backward, causal / key-padding masks
//   maxpool_bwd, relu/gelu backward, embedding backward, small data-movement helpers
//
// Reference semantics: torch autograd of feature_extractor/resnet.py:205-245, seq_modeling/vit/vision_transformer.py:26-122,
// prediction_head/tfm.py:103-118 as driven by engine/training.py:76-164.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "conv_common.h"
#include "kernels.h"

namespace d2t {

namespace {
__device__ __forceinline__ float wsum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wmax(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
constexpr int EW_THREADS = 256;
inline int ew_grid(size_t n4) { return (int)std::min<size_t>((n4 + EW_THREADS - 1) / EW_THREADS, 65535u * 16u); }
}  // namespace

// ---------------------------------------------------------------------------
// wgrad: out[z][tap][m][n] = sum_{p in chunk z} A[p][m] * X[src(p, tap)][n]
// Block tile BM x BN, 4 waves (2x2), K-step = 32 rows; both operand tiles are stored [k][col] in LDS exactly as
// they lie in memory (rows = pixels, contiguous channels), and v_mfma_f32_32x32x2_f32 wants A[i][k] / B[k][j] with
// i, j = lane & 31: consecutive lanes read consecutive floats of one LDS row -> conflict-free ds_read_b32.
// ---------------------------------------------------------------------------
template <int BM, int BN>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradP p) {
  constexpr int BK = 32;
  constexpr int WTM = BM / 2, WTN = BN / 2, MI = WTM / 32, NJ = WTN / 32;
  constexpr int AV = BM / 4, BV = BN / 4;               // float4 per tile row
  constexpr int ALD = (BK * AV) / 256, BLD = (BK * BV) / 256;  // float4 loads per thread per stage
  static_assert(ALD >= 1 && BLD >= 1, "tile too small");
  __shared__ __attribute__((aligned(16))) float As[2][BK][BM];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK][BN];

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int tiles_n = (p.N + BN - 1) / BN;
  const int m0 = (blockIdx.x / tiles_n) * BM, n0 = (blockIdx.x % tiles_n) * BN;
  const int tap = blockIdx.y, kh = tap / p.KW, kw = tap % p.KW;
  const int z = blockIdx.z;
  const long long r_begin = (long long)z * p.chunk;
  const long long r_end = r_begin + p.chunk < p.P ? r_begin + p.chunk : p.P;
  const int ohow = p.OH * p.OW;

  float4 ra[ALD], rb[BLD];
  auto fetch = [&](long long r0) {
#pragma unroll
    for (int i = 0; i < ALD; ++i) {
      const int idx = tid + i * 256, row = idx / AV, c4 = (idx % AV) * 4;
      const long long r = r0 + row;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r < r_end && m0 + c4 < p.M) v = *reinterpret_cast<const float4*>(p.a + r * p.lda + m0 + c4);
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < BLD; ++i) {
      const int idx = tid + i * 256, row = idx / BV, c4 = (idx % BV) * 4;
      const long long r = r0 + row;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r < r_end && n0 + c4 < p.N) {
        long long src = r;
        bool ok = true;
        if (p.geom) {
          const int b = (int)(r / ohow), rem = (int)(r - (long long)b * ohow);
          const int oh = rem / p.OW, ow = rem - oh * p.OW;
          const int ih = oh * p.SH - p.PH + kh, iw = ow * p.SW - p.PW + kw;
          ok = (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
          src = ((long long)b * p.H + ih) * p.W + iw;
        }
        if (ok) v = *reinterpret_cast<const float4*>(p.b + src * p.ldb + n0 + c4);
      }
      rb[i] = v;
    }
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int i = 0; i < ALD; ++i) {
      const int idx = tid + i * 256, row = idx / AV, c4 = (idx % AV) * 4;
      *reinterpret_cast<float4*>(&As[buf][row][c4]) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < BLD; ++i) {
      const int idx = tid + i * 256, row = idx / BV, c4 = (idx % BV) * 4;
      *reinterpret_cast<float4*>(&Bs[buf][row][c4]) = rb[i];
    }
  };

  const int wm = wave >> 1, wn = wave & 1, r = lane & 31, h = lane >> 5;
  f32x16 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int steps = (int)((r_end - r_begin + BK - 1) / BK);
  if (steps > 0) {
    fetch(r_begin);
    stash(0);
  }
  __syncthreads();
  for (int st = 0; st < steps; ++st) {
    const int cur = st & 1;
    if (st + 1 < steps) fetch(r_begin + (long long)(st + 1) * BK);
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      float fa[MI], fb[NJ];
#pragma unroll
      for (int i = 0; i < MI; ++i) fa[i] = As[cur][2 * kk + h][wm * WTM + i * 32 + r];
#pragma unroll
      for (int j = 0; j < NJ; ++j) fb[j] = Bs[cur][2 * kk + h][wn * WTN + j * 32 + r];
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    if (st + 1 < steps) stash(cur ^ 1);
    __syncthreads();
  }
  float* out = p.part + ((size_t)z * p.taps + tap) * p.M * p.N;
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int n = n0 + wn * WTN + j * 32 + r;
      if (n >= p.N) continue;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int m = m0 + wm * WTM + i * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
        if (m < p.M) out[(size_t)m * p.N + n] = acc[i][j][reg];
      }
    }
}

// ---------------------------------------------------------------------------
// Split-bf16 weight gradient (conv_precision = bf16x3): the same TN GEMM on v_mfma_f32_32x32x16_bf16 with
// dz = hi + lo, x = hi + lo and three MFMAs per product.  Both operand tiles stay pixel-major in LDS (rows = pixels,
// exactly as they are loaded and split); the MFMA wants, per lane, eight consecutive K (= pixel) values of ONE column,
// which is what gfx950's transposing LDS read delivers: ds_read_b64_tr_b16 hands lane i of a 16-lane group column i of
// a 4-row x 16-column block.  Two of them per fragment.  LDS rows are 320 B (256 B of data + 64 B pad): a 32-lane
// half reads 4 rows x 64 B, and a row stride of 64 (mod 256) bytes makes those 256 bytes hit all 64 banks once.
// Block tile 128 x 128, 4 waves (wave tile 64 x 64), K-step = 16 pixels, double-buffered, 40 KB of LDS.
// ---------------------------------------------------------------------------
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef short s4_t __attribute__((ext_vector_type(4)));
typedef short s8_t __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s4_t* lds_s4_ptr;

__device__ __forceinline__ void split4_bf16(const float4 v, uint2& hi, uint2& lo) {
  const unsigned x0 = __float_as_uint(v.x), x1 = __float_as_uint(v.y), x2 = __float_as_uint(v.z), x3 = __float_as_uint(v.w);
  hi.x = (x0 >> 16) | (x1 & 0xFFFF0000u);
  hi.y = (x2 >> 16) | (x3 & 0xFFFF0000u);
  const __bf16 l0 = (__bf16)(v.x - __uint_as_float(x0 & 0xFFFF0000u)), l1 = (__bf16)(v.y - __uint_as_float(x1 & 0xFFFF0000u));
  const __bf16 l2 = (__bf16)(v.z - __uint_as_float(x2 & 0xFFFF0000u)), l3 = (__bf16)(v.w - __uint_as_float(x3 & 0xFFFF0000u));
  lo.x = (unsigned)*reinterpret_cast<const unsigned short*>(&l0) | ((unsigned)*reinterpret_cast<const unsigned short*>(&l1) << 16);
  lo.y = (unsigned)*reinterpret_cast<const unsigned short*>(&l2) | ((unsigned)*reinterpret_cast<const unsigned short*>(&l3) << 16);
}

__global__ __launch_bounds__(256) void wgrad_bf16x3_kernel(const WgradP p) {
  constexpr int BM = 128, BN = 128, BK = 16, LDR = 160;  // LDR: uint16 per LDS row (320 B)
  constexpr int PLANE = BK * LDR;                         // uint16 per plane per stage
  __shared__ __attribute__((aligned(16))) unsigned short sm[2][4][PLANE];  // [stage][A_hi, A_lo, B_hi, B_lo]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int tiles_n = (p.N + BN - 1) / BN;
  const int m0 = (blockIdx.x / tiles_n) * BM, n0 = (blockIdx.x % tiles_n) * BN;
  const int tap = blockIdx.y, kh = tap / p.KW, kw = tap % p.KW;
  const long long r_begin = (long long)blockIdx.z * p.chunk;
  const long long r_end = r_begin + p.chunk < p.P ? r_begin + p.chunk : p.P;
  const int ohow = p.OH * p.OW;
  // staging: 16 rows x 32 float4 per operand = 512 float4 -> two per thread
  float4 ra[2], rb[2];
  auto fetch = [&](long long r0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = tid + i * 256, row = idx >> 5, c4 = (idx & 31) * 4;
      const long long r = r0 + row;
      float4 va = make_float4(0.f, 0.f, 0.f, 0.f), vb = va;
      if (r < r_end) {
        if (m0 + c4 < p.M) va = *reinterpret_cast<const float4*>(p.a + r * p.lda + m0 + c4);
        if (n0 + c4 < p.N) {
          long long src = r;
          bool ok = true;
          if (p.geom) {
            const int b = (int)(r / ohow), rem = (int)(r - (long long)b * ohow);
            const int oh = rem / p.OW, ow = rem - oh * p.OW;
            const int ih = oh * p.SH - p.PH + kh, iw = ow * p.SW - p.PW + kw;
            ok = (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
            src = ((long long)b * p.H + ih) * p.W + iw;
          }
          if (ok) vb = *reinterpret_cast<const float4*>(p.b + src * p.ldb + n0 + c4);
        }
      }
      ra[i] = va; rb[i] = vb;
    }
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = tid + i * 256, row = idx >> 5, c4 = (idx & 31) * 4;
      uint2 hi, lo;
      split4_bf16(ra[i], hi, lo);
      *reinterpret_cast<uint2*>(&sm[buf][0][row * LDR + c4]) = hi;
      *reinterpret_cast<uint2*>(&sm[buf][1][row * LDR + c4]) = lo;
      split4_bf16(rb[i], hi, lo);
      *reinterpret_cast<uint2*>(&sm[buf][2][row * LDR + c4]) = hi;
      *reinterpret_cast<uint2*>(&sm[buf][3][row * LDR + c4]) = lo;
    }
  };
  const int wm = wave >> 1, wn = wave & 1, r = lane & 31, h = lane >> 5;
  // transposing read: group g = lane / 16 covers columns 16 (g & 1) .. +15 of a 32-column fragment and K rows 8 (g >> 1) .. +7
  const int g = lane >> 4, l16 = lane & 15;
  const int tr_off = ((8 * (g >> 1) + (l16 >> 2)) * LDR + 16 * (g & 1) + 4 * (l16 & 3));  // uint16 units, rows +0..3
  auto frag = [&](const unsigned short* plane, int col0) -> bf16x8_t {
    const unsigned short* q = plane + tr_off + col0;
    const s4_t lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(q));
    const s4_t hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(q + 4 * LDR));
    s8_t v = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
    return __builtin_bit_cast(bf16x8_t, v);
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const int steps = (int)((r_end - r_begin + BK - 1) / BK);
  if (steps > 0) { fetch(r_begin); stash(0); }
  __syncthreads();
  for (int st = 0; st < steps; ++st) {
    const int cur = st & 1;
    if (st + 1 < steps) fetch(r_begin + (long long)(st + 1) * BK);
    bf16x8_t ah[2], al[2], bh[2], bl[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      ah[i] = frag(sm[cur][0], wm * 64 + i * 32);
      al[i] = frag(sm[cur][1], wm * 64 + i * 32);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      bh[j] = frag(sm[cur][2], wn * 64 + j * 32);
      bl[j] = frag(sm[cur][3], wn * 64 + j * 32);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
      }
    if (st + 1 < steps) stash(cur ^ 1);
    __syncthreads();
  }
  float* out = p.part + ((size_t)blockIdx.z * p.taps + tap) * p.M * p.N;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + wn * 64 + j * 32 + r;
      if (n >= p.N) continue;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int m = m0 + wm * 64 + i * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
        if (m < p.M) out[(size_t)m * p.N + n] = acc[i][j][reg];
      }
    }
}

// ---------------------------------------------------------------------------
// Split-bf16 weight gradient on operand RECORDS (the convolution layers whose dz and input already exist as
// [pixel][32 x hi | 32 x lo] records: the BatchNorm kernels write them for the forward / data-gradient convolutions).
// Same products as wgrad_bf16x3_kernel -- lo*hi, hi*lo, hi*hi into one fp32 accumulator -- but nothing is split or staged
// through registers: a K-step's 16 pixel rows of both operands travel global -> LDS by LDS-DMA (16 bytes per lane), three
// stages deep with a counted vmcnt wait and ONE raw barrier per step (the scheme of conv_bf16x3p.hip), and three blocks
// share a CU so that one block's barrier is covered by the others' MFMAs.
//   LDS stage = A [16 rows][512 B] | B [16 rows][512 B]; a row = the tile's four records of one pixel = 32 chunks of
//   16 bytes (chunk 8 g + 0..3: hi of group g, 8 g + 4..7: lo).  Row stride 512 B puts every row on the same banks, so
//   chunk c of row r is stored at position c ^ ((r & 3) << 2): the four rows of a transposing read (ds_read_b64_tr_b16:
//   4 rows x 64 B per 32 lanes) then start 64 B apart modulo 256 B and cover the 64 banks once.  The LDS-DMA writes lane
//   l of a wave at base + 16 l, so the swizzle is applied on the SOURCE side: the lane fetches the chunk that belongs at
//   its position.
// ---------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void* lds_void_ptr;
template <int N>
__device__ __forceinline__ void wg_wait_vm() {
  __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));
}

// chunk swizzle of an LDS row of ROWB bytes: the four consecutive rows of a transposing read must start in four different
// 64-byte slots modulo 256 bytes.  Rows of 256 bytes or more all start in the same slot: XOR the row's low two bits into
// bits 2-3 of the chunk index; 128-byte rows alternate between two slots: flip bit 2 (hi <-> lo half) on rows 2, 3 mod 4.
template <int ROWB>
__device__ __forceinline__ int wg_swz(int row) {
  static_assert(ROWB >= 128, "a row holds at least one record");
  return ROWB >= 256 ? (row & 3) << 2 : ((row >> 1) & 1) << 2;
}

template <int MI, int NJ, int WM, int WN, int ABL = 0>
// (ABL: timing probes -- 1 no MFMAs, 2 no fragment reads, 3 no LDS-DMA, 4 plain ds_read_b64, 5 the step's LDS-DMA issued
// in one burst behind the barrier)
// wave tile (32 MI) x (32 NJ), WM x WN waves.  <4,2,2,4> 256 x 256 with 512 threads, one block per CU (a third less
// L2 -> LDS traffic per MFMA than 256 x 128); <4,2,2,2> 256 x 128, two; <2,2,2,2> 128 x 128, three; and for the narrow
// layers at the front of the network <2,2,2,1> 128 x 64 (two waves) and <2,1,1,1> 64 x 32 (one wave)
// (second launch bound = waves per SIMD the register budget must allow: 3 for the 128 x 128 tile, 2 otherwise)
__global__ __launch_bounds__(64 * WM * WN, WM * WN == 4 && MI == 2 ? 3 : 2)
void wgrad_rec_kernel(const WgradP p) {
  constexpr int BK = 16, NS = 3, BM = 32 * MI * WM, BN = 32 * NJ * WN, NT = 64 * WM * WN;
  constexpr int AROWB = BM * 4, BROWB = BN * 4;      // bytes per LDS row: the tile's records of one pixel
  constexpr int AOPB = BK * AROWB, BOPB = BK * BROWB, STAGE = AOPB + BOPB;
  constexpr int ACH = AROWB / 16, BCH = BROWB / 16;  // 16-byte chunks per row
  constexpr int ARPI = NT / ACH, ANI = BK / ARPI;    // rows per block-wide DMA instruction, instructions per step
  constexpr int BRPI = NT / BCH, BNI = BK / BRPI;
  constexpr int PIECES = ANI + BNI, IBYTES = NT * 16;
  static_assert(ARPI % 4 == 0 && BRPI % 4 == 0, "the source-side swizzle needs (row + k RPI) & 3 == row & 3");
  __shared__ __attribute__((aligned(1024))) unsigned char smem[NS * STAGE];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int tiles_n = p.N / BN;
  const int m0 = (blockIdx.x / tiles_n) * BM, n0 = (blockIdx.x % tiles_n) * BN;
  const int tap = blockIdx.y, kh = tap / p.KW, kw = tap % p.KW;
  const long long r_begin = (long long)blockIdx.z * p.chunk;
  const long long r_end = r_begin + p.chunk < p.P ? r_begin + p.chunk : p.P;
  const int KT = (int)((r_end - r_begin + BK - 1) / BK);
  const unsigned char* zero = reinterpret_cast<const unsigned char*>(p.zero) + (lane & 15) * 16;

  // ---- LDS-DMA source side.  A: rows arow + ARPI i (i < ANI) of every stage, chunk position acpos; B: rows brow + BRPI i ----
  const int arow = tid / ACH, acpos = tid % ACH, brow = tid / BCH, bcpos = tid % BCH;
  const int acsrc = acpos ^ wg_swz<AROWB>(arow), bcsrc = bcpos ^ wg_swz<BROWB>(brow);
  const size_t arow_b = (size_t)p.M * 4, brow_b = (size_t)p.N * 4;  // bytes per pixel row of the record arrays
  const unsigned char* a_src = reinterpret_cast<const unsigned char*>(p.a_rec) + (size_t)(r_begin + arow) * arow_b +
                               (size_t)(m0 / 32 + (acsrc >> 3)) * 128 + (acsrc & 7) * 16;
  const unsigned char* b_base = reinterpret_cast<const unsigned char*>(p.b_rec) + (size_t)(n0 / 32 + (bcsrc >> 3)) * 128 + (bcsrc & 7) * 16;
  long long rr = r_begin;  // first pixel row of the step about to be issued
  int pb[BNI], poh[BNI], pow_[BNI];
#pragma unroll
  for (int i = 0; i < BNI; ++i) {
    const long long r = r_begin + brow + BRPI * i;
    const int ohow = p.OH * p.OW;
    pb[i] = (int)(r / ohow);
    const int rem = (int)(r - (long long)pb[i] * ohow);
    poh[i] = rem / p.OW;
    pow_[i] = rem - poh[i] * p.OW;
  }
  long long rrb = r_begin;
  // one LDS-DMA instruction of the stage: pieces 0 .. ANI-1 = the A rows, ANI .. ANI+BNI-1 = the B rows
  auto issue_piece = [&](int stage, int k) {
    unsigned char* sa = smem + stage * STAGE + wave * 1024;  // wave-uniform bases; the hardware adds 16 * lane
    if (k < ANI) {
      const bool live = rr + arow + ARPI * k < r_end;
      const unsigned char* as = live ? a_src + (size_t)(ARPI * k) * arow_b : zero;
      __builtin_amdgcn_global_load_lds(as, (lds_void_ptr)(sa + k * IBYTES), 16, 0, 0);
      if (k == ANI - 1) { a_src += (size_t)BK * arow_b; rr += BK; }
    } else {
      const int i = k - ANI;
      const bool live = rrb + brow + BRPI * i < r_end;
      const int ih = poh[i] * p.SH - p.PH + kh, iw = pow_[i] * p.SW - p.PW + kw;
      const bool ok = live && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
      const unsigned char* bs = ok ? b_base + ((size_t)((long long)pb[i] * p.H + ih) * p.W + iw) * brow_b : zero;
      __builtin_amdgcn_global_load_lds(bs, (lds_void_ptr)(sa + AOPB + i * IBYTES), 16, 0, 0);
      pow_[i] += BK;  // advance to the same piece of the next step
      while (pow_[i] >= p.OW) {
        pow_[i] -= p.OW;
        if (++poh[i] == p.OH) { poh[i] = 0; ++pb[i]; }
      }
      if (i == BNI - 1) rrb += BK;
    }
  };
  auto issue = [&](int stage) {
#pragma unroll
    for (int k = 0; k < PIECES; ++k) issue_piece(stage, k);
  };

  // ---- fragment side ----
  const int wm = wave / WN, wn = wave % WN, r = lane & 31, h = lane >> 5;
  const int g16 = lane >> 4, l16 = lane & 15;
  // (the second transposing read of a fragment is four rows further down: same swizzle term)
  const int frow = 8 * (g16 >> 1) + (l16 >> 2), fsub = (g16 & 1) * 2 + ((l16 & 3) >> 1);
  const int fxa = wg_swz<AROWB>(frow), fxb = wg_swz<BROWB>(frow);
  int offa[MI][2], offb[NJ][2];  // [fragment][hi, lo]
#pragma unroll
  for (int part = 0; part < 2; ++part) {
#pragma unroll
    for (int i = 0; i < MI; ++i)
      offa[i][part] = frow * AROWB + (l16 & 1) * 8 + (((((wm * MI + i) << 3) | (part << 2) | fsub) ^ fxa) << 4);
#pragma unroll
    for (int j = 0; j < NJ; ++j)
      offb[j][part] = AOPB + frow * BROWB + (l16 & 1) * 8 + (((((wn * NJ + j) << 3) | (part << 2) | fsub) ^ fxb) << 4);
  }
  auto frag = [&](const unsigned char* q, int rowb) -> bf16x8_t {
    if (ABL == 4) {
      const s4_t a4 = *reinterpret_cast<const s4_t*>(q), b4 = *reinterpret_cast<const s4_t*>(q + 4 * rowb);
      s8_t v = {a4[0], a4[1], a4[2], a4[3], b4[0], b4[1], b4[2], b4[3]};
      return __builtin_bit_cast(bf16x8_t, v);
    }
    const s4_t lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(q));
    const s4_t hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(q + 4 * rowb));
    s8_t v = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
    return __builtin_bit_cast(bf16x8_t, v);
  };
  f32x16 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  if (KT > 0) issue(0);
  if (KT > 1) issue(1);
  int cur = 0, nxt2 = 2;
  constexpr bool burst = ABL == 3 || ABL == 5;
  bf16x8_t ah[MI], al[MI], bh[NJ], bl[NJ];
  for (int kt = 0; kt < KT; ++kt) {
    if (kt + 1 < KT) wg_wait_vm<PIECES>(); else wg_wait_vm<0>();  // this wave's pieces of step kt have landed
    __builtin_amdgcn_s_barrier();  // ... and everybody else's; nobody reads stage kt-1 any more
    if (burst && kt + 2 < KT && (ABL != 3 || kt + 2 < 3)) issue(nxt2);
    __builtin_amdgcn_sched_barrier(0);
    const unsigned char* st = smem + cur * STAGE;
    if (ABL != 2 || kt == 0) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        bh[j] = frag(st + offb[j][0], BROWB);
        bl[j] = frag(st + offb[j][1], BROWB);
      }
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        ah[i] = frag(st + offa[i][0], AROWB);
        al[i] = frag(st + offa[i][1], AROWB);
      }
    }
    if (ABL == 1) {
#pragma unroll
      for (int i = 0; i < MI; ++i) asm volatile("" ::"v"(ah[i]), "v"(al[i]));
#pragma unroll
      for (int j = 0; j < NJ; ++j) asm volatile("" ::"v"(bh[j]), "v"(bl[j]));
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        if (ABL != 1 || kt == 0) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
        // the LDS-DMA of step kt+2: one instruction behind each (i, j) group of MFMAs, the last PIECES groups of the step.
        // (In one burst behind the barrier the waves of a block queue up in the vector-memory path together and the
        // MFMAs wait behind them: 951 us on the dominant layer against 833 us this way; all at once after the first /
        // second row of groups: 867 / 847 us.)
        // The narrow tiles have more pieces than groups: PPG pieces behind each group from the first on.
        if (!burst && kt + 2 < KT) {
          constexpr int GROUPS = MI * NJ, PPG = (PIECES + GROUPS - 1) / GROUPS;
          constexpr int first = PPG == 1 ? GROUPS - PIECES : 0;
          const int g = i * NJ + j - first;
          if (g >= 0) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < PPG; ++q)
              if (g * PPG + q < PIECES) issue_piece(nxt2, g * PPG + q);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the fragment reads have returned before the next barrier (WAR on the stage)
    cur = cur == 2 ? 0 : cur + 1;
    nxt2 = nxt2 == 2 ? 0 : nxt2 + 1;
  }
  float* out = p.part + ((size_t)blockIdx.z * p.taps + tap) * p.M * p.N;
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int n = n0 + wn * 32 * NJ + j * 32 + r;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int m = m0 + wm * 32 * MI + i * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
        out[(size_t)m * p.N + n] = acc[i][j][reg];
      }
    }
}

// block tile of the record kernel: 0 = 128 x 128, 1 = 256 x 128, 2 = 256 x 256 (D2T_WGRAD_WIDE caps these three),
// 3 = 128 x 64 (Cin = 64), 4 = 64 x 32 (conv0_2: 32 -> 64 channels); -1: the channel counts fit none of them
int wgrad_rec_shape(int M, int N) {
  static const int mode = D2T_PROBE_ENV_STR("D2T_WGRAD_WIDE") ? D2T_PROBE_ENV("D2T_WGRAD_WIDE") : 2;
  static const bool narrow = !(D2T_PROBE_ENV_STR("D2T_WGRAD_NARROW") && D2T_PROBE_ENV("D2T_WGRAD_NARROW") == 0);
  if (M % 128 == 0 && N % 128 == 0) {
    if (mode >= 2 && M % 256 == 0 && N % 256 == 0) return 2;
    if (mode >= 1 && M % 256 == 0) return 1;
    return 0;
  }
  if (narrow && M % 128 == 0 && N == 64) return 3;
  if (narrow && M == 64 && N == 32) return 4;
  return -1;
}
hipError_t launch_wgrad(const WgradP& p, hipStream_t s) {
  if (p.M <= 0 || p.N <= 0 || p.P <= 0) return hipSuccess;
  if (p.M % 4 || p.N % 4 || p.lda % 4 || p.ldb % 4 || p.S < 1 || p.chunk < 1 || p.taps < 1) return hipErrorInvalidValue;
  if (p.a_rec) {  // record operands: see wgrad_rec_ok
    if (!p.b_rec || !p.zero || !p.geom || !p.bf16x3 || wgrad_rec_shape(p.M, p.N) < 0 || p.chunk % 16) return hipErrorInvalidValue;
    static const int abl = D2T_PROBE_ENV("D2T_WGRAD_ABL");
    const int shape = wgrad_rec_shape(p.M, p.N);
    if (shape == 2) {
      dim3 grid((p.M / 256) * (p.N / 256), p.taps, p.S);
      hipLaunchKernelGGL((wgrad_rec_kernel<4, 2, 2, 4, 0>), grid, dim3(512), 0, s, p);
    } else if (shape == 1) {
      dim3 grid((p.M / 256) * (p.N / 128), p.taps, p.S);
      if (abl == 1) hipLaunchKernelGGL((wgrad_rec_kernel<4, 2, 2, 2, 1>), grid, dim3(256), 0, s, p);
      else if (abl == 2) hipLaunchKernelGGL((wgrad_rec_kernel<4, 2, 2, 2, 2>), grid, dim3(256), 0, s, p);
      else if (abl == 3) hipLaunchKernelGGL((wgrad_rec_kernel<4, 2, 2, 2, 3>), grid, dim3(256), 0, s, p);
      else if (abl == 4) hipLaunchKernelGGL((wgrad_rec_kernel<4, 2, 2, 2, 4>), grid, dim3(256), 0, s, p);
      else if (abl == 5) hipLaunchKernelGGL((wgrad_rec_kernel<4, 2, 2, 2, 5>), grid, dim3(256), 0, s, p);
      else hipLaunchKernelGGL((wgrad_rec_kernel<4, 2, 2, 2, 0>), grid, dim3(256), 0, s, p);
    } else if (shape == 0) {
      dim3 grid((p.M / 128) * (p.N / 128), p.taps, p.S);
      hipLaunchKernelGGL((wgrad_rec_kernel<2, 2, 2, 2, 0>), grid, dim3(256), 0, s, p);
    } else if (shape == 3) {
      dim3 grid((p.M / 128) * (p.N / 64), p.taps, p.S);
      hipLaunchKernelGGL((wgrad_rec_kernel<2, 2, 2, 1, 0>), grid, dim3(128), 0, s, p);
    } else if (shape == 4) {
      dim3 grid((p.M / 64) * (p.N / 32), p.taps, p.S);
      hipLaunchKernelGGL((wgrad_rec_kernel<2, 1, 1, 1, 0>), grid, dim3(64), 0, s, p);
    } else {
      return hipErrorInvalidValue;
    }
  } else if (p.M <= 64 || p.N <= 64) {
    dim3 grid(((p.M + 63) / 64) * ((p.N + 63) / 64), p.taps, p.S);
    hipLaunchKernelGGL((wgrad_kernel<64, 64>), grid, dim3(256), 0, s, p);
  } else {
    dim3 grid(((p.M + 127) / 128) * ((p.N + 127) / 128), p.taps, p.S);
    if (p.bf16x3) hipLaunchKernelGGL(wgrad_bf16x3_kernel, grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((wgrad_kernel<128, 128>), grid, dim3(256), 0, s, p);
  }
  return hipGetLastError();
}

// dst = (accumulate ? dst : 0) + sum_z part[z][tap][m][n];  layout 0: dst[m][n] (taps == 1);  layout 1: OIHW dst[m][n][tap]
__global__ void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dst, int S, int taps, int M, int N,
                                    int layout, int accumulate) {
  const size_t total = (size_t)taps * M * N;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    float v = 0.f;
#pragma unroll 8
    for (int z = 0; z < S; ++z) v += part[(size_t)z * total + i];
    const int n = (int)(i % N), m = (int)((i / N) % M), tap = (int)(i / ((size_t)M * N));
    const size_t o = layout == 1 ? ((size_t)m * N + n) * taps + tap : (size_t)m * N + n;
    dst[o] = accumulate ? dst[o] + v : v;
  }
}
// few outputs, many partials (the stem's 288 filter taps over ~2000 pixel chunks, conv0_2's 18432 over 227): one block per output, its 256 threads
// take every 256th partial and a fixed-order tree adds them (a thread per output would walk the partials serially)
__global__ __launch_bounds__(256) void wgrad_reduce_small_kernel(const float* __restrict__ part, float* __restrict__ dst, int S,
                                                                 int taps, int M, int N, int layout, int accumulate) {
  __shared__ float red[256];
  const size_t total = (size_t)taps * M * N, i = blockIdx.x;
  float v = 0.f;
  for (int z = threadIdx.x; z < S; z += 256) v += part[(size_t)z * total + i];
  red[threadIdx.x] = v;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const int n = (int)(i % N), m = (int)((i / N) % M), tap = (int)(i / ((size_t)M * N));
    const size_t o = layout == 1 ? ((size_t)m * N + n) * taps + tap : (size_t)m * N + n;
    dst[o] = accumulate ? dst[o] + red[0] : red[0];
  }
}
hipError_t launch_wgrad_reduce(const float* part, float* dst, int S, int taps, int M, int N, int layout, int accumulate,
                               hipStream_t s) {
  const size_t total = (size_t)taps * M * N;
  if ((total <= 4096 && S >= 256) || (total <= 65536 && S >= 128)) {
    hipLaunchKernelGGL(wgrad_reduce_small_kernel, dim3((unsigned)total), dim3(256), 0, s, part, dst, S, taps, M, N, layout,
                       accumulate);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 4096)), dim3(256), 0, s, part,
                     dst, S, taps, M, N, layout, accumulate);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Column reductions over the rows of row-major [R][C] matrices -> part[chunk][2][C]
// ---------------------------------------------------------------------------
// rows per block: about 2048 blocks per launch (eight per CU -- these kernels are HBM streams and one block per CU, which
// 2048-row chunks gave on the large maps, kept them near 2 TB/s), at least 128 rows each
static int colreduce_rows(long long R, int C) {
  const long long col_blocks = (C + 63) / 64, chunks = std::max<long long>(1, 2048 / col_blocks);
  const long long rows = ((R + chunks - 1) / chunks + 15) / 16 * 16;
  return (int)std::max<long long>(128, rows);
}
__global__ __launch_bounds__(256) void colreduce_kernel(const ColRedP p, int rows_per_block) {
  __shared__ float red[16][2][64];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int c = blockIdx.x * 64 + tx * 4;
  const long long r0 = (long long)blockIdx.y * rows_per_block;
  const long long r1 = r0 + rows_per_block < p.R ? r0 + rows_per_block : p.R;
  float s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f};
  if (c < p.C) {
    float mu[4] = {0, 0, 0, 0}, rs[4] = {1, 1, 1, 1};
    if (p.mode == CR_BN_BWD) {
#pragma unroll
      for (int k = 0; k < 4; ++k) { mu[k] = p.mean[c + k]; rs[k] = p.rstd[c + k]; }
    }
#pragma unroll 8
    for (long long r = r0 + ty; r < r1; r += 16) {
      const size_t off = (size_t)r * p.C + c;
      const float4 a4 = *reinterpret_cast<const float4*>(p.a + off);
      const float a[4] = {a4.x, a4.y, a4.z, a4.w};
      if (p.mode == CR_SUM) {
#pragma unroll
        for (int k = 0; k < 4; ++k) s0[k] += a[k];
      } else if (p.mode == CR_SUM_SQ) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { s0[k] += a[k]; s1[k] = fmaf(a[k], a[k], s1[k]); }
      } else if (p.mode == CR_BN_BWD) {  // a = dy, y = post-activation output (nullable), z = pre-BN conv output
        const float4 z4 = *reinterpret_cast<const float4*>(p.z + off);
        const float z[4] = {z4.x, z4.y, z4.z, z4.w};
        float g[4] = {a[0], a[1], a[2], a[3]};
        if (p.y) {
          const float4 y4 = *reinterpret_cast<const float4*>(p.y + off);
          const float y[4] = {y4.x, y4.y, y4.z, y4.w};
#pragma unroll
          for (int k = 0; k < 4; ++k) g[k] = y[k] > 0.f ? g[k] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) { s0[k] += g[k]; s1[k] = fmaf(g[k], (z[k] - mu[k]) * rs[k], s1[k]); }
      } else {  // CR_LN_BWD: a = dy, z = LayerNorm input, mean / rstd per ROW
        const float4 z4 = *reinterpret_cast<const float4*>(p.z + off);
        const float z[4] = {z4.x, z4.y, z4.z, z4.w};
        const float m = p.mean[r], q = p.rstd[r];
#pragma unroll
        for (int k = 0; k < 4; ++k) { s0[k] += a[k]; s1[k] = fmaf(a[k], (z[k] - m) * q, s1[k]); }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) { red[ty][0][tx * 4 + k] = s0[k]; red[ty][1][tx * 4 + k] = s1[k]; }
  __syncthreads();
  if (threadIdx.x < 128) {
    const int which = threadIdx.x >> 6, col = threadIdx.x & 63;
    float v = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t) v += red[t][which][col];
    const int cc = blockIdx.x * 64 + col;
    if (cc < p.C) p.part[((size_t)blockIdx.y * 2 + which) * p.C + cc] = v;
  }
}
int colreduce_chunks(long long R, int C) {
  const int rows = colreduce_rows(R, C);
  return (int)((R + rows - 1) / rows);
}
hipError_t launch_colreduce(const ColRedP& p, hipStream_t s) {
  if (p.C % 4 || p.R <= 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(colreduce_kernel, dim3((p.C + 63) / 64, colreduce_chunks(p.R, p.C)), dim3(256), 0, s, p,
                     colreduce_rows(p.R, p.C));
  return hipGetLastError();
}
// out0[c] = (acc ? out0[c] : 0) + sum_chunks part[.][0][c];  out1 likewise (nullable)
// sums over the chunks of part[chunk][2][C] for 32 channels per 256-thread block: eight threads per channel take every
// eighth chunk (double accumulators), thread 0 of the eight adds their partial sums in a fixed order
__device__ __forceinline__ void chunk_sums(const float* __restrict__ part, int chunks, int C, int c, int lane8, double (*red)[8][32],
                                           double& a, double& b) {
  a = 0.0; b = 0.0;
  if (c < C) {
#pragma unroll 8
    for (int k = lane8; k < chunks; k += 8) { a += part[((size_t)k * 2) * C + c]; b += part[((size_t)k * 2 + 1) * C + c]; }
  }
  red[0][lane8][threadIdx.x & 31] = a;
  red[1][lane8][threadIdx.x & 31] = b;
  __syncthreads();
  if (lane8 == 0) {
    a = 0.0; b = 0.0;
#pragma unroll
    for (int t = 0; t < 8; ++t) { a += red[0][t][threadIdx.x & 31]; b += red[1][t][threadIdx.x & 31]; }
  }
}
__global__ __launch_bounds__(256) void colreduce_final_kernel(const float* __restrict__ part, int chunks, int C, float* out0,
                                                              float* out1, int accumulate) {
  __shared__ double red[2][8][32];
  const int c = blockIdx.x * 32 + (threadIdx.x & 31), lane8 = threadIdx.x >> 5;
  double a, b;
  chunk_sums(part, chunks, C, c, lane8, red, a, b);
  if (lane8 != 0 || c >= C) return;
  if (out0) out0[c] = (accumulate ? out0[c] : 0.f) + (float)a;
  if (out1) out1[c] = (accumulate ? out1[c] : 0.f) + (float)b;
}
hipError_t launch_colreduce_final(const float* part, int chunks, int C, float* out0, float* out1, int accumulate,
                                  hipStream_t s) {
  hipLaunchKernelGGL(colreduce_final_kernel, dim3((C + 31) / 32), dim3(256), 0, s, part, chunks, C, out0, out1, accumulate);
  return hipGetLastError();
}
// BatchNorm batch statistics from (sum, sum of squares) partials; running statistics updated in place
// (momentum 0.1, unbiased variance: nn.BatchNorm2d defaults used by resnet.py).
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ part, int chunks, int C, long long R, float eps,
                                                          float momentum, float* mean, float* rstd, float* run_mean,
                                                          float* run_var) {
  __shared__ double red[2][8][32];
  const int c = blockIdx.x * 32 + (threadIdx.x & 31), lane8 = threadIdx.x >> 5;
  double a, b;
  chunk_sums(part, chunks, C, c, lane8, red, a, b);
  if (lane8 != 0 || c >= C) return;
  const double m = a / (double)R;
  double var = b / (double)R - m * m;
  if (var < 0.0) var = 0.0;
  mean[c] = (float)m;
  rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (run_mean) {
    const double unb = R > 1 ? var * (double)R / (double)(R - 1) : var;
    run_mean[c] = (float)((1.0 - momentum) * run_mean[c] + momentum * m);
    run_var[c] = (float)((1.0 - momentum) * run_var[c] + momentum * unb);
  }
}
hipError_t launch_bn_finalize(const float* part, int chunks, int C, long long R, float eps, float momentum, float* mean,
                              float* rstd, float* run_mean, float* run_var, hipStream_t s) {
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 31) / 32), dim3(256), 0, s, part, chunks, C, R, eps, momentum, mean,
                     rstd, run_mean, run_var);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// BatchNorm apply (train forward) and backward apply, float4 over [R][C]
// ---------------------------------------------------------------------------
// the four values of flat float4 index i4 of a row-major [rows][C] tensor (C % 32 == 0) as split-bf16 record entries
// (conv_common.h plane_idx): what the LDS-DMA convolution kernels read
__device__ __forceinline__ void store_split4(uint16_t* planes, size_t i4, const float (&v)[4]) {
  uint16_t hi[4], lo[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) split_f32(v[k], hi[k], lo[k]);
  const size_t e = i4 * 4;
  uint16_t* dst = planes + (e >> 5) * 64 + (e & 31);
  *reinterpret_cast<uint2*>(dst) = make_uint2(hi[0] | (uint32_t)hi[1] << 16, hi[2] | (uint32_t)hi[3] << 16);
  *reinterpret_cast<uint2*>(dst + 32) = make_uint2(lo[0] | (uint32_t)lo[1] << 16, lo[2] | (uint32_t)lo[3] << 16);
}
__global__ void bn_apply_kernel(const float* __restrict__ z, const float* __restrict__ mean, const float* __restrict__ rstd,
                                const float* __restrict__ g, const float* __restrict__ b, const float* __restrict__ res,
                                float* __restrict__ y, size_t n4, int C, int relu, uint16_t* __restrict__ planes) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)((i * 4) % C);
    const float4 v = reinterpret_cast<const float4*>(z)[i];
    float o[4] = {v.x, v.y, v.z, v.w};
    float r4[4] = {0, 0, 0, 0};
    if (res) { const float4 t = reinterpret_cast<const float4*>(res)[i]; r4[0] = t.x; r4[1] = t.y; r4[2] = t.z; r4[3] = t.w; }
    // (per-channel parameters as 16-byte loads: c and C are multiples of 4)
    const float4 m4 = *reinterpret_cast<const float4*>(mean + c), q4 = *reinterpret_cast<const float4*>(rstd + c);
    const float4 g4 = *reinterpret_cast<const float4*>(g + c), b4 = *reinterpret_cast<const float4*>(b + c);
    const float mm[4] = {m4.x, m4.y, m4.z, m4.w}, qq[4] = {q4.x, q4.y, q4.z, q4.w};
    const float gg[4] = {g4.x, g4.y, g4.z, g4.w}, bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float t = (o[k] - mm[k]) * qq[k] * gg[k] + bb[k] + r4[k];
      o[k] = relu ? fmaxf(t, 0.f) : t;
    }
    reinterpret_cast<float4*>(y)[i] = make_float4(o[0], o[1], o[2], o[3]);
    if (planes) store_split4(planes, i, o);
  }
}
hipError_t launch_bn_apply(const float* z, const float* mean, const float* rstd, const float* g, const float* b,
                           const float* res, float* y, long long R, int C, int relu, hipStream_t s, uint16_t* planes) {
  const size_t n4 = (size_t)R * C / 4;
  if (planes && C % 32 != 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(bn_apply_kernel, dim3(ew_grid(n4)), dim3(EW_THREADS), 0, s, z, mean, rstd, g, b, res, y, n4, C, relu, planes);
  return hipGetLastError();
}
// dz = gamma*rstd * (g - s0/R - xhat * s1/R),  g = dy * (y > 0) (y nullable = no ReLU);  gout (nullable) receives g
// (the gradient of the residual branch that was added before the ReLU).
__global__ void bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ y, const float* __restrict__ z,
                                    const float* __restrict__ mean, const float* __restrict__ rstd,
                                    const float* __restrict__ gamma, const float* __restrict__ s0,
                                    const float* __restrict__ s1, float invR, float* __restrict__ dz,
                                    float* __restrict__ gout, size_t n4, int C, uint16_t* __restrict__ planes) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)((i * 4) % C);
    const float4 d4 = reinterpret_cast<const float4*>(dy)[i];
    const float4 z4 = reinterpret_cast<const float4*>(z)[i];
    float g[4] = {d4.x, d4.y, d4.z, d4.w};
    const float zz[4] = {z4.x, z4.y, z4.z, z4.w};
    if (y) {
      const float4 y4 = reinterpret_cast<const float4*>(y)[i];
      const float yy[4] = {y4.x, y4.y, y4.z, y4.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) g[k] = yy[k] > 0.f ? g[k] : 0.f;
    }
    float o[4];
    const float4 m4 = *reinterpret_cast<const float4*>(mean + c), q4 = *reinterpret_cast<const float4*>(rstd + c);
    const float4 g4 = *reinterpret_cast<const float4*>(gamma + c);
    const float4 a4 = *reinterpret_cast<const float4*>(s0 + c), b4 = *reinterpret_cast<const float4*>(s1 + c);
    const float mm[4] = {m4.x, m4.y, m4.z, m4.w}, qq[4] = {q4.x, q4.y, q4.z, q4.w}, gm[4] = {g4.x, g4.y, g4.z, g4.w};
    const float sa[4] = {a4.x, a4.y, a4.z, a4.w}, sb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float xh = (zz[k] - mm[k]) * qq[k];
      o[k] = gm[k] * qq[k] * (g[k] - sa[k] * invR - xh * sb[k] * invR);
    }
    reinterpret_cast<float4*>(dz)[i] = make_float4(o[0], o[1], o[2], o[3]);
    if (planes) store_split4(planes, i, o);
    if (gout) reinterpret_cast<float4*>(gout)[i] = make_float4(g[0], g[1], g[2], g[3]);
  }
}
hipError_t launch_bn_bwd_apply(const float* dy, const float* y, const float* z, const float* mean, const float* rstd,
                               const float* gamma, const float* s0, const float* s1, float* dz, float* gout, long long R,
                               int C, hipStream_t s, uint16_t* planes) {
  const size_t n4 = (size_t)R * C / 4;
  if (planes && C % 32 != 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(ew_grid(n4)), dim3(EW_THREADS), 0, s, dy, y, z, mean, rstd, gamma, s0, s1,
                     1.f / (float)R, dz, gout, n4, C, planes);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Elementwise: out = f(a, b)
// ---------------------------------------------------------------------------
__global__ void ew_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, size_t n4,
                          int op) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const float4 x4 = reinterpret_cast<const float4*>(a)[i];
    float x[4] = {x4.x, x4.y, x4.z, x4.w}, y[4] = {0, 0, 0, 0}, o[4];
    if (b) { const float4 y4 = reinterpret_cast<const float4*>(b)[i]; y[0] = y4.x; y[1] = y4.y; y[2] = y4.z; y[3] = y4.w; }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      switch (op) {
        case EW_ADD: o[k] = x[k] + y[k]; break;
        case EW_RELU_BWD: o[k] = y[k] > 0.f ? x[k] : 0.f; break;  // a = dy, b = forward output
        case EW_RELU: o[k] = fmaxf(x[k], 0.f); break;
        case EW_GELU: o[k] = 0.5f * x[k] * (1.f + erff(x[k] * 0.70710678118654752440f)); break;
        case EW_GELU_BWD: {  // a = dy, b = pre-activation u: d/du [u * Phi(u)] = Phi(u) + u * phi(u)
          const float u = y[k];
          const float cdf = 0.5f * (1.f + erff(u * 0.70710678118654752440f));
          const float pdf = 0.3989422804014327f * expf(-0.5f * u * u);
          o[k] = x[k] * (cdf + u * pdf);
          break;
        }
        default: o[k] = x[k];
      }
    }
    reinterpret_cast<float4*>(out)[i] = make_float4(o[0], o[1], o[2], o[3]);
  }
}
hipError_t launch_ew(const float* a, const float* b, float* out, size_t n, int op, hipStream_t s) {
  if (n % 4) return hipErrorInvalidValue;
  if (!n) return hipSuccess;
  hipLaunchKernelGGL(ew_kernel, dim3(ew_grid(n / 4)), dim3(EW_THREADS), 0, s, a, b, out, n / 4, op);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// MaxPool2d(k=2) backward, gather form (deterministic): every input pixel collects dy of the windows whose
// FIRST maximum (scan order kh, kw; padding = -inf) it is -- the rule of torch's max_pool2d_with_indices.
// ---------------------------------------------------------------------------
__global__ void maxpool_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx, int B,
                                   int H, int W, int C, int OH, int OW, int SH, int SW, int PH, int PW, int KW) {
  const int C4 = C >> 2;  // four channels per thread (C % 4 == 0): one set of index arithmetic, 16-byte accesses
  const size_t total = (size_t)B * H * W * C4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4) * 4;
    const int w = (int)((i / C4) % W), h = (int)((i / ((size_t)C4 * W)) % H), b = (int)(i / ((size_t)C4 * W * H));
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    // windows (oh, ow) that contain (h, w): oh*SH - PH <= h <= oh*SH - PH + 1
    const int oh_lo = h + PH - 1 > 0 ? (h + PH - 1 + SH - 1) / SH : 0, oh_hi = min((h + PH) / SH, OH - 1);
    const int ow_lo = w + PW - (KW - 1) > 0 ? (w + PW - (KW - 1) + SW - 1) / SW : 0, ow_hi = min((w + PW) / SW, OW - 1);
    for (int oh = oh_lo; oh <= oh_hi; ++oh)
      for (int ow = ow_lo; ow <= ow_hi; ++ow) {
        float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        bool mine[4] = {false, false, false, false};  // is (h, w) the first maximum so far
        bool any = false;
        for (int kh = 0; kh < 2; ++kh)
          for (int kw = 0; kw < KW; ++kw) {
            const int ih = oh * SH - PH + kh, iw = ow * SW - PW + kw;
            if ((unsigned)ih >= (unsigned)H || (unsigned)iw >= (unsigned)W) continue;
            const float4 v4 = *reinterpret_cast<const float4*>(x + (((size_t)b * H + ih) * W + iw) * C + c);
            const float v[4] = {v4.x, v4.y, v4.z, v4.w};
            const bool here = ih == h && iw == w;
#pragma unroll
            for (int k = 0; k < 4; ++k)
              if (v[k] > best[k] || !any) { best[k] = v[k]; mine[k] = here; }
            any = true;
          }
        const float4 d4 = *reinterpret_cast<const float4*>(dy + (((size_t)b * OH + oh) * OW + ow) * C + c);
        const float d[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (mine[k]) acc[k] += d[k];
      }
    *reinterpret_cast<float4*>(dx + i * 4) = make_float4(acc[0], acc[1], acc[2], acc[3]);
  }
}
hipError_t launch_maxpool_bwd(const float* x, const float* dy, float* dx, int B, int H, int W, int C, int SH, int SW, int PH,
                              int PW, hipStream_t s, int KW) {  // window 2 x KW (KW = 1: VGG's (2,1) pools)
  const int OH = (H + 2 * PH - 2) / SH + 1, OW = (W + 2 * PW - KW) / SW + 1;
  if (C % 4) return hipErrorInvalidValue;
  const size_t total = (size_t)B * H * W * (C / 4);
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 1u << 20)), dim3(256), 0, s, x,
                     dy, dx, B, H, W, C, OH, OW, SH, SW, PH, PW, KW);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Fused cross-entropy (nn.CrossEntropyLoss(ignore_index, reduction='none') of engine/training.py:50-53,83,90): one wave per
// row of the [rows][V] logits.  Forward: lse = max + log(sum exp(x - max)); loss = lse - x[target] (0 where target ==
// ignore_index); backward: dx[v] = (exp(x[v] - lse) - [v == target]) * dloss, 0 for ignored rows.  Replaces torch's
// log_softmax + nll_loss pair (four passes over the logits) with one pass each way.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ce_fwd_kernel(const float* __restrict__ x, const int64_t* __restrict__ tgt,
                                                     float* __restrict__ loss, float* __restrict__ lse, int rows, int V,
                                                     long long ignore) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + wave;
  if (row >= rows) return;
  const float* xr = x + (size_t)row * V;
  float m = -INFINITY;
  for (int v = lane; v < V; v += 64) m = fmaxf(m, xr[v]);
  m = wmax(m);
  float sum = 0.f;
  for (int v = lane; v < V; v += 64) sum += expf(xr[v] - m);
  sum = wsum(sum);
  if (lane == 0) {
    const float l = m + logf(sum);
    const long long t = tgt[row];
    lse[row] = l;
    loss[row] = (t == ignore || t < 0 || t >= V) ? 0.f : l - xr[t];
  }
}
__global__ __launch_bounds__(256) void ce_bwd_kernel(const float* __restrict__ x, const int64_t* __restrict__ tgt,
                                                     const float* __restrict__ lse, const float* __restrict__ dloss,
                                                     float* __restrict__ dx, int rows, int V, long long ignore) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + wave;
  if (row >= rows) return;
  const float* xr = x + (size_t)row * V;
  float* dr = dx + (size_t)row * V;
  const long long t = tgt[row];
  const bool live = !(t == ignore || t < 0 || t >= V);
  const float g = live ? dloss[row] : 0.f, l = lse[row];
  for (int v = lane; v < V; v += 64) dr[v] = live ? (expf(xr[v] - l) - (v == t ? 1.f : 0.f)) * g : 0.f;
}
hipError_t launch_ce_fwd(const float* x, const int64_t* tgt, float* loss, float* lse, int rows, int V, long long ignore,
                         hipStream_t s) {
  if (rows <= 0) return hipSuccess;
  hipLaunchKernelGGL(ce_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, x, tgt, loss, lse, rows, V, ignore);
  return hipGetLastError();
}
hipError_t launch_ce_bwd(const float* x, const int64_t* tgt, const float* lse, const float* dloss, float* dx, int rows, int V,
                         long long ignore, hipStream_t s) {
  if (rows <= 0) return hipSuccess;
  hipLaunchKernelGGL(ce_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, x, tgt, lse, dloss, dx, rows, V, ignore);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// The discrete decisions of a training forward, for tests that replay them in the oracle (d2t_train_read_decision):
// ReLU keep masks (y > 0) and, per max-pool output element, which window element (kh*2 + kw) the backward routes the
// gradient to -- the same first-maximum scan as maxpool_bwd_kernel.
// ---------------------------------------------------------------------------
__global__ void relu_mask_kernel(const float* __restrict__ y, uint8_t* __restrict__ m, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) m[i] = y[i] > 0.f;
}
hipError_t launch_relu_mask(const float* y, uint8_t* m, size_t n, hipStream_t s) {
  hipLaunchKernelGGL(relu_mask_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 1u << 20)), dim3(256), 0, s, y, m, n);
  return hipGetLastError();
}
__global__ void pool_argmax_kernel(const float* __restrict__ x, uint8_t* __restrict__ k, int B, int H, int W, int C, int OH,
                                   int OW, int SH, int SW, int PH, int PW, int KW) {
  const size_t total = (size_t)B * OH * OW * C;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const int ow = (int)((i / C) % OW), oh = (int)((i / ((size_t)C * OW)) % OH), b = (int)(i / ((size_t)C * OW * OH));
    float best = -INFINITY;
    int bk = -1;
    for (int kh = 0; kh < 2; ++kh)
      for (int kw = 0; kw < KW; ++kw) {
        const int ih = oh * SH - PH + kh, iw = ow * SW - PW + kw;
        if ((unsigned)ih >= (unsigned)H || (unsigned)iw >= (unsigned)W) continue;
        const float v = x[(((size_t)b * H + ih) * W + iw) * C + c];
        if (v > best || bk < 0) { best = v; bk = kh * KW + kw; }
      }
    k[i] = (uint8_t)bk;
  }
}
hipError_t launch_pool_argmax(const float* x, uint8_t* k, int B, int H, int W, int C, int SH, int SW, int PH, int PW,
                              hipStream_t s, int KW) {
  const int OH = (H + 2 * PH - 2) / SH + 1, OW = (W + 2 * PW - KW) / SW + 1;
  const size_t total = (size_t)B * OH * OW * C;
  hipLaunchKernelGGL(pool_argmax_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 1u << 20)), dim3(256), 0, s, x, k,
                     B, H, W, C, OH, OW, SH, SW, PH, PW, KW);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// LayerNorm with saved statistics / backward.  One wave per row, D = 128, 256 or 512.
// ---------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void ln_train_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                       const float* __restrict__ b, float* __restrict__ y,
                                                       float* __restrict__ mean, float* __restrict__ rstd, int rows,
                                                       float eps) {
  constexpr int PER = D / 64;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  float v[PER], s = 0.f;
#pragma unroll
  for (int k = 0; k < PER; ++k) { v[k] = x[(size_t)row * D + lane + 64 * k]; s += v[k]; }
  const float m = wsum(s) / D;
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < PER; ++k) q = fmaf(v[k] - m, v[k] - m, q);
  const float r = rsqrtf(wsum(q) / D + eps);
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int c = lane + 64 * k;
    y[(size_t)row * D + c] = (v[k] - m) * r * g[c] + b[c];
  }
  if (lane == 0) { mean[row] = m; rstd[row] = r; }
}
hipError_t launch_ln_train(const float* x, const float* g, const float* b, float* y, float* mean, float* rstd, int rows,
                           int D, float eps, hipStream_t s) {
  if (D == 128) hipLaunchKernelGGL(ln_train_kernel<128>, dim3((rows + 3) / 4), dim3(256), 0, s, x, g, b, y, mean, rstd, rows, eps);
  else if (D == 256) hipLaunchKernelGGL(ln_train_kernel<256>, dim3((rows + 3) / 4), dim3(256), 0, s, x, g, b, y, mean, rstd, rows, eps);
  else if (D == 512) hipLaunchKernelGGL(ln_train_kernel<512>, dim3((rows + 3) / 4), dim3(256), 0, s, x, g, b, y, mean, rstd, rows, eps);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}
// dx = rstd * (dyg - mean_c(dyg) - xhat * mean_c(dyg * xhat)) (+ add),  dyg = dy * gamma
template <int D>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const float* __restrict__ g, const float* __restrict__ add,
                                                     float* __restrict__ dx, int rows) {
  constexpr int PER = D / 64;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float m = mean[row], r = rstd[row];
  float dg[PER], xh[PER], a = 0.f, bsum = 0.f;
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int c = lane + 64 * k;
    dg[k] = dy[(size_t)row * D + c] * g[c];
    xh[k] = (x[(size_t)row * D + c] - m) * r;
    a += dg[k];
    bsum = fmaf(dg[k], xh[k], bsum);
  }
  a = wsum(a) / D;
  bsum = wsum(bsum) / D;
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int c = lane + 64 * k;
    float o = r * (dg[k] - a - xh[k] * bsum);
    if (add) o += add[(size_t)row * D + c];
    dx[(size_t)row * D + c] = o;
  }
}
hipError_t launch_ln_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* g,
                         const float* add, float* dx, int rows, int D, hipStream_t s) {
  if (D == 128) hipLaunchKernelGGL(ln_bwd_kernel<128>, dim3((rows + 3) / 4), dim3(256), 0, s, dy, x, mean, rstd, g, add, dx, rows);
  else if (D == 256) hipLaunchKernelGGL(ln_bwd_kernel<256>, dim3((rows + 3) / 4), dim3(256), 0, s, dy, x, mean, rstd, g, add, dx, rows);
  else if (D == 512) hipLaunchKernelGGL(ln_bwd_kernel<512>, dim3((rows + 3) / 4), dim3(256), 0, s, dy, x, mean, rstd, g, add, dx, rows);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Attention with saved probabilities.  q/k/v/o are addressed as  base + (b*L + i)*ld + head*HD ; one block per
// (batch, head); K and V of the head live in LDS.  mask: causal (key j <= query i) and/or key padding
// (keytok[b][j] == pad_id masked) -- nn.MultiheadAttention's additive -inf masks (tfm.py:74-91).
// probs [B][heads][Lq][Lk] is written for the backward pass.
// ---------------------------------------------------------------------------
// GKV (memories too long for a head's K and V to sit in LDS: crops beyond about 600 tokens, the shipped max_dimension [800, 800]
// gives 2526): the same loops read K / V rows from global memory (L2) -- slow, and the same sums in the same order.
template <int HD, bool GKV = false>
__global__ __launch_bounds__(1024) void attn_train_fwd_kernel(const AttnTrainP p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int b = blockIdx.x / p.heads, hh = blockIdx.x % p.heads;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, NT = blockDim.x, NW = NT >> 6;
  const float* Ks;  // key j, channel c at Ks[j * KLD + c]
  const float* Vs;
  float* Ps;        // [waves][Lk]
  int KLD, VLD;
  if (GKV) {
    Ks = p.k + (size_t)b * p.Lk * p.ldk + hh * HD; KLD = p.ldk;
    Vs = p.v + (size_t)b * p.Lk * p.ldv + hh * HD; VLD = p.ldv;
    Ps = sm;
  } else {
    float* ks = sm;                              // [Lk][HD+1]
    float* vs = ks + (size_t)p.Lk * (HD + 1);    // [Lk][HD]
    Ps = vs + (size_t)p.Lk * HD;
    for (int i = tid; i < p.Lk * HD; i += NT) {
      const int j = i / HD, c = i % HD;
      ks[j * (HD + 1) + c] = p.k[((size_t)b * p.Lk + j) * p.ldk + hh * HD + c];
      vs[j * HD + c] = p.v[((size_t)b * p.Lk + j) * p.ldv + hh * HD + c];
    }
    __syncthreads();
    Ks = ks; KLD = HD + 1;
    Vs = vs; VLD = HD;
  }
  const float scale = rsqrtf((float)HD);
  float* P = Ps + (size_t)wave * p.Lk;
  for (int i = wave; i < p.Lq; i += NW) {
    const float* qp = p.q + ((size_t)b * p.Lq + i) * p.ldq + hh * HD;
    float q[HD];
#pragma unroll
    for (int c = 0; c < HD; ++c) q[c] = qp[c];
    float mx = -INFINITY;
    for (int j = lane; j < p.Lk; j += 64) {
      bool ok = !(p.causal && j > i);
      if (ok && p.keytok) ok = p.keytok[(size_t)b * p.Lk + j] != p.pad_id;
      float a = -INFINITY;
      if (ok) {
        a = 0.f;
#pragma unroll
        for (int c = 0; c < HD; ++c) a = fmaf(q[c], Ks[(size_t)j * KLD + c], a);
        a *= scale;
      }
      P[j] = a;
      mx = fmaxf(mx, a);
    }
    mx = wmax(mx);
    float sum = 0.f;
    for (int j = lane; j < p.Lk; j += 64) {
      const float e = P[j] == -INFINITY ? 0.f : expf(P[j] - mx);
      P[j] = e;
      sum += e;
    }
    sum = wsum(sum);
    const float inv = 1.f / sum;
    float* prow = p.probs + (((size_t)b * p.heads + hh) * p.Lq + i) * p.Lk;
    const uint8_t* mrow = p.dropmask ? p.dropmask + (((size_t)b * p.heads + hh) * p.Lq + i) * p.Lk : nullptr;
    for (int j = lane; j < p.Lk; j += 64) {
      const float w = P[j] * inv;
      prow[j] = w;                                            // saved: the softmax itself
      P[j] = mrow ? w * (mrow[j] ? p.dropscale : 0.f) : w;    // used: after dropout (nn.MultiheadAttention dropout)
    }
    // o[c] = sum_j P[j] * V[j][c]: lane -> (c = lane % HD, key phase = lane / HD)
    constexpr int PH = 64 / HD;
    const int c = lane % HD, ph = lane / HD;
    float o = 0.f;
#pragma unroll 8
    for (int j = ph; j < p.Lk; j += PH) o = fmaf(P[j], Vs[(size_t)j * VLD + c], o);
    if (PH == 2) o += __shfl_xor(o, 32, 64);
    if (lane < HD) p.o[((size_t)b * p.Lq + i) * p.ldo + hh * HD + c] = o;
  }
}
static hipError_t attn_lds(const void* fn, size_t lds) {
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}
// One block per (batch, head) leaves a CU with a single block; sixteen waves in it (four per SIMD) hide the latency of the
// per-query loops that four could not (forward 382 -> ~130 us, backward 656 -> ~250 us at B = 32, 261 tokens).  Fewer when
// the per-wave LDS rows do not fit.
static int attn_waves(size_t base_bytes, int Lk) {
  for (int nw = 16; nw > 4; nw -= 4)
    if (base_bytes + (size_t)nw * Lk * 4 <= 160 * 1024) return nw;
  return 4;
}
// waves of the global-K/V forms: as many per-wave rows of Lk floats as fit (a multiple of four, at most sixteen)
static int attn_waves_gkv(int Lk) {
  int nw = (int)((size_t)160 * 1024 / ((size_t)Lk * 4)) & ~3;
  return nw > 16 ? 16 : nw;
}
hipError_t launch_attn_train_fwd(const AttnTrainP& p, hipStream_t s) {
  const size_t base = ((size_t)p.Lk * (p.hd + 1) + (size_t)p.Lk * p.hd) * 4;
  hipError_t e;
  if (base + (size_t)4 * p.Lk * 4 > 160 * 1024) {  // K and V of a head do not fit in LDS beside four score rows
    const int nwg = attn_waves_gkv(p.Lk);
    if (nwg < 4 || (p.hd != 32 && p.hd != 64)) return hipErrorInvalidValue;
    const size_t ldsg = (size_t)nwg * p.Lk * 4;
    if (p.hd == 32) {
      if ((e = attn_lds(reinterpret_cast<const void*>(attn_train_fwd_kernel<32, true>), ldsg)) != hipSuccess) return e;
      hipLaunchKernelGGL((attn_train_fwd_kernel<32, true>), dim3(p.B * p.heads), dim3(nwg * 64), ldsg, s, p);
    } else {
      if ((e = attn_lds(reinterpret_cast<const void*>(attn_train_fwd_kernel<64, true>), ldsg)) != hipSuccess) return e;
      hipLaunchKernelGGL((attn_train_fwd_kernel<64, true>), dim3(p.B * p.heads), dim3(nwg * 64), ldsg, s, p);
    }
    return hipGetLastError();
  }
  const int nw = attn_waves(base, p.Lk);
  const size_t lds = base + (size_t)nw * p.Lk * 4;
  if (p.hd == 32) {
    if ((e = attn_lds(reinterpret_cast<const void*>(attn_train_fwd_kernel<32>), lds)) != hipSuccess) return e;
    hipLaunchKernelGGL(attn_train_fwd_kernel<32>, dim3(p.B * p.heads), dim3(nw * 64), lds, s, p);
  } else if (p.hd == 64) {
    if ((e = attn_lds(reinterpret_cast<const void*>(attn_train_fwd_kernel<64>), lds)) != hipSuccess) return e;
    hipLaunchKernelGGL(attn_train_fwd_kernel<64>, dim3(p.B * p.heads), dim3(nw * 64), lds, s, p);
  } else {
    return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// Backward: dV = P^T dO;  dP = dO V^T;  dS = P * (dP - rowsum(dP * P));  dQ = scale * dS K;  dK = scale * dS^T Q.
// One block per (batch, head); dS overwrites the saved probabilities in place (phase 2), phases separated by
// block barriers.  p.o carries dO; p.dq / p.dk / p.dv use the q / k / v strides.
// (GKV: K / V rows from global memory, as in the forward kernel; any multiple of 64 threads)
template <int HD, bool GKV = false>
__global__ __launch_bounds__(1024) void attn_train_bwd_kernel(const AttnTrainP p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int b = blockIdx.x / p.heads, hh = blockIdx.x % p.heads;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, NT = blockDim.x, NW = NT >> 6;
  const float* Ks;  // key j, channel c at Ks[j * KLD + c]
  const float* Vs;
  float* Ds;        // [waves][Lk]: the dS row a wave is working on
  int KLD, VLD;
  if (GKV) {
    Ks = p.k + (size_t)b * p.Lk * p.ldk + hh * HD; KLD = p.ldk;
    Vs = p.v + (size_t)b * p.Lk * p.ldv + hh * HD; VLD = p.ldv;
    Ds = sm;
  } else {
    float* ks = sm;                            // [Lk][HD+1]
    float* vs = ks + (size_t)p.Lk * (HD + 1);  // [Lk][HD+1]
    Ds = vs + (size_t)p.Lk * (HD + 1);
    for (int i = tid; i < p.Lk * HD; i += NT) {
      const int j = i / HD, c = i % HD;
      ks[j * (HD + 1) + c] = p.k[((size_t)b * p.Lk + j) * p.ldk + hh * HD + c];
      vs[j * (HD + 1) + c] = p.v[((size_t)b * p.Lk + j) * p.ldv + hh * HD + c];
    }
    Ks = ks; KLD = HD + 1;
    Vs = vs; VLD = HD + 1;
  }
  float* probs = p.probs + ((size_t)b * p.heads + hh) * p.Lq * p.Lk;
  constexpr int PH = 64 / HD;
  const int c = lane % HD, ph = lane / HD;
  const uint8_t* dmask = p.dropmask ? p.dropmask + ((size_t)b * p.heads + hh) * p.Lq * p.Lk : nullptr;
  // phase 1: dV[j][c] = sum_i (P o D)[i][j] * dO[i][c]   (wave per key; D = dropout keep mask * scale)
  for (int j = wave; j < p.Lk; j += NW) {
    float a = 0.f;
#pragma unroll 8
    for (int i = ph; i < p.Lq; i += PH) {
      float w = probs[(size_t)i * p.Lk + j];
      if (dmask) w *= dmask[(size_t)i * p.Lk + j] ? p.dropscale : 0.f;
      a = fmaf(w, p.o[((size_t)b * p.Lq + i) * p.ldo + hh * HD + c], a);
    }
    if (PH == 2) a += __shfl_xor(a, 32, 64);
    if (lane < HD) p.dv[((size_t)b * p.Lk + j) * p.ldv + hh * HD + c] = a;
  }
  __syncthreads();
  const float scale = rsqrtf((float)HD);
  // phase 2: per query row: dS, dQ (wave per query)
  for (int i = wave; i < p.Lq; i += NW) {
    const float* dop = p.o + ((size_t)b * p.Lq + i) * p.ldo + hh * HD;
    float d_o[HD];
#pragma unroll
    for (int cc = 0; cc < HD; ++cc) d_o[cc] = dop[cc];
    float* prow = probs + (size_t)i * p.Lk;
    const uint8_t* mrow = dmask ? dmask + (size_t)i * p.Lk : nullptr;
    float dsum = 0.f;
    for (int j = lane; j < p.Lk; j += 64) {
      float dp = 0.f;
#pragma unroll
      for (int cc = 0; cc < HD; ++cc) dp = fmaf(d_o[cc], Vs[(size_t)j * VLD + cc], dp);
      if (mrow) dp *= mrow[j] ? p.dropscale : 0.f;
      dsum = fmaf(dp, prow[j], dsum);
    }
    dsum = wsum(dsum);
    for (int j = lane; j < p.Lk; j += 64) {
      float dp = 0.f;
#pragma unroll
      for (int cc = 0; cc < HD; ++cc) dp = fmaf(d_o[cc], Vs[(size_t)j * VLD + cc], dp);
      if (mrow) dp *= mrow[j] ? p.dropscale : 0.f;
      const float ds = prow[j] * (dp - dsum);
      prow[j] = ds;  // read again by phase 3 (after the block barrier)
      Ds[wave * p.Lk + j] = ds;
    }
    float a = 0.f;
#pragma unroll 8
    for (int j = ph; j < p.Lk; j += PH) a = fmaf(Ds[wave * p.Lk + j], Ks[(size_t)j * KLD + c], a);
    if (PH == 2) a += __shfl_xor(a, 32, 64);
    if (lane < HD) p.dq[((size_t)b * p.Lq + i) * p.ldq + hh * HD + c] = a * scale;
  }
  __syncthreads();
  // phase 3: dK[j][c] = scale * sum_i dS[i][j] * Q[i][c]   (wave per key)
  for (int j = wave; j < p.Lk; j += NW) {
    float a = 0.f;
#pragma unroll 8
    for (int i = ph; i < p.Lq; i += PH) a = fmaf(probs[(size_t)i * p.Lk + j], p.q[((size_t)b * p.Lq + i) * p.ldq + hh * HD + c], a);
    if (PH == 2) a += __shfl_xor(a, 32, 64);
    if (lane < HD) p.dk[((size_t)b * p.Lk + j) * p.ldk + hh * HD + c] = a * scale;
  }
}
// Same mathematics with coalesced accesses: dO and Q of the head are staged in LDS too, and the two column-wise
// reductions (dV over queries with P, dK over queries with dS) run with lane = key (consecutive lanes read consecutive
// probabilities of one query row) and wave = group of HD/4 channels.  Needs (2*Lk*(HD+1) + 2*Lq*HD + 4*Lk) floats of LDS.
template <int HD>
__global__ __launch_bounds__(1024) void attn_train_bwd_fast_kernel(const AttnTrainP p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* Ks = sm;                             // [Lk][HD+1]
  float* Vs = Ks + (size_t)p.Lk * (HD + 1);   // [Lk][HD+1]
  float* dOs = Vs + (size_t)p.Lk * (HD + 1);  // [Lq][HD]
  float* Qs = dOs + (size_t)p.Lq * HD;        // [Lq][HD]
  float* Ds = Qs + (size_t)p.Lq * HD;         // [waves][Lk]
  const int b = blockIdx.x / p.heads, hh = blockIdx.x % p.heads;
  // a multiple of four waves: wave & 3 = channel group of the column reductions, wave >> 2 = which 64-key blocks it takes
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, NT = blockDim.x, NW = NT >> 6;
  const int cgrp = wave & 3, kgrp = wave >> 2, nkg = NW >> 2;
  for (int i = tid; i < p.Lk * HD; i += NT) {
    const int j = i / HD, c = i % HD;
    Ks[j * (HD + 1) + c] = p.k[((size_t)b * p.Lk + j) * p.ldk + hh * HD + c];
    Vs[j * (HD + 1) + c] = p.v[((size_t)b * p.Lk + j) * p.ldv + hh * HD + c];
  }
  for (int i = tid; i < p.Lq * HD; i += NT) {
    const int r = i / HD, c = i % HD;
    dOs[i] = p.o[((size_t)b * p.Lq + r) * p.ldo + hh * HD + c];
    Qs[i] = p.q[((size_t)b * p.Lq + r) * p.ldq + hh * HD + c];
  }
  __syncthreads();
  float* probs = p.probs + ((size_t)b * p.heads + hh) * p.Lq * p.Lk;
  constexpr int CG = HD / 4;  // channels per wave
  // column reduction out[j][c] = alpha * sum_i M[i][j] * R[i][c]
  const uint8_t* dmask = p.dropmask ? p.dropmask + ((size_t)b * p.heads + hh) * p.Lq * p.Lk : nullptr;
  auto colred = [&](const float* R, float* out, int ld, float alpha, const uint8_t* dm) {
    for (int j0 = kgrp * 64; j0 < p.Lk; j0 += 64 * nkg) {
      const int j = j0 + lane;
      float acc[CG];
#pragma unroll
      for (int c = 0; c < CG; ++c) acc[c] = 0.f;
      if (j < p.Lk) {
#pragma unroll 8
        for (int i = 0; i < p.Lq; ++i) {
          float m = probs[(size_t)i * p.Lk + j];
          if (dm) m *= dm[(size_t)i * p.Lk + j] ? p.dropscale : 0.f;
          const float* r = R + i * HD + cgrp * CG;
#pragma unroll
          for (int c = 0; c < CG; ++c) acc[c] = fmaf(m, r[c], acc[c]);
        }
        float* o = out + ((size_t)b * p.Lk + j) * ld + hh * HD + cgrp * CG;
#pragma unroll
        for (int c = 0; c < CG; ++c) o[c] = acc[c] * alpha;
      }
    }
  };
  if (!(p.probe & 1)) colred(dOs, p.dv, p.ldv, 1.f, dmask);  // phase 1: dV = (P o D)^T dO
  __syncthreads();
  const float scale = rsqrtf((float)HD);
  constexpr int PH = 64 / HD;
  const int c = lane % HD, ph = lane / HD;
  for (int i = wave; i < p.Lq && !(p.probe & 2); i += NW) {  // phase 2: dS (in place) and dQ, one wave per query row
    const float* d_o = dOs + i * HD;
    float* prow = probs + (size_t)i * p.Lk;
    float dsum = 0.f;
    for (int j = lane; j < p.Lk; j += 64) {
      float dp = 0.f;
#pragma unroll
      for (int cc = 0; cc < HD; ++cc) dp = fmaf(d_o[cc], Vs[j * (HD + 1) + cc], dp);
      if (dmask) dp *= dmask[(size_t)i * p.Lk + j] ? p.dropscale : 0.f;
      Ds[wave * p.Lk + j] = dp;
      dsum = fmaf(dp, prow[j], dsum);
    }
    dsum = wsum(dsum);
    for (int j = lane; j < p.Lk; j += 64) {
      const float ds = prow[j] * (Ds[wave * p.Lk + j] - dsum);
      prow[j] = ds;
      Ds[wave * p.Lk + j] = ds;
    }
    float a = 0.f;
#pragma unroll 8
    for (int j = ph; j < p.Lk; j += PH) a = fmaf(Ds[wave * p.Lk + j], Ks[j * (HD + 1) + c], a);
    if (PH == 2) a += __shfl_xor(a, 32, 64);
    if (lane < HD) p.dq[((size_t)b * p.Lq + i) * p.ldq + hh * HD + c] = a * scale;
  }
  __syncthreads();
  if (!(p.probe & 4)) colred(Qs, p.dk, p.ldk, scale, nullptr);  // phase 3: dK = scale * dS^T Q
}

hipError_t launch_attn_train_bwd(const AttnTrainP& p_in, hipStream_t s) {
  AttnTrainP p = p_in;
  static const int probe = D2T_PROBE_ENV("D2T_ATTN_BWD_PROBE");  // timing probe: skip phases (bit mask)
  p.probe = probe;
  const size_t lds = ((size_t)2 * p.Lk * (p.hd + 1) + 4 * (size_t)p.Lk) * 4;
  const size_t base_fast = ((size_t)2 * p.Lk * (p.hd + 1) + (size_t)2 * p.Lq * p.hd) * 4;
  const int nw = attn_waves(base_fast, p.Lk);
  const size_t lds_fast = base_fast + (size_t)nw * p.Lk * 4;
  hipError_t e;
  if (lds_fast <= 160 * 1024 && (p.hd == 32 || p.hd == 64)) {
    if (p.hd == 32) {
      if ((e = attn_lds(reinterpret_cast<const void*>(attn_train_bwd_fast_kernel<32>), lds_fast)) != hipSuccess) return e;
      hipLaunchKernelGGL(attn_train_bwd_fast_kernel<32>, dim3(p.B * p.heads), dim3(nw * 64), lds_fast, s, p);
    } else {
      if ((e = attn_lds(reinterpret_cast<const void*>(attn_train_bwd_fast_kernel<64>), lds_fast)) != hipSuccess) return e;
      hipLaunchKernelGGL(attn_train_bwd_fast_kernel<64>, dim3(p.B * p.heads), dim3(nw * 64), lds_fast, s, p);
    }
    return hipGetLastError();
  }
  if (p.hd != 32 && p.hd != 64) return hipErrorInvalidValue;
  if (lds > 160 * 1024) {  // K and V of a head do not fit in LDS: rows from global memory
    const int nwg = attn_waves_gkv(p.Lk);
    if (nwg < 4) return hipErrorInvalidValue;
    const size_t ldsg = (size_t)nwg * p.Lk * 4;
    if (p.hd == 32) {
      if ((e = attn_lds(reinterpret_cast<const void*>(attn_train_bwd_kernel<32, true>), ldsg)) != hipSuccess) return e;
      hipLaunchKernelGGL((attn_train_bwd_kernel<32, true>), dim3(p.B * p.heads), dim3(nwg * 64), ldsg, s, p);
    } else {
      if ((e = attn_lds(reinterpret_cast<const void*>(attn_train_bwd_kernel<64, true>), ldsg)) != hipSuccess) return e;
      hipLaunchKernelGGL((attn_train_bwd_kernel<64, true>), dim3(p.B * p.heads), dim3(nwg * 64), ldsg, s, p);
    }
    return hipGetLastError();
  }
  if (p.hd == 32) {
    if ((e = attn_lds(reinterpret_cast<const void*>(attn_train_bwd_kernel<32>), lds)) != hipSuccess) return e;
    hipLaunchKernelGGL(attn_train_bwd_kernel<32>, dim3(p.B * p.heads), dim3(256), lds, s, p);
  } else {
    if ((e = attn_lds(reinterpret_cast<const void*>(attn_train_bwd_kernel<64>), lds)) != hipSuccess) return e;
    hipLaunchKernelGGL(attn_train_bwd_kernel<64>, dim3(p.B * p.heads), dim3(256), lds, s, p);
  }
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Embedding: x[r][:] = E[tok[r]][:] * scale + pe[r % L][:];  backward (deterministic, block per vocabulary row):
// dE[v][:] = scale * sum_{r: tok[r] == v} dx[r][:], zero for the padding row (nn.Embedding padding_idx).
// ---------------------------------------------------------------------------
__global__ void embed_train_kernel(const float* __restrict__ E, const float* __restrict__ pe, const int64_t* __restrict__ tok,
                                   float* __restrict__ x, int rows, int L, int D, float scale) {
  const int r = blockIdx.x;
  const int64_t t = tok[r];
  for (int c = threadIdx.x; c < D; c += blockDim.x) x[(size_t)r * D + c] = E[(size_t)t * D + c] * scale + pe[(size_t)(r % L) * D + c];
}
hipError_t launch_embed_train(const float* E, const float* pe, const int64_t* tok, float* x, int rows, int L, int D,
                              float scale, hipStream_t s) {
  hipLaunchKernelGGL(embed_train_kernel, dim3(rows), dim3(256), 0, s, E, pe, tok, x, rows, L, D, scale);
  return hipGetLastError();
}
// One block per vocabulary row v (deterministic: rows are added in ascending order).  The block first lists the rows whose
// token is v -- 256 rows per pass, positions by ballot prefix -- and then adds only those (a token occurs in a handful of
// the B*L rows; scanning every row for every column was 0.37 ms).  D <= 1024.
__global__ __launch_bounds__(256) void embed_bwd_kernel(const float* __restrict__ dx, const int64_t* __restrict__ tok,
                                                        float* __restrict__ dE, int rows, int D, float scale, int pad_id) {
  constexpr int CAP = 1024;
  __shared__ int list[CAP + 256];
  __shared__ int wcnt[4];
  const int v = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  int n = 0;  // rows listed and not yet added (block-uniform)
  auto flush = [&]() {
    __syncthreads();
    for (int k = 0; k < n; ++k) {
      const float* src = dx + (size_t)list[k] * D;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = tid + j * 256;
        if (c < D) acc[j] += src[c];
      }
    }
    __syncthreads();
    n = 0;
  };
  if (v != pad_id) {
    for (int base = 0; base < rows; base += 256) {
      const int r = base + tid;
      const bool hit = r < rows && tok[r] == v;
      const unsigned long long m = __ballot(hit);
      if (lane == 0) wcnt[wave] = __popcll(m);
      __syncthreads();
      int off = n;
      for (int w = 0; w < wave; ++w) off += wcnt[w];
      if (hit) list[off + __popcll(m & ((1ull << lane) - 1ull))] = r;
      const int add = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
      __syncthreads();
      n += add;
      if (n >= CAP) flush();
    }
    flush();
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = tid + j * 256;
    if (c < D) dE[(size_t)v * D + c] = acc[j] * scale;
  }
}
hipError_t launch_embed_bwd(const float* dx, const int64_t* tok, float* dE, int rows, int V, int D, float scale, int pad_id,
                            hipStream_t s) {
  if (D > 1024) return hipErrorInvalidValue;
  hipLaunchKernelGGL(embed_bwd_kernel, dim3(V), dim3(256), 0, s, dx, tok, dE, rows, D, scale, pad_id);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Dropout: keep masks from Philox4x32-10 (counter-based, so forward and backward see the same mask without storing
// random state; masks are materialised as bytes because several kernels consume them)
// ---------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32(unsigned long long counter, unsigned long long stream_id, unsigned k0, unsigned k1,
                                           unsigned (&out)[4]) {
  unsigned c0 = (unsigned)counter, c1 = (unsigned)(counter >> 32), c2 = (unsigned)stream_id, c3 = (unsigned)(stream_id >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__global__ void dropout_mask_kernel(uint8_t* __restrict__ mask, size_t n, unsigned thresh16, unsigned long long seed,
                                    unsigned long long stream_id) {
  const size_t groups = (n + 7) / 8;  // eight 16-bit draws per Philox call
  for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < groups; g += (size_t)gridDim.x * blockDim.x) {
    unsigned r[4];
    philox4x32(g, stream_id, (unsigned)seed, (unsigned)(seed >> 32), r);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const size_t i = g * 8 + k;
      if (i < n) mask[i] = ((r[k >> 1] >> (16 * (k & 1))) & 0xFFFFu) >= thresh16 ? 1 : 0;
    }
  }
}
hipError_t launch_dropout_mask(uint8_t* mask, size_t n, float p, unsigned long long seed, unsigned long long stream_id,
                               hipStream_t s) {
  if (!n) return hipSuccess;
  const unsigned thresh = (unsigned)(p * 65536.0f + 0.5f);  // drop when the 16-bit draw is below p * 2^16
  const size_t groups = (n + 7) / 8;
  hipLaunchKernelGGL(dropout_mask_kernel, dim3((unsigned)std::min<size_t>((groups + 255) / 256, 1u << 16)), dim3(256), 0, s, mask,
                     n, thresh, seed, stream_id);
  return hipGetLastError();
}
__global__ void apply_mask_kernel(const float* __restrict__ a, const uint8_t* __restrict__ m, float scale,
                                  float* __restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    out[i] = m[i] ? a[i] * scale : 0.f;
}
hipError_t launch_apply_mask(const float* a, const uint8_t* mask, float scale, float* out, size_t n, hipStream_t s) {
  if (!n) return hipSuccess;
  hipLaunchKernelGGL(apply_mask_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 1u << 20)), dim3(256), 0, s, a, mask,
                     scale, out, n);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Data movement
// ---------------------------------------------------------------------------
// dst[r][0..Cd) = src[r][0..Cs) zero-padded on the right (Cd >= Cs)
__global__ void pad_cols_kernel(const float* __restrict__ src, float* __restrict__ dst, size_t rows, int Cs, int Cd) {
  const size_t total = rows * Cd;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % Cd);
    const size_t r = i / Cd;
    dst[i] = c < Cs ? src[r * Cs + c] : 0.f;
  }
}
hipError_t launch_pad_cols(const float* src, float* dst, size_t rows, int Cs, int Cd, hipStream_t s) {
  const size_t total = rows * Cd;
  hipLaunchKernelGGL(pad_cols_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 1u << 20)), dim3(256), 0, s, src, dst,
                     rows, Cs, Cd);
  return hipGetLastError();
}
// dst[r][0..cols) = src[r][0..cols) between two row-major matrices of different leading dimensions
__global__ void copy2d_kernel(const float* __restrict__ src, int lds, float* __restrict__ dst, int ldd, size_t rows, int cols) {
  const size_t total = rows * cols;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % cols);
    const size_t r = i / cols;
    dst[r * ldd + c] = src[r * lds + c];
  }
}
hipError_t launch_copy2d(const float* src, int lds, float* dst, int ldd, size_t rows, int cols, hipStream_t s) {
  const size_t total = rows * cols;
  hipLaunchKernelGGL(copy2d_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 1u << 20)), dim3(256), 0, s, src, lds,
                     dst, ldd, rows, cols);
  return hipGetLastError();
}
// Zero-dilated, zero-padded copy of an NHWC map: dst[b][oh*SH + OFFH][ow*SW + OFFW][c] = src[b][oh][ow][c], rest 0.
// (input of the stride-1 convolution that evaluates the data gradient of a strided convolution)
__global__ void dilate_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int OH, int OW, int C, int DH,
                              int DW, int SH, int SW, int offh, int offw) {
  const size_t total = (size_t)B * DH * DW * C;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const int w = (int)((i / C) % DW), h = (int)((i / ((size_t)C * DW)) % DH), b = (int)(i / ((size_t)C * DW * DH));
    const int hh = h - offh, ww = w - offw;
    float v = 0.f;
    if (hh >= 0 && ww >= 0 && hh % SH == 0 && ww % SW == 0 && hh / SH < OH && ww / SW < OW)
      v = src[(((size_t)b * OH + hh / SH) * OW + ww / SW) * C + c];
    dst[i] = v;
  }
}
hipError_t launch_dilate(const float* src, float* dst, int B, int OH, int OW, int C, int DH, int DW, int SH, int SW, int offh,
                         int offw, hipStream_t s) {
  const size_t total = (size_t)B * DH * DW * C;
  hipLaunchKernelGGL(dilate_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 1u << 20)), dim3(256), 0, s, src, dst, B,
                     OH, OW, C, DH, DW, SH, SW, offh, offw);
  return hipGetLastError();
}
// Weights of the data-gradient convolution: out[ci][co][kh][kw] = w[co][ci][KH-1-kh][KW-1-kw]  (OIHW -> flipped IOHW)
__global__ void flip_oihw_kernel(const float* __restrict__ w, float* __restrict__ out, int Cout, int Cin, int KH, int KW) {
  const size_t total = (size_t)Cout * Cin * KH * KW;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int kw = (int)(i % KW), kh = (int)((i / KW) % KH), co = (int)((i / ((size_t)KW * KH)) % Cout);
    const int ci = (int)(i / ((size_t)KW * KH * Cout));
    out[i] = w[(((size_t)co * Cin + ci) * KH + (KH - 1 - kh)) * KW + (KW - 1 - kw)];
  }
}
hipError_t launch_flip_oihw(const float* w, float* out, int Cout, int Cin, int KH, int KW, hipStream_t s) {
  const size_t total = (size_t)Cout * Cin * KH * KW;
  hipLaunchKernelGGL(flip_oihw_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 4096)), dim3(256), 0, s, w, out, Cout,
                     Cin, KH, KW);
  return hipGetLastError();
}
// rows gather / scatter between a [B][n+skip] token layout and the compact [B][n] one:
// dst[b*n + i][:] = src[(b*(n+skip) + skip + i)][:]  (gather)  or the inverse with zero skip rows (scatter)
__global__ void token_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int n, int skip, int D,
                                  int scatter) {
  const size_t total = (size_t)B * (scatter ? n + skip : n) * D;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % D);
    const size_t row = i / D;
    if (!scatter) {
      const size_t b = row / n, t = row % n;
      dst[i] = src[((b * (n + skip)) + skip + t) * D + c];
    } else {
      const size_t b = row / (n + skip), t = row % (n + skip);
      dst[i] = t < (size_t)skip ? 0.f : src[(b * n + (t - skip)) * D + c];
    }
  }
}
hipError_t launch_token_rows(const float* src, float* dst, int B, int n, int skip, int D, int scatter, hipStream_t s) {
  const size_t total = (size_t)B * (scatter ? n + skip : n) * D;
  hipLaunchKernelGGL(token_rows_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 1u << 20)), dim3(256), 0, s, src,
                     dst, B, n, skip, D, scatter);
  return hipGetLastError();
}
// out[c] = sum_b x[(b*stride_rows + row) * D + c]   (cls-token gradient)
__global__ void sum_rows_strided_kernel(const float* __restrict__ x, float* __restrict__ out, int B, long long stride_rows,
                                        int row, int D) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= D) return;
  float a = 0.f;
  for (int b = 0; b < B; ++b) a += x[((size_t)b * stride_rows + row) * D + c];
  out[c] = a;
}
hipError_t launch_sum_rows_strided(const float* x, float* out, int B, long long stride_rows, int row, int D, hipStream_t s) {
  hipLaunchKernelGGL(sum_rows_strided_kernel, dim3((D + 127) / 128), dim3(128), 0, s, x, out, B, stride_rows, row, D);
  return hipGetLastError();
}

// Stem convolution (Cin = 1, 3x3, pad 1) in training mode: raw weights, no bias / BN (z = conv(x)); and its
// weight gradient dW[co][kh][kw] = sum_p dz[p][co] * x[p + tap]  (block per (co-group), deterministic).
__global__ __launch_bounds__(256) void stem_raw_kernel(const float* __restrict__ img, const float* __restrict__ w,
                                                       float* __restrict__ z, int B, int H, int W, int Cout) {
  // a thread owns four consecutive output channels of one pixel: nine image loads serve all four, the filter comes
  // from LDS as [tap][co] float4s, the store is 16 bytes
  __shared__ __attribute__((aligned(16))) float ws[9][64];
  for (int i = threadIdx.x; i < 9 * Cout; i += 256) ws[i % 9][i / 9] = w[i];  // w is [co][9]
  __syncthreads();
  const int C4 = Cout >> 2;
  const size_t total = (size_t)B * H * W * C4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int co = (int)(i % C4) * 4;
    const int x = (int)((i / C4) % W), y = (int)((i / ((size_t)C4 * W)) % H), b = (int)(i / ((size_t)C4 * W * H));
    float a[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int ih = y + kh - 1, iw = x + kw - 1;
        if ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W) {
          const float v = img[((size_t)b * H + ih) * W + iw];
          const float4 w4 = *reinterpret_cast<const float4*>(&ws[kh * 3 + kw][co]);
          a[0] = fmaf(v, w4.x, a[0]); a[1] = fmaf(v, w4.y, a[1]); a[2] = fmaf(v, w4.z, a[2]); a[3] = fmaf(v, w4.w, a[3]);
        }
      }
    *reinterpret_cast<float4*>(z + i * 4) = make_float4(a[0], a[1], a[2], a[3]);
  }
}
hipError_t launch_stem_raw(const float* img, const float* w, float* z, int B, int H, int W, int Cout, hipStream_t s) {
  if (Cout % 4 || Cout > 64) return hipErrorInvalidValue;
  const size_t total = (size_t)B * H * W * (Cout / 4);
  hipLaunchKernelGGL(stem_raw_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 1u << 16)), dim3(256), 0, s, img, w, z,
                     B, H, W, Cout);
  return hipGetLastError();
}
// part[chunk][tap][co]: chunked over pixels; reduced by launch_wgrad_reduce with M = Cout, N = 1 ... (taps = 9)
template <int CO>  // output channels: 32 (ResNet conv0_1) or 64 (VGG's first convolution)
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const float* __restrict__ img, const float* __restrict__ dz,
                                                         float* __restrict__ part, int B, int H, int W, int chunk) {
  // thread -> (co = tid % CO, pixel lane = tid / CO)
  constexpr int NPL = 256 / CO;
  __shared__ float red[NPL][9][CO];
  const int co = threadIdx.x % CO, pl = threadIdx.x / CO;
  const long long P = (long long)B * H * W;
  const long long r0 = (long long)blockIdx.x * chunk, r1 = r0 + chunk < P ? r0 + chunk : P;
  float acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (long long r = r0 + pl; r < r1; r += NPL) {
    const int x = (int)(r % W), y = (int)((r / W) % H);
    const long long b = r / ((long long)W * H);
    const float g = dz[(size_t)r * CO + co];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int ih = y + kh - 1, iw = x + kw - 1;
        if ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W)
          acc[kh * 3 + kw] = fmaf(g, img[((size_t)b * H + ih) * W + iw], acc[kh * 3 + kw]);
      }
  }
#pragma unroll
  for (int t = 0; t < 9; ++t) red[pl][t][co] = acc[t];
  __syncthreads();
  for (int i = threadIdx.x; i < 9 * CO; i += 256) {
    const int t = i / CO, c = i % CO;
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < NPL; ++k) v += red[k][t][c];
    part[((size_t)blockIdx.x * 9 + t) * CO + c] = v;  // [chunk][tap][co] == wgrad partial layout with M = Cout, N = 1
  }
}
hipError_t launch_stem_wgrad(const float* img, const float* dz, float* part, int B, int H, int W, int Cout, int chunk,
                             int nchunks, hipStream_t s) {
  if (Cout == 32) hipLaunchKernelGGL(stem_wgrad_kernel<32>, dim3(nchunks), dim3(256), 0, s, img, dz, part, B, H, W, chunk);
  else if (Cout == 64) hipLaunchKernelGGL(stem_wgrad_kernel<64>, dim3(nchunks), dim3(256), 0, s, img, dz, part, B, H, W, chunk);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// GlobalContext block in the training step (addon_module/visual_attention.py:147-165): x [B][HW][C]
// ---------------------------------------------------------------------------
// partial[b][z][c] = sum over the z-th pixel chunk of (w[b][p]) x[b][p][c];  grid (B, GC_CHUNKS), thread = 4 channels x
// pixel phase
__global__ __launch_bounds__(256) void gc_wpool_kernel(const float* __restrict__ x, const float* __restrict__ wts,
                                                       float* __restrict__ part, int HW, int C) {
  __shared__ float4 red[256];
  const int b = blockIdx.x, z = blockIdx.y, tid = threadIdx.x;
  const int q4 = C / 4;                  // float4 per pixel row
  const int cq = tid % q4, ph = tid / q4, nph = 256 / q4;  // C <= 512 -> q4 <= 128 -> at least two pixel phases
  const int per = (HW + GC_CHUNKS - 1) / GC_CHUNKS, p0 = z * per, p1 = min(HW, p0 + per);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (ph < nph)
    for (int p = p0 + ph; p < p1; p += nph) {
      const float w = wts ? wts[(size_t)b * HW + p] : 1.f;
      const float4 v = *reinterpret_cast<const float4*>(x + ((size_t)b * HW + p) * C + cq * 4);
      acc.x = fmaf(w, v.x, acc.x); acc.y = fmaf(w, v.y, acc.y); acc.z = fmaf(w, v.z, acc.z); acc.w = fmaf(w, v.w, acc.w);
    }
  red[tid] = acc;
  __syncthreads();
  if (tid < q4) {  // fixed order over the phases: deterministic
    float4 t = red[tid];
    for (int k = 1; k < nph; ++k) {
      const float4 u = red[k * q4 + tid];
      t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
    }
    *reinterpret_cast<float4*>(part + ((size_t)b * GC_CHUNKS + z) * C + tid * 4) = t;
  }
}
__global__ void gc_wpool_final_kernel(const float* __restrict__ part, float* __restrict__ out, int B, int C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * C) return;
  const int b = i / C, c = i % C;
  float t = 0.f;
  for (int z = 0; z < GC_CHUNKS; ++z) t += part[((size_t)b * GC_CHUNKS + z) * C + c];
  out[i] = t;
}
hipError_t launch_gc_wpool(const float* x, const float* wts, float* part, float* out, int B, int HW, int C, hipStream_t s) {
  if (C % 4 || C > 512 || 256 / (C / 4) < 1) return hipErrorInvalidValue;
  hipLaunchKernelGGL(gc_wpool_kernel, dim3(B, GC_CHUNKS), dim3(256), 0, s, x, wts, part, HW, C);
  hipLaunchKernelGGL(gc_wpool_final_kernel, dim3((B * C + 255) / 256), dim3(256), 0, s, part, out, B, C);
  return hipGetLastError();
}
__global__ void gc_bcast_add_kernel(const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ out, size_t n4,
                                    int HW, int C) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const size_t e = i * 4;
    const int c = (int)(e % C);
    const size_t b = e / ((size_t)HW * C);
    float4 v = reinterpret_cast<const float4*>(x)[i];
    const float4 a = *reinterpret_cast<const float4*>(y + b * C + c);
    v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
    reinterpret_cast<float4*>(out)[i] = v;
  }
}
hipError_t launch_gc_bcast_add(const float* x, const float* y, float* out, int B, int HW, int C, hipStream_t s) {
  const size_t n4 = (size_t)B * HW * C / 4;
  hipLaunchKernelGGL(gc_bcast_add_kernel, dim3(ew_grid(n4)), dim3(EW_THREADS), 0, s, x, y, out, n4, HW, C);
  return hipGetLastError();
}
// one block per image: softmax statistics of the logits, then per pixel a = softmax, da = dctx . x; then dl = a (da - s)
// (da is taken against the pooled vector, dctx . (x[p] - ctx): dl = a (da - sum a da) cancels dctx . ctx, which is three
// orders of magnitude larger than what is left when the positions of a map resemble each other)
__global__ __launch_bounds__(512) void gc_pool_bwd_weights_kernel(const float* __restrict__ x, const float* __restrict__ logits,
                                                                  const float* __restrict__ dctx, const float* __restrict__ ctx,
                                                                  float* __restrict__ a_out, float* __restrict__ dl,
                                                                  float* __restrict__ sum_dl, int HW, int C) {
  __shared__ float red[24], dc[512], cx[512];
  const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const float* l = logits + (size_t)b * HW;
  float* av = a_out + (size_t)b * HW;
  float* dv = dl + (size_t)b * HW;
  if (tid < C) { dc[tid] = dctx[(size_t)b * C + tid]; cx[tid] = ctx[(size_t)b * C + tid]; }
  float m = -INFINITY;
  for (int p = tid; p < HW; p += 512) m = fmaxf(m, l[p]);
  m = wmax(m);
  if (lane == 0) red[wave] = m;
  __syncthreads();
  m = red[0];
#pragma unroll
  for (int w = 1; w < 8; ++w) m = fmaxf(m, red[w]);
  float sum = 0.f;
  for (int p = tid; p < HW; p += 512) sum += expf(l[p] - m);
  sum = wsum(sum);
  if (lane == 0) red[8 + wave] = sum;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int w = 0; w < 8; ++w) tot += red[8 + w];
  const float inv = 1.f / tot;
  // da per pixel: one wave per pixel, lanes over channels
  float sacc = 0.f;
  for (int p = wave; p < HW; p += 8) {
    const float* xr = x + ((size_t)b * HW + p) * C;
    float d = 0.f;
    for (int c = lane; c < C; c += 64) d = fmaf(dc[c], xr[c] - cx[c], d);
    d = wsum(d);
    const float a = expf(l[p] - m) * inv;
    if (lane == 0) { av[p] = a; dv[p] = d; }
    sacc = fmaf(a, d, sacc);  // identical in every lane of the wave
  }
  if (lane == 0) red[16 + wave] = sacc;
  __syncthreads();
  float sdot = 0.f;
#pragma unroll
  for (int w = 0; w < 8; ++w) sdot += red[16 + w];
  __syncthreads();
  float sd = 0.f;
  for (int p = tid; p < HW; p += 512) {
    const float v = av[p] * (dv[p] - sdot);
    dv[p] = v;
    sd += v;
  }
  sd = wsum(sd);
  if (lane == 0) red[wave] = sd;
  __syncthreads();
  if (tid == 0) {
    float t = 0.f;
    for (int w = 0; w < 8; ++w) t += red[w];
    sum_dl[b] = t;
  }
}
hipError_t launch_gc_pool_bwd_weights(const float* x, const float* logits, const float* dctx, const float* ctx, float* a,
                                      float* dl, float* sum_dl, int B, int HW, int C, hipStream_t s) {
  if (C > 512) return hipErrorInvalidValue;
  hipLaunchKernelGGL(gc_pool_bwd_weights_kernel, dim3(B), dim3(512), 0, s, x, logits, dctx, ctx, a, dl, sum_dl, HW, C);
  return hipGetLastError();
}
__global__ void gc_pool_bwd_dx_kernel(const float* __restrict__ a, const float* __restrict__ dl, const float* __restrict__ dctx,
                                      const float* __restrict__ wg, float* __restrict__ dx, size_t n4, int HW, int C) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const size_t e = i * 4;
    const int c = (int)(e % C);
    const size_t row = e / C, b = row / HW;
    const float av = a[row], dv = dl[row];
    const float4 d = *reinterpret_cast<const float4*>(dctx + b * C + c);
    const float4 w = *reinterpret_cast<const float4*>(wg + c);
    reinterpret_cast<float4*>(dx)[i] = make_float4(fmaf(av, d.x, dv * w.x), fmaf(av, d.y, dv * w.y), fmaf(av, d.z, dv * w.z),
                                                   fmaf(av, d.w, dv * w.w));
  }
}
__global__ __launch_bounds__(64) void sum_small_kernel(const float* __restrict__ a, int n, float* __restrict__ dst) {
  float t = 0.f;
  for (int i = threadIdx.x; i < n; i += 64) t += a[i];
  t = wsum(t);
  if (threadIdx.x == 0) dst[0] = t;
}
hipError_t launch_sum_small(const float* a, int n, float* dst, hipStream_t s) {
  hipLaunchKernelGGL(sum_small_kernel, dim3(1), dim3(64), 0, s, a, n, dst);
  return hipGetLastError();
}
hipError_t launch_gc_pool_bwd_dx(const float* a, const float* dl, const float* dctx, const float* wg, float* dx, int B, int HW,
                                 int C, hipStream_t s) {
  const size_t n4 = (size_t)B * HW * C / 4;
  hipLaunchKernelGGL(gc_pool_bwd_dx_kernel, dim3(ew_grid(n4)), dim3(EW_THREADS), 0, s, a, dl, dctx, wg, dx, n4, HW, C);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// VGG + BidirectionalLSTM encoder in the training step
// ---------------------------------------------------------------------------
__global__ void bias_add_kernel(float* __restrict__ z, const float* __restrict__ bias, size_t n4, int C) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)((i * 4) % C);
    float4 v = reinterpret_cast<float4*>(z)[i];
    const float4 b = *reinterpret_cast<const float4*>(bias + c);
    v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
    reinterpret_cast<float4*>(z)[i] = v;
  }
}
hipError_t launch_bias_add(float* z, const float* bias, long long rows, int C, hipStream_t s) {
  if (C % 4) return hipErrorInvalidValue;
  const size_t n4 = (size_t)rows * C / 4;
  hipLaunchKernelGGL(bias_add_kernel, dim3(ew_grid(n4)), dim3(EW_THREADS), 0, s, z, bias, n4, C);
  return hipGetLastError();
}
__global__ void scale_kernel(const float* __restrict__ a, float* __restrict__ out, size_t n, float alpha) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = alpha * a[i];
}
hipError_t launch_scale(const float* a, float* out, size_t n, float alpha, hipStream_t s) {
  hipLaunchKernelGGL(scale_kernel, dim3(ew_grid((n + 3) / 4)), dim3(EW_THREADS), 0, s, a, out, n, alpha);
  return hipGetLastError();
}
__global__ void mean_h_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, size_t total, int H, int W, int C, float inv) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C), w = (int)((i / C) % W);
    const size_t b = i / ((size_t)C * W * H);
    dx[i] = dy[(b * W + w) * C + c] * inv;
  }
}
hipError_t launch_mean_h_bwd(const float* dy, float* dx, int B, int H, int W, int C, hipStream_t s) {
  const size_t total = (size_t)B * H * W * C;
  hipLaunchKernelGGL(mean_h_bwd_kernel, dim3(ew_grid((total + 3) / 4)), dim3(EW_THREADS), 0, s, dy, dx, total, H, W, C, 1.f / H);
  return hipGetLastError();
}

__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + expf(-x)); }
constexpr int BL_RB = 4;  // batch rows per block, as the inference kernel (recurrent.hip)

// grid (2 directions, ceil(B / 4)); 1024 threads = the 4H gate rows (H = 256); the same arithmetic as bilstm_kernel
__global__ __launch_bounds__(1024) void bilstm_train_fwd_kernel(const float* __restrict__ g, const float* __restrict__ whh_t,
                                                                float* __restrict__ out, float* __restrict__ sv_gates,
                                                                float* __restrict__ sv_c, int B, int T, int H) {
  __shared__ float h_s[BL_RB][256], c_s[BL_RB][256], gate_s[BL_RB][1024];
  const int dir = blockIdx.x, b0 = blockIdx.y * BL_RB, r = threadIdx.x;
  const int G4 = 4 * H;
  const float* Wt = whh_t + (size_t)dir * H * G4;
  for (int i = r; i < BL_RB * H; i += 1024) { (&h_s[0][0])[i] = 0.f; (&c_s[0][0])[i] = 0.f; }
  __syncthreads();
  for (int step = 0; step < T; ++step) {
    const int t = dir == 0 ? step : T - 1 - step;
    float acc[BL_RB];
#pragma unroll
    for (int b = 0; b < BL_RB; ++b) acc[b] = (b0 + b < B) ? g[((size_t)(b0 + b) * T + t) * (2 * G4) + dir * G4 + r] : 0.f;
#pragma unroll 8
    for (int k = 0; k < H; ++k) {
      const float w = Wt[(size_t)k * G4 + r];
#pragma unroll
      for (int b = 0; b < BL_RB; ++b) acc[b] = fmaf(h_s[b][k], w, acc[b]);
    }
#pragma unroll
    for (int b = 0; b < BL_RB; ++b) gate_s[b][r] = acc[b];
    __syncthreads();
    {
      const int b = r >> 8, j = r & 255;
      if (b0 + b < B) {
        const float ig = sigm(gate_s[b][j]), fg = sigm(gate_s[b][H + j]);
        const float gg = tanhf(gate_s[b][2 * H + j]), og = sigm(gate_s[b][3 * H + j]);
        const float cc = fg * c_s[b][j] + ig * gg;
        const float hh = og * tanhf(cc);
        c_s[b][j] = cc;
        h_s[b][j] = hh;
        const size_t row = (size_t)(b0 + b) * T + t;
        out[row * (2 * H) + dir * H + j] = hh;
        float* sg = sv_gates + row * (2 * G4) + dir * G4;
        sg[j] = ig; sg[H + j] = fg; sg[2 * H + j] = gg; sg[3 * H + j] = og;
        sv_c[row * (2 * H) + dir * H + j] = cc;
      }
    }
    __syncthreads();
  }
}
hipError_t launch_bilstm_train_fwd(const float* gates, const float* whh_t, float* out, float* sv_gates, float* sv_c, int B, int T,
                                   int H, hipStream_t s) {
  if (H != 256 || B < 1 || T < 1) return hipErrorInvalidValue;
  hipLaunchKernelGGL(bilstm_train_fwd_kernel, dim3(2, (B + BL_RB - 1) / BL_RB), dim3(1024), 0, s, gates, whh_t, out, sv_gates, sv_c,
                     B, T, H);
  return hipGetLastError();
}

// Backward through time, same grid.  Per step (reverse processing order): phase 1, thread = (row, hidden unit): the cell's
// gradients and the pre-activation gate gradients (kept in LDS and written out); phase 2, thread = (quarter of the 4H gate
// rows, hidden unit k): dh_prev[b][k] = sum_r dgate[b][r] W_hh[r][k] for the block's four rows, each weight read once.
__global__ __launch_bounds__(1024) void bilstm_train_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ sv_gates,
                                                                const float* __restrict__ sv_c, const float* __restrict__ whh_fwd,
                                                                const float* __restrict__ whh_rev, float* __restrict__ dgates,
                                                                int B, int T, int H) {
  __shared__ float dh_s[BL_RB][256], dc_s[BL_RB][256], dg_s[BL_RB][1024], part_s[4][BL_RB][256];
  const int dir = blockIdx.x, b0 = blockIdx.y * BL_RB, tid = threadIdx.x;
  const int G4 = 4 * H;
  const float* W = dir == 0 ? whh_fwd : whh_rev;  // [4H][H]
  for (int i = tid; i < BL_RB * H; i += 1024) { (&dh_s[0][0])[i] = 0.f; (&dc_s[0][0])[i] = 0.f; }
  __syncthreads();
  for (int step = T - 1; step >= 0; --step) {
    const int t = dir == 0 ? step : T - 1 - step;
    const int tp = dir == 0 ? t - 1 : t + 1;  // the step processed before t (none when step == 0)
    {
      const int b = tid >> 8, j = tid & 255;
      float di = 0.f, df = 0.f, dgg = 0.f, dop = 0.f;
      if (b0 + b < B) {
        const size_t row = (size_t)(b0 + b) * T + t;
        const float* sg = sv_gates + row * (2 * G4) + dir * G4;
        const float ig = sg[j], fg = sg[H + j], gg = sg[2 * H + j], og = sg[3 * H + j];
        const float cc = sv_c[row * (2 * H) + dir * H + j];
        const float cp = step > 0 ? sv_c[((size_t)(b0 + b) * T + tp) * (2 * H) + dir * H + j] : 0.f;
        const float dh = dout[row * (2 * H) + dir * H + j] + dh_s[b][j];
        const float tc = tanhf(cc);
        const float dc = dc_s[b][j] + dh * og * (1.f - tc * tc);
        di = dc * gg * ig * (1.f - ig);
        df = dc * cp * fg * (1.f - fg);
        dgg = dc * ig * (1.f - gg * gg);
        dop = dh * tc * og * (1.f - og);
        dc_s[b][j] = dc * fg;
        float* o = dgates + row * (2 * G4) + dir * G4;
        o[j] = di; o[H + j] = df; o[2 * H + j] = dgg; o[3 * H + j] = dop;
      }
      dg_s[b][j] = di; dg_s[b][H + j] = df; dg_s[b][2 * H + j] = dgg; dg_s[b][3 * H + j] = dop;
    }
    __syncthreads();
    {
      const int q = tid >> 8, k = tid & 255;
      float acc[BL_RB];
#pragma unroll
      for (int b = 0; b < BL_RB; ++b) acc[b] = 0.f;
      const int r0 = q * 256;
#pragma unroll 8
      for (int r = r0; r < r0 + 256; ++r) {
        const float w = W[(size_t)r * H + k];
#pragma unroll
        for (int b = 0; b < BL_RB; ++b) acc[b] = fmaf(dg_s[b][r], w, acc[b]);
      }
#pragma unroll
      for (int b = 0; b < BL_RB; ++b) part_s[q][b][k] = acc[b];
    }
    __syncthreads();
    {
      const int b = tid >> 8, k = tid & 255;
      dh_s[b][k] = (part_s[0][b][k] + part_s[1][b][k]) + (part_s[2][b][k] + part_s[3][b][k]);
    }
    __syncthreads();
  }
}
hipError_t launch_bilstm_train_bwd(const float* dout, const float* sv_gates, const float* sv_c, const float* whh_fwd,
                                   const float* whh_rev, float* dgates, int B, int T, int H, hipStream_t s) {
  if (H != 256 || B < 1 || T < 1) return hipErrorInvalidValue;
  hipLaunchKernelGGL(bilstm_train_bwd_kernel, dim3(2, (B + BL_RB - 1) / BL_RB), dim3(1024), 0, s, dout, sv_gates, sv_c, whh_fwd,
                     whh_rev, dgates, B, T, H);
  return hipGetLastError();
}
__global__ void bilstm_hprev_kernel(const float* __restrict__ out, float* __restrict__ hf, float* __restrict__ hr, int B, int T,
                                    int H) {
  const size_t total = (size_t)B * T * H;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int j = (int)(i % H), t = (int)((i / H) % T);
    const size_t b = i / ((size_t)H * T);
    hf[i] = t > 0 ? out[((b * T + t - 1) * 2) * H + j] : 0.f;
    hr[i] = t + 1 < T ? out[((b * T + t + 1) * 2 + 1) * H + j] : 0.f;
  }
}
hipError_t launch_bilstm_hprev(const float* out, float* hprev_fwd, float* hprev_rev, int B, int T, int H, hipStream_t s) {
  const size_t total = (size_t)B * T * H;
  hipLaunchKernelGGL(bilstm_hprev_kernel, dim3(ew_grid((total + 3) / 4)), dim3(EW_THREADS), 0, s, out, hprev_fwd, hprev_rev, B, T, H);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// many small device-to-device copies in one launch (d2t_train_gather)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void multi_copy_kernel(const CopyChunk* __restrict__ table) {
  const CopyChunk c = table[blockIdx.x];
  const bool vec = ((reinterpret_cast<uintptr_t>(c.src) | reinterpret_cast<uintptr_t>(c.dst)) & 15) == 0;
  long long i = threadIdx.x;
  if (vec) {
    const long long n4 = c.n >> 2;
    for (; i < n4; i += 256) reinterpret_cast<float4*>(c.dst)[i] = reinterpret_cast<const float4*>(c.src)[i];
    for (i = (n4 << 2) + threadIdx.x; i < c.n; i += 256) c.dst[i] = c.src[i];
  } else {
    for (; i < c.n; i += 256) c.dst[i] = c.src[i];
  }
}
hipError_t launch_multi_copy(const CopyChunk* table, int chunks, hipStream_t s) {
  if (chunks <= 0) return hipSuccess;
  hipLaunchKernelGGL(multi_copy_kernel, dim3(chunks), dim3(256), 0, s, table);
  return hipGetLastError();
}

}  // namespace d2t
