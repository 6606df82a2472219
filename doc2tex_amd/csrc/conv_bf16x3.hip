// Implicit-GEMM convolution on the bf16 matrix cores with fp32-class accuracy ("bf16x3").
//
// Every fp32 operand is split as x = hi + lo with hi = the upper 16 bits of x (a bf16) and
// lo = bf16(x - hi); the product is accumulated in fp32 as  a_hi*b_hi + a_hi*b_lo + a_lo*b_hi
// (the dropped a_lo*b_lo term is ~2^-18 relative).  Three v_mfma_f32_32x32x16_bf16 replace eight
// v_mfma_f32_32x32x2_f32 for the same K=16 slice: 5.3x the fp32-MFMA rate at ~2^-17 relative
// product error.  Measured end to end on the headline model (CPU emulation in tools/, GPU parity
// tests): encoder memory error 1e-4, logits error 2.4e-5 (budget 1e-3), greedy tokens unchanged.
//
// Same GEMM view, tile shape (128x128x32, 2x2 waves, 64x64 wave tile), loader geometry and
// epilogue as conv_mfma.hip.  Weights arrive pre-split (two bf16 planes, OHWI); activations stay
// fp32 in HBM and are split in registers while staging.  LDS holds four bf16 planes per stage
// ([rows][32] = 64-byte rows); the 16-byte k-chunk index is XOR-swizzled with (row>>2)&3 so that
// the 16-lane groups of every ds_read_b128 touch all 64 banks exactly once.
// MFMA operand map (32x32x16 bf16): lane l holds A[row l&31][k = 8*(l>>5) + j], j = 0..7, and the
// same for B with its column: one 16-byte chunk per lane per k16-step.
#include <cstdlib>
#include <type_traits>

#include "conv_common.h"

namespace d2t {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int XBK = 32;             // K-step
constexpr int XROW = 64;            // bytes per LDS row of one plane (32 bf16)

__device__ __forceinline__ int swz_chunk(int row, int c) { return c ^ ((row >> 2) & 3); }

// four consecutive-k floats -> packed hi (truncation) and lo (RNE of the remainder) bf16 quads
__device__ __forceinline__ void split4(const float4 v, uint2& hi, uint2& lo) {
  const unsigned x0 = __float_as_uint(v.x), x1 = __float_as_uint(v.y), x2 = __float_as_uint(v.z),
                 x3 = __float_as_uint(v.w);
  hi.x = __builtin_amdgcn_perm(x1, x0, 0x07060302u);  // [x0.hi16 | x1.hi16 << 16]
  hi.y = __builtin_amdgcn_perm(x3, x2, 0x07060302u);
  const f32x2 r01 = {v.x - __uint_as_float(x0 & 0xFFFF0000u), v.y - __uint_as_float(x1 & 0xFFFF0000u)};
  const f32x2 r23 = {v.z - __uint_as_float(x2 & 0xFFFF0000u), v.w - __uint_as_float(x3 & 0xFFFF0000u)};
  const bf16x2 l01 = __builtin_convertvector(r01, bf16x2), l23 = __builtin_convertvector(r23, bf16x2);
  lo.x = *reinterpret_cast<const unsigned*>(&l01);
  lo.y = *reinterpret_cast<const unsigned*>(&l23);
}

template <int BM, int BN, int WM, int WN, bool DEEP>  // block tile, wave grid (WM x WN waves), pipelining depth
__global__ __launch_bounds__(WM * WN * 64, (WM * WN) / 2) void conv_bf16x3_kernel(const ConvP p) {
  constexpr int NT = WM * WN * 64;
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int MI = WTM / 32, NJ = WTN / 32;
  constexpr int RPP = NT / 8;          // A rows staged per pass (8 threads x float4 per row)
  constexpr int AR = BM / RPP;         // A rows per thread per K-step (one float4 each)
  constexpr int BC = BN * 4 / NT;      // B 16-byte chunks per thread per plane
  static_assert(AR >= 1 && BC >= 1 && MI >= 1 && NJ >= 1, "tile too small for the wave grid");
  constexpr int PLANE_A = BM * XROW, PLANE_B = BN * XROW;
  constexpr int STAGE = 2 * PLANE_A + 2 * PLANE_B;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];
  static_assert(BM * BN * 4 <= 2 * STAGE, "the wide epilogue's fp32 tile must fit in the staging area");

  const int nt = (p.Cout + BN - 1) / BN;
  const int logical = xcd_logical_tile();
  const int m0 = (logical / nt) * BM;
  const int n0 = (logical % nt) * BN;
  const int tid = threadIdx.x;
  const int kq = tid & 7, lrow = tid >> 3;

  int a_ih0[AR], a_iw0[AR], a_pix[AR];
  const int ohow = p.OH * p.OW;
#pragma unroll
  for (int i = 0; i < AR; ++i) {
    const int m = m0 + lrow + RPP * i;
    if (m < p.M) {
      const int b = m / ohow, rem = m - b * ohow;
      const int oh = rem / p.OW, ow = rem - oh * p.OW;
      a_ih0[i] = oh * p.SH - p.PH;
      a_iw0[i] = ow * p.SW - p.PW;
      a_pix[i] = b * p.H * p.W;
    } else {
      a_ih0[i] = -0x40000000;
      a_iw0[i] = 0;
      a_pix[i] = 0;
    }
  }
  // B chunks of this thread: chunk id q = tid + 256*j -> row q>>2, k-chunk q&3
  size_t b_off[BC];
  bool b_ok[BC];
#pragma unroll
  for (int j = 0; j < BC; ++j) {
    const int q = tid + NT * j, row = q >> 2, c = q & 3;
    const int n = n0 + row;
    b_ok[j] = n < p.Cout;
    b_off[j] = (size_t)(b_ok[j] ? n : 0) * p.K + c * 8;
  }

  // Two register sets for the staged K-steps: loads for K-step t+2 are issued at the top of
  // iteration t, while the set fetched one iteration earlier (K-step t+1) is converted and written to
  // LDS in small pieces BETWEEN the MFMAs of step t, so VALU / LDS-store issue hides under the matrix pipe.
  float4 ra[2][AR];
  uint4 rbh[2][BC], rbl[2][BC];
  int kh = 0, kw = 0, c0 = 0;
  const int KT = p.K / XBK;

  auto fetch = [&](int kt, float4 (&xa)[AR], uint4 (&xh)[BC], uint4 (&xl)[BC]) {
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      // branch-free: out-of-image taps read a valid address (the tensor base) and are zeroed by a select
      const int ih = a_ih0[i] + kh, iw = a_iw0[i] + kw;
      const bool ok = (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
      const size_t off = ok ? (size_t)(a_pix[i] + ih * p.W + iw) * p.Cin + c0 + kq * 4 : (size_t)(kq * 4);
      const float4 v = *reinterpret_cast<const float4*>(p.in + off);
      const unsigned msk = ok ? 0xFFFFFFFFu : 0u;  // bit-mask instead of a select keeps the load unconditional
      xa[i] = make_float4(__uint_as_float(__float_as_uint(v.x) & msk), __uint_as_float(__float_as_uint(v.y) & msk),
                          __uint_as_float(__float_as_uint(v.z) & msk), __uint_as_float(__float_as_uint(v.w) & msk));
    }
#pragma unroll
    for (int j = 0; j < BC; ++j) {
      const uint4 vh = *reinterpret_cast<const uint4*>(p.w_hi + b_off[j] + (size_t)kt * XBK);
      const uint4 vl = *reinterpret_cast<const uint4*>(p.w_lo + b_off[j] + (size_t)kt * XBK);
      const unsigned msk = b_ok[j] ? 0xFFFFFFFFu : 0u;
      xh[j] = make_uint4(vh.x & msk, vh.y & msk, vh.z & msk, vh.w & msk);
      xl[j] = make_uint4(vl.x & msk, vl.y & msk, vl.z & msk, vl.w & msk);
    }
    // next K-step: taps cycle fastest, then the 32-channel chunk -> the nine taps of a chunk re-read the
    // same 128-byte channel slices of neighbouring pixels back to back (L2 / L1 hits instead of HBM)
    if (++kw == p.KW) {
      kw = 0;
      if (++kh == p.KH) { kh = 0; c0 += 32; }
    }
  };
  // piece q of the staging work of one K-step: q < AR -> A row group q, else B chunk q - AR
  auto stage_piece = [&](int buf, int q, const float4 (&xa)[AR], const uint4 (&xh)[BC], const uint4 (&xl)[BC]) {
    unsigned char* ah = smem + buf * STAGE;
    unsigned char* al = ah + PLANE_A;
    unsigned char* bh = al + PLANE_A;
    unsigned char* bl = bh + PLANE_B;
    if (q < AR) {
      const int row = lrow + RPP * q;
      const int off = row * XROW + swz_chunk(row, kq >> 1) * 16 + (kq & 1) * 8;
      uint2 hi, lo;
      split4(xa[q], hi, lo);
      *reinterpret_cast<uint2*>(ah + off) = hi;
      *reinterpret_cast<uint2*>(al + off) = lo;
    } else if (q - AR < BC) {
      const int j = q - AR;
      const int qq = tid + NT * j, row = qq >> 2, c = qq & 3;
      const int off = row * XROW + swz_chunk(row, c) * 16;
      *reinterpret_cast<uint4*>(bh + off) = xh[j];
      *reinterpret_cast<uint4*>(bl + off) = xl[j];
    }
  };

  const int wave = tid >> 6, lane = tid & 63;
  const int wm = wave / WN, wn = wave % WN;
  const int r = lane & 31, h = lane >> 5;

  f32x16 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // prologue: K-step 0 -> LDS[0]; K-step 1 in flight in set 1
  fetch(0, ra[0], rbh[0], rbl[0]);
#pragma unroll
  for (int q = 0; q < AR + BC; ++q) stage_piece(0, q, ra[0], rbh[0], rbl[0]);
  if (KT > 1) fetch(1, ra[1], rbh[1], rbl[1]);
  __syncthreads();

  constexpr int NPIECE = AR + BC;
  // K-step kt from LDS[kt&1]; if STAGE_NEXT the set (sa,sh,sl) = K-step kt+1 is converted and written
  // to the other LDS buffer one piece per MFMA triple, and (FETCH) K-step kt+2 is loaded into (na,nh,nl).
  auto compute = [&](int kt, auto stage_next, auto fetch_next, const float4 (&sa)[AR], const uint4 (&sh)[BC],
                     const uint4 (&sl)[BC], float4 (&na)[AR], uint4 (&nh)[BC], uint4 (&nl)[BC]) {
    constexpr bool STAGE_NEXT = decltype(stage_next)::value, FETCH = decltype(fetch_next)::value;
    const int cur = kt & 1;
    const unsigned char* ah = smem + cur * STAGE;
    const unsigned char* al = ah + PLANE_A;
    const unsigned char* bh = al + PLANE_A;
    const unsigned char* bl = bh + PLANE_B;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int c = 2 * kk + h;
      bf16x8 fah[MI], fal[MI], fbh[NJ], fbl[NJ];
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int row = wm * WTM + i * 32 + r;
        const int off = row * XROW + swz_chunk(row, c) * 16;
        fah[i] = *reinterpret_cast<const bf16x8*>(ah + off);
        fal[i] = *reinterpret_cast<const bf16x8*>(al + off);
      }
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int row = wn * WTN + j * 32 + r;
        const int off = row * XROW + swz_chunk(row, c) * 16;
        fbh[j] = *reinterpret_cast<const bf16x8*>(bh + off);
        fbl[j] = *reinterpret_cast<const bf16x8*>(bl + off);
      }
      if (FETCH && kk == 0) fetch(kt + 2, na, nh, nl);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fal[i], fbh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fah[i], fbl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fah[i], fbh[j], acc[i][j], 0, 0, 0);
          const int piece = (kk * MI + i) * NJ + j;  // compile-time after unrolling
          if (STAGE_NEXT && DEEP && piece < NPIECE) stage_piece(cur ^ 1, piece, sa, sh, sl);
        }
    }
    if (STAGE_NEXT) {  // leftovers (or, without DEEP, the whole staging: the loads were issued this iteration)
#pragma unroll
      for (int piece = DEEP ? 2 * MI * NJ : 0; piece < NPIECE; ++piece) stage_piece(cur ^ 1, piece, sa, sh, sl);
    }
    __syncthreads();
  };
  using T_ = std::integral_constant<bool, true>;
  using F_ = std::integral_constant<bool, false>;

  if constexpr (!DEEP) {
    // occupancy-driven variant (4 waves per SIMD): one register set, K-step t+1 fetched at the top of
    // iteration t and staged after its MFMAs; other waves cover the staging time
    for (int kt2 = 0; kt2 < KT; ++kt2) {
      if (kt2 + 1 < KT) {
        if (kt2 > 0) fetch(kt2 + 1, ra[1], rbh[1], rbl[1]);  // (K-step 1 was fetched in the prologue)
        compute(kt2, T_{}, F_{}, ra[1], rbh[1], rbl[1], ra[0], rbh[0], rbl[0]);
      } else {
        compute(kt2, F_{}, F_{}, ra[1], rbh[1], rbl[1], ra[0], rbh[0], rbl[0]);
      }
    }
    if (wide_epilogue_full_ok(p)) conv_epilogue_wide_full<BM, BN, NT, MI, NJ>(p, acc, smem, m0, n0, wm * WTM, wn * WTN, r, h, tid);
    else conv_epilogue<MI, NJ>(p, acc, m0 + wm * WTM, n0 + wn * WTN, r, h);
    return;
  }
  // steady state (K-steps kt+1 and kt+2 exist), unrolled by two so the register sets are static
  int kt = 0;
  for (; kt + 3 < KT; kt += 2) {
    compute(kt, T_{}, T_{}, ra[1], rbh[1], rbl[1], ra[0], rbh[0], rbl[0]);
    compute(kt + 1, T_{}, T_{}, ra[0], rbh[0], rbl[0], ra[1], rbh[1], rbl[1]);
  }
  // tail: at most three K-steps left; set parity continues from the loop (kt is even)
  if (kt + 2 < KT) {        // three left: kt, kt+1, kt+2
    compute(kt, T_{}, T_{}, ra[1], rbh[1], rbl[1], ra[0], rbh[0], rbl[0]);
    compute(kt + 1, T_{}, F_{}, ra[0], rbh[0], rbl[0], ra[1], rbh[1], rbl[1]);
    compute(kt + 2, F_{}, F_{}, ra[1], rbh[1], rbl[1], ra[0], rbh[0], rbl[0]);
  } else if (kt + 1 < KT) { // two left
    compute(kt, T_{}, F_{}, ra[1], rbh[1], rbl[1], ra[0], rbh[0], rbl[0]);
    compute(kt + 1, F_{}, F_{}, ra[0], rbh[0], rbl[0], ra[1], rbh[1], rbl[1]);
  } else if (kt < KT) {     // one left
    compute(kt, F_{}, F_{}, ra[1], rbh[1], rbl[1], ra[0], rbh[0], rbl[0]);
  }
  if (wide_epilogue_full_ok(p)) conv_epilogue_wide_full<BM, BN, NT, MI, NJ>(p, acc, smem, m0, n0, wm * WTM, wn * WTN, r, h, tid);
  else conv_epilogue<MI, NJ>(p, acc, m0 + wm * WTM, n0 + wn * WTN, r, h);
}

// ---------------------------------------------------------------------------
// Split-activation kernel with LDS-DMA staging.  The input arrives as split-bf16 records written by the
// producing layer's epilogue (conv_common.h plane_idx), so staging A is a pure 16-byte copy like B: no
// conversion and almost no address arithmetic in the K loop (the on-the-fly variant above spends ~5 VALU
// instructions per MFMA on it).  Per output row a 16-bit validity mask over the filter taps and a signed
// element offset are computed once.  Both operands are pure copies, so every 16-byte chunk
// goes global -> LDS directly (global_load_lds_dwordx4: no staging registers, no ds_write).  The LDS
// destination of a wave instruction is linear (base + lane*16 = 16 rows x 64 B of one plane), so the
// XOR swizzle of the k-chunk is applied to the per-lane SOURCE address and again on the ds_read side
// (cdna guide rule 21).  Out-of-image taps / rows beyond Cout point the source at a zero page.
// ---------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void* lds_ptr;

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// F16 (ConvP::f16, fp16x2 mode): the A records hold fp16 in their hi half (no lo plane is staged or read), the weight planes
// are fp16 hi / lo, a product is x * w_lo + x * w_hi on v_mfma_f32_32x32x16_f16
template <int BM, int BN, int WM, bool F16 = false>  // wave grid WM x 2
__device__ __forceinline__ void conv_bf16x3g_body(const ConvP& p, unsigned char* smem) {
  constexpr int NW = WM * 2;
  constexpr int WTM = BM / WM, WTN = BN / 2;
  constexpr int MI = WTM / 32, NJ = WTN / 32;
  constexpr int AJ = BM / (16 * NW);  // 16-row (1 KiB) pieces per wave per A plane
  constexpr int BJ = BN / (16 * NW);
  static_assert(AJ >= 1 && BJ >= 1, "tile too small for the wave count");
  constexpr int PLANE_A = BM * XROW, PLANE_B = BN * XROW;
  constexpr int STAGE = 2 * PLANE_A + 2 * PLANE_B;

  const int nt = (p.Cout + BN - 1) / BN;
  const int ntiles = nt * ((p.M - p.m_base + BM - 1) / BM);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int lr = lane >> 2, pos = lane & 3;
  // Persistent over tiles: the grid may be smaller than the tile count (p.max_blocks), which leaves
  // block slots free for the latency-bound decode kernels of the previous batch (pipelined mode).
  // Tile order: in every round the blocks that share an XCD (blockIdx % 8) take consecutive tiles.
  const int G = gridDim.x, gx = G >> 3;
  for (int tile0 = 0; tile0 < ntiles; tile0 += G) {
  int logical = tile0 + blockIdx.x;
  if ((G & 7) == 0) logical = tile0 + (blockIdx.x & 7) * gx + (blockIdx.x >> 3);
  if (logical >= ntiles) break;  // block-uniform
  const int m0 = p.m_base + (logical / nt) * BM;
  const int n0 = (logical % nt) * BN;

  int a_off[AJ];
  unsigned a_mask[AJ];
  const int ohow = p.OH * p.OW;
#pragma unroll
  for (int j = 0; j < AJ; ++j) {
    const int row = (wave * AJ + j) * 16 + lr;
    const int c = swz_chunk(row, pos);  // the chunk that belongs at LDS position `pos` of this row
    const int m = m0 + row;
    a_off[j] = 0;
    a_mask[j] = 0;
    if (m < p.M) {
      int b, oh, ow;
      conv_row_coords(p, m, b, oh, ow);
      const int ih0 = oh * p.SH - p.PH, iw0 = ow * p.SW - p.PW;
      a_off[j] = ((b * p.H + ih0) * p.W + iw0) * p.Cin * 2 + c * 8;
      for (int kh = 0; kh < p.KH; ++kh)
        for (int kw = 0; kw < p.KW; ++kw)
          if ((unsigned)(ih0 + kh) < (unsigned)p.H && (unsigned)(iw0 + kw) < (unsigned)p.W)
            a_mask[j] |= 1u << (kh * p.KW + kw);
    }
  }
  const uint16_t* b_hi[BJ];
  const uint16_t* b_lo[BJ];
#pragma unroll
  for (int j = 0; j < BJ; ++j) {
    const int row = (wave * BJ + j) * 16 + lr;
    const int c = swz_chunk(row, pos);
    const int n = n0 + row;
    const bool ok = n < p.Cout;
    b_hi[j] = ok ? p.w_hi + (size_t)n * p.K + c * 8 : nullptr;
    b_lo[j] = ok ? p.w_lo + (size_t)n * p.K + c * 8 : nullptr;
  }
  const uint16_t* zero = reinterpret_cast<const uint16_t*>(p.zero16);
  int kh = 0, kw = 0, c0 = 0;
  const int KT = p.K / XBK;

  auto issue = [&](int kt, int buf) {
    unsigned char* ah = smem + buf * STAGE;
    unsigned char* al = ah + PLANE_A;
    unsigned char* bh = al + PLANE_A;
    unsigned char* bl = bh + PLANE_B;
    const int tap = kh * p.KW + kw;
    const int tapoff = ((kh * p.W + kw) * p.Cin + c0) * 2;
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
      const bool ok = (a_mask[j] >> tap) & 1u;
      const uint16_t* src = ok ? p.in_hi + (a_off[j] + tapoff) : zero;
      const int piece = (wave * AJ + j) * 1024;
      __builtin_amdgcn_global_load_lds(src, (lds_ptr)(ah + piece), 16, 0, 0);
      if (!F16) __builtin_amdgcn_global_load_lds(ok ? src + 32 : zero, (lds_ptr)(al + piece), 16, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
      const int piece = (wave * BJ + j) * 1024;
      __builtin_amdgcn_global_load_lds(b_hi[j] ? b_hi[j] + (size_t)kt * XBK : zero, (lds_ptr)(bh + piece), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(b_lo[j] ? b_lo[j] + (size_t)kt * XBK : zero, (lds_ptr)(bl + piece), 16, 0, 0);
    }
    if (++kw == p.KW) {
      kw = 0;
      if (++kh == p.KH) { kh = 0; c0 += 32; }
    }
  };

  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, h = lane >> 5;
  f32x16 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  issue(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int kt = 0; kt < KT; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < KT) issue(kt + 1, cur ^ 1);  // DMA for the next K-step flies under this step's MFMAs
    __builtin_amdgcn_sched_barrier(0);         // keep the DMA issue ahead of the ds_reads / MFMAs
    const unsigned char* ah = smem + cur * STAGE;
    const unsigned char* al = ah + PLANE_A;
    const unsigned char* bh = al + PLANE_A;
    const unsigned char* bl = bh + PLANE_B;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int c = 2 * kk + h;
      bf16x8 fah[MI], fal[MI], fbh[NJ], fbl[NJ];
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int row = wm * WTM + i * 32 + r;
        const int off = row * XROW + swz_chunk(row, c) * 16;
        fah[i] = *reinterpret_cast<const bf16x8*>(ah + off);
        if (!F16) fal[i] = *reinterpret_cast<const bf16x8*>(al + off);
      }
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int row = wn * WTN + j * 32 + r;
        const int off = row * XROW + swz_chunk(row, c) * 16;
        fbh[j] = *reinterpret_cast<const bf16x8*>(bh + off);
        fbl[j] = *reinterpret_cast<const bf16x8*>(bl + off);
      }
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          if (F16) {
            const f16x8 xa = __builtin_bit_cast(f16x8, fah[i]);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xa, __builtin_bit_cast(f16x8, fbl[j]), acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xa, __builtin_bit_cast(f16x8, fbh[j]), acc[i][j], 0, 0, 0);
            continue;
          }
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fal[i], fbh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fah[i], fbl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fah[i], fbh[j], acc[i][j], 0, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's DMA pieces have landed
    __syncthreads();                                   // ... and everyone else's; reads of `cur` are done
  }
  if (wide_epilogue_ok(p)) {  // block-uniform
    static_assert(BM * BN * 4 <= 2 * STAGE, "the fp32 tile must fit in the staging area");
    if (p.pool2) conv_epilogue_wide_pool<BM, BN, NW * 64, MI, NJ>(p, acc, smem, m0, n0, wm * WTM, wn * WTN, r, h, tid);
    else conv_epilogue_wide<BM, BN, NW * 64, MI, NJ>(p, acc, smem, m0, n0, wm * WTM, wn * WTN, r, h, tid);
  } else {
    conv_epilogue<MI, NJ>(p, acc, m0 + wm * WTM, n0 + wn * WTN, r, h);
  }
  }  // tile loop
}

// non-template entry points (the host-side stub of a __global__ template using the LDS-DMA builtin is not emitted)
__global__ __launch_bounds__(256, 2) void conv_bf16x3g_128x128(const ConvP p) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * (2 * 128 * XROW + 2 * 128 * XROW)];
  conv_bf16x3g_body<128, 128, 2>(p, smem);
}
__global__ __launch_bounds__(512, 4) void conv_bf16x3g_128x128_w8(const ConvP p) {  // 8 waves, wave tile 32x64
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * (2 * 128 * XROW + 2 * 128 * XROW)];
  conv_bf16x3g_body<128, 128, 4>(p, smem);
}
// same code under its own symbol for the dominant GEMM shape (512 -> 512 channels, 3x3: K = 4608), so that
// rocprofv3's per-kernel rows separate it from the other layers
__global__ __launch_bounds__(512, 4) void conv_bf16x3g_128x128_w8_k4608(const ConvP p) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * (2 * 128 * XROW + 2 * 128 * XROW)];
  conv_bf16x3g_body<128, 128, 4>(p, smem);
}
__global__ __launch_bounds__(256, 3) void conv_bf16x3g_128x64(const ConvP p) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * (2 * 128 * XROW + 2 * 64 * XROW)];
  conv_bf16x3g_body<128, 64, 2>(p, smem);
}

// fp16x2 builds (ConvP::f16)
__global__ __launch_bounds__(256, 3) void conv_f16x2g_128x64(const ConvP p) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * (2 * 128 * XROW + 2 * 64 * XROW)];
  conv_bf16x3g_body<128, 64, 2, true>(p, smem);
}
__global__ __launch_bounds__(512, 4) void conv_f16x2g_128x128_w8(const ConvP p) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * (2 * 128 * XROW + 2 * 128 * XROW)];
  conv_bf16x3g_body<128, 128, 4, true>(p, smem);
}

// The 128-row LDS-DMA kernels over output rows [p.m_base, p.M) with a column tile of `bn` (64 | 128): the pipelined kernel
// hands the rows of its last, partial round of tiles to this one (conv_bf16x3p.hip).
hipError_t launch_conv_bf16x3g_rows(const ConvP& p, int bn, hipStream_t s) {
  if (p.M <= p.m_base) return hipSuccess;
  if (!p.in_hi || !p.w_hi || !p.w_lo || !p.zero16 || p.m_base < 0 || (bn != 64 && bn != 128)) return hipErrorInvalidValue;
  const int tiles = ((p.M - p.m_base + 127) / 128) * ((p.Cout + bn - 1) / bn);
  if (bn == 64) hipLaunchKernelGGL(conv_bf16x3g_128x64, dim3(tiles), dim3(256), 0, s, p);
  else hipLaunchKernelGGL(conv_bf16x3g_128x128_w8, dim3(tiles), dim3(512), 0, s, p);
  return hipGetLastError();
}

hipError_t launch_conv_bf16x3(const ConvP& p, hipStream_t s) {
  if (p.M <= 0 || p.Cout <= 0) return hipSuccess;
  if (p.pool2) {  // fused 2x2 max-pool: split-record LDS-DMA kernels with the wide epilogue only (g body, 16x16x32 pipelined body)
    const bool wide = p.store_mode == STORE_ROWS && p.rows_per_img == 0 && !p.row_add && (p.Cout & 31) == 0;
    if (!p.in_hi || !wide || p.res || p.res_hi || (p.M & 3) || p.M != 4 * p.B * (p.OH / 2) * (p.OW / 2) ||
        (p.Cout >= 128 && p.pipelined != 3 && p.pipelined != 6 && p.pipelined != 7) || (p.Cout > 64 && p.Cout < 128))
      return hipErrorInvalidValue;
  }
  if (!p.w_hi || !p.w_lo || p.Cin % XBK != 0 || p.K != p.KH * p.KW * p.Cin + p.Cin2 || p.m_base != 0) return hipErrorInvalidValue;
  // a second (1x1) input along K: the DMA issuer of the pipelined 16x16x32 kernel only (no tail hand-over, not the patch forms)
  if (p.Cin2 && (!p.in_hi || !p.in2_hi || p.Cin2 % XBK || p.Cout < 128 || (p.pipelined != 3 && p.pipelined != 6 && p.pipelined != 7) || p.pool2 ||
                 p.OH != p.H || p.OW != p.W))
    return hipErrorInvalidValue;
  if (p.f16 && (!p.in_hi || (p.Cout >= 128 && p.pipelined != 3))) return hipErrorInvalidValue;  // fp16 records: split-record kernels only
  const int mt = (p.M + 127) / 128;
  if (p.in_hi) {  // split-bf16 input planes
    if (!p.zero16 || p.KH * p.KW > 16 || (long long)p.B * p.H * p.W * p.Cin * 2 > 0x7fffffffLL) return hipErrorInvalidValue;
    // layers with >= 128 output channels: the pipelined 256x128 kernel (one block per CU, three LDS stages); dispatch by
    // layer shape only (never by the row count), so a sample's results do not depend on its batch
    if (p.pipelined && p.Cout >= 128) return launch_conv_bf16x3p(p, s);
    int tiles = mt * ((p.Cout + (p.Cout <= 64 ? 63 : 127)) / (p.Cout <= 64 ? 64 : 128));
    const int grid = (p.max_blocks > 0 && tiles > p.max_blocks) ? p.max_blocks : tiles;
    if (p.Cout <= 64) {
      // 48 KB of LDS and 4 waves per block: three blocks fit on a CU (launch bounds 256 x 3), which this short-K layer
      // (conv0_2: 9 K-steps per tile) uses to overlap one tile's prologue / epilogue with another's MFMAs whenever the
      // grid is not capped (no slots reserved for the decode stream): 1.35 -> 0.96 ms alone.  With a cap it stays at
      // the cap (three per CU measured 0.5 % slower end to end in pipelined serving).
      const int grid3 = grid;
      if (p.f16) hipLaunchKernelGGL(conv_f16x2g_128x64, dim3(grid3), dim3(256), 0, s, p);
      else hipLaunchKernelGGL(conv_bf16x3g_128x64, dim3(grid3), dim3(256), 0, s, p);
    } else {
      // 8 waves per tile (4 per SIMD at two blocks per CU) measured 1-2 % faster end to end than 4 waves
      static const bool w4 = D2T_PROBE_ENV("D2T_BF16X3_WAVES") == 4;
      if (p.f16) hipLaunchKernelGGL(conv_f16x2g_128x128_w8, dim3(grid), dim3(512), 0, s, p);
      else if (w4) hipLaunchKernelGGL(conv_bf16x3g_128x128, dim3(grid), dim3(256), 0, s, p);
      else if (p.K == 4608 && p.Cout == 512) hipLaunchKernelGGL(conv_bf16x3g_128x128_w8_k4608, dim3(grid), dim3(512), 0, s, p);
      else hipLaunchKernelGGL(conv_bf16x3g_128x128_w8, dim3(grid), dim3(512), 0, s, p);
    }
    return hipGetLastError();
  }
  // fp32 rows in (the ViT encoder's linears, M = B * 261: one to two rounds of 128 x 128 tiles): eight waves per tile measured
  // 15-20 % faster there than four (round 4, tools/probe/conv_per_launch.py: 229 -> 194 us per ViT block); same products in the
  // same order per output element, so the two are bit-identical
  static const int variant = D2T_PROBE_ENV_STR("D2T_BF16X3_WAVES") ? D2T_PROBE_ENV("D2T_BF16X3_WAVES") : 8;
  if (p.Cout <= 64) {
    hipLaunchKernelGGL((conv_bf16x3_kernel<128, 64, 2, 2, true>), dim3(mt * ((p.Cout + 63) / 64)), dim3(256), 0, s, p);
  } else if (variant == 4) {
    hipLaunchKernelGGL((conv_bf16x3_kernel<128, 128, 2, 2, true>), dim3(mt * ((p.Cout + 127) / 128)), dim3(256), 0, s, p);
  } else {
    // 8 waves per 128x128 tile (wave tile 32x64): 4 waves per SIMD at 2 blocks/CU keep the matrix pipe fed
    hipLaunchKernelGGL((conv_bf16x3_kernel<128, 128, 4, 2, false>), dim3(mt * ((p.Cout + 127) / 128)), dim3(512), 0, s, p);
  }
  return hipGetLastError();
}

__global__ void split_bf16_kernel(const float* __restrict__ w, uint16_t* __restrict__ hi, uint16_t* __restrict__ lo,
                                  size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float x = w[i];
    const __bf16 h = (__bf16)x;  // v_cvt_pk_bf16_f32: round to nearest even
    const __bf16 l = (__bf16)(x - (float)h);
    hi[i] = *reinterpret_cast<const uint16_t*>(&h);
    lo[i] = *reinterpret_cast<const uint16_t*>(&l);
  }
}
__global__ void split_f16_kernel(const float* __restrict__ w, uint16_t* __restrict__ hi, uint16_t* __restrict__ lo, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float x = w[i];
    const uint16_t h = f32_to_f16_bits(x);
    hi[i] = h;
    lo[i] = f32_to_f16_bits(x - f16_bits_to_f32(h));
  }
}
__global__ void split_act_kernel(const float* __restrict__ x, uint16_t* __restrict__ planes, size_t rows, int C, int f16) {
  const size_t n = rows * C;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint16_t hi, lo;
    split_rec(x[i], hi, lo, f16);
    const size_t o = plane_idx(i / C, (int)(i % C), C);
    planes[o] = hi;
    planes[o + 32] = lo;
  }
}
__global__ void merge_act_kernel(const uint16_t* __restrict__ planes, float* __restrict__ x, size_t rows, int C, int f16) {
  const size_t n = rows * C;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const size_t o = plane_idx(i / C, (int)(i % C), C);
    x[i] = join_rec(planes[o], planes[o + 32], f16);
  }
}
// the same, eight channels per thread: two 16-byte loads, one 16-byte store into each half of the record (C % 32 == 0)
__global__ void split_act8_kernel(const float* __restrict__ x, uint16_t* __restrict__ planes, size_t n8, int f16) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
    const float4 a = reinterpret_cast<const float4*>(x)[2 * i], b = reinterpret_cast<const float4*>(x)[2 * i + 1];
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    uint16_t hi[8], lo[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) split_rec(v[k], hi[k], lo[k], f16);
    // element e = 8 i of the row-major [rows][C] tensor lies in record e / 32 at position e % 32 (C % 32 == 0)
    const size_t rec = i >> 2, pos = (i & 3) * 8;
    uint16_t* dst = planes + rec * 64 + pos;
    *reinterpret_cast<uint4*>(dst) = make_uint4(hi[0] | (uint32_t)hi[1] << 16, hi[2] | (uint32_t)hi[3] << 16,
                                                hi[4] | (uint32_t)hi[5] << 16, hi[6] | (uint32_t)hi[7] << 16);
    *reinterpret_cast<uint4*>(dst + 32) = make_uint4(lo[0] | (uint32_t)lo[1] << 16, lo[2] | (uint32_t)lo[3] << 16,
                                                     lo[4] | (uint32_t)lo[5] << 16, lo[6] | (uint32_t)lo[7] << 16);
  }
}
hipError_t launch_split_act(const float* x, uint16_t* planes, size_t rows, int C, hipStream_t s, int f16) {
  const size_t n = rows * C;
  if (C % 32 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {
    const size_t n8 = n / 8;
    hipLaunchKernelGGL(split_act8_kernel, dim3((unsigned)((n8 + 255) / 256 < 8192 ? (n8 + 255) / 256 : 8192)), dim3(256), 0, s, x,
                       planes, n8, f16);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(split_act_kernel, dim3((unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096)), dim3(256), 0,
                     s, x, planes, rows, C, f16);
  return hipGetLastError();
}
hipError_t launch_merge_act(const uint16_t* planes, float* x, size_t rows, int C, hipStream_t s, int f16) {
  const size_t n = rows * C;
  hipLaunchKernelGGL(merge_act_kernel, dim3((unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096)), dim3(256), 0,
                     s, planes, x, rows, C, f16);
  return hipGetLastError();
}

hipError_t launch_split_f16(const float* w, uint16_t* hi, uint16_t* lo, size_t n, hipStream_t s) {
  hipLaunchKernelGGL(split_f16_kernel, dim3((unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096)), dim3(256), 0,
                     s, w, hi, lo, n);
  return hipGetLastError();
}
hipError_t launch_split_bf16(const float* w, uint16_t* hi, uint16_t* lo, size_t n, hipStream_t s) {
  hipLaunchKernelGGL(split_bf16_kernel, dim3((unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096)), dim3(256), 0,
                     s, w, hi, lo, n);
  return hipGetLastError();
}

}  // namespace d2t
