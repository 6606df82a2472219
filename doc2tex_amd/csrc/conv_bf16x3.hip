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

template <int BM, int BN>
__global__ __launch_bounds__(256, 2) void conv_bf16x3_kernel(const ConvP p) {
  constexpr int WTM = BM / 2, WTN = BN / 2;
  constexpr int MI = WTM / 32, NJ = WTN / 32;
  constexpr int AR = BM / 32;          // A rows per thread per K-step (one float4 each)
  constexpr int BC = BN * 4 / 256;     // B 16-byte chunks per thread per plane
  constexpr int PLANE_A = BM * XROW, PLANE_B = BN * XROW;
  constexpr int STAGE = 2 * PLANE_A + 2 * PLANE_B;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];

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
    const int m = m0 + lrow + 32 * i;
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
    const int q = tid + 256 * j, row = q >> 2, c = q & 3;
    const int n = n0 + row;
    b_ok[j] = n < p.Cout;
    b_off[j] = (size_t)(b_ok[j] ? n : 0) * p.K + c * 8;
  }

  float4 ra[AR];
  uint4 rbh[BC], rbl[BC];
  int kh = 0, kw = 0, c0 = 0;
  const int KT = p.K / XBK;

  auto fetch = [&](int kt) {
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      const int ih = a_ih0[i] + kh, iw = a_iw0[i] + kw;
      if ((unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W) {
        const float* src = p.in + (size_t)(a_pix[i] + ih * p.W + iw) * p.Cin + c0 + kq * 4;
        ra[i] = *reinterpret_cast<const float4*>(src);
      } else {
        ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
#pragma unroll
    for (int j = 0; j < BC; ++j) {
      if (b_ok[j]) {
        rbh[j] = *reinterpret_cast<const uint4*>(p.w_hi + b_off[j] + (size_t)kt * XBK);
        rbl[j] = *reinterpret_cast<const uint4*>(p.w_lo + b_off[j] + (size_t)kt * XBK);
      } else {
        rbh[j] = make_uint4(0, 0, 0, 0);
        rbl[j] = make_uint4(0, 0, 0, 0);
      }
    }
    c0 += XBK;
    if (c0 == p.Cin) {
      c0 = 0;
      if (++kw == p.KW) { kw = 0; ++kh; }
    }
  };
  auto stage = [&](int buf) {
    unsigned char* ah = smem + buf * STAGE;
    unsigned char* al = ah + PLANE_A;
    unsigned char* bh = al + PLANE_A;
    unsigned char* bl = bh + PLANE_B;
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      const int row = lrow + 32 * i;
      const int off = row * XROW + swz_chunk(row, kq >> 1) * 16 + (kq & 1) * 8;
      uint2 hi, lo;
      split4(ra[i], hi, lo);
      *reinterpret_cast<uint2*>(ah + off) = hi;
      *reinterpret_cast<uint2*>(al + off) = lo;
    }
#pragma unroll
    for (int j = 0; j < BC; ++j) {
      const int q = tid + 256 * j, row = q >> 2, c = q & 3;
      const int off = row * XROW + swz_chunk(row, c) * 16;
      *reinterpret_cast<uint4*>(bh + off) = rbh[j];
      *reinterpret_cast<uint4*>(bl + off) = rbl[j];
    }
  };

  const int wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, h = lane >> 5;

  f32x16 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  fetch(0);
  stage(0);
  __syncthreads();

  for (int kt = 0; kt < KT; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < KT) fetch(kt + 1);
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
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fal[i], fbh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fah[i], fbl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fah[i], fbh[j], acc[i][j], 0, 0, 0);
        }
    }
    if (kt + 1 < KT) stage(cur ^ 1);
    __syncthreads();
  }
  conv_epilogue<MI, NJ>(p, acc, m0 + wm * WTM, n0 + wn * WTN, r, h);
}

hipError_t launch_conv_bf16x3(const ConvP& p, hipStream_t s) {
  if (p.M <= 0 || p.Cout <= 0) return hipSuccess;
  if (!p.w_hi || !p.w_lo || p.Cin % XBK != 0 || p.K != p.KH * p.KW * p.Cin) return hipErrorInvalidValue;
  const int mt = (p.M + 127) / 128;
  if (p.Cout <= 64) {
    hipLaunchKernelGGL((conv_bf16x3_kernel<128, 64>), dim3(mt * ((p.Cout + 63) / 64)), dim3(256), 0, s, p);
  } else {
    hipLaunchKernelGGL((conv_bf16x3_kernel<128, 128>), dim3(mt * ((p.Cout + 127) / 128)), dim3(256), 0, s, p);
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
hipError_t launch_split_bf16(const float* w, uint16_t* hi, uint16_t* lo, size_t n, hipStream_t s) {
  hipLaunchKernelGGL(split_bf16_kernel, dim3((unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096)), dim3(256), 0,
                     s, w, hi, lo, n);
  return hipGetLastError();
}

}  // namespace d2t
