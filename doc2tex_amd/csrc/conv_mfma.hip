// Implicit-GEMM convolution / linear kernel on the gfx950 fp32 matrix cores.
//
// Replaces the reference's nn.Conv2d + BatchNorm2d(eval) + ReLU (+ residual)
// sequences (feature_extractor/resnet.py:205-245, :32-48), the HybridEmbed patch
// projection (seq_modeling/addon_module/patchembed.py:135) and the large-M
// nn.Linear layers of the ViT blocks (seq_modeling/vit/vision_transformer.py:26-32,61-81).
//
// GEMM view: M = B*OH*OW output pixels, N = Cout, K = KH*KW*Cin.  Activations are
// NHWC so that for one filter tap the K-slice of a pixel is Cin contiguous floats;
// weights are OHWI ([Cout][K], BN folded), so a K-step of both operands is one
// 128-byte run per row.  v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD):
// lane l supplies A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31].  The k order
// inside a 8-wide chunk is permuted identically for A and B (lane half h takes
// k = 4h..4h+3) so each lane fetches its four k values with one ds_read_b128.
//
// Block = 256 threads = 2x2 waves; block tile BM x BN x 32, register-prefetched
// double-buffered LDS (rows padded to 36 floats: conflict-free b128 reads and
// writes), one barrier per K-step.  Epilogue fuses bias, residual, ReLU / exact
// GELU, positional-table add and the output row / head-split remaps.
#include "conv_common.h"

namespace d2t {

constexpr int BK = 32;
constexpr int LDS_LD = 36;  // padded row length in floats (144 B = 9 x 16 B)

template <int BM, int BN>
__global__ __launch_bounds__(256, 2) void conv_mfma_kernel(const ConvP p) {
  constexpr int WTM = BM / 2, WTN = BN / 2;  // wave tile
  constexpr int MI = WTM / 32, NJ = WTN / 32;
  constexpr int AR = BM / 32, BR = BN / 32;  // rows each thread stages per K-step
  __shared__ __attribute__((aligned(16))) float smem[2 * (BM + BN) * LDS_LD];
  float* const As = smem;                     // [2][BM][LDS_LD]
  float* const Bs = smem + 2 * BM * LDS_LD;   // [2][BN][LDS_LD]

  // XCD-aware tile order: consecutive logical tiles (same A rows, neighbouring
  // pixels) run on the same XCD so they share its L2 (bijective remap).
  const int nt = (p.Cout + BN - 1) / BN;
  const int logical = xcd_logical_tile();
  const int m0 = (logical / nt) * BM;
  const int n0 = (logical % nt) * BN;

  const int tid = threadIdx.x;
  const int kq = tid & 7;     // float4 slot inside the 32-float K-step
  const int lrow = tid >> 3;  // 0..31

  // per-thread A rows: output pixel -> top-left input coordinate
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
      a_ih0[i] = -0x40000000;  // never in range
      a_iw0[i] = 0;
      a_pix[i] = 0;
    }
  }
  const float* b_ptr[BR];
  bool b_ok[BR];
#pragma unroll
  for (int i = 0; i < BR; ++i) {
    const int n = n0 + lrow + 32 * i;
    b_ok[i] = n < p.Cout;
    b_ptr[i] = p.w + (size_t)(b_ok[i] ? n : 0) * p.K + kq * 4;
  }

  float4 ra[AR], rb[BR];
  int kh = 0, kw = 0, c0 = 0;  // position of the NEXT K-step to fetch
  const int KT = p.K / BK;

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
    for (int i = 0; i < BR; ++i) {
      rb[i] = b_ok[i] ? *reinterpret_cast<const float4*>(b_ptr[i] + (size_t)kt * BK)
                      : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // next K-step: taps cycle fastest, then the 32-channel chunk -> the nine taps of a chunk re-read the
    // same 128-byte channel slices of neighbouring pixels back to back (L2 / L1 hits instead of HBM)
    if (++kw == p.KW) {
      kw = 0;
      if (++kh == p.KH) { kh = 0; c0 += 32; }
    }
  };
  auto stage = [&](int buf) {
    float* a = As + buf * BM * LDS_LD;
    float* b = Bs + buf * BN * LDS_LD;
#pragma unroll
    for (int i = 0; i < AR; ++i) *reinterpret_cast<float4*>(a + (lrow + 32 * i) * LDS_LD + kq * 4) = ra[i];
#pragma unroll
    for (int i = 0; i < BR; ++i) *reinterpret_cast<float4*>(b + (lrow + 32 * i) * LDS_LD + kq * 4) = rb[i];
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
    if (kt + 1 < KT) fetch(kt + 1);  // global loads in flight under the MFMAs
    const float* a = As + cur * BM * LDS_LD + (wm * WTM + r) * LDS_LD + h * 4;
    const float* b = Bs + cur * BN * LDS_LD + (wn * WTN + r) * LDS_LD + h * 4;
#pragma unroll
    for (int kc = 0; kc < BK / 8; ++kc) {
      float4 fa[MI], fb[NJ];
#pragma unroll
      for (int i = 0; i < MI; ++i) fa[i] = *reinterpret_cast<const float4*>(a + i * 32 * LDS_LD + kc * 8);
#pragma unroll
      for (int j = 0; j < NJ; ++j) fb[j] = *reinterpret_cast<const float4*>(b + j * 32 * LDS_LD + kc * 8);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].x, fb[j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].y, fb[j].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].z, fb[j].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].w, fb[j].w, acc[i][j], 0, 0, 0);
        }
    }
    if (kt + 1 < KT) stage(cur ^ 1);
    __syncthreads();
  }

  static_assert((size_t)BM * BN <= 2 * (BM + BN) * LDS_LD, "the wide epilogue's fp32 tile must fit in the staging area");
  if (wide_epilogue_full_ok(p))
    conv_epilogue_wide_full<BM, BN, 256, MI, NJ>(p, acc, reinterpret_cast<unsigned char*>(smem), m0, n0, wm * WTM, wn * WTN, r, h, tid);
  else
    conv_epilogue<MI, NJ>(p, acc, m0 + wm * WTM, n0 + wn * WTN, r, h);
}

template <int BM, int BN>
static hipError_t launch_cfg(const ConvP& p, hipStream_t s) {
  const int mt = (p.M + BM - 1) / BM, nt = (p.Cout + BN - 1) / BN;
  hipLaunchKernelGGL((conv_mfma_kernel<BM, BN>), dim3(mt * nt), dim3(256), 0, s, p);
  return hipGetLastError();
}

hipError_t launch_conv(const ConvP& p, hipStream_t s) {
  if (p.M <= 0 || p.Cout <= 0) return hipSuccess;
  if (p.in_hi) return launch_conv_bf16x3(p, s);  // split-bf16 input planes: only that kernel reads them
  // bf16 weight planes select the split-bf16 kernel whatever the problem size, so that a row's result never depends on
  // how many other rows (batch size, shard) share the launch
  if (p.w_hi && p.w_lo && p.Cout >= 64) return launch_conv_bf16x3(p, s);
  if (p.Cin % BK != 0 || p.K != p.KH * p.KW * p.Cin) return hipErrorInvalidValue;
  const long long tiles128 = (long long)((p.M + 127) / 128) * ((p.Cout + 127) / 128);
  if (p.Cout <= 64) {
    if ((long long)((p.M + 127) / 128) >= 256) return launch_cfg<128, 64>(p, s);
    return launch_cfg<64, 64>(p, s);
  }
  if (tiles128 >= 256) return launch_cfg<128, 128>(p, s);
  return launch_cfg<64, 64>(p, s);
}

}  // namespace d2t
