// Winograd F(2x2, 3x3) form of the split-bf16 3x3 / stride 1 / pad 1 convolution (round 3; the sixteen 512 -> 512 layers
// of the backbone @16x129: 21 of the encoder's 36 ms).
//
// Direct form: every output pixel costs 9 * Cin products per output channel.  Winograd's minimal filtering computes a 2x2
// block of outputs from a 4x4 input window with 16 products instead of 36 (Lavin & Gray 2016):
//     Y = A^T [ (G g G^T) .* (B^T d B) ] A          g 3x3 filter, d 4x4 input tile, .* element-wise, summed over Cin
// i.e. 16 independent GEMMs  M_c[tile][cout] = sum_cin V_c[tile][cin] * U_c[cout][cin]  (c = the 16 tile components) over
// T = B * ceil(H/2) * ceil(W/2) tiles, followed by the 4x4 -> 2x2 output transform: 2.25 x fewer matrix products.
// Numerics with split-bf16 operands (V and U split into hi + lo bf16 like the activations and weights of the direct
// kernels, three MFMAs per product, fp32 accumulate): tools/winograd_study.py emulates the whole network on the
// reference's fixtures -- greedy tokens exact everywhere, max |dlogit| 0.7 ... 1.07 x the direct form's (docs/winograd_study_r03.txt).
//
// Three kernels:
//   wino_weight_kernel   once per layer at weight-packing time: U = G g G^T in fp32 from the folded OHWI weights, split into
//                        bf16 planes u_hi / u_lo [16][Cout][Cin]
//   wino_input_kernel    per launch: V = B^T d B for every tile and channel (zero padding applied here), written as split
//                        records [16][T][Cin] -- 4 x the input bytes, the price of the method (HBM-bound, ~0.3 ms at B = 64)
//   wino_gemm_kernel     the 16 GEMMs AND the output transform in one pass: a block owns 96 tiles x 128 output channels
//                        and walks K = 16 components x Cin; after each component's Cin / 32 K-steps every wave folds its
//                        16x16 accumulators into the four 2x2 output accumulators with the component's (0, +-1)
//                        coefficients (registers only: M never goes to memory), then bias / residual / ReLU / split and the
//                        record stores of the direct kernels' epilogue.  12 compute waves of 32 x 32
//                        (v_mfma_f32_16x16x32_bf16; five accumulator sets, <= 128 VGPRs: 4 waves per SIMD) + 4 loader waves
//                        that issue the LDS-DMA three K-steps ahead (four 28 KB stages, counted vmcnt, one barrier per K-step).
// A sample's outputs depend only on its own tiles and the fixed K order: independent of the batch it is computed in.
#include <cstdlib>

#include "conv_common.h"

namespace d2t {

typedef __bf16 wbf16x8 __attribute__((ext_vector_type(8)));
typedef float wf32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* wlds_ptr;

namespace {
constexpr int WBK = 32;                  // K-step (channels)
constexpr int WTM = 96, WTN = 128;       // block tile: Winograd tiles x output channels
constexpr int WCW = 12;                  // compute waves (3 x 4 wave tiles of 32 x 32); waves 12..15 are loaders
constexpr int WSTAGE = WTM * 128 + 2 * WTN * 64;  // A records (128 B per row) + B hi plane + B lo plane (64 B per row) = 28 KB
constexpr int WNS = 4;                   // LDS stages
__device__ __forceinline__ int w_aswz(int row, int c) { return c ^ ((row >> 1) & 7); }
__device__ __forceinline__ int w_pswz16(int row, int c) { return c ^ ((row >> 2) & 2); }
template <int N>
__device__ __forceinline__ void w_wait_vm() {
  __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));
}
}  // namespace

// ---- U = G g G^T ------------------------------------------------------------------------------------------------------
// w: folded weights in the packed K order of the direct kernels ([Cout][K], K = ((c >> 5) * 9 + tap) * 32 + (c & 31))
__global__ void wino_weight_kernel(const float* __restrict__ w, uint16_t* __restrict__ u_hi, uint16_t* __restrict__ u_lo, int Cout,
                                   int Cin) {
  const long long total = (long long)Cout * Cin;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % Cin), o = (int)(idx / Cin);
    float g[3][3];
#pragma unroll
    for (int t = 0; t < 9; ++t) g[t / 3][t % 3] = w[(size_t)o * Cin * 9 + ((size_t)(c >> 5) * 9 + t) * 32 + (c & 31)];
    float t4[4][3];  // G g, G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      t4[0][j] = g[0][j];
      t4[1][j] = 0.5f * (g[0][j] + g[1][j] + g[2][j]);
      t4[2][j] = 0.5f * (g[0][j] - g[1][j] + g[2][j]);
      t4[3][j] = g[2][j];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float u[4] = {t4[i][0], 0.5f * (t4[i][0] + t4[i][1] + t4[i][2]), 0.5f * (t4[i][0] - t4[i][1] + t4[i][2]), t4[i][2]};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const __bf16 h = (__bf16)u[j];  // round to nearest even, as launch_split_bf16 does for the direct kernels' weights
        const __bf16 l = (__bf16)(u[j] - (float)h);
        const size_t dst = ((size_t)(i * 4 + j) * Cout + o) * Cin + c;
        u_hi[dst] = *reinterpret_cast<const uint16_t*>(&h);
        u_lo[dst] = *reinterpret_cast<const uint16_t*>(&l);
      }
    }
  }
}

hipError_t launch_wino_weights(const float* w_packed, uint16_t* u_hi, uint16_t* u_lo, int Cout, int Cin, hipStream_t s) {
  const long long total = (long long)Cout * Cin;
  hipLaunchKernelGGL(wino_weight_kernel, dim3((unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096)), dim3(256), 0, s,
                     w_packed, u_hi, u_lo, Cout, Cin);
  return hipGetLastError();
}

// ---- V = B^T d B ------------------------------------------------------------------------------------------------------
// One thread: one tile x eight channels.  in_hi: split records of x [B*H*W][Cin]; v: split records [16][T][Cin].
__global__ __launch_bounds__(256) void wino_input_kernel(const uint16_t* __restrict__ in_hi, uint16_t* __restrict__ v, int B, int H,
                                                         int W, int Cin, int th, int tw) {
  const int oct = Cin >> 3;
  const long long T = (long long)B * th * tw, total = T * oct;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int o8 = (int)(idx % oct);
    const long long tile = idx / oct;
    const int tx = (int)(tile % tw), ty = (int)((tile / tw) % th), b = (int)(tile / ((long long)tw * th));
    const int c = o8 * 8;
    const size_t rec = (size_t)(c & ~31) * 2 + (c & 31);  // offset of the hi part inside a pixel's Cin * 2 uint16
    float d[4][4][8];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int iy = 2 * ty - 1 + r;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int ix = 2 * tx - 1 + q;
        if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
          const uint16_t* px = in_hi + ((size_t)(b * H + iy) * W + ix) * Cin * 2 + rec;
          const uint4 hh = *reinterpret_cast<const uint4*>(px), ll = *reinterpret_cast<const uint4*>(px + 32);
          const unsigned hv[4] = {hh.x, hh.y, hh.z, hh.w}, lv[4] = {ll.x, ll.y, ll.z, ll.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            d[r][q][2 * e] = __uint_as_float(hv[e] << 16) + __uint_as_float(lv[e] << 16);
            d[r][q][2 * e + 1] = __uint_as_float(hv[e] & 0xFFFF0000u) + __uint_as_float(lv[e] & 0xFFFF0000u);
          }
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) d[r][q][e] = 0.f;
        }
      }
    }
    // B^T = [[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]]: rows first, then columns
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d0 = d[0][q][e], d1 = d[1][q][e], d2 = d[2][q][e], d3 = d[3][q][e];
        d[0][q][e] = d0 - d2; d[1][q][e] = d1 + d2; d[2][q][e] = d2 - d1; d[3][q][e] = d1 - d3;
      }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float o[4][8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float t0 = d[i][0][e], t1 = d[i][1][e], t2 = d[i][2][e], t3 = d[i][3][e];
        o[0][e] = t0 - t2; o[1][e] = t1 + t2; o[2][e] = t2 - t1; o[3][e] = t1 - t3;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        uint16_t hi[8], lo[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) split_f32(o[j][e], hi[e], lo[e]);
        uint16_t* dst = v + ((size_t)(i * 4 + j) * T + tile) * Cin * 2 + rec;
        *reinterpret_cast<uint4*>(dst) = make_uint4(hi[0] | (unsigned)hi[1] << 16, hi[2] | (unsigned)hi[3] << 16,
                                                    hi[4] | (unsigned)hi[5] << 16, hi[6] | (unsigned)hi[7] << 16);
        *reinterpret_cast<uint4*>(dst + 32) = make_uint4(lo[0] | (unsigned)lo[1] << 16, lo[2] | (unsigned)lo[3] << 16,
                                                         lo[4] | (unsigned)lo[5] << 16, lo[6] | (unsigned)lo[7] << 16);
      }
    }
  }
}

// ---- the 16 GEMMs + output transform ---------------------------------------------------------------------------------
struct WinoP {
  const uint16_t* v;      // [16][T][Cin] split records
  const uint16_t* u_hi;   // [16][Cout][Cin]
  const uint16_t* u_lo;
  const float* bias;
  const uint16_t* res_hi; // split records [B*H*W][Cout] or nullptr
  const float* res;       // fp32 [B*H*W][Cout] or nullptr
  uint16_t* out_hi;       // split records, or
  float* out;             // fp32
  const void* zero16;     // >= 1 KiB of zeros
  int B, H, W, Cin, Cout, th, tw, T, act;
};

__global__ __launch_bounds__(1024, 1) void wino_gemm_kernel(const WinoP p) {
  // 12 compute waves (3 x 4 grid of 32 x 32 wave tiles over the 96 x 128 block tile) + 4 loader waves (one per SIMD) that own
  // the whole vector-memory side: an LDS-DMA instruction costs its issuing wave 100+ cycles, and with every compute wave
  // issuing its own pieces a K-step measured 2200 cycles against 576 of MFMA work per SIMD.
  __shared__ __attribute__((aligned(1024))) unsigned char smem[WNS * WSTAGE];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int nt = (p.Cout + WTN - 1) / WTN;
  const int ntiles = nt * ((p.T + WTM - 1) / WTM);
  const int nch = p.Cin / WBK, KT = 16 * nch;
  const uint16_t* zero = reinterpret_cast<const uint16_t*>(p.zero16);
  const int G = gridDim.x, xq = G >> 3, xr = G & 7, xcd = blockIdx.x & 7;
  const int slot = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (blockIdx.x >> 3);
  const bool loader = wave >= WCW;  // wave-uniform

  for (int bt = slot; bt < ntiles; bt += G) {
    const int t0 = (bt / nt) * WTM, n0 = (bt % nt) * WTN;
    if (loader) {
      const int lw = wave - WCW;
      // A: 12 pieces of 8 rows (one 128-byte record per row), three per loader; lane -> (row, physical 16-byte chunk)
      const uint16_t* a_src[3];
      const size_t a_comp = (size_t)p.T * p.Cin * 2;  // uint16 per component plane
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int row = (lw * 3 + j) * 8 + (lane >> 3);
        a_src[j] = t0 + row < p.T ? p.v + ((size_t)(t0 + row) * p.Cin) * 2 + w_aswz(row, lane & 7) * 8 : nullptr;
      }
      // B: 8 pieces of 16 rows (64 bytes per row) in each of the hi / lo planes, two of each per loader
      const uint16_t *bh_src[2], *bl_src[2];
      const size_t b_comp = (size_t)p.Cout * p.Cin;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int row = (lw * 2 + j) * 16 + (lane >> 2);
        const bool ok = n0 + row < p.Cout;
        const size_t off = (size_t)(n0 + row) * p.Cin + w_pswz16(row, lane & 3) * 8;
        bh_src[j] = ok ? p.u_hi + off : nullptr;
        bl_src[j] = ok ? p.u_lo + off : nullptr;
      }
      auto issue = [&](int kt) {
        const int comp = kt / nch, c0 = (kt - comp * nch) * WBK;
        unsigned char* st = smem + (kt & (WNS - 1)) * WSTAGE;
#pragma unroll
        for (int j = 0; j < 3; ++j)
          __builtin_amdgcn_global_load_lds(a_src[j] ? a_src[j] + comp * a_comp + (size_t)c0 * 2 : zero, (wlds_ptr)(st + (lw * 3 + j) * 1024), 16, 0, 0);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          __builtin_amdgcn_global_load_lds(bh_src[j] ? bh_src[j] + comp * b_comp + c0 : zero, (wlds_ptr)(st + WTM * 128 + (lw * 2 + j) * 1024), 16, 0, 0);
          __builtin_amdgcn_global_load_lds(bl_src[j] ? bl_src[j] + comp * b_comp + c0 : zero, (wlds_ptr)(st + WTM * 128 + WTN * 64 + (lw * 2 + j) * 1024), 16, 0, 0);
        }
      };
      constexpr int PS = 7;  // pieces per loader and K-step
      issue(0);
      if (KT > 1) issue(1);
      if (KT > 2) issue(2);
      for (int kt = 0; kt < KT; ++kt) {
        // this loader's pieces of K-step kt have landed (those of kt+1, kt+2 may still be in flight) ...
        if (kt + 2 < KT) w_wait_vm<2 * PS>(); else if (kt + 1 < KT) w_wait_vm<PS>(); else w_wait_vm<0>();
        __builtin_amdgcn_s_barrier();  // ... and the other loaders'; the compute waves are done reading K-step kt-1
        if (kt + 3 < KT) issue(kt + 3);
      }
      __builtin_amdgcn_s_barrier();  // (compute waves: epilogue done, the stages may be overwritten)
      continue;
    }
    const int wm = wave >> 2, wn = wave & 3;
    const int r = lane & 15, q = lane >> 4;
    wf32x4 y[2][2][2][2];  // [a][b][i][j]: the 2x2 outputs of this wave's 32 tiles x 32 channels
    wf32x4 macc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) y[a][b][i][j] = wf32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) macc[i][j] = wf32x4{0.f, 0.f, 0.f, 0.f};
    int offa[2], offal[2], offb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = wm * 32 + i * 16 + r;
      offa[i] = row * 128 + w_aswz(row, q) * 16;
      offal[i] = row * 128 + w_aswz(row, 4 + q) * 16;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = wn * 32 + j * 16 + r;
      offb[j] = WTM * 128 + row * 64 + w_pswz16(row, q) * 16;
    }

    int kin = 0, comp = 0;  // K-step inside the component, component
    for (int kt = 0; kt < KT; ++kt) {
      __builtin_amdgcn_s_barrier();  // stage kt is complete for everyone; nobody reads the stage of K-step kt-1 any more
      const unsigned char* st = smem + (kt & (WNS - 1)) * WSTAGE;
      wbf16x8 fah[2], fal[2], fbh[2], fbl[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        fah[i] = *reinterpret_cast<const wbf16x8*>(st + offa[i]);
        fal[i] = *reinterpret_cast<const wbf16x8*>(st + offal[i]);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        fbh[j] = *reinterpret_cast<const wbf16x8*>(st + offb[j]);
        fbl[j] = *reinterpret_cast<const wbf16x8*>(st + offb[j] + WTN * 64);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          wf32x4 c = macc[i][j];
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fal[i], fbh[j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fah[i], fbl[j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fah[i], fbh[j], c, 0, 0, 0);
          macc[i][j] = c;
        }
      if (++kin == nch) {  // component (xi, nu) complete: Y[a][b] += A^T[a][xi] * A^T[b][nu] * M, A^T = [[1,1,1,0],[0,1,-1,-1]]
        const int xi = comp >> 2, nu = comp & 3;
        const float ca0 = xi < 3 ? 1.f : 0.f, ca1 = xi == 0 ? 0.f : (xi == 1 ? 1.f : -1.f);
        const float cb0 = nu < 3 ? 1.f : 0.f, cb1 = nu == 0 ? 0.f : (nu == 1 ? 1.f : -1.f);
        const float s00 = ca0 * cb0, s01 = ca0 * cb1, s10 = ca1 * cb0, s11 = ca1 * cb1;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float m = macc[i][j][e];
              y[0][0][i][j][e] = fmaf(s00, m, y[0][0][i][j][e]);
              y[0][1][i][j][e] = fmaf(s01, m, y[0][1][i][j][e]);
              y[1][0][i][j][e] = fmaf(s10, m, y[1][0][i][j][e]);
              y[1][1][i][j][e] = fmaf(s11, m, y[1][1][i][j][e]);
            }
            macc[i][j] = wf32x4{0.f, 0.f, 0.f, 0.f};
          }
        kin = 0;
        ++comp;
      }
    }
    // ---- epilogue: lane (q, r) holds tiles t0 + 32 wm + 16 i + 4 q + e, channel n0 + 32 wn + 16 j + r ----
    const int thw = p.th * p.tw;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + wn * 32 + j * 16 + r;
      if (n >= p.Cout) continue;
      const float bias = p.bias ? p.bias[n] : 0.f;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int tile = t0 + wm * 32 + i * 16 + 4 * q + e;
          if (tile >= p.T) continue;
          const int bimg = tile / thw, rem = tile - bimg * thw;
          const int ty = rem / p.tw, tx = rem - ty * p.tw;
#pragma unroll
          for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
              const int oy = 2 * ty + a, ox = 2 * tx + b;
              if (oy >= p.H || ox >= p.W) continue;
              const size_t m = ((size_t)bimg * p.H + oy) * p.W + ox;
              float val = y[a][b][i][j][e] + bias;
              const size_t off = m * p.Cout + n;
              const size_t pi = plane_idx(m, n, p.Cout);
              if (p.res) val += p.res[off];
              if (p.res_hi) val += bf16_bits_to_f32(p.res_hi[pi]) + bf16_bits_to_f32(p.res_hi[pi + 32]);
              val = apply_act(val, p.act);
              if (p.out_hi) {
                uint16_t hi, lo;
                split_f32(val, hi, lo);
                p.out_hi[pi] = hi;
                p.out_hi[pi + 32] = lo;
              } else {
                p.out[off] = val;
              }
            }
        }
    }
    __builtin_amdgcn_s_barrier();  // every compute wave is done with the last stages: the loaders may start the next tile
  }
}

// 3x3 / stride 1 / pad 1, Cin % 32 == 0, split-record input: V into `v_ws` (wino_workspace_bytes), then the fused GEMM.
size_t wino_workspace_bytes(int B, int H, int W, int Cin) {
  return (size_t)16 * B * ((H + 1) / 2) * ((W + 1) / 2) * Cin * 4;
}

bool wino_applicable(const ConvP& p) {
  return p.in_hi && p.KH == 3 && p.KW == 3 && p.SH == 1 && p.SW == 1 && p.PH == 1 && p.PW == 1 && p.OH == p.H && p.OW == p.W &&
         p.Cin % 32 == 0 && p.Cout % 32 == 0 && p.store_mode == STORE_ROWS && p.rows_per_img == 0 && !p.row_add && p.m_base == 0 &&
         !p.pool2 && !p.Cin2;
}

hipError_t launch_conv_winograd(const ConvP& p, const uint16_t* u_hi, const uint16_t* u_lo, uint16_t* v_ws, hipStream_t s) {
  if (!wino_applicable(p) || !u_hi || !u_lo || !v_ws || !p.zero16) return hipErrorInvalidValue;
  static int cus = 0;
  if (!cus) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1)
      return hipErrorInvalidDevice;
    cus = n;
  }
  const int th = (p.H + 1) / 2, tw = (p.W + 1) / 2;
  const long long T = (long long)p.B * th * tw;
  if (T * p.Cin * 2 * 16 > 0x7fffffffffffLL || T > 0x7fffffff) return hipErrorInvalidValue;
  {
    const long long total = T * (p.Cin / 8);
    const unsigned blocks = (unsigned)((total + 255) / 256 < (1 << 20) ? (total + 255) / 256 : (1 << 20));
    hipLaunchKernelGGL(wino_input_kernel, dim3(blocks), dim3(256), 0, s, p.in_hi, v_ws, p.B, p.H, p.W, p.Cin, th, tw);
  }
  WinoP q{};
  q.v = v_ws; q.u_hi = u_hi; q.u_lo = u_lo; q.bias = p.bias; q.res_hi = p.res_hi; q.res = p.res; q.out_hi = p.out_hi; q.out = p.out;
  q.zero16 = p.zero16; q.B = p.B; q.H = p.H; q.W = p.W; q.Cin = p.Cin; q.Cout = p.Cout; q.th = th; q.tw = tw; q.T = (int)T; q.act = p.act;
  const int ntiles = ((p.Cout + WTN - 1) / WTN) * (int)((T + WTM - 1) / WTM);
  int grid = cus - (p.reserved_cus > 0 ? p.reserved_cus : 0);
  if (grid < 8) grid = 8;
  if (grid > ntiles) grid = ntiles;
  hipLaunchKernelGGL(wino_gemm_kernel, dim3(grid), dim3(1024), 0, s, q);
  return hipGetLastError();
}

}  // namespace d2t
