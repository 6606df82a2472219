// Probe of ds_read_b64_tr_b16 (gfx950): prints, for every lane, the 4 elements it receives from a 64-column image whose
// element (row, col) holds row * 64 + col.  Build: hipcc --offload-arch=gfx950 tr_probe.hip -o tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void k(short* out) {
  __shared__ short sm[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) sm[i] = (short)i;
  __syncthreads();
  const int lane = threadIdx.x & 15, g = threadIdx.x >> 4;
  // group g reads the block of 4 rows x 16 columns at rows 4g.., columns 0..15; lane 4q+p supplies row q, cols 4p..4p+3
  __attribute__((address_space(3))) v4s* p =
      (__attribute__((address_space(3))) v4s*)(sm + (4 * g + (lane >> 2)) * 64 + (lane & 3) * 4);
  v4s v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p);
  for (int e = 0; e < 4; ++e) out[threadIdx.x * 4 + e] = v[e];
}
int main() {
  short* d;
  hipMalloc(&d, 256 * 2);
  k<<<1, 64>>>(d);
  short h[256];
  hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) printf("lane %2d: (%d,%d) (%d,%d) (%d,%d) (%d,%d)\n", l, h[l*4]/64, h[l*4]%64, h[l*4+1]/64, h[l*4+1]%64, h[l*4+2]/64, h[l*4+2]%64, h[l*4+3]/64, h[l*4+3]%64);
  return 0;
}
