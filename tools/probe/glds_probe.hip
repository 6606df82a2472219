// Probe: where does global_load_lds_dwordx4 put each lane's 16 bytes?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_ptr;
__global__ void k(const uint4* src, uint4* out, int perm) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[4 * 1024];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned char* dst = smem + wave * 1024;
  const uint4* s = src + wave * 64 + (perm ? (lane ^ 5) : lane);
  __builtin_amdgcn_global_load_lds(s, (lds_ptr)dst, 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  out[threadIdx.x] = *reinterpret_cast<uint4*>(smem + threadIdx.x * 16);
}
int main() {
  const int n = 256;
  std::vector<uint4> h(n);
  for (int i = 0; i < n; ++i) h[i] = make_uint4(i, i * 10, i * 100, i * 1000);
  uint4 *d, *o;
  hipMalloc(&d, n * 16); hipMalloc(&o, n * 16);
  hipMemcpy(d, h.data(), n * 16, hipMemcpyHostToDevice);
  for (int perm = 0; perm < 2; ++perm) {
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, d, o, perm);
    std::vector<uint4> r(n);
    hipMemcpy(r.data(), o, n * 16, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; ++i) {
      const int lane = i & 63, wave = i >> 6;
      const int expect = wave * 64 + (perm ? (lane ^ 5) : lane);
      if ((int)r[i].x != expect || (int)r[i].y != expect * 10) { if (bad < 8) printf("perm %d slot %d got %u,%u expect %d\n", perm, i, r[i].x, r[i].y, expect); ++bad; }
    }
    printf("perm %d: %d mismatches\n", perm, bad);
  }
  return 0;
}
