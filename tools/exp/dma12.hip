// experiment: layout of buffer_load_dwordx3 ... lds (12-byte LDS-DMA) on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const float* src, float* dst, int n) {
  extern __shared__ float lds[];
  for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = -1.f;
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, n * 4, 0x00020000);
  int lane = threadIdx.x & 63;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 12, lane * 12, 0, 0, 0);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(lds + 192), 12, lane * 12, 768, 0, 0);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  for (int i = threadIdx.x; i < 1024; i += 64) dst[i] = lds[i];
}
int main() {
  std::vector<float> h(4096);
  for (int i = 0; i < 4096; ++i) h[i] = (float)i;
  float *s, *d;
  hipMalloc(&s, 4096 * 4); hipMalloc(&d, 1024 * 4);
  hipMemcpy(s, h.data(), 4096 * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, s, d, 4096);
  std::vector<float> o(1024);
  hipMemcpy(o.data(), d, 1024 * 4, hipMemcpyDeviceToHost);
  for (int i = 0; i < 400; ++i) { printf("%g ", o[i]); if (i % 24 == 23) printf("\n"); }
  printf("\n");
  return 0;
}
