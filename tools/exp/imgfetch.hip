// experiment: how fast does every CU get the SAME 84 KB (the packed weight image of the fused train step) into LDS when all
// 256 workgroups ask for it at once?  Variants: same buffer / own buffer per workgroup; pieces in natural order / rotated per
// workgroup; LDS-DMA / plain loads + ds_write; one workgroup alone (latency floor).
// build: hipcc -O3 --offload-arch=gfx950 -o imgfetch imgfetch.hip ; run: ./imgfetch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %d at %d\n", (int)e, __LINE__); return 1; } } while (0)
typedef __attribute__((address_space(3))) void *lp_t;
constexpr int PIECES = 84, WAVES = 8;

template <int MODE>   // 0 = LDS-DMA, 1 = plain loads + ds_write
__global__ __launch_bounds__(64 * WAVES) void fetch(const float *src, size_t wg_stride_floats, int rotate, unsigned long long *cyc, float *out) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const float *mine = src + (size_t)blockIdx.x * wg_stride_floats;
  const int rot = rotate ? (int)((blockIdx.x * rotate) % PIECES) : 0;
  const unsigned long long t0 = __builtin_readcyclecounter();
  if (MODE == 0) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(mine), 0, PIECES * 1024, 0x00020000);
    const unsigned base = (unsigned)(uintptr_t)(lp_t)lds;
#pragma unroll
    for (int i = 0; i < (PIECES + WAVES - 1) / WAVES; ++i) {
      int pc = wave + i * WAVES;
      if (pc < PIECES) {
        pc = pc + rot >= PIECES ? pc + rot - PIECES : pc + rot;
        pc = __builtin_amdgcn_readfirstlane(pc);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lp_t)(uintptr_t)(base + pc * 1024), 16, (pc * 64 + lane) * 16, 0, 0, 0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    float4 v[(PIECES + WAVES - 1) / WAVES];
#pragma unroll
    for (int i = 0; i < (PIECES + WAVES - 1) / WAVES; ++i) {
      int pc = wave + i * WAVES;
      pc = pc < PIECES ? pc : 0;
      pc = pc + rot >= PIECES ? pc + rot - PIECES : pc + rot;
      v[i] = reinterpret_cast<const float4 *>(mine)[pc * 64 + lane];
    }
#pragma unroll
    for (int i = 0; i < (PIECES + WAVES - 1) / WAVES; ++i) {
      int pc = wave + i * WAVES;
      if (pc < PIECES) {
        pc = pc + rot >= PIECES ? pc + rot - PIECES : pc + rot;
        reinterpret_cast<float4 *>(lds)[pc * 64 + lane] = v[i];
      }
    }
  }
  __syncthreads();
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  if (lds[threadIdx.x * 7] == 12345.678f) out[0] = 1.f;
}

// the image is rewritten between steps (AdamW refreshes it from a few workgroups): 4-byte scattered stores like the optimizer's
__global__ __launch_bounds__(256) void rewrite(float *img, size_t n, float v) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) img[i] = v;
}

int main() {
  const size_t img = (size_t)PIECES * 256;   // floats
  const int NB = 256;
  float *src, *out;
  unsigned long long *cyc;
  CK(hipMalloc(&src, img * 4 * NB));
  CK(hipMalloc(&out, 64));
  CK(hipMalloc(&cyc, NB * 8));
  CK(hipMemset(src, 0, img * 4 * NB));
  CK(hipFuncSetAttribute((const void *)fetch<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute((const void *)fetch<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  std::vector<unsigned long long> h(NB);
  int g_rewrite = 0;
  auto run = [&](const char *name, int mode, int blocks, size_t stride, int rotate) -> int {
    for (int rep = 0; rep < 3; ++rep) {
      if (g_rewrite) hipLaunchKernelGGL(rewrite, dim3(79), dim3(256), 0, 0, src, stride ? img * NB : img, (float)rep);
      if (mode == 0) hipLaunchKernelGGL(fetch<0>, dim3(blocks), dim3(64 * WAVES), PIECES * 1024, 0, src, stride, rotate, cyc, out);
      else hipLaunchKernelGGL(fetch<1>, dim3(blocks), dim3(64 * WAVES), PIECES * 1024, 0, src, stride, rotate, cyc, out);
      CK(hipDeviceSynchronize());
    }
    CK(hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost));
    std::vector<unsigned long long> s(h.begin(), h.begin() + blocks);
    std::sort(s.begin(), s.end());
    double avg = 0;
    for (auto v : s) avg += (double)v;
    printf("%-58s blocks %3d: min %6llu  median %6llu  max %6llu  avg %8.0f cycles\n", name, blocks, s.front(), s[blocks / 2], s.back(), avg / blocks);
    return 0;
  };
  run("LDS-DMA, same image, natural order", 0, 256, 0, 0);
  run("LDS-DMA, same image, rotated by 7 pieces per workgroup", 0, 256, 0, 7);
  run("LDS-DMA, same image, rotated by 1 piece per workgroup", 0, 256, 0, 1);
  run("LDS-DMA, own image per workgroup (21 MB)", 0, 256, img, 0);
  run("LDS-DMA, same image, ONE workgroup", 0, 1, 0, 0);
  run("LDS-DMA, same image, 8 workgroups (one per XCD)", 0, 8, 0, 0);
  run("LDS-DMA, same image, 32 workgroups", 0, 32, 0, 0);
  run("LDS-DMA, same image, 128 workgroups", 0, 128, 0, 0);
  run("plain loads + ds_write, same image, natural order", 1, 256, 0, 0);
  run("plain loads + ds_write, same image, rotated by 7", 1, 256, 0, 7);
  run("plain loads + ds_write, own image per workgroup", 1, 256, img, 0);
  run("plain loads + ds_write, ONE workgroup", 1, 1, 0, 0);
  g_rewrite = 1;
  printf("---- the image rewritten by another kernel before every launch ----\n");
  run("LDS-DMA, same image, natural order", 0, 256, 0, 0);
  run("LDS-DMA, same image, rotated by 7 pieces per workgroup", 0, 256, 0, 7);
  run("LDS-DMA, own image per workgroup (21 MB)", 0, 256, img, 0);
  run("LDS-DMA, same image, ONE workgroup", 0, 1, 0, 0);
  run("LDS-DMA, same image, 8 workgroups (one per XCD)", 0, 8, 0, 0);
  run("LDS-DMA, same image, 32 workgroups", 0, 32, 0, 0);
  run("plain loads + ds_write, same image, natural order", 1, 256, 0, 0);
  run("plain loads + ds_write, own image per workgroup", 1, 256, img, 0);
  return 0;
}
