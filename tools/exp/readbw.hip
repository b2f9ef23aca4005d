// experiment: read-only streaming bandwidth on MI355X -- plain 16-byte loads vs LDS-DMA (buffer_load ... lds), 1.2 GB
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %d at %d\n", (int)e, __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(256) void read_plain(const float4* __restrict__ src, size_t n4, float* out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  float a = 0.f;
  for (; i + 3 * stride < n4; i += 4 * stride) {
    float4 v0 = src[i], v1 = src[i + stride], v2 = src[i + 2 * stride], v3 = src[i + 3 * stride];
    a += v0.x + v1.y + v2.z + v3.w;
  }
  if (a == 12345.678f) out[0] = a;
}

typedef __attribute__((address_space(3))) void* lp_t;
template <int PIECE>
__device__ __forceinline__ void dma(__amdgpu_buffer_rsrc_t r, lp_t dst, int vo, int so) {
  if constexpr (PIECE == 16) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, dst, 16, vo, so, 0, 0);
  else __builtin_amdgcn_raw_ptr_buffer_load_lds(r, dst, 12, vo, so, 0, 0);
}
// each wave: per iteration NI LDS-DMA instructions of 1 KB, wait, read one value per lane, next
template <int NI, int PIECE = 16, int MISALIGN = 0>
__global__ __launch_bounds__(256) void read_dma(const float* src, size_t bytes, float* out, int iters_per_wave) {
  extern __shared__ float4 lds[];
  typedef __attribute__((address_space(3))) void* lp;
  const int lane = threadIdx.x & 63, wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned base = (unsigned)(uintptr_t)(lp)lds + wib * NI * 1024;
  const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const size_t wave_bytes = (size_t)iters_per_wave * NI * 64 * PIECE;
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)src + wave * wave_bytes + MISALIGN), 0, (int)wave_bytes, 0x00020000);
  float a = 0.f;
#pragma unroll
  for (int i = 0; i < NI; ++i) dma<PIECE>(r, (lp)(uintptr_t)(base + i * 1024), lane * PIECE + i * 64 * PIECE, 0);
  for (int it = 0; it < iters_per_wave; ++it) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    a += ((__attribute__((address_space(3))) const float*)(uintptr_t)base)[lane];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (it + 1 < iters_per_wave) {
      const int so = (it + 1) * NI * 64 * PIECE;
#pragma unroll
      for (int i = 0; i < NI; ++i) dma<PIECE>(r, (lp)(uintptr_t)(base + i * 1024), lane * PIECE + i * 64 * PIECE, so);
    }
  }
  if (a == 12345.678f) out[0] = a;
}

// the chain kernel's staging shape: per wave and iteration 4992 B from stream A (16-byte pieces), 1664 B from stream B (16-byte
// pieces), 2496 B from stream C (12-byte pieces); MASKED: surplus lanes of the last instruction of a group switched off
template <bool MASKED, bool ONEBUF, int MODE = 0>
__global__ __launch_bounds__(256) void read_three(const float* a, const float* b, const float* c, float* out, int iters, int waves_total) {
  extern __shared__ float4 lds[];
  const int lane = threadIdx.x & 63, wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned base = (unsigned)(uintptr_t)(lp_t)lds + wib * 11264;
  const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (wave >= (size_t)waves_total) return;
  const size_t wa = (size_t)iters * 4992, wb = (size_t)iters * 1664, wc = (size_t)iters * 2496;
  if (MODE != 0) {
    // MODE 1: frame-major -- iteration `it` of wave w reads chunk (it * waves + w); MODE 2: clip-major rows -- the wave's region
    // is 8 clips x iters rows; iteration `it` reads row `it` of each clip (row = 624 / 208 / 312 bytes)
    __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)a, 0, 0x7ffffff0, 0x00020000);
    __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)b, 0, 0x7ffffff0, 0x00020000);
    __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void*)c, 0, 0x7ffffff0, 0x00020000);
    int va[5], vb[2], vc[4];
    for (int i = 0; i < 5; ++i) { int p = i * 64 + lane; va[i] = MODE == 1 ? (int)(wave * 4992) + p * 16 : (int)(wave * wa) + (p / 39) * iters * 624 + (p % 39) * 16; if (p >= 312) va[i] = 0x7ffffff0; }
    for (int i = 0; i < 2; ++i) { int p = i * 64 + lane; vb[i] = MODE == 1 ? (int)(wave * 1664) + p * 16 : (int)(wave * wb) + (p / 13) * iters * 208 + (p % 13) * 16; if (p >= 104) vb[i] = 0x7ffffff0; }
    for (int i = 0; i < 4; ++i) { int p = i * 64 + lane; vc[i] = MODE == 1 ? (int)(wave * 2496) + p * 12 : (int)(wave * wc) + (p / 26) * iters * 312 + (p % 26) * 12; if (p >= 208) vc[i] = 0x7ffffff0; }
    float acc = 0.f;
    auto issue = [&](int it) {
      const int sa = MODE == 1 ? it * waves_total * 4992 : it * 624, sb = MODE == 1 ? it * waves_total * 1664 : it * 208, sc = MODE == 1 ? it * waves_total * 2496 : it * 312;
#pragma unroll
      for (int i = 0; i < 5; ++i) if (i < 4 || lane < 56) dma<16>(ra, (lp_t)(uintptr_t)(base + i * 1024), va[i], sa);
#pragma unroll
      for (int i = 0; i < 2; ++i) if (i < 1 || lane < 40) dma<16>(rb, (lp_t)(uintptr_t)(base + 5120 + i * 1024), vb[i], sb);
#pragma unroll
      for (int i = 0; i < 4; ++i) if (i < 3 || lane < 16) dma<12>(rc, (lp_t)(uintptr_t)(base + 7168 + i * 1024), vc[i], sc);
    };
    issue(0);
    for (int it = 0; it < iters; ++it) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      acc += ((__attribute__((address_space(3))) const float*)(uintptr_t)base)[lane];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (it + 1 < iters) issue(it + 1);
    }
    if (acc == 12345.678f) out[0] = acc;
    return;
  }
  __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)a + wave * wa), 0, (int)wa, 0x00020000);
  __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)(ONEBUF ? a : b) + wave * wb + (ONEBUF ? (size_t)waves_total * wa : 0)), 0, (int)wb, 0x00020000);
  __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)(ONEBUF ? a : c) + wave * wc + (ONEBUF ? (size_t)waves_total * (wa + wb) : 0)), 0, (int)wc, 0x00020000);
  float acc = 0.f;
  auto issue = [&](int it) {
#pragma unroll
    for (int i = 0; i < 5; ++i) if (!MASKED || i < 4 || lane < 56) dma<16>(ra, (lp_t)(uintptr_t)(base + i * 1024), lane * 16 + i * 1024, it * 4992);
#pragma unroll
    for (int i = 0; i < 2; ++i) if (!MASKED || i < 1 || lane < 40) dma<16>(rb, (lp_t)(uintptr_t)(base + 5120 + i * 1024), lane * 16 + i * 1024, it * 1664);
#pragma unroll
    for (int i = 0; i < 4; ++i) if (!MASKED || i < 3 || lane < 16) dma<12>(rc, (lp_t)(uintptr_t)(base + 7168 + i * 1024), lane * 12 + i * 768, it * 2496);
  };
  issue(0);
  for (int it = 0; it < iters; ++it) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    acc += ((__attribute__((address_space(3))) const float*)(uintptr_t)base)[lane];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (it + 1 < iters) issue(it + 1);
  }
  if (acc == 12345.678f) out[0] = acc;
}

int main() {
  const size_t bytes = (size_t)1200 << 20;
  float *src, *out;
  CK(hipMalloc(&src, bytes)); CK(hipMalloc(&out, 64));
  CK(hipMemset(src, 0, bytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time = [&](auto launch, const char* name) {
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-40s %8.1f us  %6.2f TB/s\n", name, ms * 200.f, bytes / (ms / 5 * 1e-3) / 1e12);
  };
  for (int blocks : {1024, 2048, 4096, 8192})
    time([&] { hipLaunchKernelGGL(read_plain, dim3(blocks), dim3(256), 0, 0, (const float4*)src, bytes / 16, out); },
         blocks == 1024 ? "plain float4 loads, 1024 blocks" : blocks == 2048 ? "plain, 2048 blocks" : blocks == 4096 ? "plain, 4096 blocks" : "plain, 8192 blocks");
  // DMA: waves = bytes / (iters * NI KB); 4 waves per block
  {
    const int iters = 16;
    { constexpr int NI = 9; size_t waves = bytes / ((size_t)iters * NI * 1024);
      time([&] { hipLaunchKernelGGL(read_dma<NI>, dim3(waves / 4), dim3(256), 4 * NI * 1024, 0, src, bytes, out, iters); }, "LDS-DMA 9 KB per wave-iter, 16 iters"); }
    { constexpr int NI = 16; size_t waves = bytes / ((size_t)iters * NI * 1024);
      time([&] { hipLaunchKernelGGL(read_dma<NI>, dim3(waves / 4), dim3(256), 4 * NI * 1024, 0, src, bytes, out, iters); }, "LDS-DMA 16 KB per wave-iter, 16 iters"); }
    { constexpr int NI = 4; size_t waves = bytes / ((size_t)iters * NI * 1024);
      time([&] { hipLaunchKernelGGL(read_dma<NI>, dim3(waves / 4), dim3(256), 4 * NI * 1024, 0, src, bytes, out, iters); }, "LDS-DMA 4 KB per wave-iter, 16 iters"); }
    { constexpr int NI = 12; size_t waves = bytes / ((size_t)iters * NI * 768);
      time([&] { hipLaunchKernelGGL((read_dma<NI, 12>), dim3(waves / 4), dim3(256), 4 * NI * 1024, 0, src, bytes, out, iters); }, "LDS-DMA 12-byte pieces, 9 KB per wave-iter"); }
    { constexpr int NI = 9; size_t waves = bytes / ((size_t)iters * NI * 1024) - 4;
      time([&] { hipLaunchKernelGGL((read_dma<NI, 16, 8>), dim3(waves / 4), dim3(256), 4 * NI * 1024, 0, src, bytes, out, iters); }, "LDS-DMA 16-byte pieces at +8 bytes"); }
    { constexpr int NI = 9; const int it2 = 64; size_t waves = bytes / ((size_t)it2 * NI * 1024);
      time([&] { hipLaunchKernelGGL(read_dma<NI>, dim3(waves / 4), dim3(256), 4 * NI * 1024, 0, src, bytes, out, it2); }, "LDS-DMA 9 KB per wave-iter, 64 iters"); }
  }
  {
    const int iters = 16, waves = 8192;
    float *b, *c; CK(hipMalloc(&b, (size_t)waves * iters * 1664)); CK(hipMalloc(&c, (size_t)waves * iters * 2496));
    CK(hipMemset(b, 0, (size_t)waves * iters * 1664)); CK(hipMemset(c, 0, (size_t)waves * iters * 2496));
    const size_t tot = (size_t)waves * iters * 9152;
    auto time3 = [&](auto launch, const char* name) {
      launch(); hipDeviceSynchronize();
      hipEventRecord(e0);
      for (int i = 0; i < 5; ++i) launch();
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("%-52s %8.1f us  %6.2f TB/s\n", name, ms * 200.f, tot / (ms / 5 * 1e-3) / 1e12);
    };
    time3([&] { hipLaunchKernelGGL((read_three<true, false>), dim3(waves / 4), dim3(256), 4 * 11264, 0, src, b, c, out, iters, waves); }, "three streams 4992+1664+2496 B, masked tails");
    time3([&] { hipLaunchKernelGGL((read_three<false, false>), dim3(waves / 4), dim3(256), 4 * 11264, 0, src, b, c, out, iters, waves); }, "three streams, unmasked (over-reads into the next)");
    time3([&] { hipLaunchKernelGGL((read_three<true, false, 1>), dim3(waves / 4), dim3(256), 4 * 11264, 0, src, b, c, out, iters, waves); }, "three streams, frame-major order");
    time3([&] { hipLaunchKernelGGL((read_three<true, false, 2>), dim3(waves / 4), dim3(256), 4 * 11264, 0, src, b, c, out, iters, waves); }, "three streams, clip-major rows (the kernel's layout)");
    time3([&] { hipLaunchKernelGGL((read_three<true, true>), dim3(waves / 4), dim3(256), 4 * 11264, 0, src, b, c, out, iters, waves); }, "same chunks, one buffer (three regions)");
  }
  return 0;
}
