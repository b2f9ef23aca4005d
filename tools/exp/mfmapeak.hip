// Developer experiment: what the fp32 MFMA pipes deliver when nothing else is going on -- independent v_mfma_f32_32x32x2_f32
// chains from registers, no memory traffic -- and the shader clock the chip holds while they run (s_memtime against the
// 100 MHz wall clock). The nominal 157.3 TFLOP/s (MI355X_MICROARCH.md) is 256 CUs x 4 SIMDs x 64 FLOP/clk x 2.4 GHz; a sustained
// run settles at the clock the power limit allows, which is the ceiling any fp32 GEMM can be priced against on this box.
//   hipcc --offload-arch=gfx950 -O3 -o tools/exp/mfmapeak tools/exp/mfmapeak.hip && tools/exp/mfmapeak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void mfma_loop(float *out, unsigned long long *clk, int iters, float a0) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  const float a = a0 + threadIdx.x * 1e-9f, b = 1.f - a0;
  const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 16; ++j) s += acc[i][j];
  if (s == 12345.678f) out[0] = s;                       // (keeps the chains alive)
  if (blockIdx.x == 300 && threadIdx.x == 0) clk[0] = c1 - c0, clk[1] = w1 - w0;
}

int main() {
  float *out;
  unsigned long long *clk, h[2];
  hipMalloc(&out, 4), hipMalloc(&clk, 16);
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  const int grid = 256 * 2;                              // two 4-wave workgroups per CU: two wavefronts per SIMD
  for (int iters : {200, 2000, 20000, 200000, 200000, 200000}) {
    hipLaunchKernelGGL(mfma_loop, dim3(grid), dim3(256), 0, 0, out, clk, iters, 0.25f);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(mfma_loop, dim3(grid), dim3(256), 0, 0, out, clk, iters, 0.25f);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double flop = (double)grid * 4 * iters * 16 * 4096.0;
    printf("iters %7d: %9.3f ms  %6.1f TFLOP/s  shader clock %.0f MHz  (cycles per MFMA per SIMD: %.1f)\n", iters, ms, flop / ms * 1e-9,
           h[0] / (h[1] / 100.0), (double)h[0] / (iters * 16 * 2));
  }
  return 0;
}
