// experiment: what does a batch of ds_bpermute_b32 cost a wave (the time scans of train_stream_kernel: 36 per round), alone and with
// the CU's other waves doing the same; against the same move done with VALU lane moves (v_permlane32_swap / v_permlane16_swap / DPP).
// build: hipcc -O3 --offload-arch=gfx950 -o bpermbw bpermbw.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %d at %d\n", (int)e, __LINE__); return 1; } } while (0)
constexpr int NV = 36, ROUNDS = 64;

template <int MODE>   // 0: ds_bpermute from lane - 32; 1: v_permlane32_swap; 2: ds_bpermute lane - 8; 3: ds_swizzle? (not used)
__global__ void k(unsigned long long *cyc, float *out) {
  const int lane = threadIdx.x & 63;
  float v[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = (float)(lane * (i + 1));
  const int addr32 = ((lane - 32) & 63) * 4, addr8 = ((lane - 8) & 63) * 4;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int r = 0; r < ROUNDS; ++r) {
    float q[NV];
    if (MODE == 0 || MODE == 2) {
#pragma unroll
      for (int i = 0; i < NV; ++i) q[i] = __int_as_float(__builtin_amdgcn_ds_bpermute(MODE == 0 ? addr32 : addr8, __float_as_int(v[i])));
    } else {
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        // q[lane] = v[lane ^ 32] for every lane (the scan needs only the upper half's view of the lower half)
        auto rr = __builtin_amdgcn_permlane32_swap(__float_as_int(v[i]), __float_as_int(v[i]), false, false);
        q[i] = __int_as_float(lane < 32 ? rr[1] : rr[0]);
      }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = fmaf(v[i], 0.5f, q[i]);     // (one VALU per value: the scan has ~4)
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) s += v[i];
  if (s == 12345.678f) out[0] = s;
  if (lane == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

int main() {
  unsigned long long *cyc;
  float *out;
  CK(hipMalloc(&cyc, 8 * 4096));
  CK(hipMalloc(&out, 64));
  std::vector<unsigned long long> h(4096);
  auto run = [&](const char *name, int mode, int waves) -> int {
    for (int rep = 0; rep < 2; ++rep) {
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(64 * waves), 0, 0, cyc, out);
      if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(64 * waves), 0, 0, cyc, out);
      if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(64 * waves), 0, 0, cyc, out);
      CK(hipDeviceSynchronize());
    }
    CK(hipMemcpy(h.data(), cyc, 256 * waves * 8, hipMemcpyDeviceToHost));
    double avg = 0;
    for (int i = 0; i < 256 * waves; ++i) avg += (double)h[i];
    avg /= 256 * waves;
    printf("%-44s %d waves/CU: %8.0f cycles per round of %d values = %6.1f per value\n", name, waves, avg / ROUNDS, NV, avg / ROUNDS / NV);
    return 0;
  };
  for (int w : {1, 2, 4, 8}) run("ds_bpermute (lane - 32) + 1 fma", 0, w);
  for (int w : {1, 2, 4, 8}) run("ds_bpermute (lane - 8) + 1 fma", 2, w);
  for (int w : {1, 2, 4, 8}) run("v_permlane32_swap + select + 1 fma", 1, w);
  return 0;
}
