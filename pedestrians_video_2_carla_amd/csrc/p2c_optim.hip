// p2c_optim.hip -- fused AdamW / Adam step over ONE flat fp32 parameter buffer (gfx950).
//
// The reference optimises with torch.optim.AdamW (modules/flow/base_model.py:156-158 configure_optimizers). The trainer
// here keeps every trainable parameter as a view of one flat buffer (parallel/flat.py), so the whole optimizer step is
// one element-wise pass: 4 reads + 3 writes (+1 when it also leaves the gradient zeroed) per parameter, one launch.
// Everything that varies between steps lives in device memory (step counter, hyper-parameters), so the launch can sit in
// a HIP graph and still follow a learning-rate scheduler.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/p2c.h"
#include "p2c_adam_math.h"

namespace p2c_optim {

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool ADAMW>
__global__ __launch_bounds__(256) void adamw_kernel(const p2c_adamw_desc d) {
  const float step = *d.step + 1.f;      // every workgroup reads the counter before it takes its completion ticket
  __shared__ Coefs sc;                   // the double-precision pow() of the bias corrections: once per workgroup
  if (threadIdx.x == 0) sc = coefs(d, step);
  __syncthreads();
  const Coefs c = sc;
  const int64_t n4 = d.n >> 2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  f32x4 *P = reinterpret_cast<f32x4 *>(d.param), *G = reinterpret_cast<f32x4 *>(d.grad);
  f32x4 *M = reinterpret_cast<f32x4 *>(d.exp_avg), *V = reinterpret_cast<f32x4 *>(d.exp_avg_sq);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    f32x4 p = P[i], g = G[i], m = M[i], v = V[i];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float pk = p[k], mk = m[k], vk = v[k];
      update<ADAMW>(c, pk, g[k], mk, vk);
      p[k] = pk, m[k] = mk, v[k] = vk;
    }
    P[i] = p, M[i] = m, V[i] = v;
    if (d.zero_grad) G[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (d.scatter_idx) {                 // second copy of selected parameters in a consumer's own layout
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int j = d.scatter_idx[4 * i + k];
        if (j >= 0) d.scatter_dst[j] = p[k];
      }
    }
  }
  for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < d.n; i += stride) {
    float p = d.param[i], m = d.exp_avg[i], v = d.exp_avg_sq[i];
    update<ADAMW>(c, p, d.grad[i], m, v);
    d.param[i] = p, d.exp_avg[i] = m, d.exp_avg_sq[i] = v;
    if (d.zero_grad) d.grad[i] = 0.f;
    if (d.scatter_idx) {
      const int j = d.scatter_idx[i];
      if (j >= 0) d.scatter_dst[j] = p;
    }
  }
  // The last workgroup to finish publishes the new step count. No fence: the counter is independent of the parameter
  // stores, every workgroup has consumed its (start-of-kernel) read of the old value before it draws a ticket, and a
  // device-scope release fence (cross-XCD L2 write-back) would cost ~4 us -- as much as the whole kernel.
  __syncthreads();
  if (threadIdx.x == 0) {
    if (atomicAdd(d.ticket, 1) == (int)gridDim.x - 1) {
      *d.step = step;
      *d.ticket = 0;
    }
  }
}

}  // namespace p2c_optim

extern "C" int p2c_adamw_step(const p2c_adamw_desc *desc, void *stream_) {
  if (!desc || !desc->param || !desc->grad || !desc->exp_avg || !desc->exp_avg_sq || !desc->step || !desc->ticket ||
      !desc->hyper)
    return P2C_E_NULL;
  if (desc->n < 0) return P2C_E_SHAPE;
  if ((desc->scatter_idx == nullptr) != (desc->scatter_dst == nullptr)) return P2C_E_NULL;
  if (desc->n == 0) return 0;
  for (const void *p : {(const void *)desc->param, (const void *)desc->grad, (const void *)desc->exp_avg,
                        (const void *)desc->exp_avg_sq})
    if (reinterpret_cast<uintptr_t>(p) & 15) return P2C_E_SHAPE;       // flat buffers are 16-byte aligned
  const int64_t n4 = (desc->n + 3) >> 2;
  int64_t blocks = (n4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (desc->adamw)
    hipLaunchKernelGGL(p2c_optim::adamw_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream_, *desc);
  else
    hipLaunchKernelGGL(p2c_optim::adamw_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream_, *desc);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}
