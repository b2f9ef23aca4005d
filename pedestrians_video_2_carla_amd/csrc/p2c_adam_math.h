// p2c_adam_math.h -- the AdamW / Adam update formula shared by the stand-alone optimizer kernel (p2c_optim.hip) and the
// fused "reduce + update" tail of the MLP backward (p2c_mlp.hip): one definition, bit-identical results.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/p2c.h"

namespace p2c_optim {

struct Coefs {
  float lr_wd, beta1, one_m_beta1, beta2, one_m_beta2, step_size, inv_bc2_sqrt, eps, wd, grad_scale;
};

// beta ** step for the integer step count, by squaring, in double (~50 multiplications instead of the library pow()'s
// several hundred fp64 instructions on one thread); both agree to a few ulp of double, far below the fp32 the result is
// cast to.
__device__ __forceinline__ double ipow(double b, unsigned n) {
  double r = 1.0;
  for (; n; n >>= 1, b *= b)
    if (n & 1u) r *= b;
  return r;
}

// torch/optim/adamw.py (_single_tensor_adamw) / ATen fused_adam_utils.cuh: bias corrections in double, update in fp32
__device__ __forceinline__ Coefs coefs(const p2c_adamw_desc &d, float step) {
  const float lr = d.hyper[0], b1 = d.hyper[1], b2 = d.hyper[2], eps = d.hyper[3], wd = d.hyper[4], gs = d.hyper[5];
  const unsigned n = (unsigned)step;      // the step counter holds an integer
  const double bc1 = 1.0 - ipow((double)b1, n), bc2 = 1.0 - ipow((double)b2, n);
  Coefs c;
  c.lr_wd = lr * wd, c.beta1 = b1, c.one_m_beta1 = 1.f - b1, c.beta2 = b2, c.one_m_beta2 = 1.f - b2;
  c.step_size = (float)((double)lr / bc1), c.inv_bc2_sqrt = (float)(1.0 / sqrt(bc2)), c.eps = eps, c.wd = wd;
  c.grad_scale = gs;
  return c;
}

// No FMA contraction in here: the formula is inlined into three kernels (adamw_kernel, the MLP's gradient reductions, the
// two-launch train step) and must round the same way in each, whatever the surrounding code lets the compiler fuse.
template <bool ADAMW>
__device__ __forceinline__ void update(const Coefs &c, float &p, float g, float &m, float &v) {
#pragma clang fp contract(off)
  g *= c.grad_scale;
  if (ADAMW) p -= c.lr_wd * p;          // decoupled weight decay
  else g += c.wd * p;                   // L2 penalty (Adam)
  m += c.one_m_beta1 * (g - m);         // lerp(exp_avg, grad, 1 - beta1)
  v = c.beta2 * v + c.one_m_beta2 * g * g;
  const float denom = sqrtf(v) * c.inv_bc2_sqrt + c.eps;
  p -= c.step_size * m / denom;
}

}  // namespace p2c_optim
