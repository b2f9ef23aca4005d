// p2c_norm.hip -- K15: LayerNorm over the last dimension for MANY short rows, forward and backward (gfx950).
//
// The build's PoseTransformer (modules/movements/pose_former/pose_transformer.py; the reference binds the third-party model in
// modules/movements/pose_former/pose_former.py:33-76) normalises 546 624 rows of 32 channels in its spatial blocks and 21 024
// rows of 832 in its temporal ones, 18 times per step. The framework kernels give a whole workgroup-pass to a row and take
// ~0.27 ms per call forward, ~0.5 ms backward, on 70-140 MB of traffic (17-35 us at the HBM roofline). Here a row belongs to
// G lanes of a wave (G = 8 for 32 channels ... 64 for up to 1 024), 16 bytes per lane and load: a wave reads 1 KB contiguous
// per instruction, the statistics are butterfly sums over the G lanes, and nothing but the row itself, gamma / beta and
// two floats of statistics moves. HBM-bound by construction.
//   forward : y = (x - mean) * rstd * gamma + beta, rstd = 1 / sqrt(var + eps) (biased variance, as torch.nn.LayerNorm);
//             mean, rstd (rows) are saved for the backward.
//   backward: xh = (x - mean) rstd;  gg = g * gamma;  dx = rstd * (gg - mean_c(gg) - xh * mean_c(gg * xh));
//             d gamma = sum_rows g * xh, d beta = sum_rows g: a thread always meets the same channels, so it sums them in
//             registers over its rows; the workgroup's threads of equal channels meet in LDS, and a second launch adds the
//             workgroup partials in order (bitwise reproducible).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/p2c.h"

namespace p2c_norm {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int THREADS = 256;

struct Args {
  const float *x, *gamma, *beta, *gy, *gx_add;
  float *y, *mean, *rstd, *gx, *g_gamma, *g_beta, *partials;
  int64_t rows;
  int32_t D, accumulate, n_blocks;
  float eps;
};

template <int G>
__device__ __forceinline__ float group_sum(float v) {         // butterfly over the G lanes of a row
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// G lanes per row, KV float4 per lane: channel of (k, l) = (k G + l) * 4
template <int G, int KV>
__global__ __launch_bounds__(THREADS) void ln_fwd_kernel(const Args a) {
  constexpr int RPW = 64 / G, RPB = RPW * (THREADS / 64);     // rows per wave / per workgroup pass
  const int lane = threadIdx.x & 63, l = lane % G, slot = (threadIdx.x >> 6) * RPW + lane / G;
  const int D = a.D;
  f32x4 gm[KV], bt[KV];
  bool on[KV];
#pragma unroll
  for (int k = 0; k < KV; ++k) {
    const int col = (k * G + l) * 4;
    on[k] = col < D;
#pragma unroll
    for (int c = 0; c < 4; ++c)                  // (scalar reads: parameters inside a flat buffer are only 4-byte aligned)
      gm[k][c] = on[k] ? a.gamma[col + c] : 0.f, bt[k][c] = on[k] ? a.beta[col + c] : 0.f;
  }
  const float inv_d = 1.f / (float)D;
  for (int64_t r0 = (int64_t)blockIdx.x * RPB; r0 < a.rows; r0 += (int64_t)gridDim.x * RPB) {
    const int64_t r = r0 + slot;
    const bool live = r < a.rows;
    f32x4 v[KV];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < KV; ++k) {
      v[k] = (live && on[k]) ? *reinterpret_cast<const f32x4 *>(a.x + r * D + (k * G + l) * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
      s += (v[k][0] + v[k][1]) + (v[k][2] + v[k][3]);
    }
    const float mean = group_sum<G>(s) * inv_d;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < KV; ++k)
      if (on[k]) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float dlt = v[k][c] - mean;
          q = fmaf(dlt, dlt, q);
        }
      }
    const float rstd = rsqrtf(group_sum<G>(q) * inv_d + a.eps);
    if (!live) continue;
#pragma unroll
    for (int k = 0; k < KV; ++k)
      if (on[k]) {
        f32x4 o;
#pragma unroll
        for (int c = 0; c < 4; ++c) o[c] = fmaf((v[k][c] - mean) * rstd, gm[k][c], bt[k][c]);
        *reinterpret_cast<f32x4 *>(a.y + r * D + (k * G + l) * 4) = o;
      }
    if (l == 0) a.mean[r] = mean, a.rstd[r] = rstd;
  }
}

template <int G, int KV>
__global__ __launch_bounds__(THREADS) void ln_bwd_kernel(const Args a) {
  constexpr int RPW = 64 / G, RPB = RPW * (THREADS / 64), SLOTS = RPB;
  __shared__ float red[SLOTS][G * KV * 4 * 2 + 1];           // [row slot][channel slot of the lane, gamma | beta]
  const int lane = threadIdx.x & 63, l = lane % G, slot = (threadIdx.x >> 6) * RPW + lane / G;
  const int D = a.D;
  f32x4 gm[KV], dg[KV], db[KV];
  bool on[KV];
#pragma unroll
  for (int k = 0; k < KV; ++k) {
    const int col = (k * G + l) * 4;
    on[k] = col < D;
#pragma unroll
    for (int c = 0; c < 4; ++c) gm[k][c] = on[k] ? a.gamma[col + c] : 0.f;
    dg[k] = (f32x4){0.f, 0.f, 0.f, 0.f}, db[k] = dg[k];
  }
  const float inv_d = 1.f / (float)D;
  for (int64_t r0 = (int64_t)blockIdx.x * RPB; r0 < a.rows; r0 += (int64_t)gridDim.x * RPB) {
    const int64_t r = r0 + slot;
    const bool live = r < a.rows;
    const float mean = live ? a.mean[r] : 0.f, rstd = live ? a.rstd[r] : 0.f;
    f32x4 xh[KV], gg[KV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < KV; ++k) {
      const bool ok = live && on[k];
      const f32x4 xv = ok ? *reinterpret_cast<const f32x4 *>(a.x + r * D + (k * G + l) * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
      const f32x4 gv = ok ? *reinterpret_cast<const f32x4 *>(a.gy + r * D + (k * G + l) * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        xh[k][c] = ok ? (xv[c] - mean) * rstd : 0.f;
        gg[k][c] = gv[c] * gm[k][c];
        s1 += gg[k][c];
        s2 = fmaf(gg[k][c], xh[k][c], s2);
        dg[k][c] = fmaf(gv[c], xh[k][c], dg[k][c]);
        db[k][c] += gv[c];
      }
    }
    const float m1 = group_sum<G>(s1) * inv_d, m2 = group_sum<G>(s2) * inv_d;
    if (!live) continue;
#pragma unroll
    for (int k = 0; k < KV; ++k)
      if (on[k]) {
        f32x4 o;
#pragma unroll
        for (int c = 0; c < 4; ++c) o[c] = rstd * (gg[k][c] - m1 - xh[k][c] * m2);
        if (a.gx_add) o += *reinterpret_cast<const f32x4 *>(a.gx_add + r * D + (k * G + l) * 4);   // the residual branch's gradient
        *reinterpret_cast<f32x4 *>(a.gx + r * D + (k * G + l) * 4) = o;
      }
  }
  // the SLOTS threads that own the same channels: fixed-order sum through LDS, one partial row per workgroup
#pragma unroll
  for (int k = 0; k < KV; ++k)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      red[slot][((k * G + l) * 4 + c) * 2] = dg[k][c];
      red[slot][((k * G + l) * 4 + c) * 2 + 1] = db[k][c];
    }
  __syncthreads();
  float *out = a.partials + (size_t)blockIdx.x * 2 * D;
  for (int i = threadIdx.x; i < 2 * D; i += THREADS) {
    const int which = i / D, col = i - which * D;             // (which = 0: gamma, 1: beta)
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < SLOTS; ++q) s += red[q][col * 2 + which];
    out[i] = s;
  }
}

// partial rows added in a fixed order: thread (channel, part) sums the workgroups b = part (mod 8), eight loads in flight;
// the eight parts of a channel meet in LDS in part order
__global__ __launch_bounds__(256) void ln_bwd_finish_kernel(const Args a) {
  __shared__ float red[8][33];
  const int ch = threadIdx.x & 31, part = threadIdx.x >> 5, i = blockIdx.x * 32 + ch, n2 = 2 * a.D;
  float s = 0.f;
  if (i < n2) {
    int b = part;
    for (; b + 56 < a.n_blocks; b += 64) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = a.partials[(size_t)(b + 8 * u) * n2 + i];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; b < a.n_blocks; b += 8) s += a.partials[(size_t)b * n2 + i];
  }
  red[part][ch] = s;
  __syncthreads();
  if (part == 0 && i < n2) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) t += red[q][ch];
    float *dst = (i < a.D) ? a.g_gamma + i : a.g_beta + (i - a.D);
    *dst = a.accumulate ? *dst + t : t;
  }
}

static int shape_ok(int64_t rows, int32_t D) { return rows >= 0 && D >= 4 && D <= 1024 && (D & 3) == 0; }
static int blocks_for(int64_t rows, int rpb) {
  int64_t b = (rows + rpb - 1) / rpb;
  if (b > 1024) b = 1024;                                     // 4 workgroups per CU; each strides over the rows
  return b < 1 ? 1 : (int)b;
}
static int rows_per_block(int32_t D) { return D <= 32 ? 32 : D <= 64 ? 16 : D <= 128 ? 8 : 4; }

#define P2C_LN_DISPATCH(KERNEL, grid)                                                                          \
  if (D <= 32) hipLaunchKernelGGL((KERNEL<8, 1>), grid, dim3(THREADS), 0, (hipStream_t)stream, a);              \
  else if (D <= 64) hipLaunchKernelGGL((KERNEL<16, 1>), grid, dim3(THREADS), 0, (hipStream_t)stream, a);        \
  else if (D <= 128) hipLaunchKernelGGL((KERNEL<32, 1>), grid, dim3(THREADS), 0, (hipStream_t)stream, a);       \
  else if (D <= 256) hipLaunchKernelGGL((KERNEL<64, 1>), grid, dim3(THREADS), 0, (hipStream_t)stream, a);       \
  else if (D <= 512) hipLaunchKernelGGL((KERNEL<64, 2>), grid, dim3(THREADS), 0, (hipStream_t)stream, a);       \
  else hipLaunchKernelGGL((KERNEL<64, 4>), grid, dim3(THREADS), 0, (hipStream_t)stream, a);

}  // namespace p2c_norm

extern "C" int p2c_layernorm_supported(int32_t D) { return p2c_norm::shape_ok(0, D); }

extern "C" int64_t p2c_layernorm_workspace_floats(int64_t rows, int32_t D) {
  if (!p2c_norm::shape_ok(rows, D)) return 0;
  return (int64_t)p2c_norm::blocks_for(rows, p2c_norm::rows_per_block(D)) * 2 * D;
}

extern "C" int p2c_layernorm_fwd(const float *x, const float *gamma, const float *beta, float *y, float *mean, float *rstd,
                                 int64_t rows, int32_t D, float eps, void *stream) {
  using namespace p2c_norm;
  if (!x || !gamma || !beta || !y || !mean || !rstd) return P2C_E_NULL;
  if (!shape_ok(rows, D)) return P2C_E_SHAPE;
  if (((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) != 0) return P2C_E_SHAPE;
  if (rows == 0) return 0;
  Args a{};
  a.x = x, a.gamma = gamma, a.beta = beta, a.y = y, a.mean = mean, a.rstd = rstd, a.rows = rows, a.D = D, a.eps = eps;
  const dim3 grid((unsigned)blocks_for(rows, rows_per_block(D)));
  P2C_LN_DISPATCH(ln_fwd_kernel, grid)
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

extern "C" int p2c_layernorm_bwd(const float *x, const float *gamma, const float *mean, const float *rstd, const float *gy,
                                 const float *gx_add, float *gx, float *g_gamma, float *g_beta, int32_t accumulate,
                                 float *partials, int64_t rows, int32_t D, void *stream) {
  using namespace p2c_norm;
  if (!x || !gamma || !mean || !rstd || !gy || !gx || !g_gamma || !g_beta || !partials) return P2C_E_NULL;
  if (!shape_ok(rows, D)) return P2C_E_SHAPE;
  if (((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(gy) | reinterpret_cast<uintptr_t>(gx)) & 15) != 0) return P2C_E_SHAPE;
  Args a{};
  a.x = x, a.gamma = gamma, a.mean = const_cast<float *>(mean), a.rstd = const_cast<float *>(rstd), a.gy = gy, a.gx = gx, a.g_gamma = g_gamma, a.g_beta = g_beta;
  a.partials = partials, a.rows = rows, a.D = D, a.accumulate = accumulate, a.gx_add = gx_add;
  if (gx_add && (reinterpret_cast<uintptr_t>(gx_add) & 15) != 0) return P2C_E_SHAPE;
  a.n_blocks = blocks_for(rows, rows_per_block(D));
  if (rows > 0) {
    const dim3 grid((unsigned)a.n_blocks);
    P2C_LN_DISPATCH(ln_bwd_kernel, grid)
  } else {
    a.n_blocks = 0;
  }
  hipLaunchKernelGGL(ln_bwd_finish_kernel, dim3((unsigned)((2 * D + 31) / 32)), dim3(256), 0, (hipStream_t)stream, a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}
