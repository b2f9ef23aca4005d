// p2c_attn.hip -- K14: multi-head self-attention over SHORT token sequences, one launch forward, one backward (gfx950).
//
// PoseFormer (reference modules/movements/pose_former/pose_former.py:33-76 binds third_party PoseTransformer; the build's own
// restatement is modules/movements/pose_former/pose_transformer.py) attends over 26 joint tokens of width 32 (8 heads of FOUR
// channels) in its spatial blocks and over 9 frame tokens of width 832 (8 heads of 104) in its temporal blocks: at cfg5 21 024
// and 2 336 sequences per block and step. The framework's fused attention is built for long sequences and head widths of
// 32-256: on these shapes it took 0.35 ms forward and 1.1 ms backward per block (four launches) for 7-60 MFLOP. Here a
// workgroup owns one sequence: its q, k, v rows (and the output gradient) sit in LDS, the N x N score matrices of all heads
// too, and every phase is a flat loop over independent outputs -- the work is a few hundred FMAs per thread, so the launch
// is bound by streaming the rows (13-120 KB per sequence) through HBM. With head widths of 4 there is nothing for a 16-wide
// MFMA tile to chew on (K = 4, 26 x 26 scores): plain VALU dot products.
//   qkv (S, N, 3, Hh, D) = the qkv Linear's output viewed; out (S, N, Hh*D);  P = softmax(scale * q k^T) per head.
// Backward recomputes P from q, k (no saved probabilities):
//   dV = P^T dO;  dP = dO V^T;  dS = P * (dP - rowsum(dP * P));  dQ = scale dS K;  dK = scale dS^T Q.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/p2c.h"

namespace p2c_attn {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Args {
  const float *qkv;   // (S, N, 3, Hh, D)
  const float *g_out; // (S, N, Hh*D)        [bwd]
  float *out;         // (S, N, Hh*D)        [fwd]
  float *g_qkv;       // (S, N, 3, Hh, D)    [bwd]
  float scale;
  int32_t S, N, Hh, D;
};

__device__ __forceinline__ void copy_in(float *dst, const float *src, int n) {      // n % 4 == 0, both 16-byte aligned
  const f32x4 *s4 = reinterpret_cast<const f32x4 *>(src);
  f32x4 *d4 = reinterpret_cast<f32x4 *>(dst);
  for (int i = threadIdx.x; i < (n >> 2); i += blockDim.x) d4[i] = s4[i];
}

// P[h][i][j] = softmax_j(scale * <q_i, k_j>_h) into `P` (Hh * N * N floats); rows = the qkv image in LDS, row pitch 3E
__device__ __forceinline__ void probabilities(const float *rows, float *P, float scale, int N, int Hh, int D) {
  const int E = Hh * D, NN = N * N;
  for (int idx = threadIdx.x; idx < Hh * NN; idx += blockDim.x) {
    const int h = idx / NN, r = idx - h * NN, i = r / N, j = r - i * N;
    const float *q = rows + i * 3 * E + h * D, *k = rows + j * 3 * E + E + h * D;
    float s0 = 0.f, s1 = 0.f;
    int d = 0;
    for (; d + 1 < D; d += 2) s0 = fmaf(q[d], k[d], s0), s1 = fmaf(q[d + 1], k[d + 1], s1);
    if (d < D) s0 = fmaf(q[d], k[d], s0);
    P[idx] = (s0 + s1) * scale;
  }
  __syncthreads();
  for (int row = threadIdx.x; row < Hh * N; row += blockDim.x) {
    float *p = P + row * N;
    float m = p[0];
    for (int j = 1; j < N; ++j) m = fmaxf(m, p[j]);
    float sum = 0.f;
    for (int j = 0; j < N; ++j) {
      const float e = __expf(p[j] - m);
      p[j] = e, sum += e;
    }
    const float inv = 1.f / sum;
    for (int j = 0; j < N; ++j) p[j] *= inv;
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void attn_fwd_kernel(const Args a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int N = a.N, Hh = a.Hh, D = a.D, E = Hh * D;
  float *rows = lds, *P = lds + N * 3 * E;
  for (int s = blockIdx.x; s < a.S; s += gridDim.x) {
    copy_in(rows, a.qkv + (size_t)s * N * 3 * E, N * 3 * E);
    __syncthreads();
    probabilities(rows, P, a.scale, N, Hh, D);
    float *o = a.out + (size_t)s * N * E;
    for (int idx = threadIdx.x; idx < N * E; idx += blockDim.x) {
      const int i = idx / E, e = idx - i * E, h = e / D;
      const float *p = P + (h * N + i) * N, *v = rows + 2 * E + e;
      float acc = 0.f;
      for (int j = 0; j < N; ++j) acc = fmaf(p[j], v[j * 3 * E], acc);
      o[idx] = acc;
    }
    __syncthreads();                               // the image is rewritten by the next sequence
  }
}

__global__ __launch_bounds__(256) void attn_bwd_kernel(const Args a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int N = a.N, Hh = a.Hh, D = a.D, E = Hh * D, NN = N * N;
  float *rows = lds, *dO = rows + N * 3 * E, *P = dO + N * E, *dS = P + Hh * NN;
  for (int s = blockIdx.x; s < a.S; s += gridDim.x) {
    copy_in(rows, a.qkv + (size_t)s * N * 3 * E, N * 3 * E);
    copy_in(dO, a.g_out + (size_t)s * N * E, N * E);
    __syncthreads();
    probabilities(rows, P, a.scale, N, Hh, D);
    // dP[h][i][j] = <dO_i, v_j>_h
    for (int idx = threadIdx.x; idx < Hh * NN; idx += blockDim.x) {
      const int h = idx / NN, r = idx - h * NN, i = r / N, j = r - i * N;
      const float *g = dO + i * E + h * D, *v = rows + j * 3 * E + 2 * E + h * D;
      float s0 = 0.f, s1 = 0.f;
      int d = 0;
      for (; d + 1 < D; d += 2) s0 = fmaf(g[d], v[d], s0), s1 = fmaf(g[d + 1], v[d + 1], s1);
      if (d < D) s0 = fmaf(g[d], v[d], s0);
      dS[idx] = s0 + s1;
    }
    __syncthreads();
    // dS = P * (dP - sum_j dP P), times the score scale
    for (int row = threadIdx.x; row < Hh * N; row += blockDim.x) {
      float *ds = dS + row * N;
      const float *p = P + row * N;
      float rs = 0.f;
      for (int j = 0; j < N; ++j) rs = fmaf(ds[j], p[j], rs);
      for (int j = 0; j < N; ++j) ds[j] = p[j] * (ds[j] - rs) * a.scale;
    }
    __syncthreads();
    float *g = a.g_qkv + (size_t)s * N * 3 * E;
    for (int idx = threadIdx.x; idx < N * 3 * E; idx += blockDim.x) {
      const int n = idx / (3 * E), r = idx - n * 3 * E, which = r / E, e = r - which * E, h = e / D;
      float acc = 0.f;
      if (which == 0) {                            // dQ_n = sum_j dS[n][j] k_j
        const float *ds = dS + (h * N + n) * N, *k = rows + E + e;
        for (int j = 0; j < N; ++j) acc = fmaf(ds[j], k[j * 3 * E], acc);
      } else if (which == 1) {                     // dK_n = sum_i dS[i][n] q_i
        const float *ds = dS + h * NN + n, *q = rows + e;
        for (int i = 0; i < N; ++i) acc = fmaf(ds[i * N], q[i * 3 * E], acc);
      } else {                                     // dV_n = sum_i P[i][n] dO_i
        const float *p = P + h * NN + n, *go = dO + e;
        for (int i = 0; i < N; ++i) acc = fmaf(p[i * N], go[i * E], acc);
      }
      g[idx] = acc;
    }
    __syncthreads();
  }
}

static int check(const Args &a, bool bwd, size_t *lds) {
  if (a.S < 0 || a.N < 1 || a.N > 64 || a.Hh < 1 || a.D < 1 || ((a.Hh * a.D) & 3)) return P2C_E_SHAPE;
  const size_t E = (size_t)a.Hh * a.D, NN = (size_t)a.N * a.N;
  *lds = sizeof(float) * (bwd ? a.N * 4 * E + 2 * a.Hh * NN : a.N * 3 * E + a.Hh * NN);
  return *lds <= 156 * 1024 ? 0 : P2C_E_SHAPE;
}

}  // namespace p2c_attn

extern "C" int p2c_attn_small_supported(int32_t N, int32_t heads, int32_t head_dim) {
  p2c_attn::Args a{};
  a.N = N, a.Hh = heads, a.D = head_dim;
  size_t lds;
  return p2c_attn::check(a, true, &lds) == 0;
}

extern "C" int p2c_attn_small_fwd(const float *qkv, float *out, float scale, int32_t S, int32_t N, int32_t heads, int32_t head_dim,
                                  void *stream) {
  using namespace p2c_attn;
  if (!qkv || !out) return P2C_E_NULL;
  Args a{};
  a.qkv = qkv, a.out = out, a.scale = scale, a.S = S, a.N = N, a.Hh = heads, a.D = head_dim;
  size_t lds;
  int rc = check(a, false, &lds);
  if (rc) return rc;
  if (S == 0) return 0;
  static bool allowed = false;
  if (!allowed) {
    (void)hipFuncSetAttribute((const void *)attn_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
    (void)hipFuncSetAttribute((const void *)attn_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
    allowed = true;
  }
  const int per_cu = (int)((156 * 1024) / lds) < 1 ? 1 : (int)((156 * 1024) / lds);
  int grid = 256 * (per_cu > 8 ? 8 : per_cu) * 4;               // a few rounds of resident workgroups; each strides over S
  if (grid > S) grid = S;
  hipLaunchKernelGGL(attn_fwd_kernel, dim3((unsigned)grid), dim3(256), lds, (hipStream_t)stream, a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

extern "C" int p2c_attn_small_bwd(const float *qkv, const float *g_out, float *g_qkv, float scale, int32_t S, int32_t N,
                                  int32_t heads, int32_t head_dim, void *stream) {
  using namespace p2c_attn;
  if (!qkv || !g_out || !g_qkv) return P2C_E_NULL;
  Args a{};
  a.qkv = qkv, a.g_out = g_out, a.g_qkv = g_qkv, a.scale = scale, a.S = S, a.N = N, a.Hh = heads, a.D = head_dim;
  size_t lds;
  int rc = check(a, true, &lds);
  if (rc) return rc;
  if (S == 0) return 0;
  static bool allowed = false;
  if (!allowed) {
    (void)hipFuncSetAttribute((const void *)attn_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
    (void)hipFuncSetAttribute((const void *)attn_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
    allowed = true;
  }
  const int per_cu = (int)((156 * 1024) / lds) < 1 ? 1 : (int)((156 * 1024) / lds);
  int grid = 256 * (per_cu > 8 ? 8 : per_cu) * 4;
  if (grid > S) grid = S;
  hipLaunchKernelGGL(attn_bwd_kernel, dim3((unsigned)grid), dim3(256), lds, (hipStream_t)stream, a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}
