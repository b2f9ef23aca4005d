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

// Index arithmetic: every phase is a flat loop whose index splits into (row, column) by a RUNTIME divisor that is uniform
// over the launch; a 32-bit integer division is ~40 instructions on this ISA, the float reciprocal below 3 (exact for the
// index ranges here: indices < 2^20, divisors < 2^13).
struct Div {
  float inv;
  int d;
  __device__ __forceinline__ explicit Div(int d_) : inv(1.f / (float)d_), d(d_) {}
  __device__ __forceinline__ int quot(int x) const { return (int)(((float)x + 0.5f) * inv); }
};

// Token rows in LDS carry a pitch of (row + 4) floats: 16-byte aligned, and the row stride is no longer a multiple of 32
// banks (3E = 96 / 2 496 floats: lanes that walk over tokens would all hit one or two banks).
__device__ __forceinline__ void copy_rows(float *dst, int pitch, const float *src, int rows, int row_floats, const Div &per_row) {
  const f32x4 *s4 = reinterpret_cast<const f32x4 *>(src);
  const int r4 = row_floats >> 2, n4 = rows * r4, nt = blockDim.x;
  constexpr int U = 8;                            // loads in flight per thread (a 90 KB sequence image is 23 per thread)
  for (int i0 = threadIdx.x; i0 < n4; i0 += nt * U) {
    f32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = (i0 + u * nt < n4) ? s4[i0 + u * nt] : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + u * nt;
      if (i >= n4) continue;
      const int n = per_row.quot(i), c = i - n * r4;
      *reinterpret_cast<f32x4 *>(dst + n * pitch + 4 * c) = v[u];
    }
  }
}

// S[h][i][j] = <a_i, b_j>_h for all heads: a rows at `ar` (pitch ap, head h at column h D), b rows at `br` (pitch bp)
__device__ __forceinline__ void head_dots(float *S, const float *ar, int ap, const float *br, int bp, float scale, int N, int Hh,
                                          int D, const Div &dN) {
  const int NN = N * N;
  for (int h = 0; h < Hh; ++h)
    for (int r = threadIdx.x; r < NN; r += blockDim.x) {
      const int i = dN.quot(r), j = r - i * N;
      const float *a = ar + i * ap + h * D, *b = br + j * bp + h * D;
      float s0 = 0.f, s1 = 0.f;
      int d = 0;
      for (; d + 3 < D; d += 4) {
        const f32x4 av = *reinterpret_cast<const f32x4 *>(a + d), bv = *reinterpret_cast<const f32x4 *>(b + d);
        s0 = fmaf(av[0], bv[0], s0), s1 = fmaf(av[1], bv[1], s1), s0 = fmaf(av[2], bv[2], s0), s1 = fmaf(av[3], bv[3], s1);
      }
      for (; d < D; ++d) s0 = fmaf(a[d], b[d], s0);
      S[h * NN + r] = (s0 + s1) * scale;
    }
}

// A thread owns a row of N <= 64 scores: all of it is read into registers by one burst of LDS loads (compile-time bound NB,
// guarded by the runtime N), reduced there and written back -- a loop with a load-use-store body per element pays one LDS
// round trip per element, and only Hh * N of the 256 threads have a row.
template <int NB>
__device__ __forceinline__ void softmax_rows_nb(float *P, int rows, int N) {
  for (int row = threadIdx.x; row < rows; row += blockDim.x) {
    float *p = P + row * N;
    float v[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) v[j] = (j < N) ? p[j] : -3.0e38f;
    float m = v[0];
#pragma unroll
    for (int j = 1; j < NB; ++j) m = fmaxf(m, v[j]);
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < NB; ++j) v[j] = __expf(v[j] - m), sum += v[j];       // (padding: exp(-huge) = 0)
    const float inv = 1.f / sum;
#pragma unroll
    for (int j = 0; j < NB; ++j)
      if (j < N) p[j] = v[j] * inv;
  }
}
__device__ __forceinline__ void softmax_rows(float *P, int rows, int N) {
  if (N <= 16) softmax_rows_nb<16>(P, rows, N);
  else if (N <= 32) softmax_rows_nb<32>(P, rows, N);
  else softmax_rows_nb<64>(P, rows, N);
}
// dS = P * (dP - sum_j dP P) * scale, row by row, same register scheme
template <int NB>
__device__ __forceinline__ void ds_rows_nb(float *dS, const float *P, int rows, int N, float scale) {
  for (int row = threadIdx.x; row < rows; row += blockDim.x) {
    float *ds = dS + row * N;
    const float *p = P + row * N;
    float dv[NB], pv[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) dv[j] = (j < N) ? ds[j] : 0.f, pv[j] = (j < N) ? p[j] : 0.f;
    float rs = 0.f;
#pragma unroll
    for (int j = 0; j < NB; ++j) rs = fmaf(dv[j], pv[j], rs);
#pragma unroll
    for (int j = 0; j < NB; ++j)
      if (j < N) ds[j] = pv[j] * (dv[j] - rs) * scale;
  }
}
__device__ __forceinline__ void ds_rows(float *dS, const float *P, int rows, int N, float scale) {
  if (N <= 16) ds_rows_nb<16>(dS, P, rows, N, scale);
  else if (N <= 32) ds_rows_nb<32>(dS, P, rows, N, scale);
  else ds_rows_nb<64>(dS, P, rows, N, scale);
}

// y[n] = sum_k m[n * sn + k * sk] * x[k * xp] for n < N, written to y[n * yp]: the thread's N values of x are read into
// registers once (compile-time bound NB), the matrix entries are wave-uniform LDS broadcasts. For wide rows (E >= the
// workgroup) this replaces "one output per thread and round": 9 + 81 LDS reads per 9 outputs instead of 162, and the index
// split once per channel instead of once per output.
template <int NB>
__device__ __forceinline__ void channel_contract(const float *m, int sn, int sk, const float *x, int xp, float *y, int yp, int N) {
  float xv[NB];
#pragma unroll
  for (int k = 0; k < NB; ++k) xv[k] = (k < N) ? x[k * xp] : 0.f;
  for (int n = 0; n < N; ++n) {
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int k = 0; k < NB; k += 2) {
      if (k < N) a0 = fmaf(m[n * sn + k * sk], xv[k], a0);
      if (k + 1 < N) a1 = fmaf(m[n * sn + (k + 1) * sk], xv[k + 1], a1);
    }
    y[n * yp] = a0 + a1;
  }
}

__global__ __launch_bounds__(256) void attn_fwd_kernel(const Args a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int N = a.N, Hh = a.Hh, D = a.D, E = Hh * D, RP = 3 * E + 4, NN = N * N;
  const bool vec = (D & 3) == 0;                  // head columns 16-byte aligned: 128-bit LDS reads in the dot products
  float *rows = lds, *P = lds + N * RP;
  const Div dN(N), dRow(3 * E >> 2), dD(D), dE(E);
  for (int s = blockIdx.x; s < a.S; s += gridDim.x) {
    copy_rows(rows, RP, a.qkv + (size_t)s * N * 3 * E, N, 3 * E, dRow);
    __syncthreads();
    if (vec) {
      head_dots(P, rows, RP, rows + E, RP, a.scale, N, Hh, D, dN);
    } else {                                      // (odd head widths: scalar path)
      for (int h = 0; h < Hh; ++h)
        for (int r = threadIdx.x; r < NN; r += blockDim.x) {
          const int i = dN.quot(r), j = r - i * N;
          float acc = 0.f;
          for (int d = 0; d < D; ++d) acc = fmaf(rows[i * RP + h * D + d], rows[j * RP + E + h * D + d], acc);
          P[h * NN + r] = acc * a.scale;
        }
    }
    __syncthreads();
    softmax_rows(P, Hh * N, N);
    __syncthreads();
    // out[i][e] = sum_j P[h(e)][i][j] v[j][e], one output per thread and round (lanes walk over e: conflict-free v reads,
    // broadcast P reads)
    float *o = a.out + (size_t)s * N * E;
    if (E >= (int)blockDim.x && N <= 16) {
      for (int e = threadIdx.x; e < E; e += blockDim.x)
        channel_contract<16>(P + dD.quot(e) * NN, N, 1, rows + 2 * E + e, RP, o + e, E, N);
    } else
    for (int idx = threadIdx.x; idx < N * E; idx += blockDim.x) {
      const int i = dE.quot(idx), e = idx - i * E, h = dD.quot(e);
      const float *v = rows + 2 * E + e, *p = P + h * NN + i * N;
      float a0 = 0.f, a1 = 0.f;
      int j = 0;
      for (; j + 1 < N; j += 2) a0 = fmaf(p[j], v[j * RP], a0), a1 = fmaf(p[j + 1], v[(j + 1) * RP], a1);
      if (j < N) a0 = fmaf(p[j], v[j * RP], a0);
      o[idx] = a0 + a1;
    }
    __syncthreads();                               // the image is rewritten by the next sequence
  }
}

__global__ __launch_bounds__(256) void attn_bwd_kernel(const Args a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int N = a.N, Hh = a.Hh, D = a.D, E = Hh * D, RP = 3 * E + 4, GP = E + 4, NN = N * N;
  const bool vec = (D & 3) == 0;
  float *rows = lds, *dO = rows + N * RP, *P = dO + N * GP, *dS = P + Hh * NN;
  const Div dN(N), dRow(3 * E >> 2), dG(E >> 2), dD(D), d3E(3 * E);
  for (int s = blockIdx.x; s < a.S; s += gridDim.x) {
    copy_rows(rows, RP, a.qkv + (size_t)s * N * 3 * E, N, 3 * E, dRow);
    copy_rows(dO, GP, a.g_out + (size_t)s * N * E, N, E, dG);
    __syncthreads();
    if (vec) {
      head_dots(P, rows, RP, rows + E, RP, a.scale, N, Hh, D, dN);            // scores
      head_dots(dS, dO, GP, rows + 2 * E, RP, 1.f, N, Hh, D, dN);              // dP[h][i][j] = <dO_i, v_j>_h
    } else {
      for (int h = 0; h < Hh; ++h)
        for (int r = threadIdx.x; r < NN; r += blockDim.x) {
          const int i = dN.quot(r), j = r - i * N;
          float sc = 0.f, dp = 0.f;
          for (int d = 0; d < D; ++d) {
            sc = fmaf(rows[i * RP + h * D + d], rows[j * RP + E + h * D + d], sc);
            dp = fmaf(dO[i * GP + h * D + d], rows[j * RP + 2 * E + h * D + d], dp);
          }
          P[h * NN + r] = sc * a.scale, dS[h * NN + r] = dp;
        }
    }
    __syncthreads();
    softmax_rows(P, Hh * N, N);
    __syncthreads();
    ds_rows(dS, P, Hh * N, N, a.scale);            // dS = P * (dP - sum_j dP P), times the score scale
    __syncthreads();
    // one gradient element per thread and round (lanes walk over the 3E channels c of token n):
    //   dQ[n][e] = sum_j dS[n][j] k[j][e];  dK[n][e] = sum_i dS[i][n] q[i][e];  dV[n][e] = sum_i P[i][n] dO[i][e]
    float *g = a.g_qkv + (size_t)s * N * 3 * E;
    if (E >= (int)blockDim.x && N <= 16) {
      for (int c = threadIdx.x; c < 3 * E; c += blockDim.x) {
        const int which = (c >= 2 * E) ? 2 : (c >= E ? 1 : 0), e = c - which * E, h = dD.quot(e);
        const float *m = (which == 2 ? P : dS) + h * NN;
        const float *x = (which == 0) ? rows + E + e : (which == 1 ? rows + e : dO + e);
        channel_contract<16>(m, which == 0 ? N : 1, which == 0 ? 1 : N, x, which == 2 ? GP : RP, g + c, 3 * E, N);
      }
    } else
    for (int idx = threadIdx.x; idx < N * 3 * E; idx += blockDim.x) {
      const int n = d3E.quot(idx), c = idx - n * 3 * E;
      const int which = (c >= 2 * E) ? 2 : (c >= E ? 1 : 0), e = c - which * E, h = dD.quot(e);
      const float *m = (which == 2 ? P : dS) + h * NN + (which == 0 ? n * N : n);
      const float *x = (which == 0) ? rows + E + e : (which == 1 ? rows + e : dO + e);
      const int xp = (which == 2) ? GP : RP, sk = (which == 0) ? 1 : N;
      float a0 = 0.f, a1 = 0.f;
      int k = 0;
      for (; k + 1 < N; k += 2) a0 = fmaf(m[k * sk], x[k * xp], a0), a1 = fmaf(m[(k + 1) * sk], x[(k + 1) * xp], a1);
      if (k < N) a0 = fmaf(m[k * sk], x[k * xp], a0);
      g[idx] = a0 + a1;
    }
    __syncthreads();
  }
}

static int check(const Args &a, bool bwd, size_t *lds) {
  if (a.S < 0 || a.N < 1 || a.N > 64 || a.Hh < 1 || a.D < 1 || ((a.Hh * a.D) & 3)) return P2C_E_SHAPE;
  const size_t E = (size_t)a.Hh * a.D, NN = (size_t)a.N * a.N;
  *lds = sizeof(float) * (bwd ? a.N * (4 * E + 8) + 2 * a.Hh * NN : a.N * (3 * E + 4) + a.Hh * NN);
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
