// p2c_embed.hip -- K7a: the per-joint input embeddings of Seq2SeqEmbeddings as ONE grouped launch (gfx950).
//
// Reference: modules/movements/seq2seq/seq2seq_embeddings.py:53-78 -- a Python loop over 26 nn.Linear(2, 64), each writing
// its 64-wide slice of a (T, B, 26*64) tensor (26 tiny GEMMs with K = 2, 26 slice copies, and as many again backward).
// Here: one streaming kernel forward (reads 8 B, writes 256 B per joint-frame: HBM-bound on the 6.6 KB/frame it writes,
// directly in the sequence-first (t, b) row order the LSTM consumes, optionally time-reversed) and a deterministic
// two-stage reduction backward (grouped dW_j = sum_n gy_j^T x_j, db_j = sum_n gy_j). K = 2 contractions are VALU work;
// the dense LSTM GEMMs behind them stay library GEMMs (MIOpen / rocBLAS fp32 MFMA kernels).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/p2c.h"

namespace p2c_embed {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int MAXC = 4;

struct Args {
  const float *x;        // (B, T, J, C)
  const float *W;        // joint j: W + j * w_stride, (E, C) row-major
  const float *b;        // joint j: b + j * b_stride, (E)
  float *y;              // (T, B, J, E), row = (flip ? T-1-t : t) * B + b
  const float *gy;       // same layout as y
  float *gW, *gb;        // same strides as W / b
  float *partials;       // (n_chunks, J, E, C + 1)
  int64_t w_stride, b_stride;
  int32_t B, T, J, C, E, flip, n_chunks, rows_per_chunk;
};

__device__ __forceinline__ int64_t out_row(const Args &a, int64_t in_row) {
  const int64_t bb = in_row / a.T;
  const int t = (int)(in_row - bb * a.T);
  return (int64_t)(a.flip ? a.T - 1 - t : t) * a.B + bb;
}

// one thread = four consecutive embedding channels of one joint, for ROWS consecutive frames: the 4 x C weights and the
// bias are read once per thread and stay in registers; per frame it reads C floats and stores 16 bytes
constexpr int ROWS = 8;
__global__ __launch_bounds__(256) void embed_fwd_kernel(const Args a) {
  const int e4n = a.E >> 2;
  const int64_t N = (int64_t)a.B * a.T;
  const int64_t nblk = (N + ROWS - 1) / ROWS;
  const int64_t total = nblk * a.J * e4n;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int e = (int)(i % e4n) * 4;
    const int64_t bj = i / e4n;
    const int j = (int)(bj % a.J);
    const int64_t n0 = (bj / a.J) * ROWS;
    const float *w = a.W + j * a.w_stride + (int64_t)e * a.C;
    const float *bp = a.b + j * a.b_stride + e;
    float wr[4][MAXC];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int c = 0; c < MAXC; ++c) wr[k][c] = (c < a.C) ? w[k * a.C + c] : 0.f;
    const f32x4 bias = {bp[0], bp[1], bp[2], bp[3]};
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      const int64_t n = n0 + r;
      if (n < N) {
        const float *xp = a.x + (n * a.J + j) * a.C;
        f32x4 acc = bias;
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
          if (c < a.C) {
            const float xv = xp[c];
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] = fmaf(wr[k][c], xv, acc[k]);
          }
        *reinterpret_cast<f32x4 *>(a.y + (out_row(a, n) * a.J + j) * a.E + e) = acc;
      }
    }
  }
}

// stage 1: workgroup (j, chunk); thread (e = tid % 64 [+ 64 k], q = tid / 64) adds rows q, q+4, ... of the chunk
__global__ __launch_bounds__(256) void embed_bwd_partial_kernel(const Args a) {
  __shared__ float red[4][64][MAXC + 1];
  const int j = blockIdx.x, chunk = blockIdx.y;
  const int el = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int64_t N = (int64_t)a.B * a.T;
  const int64_t r0 = (int64_t)chunk * a.rows_per_chunk;
  const int64_t r1 = (r0 + a.rows_per_chunk < N) ? r0 + a.rows_per_chunk : N;
  for (int e0 = 0; e0 < a.E; e0 += 64) {
    const int e = e0 + el;
    float acc[MAXC + 1];
#pragma unroll
    for (int c = 0; c <= MAXC; ++c) acc[c] = 0.f;
    if (e < a.E) {
      for (int64_t n = r0 + q; n < r1; n += 4) {
        const float g = a.gy[(out_row(a, n) * a.J + j) * a.E + e];
        const float *xp = a.x + (n * a.J + j) * a.C;
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
          if (c < a.C) acc[c] = fmaf(g, xp[c], acc[c]);
        acc[MAXC] += g;
      }
    }
#pragma unroll
    for (int c = 0; c <= MAXC; ++c) red[q][el][c] = acc[c];
    __syncthreads();
    if (q == 0 && e < a.E) {
      float *p = a.partials + (((size_t)chunk * a.J + j) * a.E + e) * (a.C + 1);
      for (int c = 0; c < a.C; ++c) p[c] = ((red[0][el][c] + red[1][el][c]) + red[2][el][c]) + red[3][el][c];
      p[a.C] = ((red[0][el][MAXC] + red[1][el][MAXC]) + red[2][el][MAXC]) + red[3][el][MAXC];
    }
    __syncthreads();
  }
}

// stage 2: fixed-order sum over the chunks (bitwise reproducible), scattered to the weight / bias gradients
__global__ __launch_bounds__(256) void embed_bwd_reduce_kernel(const Args a) {
  const int per = a.C + 1;
  const int total = a.J * a.E * per;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  float s = 0.f;
  int ch = 0;
  for (; ch + 8 <= a.n_chunks; ch += 8) {        // eight loads in flight, added in chunk order
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = a.partials[(size_t)(ch + u) * total + i];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; ch < a.n_chunks; ++ch) s += a.partials[(size_t)ch * total + i];
  const int c = i % per, je = i / per, e = je % a.E, j = je / a.E;
  if (c < a.C) a.gW[j * a.w_stride + (int64_t)e * a.C + c] = s;
  else a.gb[j * a.b_stride + e] = s;
}

}  // namespace p2c_embed

using namespace p2c_embed;

static int chunks_for(int64_t N) {
  int64_t c = (N + 63) / 64;
  return (int)(c < 1 ? 1 : (c > 128 ? 128 : c));
}

static int fill(Args &a, const float *x, const float *W, const float *b, int64_t w_stride, int64_t b_stride, int32_t B,
                int32_t T, int32_t J, int32_t C, int32_t E, int32_t flip) {
  if (!x || !W || !b) return P2C_E_NULL;
  if (B < 0 || T < 1 || J < 1 || C < 1 || C > MAXC || E < 4 || (E & 3)) return P2C_E_SHAPE;
  a = Args{};
  a.x = x, a.W = W, a.b = b, a.w_stride = w_stride, a.b_stride = b_stride;
  a.B = B, a.T = T, a.J = J, a.C = C, a.E = E, a.flip = flip ? 1 : 0;
  const int64_t N = (int64_t)B * T;
  a.n_chunks = chunks_for(N);
  a.rows_per_chunk = (int32_t)((N + a.n_chunks - 1) / a.n_chunks);
  return 0;
}

extern "C" int64_t p2c_embed_workspace_floats(int32_t B, int32_t T, int32_t J, int32_t C, int32_t E) {
  if (B <= 0 || T <= 0) return 0;
  return (int64_t)chunks_for((int64_t)B * T) * J * E * (C + 1);
}

extern "C" int p2c_embed_fwd(const float *x, const float *W, const float *b, int64_t w_stride, int64_t b_stride, float *y,
                             int32_t B, int32_t T, int32_t J, int32_t C, int32_t E, int32_t flip, void *stream) {
  Args a;
  int rc = fill(a, x, W, b, w_stride, b_stride, B, T, J, C, E, flip);
  if (rc) return rc;
  if (!y) return P2C_E_NULL;
  if ((reinterpret_cast<uintptr_t>(y) & 15) != 0) return P2C_E_SHAPE;
  if (B == 0) return 0;
  a.y = y;
  const int64_t total = (((int64_t)B * T + ROWS - 1) / ROWS) * J * (E >> 2);
  int64_t blocks = (total + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(embed_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

extern "C" int p2c_embed_bwd(const float *x, const float *gy, int64_t w_stride, int64_t b_stride, float *gW, float *gb,
                             float *partials, int32_t B, int32_t T, int32_t J, int32_t C, int32_t E, int32_t flip,
                             void *stream) {
  Args a;
  int rc = fill(a, x, gy, gy, w_stride, b_stride, B, T, J, C, E, flip);   // W / b are not read by the backward
  if (rc) return rc;
  if (!gy || !gW || !gb || !partials) return P2C_E_NULL;
  a.gy = gy, a.gW = gW, a.gb = gb, a.partials = partials;
  hipLaunchKernelGGL(embed_bwd_partial_kernel, dim3((unsigned)J, (unsigned)a.n_chunks), dim3(256), 0, (hipStream_t)stream, a);
  const int total = J * E * (C + 1);
  hipLaunchKernelGGL(embed_bwd_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}
