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

// ---- folded input map (K7a'): embeddings composed with the encoder's first input projection -------------------------------
// Seq2SeqEmbeddings with fold_embeddings: nothing non-linear sits between the per-joint Linear(C, E) and W_ih0 of the
// encoder LSTM, so the train step runs the encoder on the raw keypoints through
//     w_eff[g, j, c] = sum_e W_ih0[g, jE + e] W_j[e, c]          b_eff[g] = b_ih0[g] + b_hh0[g] + sum_{j,e} W_ih0[g, jE + e] b_j[e]
// As framework ops that is 2 broadcast products + 2 reductions + 2 adds forward and 11 launches backward, each 4-15 us
// for 0.85 MFLOP. Here: one launch each way, fixed summation orders (bitwise reproducible).
namespace p2c_fold {

constexpr int MAXC = 4;

struct Args {
  const float *w_ih;            // (G, J*E)
  const float *W, *b;           // joint j: W + j * w_stride (E, C), b + j * b_stride (E)
  const float *b_ih, *b_hh;     // (G) or NULL
  float *w_eff, *b_eff;         // (G, J*C), (G)
  const float *g_eff, *g_b;     // gradients of the two outputs
  float *g_w_ih;                // (G, J*E): written, or added to when acc_w
  float *gW, *gb;               // strides of W / b: ADDED to
  float *g_b_ih, *g_b_hh;       // (G) or NULL: ADDED to
  int64_t w_stride, b_stride;
  int32_t G, J, E, C, acc_w;
};

// one workgroup per gate row g: its J*E weights pass through LDS (coalesced), thread (j, c) forms one output
__global__ __launch_bounds__(256) void fold_fwd_kernel(const Args a) {
  extern __shared__ float row[];                  // [J*E] + [J] partial bias sums
  const int g = blockIdx.x, JE = a.J * a.E;
  for (int i = threadIdx.x; i < JE; i += blockDim.x) row[i] = a.w_ih[(size_t)g * JE + i];
  __syncthreads();
  float *pb = row + JE;
  // threads [0, J*C): one w_eff element; threads [J*C, J*C + J): one joint's share of the bias sum (all waves busy at once)
  for (int i = threadIdx.x; i < a.J * a.C + a.J; i += blockDim.x) {
    if (i < a.J * a.C) {
      const int j = i / a.C, c = i - j * a.C;
      const float *__restrict__ w = a.W + j * a.w_stride + c;
      const float *r = row + j * a.E;
      float s0 = 0.f, s1 = 0.f;                   // two chains: the loads of W_j are independent of the sums
      int e = 0;
      for (; e + 1 < a.E; e += 2) s0 = fmaf(r[e], w[(size_t)e * a.C], s0), s1 = fmaf(r[e + 1], w[(size_t)(e + 1) * a.C], s1);
      if (e < a.E) s0 = fmaf(r[e], w[(size_t)e * a.C], s0);
      a.w_eff[(size_t)g * a.J * a.C + i] = s0 + s1;
    } else {
      const int j = i - a.J * a.C;
      const float *__restrict__ bj = a.b + j * a.b_stride;
      const float *r = row + j * a.E;
      float s0 = 0.f, s1 = 0.f;
      int e = 0;
      for (; e + 1 < a.E; e += 2) s0 = fmaf(r[e], bj[e], s0), s1 = fmaf(r[e + 1], bj[e + 1], s1);
      if (e < a.E) s0 = fmaf(r[e], bj[e], s0);
      pb[j] = s0 + s1;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int j = 0; j < a.J; ++j) s += pb[j];
    if (a.b_ih) s += a.b_ih[g];
    if (a.b_hh) s += a.b_hh[g];
    a.b_eff[g] = s;
  }
}

// one workgroup per joint j, 1024 threads = (channel e, one of 16 groups of gate rows): the joint's slice of d W_ih0 is
// written row by row while the sums over g for d W_j / d b_j run in registers; the 16 groups meet in LDS in a fixed order.
// The gradients of the two outputs for this joint sit in LDS, and the loop is unrolled with its loads in front: 16 rows per
// thread is two round trips, where 64 rows with a load-use-store body each were 64 (57 us measured at G = 256).
constexpr int GG = 16;                             // gate-row groups per workgroup
__global__ __launch_bounds__(64 * GG) void fold_bwd_kernel(const Args a) {
  extern __shared__ float lds[];                   // [G][C + 1]: g_eff[g, j, :], g_b[g]; then [GG][64][C + 1] partial sums
  const int j = blockIdx.x, JE = a.J * a.E, JC = a.J * a.C, C1 = a.C + 1;
  float *ge = lds, *part = lds + a.G * C1;
  for (int i = threadIdx.x; i < a.G * C1; i += blockDim.x) {
    const int g = i / C1, c = i - g * C1;
    ge[i] = (c < a.C) ? a.g_eff[(size_t)g * JC + j * a.C + c] : a.g_b[g];
  }
  __syncthreads();
  const int grp = threadIdx.x >> 6, l = threadIdx.x & 63;
  const int per = (a.G + GG - 1) / GG, g0 = grp * per, g1 = (g0 + per < a.G) ? g0 + per : a.G;
  for (int e0 = 0; e0 < a.E; e0 += 64) {
    const int e = e0 + l;
    const bool ok = e < a.E;
    float wj[MAXC], acc[MAXC + 1];
#pragma unroll
    for (int c = 0; c < MAXC; ++c) wj[c] = (ok && c < a.C) ? a.W[j * a.w_stride + (size_t)e * a.C + c] : 0.f, acc[c] = 0.f;
    acc[MAXC] = 0.f;
    const float bj = ok ? a.b[j * a.b_stride + e] : 0.f;
    constexpr int U = 8;
    for (int gb = g0; gb < g1 && ok; gb += U) {
      float w[U], old[U];
#pragma unroll
      for (int r = 0; r < U; ++r) {
        const bool in = gb + r < g1;
        const size_t iw = (size_t)(gb + r) * JE + j * a.E + e;
        w[r] = in ? a.w_ih[iw] : 0.f;
        old[r] = (in && a.acc_w) ? a.g_w_ih[iw] : 0.f;
      }
#pragma unroll
      for (int r = 0; r < U; ++r) {
        if (gb + r >= g1) continue;
        const float *gr = ge + (gb + r) * C1;
        const float gbv = gr[a.C];
        float s = gbv * bj;
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
          if (c < a.C) s = fmaf(gr[c], wj[c], s), acc[c] = fmaf(gr[c], w[r], acc[c]);
        acc[MAXC] = fmaf(gbv, w[r], acc[MAXC]);
        a.g_w_ih[(size_t)(gb + r) * JE + j * a.E + e] = old[r] + s;
      }
    }
    __syncthreads();                               // (previous round's readers are done)
#pragma unroll
    for (int c = 0; c <= MAXC; ++c) part[(grp * 64 + l) * (MAXC + 1) + c] = acc[c];
    __syncthreads();
    if (grp == 0 && ok) {
      float tot[MAXC + 1];
#pragma unroll
      for (int c = 0; c <= MAXC; ++c) tot[c] = 0.f;
      for (int q = 0; q < GG; ++q)
#pragma unroll
        for (int c = 0; c <= MAXC; ++c) tot[c] += part[(q * 64 + l) * (MAXC + 1) + c];
#pragma unroll
      for (int c = 0; c < MAXC; ++c)
        if (c < a.C) a.gW[j * a.w_stride + (size_t)e * a.C + c] += tot[c];
      a.gb[j * a.b_stride + e] += tot[MAXC];
    }
  }
  if (j == 0)
    for (int g = threadIdx.x; g < a.G; g += blockDim.x) {
      if (a.g_b_ih) a.g_b_ih[g] += ge[g * C1 + a.C];
      if (a.g_b_hh) a.g_b_hh[g] += ge[g * C1 + a.C];
    }
}

static int check(int32_t G, int32_t J, int32_t E, int32_t C) {
  if (G < 0 || J < 1 || E < 1 || C < 1 || C > MAXC) return P2C_E_SHAPE;
  if ((size_t)(J * E + J) * sizeof(float) > 60 * 1024) return P2C_E_SHAPE;
  return 0;
}

}  // namespace p2c_fold

extern "C" int p2c_fold_fwd(const float *w_ih, const float *W, const float *b, int64_t w_stride, int64_t b_stride,
                            const float *b_ih, const float *b_hh, float *w_eff, float *b_eff, int32_t G, int32_t J, int32_t E,
                            int32_t C, void *stream) {
  if (!w_ih || !W || !b || !w_eff || !b_eff) return P2C_E_NULL;
  int rc = p2c_fold::check(G, J, E, C);
  if (rc) return rc;
  if (G == 0) return 0;
  p2c_fold::Args a{};
  a.w_ih = w_ih, a.W = W, a.b = b, a.b_ih = b_ih, a.b_hh = b_hh, a.w_eff = w_eff, a.b_eff = b_eff;
  a.w_stride = w_stride, a.b_stride = b_stride, a.G = G, a.J = J, a.E = E, a.C = C;
  hipLaunchKernelGGL(p2c_fold::fold_fwd_kernel, dim3((unsigned)G), dim3(256), (size_t)(J * E + J) * sizeof(float), (hipStream_t)stream, a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

extern "C" int p2c_fold_bwd(const float *w_ih, const float *W, const float *b, int64_t w_stride, int64_t b_stride,
                            const float *g_eff, const float *g_b, float *g_w_ih, int32_t accumulate_w, float *gW, float *gb,
                            float *g_b_ih, float *g_b_hh, int32_t G, int32_t J, int32_t E, int32_t C, void *stream) {
  if (!w_ih || !W || !b || !g_eff || !g_b || !g_w_ih || !gW || !gb) return P2C_E_NULL;
  int rc = p2c_fold::check(G, J, E, C);
  if (rc) return rc;
  p2c_fold::Args a{};
  a.w_ih = w_ih, a.W = W, a.b = b, a.g_eff = g_eff, a.g_b = g_b, a.g_w_ih = g_w_ih, a.acc_w = accumulate_w, a.gW = gW, a.gb = gb;
  a.g_b_ih = g_b_ih, a.g_b_hh = g_b_hh, a.w_stride = w_stride, a.b_stride = b_stride, a.G = G, a.J = J, a.E = E, a.C = C;
  const size_t lds = ((size_t)G * (C + 1) + (size_t)p2c_fold::GG * 64 * (p2c_fold::MAXC + 1)) * sizeof(float);
  if (lds > 60 * 1024) return P2C_E_SHAPE;
  hipLaunchKernelGGL(p2c_fold::fold_bwd_kernel, dim3((unsigned)J), dim3(64 * p2c_fold::GG), lds, (hipStream_t)stream, a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}
