// p2c_attn.hip -- K14: multi-head self-attention over SHORT token sequences, one launch forward, one backward (gfx950).
//
// PoseFormer (reference modules/movements/pose_former/pose_former.py:33-76 binds third_party PoseTransformer; the build's own
// restatement is modules/movements/pose_former/pose_transformer.py) attends over 26 joint tokens of width 32 (8 heads of FOUR
// channels) in its spatial blocks and over 9 frame tokens of width 832 (8 heads of 104) in its temporal blocks: at cfg5 21 024
// and 2 336 sequences per block and step. The framework's fused attention is built for long sequences and head widths of
// 32-256: on these shapes it took 0.35 ms forward and 1.1 ms backward per block (four launches) for 7-60 MFLOP. Here a
// workgroup owns one sequence: its q, k, v rows (and the output gradient) sit in LDS, the N x N score matrices of all heads
// too. Two families of phases:
//   * narrow heads (D = 4 or 8: the spatial blocks): nothing for a 16-wide MFMA tile to chew on (K = 4, 26 x 26 scores) --
//     a thread owns a (head, token) row, keeps its own D channels in registers and walks over the other tokens with one
//     16-byte broadcast LDS read per step (~8 instructions per score, 6 per four gradient products);
//   * wide heads (D > 8, N <= 16: the temporal blocks, 9 tokens x 104): every product runs on the matrix cores
//     (v_mfma_f32_16x16x4_f32, exact fp32): one 16 x 16 tile holds all token pairs of a head, K = D for scores / dP, K = N
//     over D / 16 column tiles for outputs and gradients;
//   softmax (and the dS row pass of the backward) are register-cached row passes. Loop bodies are written "all loads, then
//   all arithmetic": a 90-120 KB sequence image leaves room for four waves on the CU, so exposed LDS round trips are the cost.
//   qkv (S, N, 3, Hh, D) = the qkv Linear's output viewed; out (S, N, Hh*D);  P = softmax(scale * q k^T) per head.
// Backward recomputes P from q, k (no saved probabilities):
//   dV = P^T dO;  dP = dO V^T;  dS = P * (dP - rowsum(dP * P));  dQ = scale dS K;  dK = scale dS^T Q.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/p2c.h"

namespace p2c_attn {

typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifdef P2C_ATTN_TRACE   // developer build only (tools/attntrace.py): shader-clock stamps of workgroup 0's first sequence
static __device__ unsigned long long g_attn_trace[16];
#define AT(i)                                                                              \
  do {                                                                                     \
    if (blockIdx.x == 0 && threadIdx.x == 0 && s == (int)blockIdx.x) g_attn_trace[i] = __builtin_readcyclecounter(); \
  } while (0)
#else
#define AT(i)
#endif

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

// Padding beyond N is applied ARITHMETICALLY (value * 0/1 mask [+ pad]) on values loaded through clamped indices: a select
// between a loaded value and a constant is turned back into a guarded load by the compiler -- one exec-masked block and one
// exposed LDS round trip per element (measured: 4.8k cycles for a group of 32 reads + 16 MFMAs). Everything in LDS is finite.

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

// S[h][i][j] = <a_i, b_j>_h for all heads: a rows at `ar` (pitch ap, head h at column h D), b rows at `br` (pitch bp).
// The loop bodies are written "all loads, then all FMAs": with one load-use pair per iteration a wave pays one LDS round
// trip (~100+ cycles) per pair, and a sequence image of 90-120 KB leaves room for only 4 waves on the CU to hide it.
__device__ __forceinline__ void head_dots(float *S, const float *ar, int ap, const float *br, int bp, float scale, int N, int Hh,
                                          int D, const Div &dN) {
  const int NN = N * N;
  if (D == 4 || D == 8) {     // narrow heads: a thread owns row (h, i) -- its a_i stays in registers and the b rows are walked
    for (int row = threadIdx.x; row < Hh * N; row += blockDim.x) {      // with one add per step: ~8 instructions per score
      const int h = dN.quot(row), i = row - h * N;
      const float *a = ar + i * ap + h * D, *b = br + h * D;
      const f32x4 a0 = *reinterpret_cast<const f32x4 *>(a);
      const f32x4 a1 = (D == 8) ? *reinterpret_cast<const f32x4 *>(a + 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
      float *out = S + row * N;
      int j = 0;
      for (; j + 3 < N; j += 4) {
        f32x4 bv[4], bw[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          bv[u] = *reinterpret_cast<const f32x4 *>(b + (j + u) * bp);
          if (D == 8) bw[u] = *reinterpret_cast<const f32x4 *>(b + (j + u) * bp + 4);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          float s0 = a0[0] * bv[u][0];
          s0 = fmaf(a0[1], bv[u][1], s0), s0 = fmaf(a0[2], bv[u][2], s0), s0 = fmaf(a0[3], bv[u][3], s0);
          if (D == 8) s0 = fmaf(a1[0], bw[u][0], s0), s0 = fmaf(a1[1], bw[u][1], s0), s0 = fmaf(a1[2], bw[u][2], s0), s0 = fmaf(a1[3], bw[u][3], s0);
          out[j + u] = s0 * scale;
        }
      }
      for (; j < N; ++j) {
        const f32x4 bv = *reinterpret_cast<const f32x4 *>(b + j * bp);
        float s0 = a0[0] * bv[0];
        s0 = fmaf(a0[1], bv[1], s0), s0 = fmaf(a0[2], bv[2], s0), s0 = fmaf(a0[3], bv[3], s0);
        if (D == 8) {
          const f32x4 bw = *reinterpret_cast<const f32x4 *>(b + j * bp + 4);
          s0 = fmaf(a1[0], bw[0], s0), s0 = fmaf(a1[1], bw[1], s0), s0 = fmaf(a1[2], bw[2], s0), s0 = fmaf(a1[3], bw[3], s0);
        }
        out[j] = s0 * scale;
      }
    }
    return;
  }
  for (int h = 0; h < Hh; ++h)
    for (int r = threadIdx.x; r < NN; r += blockDim.x) {
      const int i = dN.quot(r), j = r - i * N;
      const float *a = ar + i * ap + h * D, *b = br + j * bp + h * D;
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
      int d = 0;
      for (; d + 15 < D; d += 16) {               // 8 x 16-byte loads in flight, then 16 FMAs on four chains
        f32x4 av[4], bv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) av[u] = *reinterpret_cast<const f32x4 *>(a + d + 4 * u), bv[u] = *reinterpret_cast<const f32x4 *>(b + d + 4 * u);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          s0 = fmaf(av[u][0], bv[u][0], s0), s1 = fmaf(av[u][1], bv[u][1], s1);
          s2 = fmaf(av[u][2], bv[u][2], s2), s3 = fmaf(av[u][3], bv[u][3], s3);
        }
      }
      for (; d + 3 < D; d += 4) {
        const f32x4 av = *reinterpret_cast<const f32x4 *>(a + d), bv = *reinterpret_cast<const f32x4 *>(b + d);
        s0 = fmaf(av[0], bv[0], s0), s1 = fmaf(av[1], bv[1], s1), s2 = fmaf(av[2], bv[2], s2), s3 = fmaf(av[3], bv[3], s3);
      }
      S[h * NN + r] = ((s0 + s1) + (s2 + s3)) * scale;
    }
}

// ---- wide heads on the matrix cores (N <= 16 tokens, head width a multiple of 4, > 8) ----------------------------------------
// One 16 x 16 tile (v_mfma_f32_16x16x4_f32, exact fp32) covers all token pairs of a head: scores and dP are K = D products
// (26 k-steps for the 104-wide temporal heads), the outputs / gradients are K = N products over D / 16 column tiles. Lane
// (r = lane & 15, g = lane >> 4) feeds A[r][4 ks + g] and B[4 ks + g][r] and receives D[4 g .. 4 g + 3][r]; with the padded row
// pitches (2 500 / 836 floats = 4 banks mod 64) the 64 scalar LDS reads of an operand fall into 64 different banks.
// S[h][i][j] = scale * <a_i, b_j>_h for i, j < N; heads are dealt to the waves
__device__ __forceinline__ void head_dots_mfma(float *S, const float *ar, int ap, const float *br, int bp, float scale, int N, int Hh,
                                               int D) {
  const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
  const int rc = r < N ? r : N - 1;               // (clamped row: its products land in tile rows / columns >= N, never stored)
  for (int h = wave; h < Hh; h += waves) {
    const float *a = ar + rc * ap + h * D + g, *b = br + rc * bp + h * D + g;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
    int ks = 0;
    for (; ks + 7 < (D >> 2); ks += 8) {          // 16 operand reads in flight, two accumulator chains
      float av[8], bv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) av[u] = a[4 * (ks + u)], bv[u] = b[4 * (ks + u)];
#pragma unroll
      for (int u = 0; u < 8; u += 2) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u + 1], bv[u + 1], acc1, 0, 0, 0);
      }
    }
    for (; ks < (D >> 2); ++ks) acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * ks], b[4 * ks], acc0, 0, 0, 0);
    const f32x4 acc = acc0 + acc1;
    if (r < N) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (4 * g + q < N) S[(h * N + 4 * g + q) * N + r] = acc[q] * scale;
    }
  }
}
// y[n][col] = sum_k M(n, k) * x_k[col] for every head h, n < N, col < D, with M(n, k) = m[h N N + n * sn + k * sk].
// Heads are dealt to the waves; a wave reads the head's M operand once (four k-steps, zero beyond N) and runs up to eight
// 16-column tiles against it: all B reads of the group in flight, then the MFMAs on independent accumulators, then stores.
__device__ __forceinline__ void contract_mfma(const float *m, int sn, int sk, const float *x, int xp, float *y, int yp, int N, int Hh,
                                              int D) {
  const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
  const int NN = N * N, ctiles = (D + 15) >> 4, rc = r < N ? r : N - 1;
  constexpr int CT = 8;
  int kc[4];
  float km[4], am[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const bool kok = 4 * ks + g < N;
    kc[ks] = kok ? 4 * ks + g : N - 1, km[ks] = kok ? 1.f : 0.f, am[ks] = (kok && r < N) ? 1.f : 0.f;
  }
  for (int h = wave; h < Hh; h += waves) {
    const float *mh = m + h * NN + rc * sn;
    float av[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) av[ks] = mh[kc[ks] * sk] * am[ks];
    for (int c0 = 0; c0 < ctiles; c0 += CT) {
      float bv[CT][4];
#pragma unroll
      for (int u = 0; u < CT; ++u) {
        const int col = (c0 + u) * 16 + r;
        const float cm = col < D ? 1.f : 0.f;
        const float *xh = x + h * D + (col < D ? col : 0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) bv[u][ks] = xh[kc[ks] * xp] * (km[ks] * cm);
      }
      f32x4 acc[CT];
#pragma unroll
      for (int u = 0; u < CT; ++u) acc[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int u = 0; u < CT; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks], bv[u][ks], acc[u], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < CT; ++u) {
        const int col = (c0 + u) * 16 + r;
        if (col < D) {
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (4 * g + q < N) y[(4 * g + q) * yp + h * D + col] = acc[u][q];
        }
      }
    }
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
    for (int j = 0; j < NB; ++j) {                // (clamped index, not a guarded load: no branch between the loads)
      v[j] = fmaf(p[j < N ? j : N - 1], (j < N) ? 1.f : 0.f, (j < N) ? 0.f : -3.0e38f);
    }
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
// both row passes of the backward in one: the thread that owns row (h, i) reads scores and dP once
template <int NB>
__device__ __forceinline__ void softmax_ds_rows_nb(float *P, float *dS, int rows, int N, float scale) {
  for (int row = threadIdx.x; row < rows; row += blockDim.x) {
    float *p = P + row * N, *ds = dS + row * N;
    float v[NB], dv[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int jc = j < N ? j : N - 1;
      const float mk = (j < N) ? 1.f : 0.f;
      v[j] = fmaf(p[jc], mk, (j < N) ? 0.f : -3.0e38f), dv[j] = ds[jc] * mk;
    }
    float m = v[0];
#pragma unroll
    for (int j = 1; j < NB; ++j) m = fmaxf(m, v[j]);
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < NB; ++j) v[j] = __expf(v[j] - m), sum += v[j];
    const float inv = 1.f / sum;
    float rs = 0.f;
#pragma unroll
    for (int j = 0; j < NB; ++j) v[j] *= inv, rs = fmaf(dv[j], v[j], rs);
#pragma unroll
    for (int j = 0; j < NB; ++j)
      if (j < N) p[j] = v[j], ds[j] = v[j] * (dv[j] - rs) * scale;
  }
}
// Narrow heads (D = 4 or 8): a thread owns all D channels of one (head, token) output: y[0..D) = sum_k m[k * sk] * x_k[0..D)
// with x_k one or two 16-byte LDS reads (wave-uniform per head: broadcast) and m a scalar read -- 6 instructions per four
// products where the one-output-per-thread loop needs 12, and one index split per D outputs.
template <int D>
__device__ __forceinline__ void vec_contract(const float *m, int sk, const float *x, int xp, float *y, int N) {
  f32x4 acc[D / 4];
#pragma unroll
  for (int v = 0; v < D / 4; ++v) acc[v] = (f32x4){0.f, 0.f, 0.f, 0.f};
  int k = 0;
  for (; k + 3 < N; k += 4) {
    float mv[4];
    f32x4 xv[4][D / 4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      mv[u] = m[(k + u) * sk];
#pragma unroll
      for (int v = 0; v < D / 4; ++v) xv[u][v] = *reinterpret_cast<const f32x4 *>(x + (k + u) * xp + 4 * v);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int v = 0; v < D / 4; ++v)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[v][c] = fmaf(mv[u], xv[u][v][c], acc[v][c]);
  }
  for (; k < N; ++k) {
    const float mv = m[k * sk];
#pragma unroll
    for (int v = 0; v < D / 4; ++v) {
      const f32x4 xv = *reinterpret_cast<const f32x4 *>(x + k * xp + 4 * v);
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[v][c] = fmaf(mv, xv[c], acc[v][c]);
    }
  }
#pragma unroll
  for (int v = 0; v < D / 4; ++v) *reinterpret_cast<f32x4 *>(y + 4 * v) = acc[v];
}

// One instantiation per family (each holds only its own path: inlined together the paths cost 256 VGPRs + 138 AGPRs and 300
// spilled SGPRs, and the scheduler interleaved nothing):
//   F_NARROW4 / F_NARROW8: D = 4 / 8;  F_WIDE: D % 4 == 0, D > 8, N <= 16 (matrix cores);  F_GENERIC: everything else.
enum Family { F_GENERIC = 0, F_NARROW4 = 1, F_NARROW8 = 2, F_WIDE = 3 };

template <int F, int NB>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const Args a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int N = a.N, Hh = a.Hh, D = (F == F_NARROW4) ? 4 : (F == F_NARROW8 ? 8 : a.D), E = Hh * D, RP = 3 * E + 4, NN = N * N;
  float *rows = lds, *P = lds + N * RP;
  const Div dN(N), dRow(3 * E >> 2), dD(D), dE(E);
  for (int s = blockIdx.x; s < a.S; s += gridDim.x) {
    copy_rows(rows, RP, a.qkv + (size_t)s * N * 3 * E, N, 3 * E, dRow);
    __syncthreads();
    if constexpr (F == F_WIDE) {
      head_dots_mfma(P, rows, RP, rows + E, RP, a.scale, N, Hh, D);
    } else if constexpr (F == F_NARROW4 || F == F_NARROW8) {
      head_dots(P, rows, RP, rows + E, RP, a.scale, N, Hh, D, dN);
    } else {
      if ((D & 3) == 0) {
        head_dots(P, rows, RP, rows + E, RP, a.scale, N, Hh, D, dN);
      } else {                                    // (odd head widths: scalar path)
        for (int h = 0; h < Hh; ++h)
          for (int r = threadIdx.x; r < NN; r += blockDim.x) {
            const int i = dN.quot(r), j = r - i * N;
            float acc = 0.f;
            for (int d = 0; d < D; ++d) acc = fmaf(rows[i * RP + h * D + d], rows[j * RP + E + h * D + d], acc);
            P[h * NN + r] = acc * a.scale;
          }
      }
    }
    __syncthreads();
    softmax_rows_nb<NB>(P, Hh * N, N);            // (NB = 16 / 32 / 64 >= N: one row width per instantiation)
    __syncthreads();
    float *o = a.out + (size_t)s * N * E;
    if constexpr (F == F_NARROW4 || F == F_NARROW8) {   // thread (h, i): the D channels of head h of query i
      for (int row = threadIdx.x; row < Hh * N; row += blockDim.x) {
        const int h = dN.quot(row), i = row - h * N;
        vec_contract<(F == F_NARROW4) ? 4 : 8>(P + row * N, 1, rows + 2 * E + h * D, RP, o + i * E + h * D, N);
      }
    } else if constexpr (F == F_WIDE) {
      contract_mfma(P, N, 1, rows + 2 * E, RP, o, E, N, Hh, D);
    } else {
      // out[i][e] = sum_j P[h(e)][i][j] v[j][e], one output per thread and round (lanes walk over e: conflict-free v reads,
      // broadcast P reads)
      for (int idx = threadIdx.x; idx < N * E; idx += blockDim.x) {
        const int i = dE.quot(idx), e = idx - i * E, h = dD.quot(e);
        const float *v = rows + 2 * E + e, *p = P + h * NN + i * N;
        float a0 = 0.f, a1 = 0.f;
        int j = 0;
        for (; j + 3 < N; j += 4) {
          const float p0 = p[j], p1 = p[j + 1], p2 = p[j + 2], p3 = p[j + 3];
          const float v0 = v[j * RP], v1 = v[(j + 1) * RP], v2 = v[(j + 2) * RP], v3 = v[(j + 3) * RP];
          a0 = fmaf(p0, v0, a0), a1 = fmaf(p1, v1, a1), a0 = fmaf(p2, v2, a0), a1 = fmaf(p3, v3, a1);
        }
        for (; j < N; ++j) a0 = fmaf(p[j], v[j * RP], a0);
        o[idx] = a0 + a1;
      }
    }
    __syncthreads();                               // the image is rewritten by the next sequence
  }
}

template <int F, int NB>
__global__ __launch_bounds__(256) void attn_bwd_kernel(const Args a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int N = a.N, Hh = a.Hh, D = (F == F_NARROW4) ? 4 : (F == F_NARROW8 ? 8 : a.D), E = Hh * D, RP = 3 * E + 4, GP = E + 4;
  const int NN = N * N;
  float *rows = lds, *dO = rows + N * RP, *P = dO + N * GP, *dS = P + Hh * NN;
  const Div dN(N), dRow(3 * E >> 2), dG(E >> 2), dD(D), d3E(3 * E);
  for (int s = blockIdx.x; s < a.S; s += gridDim.x) {
    AT(0);
    copy_rows(rows, RP, a.qkv + (size_t)s * N * 3 * E, N, 3 * E, dRow);
    copy_rows(dO, GP, a.g_out + (size_t)s * N * E, N, E, dG);
    __syncthreads();
    AT(1);
    if constexpr (F == F_WIDE) {
      head_dots_mfma(P, rows, RP, rows + E, RP, a.scale, N, Hh, D);           // scores
      head_dots_mfma(dS, dO, GP, rows + 2 * E, RP, 1.f, N, Hh, D);            // dP[h][i][j] = <dO_i, v_j>_h
    } else {
      if (F != F_GENERIC || (D & 3) == 0) {
        head_dots(P, rows, RP, rows + E, RP, a.scale, N, Hh, D, dN);          // scores
        head_dots(dS, dO, GP, rows + 2 * E, RP, 1.f, N, Hh, D, dN);            // dP
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
    }
    __syncthreads();
    AT(2);
    softmax_ds_rows_nb<NB>(P, dS, Hh * N, N, a.scale);    // P = softmax(scores); dS = P * (dP - sum_j dP P), times the score scale
    __syncthreads();
    AT(3);
    //   dQ[n][e] = sum_j dS[n][j] k[j][e];  dK[n][e] = sum_i dS[i][n] q[i][e];  dV[n][e] = sum_i P[i][n] dO[i][e]
    float *g = a.g_qkv + (size_t)s * N * 3 * E;
    if constexpr (F == F_NARROW4 || F == F_NARROW8) {   // thread (q | k | v, h, n): the D channels of that head and token
      const int per = Hh * N;
      for (int r = threadIdx.x; r < 3 * per; r += blockDim.x) {
        const int which = (r >= 2 * per) ? 2 : (r >= per ? 1 : 0), row = r - which * per, h = dN.quot(row), n = row - h * N;
        const float *m = (which == 2 ? P : dS) + h * NN + (which == 0 ? n * N : n);
        const float *x = (which == 0 ? rows + E : (which == 1 ? rows : dO)) + h * D;
        float *y = g + n * 3 * E + which * E + h * D;
        vec_contract<(F == F_NARROW4) ? 4 : 8>(m, which == 0 ? 1 : N, x, which == 2 ? GP : RP, y, N);
      }
    } else if constexpr (F == F_WIDE) {
      contract_mfma(dS, N, 1, rows + E, RP, g, 3 * E, N, Hh, D);              // dQ = dS K
      AT(5);
      contract_mfma(dS, 1, N, rows, RP, g + E, 3 * E, N, Hh, D);              // dK = dS^T Q
      AT(6);
      contract_mfma(P, 1, N, dO, GP, g + 2 * E, 3 * E, N, Hh, D);             // dV = P^T dO
      AT(7);
    } else {
      // one gradient element per thread and round (lanes walk over the 3E channels c of token n)
      for (int idx = threadIdx.x; idx < N * 3 * E; idx += blockDim.x) {
        const int n = d3E.quot(idx), c = idx - n * 3 * E;
        const int which = (c >= 2 * E) ? 2 : (c >= E ? 1 : 0), e = c - which * E, h = dD.quot(e);
        const float *m = (which == 2 ? P : dS) + h * NN + (which == 0 ? n * N : n);
        const float *x = (which == 0) ? rows + E + e : (which == 1 ? rows + e : dO + e);
        const int xp = (which == 2) ? GP : RP, sk = (which == 0) ? 1 : N;
        float a0 = 0.f, a1 = 0.f;
        int k = 0;
        for (; k + 3 < N; k += 4) {
          const float m0 = m[k * sk], m1 = m[(k + 1) * sk], m2 = m[(k + 2) * sk], m3 = m[(k + 3) * sk];
          const float x0 = x[k * xp], x1 = x[(k + 1) * xp], x2 = x[(k + 2) * xp], x3 = x[(k + 3) * xp];
          a0 = fmaf(m0, x0, a0), a1 = fmaf(m1, x1, a1), a0 = fmaf(m2, x2, a0), a1 = fmaf(m3, x3, a1);
        }
        for (; k < N; ++k) a0 = fmaf(m[k * sk], x[k * xp], a0);
        g[idx] = a0 + a1;
      }
    }
    __syncthreads();
    AT(4);
  }
}

static int family(int N, int D) { return D == 4 ? F_NARROW4 : D == 8 ? F_NARROW8 : ((D & 3) == 0 && D > 8 && N <= 16) ? F_WIDE : F_GENERIC; }

static int check(const Args &a, bool bwd, size_t *lds) {
  if (a.S < 0 || a.N < 1 || a.N > 64 || a.Hh < 1 || a.D < 1 || ((a.Hh * a.D) & 3)) return P2C_E_SHAPE;
  const size_t E = (size_t)a.Hh * a.D, NN = (size_t)a.N * a.N;
  *lds = sizeof(float) * (bwd ? a.N * (4 * E + 8) + 2 * a.Hh * NN : a.N * (3 * E + 4) + a.Hh * NN);
  return *lds <= 156 * 1024 ? 0 : P2C_E_SHAPE;
}

#define P2C_ATTN_ROW(K, F) {(const void *)K<F, 16>, (const void *)K<F, 32>, (const void *)K<F, 64>}
static const void *const fwd_kernels[4][3] = {P2C_ATTN_ROW(attn_fwd_kernel, F_GENERIC), P2C_ATTN_ROW(attn_fwd_kernel, F_NARROW4),
                                              P2C_ATTN_ROW(attn_fwd_kernel, F_NARROW8), P2C_ATTN_ROW(attn_fwd_kernel, F_WIDE)};
static const void *const bwd_kernels[4][3] = {P2C_ATTN_ROW(attn_bwd_kernel, F_GENERIC), P2C_ATTN_ROW(attn_bwd_kernel, F_NARROW4),
                                              P2C_ATTN_ROW(attn_bwd_kernel, F_NARROW8), P2C_ATTN_ROW(attn_bwd_kernel, F_WIDE)};
static void allow_lds() {
  static bool done = false;
  if (done) return;
  for (int f = 0; f < 4; ++f)
    for (int n = 0; n < 3; ++n) {
      (void)hipFuncSetAttribute(fwd_kernels[f][n], hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
      (void)hipFuncSetAttribute(bwd_kernels[f][n], hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
    }
  done = true;
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
  allow_lds();
  const int per_cu = (int)((156 * 1024) / lds) < 1 ? 1 : (int)((156 * 1024) / lds);
  int grid = 256 * (per_cu > 8 ? 8 : per_cu) * 4;               // a few rounds of resident workgroups; each strides over S
  if (grid > S) grid = S;
  const void *kernel = fwd_kernels[family(N, head_dim)][N <= 16 ? 0 : N <= 32 ? 1 : 2];
  void *kargs[] = {&a};
  (void)hipLaunchKernel(kernel, dim3((unsigned)grid), dim3(256), kargs, lds, (hipStream_t)stream);
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
  allow_lds();
  const int per_cu = (int)((156 * 1024) / lds) < 1 ? 1 : (int)((156 * 1024) / lds);
  int grid = 256 * (per_cu > 8 ? 8 : per_cu) * 4;
  if (grid > S) grid = S;
  const void *kernel = bwd_kernels[family(N, head_dim)][N <= 16 ? 0 : N <= 32 ? 1 : 2];
  void *kargs[] = {&a};
  (void)hipLaunchKernel(kernel, dim3((unsigned)grid), dim3(256), kargs, lds, (hipStream_t)stream);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

#ifdef P2C_ATTN_TRACE
extern "C" __attribute__((visibility("default"))) int p2c_debug_attn_trace(unsigned long long *out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(p2c_attn::g_attn_trace), sizeof(unsigned long long) * 16);
}
#endif
