// p2c_atb.hip -- C (+)= A^T [B | 1] for tall operands (K rows >> M, N <= a few hundred columns), fp32 MFMA, gfx950.
//
// The weight and bias gradients of every dense layer around the Seq2Seq recurrences (reference
// modules/movements/seq2seq/seq2seq.py:36-94: nn.LSTM input / hidden projections, Decoder.fc_out) are contractions over
// all T*B rows of a (T*B, 4H) gradient with a (T*B, in) activation: 256 x 64 outputs from K = 8 192 rows at cfg3. The BLAS
// libraries run that shape as a split-K GEMM plus a post-pass (rocBLAS 21.7 + 5 us, hipBLASLt 60 us for 268 MFLOP) and the
// bias gradient as a separate column reduction. Here: workgroup (16x16 output tile, K slice) -- enough slices that the
// grid fills the chip, the operand loads are 64-byte row segments and only memory parallelism hides their latency -- its
// eight waves split the slice and add up through LDS in wave order; a second launch adds the slices in order (bitwise
// reproducible) and writes C and the bias gradient (= the extra column of ones). Both operands are read row-major as
// they are (no transpose copy).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "../../include/p2c.h"

namespace p2c_atb_impl {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int WAVES = 8;

struct Args {
  const float *A, *B;
  float *C, *bias, *ws;
  int64_t lda, ldb, ldc, K;
  int32_t M, N, ones, accumulate, ks, tiles;
  float *bias2;                 // second destination of the bias gradient (b_ih and b_hh of an LSTM layer receive the same sums)
  const float *a_scale;         // row k of A is multiplied by a_scale[k / rows_per_scale] as it is loaded, or NULL
  int64_t rows_per_scale;
};

// One wave accumulates a 32x32 block of C as 2x2 MFMA tiles; lane (r, kk) loads the column PAIRS 2r, 2r+1 of row kk of
// A and of B as one 8-byte access each (a whole 128-byte line per row and operand), so tile (h, g) of the block is the
// interleaved set of rows m0 + 2i + h and columns n0 + 2j + g: two loads feed four MFMAs.
typedef float f32x2 __attribute__((ext_vector_type(2)));
// Workgroup numbering: the grid is launch_blocks() = (launched blocks of C) x (slices rounded up to 8). Workgroups are handed to
// the eight XCDs round-robin, so id & 7 is the XCD: it picks the slice within a run of eight, and all blocks of C of one slice
// follow each other on ONE XCD -- the 128-byte row segments of A and B they share (a row of A feeds every column block, a row
// of B every row block) are fetched into that XCD's L2 once instead of once per XCD.
// The bias gradient (A^T times a column of ones) of a problem with N % 32 == 0 would open a column block of its own for one
// useful column: instead the workgroups of column block 0 carry two more accumulators (A pairs against a [1 0 .. 0] operand)
// and publish them in the place of that virtual block, which only the finish pass knows about.
__device__ __forceinline__ bool bias_in_block(const Args &a) { return a.ones != 0 && (a.N & 31) == 0; }
__host__ __device__ __forceinline__ int launched_tiles(int tiles, int N, int ones) {
  const int nb_count = (N + ones + 31) >> 5;
  return (ones != 0 && (N & 31) == 0) ? tiles / nb_count * (nb_count - 1) : tiles;
}
static inline int launch_blocks(const Args &a) { return launched_tiles(a.tiles, a.N, a.ones) * ((a.ks + 7) & ~7); }

__device__ __forceinline__ void atb_body(const Args &a, const int bid, f32x4 (*red)[4][64], f32x4 (*redb)[2][4]) {
  const int lane = threadIdx.x & 63, r = lane & 15, kk = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nb_count = (a.N + a.ones + 31) >> 5;              // column blocks the finish pass walks (a.tiles = row blocks x this)
  const int nb_real = bias_in_block(a) ? nb_count - 1 : nb_count;
  const int tiles_real = a.tiles / nb_count * nb_real;
  const int slice = ((bid >> 3) / tiles_real) * 8 + (bid & 7), blk_r = (bid >> 3) % tiles_real;
  if (slice >= a.ks) return;
  const int mb = blk_r / nb_real, nb = blk_r - mb * nb_real;
  const int blk = mb * nb_count + nb;
  const bool with_bias = bias_in_block(a) && nb == 0;
  const float onev = r == 0 ? 1.f : 0.f;
  const int m = mb * 32 + 2 * r, n = nb * 32 + 2 * r;          // first column of this lane's pair
  const bool vec_a = (m + 1 < a.M) && ((a.lda & 1) == 0) && ((reinterpret_cast<uintptr_t>(a.A) & 7) == 0);
  const bool vec_b = (n + 1 < a.N) && ((a.ldb & 1) == 0) && ((reinterpret_cast<uintptr_t>(a.B) & 7) == 0);
  const int64_t per_slice = ((a.K + (int64_t)a.ks * WAVES * 4 - 1) / ((int64_t)a.ks * WAVES * 4)) * WAVES * 4;
  const int64_t s0 = slice * per_slice, chunk = per_slice / WAVES;             // rows per wave: a multiple of the MFMA k
  const int64_t k0 = s0 + wave * chunk, kend = (k0 + chunk < a.K) ? k0 + chunk : a.K;
  const float *ap = a.A + m, *bp = a.B + n;
  auto col = [&](const float *p, int64_t off, int c, int limit, bool one_ok) -> float {   // scalar edge path
    return c < limit ? p[off] : ((one_ok && c == limit) ? 1.f : 0.f);
  };
  f32x4 acc[2][2] = {{{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}};
  f32x4 accb[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#ifndef P2C_ATB_U
#define P2C_ATB_U 4
#endif
  constexpr int U = P2C_ATB_U;
  // The lane's rows are k0 + kk, + 4, + 8, ...: its two operand pointers and the index of its row's scale factor are CARRIED
  // from row to row (one division per wave, then add / compare), and whole steps of 4 U rows run without per-row tests. (With
  // the row product, the 64-bit division `row / rows_per_scale` and a branch per row inside the loop, a step of the scaled form
  // was ~660 scalar + vector instructions around its 16 MFMAs.)
  const bool scaled = a.a_scale != nullptr;
  const int64_t rps = scaled ? a.rows_per_scale : 1;
  int64_t sidx = (k0 + kk) / rps, srem = (k0 + kk) % rps;
  const int64_t sq4 = 4 / rps, sr4 = 4 % rps;
  const float *pa = ap + (k0 + kk) * a.lda, *pb = bp + (k0 + kk) * a.ldb;
  const int64_t sa = 4 * a.lda, sb = 4 * a.ldb;
  // (whole: every lane of the workgroup takes the 8-byte loads -- the block lies inside both matrices -- so nothing in the
  // step depends on the lane but the addresses)
  auto step = [&](auto guarded, auto whole, int64_t k) {
    f32x2 av[U], bv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool ok = !decltype(guarded)::value || k + 4 * u + kk < kend;
      av[u] = (f32x2){0.f, 0.f}, bv[u] = (f32x2){0.f, 0.f};
      if (ok) {
        if (decltype(whole)::value || vec_a) av[u] = *reinterpret_cast<const f32x2 *>(pa);
        else av[u] = (f32x2){col(pa, 0, m, a.M, false), col(pa, 1, m + 1, a.M, false)};
        if (scaled) av[u] *= a.a_scale[sidx];
        if (decltype(whole)::value || vec_b) bv[u] = *reinterpret_cast<const f32x2 *>(pb);
        else bv[u] = (f32x2){col(pb, 0, n, a.N, a.ones != 0), col(pb, 1, n + 1, a.N, a.ones != 0)};
      }
      pa += sa, pb += sb;
      if (scaled) {
        sidx += sq4, srem += sr4;
        if (srem >= rps) srem -= rps, ++sidx;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int g = 0; g < 2; ++g)
          acc[h][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][h], bv[u][g], acc[h][g], 0, 0, 0);
      if (with_bias) {
#pragma unroll
        for (int h = 0; h < 2; ++h) accb[h] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][h], onev, accb[h], 0, 0, 0);
      }
    }
  };
  const bool whole = mb * 32 + 32 <= a.M && nb * 32 + 32 <= a.N && ((a.lda | a.ldb) & 1) == 0 &&
                     ((reinterpret_cast<uintptr_t>(a.A) | reinterpret_cast<uintptr_t>(a.B)) & 7) == 0;
  int64_t k = k0;
  if (whole) {
    for (; k + 4 * U <= kend; k += 4 * U) step(std::false_type{}, std::true_type{}, k);
    if (k < kend) step(std::true_type{}, std::true_type{}, k);
  } else {
    for (; k + 4 * U <= kend; k += 4 * U) step(std::false_type{}, std::false_type{}, k);
    if (k < kend) step(std::true_type{}, std::false_type{}, k);
  }
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int g = 0; g < 2; ++g) red[wave][h * 2 + g][lane] = acc[h][g];
  if (with_bias && r == 0) redb[wave][0][kk] = accb[0], redb[wave][1][kk] = accb[1];     // column 0 of the two bias tiles
  __syncthreads();
  // waves 0..3 each finish one of the four tiles (slices of K added in wave order); waves 4, 5 the two bias tiles
  if (wave >= 4) {
    if (with_bias && wave < 6 && lane < 4) {
      f32x4 s = redb[0][wave - 4][lane];
#pragma unroll
      for (int w = 1; w < WAVES; ++w) s += redb[w][wave - 4][lane];
      // tile (h = wave - 4, g = 0) of the virtual column block nb_real, lane (r = 0, kk = lane)
      reinterpret_cast<f32x4 *>(a.ws)[(((size_t)slice * a.tiles + blk + nb_real) * 4 + (wave - 4) * 2) * 64 + 16 * lane] = s;
    }
    return;
  }
  f32x4 s = red[0][wave][lane];
#pragma unroll
  for (int w = 1; w < WAVES; ++w) s += red[w][wave][lane];
  reinterpret_cast<f32x4 *>(a.ws)[(((size_t)slice * a.tiles + blk) * 4 + wave) * 64 + lane] = s;
}

__global__ __launch_bounds__(64 * WAVES) void atb_kernel(const Args a) {
  __shared__ f32x4 red[WAVES][4][64];
  __shared__ f32x4 redb[WAVES][2][4];
  atb_body(a, blockIdx.x, red, redb);
}

// slices added in order, one thread per (block, tile, lane)
__device__ __forceinline__ void finish_body(const Args &a, const int bid) {
  const int idx = bid * 256 + threadIdx.x, lane = idx & 63, tile = (idx >> 6) & 3, blk = idx >> 8;
  if (blk >= a.tiles) return;
  const f32x4 *p = reinterpret_cast<const f32x4 *>(a.ws) + ((size_t)blk * 4 + tile) * 64 + lane;
  const size_t stride = (size_t)a.tiles * 256;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  int k = 0;
  for (; k + 8 <= a.ks; k += 8) {                              // eight slices in flight, added in slice order
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(k + u) * stride];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; k < a.ks; ++k) s += p[(size_t)k * stride];
  const int nb_count = (a.N + a.ones + 31) >> 5;
  const int mb = blk / nb_count, nb = blk - mb * nb_count, h = tile >> 1, g = tile & 1;
  const int col = nb * 32 + 2 * (lane & 15) + g;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = mb * 32 + 2 * (4 * (lane >> 4) + i) + h;
    if (row >= a.M) continue;
    if (col < a.N) {
      float *c = a.C + (int64_t)row * a.ldc + col;
      *c = (a.accumulate & 1) ? *c + s[i] : s[i];
    } else if (a.ones && col == a.N && a.bias) {
      a.bias[row] = (a.accumulate & 2) ? a.bias[row] + s[i] : s[i];
      if (a.bias2) a.bias2[row] = (a.accumulate & 2) ? a.bias2[row] + s[i] : s[i];
    }
  }
}
__global__ __launch_bounds__(256) void atb_finish_kernel(const Args a) { finish_body(a, blockIdx.x); }

// ---- grouped form: up to MAXG independent problems behind one launch pair. The backward of a Seq2Seq layer stack is a
// row of such contractions (decoder: dW_ih0, dW_ih1, dW_fc + db_fc, dW_hh0 + db, dW_hh1 + db); launched one by one each
// pair costs ~2 x 2-5 us of launch + tail on top of its ~10 us of work. The problems ride in the kernel arguments; a
// workgroup finds its problem by a scan over at most MAXG prefix sums.
constexpr int MAXG = 8;
struct Group {
  Args p[MAXG];
  int32_t first[MAXG + 1], ffirst[MAXG + 1];      // first workgroup of problem i in the main / the finish launch
  int32_t n;
};
__global__ __launch_bounds__(64 * WAVES) void atb_group_kernel(const Group g) {
  __shared__ f32x4 red[WAVES][4][64];
  __shared__ f32x4 redb[WAVES][2][4];
  int i = 0;
  while (i + 1 < g.n && (int)blockIdx.x >= g.first[i + 1]) ++i;
  atb_body(g.p[i], blockIdx.x - g.first[i], red, redb);
}
__global__ __launch_bounds__(256) void atb_group_finish_kernel(const Group g) {
  int i = 0;
  while (i + 1 < g.n && (int)blockIdx.x >= g.ffirst[i + 1]) ++i;
  finish_body(g.p[i], blockIdx.x - g.ffirst[i]);
}

}  // namespace p2c_atb_impl

static inline int slices_for(int tiles, int64_t K) {
  int ks = 1024 / (tiles < 1 ? 1 : tiles);                     // ~1024 workgroups: four per CU
  const int64_t max_ks = (K + 255) / 256;                      // at least 256 rows per slice
  if (ks > max_ks) ks = (int)max_ks;
  return ks < 1 ? 1 : (ks > 256 ? 256 : ks);   // (few tiles over very many rows: the transformer's 546 624 x 96 x 32)
}

extern "C" int64_t p2c_atb_workspace_floats(int64_t K, int32_t M, int32_t N, int32_t with_bias) {
  if (K < 0 || M < 1 || N < 1) return 0;
  const int blocks = ((M + 31) / 32) * ((N + (with_bias ? 1 : 0) + 31) / 32);
  return (int64_t)slices_for(blocks, K) * blocks * 1024;
}

extern "C" int p2c_atb_scaled(const float *A, int64_t lda, const float *B, int64_t ldb, int64_t K, int32_t M, int32_t N, float *C,
                              int64_t ldc, float *bias_out, int32_t accumulate, const float *a_scale, int64_t rows_per_scale,
                              float *workspace, void *stream) {
  using namespace p2c_atb_impl;
  if (!A || !B || !C || !workspace) return P2C_E_NULL;
  if (K < 0 || M < 1 || N < 1 || lda < M || ldb < N || ldc < N || (a_scale && rows_per_scale < 1)) return P2C_E_SHAPE;
  Args a{A, B, C, bias_out, workspace, lda, ldb, ldc, K, M, N, bias_out ? 1 : 0, accumulate & 3, 1, 0, nullptr, a_scale,
         rows_per_scale};
  a.tiles = ((M + 31) / 32) * ((N + a.ones + 31) / 32);      // 32x32 blocks of C
  a.ks = slices_for(a.tiles, K);
  hipLaunchKernelGGL(atb_kernel, dim3((unsigned)launch_blocks(a)), dim3(64 * WAVES), 0, (hipStream_t)stream, a);
  hipLaunchKernelGGL(atb_finish_kernel, dim3((unsigned)a.tiles), dim3(256), 0, (hipStream_t)stream, a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

extern "C" int p2c_atb(const float *A, int64_t lda, const float *B, int64_t ldb, int64_t K, int32_t M, int32_t N, float *C,
                       int64_t ldc, float *bias_out, int32_t accumulate, float *workspace, void *stream) {
  return p2c_atb_scaled(A, lda, B, ldb, K, M, N, C, ldc, bias_out, accumulate, nullptr, 1, workspace, stream);
}

static int group_fill(const p2c_atb_problem *p, int32_t n, float *workspace, p2c_atb_impl::Group &g, int64_t *ws_total) {
  using namespace p2c_atb_impl;
  if (!p) return P2C_E_NULL;
  if (n < 1 || n > MAXG) return P2C_E_SHAPE;
  g.n = n, g.first[0] = g.ffirst[0] = 0;
  int64_t off = 0;
  for (int i = 0; i < n; ++i) {
    const p2c_atb_problem &q = p[i];
    if (!q.a || !q.b || !q.out) return P2C_E_NULL;
    if (q.K < 0 || q.M < 1 || q.N < 1 || q.a_stride < q.M || q.b_stride < q.N || q.out_stride < q.N) return P2C_E_SHAPE;
    if (q.bias_out2 && !q.bias_out) return P2C_E_NULL;
    Args &a = g.p[i];
    a = Args{q.a, q.b, q.out, q.bias_out, workspace ? workspace + off : nullptr, q.a_stride, q.b_stride, q.out_stride, q.K,
             q.M, q.N, q.bias_out ? 1 : 0, q.flags & 3, 1, 0, q.bias_out2, nullptr, 1};
    a.tiles = ((q.M + 31) / 32) * ((q.N + a.ones + 31) / 32);
    a.ks = slices_for(a.tiles, q.K);
    g.first[i + 1] = g.first[i] + launch_blocks(a);        // (a multiple of 8: every problem starts on XCD 0)
    g.ffirst[i + 1] = g.ffirst[i] + a.tiles;
    off += (int64_t)a.ks * a.tiles * 1024;
  }
  *ws_total = off;
  return 0;
}

extern "C" int64_t p2c_atb_group_workspace_floats(const p2c_atb_problem *p, int32_t n) {
  p2c_atb_impl::Group g;
  int64_t total = 0;
  return group_fill(p, n, nullptr, g, &total) ? 0 : total;
}

extern "C" int p2c_atb_group(const p2c_atb_problem *p, int32_t n, float *workspace, void *stream) {
  using namespace p2c_atb_impl;
  if (!workspace) return P2C_E_NULL;
  Group g;
  int64_t total = 0;
  int rc = group_fill(p, n, workspace, g, &total);
  if (rc) return rc;
  hipLaunchKernelGGL(atb_group_kernel, dim3((unsigned)g.first[n]), dim3(64 * WAVES), 0, (hipStream_t)stream, g);
  hipLaunchKernelGGL(atb_group_finish_kernel, dim3((unsigned)g.ffirst[n]), dim3(256), 0, (hipStream_t)stream, g);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}
