// p2c_gemm.hip -- dense layers on fp32 MFMA with a fused epilogue (K16): C = epilogue(A op(B)).
//
// What it replaces: the library GEMMs behind every nn.Linear of the model plugins -- PoseTransformer's qkv / proj / fc1 / fc2
// (reference modules/movements/pose_former/pose_former.py:62-76 binds the transformer; 2.8 TFLOP of fp32 GEMMs per cfg5 step)
// and the input projections / input gradients around the Seq2Seq recurrences (movements/seq2seq/seq2seq.py:21-94) -- together
// with the element-wise launches that followed them: bias, GELU (exact erf form, torch.nn.GELU()), its derivative in the
// backward, the per-sample stochastic-depth factor and the residual add. fp32 in, fp32 accumulate: v_mfma_f32_32x32x2_f32 is
// bit-for-bit an fmaf chain in k order (no TF32-like shortcut exists on gfx950), 64 FLOP/clk/SIMD = the fp32 peak.
//
//   trans_b = 1 ("NT"): B is (N, K) row-major -- y = x W^T, nn.Linear's forward;
//   trans_b = 0 ("NN"): B is (K, N) row-major -- dx = dy W, its input gradient.
//   epilogue, in this order:  v = acc + bias[n];  act 1: (aux_out = v;) v = gelu(v);  act 2: v *= gelu'(aux[m][n]);
//                             v *= row_scale[m / rows_per_scale];  v += residual[m][n];  C[m][n] = v.
//
// Tiling: a workgroup of four wavefronts owns a 128 x BN tile of C (BN = 128: 2 x 2 waves of 64 x 64; BN = 64 / 32: 4 x 1 waves
// of 32 x BN -- the spatial blocks of PoseFormer have 32..96 output features over 546 624 rows and are pure streaming).
// K advances in steps of 32: the next A / B slabs (128 x 32 floats each: whole 128-byte lines per row) are loaded into
// registers while the current ones are multiplied out of LDS, where they sit k-major ([k][m], pitch 129: conflict-free both
// for the transposing stores and for the MFMA operand reads -- lane l reads row l & 31 of k-step l >> 5).
// The weight gradient dW = dy^T x is K12 (p2c_atb.hip: K >> M, N).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/p2c.h"

namespace p2c_gemm_impl {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BK = 32, NTH = 256;

#ifdef P2C_GEMM_TRACE   // developer build only (tools/gemmtrace.py): shader-clock stamps of one workgroup of gemm_kernel
#ifndef P2C_GEMM_TRACE_BLOCK
#define P2C_GEMM_TRACE_BLOCK 2000
#endif
static __device__ unsigned long long g_gemm_trace[65];      // [62], [63]: wall clock at stamps 0 and 61; [64]: at stamp 0 of workgroup 0
static __device__ int g_gemm_trace_block = P2C_GEMM_TRACE_BLOCK;
#define GT(i)                                                                                                     \
  do {                                                                                                            \
    if ((i) == 0 && blockIdx.x == 0 && threadIdx.x == 0) g_gemm_trace[64] = wall_clock64();                      \
    if ((int)blockIdx.x == g_gemm_trace_block && threadIdx.x == 0 && (i) < 62) {                                 \
      g_gemm_trace[i] = __builtin_readcyclecounter();                                                             \
      if ((i) == 0 || (i) == 61) g_gemm_trace[(i) == 0 ? 62 : 63] = wall_clock64();   /* 100 MHz: the shader clock under load */ \
    }                                                                                                             \
  } while (0)
#else
#define GT(i)
#endif
constexpr int LDA = BM + 1;          // pitch of the k-major A slab (floats)

__device__ __forceinline__ float gelu(float v) { return 0.5f * v * (1.f + erff(v * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad(float z) {
  return 0.5f * (1.f + erff(z * 0.70710678118654752f)) + z * 0.39894228040143268f * __expf(-0.5f * z * z);
}

// row-major slabs (A of the NT / NN forms, B of the NT form): (rows x BK) floats from a row-major (rows, K) matrix. 16-byte piece
// q = t + 256 i is row q / 8, k-piece q % 8: the eight lanes of a row take one whole 128-byte line per instruction. (The first
// version gave a thread 16 consecutive k of a row: each of its four load instructions touched 32 lines for 32 bytes each, the
// same lines four times over -- the timeline of a workgroup, tools/gemmtrace.py, showed the ISSUE of the eight loads of a
// k-tile taking up to 19 k cycles with three workgroups per CU: the vector-memory path was saturated with line requests.)
struct SlabRegs {
  f32x4 v[4];
};
template <bool VEC>
__device__ __forceinline__ void load_rows(const float *base, int64_t ld, int row0, int n_rows, int k0, int K, SlabRegs &r, int rows_in_tile,
                                          int tid) {
  const int kp = (tid & 7) * 4, k = k0 + kp;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (tid >> 3) + 32 * i;
    const bool row_ok = row < rows_in_tile && row0 + row < n_rows;
    const float *p = base + (int64_t)(row0 + (row_ok ? row : 0)) * ld + k;
    if (VEC) {
      r.v[i] = (row_ok && k < K) ? *reinterpret_cast<const f32x4 *>(p) : (f32x4){0.f, 0.f, 0.f, 0.f};
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) r.v[i][e] = (row_ok && k + e < K) ? p[e] : 0.f;
    }
  }
}
// ... into the k-major LDS slab [k][row] (pitch `ld` odd: the 64 lanes of a store -- 8 rows x 8 k-pieces -- fall two per bank)
__device__ __forceinline__ void store_rows_transposed(float *slab, int ld, const SlabRegs &r, int rows_in_tile, int tid) {
  const int kp = (tid & 7) * 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (tid >> 3) + 32 * i;
    if (row < rows_in_tile) {
#pragma unroll
      for (int e = 0; e < 4; ++e) slab[(kp + e) * ld + row] = r.v[i][e];
    }
  }
}
// k-major slabs (B of the NN form, A and B of the TN form): (BK x W) floats from a row-major (K, W') matrix, stored as they are
// ([k][n], pitch W + 4 floats: 16-byte aligned rows). 16-byte piece q = t + 256 i of the slab is row q / (W / 4), column piece
// q % (W / 4): consecutive lanes take consecutive pieces -- whole 128-byte lines from memory and conflict-free ds_write_b128
// into LDS. (The first version gave a thread 16 consecutive floats of a row and wrote them as dwords: lanes 64 bytes apart hit
// 16 of 32 banks -- SQ_LDS_BANK_CONFLICT 35 M / 71 M cycles per NN / TN launch at 21 024 x 2 496 x 832, none in the NT form.)
template <int W>
struct KnMap {
  static constexpr int PIECES = BK * W / 4 / NTH;        // pieces per thread: 4 / 2 / 1
  static constexpr int PPR = W / 4;                      // pieces per k-row
  __device__ static __forceinline__ int row(int i, int tid) { return (tid + i * NTH) / PPR; }
  __device__ static __forceinline__ int col(int i, int tid) { return ((tid + i * NTH) % PPR) * 4; }
};
template <int W, bool VEC>
__device__ __forceinline__ void load_kn(const float *base, int64_t ld, int k0, int K, int n0, int N, SlabRegs &r, int tid) {
  using M = KnMap<W>;
#pragma unroll
  for (int i = 0; i < M::PIECES; ++i) {
    const int k = k0 + M::row(i, tid), n = n0 + M::col(i, tid);
    const bool k_ok = k < K;
    const float *p = base + (int64_t)(k_ok ? k : 0) * ld + n;
    if (VEC) {
      r.v[i] = (k_ok && n < N) ? *reinterpret_cast<const f32x4 *>(p) : (f32x4){0.f, 0.f, 0.f, 0.f};
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) r.v[i][e] = (k_ok && n + e < N) ? p[e] : 0.f;
    }
  }
}
template <int W>
__device__ __forceinline__ void store_kn(float *slab, int ld, const SlabRegs &r, int tid) {
  using M = KnMap<W>;
#pragma unroll
  for (int i = 0; i < M::PIECES; ++i) *reinterpret_cast<f32x4 *>(slab + M::row(i, tid) * ld + M::col(i, tid)) = r.v[i];
}

// ---- the same slabs through buffer loads (FAST forms: 16-byte-aligned operands, K a multiple of BK, every operand below 2 GB).
// A thread's four (two, one) pieces sit at the same place of every k-tile: their byte offsets are formed ONCE, with pieces outside
// the matrix (rows past M or N, columns past N) parked beyond the buffer's records -- the range check of a raw buffer looks at
// the lane offset only and returns zeros there -- and the k-tile is the scalar offset of the instruction. A k-tile's fetch is
// its 6 load instructions and one scalar add; with pointers, every piece carried a 64-bit row product, two compares and a
// branch around the load, and the timeline (tools/gemmtrace.py) showed the ISSUE of a k-tile's loads taking 2 000 - 3 900 cycles
// -- the largest part of what a wave spends outside its MFMA phase.
constexpr int OOB_OFF = 0x7fffff00;
struct SlabOff {
  int v[4];
};
__device__ __forceinline__ __amdgpu_buffer_rsrc_t operand_rsrc(const float *base, int64_t rows, int64_t ld, int64_t cols) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, (int)(((rows - 1) * ld + cols) * 4), 0x00020000);
}
__device__ __forceinline__ SlabOff rows_offsets(int64_t ld, int row0, int n_rows, int rows_in_tile, int tid) {
  SlabOff o;
  const int kp = (tid & 7) * 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (tid >> 3) + 32 * i;
    o.v[i] = (row < rows_in_tile && row0 + row < n_rows) ? (int)(((int64_t)(row0 + row) * ld + kp) * 4) : OOB_OFF;
  }
  return o;
}
template <int W>
__device__ __forceinline__ SlabOff kn_offsets(int64_t ld, int n0, int N, int tid) {
  using M = KnMap<W>;
  SlabOff o;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (i >= M::PIECES) { o.v[i] = OOB_OFF; continue; }
    const int n = n0 + M::col(i, tid);
    o.v[i] = n < N ? (int)(((int64_t)M::row(i, tid) * ld + n) * 4) : OOB_OFF;
  }
  return o;
}
template <int PIECES>
__device__ __forceinline__ void load_pieces(__amdgpu_buffer_rsrc_t rs, const SlabOff &o, int soff, SlabRegs &r) {
#pragma unroll
  for (int i = 0; i < PIECES; ++i) r.v[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, o.v[i], soff, 0));
}

// Workgroups are dealt to the eight XCDs round-robin by their linear id, and every XCD has its own L2: with tile = id, the
// tiles that share an operand slab (one row of tiles shares A, one column B) are spread over all eight L2s. xcd_tile gives
// XCD x the contiguous tile range [x n / 8, (x + 1) n / 8): what runs side by side on an XCD is a compact patch of C.
// The grid is rounded up to a multiple of 8; ids whose tile falls outside leave at once.
#ifndef P2C_GEMM_XCD_SWIZZLE
#define P2C_GEMM_XCD_SWIZZLE 1
#endif
__device__ __forceinline__ int xcd_tile(int id, int n) {
#if P2C_GEMM_XCD_SWIZZLE
  return (id & 7) * ((n + 7) >> 3) + (id >> 3);
#else
  return id;
#endif
}
static inline unsigned xcd_grid(int64_t n) { return (unsigned)(((n + 7) >> 3) << 3); }

// TN form (weight gradients of wide layers, dW = dy^T x: A = dy is (K, M) row-major, B = x is (K, N) row-major, K = rows >> M, N):
// split over K in `slices` (blockIdx.y), every slice writes its own (M, N) slab of the workspace, tn_finish_kernel adds the
// slabs in slice order (bitwise reproducible, no atomics) into C, written or accumulated.
struct TnArgs {
  const float *a, *b;
  float *ws, *c;
  int64_t lda, ldb, ldc;
  int32_t M, N, K, k_chunk, slices, accumulate;
  const float *row_scale;       // A's row k is multiplied by row_scale[k / rows_per_scale] as it is loaded (dy of a layer whose
  int32_t rows_per_scale;       // output went through a per-sample stochastic-depth factor), or NULL
  float *bias, *ws_bias;        // column sums of the (scaled) A = the bias gradient, or NULL; (slices, M) slabs
  int32_t bias_accumulate;
};
template <int BN, bool VEC, bool FAST = false>
__global__ __launch_bounds__(NTH, BN == 128 ? 3 : (BN == 64 ? 5 : 6)) void gemm_tn_kernel(const TnArgs d) {
  constexpr int WM = (BN == 128) ? 2 : 4, TM = (BN == 128) ? 2 : 1, TN = (BN == 128) ? 2 : BN / 32;
  __shared__ __attribute__((aligned(16))) float As[BK * (BM + 4)];
  __shared__ __attribute__((aligned(16))) float Bs[BK * (BN + 4)];
  const int n_tiles = (d.N + BN - 1) / BN, tiles = ((d.M + BM - 1) / BM) * n_tiles;
  const int id = xcd_tile((int)blockIdx.x, tiles * d.slices);
  if (id >= tiles * d.slices) return;
  const int slice = id / tiles, tile = id - slice * tiles;
  const int tm = tile / n_tiles, tn = tile % n_tiles;
  const int m0 = tm * BM, n0 = tn * BN;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave % WM, wn = wave / WM, li = lane & 31, lk = lane >> 5;
  const int k_begin = slice * d.k_chunk, k_end = k_begin + d.k_chunk < d.K ? k_begin + d.k_chunk : d.K;
  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
  SlabRegs ra, rb;
  const int nk = (k_end - k_begin + BK - 1) / BK;
  const bool sums = d.bias && tn == 0 && (int)threadIdx.x < BM;
  float colsum = 0.f;
  SlabOff oa, ob;
  __amdgpu_buffer_rsrc_t rsa, rsb;
  if (FAST) {
    rsa = operand_rsrc(d.a, d.K, d.lda, d.M), rsb = operand_rsrc(d.b, d.K, d.ldb, d.N);
    oa = kn_offsets<BM>(d.lda, m0, d.M, (int)threadIdx.x), ob = kn_offsets<BN>(d.ldb, n0, d.N, (int)threadIdx.x);
  }
  // index of a row's factor: below 2^21 rows floor((k + 0.5) / rows_per_scale) in fp32 is exact (the quotient's rounding error,
  // q * 2^-22, stays below the 0.5 / rows_per_scale that separates it from the next integer): three instructions instead of the
  // ~35 of an integer division, four times per k-tile and thread
  const bool fdiv = d.K <= (1 << 21);
  const float inv_rps = d.row_scale ? 1.f / (float)d.rows_per_scale : 0.f;
  auto fetch = [&](int kt) {
    if (FAST) load_pieces<KnMap<BM>::PIECES>(rsa, oa, (int)((k_begin + kt * BK) * d.lda * 4), ra);
    else load_kn<BM, VEC>(d.a, d.lda, k_begin + kt * BK, k_end, m0, d.M, ra, (int)threadIdx.x);
    if (d.row_scale) {
#pragma unroll
      for (int i = 0; i < KnMap<BM>::PIECES; ++i) {
        const int k = k_begin + kt * BK + KnMap<BM>::row(i, (int)threadIdx.x);
        const int idx = fdiv ? (int)(((float)k + 0.5f) * inv_rps) : k / d.rows_per_scale;
        ra.v[i] *= (FAST || k < k_end) ? d.row_scale[idx] : 0.f;
      }
    }
    if (FAST) load_pieces<KnMap<BN>::PIECES>(rsb, ob, (int)((k_begin + kt * BK) * d.ldb * 4), rb);
    else load_kn<BN, VEC>(d.b, d.ldb, k_begin + kt * BK, k_end, n0, d.N, rb, (int)threadIdx.x);
  };
  auto commit = [&]() {
    store_kn<BM>(As, BM + 4, ra, (int)threadIdx.x);
    store_kn<BN>(Bs, BN + 4, rb, (int)threadIdx.x);
  };
  if (nk > 0) {
    fetch(0);
    commit();
  }
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) fetch(kt + 1);
    // operand fragments of k-step ks + 1 are read from LDS BEFORE the MFMAs of k-step ks are issued (two register sets): the
    // compiler keeps source order here, and with read -> wait -> four MFMAs per step every step exposed an LDS round trip
    float af[2][TM], bf[2][TN];
    auto frag = [&](int ks, int buf) {
      const int k = ks * 2 + lk;
#pragma unroll
      for (int a = 0; a < TM; ++a) af[buf][a] = As[k * (BM + 4) + (wm * TM + a) * 32 + li];
#pragma unroll
      for (int b = 0; b < TN; ++b) bf[buf][b] = Bs[k * (BN + 4) + (wn * TN + b) * 32 + li];
    };
    frag(0, 0);
#pragma unroll
    for (int ks = 0; ks < BK / 2; ++ks) {
      if (ks + 1 < BK / 2) frag(ks + 1, (ks + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);           // (keep the reads of step ks + 1 ahead of the MFMAs of step ks)
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[ks & 1][a], bf[ks & 1][b], acc[a][b], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (sums) {
#pragma unroll
      for (int k = 0; k < BK; ++k) colsum += As[k * (BM + 4) + (int)threadIdx.x];
    }
    __syncthreads();
    if (kt + 1 < nk) {
      commit();
      __syncthreads();
    }
  }
  if (sums && m0 + (int)threadIdx.x < d.M) d.ws_bias[(int64_t)slice * d.M + m0 + (int)threadIdx.x] = colsum;
  float *slab = d.ws + (int64_t)slice * d.M * d.N;
#pragma unroll
  for (int b = 0; b < TN; ++b) {
    const int n = n0 + (wn * TN + b) * 32 + li;
    if (n >= d.N) continue;
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + (wm * TM + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        if (m < d.M) slab[(int64_t)m * d.N + n] = acc[a][b][r];
      }
  }
}
// Slabs added in a FIXED order (bitwise reproducible): a workgroup owns 32 outputs, thread (part p, output i) adds slices p,
// p + 8, ... in that order, eight loads in flight, and the eight parts of an output meet in LDS in part order. (A skinny dW from
// 546 624 rows has ~1 000 slices: one thread walking them one after the other is ~1 000 dependent round trips, 1.2 ms.)
__device__ __forceinline__ float tn_slab_sum(const float *ws, int64_t stride, int64_t i, int slices, int part) {
  float v = 0.f;
  int s = part;
  for (; s + 56 < slices; s += 64) {
    float t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = ws[(int64_t)(s + 8 * u) * stride + i];
#pragma unroll
    for (int u = 0; u < 8; ++u) v += t[u];
  }
  for (; s < slices; s += 8) v += ws[(int64_t)s * stride + i];
  return v;
}
__global__ __launch_bounds__(256) void tn_finish_kernel(const TnArgs d) {
  __shared__ float red[8][33];
  const int64_t total = (int64_t)d.M * d.N, n_all = total + (d.bias ? d.M : 0);
  const int e = threadIdx.x & 31, part = threadIdx.x >> 5;
  for (int64_t base = (int64_t)blockIdx.x * 32; base < n_all; base += (int64_t)gridDim.x * 32) {
    const int64_t i = base + e;
    float v = 0.f;
    if (i < total) v = tn_slab_sum(d.ws, total, i, d.slices, part);
    else if (i < n_all) v = tn_slab_sum(d.ws_bias, d.M, i - total, d.slices, part);
    red[part][e] = v;
    __syncthreads();
    if (part == 0 && i < n_all) {
      float t = red[0][e];
#pragma unroll
      for (int p = 1; p < 8; ++p) t += red[p][e];
      if (i < total) {
        float *o = d.c + (i / d.N) * d.ldc + (i % d.N);
        *o = d.accumulate ? *o + t : t;
      } else {
        float *o = d.bias + (i - total);
        *o = d.bias_accumulate ? *o + t : t;
      }
    }
    __syncthreads();
  }
}
// More than eight slabs with 16-byte-aligned outputs: the eight-part scheme of tn_finish_kernel on FOUR neighbouring outputs per
// thread (same order of additions per output, so the same bits; a quarter of the load instructions).
__device__ __forceinline__ f32x4 tn_slab_sum4(const float *ws, int64_t stride, int64_t i, int slices, int part) {
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  int s = part;
  for (; s + 56 < slices; s += 64) {
    f32x4 t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = *reinterpret_cast<const f32x4 *>(ws + (int64_t)(s + 8 * u) * stride + i);
#pragma unroll
    for (int u = 0; u < 8; ++u) v += t[u];
  }
  for (; s < slices; s += 8) v += *reinterpret_cast<const f32x4 *>(ws + (int64_t)s * stride + i);
  return v;
}
__global__ __launch_bounds__(256) void tn_finish_vec8_kernel(const TnArgs d) {
  __shared__ f32x4 red[8][33];
  const int64_t total = (int64_t)d.M * d.N, nv = total >> 2, n_all = nv + (d.bias ? d.M : 0);
  const int e = threadIdx.x & 31, part = threadIdx.x >> 5;
  for (int64_t base = (int64_t)blockIdx.x * 32; base < n_all; base += (int64_t)gridDim.x * 32) {
    const int64_t j = base + e;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (j < nv) v = tn_slab_sum4(d.ws, total, j * 4, d.slices, part);
    else if (j < n_all) v[0] = tn_slab_sum(d.ws_bias, d.M, j - nv, d.slices, part);
    red[part][e] = v;
    __syncthreads();
    if (part == 0 && j < n_all) {
      f32x4 t = red[0][e];
#pragma unroll
      for (int p = 1; p < 8; ++p) t += red[p][e];
      if (j < nv) {
        const int64_t i = j * 4;
        f32x4 *o = reinterpret_cast<f32x4 *>(d.c + (i / d.N) * d.ldc + (i % d.N));
        *o = d.accumulate ? *o + t : t;
      } else {
        float *o = d.bias + (j - nv);
        *o = d.bias_accumulate ? *o + t[0] : t[0];
      }
    }
    __syncthreads();
  }
}
// Few slices (the wide layers: 2 .. 8 slabs of up to 2 M outputs each): a thread owns FOUR neighbouring outputs and adds their
// slabs in slice order from 16-byte loads, all of them in flight at once -- the same sums, bit for bit, as the eight-part form
// above gives for <= 8 slices (each part holds one slab), at the rate of a streaming pass instead of 4-byte loads by a
// quarter of the threads (35 us -> see DESIGN.md for 5 slabs of 2 496 x 832).
__global__ __launch_bounds__(256) void tn_finish_vec_kernel(const TnArgs d) {
  const int64_t total = (int64_t)d.M * d.N, nv = total >> 2, n_all = nv + (d.bias ? d.M : 0);
  for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < n_all; j += (int64_t)gridDim.x * 256) {
    if (j < nv) {
      const int64_t i = j * 4;
      f32x4 t[8];
#pragma unroll
      for (int s = 0; s < 8; ++s)
        t[s] = s < d.slices ? *reinterpret_cast<const f32x4 *>(d.ws + (int64_t)s * total + i) : (f32x4){0.f, 0.f, 0.f, 0.f};
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 8; ++s)
        if (s < d.slices) v += t[s];
      f32x4 *o = reinterpret_cast<f32x4 *>(d.c + (i / d.N) * d.ldc + (i % d.N));
      *o = d.accumulate ? *o + v : v;
    } else {
      const int64_t b = j - nv;
      float v = 0.f;
      for (int s = 0; s < d.slices; ++s) v += d.ws_bias[(int64_t)s * d.M + b];
      d.bias[b] = d.bias_accumulate ? d.bias[b] + v : v;
    }
  }
}

// ---- epilogue of one wave's TM x TN accumulator tiles: C/D layout of the 32 x 32 tile: column = lane & 31,
// row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5).
// Which terms exist (activation, row scale, residual) and whether the wave's block lies inside the matrix are uniform over the
// launch / the wave: they pick ONE straight-line instantiation. Written as one loop with the tests inside, every element was
// its own basic block of ~150 instructions (the tests, a 64-bit row product per pointer, an integer division for the row
// scale): 5 000 instructions per wave, 50 k cycles = 15 % of a workgroup's life at 21 024 x 2 496 x 832 (tools/gemmtrace.py).
// Here a row's pointers and its scale are formed once and serve the TN columns.
// SCALE / RES: 0 = absent, 1 = present, 2 = looked up at run time (the activation forms: their erf dominates the code anyway, and
// sixteen copies of it per kernel were most of this file's compile time)
template <int ACT, int SCALE, int RES, bool FULL, int TM, int TN>
__device__ __forceinline__ void epilogue_rows(const p2c_gemm_desc &d, const f32x16 (&acc)[TM][TN], int mw, int nw, int li, int lk) {
  int n[TN];
  bool nok[TN];
  float bias[TN];
#pragma unroll
  for (int b = 0; b < TN; ++b) {
    n[b] = nw + b * 32 + li;
    nok[b] = FULL || n[b] < d.N;
    bias[b] = (d.bias && nok[b]) ? d.bias[n[b]] : 0.f;
  }
  float *const zbase = (ACT == 1) ? d.aux_out : nullptr;
  const bool scale = SCALE == 1 || (SCALE == 2 && d.row_scale != nullptr), res = RES == 1 || (RES == 2 && d.residual != nullptr);
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = mw + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
      if (!FULL && m >= d.M) continue;
      float *const crow = d.c + (int64_t)m * d.ldc;
      const float *const rrow = res ? d.residual + (int64_t)m * d.ldr : nullptr;
      const float *const xrow = (ACT == 2) ? d.aux + (int64_t)m * d.ldaux : nullptr;
      float *const zrow = (ACT == 1 && zbase) ? zbase + (int64_t)m * d.ldaux : nullptr;
      const float rs = scale ? d.row_scale[(unsigned)m / (unsigned)d.rows_per_scale] : 1.f;
#pragma unroll
      for (int b = 0; b < TN; ++b) {
        if (!nok[b]) continue;
        float v = acc[a][b][r] + bias[b];
        if (ACT == 1) {
          if (zrow) zrow[n[b]] = v;
          v = gelu(v);
        } else if (ACT == 2) {
          v *= gelu_grad(xrow[n[b]]);
        }
        if (scale) v *= rs;
        if (res) v += rrow[n[b]];
        crow[n[b]] = v;
      }
    }
}
template <int ACT, bool FULL, int TM, int TN>
__device__ __forceinline__ void epilogue_terms(const p2c_gemm_desc &d, const f32x16 (&acc)[TM][TN], int mw, int nw, int li, int lk) {
  if (ACT != 0) {
    epilogue_rows<ACT, 2, 2, FULL>(d, acc, mw, nw, li, lk);
    return;
  }
  const bool scale = d.row_scale != nullptr, res = d.residual != nullptr;
  if (scale && res) epilogue_rows<ACT, 1, 1, FULL>(d, acc, mw, nw, li, lk);
  else if (scale) epilogue_rows<ACT, 1, 0, FULL>(d, acc, mw, nw, li, lk);
  else if (res) epilogue_rows<ACT, 0, 1, FULL>(d, acc, mw, nw, li, lk);
  else epilogue_rows<ACT, 0, 0, FULL>(d, acc, mw, nw, li, lk);
}
template <int TM, int TN>
__device__ __forceinline__ void epilogue(const p2c_gemm_desc &d, const f32x16 (&acc)[TM][TN], int m0, int n0, int wm, int wn, int li, int lk) {
  const int mw = m0 + wm * TM * 32, nw = n0 + wn * TN * 32;
  const bool full = mw + TM * 32 <= d.M && nw + TN * 32 <= d.N;
  if (d.act == 1) {
    if (full) epilogue_terms<1, true>(d, acc, mw, nw, li, lk);
    else epilogue_terms<1, false>(d, acc, mw, nw, li, lk);
  } else if (d.act == 2) {
    if (full) epilogue_terms<2, true>(d, acc, mw, nw, li, lk);
    else epilogue_terms<2, false>(d, acc, mw, nw, li, lk);
  } else {
    if (full) epilogue_terms<0, true>(d, acc, mw, nw, li, lk);
    else epilogue_terms<0, false>(d, acc, mw, nw, li, lk);
  }
}

// (An LDS double-buffered main loop -- the wave stores k-tile kt + 1 into the other slab pair and requests kt + 2 in the middle of
// the MFMAs of kt, one barrier per k-tile -- was measured twice and is slower: 945 vs 846 us at 21 024 x 2 496 x 832. LDS
// operations of a wave complete in order, so the operand reads behind the 32 stores wait for them anyway, and 66 KB of slabs
// leave two workgroups per CU instead of three.)
template <int BN, bool TRANS_B, bool VEC, bool FAST = false>
__global__ __launch_bounds__(NTH, BN == 128 ? 3 : (BN == 64 ? 5 : 6)) void gemm_kernel(const p2c_gemm_desc d) {
  constexpr int WM = (BN == 128) ? 2 : 4;            // waves along m
  constexpr int TM = (BN == 128) ? 2 : 1;            // 32 x 32 MFMA tiles per wave along m ...
  constexpr int TN = (BN == 128) ? 2 : BN / 32;      // ... and along n
  constexpr int LDB = BN + 1;
  __shared__ __attribute__((aligned(16))) float As[BK * LDA];
  __shared__ __attribute__((aligned(16))) float Bs[BK * (BN + 4)];
  const int n_tiles = (d.N + BN - 1) / BN;
  const int id = xcd_tile((int)blockIdx.x, ((d.M + BM - 1) / BM) * n_tiles);
  if (id >= ((d.M + BM - 1) / BM) * n_tiles) return;
  const int tm = id / n_tiles, tn = id % n_tiles;                                // neighbours share the A rows (L2)
  const int m0 = tm * BM, n0 = tn * BN;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave % WM, wn = wave / WM;
  const int li = lane & 31, lk = lane >> 5;
  constexpr int ldb_s = TRANS_B ? LDB : BN + 4;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

  SlabRegs ra, rb;
  const int nk = (d.K + BK - 1) / BK;
  SlabOff oa, ob;
  __amdgpu_buffer_rsrc_t rsa, rsb;
  int b_step = 0;
  if (FAST) {
    rsa = operand_rsrc(d.a, d.M, d.lda, d.K);
    oa = rows_offsets(d.lda, m0, d.M, BM, (int)threadIdx.x);
    if (TRANS_B) {
      rsb = operand_rsrc(d.b, d.N, d.ldb, d.K);
      ob = rows_offsets(d.ldb, n0, d.N, BN, (int)threadIdx.x);
      b_step = BK * 4;
    } else {
      rsb = operand_rsrc(d.b, d.K, d.ldb, d.N);
      ob = kn_offsets<BN>(d.ldb, n0, d.N, (int)threadIdx.x);
      b_step = (int)(BK * d.ldb * 4);
    }
  }
  auto fetch = [&](int kt) {
    if (FAST) {
      load_pieces<4>(rsa, oa, kt * (BK * 4), ra);
      load_pieces<TRANS_B ? BN / 32 : KnMap<BN>::PIECES>(rsb, ob, kt * b_step, rb);
      return;
    }
    load_rows<VEC>(d.a, d.lda, m0, d.M, kt * BK, d.K, ra, BM, (int)threadIdx.x);
    if (TRANS_B) load_rows<VEC>(d.b, d.ldb, n0, d.N, kt * BK, d.K, rb, BN, (int)threadIdx.x);
    else load_kn<BN, VEC>(d.b, d.ldb, kt * BK, d.K, n0, d.N, rb, (int)threadIdx.x);
  };
  auto commit = [&]() {
    store_rows_transposed(As, LDA, ra, BM, (int)threadIdx.x);
    if (TRANS_B) {
      store_rows_transposed(Bs, ldb_s, rb, BN, (int)threadIdx.x);
    } else {
      store_kn<BN>(Bs, ldb_s, rb, (int)threadIdx.x);
    }
  };
  GT(0);
  fetch(0);
  GT(1);
  commit();
  GT(2);
  __syncthreads();
  GT(3);
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) fetch(kt + 1);
    if (kt < 8) GT(4 + kt * 6);
    float af[2][TM], bf[2][TN];          // (fragments of the next k-step are read before this step's MFMAs: see gemm_tn_kernel)
    auto frag = [&](int ks, int buf) {
      const int k = ks * 2 + lk;
#pragma unroll
      for (int a = 0; a < TM; ++a) af[buf][a] = As[k * LDA + (wm * TM + a) * 32 + li];
#pragma unroll
      for (int b = 0; b < TN; ++b) bf[buf][b] = Bs[k * ldb_s + (wn * TN + b) * 32 + li];
    };
    frag(0, 0);
#pragma unroll
    for (int ks = 0; ks < BK / 2; ++ks) {
      if (ks + 1 < BK / 2) frag(ks + 1, (ks + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);           // (keep the reads of step ks + 1 ahead of the MFMAs of step ks)
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[ks & 1][a], bf[ks & 1][b], acc[a][b], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (kt < 8) GT(5 + kt * 6);
    __syncthreads();
    if (kt < 8) GT(6 + kt * 6);
    if (kt + 1 < nk) {
#ifdef P2C_GEMM_TRACE
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (kt < 8) GT(7 + kt * 6);
#endif
      commit();
      if (kt < 8) GT(8 + kt * 6);
      __syncthreads();
      if (kt < 8) GT(9 + kt * 6);
    }
  }
  GT(60);

  epilogue<TM, TN>(d, acc, m0, n0, wm, wn, li, lk);
  GT(61);
}

#ifdef P2C_GEMM_EXPERIMENTS   // the two measured-and-shelved variants below: make EXTRA=-DP2C_GEMM_EXPERIMENTS (a third of this file's compile time)
// ---- the same 128 x 64 tile on v_mfma_f32_16x16x4_f32 (round 3 experiment: P2C_GEMM_MI16=1): a wave's 32 x 64 block as 2 x 4
// tiles of 16 x 16 (eight f32x4 accumulators), k advances by 4 per step: six LDS dwords feed eight MFMAs of 32 cycles. Same
// FLOP per cycle as 32 x 32 x 2; what it tests is whether the instruction shape matters for the sustained rate (the library's
// kernels at these shapes are built on 16 x 16 x 4). Measured: it does not -- NT 21 024 x 2 496 x 832: 904 vs 868 us, 832-wide
// outputs 296 vs 274 us, cfg5 37.6 vs 37.0 ms. Kept as the opt-in it was measured as.
template <bool TRANS_B, bool VEC>
__global__ __launch_bounds__(NTH, 5) void gemm16_kernel(const p2c_gemm_desc d) {
  constexpr int BN = 64, LDB = BN + 1, ldb_s = TRANS_B ? LDB : BN + 4;
  constexpr bool FAST = false;
  __shared__ __attribute__((aligned(16))) float As[BK * LDA];
  __shared__ __attribute__((aligned(16))) float Bs[BK * (BN + 4)];
  const int n_tiles = (d.N + BN - 1) / BN;
  const int id = xcd_tile((int)blockIdx.x, ((d.M + BM - 1) / BM) * n_tiles);
  if (id >= ((d.M + BM - 1) / BM) * n_tiles) return;
  const int tm = id / n_tiles, tn = id % n_tiles, m0 = tm * BM, n0 = tn * BN;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, kg = lane >> 4;
  f32x4 acc[2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  SlabRegs ra, rb;
  const int nk = (d.K + BK - 1) / BK;
  SlabOff oa, ob;
  __amdgpu_buffer_rsrc_t rsa, rsb;
  int b_step = 0;
  if (FAST) {
    rsa = operand_rsrc(d.a, d.M, d.lda, d.K);
    oa = rows_offsets(d.lda, m0, d.M, BM, (int)threadIdx.x);
    if (TRANS_B) {
      rsb = operand_rsrc(d.b, d.N, d.ldb, d.K);
      ob = rows_offsets(d.ldb, n0, d.N, BN, (int)threadIdx.x);
      b_step = BK * 4;
    } else {
      rsb = operand_rsrc(d.b, d.K, d.ldb, d.N);
      ob = kn_offsets<BN>(d.ldb, n0, d.N, (int)threadIdx.x);
      b_step = (int)(BK * d.ldb * 4);
    }
  }
  auto fetch = [&](int kt) {
    if (FAST) {
      load_pieces<4>(rsa, oa, kt * (BK * 4), ra);
      load_pieces<TRANS_B ? BN / 32 : KnMap<BN>::PIECES>(rsb, ob, kt * b_step, rb);
      return;
    }
    load_rows<VEC>(d.a, d.lda, m0, d.M, kt * BK, d.K, ra, BM, (int)threadIdx.x);
    if (TRANS_B) load_rows<VEC>(d.b, d.ldb, n0, d.N, kt * BK, d.K, rb, BN, (int)threadIdx.x);
    else load_kn<BN, VEC>(d.b, d.ldb, kt * BK, d.K, n0, d.N, rb, (int)threadIdx.x);
  };
  auto commit = [&]() {
    store_rows_transposed(As, LDA, ra, BM, (int)threadIdx.x);
    if (TRANS_B) store_rows_transposed(Bs, ldb_s, rb, BN, (int)threadIdx.x);
    else store_kn<BN>(Bs, ldb_s, rb, (int)threadIdx.x);
  };
  fetch(0);
  commit();
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) fetch(kt + 1);
    float af[2][2], bf[2][4];
    auto frag = [&](int ks, int buf) {
      const int k = ks * 4 + kg;
#pragma unroll
      for (int a = 0; a < 2; ++a) af[buf][a] = As[k * LDA + wave * 32 + a * 16 + li];
#pragma unroll
      for (int b = 0; b < 4; ++b) bf[buf][b] = Bs[k * ldb_s + b * 16 + li];
    };
    frag(0, 0);
#pragma unroll
    for (int ks = 0; ks < BK / 4; ++ks) {
      if (ks + 1 < BK / 4) frag(ks + 1, (ks + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[ks & 1][a], bf[ks & 1][b], acc[a][b], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    if (kt + 1 < nk) {
      commit();
      __syncthreads();
    }
  }
  // C/D layout of the 16 x 16 tile: column = lane & 15, row = 4 (lane >> 4) + reg
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const int n = n0 + b * 16 + li;
    if (n >= d.N) continue;
    const float bias = d.bias ? d.bias[n] : 0.f;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wave * 32 + a * 16 + 4 * kg + r;
        if (m >= d.M) continue;
        float v = acc[a][b][r] + bias;
        if (d.act == 1) {
          if (d.aux_out) d.aux_out[(int64_t)m * d.ldaux + n] = v;
          v = gelu(v);
        } else if (d.act == 2) {
          v *= gelu_grad(d.aux[(int64_t)m * d.ldaux + n]);
        }
        if (d.row_scale) v *= d.row_scale[m / d.rows_per_scale];
        if (d.residual) v += d.residual[(int64_t)m * d.ldr + n];
        d.c[(int64_t)m * d.ldc + n] = v;
      }
  }
}

// ---- producer / consumer form (round 3 experiment, OFF by default: P2C_GEMM_WS=1): eight waves per workgroup. Waves 4..7 only move data -- global loads of k-tile kt + 2
// into registers, the registers of kt + 1 into the OTHER slab pair -- and waves 0..3 only multiply out of the current pair; one
// workgroup barrier per k-tile hands a pair over. In the four-wave kernel above every wave does both jobs in turn and the
// workgroup spends more than half of a k-tile outside its MFMA phase (tools/gemmtrace.py), which co-resident workgroups do not
// cover for each other; a wave's own LDS stores cannot overlap its own operand reads either (in-order LDS queue: the measured
// double-buffered variant). Here the stores and the operand reads belong to different waves.
// Measured: correct (tests/test_gemm_gpu.py under P2C_GEMM_WS=1) and NOT faster -- 21 024 x 2 496 x 832 NT 888 vs 868 us, NN 832 x
// 2 496: 826..851 vs 759 us, cfg5 38.1..38.8 vs 37.0 ms -- with one or with three k-tiles of load look-ahead: neither the load
// latency nor the store / barrier phases of the four-wave kernel are what holds the MFMA rate at ~65 % of peak.
template <int BN, bool TRANS_B, bool VEC>
__global__ __launch_bounds__(2 * NTH, 2) void gemm_ws_kernel(const p2c_gemm_desc d) {
  constexpr int WM = (BN == 128) ? 2 : 4, TM = (BN == 128) ? 2 : 1, TN = (BN == 128) ? 2 : BN / 32;
  constexpr int LDB = BN + 1, ldb_s = TRANS_B ? LDB : BN + 4;
  constexpr int A_SZ = (BK * LDA + 3) & ~3, B_SZ = BK * (BN + 4);
  __shared__ __attribute__((aligned(16))) float As2[2 * A_SZ];
  __shared__ __attribute__((aligned(16))) float Bs2[2 * B_SZ];
  const int n_tiles = (d.N + BN - 1) / BN;
  const int id = xcd_tile((int)blockIdx.x, ((d.M + BM - 1) / BM) * n_tiles);
  if (id >= ((d.M + BM - 1) / BM) * n_tiles) return;
  const int tm = id / n_tiles, tn = id % n_tiles;
  const int m0 = tm * BM, n0 = tn * BN;
  const int nk = (d.K + BK - 1) / BK;
  const bool producer = threadIdx.x >= NTH;                 // (wave-uniform)
  if (producer) {
    const int tid = (int)threadIdx.x - NTH;
    // k-tile j travels through register set j % D: its loads are issued D k-tiles before it is stored to LDS (a k-tile of the
    // consumers is only ~2-4 k cycles: one k-tile of look-ahead is less than a loaded memory round trip)
    constexpr int D = 3;
    SlabRegs ra[D], rb[D];
    auto fetch = [&](int kt, int slot) {
      load_rows<VEC>(d.a, d.lda, m0, d.M, kt * BK, d.K, ra[slot], BM, tid);
      if (TRANS_B) load_rows<VEC>(d.b, d.ldb, n0, d.N, kt * BK, d.K, rb[slot], BN, tid);
      else load_kn<BN, VEC>(d.b, d.ldb, kt * BK, d.K, n0, d.N, rb[slot], tid);
    };
    auto commit = [&](int buf, int slot) {
      store_rows_transposed(As2 + buf * A_SZ, LDA, ra[slot], BM, tid);
      if (TRANS_B) store_rows_transposed(Bs2 + buf * B_SZ, ldb_s, rb[slot], BN, tid);
      else store_kn<BN>(Bs2 + buf * B_SZ, ldb_s, rb[slot], tid);
    };
    fetch(0, 0);
    commit(0, 0);
#pragma unroll
    for (int j = 1; j <= D; ++j)
      if (j < nk) fetch(j, j % D);
    __syncthreads();                                        // slab pair 0 is ready
    for (int kt0 = 0; kt0 < nk; kt0 += D) {
#pragma unroll
      for (int u = 0; u < D; ++u) {                         // (unrolled: the register set of a k-tile is a compile-time index)
        const int kt = kt0 + u;
        if (kt < nk) {
          if (kt + 1 < nk) {
            commit((kt + 1) & 1, (u + 1) % D);              // (the consumers left this pair at the previous barrier)
            if (kt + 1 + D < nk) fetch(kt + 1 + D, (u + 1) % D);
          }
          __syncthreads();
        }
      }
    }
    return;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave % WM, wn = wave / WM, li = lane & 31, lk = lane >> 5;
  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const float *As = As2 + (kt & 1) * A_SZ, *Bs = Bs2 + (kt & 1) * B_SZ;
    float af[2][TM], bf[2][TN];
    auto frag = [&](int ks, int buf) {
      const int k = ks * 2 + lk;
#pragma unroll
      for (int a = 0; a < TM; ++a) af[buf][a] = As[k * LDA + (wm * TM + a) * 32 + li];
#pragma unroll
      for (int b = 0; b < TN; ++b) bf[buf][b] = Bs[k * ldb_s + (wn * TN + b) * 32 + li];
    };
    frag(0, 0);
#pragma unroll
    for (int ks = 0; ks < BK / 2; ++ks) {
      if (ks + 1 < BK / 2) frag(ks + 1, (ks + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[ks & 1][a], bf[ks & 1][b], acc[a][b], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
  }
  epilogue<TM, TN>(d, acc, m0, n0, wm, wn, li, lk);
}

#endif  // P2C_GEMM_EXPERIMENTS

template <int BN, bool TRANS_B>
static void launch(const p2c_gemm_desc &d, bool vec, hipStream_t s) {
  const unsigned grid = xcd_grid((int64_t)((d.M + BM - 1) / BM) * ((d.N + BN - 1) / BN));
#ifdef P2C_GEMM_EXPERIMENTS
  static const int ws_mode = getenv("P2C_GEMM_WS") ? atoi(getenv("P2C_GEMM_WS")) : 0;
  static const int mi16 = getenv("P2C_GEMM_MI16") ? atoi(getenv("P2C_GEMM_MI16")) : 0;
  if (mi16 && vec && BN == 64) {
    hipLaunchKernelGGL((gemm16_kernel<TRANS_B, true>), dim3(grid), dim3(NTH), 0, s, d);
    return;
  }
  if (ws_mode && vec && d.K >= 256) {           // producer / consumer form: deep products
    hipLaunchKernelGGL((gemm_ws_kernel<BN, TRANS_B, true>), dim3(grid), dim3(2 * NTH), 0, s, d);
    return;
  }
#endif
  static const int no_fast = getenv("P2C_GEMM_NO_FAST") ? atoi(getenv("P2C_GEMM_NO_FAST")) : 0;      // (A/B timing)
  const int64_t a_bytes = ((int64_t)(d.M - 1) * d.lda + d.K) * 4;
  const int64_t b_bytes = (TRANS_B ? (int64_t)(d.N - 1) * d.ldb + d.K : (int64_t)(d.K - 1) * d.ldb + d.N) * 4;
  const bool fast = vec && !no_fast && d.K % BK == 0 && a_bytes < OOB_OFF && b_bytes < OOB_OFF;
  if (fast) hipLaunchKernelGGL((gemm_kernel<BN, TRANS_B, true, true>), dim3(grid), dim3(NTH), 0, s, d);
  else if (vec) hipLaunchKernelGGL((gemm_kernel<BN, TRANS_B, true>), dim3(grid), dim3(NTH), 0, s, d);
  else hipLaunchKernelGGL((gemm_kernel<BN, TRANS_B, false>), dim3(grid), dim3(NTH), 0, s, d);
}

}  // namespace p2c_gemm_impl

// Experiment knobs (tools/tn_bench.py, tools/gemm_bench.py): read from the environment ONCE -- dense() routes every fp32 2-D linear
// here, and a getenv per eager launch is host time; a workspace sized under one value and a launch under another would also
// write past the workspace. p2c_gemm_reload_env() re-reads them (the sweep tools call it after changing a variable).
static int g_env_tn_bn = -1, g_env_tn_slices = -1, g_env_bn = -1;
static bool g_env_read = false;
static void gemm_env() {
  if (g_env_read) return;
  const char *e = getenv("P2C_GEMM_TN_BN");
  g_env_tn_bn = e ? atoi(e) : 0;
  e = getenv("P2C_GEMM_TN_SLICES");
  g_env_tn_slices = e ? atoi(e) : 0;
  e = getenv("P2C_GEMM_BN");
  g_env_bn = e ? atoi(e) : 0;
  g_env_read = true;
}
extern "C" P2C_API void p2c_gemm_reload_env(void) {
  g_env_read = false;
  gemm_env();
}
static int tn_bn(int N) {
  int bn = N > 32 ? 64 : 32;                          // (64: measured ahead of 128 at 832 x 832, equal elsewhere)
  gemm_env();
  if (g_env_tn_bn == 32 || g_env_tn_bn == 64 || g_env_tn_bn == 128) bn = g_env_tn_bn;     // (experiments)
  return bn;
}
static int tn_slices(int M, int N, int K) {
  using namespace p2c_gemm_impl;
  const int bn = tn_bn(N);
  const int tiles = ((M + BM - 1) / BM) * ((N + bn - 1) / bn);
  int max_s = K / (16 * BK);                             // at least 16 k-tiles per slice
  max_s = max_s < 1 ? 1 : (max_s > 1024 ? 1024 : max_s);
  gemm_env();
  if (g_env_tn_slices >= 1) return g_env_tn_slices > max_s ? max_s : g_env_tn_slices;     // (experiments: tools/tn_bench.py sweeps it)
  if (tiles < 64) {
    // a skinny output over very many rows (the spatial blocks' 96 x 32 from 546 624 rows) is a streaming pass over A and B:
    // what counts is enough workgroups in flight to pull the rows in -- about four per CU
    const int s = 1024 / tiles;
    return s < 1 ? 1 : (s > max_s ? max_s : s);
  }
  // Otherwise the launch takes as long as the busiest CU. A CU runs its workgroups five at a time (launch bounds), and what a
  // k-tile step costs depends on how many share the MFMA pipes (timeline of a workgroup, tools/gemmtrace.py: alone a wave
  // spends ~5.7 k cycles per k-tile, 2 k of them in MFMAs; five together take ~11.6 k per step and keep the pipes ~88 % busy):
  // equal workgroups started together stay in step, so n of them on a CU cost floor(n / 5) full steps and one step of the
  // remainder -- a remainder of one is the worst buy (half a full step for a fifth of its work). On top of that every slice
  // writes and re-reads its slab (8 M N bytes at ~4 TB/s through the second launch). The slice count minimises the sum.
  if (max_s > 64) max_s = 64;
  static const double step_cost[6] = {0.0, 5.7e3, 5.7e3, 7.2e3, 9.4e3, 11.6e3};      // cycles per k-tile step, by co-resident count
  const double finish_cycles_per_slice = 8.0 * M * N / 4.0e12 * 2.35e9;
  int best = 1;
  double best_cost = 1e300;
  for (int s = 1; s <= max_s; ++s) {
    const int n_max = (tiles * s + 255) / 256;
    const double kt = (double)((K + (int64_t)s * BK - 1) / ((int64_t)s * BK));
    const double cost = kt * ((n_max / 5) * step_cost[5] + step_cost[n_max % 5]) + finish_cycles_per_slice * s;
    if (cost < best_cost) best_cost = cost, best = s;
  }
  return best;
}
extern "C" int64_t p2c_gemm_tn_workspace_floats(int32_t M, int32_t N, int32_t K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  return (int64_t)tn_slices(M, N, K) * ((int64_t)M * N + M);
}
extern "C" int p2c_gemm_tn(const float *a, int64_t lda, const float *b, int64_t ldb, float *c, int64_t ldc, int32_t M, int32_t N,
                           int32_t K, int32_t accumulate, const float *row_scale, int32_t rows_per_scale, float *bias_out,
                           float *workspace, void *stream_) {
  using namespace p2c_gemm_impl;
  if (!a || !b || !c || !workspace) return P2C_E_NULL;
  if (M <= 0 || N <= 0 || K <= 0 || lda < M || ldb < N || ldc < N) return P2C_E_SHAPE;
  if (row_scale && rows_per_scale <= 0) return P2C_E_SHAPE;
  TnArgs d;
  d.a = a, d.b = b, d.ws = workspace, d.c = c, d.lda = lda, d.ldb = ldb, d.ldc = ldc, d.M = M, d.N = N, d.K = K;
  d.slices = tn_slices(M, N, K);
  d.k_chunk = (((K + d.slices - 1) / d.slices) + BK - 1) / BK * BK;
  d.accumulate = accumulate & 1, d.bias_accumulate = (accumulate >> 1) & 1;
  d.row_scale = row_scale, d.rows_per_scale = rows_per_scale;
  d.bias = bias_out, d.ws_bias = workspace + (int64_t)d.slices * M * N;
  hipStream_t s = (hipStream_t)stream_;
  auto al = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  const bool vec = al(a) && al(b) && lda % 4 == 0 && ldb % 4 == 0 && M % 4 == 0 && N % 4 == 0;
  const int bn = tn_bn(N);
  const dim3 grid(xcd_grid((int64_t)((M + BM - 1) / BM) * ((N + bn - 1) / bn) * d.slices));
  static const int no_fast = getenv("P2C_GEMM_NO_FAST") ? atoi(getenv("P2C_GEMM_NO_FAST")) : 0;      // (A/B timing)
  const bool fast = vec && !no_fast && K % BK == 0 && ((int64_t)(K - 1) * lda + M) * 4 < OOB_OFF && ((int64_t)(K - 1) * ldb + N) * 4 < OOB_OFF;
#define P2C_TN(BN_)                                                                         \
  do {                                                                                      \
    if (fast) hipLaunchKernelGGL((gemm_tn_kernel<BN_, true, true>), grid, dim3(NTH), 0, s, d);   \
    else if (vec) hipLaunchKernelGGL((gemm_tn_kernel<BN_, true>), grid, dim3(NTH), 0, s, d);     \
    else hipLaunchKernelGGL((gemm_tn_kernel<BN_, false>), grid, dim3(NTH), 0, s, d);        \
  } while (0)
  if (bn == 128) P2C_TN(128);
  else if (bn == 64) P2C_TN(64);
  else P2C_TN(32);
#undef P2C_TN
  const int64_t total = (int64_t)M * N;
  const int64_t n_all = total + (bias_out ? M : 0);
  if (N % 4 == 0 && ldc % 4 == 0 && al(c) && al(workspace)) {
    const int64_t n_thr = total / 4 + (bias_out ? M : 0);
    if (d.slices <= 8)
      hipLaunchKernelGGL(tn_finish_vec_kernel, dim3((unsigned)((n_thr + 255) / 256 < 65536 ? (n_thr + 255) / 256 : 65536)), dim3(256), 0, s, d);
    else
      hipLaunchKernelGGL(tn_finish_vec8_kernel, dim3((unsigned)((n_thr + 31) / 32 < 65536 ? (n_thr + 31) / 32 : 65536)), dim3(256), 0, s, d);
  } else {
    hipLaunchKernelGGL(tn_finish_kernel, dim3((unsigned)((n_all + 31) / 32 < 65536 ? (n_all + 31) / 32 : 65536)), dim3(256), 0, s, d);
  }
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

#ifdef P2C_GEMM_TRACE
extern "C" __attribute__((visibility("default"))) int p2c_debug_gemm_trace_block(int block) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(p2c_gemm_impl::g_gemm_trace_block), &block, sizeof(int));
}
extern "C" __attribute__((visibility("default"))) int p2c_debug_gemm_trace(unsigned long long *out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(p2c_gemm_impl::g_gemm_trace), sizeof(unsigned long long) * 65);
}
#endif

extern "C" int p2c_gemm(const p2c_gemm_desc *desc, void *stream_) {
  using namespace p2c_gemm_impl;
  if (!desc || !desc->a || !desc->b || !desc->c) return P2C_E_NULL;
  const p2c_gemm_desc d = *desc;
  if (d.M <= 0 || d.N <= 0 || d.K <= 0 || d.lda < d.K || d.ldc < d.N) return P2C_E_SHAPE;
  if (d.ldb < (d.trans_b ? d.K : d.N)) return P2C_E_SHAPE;
  if (d.act < 0 || d.act > 2 || (d.act == 2 && !d.aux) || ((d.aux || d.aux_out) && d.ldaux < d.N)) return P2C_E_ENUM;
  if (d.row_scale && d.rows_per_scale <= 0) return P2C_E_SHAPE;
  if (d.residual && d.ldr < d.N) return P2C_E_SHAPE;
  if ((int64_t)((d.M + BM - 1) / BM) * ((d.N + 31) / 32) > 0x7fffffffll) return P2C_E_SHAPE;
  hipStream_t s = (hipStream_t)stream_;
  // 16-byte loads need 16-byte aligned rows: every leading dimension a multiple of 4 floats, bases aligned, K (NT) / N (NN) too
  auto al = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  const bool vec = al(d.a) && al(d.b) && d.lda % 4 == 0 && d.ldb % 4 == 0 && d.K % 4 == 0 && (d.trans_b || d.N % 4 == 0);
  // Column tile: 64 (four waves of 32 x 64 side by side) -- measured ahead of the 2 x 2 arrangement of 64 x 64 waves at every
  // wide shape of cfg5 (NT 832-wide outputs +6..10 %, equal at 2 496; 120 VGPRs -> four workgroups per CU), 32 for the NN form
  // (another 3..5 %) and for N <= 32. Small problems (the 8 192-row projections around the Seq2Seq recurrences) take the
  // narrowest tile to reach more CUs. P2C_GEMM_BN forces one (experiments; the 128-column instantiation stays reachable).
  int bn = d.N > 32 ? 64 : 32;
  if (!d.trans_b && d.K >= 256 && d.K < 2048) bn = 32;      // (with the buffer-load fetch the 64-column tile is ahead again from K ~ 2 000: 687 vs 714 us at 832 x 2 496)
  const int64_t row_tiles = (d.M + BM - 1) / BM;
  while (bn > 32 && row_tiles * ((d.N + bn - 1) / bn) < 512) bn >>= 1;
  // shallow products (K <= 128: the spatial blocks' 546 624 x 96 x 32) stream A and C once and are bound by that: the tile that
  // pads N least wins (N = 96: three 32-column tiles instead of one 128-column tile a quarter empty)
  if (d.K <= 128)
    for (int cand = bn >> 1; cand >= 32; cand >>= 1)
      if ((d.N + cand - 1) / cand * cand < (d.N + bn - 1) / bn * bn) bn = cand;
  gemm_env();
  if (g_env_bn == 32 || g_env_bn == 64 || g_env_bn == 128) bn = g_env_bn;     // (experiments)
  if (d.trans_b) {
    if (bn == 128) launch<128, true>(d, vec, s);
    else if (bn == 64) launch<64, true>(d, vec, s);
    else launch<32, true>(d, vec, s);
  } else {
    if (bn == 128) launch<128, false>(d, vec, s);
    else if (bn == 64) launch<64, false>(d, vec, s);
    else launch<32, false>(d, vec, s);
  }
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}
