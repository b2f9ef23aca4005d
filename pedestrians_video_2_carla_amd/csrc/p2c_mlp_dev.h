// p2c_mlp_dev.h -- device functions of the fused small MLP (shared by p2c_mlp.hip and p2c_train.hip).
// See p2c_mlp.hip for the structure ("cooperative 16-sample tile").
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "../../include/p2c.h"

namespace p2c_mlp {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int MAXW = 160;        // widest layer (padded to 16)
constexpr int TS = 16;           // samples per tile
constexpr int TP = 17;           // LDS pitch of an activation row (odd: the dW phase reads [row = lane&15][sample])
#ifndef P2C_MLP_WAVES
#define P2C_MLP_WAVES 8
#endif
constexpr int WAVES = P2C_MLP_WAVES;     // waves per workgroup
constexpr int MAX_SLOTS = 96 / WAVES;    // dW tiles per wave held in accumulators (96 tiles in all)
constexpr int NL = P2C_MLP_MAX_LAYERS;

struct MlpArgs {
  int32_t n_layers;
  int32_t dims[NL + 1];
  const float *W[NL];
  const float *b[NL];
  float *gW[NL];
  float *gb[NL];
  const float *x;
  float *y;
  const float *gy;
  float *partials;
  int64_t N;
  int32_t n_params, n_tiles_w;   // total parameters; total 16x16 tiles of the augmented weight gradients
  int32_t ld[NL];                // LDS row pitch of the padded image of W_l (floats), == 2 (mod 4), >= pad16(n_in)
  int32_t w_off[NL];             // offset of that image (floats)
  int32_t w_total;               // floats of all images
  int32_t h_off[NL + 1];         // row offset of H_l^T (l = 0..L) in the activation area; rows = pad16(dims[l]) + 16
  int32_t act_rows;              // rows of H_0 .. H_L
  int32_t vec_x, vec_y, vec_gy;  // 16-byte row loads/stores are legal (row length % 4 == 0 and base aligned)
  float *w_image;                // packed, zero-padded weight images in HBM (w_total floats), written by mlp_pack_kernel
  int32_t tab[2 * MAX_SLOTS * WAVES];   // dW tile t: LDS float offsets (relative to H) of its G rows and its H rows
  // split weight gradient (small batches): per sample tile the backward leaves H_1..H_{L-1} and G_1..G_{L-1} transposed
  // ([row][16 samples]) in `factors`; row f_off[l] (+ f_half for G) is the first row of layer l, f_rows = 2 * f_half
  float *factors;
  int32_t f_off[NL + 1], f_half, f_rows;
  // saved activations (many sample tiles per workgroup): the forward leaves H_1..H_{L-1} of every sample tile in the same
  // transposed layout (f_half rows of 16 samples), the backward loads them instead of recomputing
  float *saved;
  int32_t fw_rows_a, fw_rows_b;    // wave-per-tile forward: rows of the two ping-pong activation buffers of a wave
};

__host__ __device__ constexpr int pad16(int n) { return (n + 15) & ~15; }
// layout rules shared by the host (fill) and by the compile-time shapes below
__host__ __device__ constexpr int ld_of(int n_in) {          // LDS pitch of the image of a layer with n_in inputs
  int ld = ((((n_in + 1 + 3) >> 2) + 3) & ~3) * 4;          // k extent incl. the bias column, in 4-step blocks
  if (ld < pad16(n_in)) ld = pad16(n_in);                    // dgrad reads columns up to pad16(n_in)
  while ((ld & 3) != 2) ++ld;                                // pitch == 2 (mod 4)
  return ld;
}
__host__ __device__ constexpr int img_rows_of(int n_out) { return (n_out + 16) & ~15; }   // rows 0..n_out (unit row), padded
__host__ __device__ constexpr int act_rows_of(int n) { return pad16(n) + 16; }            // ones row / k rounding past pad16

// Shape providers. DynShape reads the layer geometry from the kernel arguments (any MLP the ABI accepts); StaticShape
// carries it in the type, so that with the layer loops unrolled every pitch, offset and trip count is an immediate:
// the integer address arithmetic that otherwise dominates these latency-bound kernels disappears.
struct DynShape {
  static constexpr bool kStatic = false;
  const MlpArgs &a;
  __device__ explicit DynShape(const MlpArgs &args) : a(args) {}
  __device__ int n_layers() const { return a.n_layers; }
  __device__ int dims(int l) const { return a.dims[l]; }
  __device__ int ld(int l) const { return a.ld[l]; }
  __device__ int w_off(int l) const { return a.w_off[l]; }
  __device__ int w_total() const { return a.w_total; }
  __device__ int h_off(int l) const { return a.h_off[l]; }
  __device__ int act_rows() const { return a.act_rows; }
};
template <int... D>
struct StaticShape {
  static constexpr bool kStatic = true;
  static constexpr int NLAY = (int)sizeof...(D) - 1;
  __device__ explicit StaticShape(const MlpArgs &) {}
  __device__ StaticShape() {}
  __host__ __device__ static constexpr int dim_at(int l) {
    constexpr int d[] = {D...};
    return d[l];
  }
  __host__ __device__ static constexpr int n_layers() { return NLAY; }
  __host__ __device__ static constexpr int dims(int l) { return dim_at(l); }
  __host__ __device__ static constexpr int ld(int l) { return ld_of(dim_at(l)); }
  __host__ __device__ static constexpr int w_off(int l) {
    int o = 0;
    for (int i = 0; i < l; ++i) o += img_rows_of(dim_at(i + 1)) * ld_of(dim_at(i));
    return o;
  }
  __host__ __device__ static constexpr int w_total() { return (w_off(NLAY) + 3) & ~3; }
  __host__ __device__ static constexpr int h_off(int l) {
    int r = 0;
    for (int i = 0; i < l; ++i) r += act_rows_of(dim_at(i));
    return r;
  }
  __host__ __device__ static constexpr int act_rows() { return h_off(NLAY + 1); }
  static bool matches(const MlpArgs &a) {
    if (a.n_layers != NLAY) return false;
    for (int l = 0; l <= NLAY; ++l)
      if (a.dims[l] != dim_at(l)) return false;
    return true;
  }
};
// the LinearAE of the reference (linear_ae.py:25-59): 26 joints x 2 in; 26 x {6, 3, 2} out (pose_changes 6-D, absolute_loc,
// pose_2d)
using LinearAE156 = StaticShape<52, 26, 13, 6, 39, 78, 156>;
using LinearAE78 = StaticShape<52, 26, 13, 6, 19, 39, 78>;
using LinearAE52 = StaticShape<52, 26, 13, 6, 13, 26, 52>;

// layer loop: fully unrolled for a static shape, a plain loop otherwise
template <class S, class F>
__device__ __forceinline__ void for_layers(const S &shape, int first, int last_exclusive, F &&f) {
  if constexpr (S::kStatic) {
#pragma unroll
    for (int l = 0; l < S::NLAY; ++l)
      if (l >= first && l < last_exclusive) f(l);
  } else {
#pragma unroll 1
    for (int l = first; l < last_exclusive; ++l) f(l);
  }
}
template <class S, class F>
__device__ __forceinline__ void for_layers_down(const S &shape, int first, int last_inclusive, F &&f) {   // first >= last
  if constexpr (S::kStatic) {
#pragma unroll
    for (int l = S::NLAY - 1; l >= 0; --l)
      if (l <= first && l >= last_inclusive) f(l);
  } else {
#pragma unroll 1
    for (int l = first; l >= last_inclusive; --l) f(l);
  }
}

// Workgroup barrier that orders LDS traffic only: __syncthreads() carries a full workgroup-scope release fence, which
// on gfx9 means s_waitcnt vmcnt(0) -- every global load in flight (the rest of the weight image, the next tile's rows)
// would have to land before any wave may pass. All data the waves exchange lives in LDS, so the fence is restricted to
// the local address space and the loads keep streaming across the barriers.
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// Workgroup copy of the packed image HBM -> LDS, split into ISSUE (all global loads of the thread in flight at once:
// the image was written by another XCD's pack kernel, so every load is an L2 miss of a few thousand cycles -- one such
// latency is paid, not one per batch) and COMMIT (LDS stores). Round u of a thread covers float4 [u * NTH, (u+1) * NTH)
// of the image, i.e. the rounds are in layer order: with a static shape the rounds a layer needs are committed right
// before that layer's barrier, so layer 0 starts as soon as the first 8 KB have landed while the big last layers (60 %
// of the bytes) are still on their way. Inside a round every workgroup starts at a different offset so that the 256 CUs
// do not ask for the same line at the same moment.
constexpr int NTH = 64 * WAVES;
static_assert((NTH & (NTH - 1)) == 0, "threads per workgroup must be a power of two");
constexpr int STAGE_U = 12;   // float4 per thread and round (one round covers 96 KB with 512 threads)
struct ImageRegs {
  f32x4 v[STAGE_U];
};
__device__ __forceinline__ int stage_index(int u, int total4, int base, bool &ok) {
  const int rot = (int)((blockIdx.x * 40503u) & (NTH - 1));
  const int i = base + u * NTH + ((threadIdx.x + rot) & (NTH - 1));
  ok = i < total4;
  return ok ? i : 0;
}
__device__ __forceinline__ void stage_issue(const float *w_image, int total4, ImageRegs &r, int base = 0, int u0 = 0,
                                            int u1 = STAGE_U) {
  const f32x4 *src = reinterpret_cast<const f32x4 *>(w_image);
#pragma unroll
  for (int u = 0; u < STAGE_U; ++u) {
    if (u < u0 || u >= u1) continue;
    bool ok;
    const int i = stage_index(u, total4, base, ok);
    r.v[u] = src[i];
  }
}
// rounds [u0, u1) (compile-time after unrolling) -> LDS
__device__ __forceinline__ void stage_commit(int total4, const ImageRegs &r, float *dst, int base = 0, int u0 = 0,
                                             int u1 = STAGE_U) {
  f32x4 *d4 = reinterpret_cast<f32x4 *>(dst);
#pragma unroll
  for (int u = 0; u < STAGE_U; ++u) {
    if (u < u0 || u >= u1) continue;
    bool ok;
    const int i = stage_index(u, total4, base, ok);
    if (ok) d4[i] = r.v[u];
  }
}
// images larger than one round (wide custom MLPs): the remaining rounds
__device__ __forceinline__ void stage_rest(const float *w_image, int total4, ImageRegs &r, float *dst) {
  const int per_round = STAGE_U * NTH;
  for (int base = per_round; base < total4; base += per_round) {
    stage_issue(w_image, total4, r, base);
    stage_commit(total4, r, dst, base);
  }
}
// Static shapes spread the ISSUE as well: the vector-memory path of a CU takes 64 B per cycle, so the ~100 KB a workgroup
// asks for at once keep every wave stuck in its load instructions for > 2 000 cycles before the first MFMA. The prologue
// asks only for what layers 0 and 1 read; slot j (= right behind the barrier of forward layer j, j = 0, 1, 2) asks for
// the next four rounds: they land while the small middle layers compute.
template <class S>
__host__ __device__ constexpr int issue_mark(int slot);   // rounds [issue_mark(j), issue_mark(j + 1)) go out in slot j - 1
// rounds that hold the images of layers 0..l
template <class S>
__host__ __device__ constexpr int static_layers() {   // 0 for the dynamic shape
  if constexpr (S::kStatic) return S::NLAY;
  else return 0;
}
template <class S>
__host__ __device__ constexpr int rounds_upto(int l) {
  if constexpr (!S::kStatic) return STAGE_U;
  else {
  const int end4 = (l + 1 >= S::n_layers()) ? (S::w_total() >> 2) : ((S::w_off(l + 1) + 3) >> 2);
  const int r = (end4 + NTH - 1) / NTH;
  return r > STAGE_U ? STAGE_U : r;
  }
}

#ifdef P2C_MLP_TRACE   // developer build only (tools/mlptrace.py): shader-clock stamps of workgroup 0
static __device__ unsigned long long g_trace[2][40];
#ifndef P2C_MLP_TRACE_BLOCK
#define P2C_MLP_TRACE_BLOCK 0
#endif
#define TR(k, i)                                                                      \
  do {                                                                                \
    if (blockIdx.x == P2C_MLP_TRACE_BLOCK && threadIdx.x == 0) {                                        \
      g_trace[k][i] = __builtin_readcyclecounter();                                   \
      if ((i) == 0 || (i) == 39) g_trace[k][(i) == 0 ? 38 : 37] = wall_clock64();    \
    }                                                                                 \
  } while (0)
#else
#define TR(k, i)
#endif

struct Lane {
  int lane, c, g, wave;   // c = lane & 15 (sample / column), g = lane >> 4
};

// ---- arithmetic of one group of FOUR k-steps (K = 16) of a 16x16 tile --------------------------------------------------------
// P2C_PREC_F32 (default everywhere): four v_mfma_f32_16x16x4_f32 -- bit-for-bit an fmaf chain in k order.
// P2C_PREC_BF16: the same 4 + 4 operand floats rounded to bf16 (RNE, v_cvt_pk_bf16_f32), ONE v_mfma_f32_16x16x16_bf16 (fp32
//   accumulate). Lane (row/col c, group g) feeds k = 4 u + g as element u of its 4-vector on BOTH operands: the instruction sums
//   over all 16 (group, element) pairs, so any pairing that is the same for A and B is a valid K order -- the operand loads of
//   the fp32 loop are reused as they are.
// P2C_PREC_BF16X3: split-bf16 -- x = hi + lo with hi = bf16(x), lo = bf16(x - hi); hi*hi + hi*lo + lo*hi (three MFMAs, the
//   lo*lo term ~2^-18 relative is dropped): ~fp32-grade products at 3/8 of the fp32 MFMA cycles.
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
struct Bf16Pair {
  s16x4 hi, lo;
};
template <int P>
__device__ __forceinline__ Bf16Pair to_bf16(const float (&v)[4]) {
  Bf16Pair r;
  bf16x4 hi;
#pragma unroll
  for (int i = 0; i < 4; ++i) hi[i] = (__bf16)v[i];
  r.hi = __builtin_bit_cast(s16x4, hi);
  r.lo = r.hi;
  if constexpr (P == P2C_PREC_BF16X3) {
    bf16x4 lo;
#pragma unroll
    for (int i = 0; i < 4; ++i) lo[i] = (__bf16)(v[i] - (float)hi[i]);
    r.lo = __builtin_bit_cast(s16x4, lo);
  }
  return r;
}
template <int P>
__device__ __forceinline__ f32x4 mfma_k16(const float (&a)[4], const float (&b)[4], f32x4 c) {
  if constexpr (P == P2C_PREC_F32) {
#pragma unroll
    for (int u = 0; u < 4; ++u) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[u], c, 0, 0, 0);
    return c;
  } else {
    const Bf16Pair A = to_bf16<P>(a), B = to_bf16<P>(b);
    if constexpr (P == P2C_PREC_BF16X3) {     // small terms first
      c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(A.lo, B.hi, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(A.hi, B.lo, c, 0, 0, 0);
    }
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(A.hi, B.hi, c, 0, 0, 0);
  }
}

// 16 consecutive rows of a row-major [N][n] HBM matrix (one contiguous span) -> LDS transposed dst[k * TP + sample];
// rows beyond N read as zero. ISSUE puts every load of the thread in flight, COMMIT (later, after other work) stores to
// LDS: the HBM latency of the next tile hides behind the current tile's phases.
constexpr int TILE_UV = (TS * MAXW / 4 + 64 * WAVES - 1) / (64 * WAVES);   // float4 per thread (vector path)
constexpr int TILE_US = (TS * MAXW + 64 * WAVES - 1) / (64 * WAVES);       // floats per thread (scalar path)
static_assert(TILE_US <= 4 * TILE_UV, "scalar path must fit the vector path's registers");
struct TileRegs {
  f32x4 v[TILE_UV];   // the scalar path keeps its floats in the same registers (v[u / 4][u % 4])
};
__device__ __forceinline__ void tile_issue(const float *src, int64_t row0, int64_t N, int n, bool vec, TileRegs &r) {
  const int64_t left = N - row0;
  const int valid = left <= 0 ? 0 : (int)(left < TS ? left : TS) * n;      // floats of this tile that exist
  const float *p = src + row0 * n;
  const int nth = blockDim.x;
  if (vec) {                                               // n % 4 == 0: a float4 never straddles two rows
    const f32x4 *p4 = reinterpret_cast<const f32x4 *>(p);
#pragma unroll
    for (int u = 0; u < TILE_UV; ++u) {
      const int i = threadIdx.x + u * nth;
      r.v[u] = (4 * i < valid) ? p4[i] : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  } else {
#pragma unroll
    for (int u = 0; u < TILE_US; ++u) {
      const int i = threadIdx.x + u * nth;
      r.v[u >> 2][u & 3] = (i < valid) ? p[i] : 0.f;
    }
  }
}
__device__ __forceinline__ void tile_commit(int n, bool vec, const TileRegs &r, float *dst) {
  const int nth = blockDim.x;
  if (vec) {
    const int total4 = (TS * n) >> 2;
#pragma unroll
    for (int u = 0; u < TILE_UV; ++u) {
      const int i = threadIdx.x + u * nth;
      if (i < total4) {
        const int e = 4 * i, sidx = e / n, k = e - sidx * n;
#pragma unroll
        for (int j = 0; j < 4; ++j) dst[(k + j) * TP + sidx] = r.v[u][j];
      }
    }
  } else {
    const int total = TS * n;
#pragma unroll
    for (int u = 0; u < TILE_US; ++u) {
      const int i = threadIdx.x + u * nth;
      if (i < total) {
        const int sidx = i / n, k = i - sidx * n;
        dst[k * TP + sidx] = r.v[u >> 2][u & 3];
      }
    }
  }
}

// Rows of the activation area that are READ but never written by a tile load or a layer epilogue must be finite (they
// meet zero weights): the padding rows of H_0 behind the x tile (row n0 = the constant one) and, in the backward, the
// padding rows of the gy tile. Everything else is rewritten for every tile before it is read.
__device__ __forceinline__ void init_rows(float *area, int row0, int row1, int one_row) {
  for (int i = row0 * TP + threadIdx.x; i < row1 * TP; i += blockDim.x) area[i] = (i / TP == one_row) ? 1.f : 0.f;
}
__device__ __forceinline__ int k_rows(int n_in) { return ((((n_in + 1 + 3) >> 2) + 3) & ~3) * 4; }   // rows the k loop reads

// out^T[n][s] = act( sum_k Waug[n][k] in^T_aug[k][s] ) for NT output tiles of this wave (nt0, nt0 + WAVES)
template <int NT, int STRIDE = WAVES, int P = P2C_PREC_F32>
__device__ __forceinline__ void layer_forward_nt(const Lane &L, const float *wl, int ld, int ksteps, int nt0, int n_out,
                                                 bool relu, const float *in, float *out, float *y_row, bool row_ok,
                                                 bool vec_y) {
  f32x4 acc[NT];
  const float *ap[NT];
#pragma unroll
  for (int h = 0; h < NT; ++h) {
    acc[h] = (f32x4){0.f, 0.f, 0.f, 0.f};
    ap[h] = wl + ((nt0 + h * STRIDE) * 16 + L.c) * ld + L.g;
  }
  const float *bp = in + L.g * TP + L.c;
  // Ping-pong software pipeline: the operands of k-group s+1 are in flight while the MFMAs of group s run (ksteps is a
  // multiple of 4, >= 4). The sched_barriers keep the compiler from sinking the loads below the MFMAs they overlap.
  float b0[4], a0[NT][4], b1[4], a1[NT][4];
  auto load = [&](float (&bv)[4], float (&av)[NT][4], int s) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      bv[u] = bp[(s + u) * 4 * TP];
#pragma unroll
      for (int h = 0; h < NT; ++h) av[h][u] = ap[h][(s + u) * 4];
    }
  };
  auto fma4 = [&](const float (&bv)[4], const float (&av)[NT][4]) {
    if constexpr (P == P2C_PREC_F32) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
#pragma unroll
        for (int h = 0; h < NT; ++h) acc[h] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[h][u], bv[u], acc[h], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int h = 0; h < NT; ++h) acc[h] = mfma_k16<P>(av[h], bv, acc[h]);
    }
  };
  load(b0, a0, 0);
  for (int s = 4;; s += 8) {
    if (s < ksteps) load(b1, a1, s);
    __builtin_amdgcn_sched_barrier(0);
    fma4(b0, a0);
    __builtin_amdgcn_sched_barrier(0);
    if (s >= ksteps) break;
    if (s + 4 < ksteps) load(b0, a0, s + 4);
    __builtin_amdgcn_sched_barrier(0);
    fma4(b1, a1);
    __builtin_amdgcn_sched_barrier(0);
    if (s + 4 >= ksteps) break;
  }
#pragma unroll
  for (int h = 0; h < NT; ++h) {
    f32x4 v = acc[h];
    const int nb = (nt0 + h * STRIDE) * 16 + 4 * L.g;   // first of this lane's 4 output rows
    if (relu) {
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
    }
    if (out) {   // the unit row of the image makes row n_out == 1, rows beyond it == 0
#pragma unroll
      for (int r = 0; r < 4; ++r) out[(nb + r) * TP + L.c] = v[r];
    }
    if (y_row && row_ok) {
      if (nb + 3 < n_out && vec_y) {
        *reinterpret_cast<f32x4 *>(y_row + nb) = v;
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (nb + r < n_out) y_row[nb + r] = v[r];
      }
    }
  }
}

// in: LDS rows [k][TP] incl. ones row; out: LDS rows and/or HBM rows y. The output tiles (rows 0..n_out, ones row
// included) are dealt round-robin to the waves.
template <int P = P2C_PREC_F32>
__device__ __forceinline__ void layer_forward(const Lane &L, const float *wl, int ld, int n_in, int n_out, bool relu,
                                              const float *in, float *out, float *y_row, bool row_ok, bool vec_y) {
  const int ksteps = (((n_in + 1 + 3) >> 2) + 3) & ~3;   // multiple of 4: image and activations are zero beyond n_in
  const int ntiles = (n_out + 16) >> 4;
  for (int nt = L.wave; nt < ntiles; nt += 2 * WAVES) {
    if (nt + WAVES < ntiles) layer_forward_nt<2, WAVES, P>(L, wl, ld, ksteps, nt, n_out, relu, in, out, y_row, row_ok, vec_y);
    else layer_forward_nt<1, WAVES, P>(L, wl, ld, ksteps, nt, n_out, relu, in, out, y_row, row_ok, vec_y);
  }
}

// gout^T[m][s] = (H^T[m][s] > 0 && m < n_in) * sum_k W[k][m] gin^T[k][s]; the m-tiles are dealt to the waves
template <int P = P2C_PREC_F32>
__device__ __forceinline__ void layer_dgrad(const Lane &L, const float *wl, int ld, int n_in, int n_out, const float *gin,
                                            const float *Hprev, float *gout) {
  const int ksteps = (((n_out + 3) >> 2) + 3) & ~3;      // multiple of 4: image rows and G rows are zero beyond n_out
  const int mtiles = (n_in + 15) >> 4;
  for (int mt = L.wave; mt < mtiles; mt += WAVES) {
    f32x4 c0 = {0.f, 0.f, 0.f, 0.f};
    const float *a0p = wl + L.g * ld + mt * 16 + L.c;
    const float *bp = gin + L.g * TP + L.c;
    float b0[4], a0[4], b1[4], a1[4];
    auto load = [&](float (&bv)[4], float (&av)[4], int s) {
#pragma unroll
      for (int u = 0; u < 4; ++u) bv[u] = bp[(s + u) * 4 * TP], av[u] = a0p[(s + u) * 4 * ld];
    };
    auto fma4 = [&](const float (&bv)[4], const float (&av)[4]) { c0 = mfma_k16<P>(av, bv, c0); };
    load(b0, a0, 0);
    for (int s = 4;; s += 8) {   // same ping-pong pipeline as layer_forward_nt
      if (s < ksteps) load(b1, a1, s);
      __builtin_amdgcn_sched_barrier(0);
      fma4(b0, a0);
      __builtin_amdgcn_sched_barrier(0);
      if (s >= ksteps) break;
      if (s + 4 < ksteps) load(b0, a0, s + 4);
      __builtin_amdgcn_sched_barrier(0);
      fma4(b1, a1);
      __builtin_amdgcn_sched_barrier(0);
      if (s + 4 >= ksteps) break;
    }
    const int mb = mt * 16 + 4 * L.g;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float hv = Hprev[(mb + r) * TP + L.c];
      gout[(mb + r) * TP + L.c] = (mb + r < n_in && hv > 0.f) ? c0[r] : 0.f;
    }
  }
}

// position of global dW_aug tile t: layer, n-tile (output neurons), m-tile (input neurons + bias column), parameter base
struct TileRef {
  int l, ntile, mtile, base;
};
__device__ __forceinline__ TileRef locate_tile(const int32_t *dims, int t) {
  TileRef r{0, 0, 0, 0};
  int rem = t;
  for (;; ++r.l) {
    int cnt = ((dims[r.l + 1] + 15) >> 4) * ((dims[r.l] + 1 + 15) >> 4);
    if (rem < cnt) break;
    rem -= cnt;
    r.base += dims[r.l + 1] * (dims[r.l] + 1);
  }
  const int mtiles = (dims[r.l] + 1 + 15) >> 4;
  r.ntile = rem / mtiles;
  r.mtile = rem - r.ntile * mtiles;
  return r;
}

// Saved activations of one sample tile <-> the H_1..H_{L-1} rows in LDS. With a static shape the next tile's rows wait in
// registers (at most two float4 per layer and thread); the generic shape copies without look-ahead.
constexpr int ACT_C = (4 * (MAXW - 1) + NTH - 1) / NTH;
struct ActRegs {
  f32x4 v[NL][ACT_C];
};
template <class S>
__device__ __forceinline__ void acts_issue(const S &sh, const MlpArgs &a, int64_t tile, int64_t n_tiles, ActRegs &r) {
  if (tile >= n_tiles) return;
  const f32x4 *src = reinterpret_cast<const f32x4 *>(a.saved) + (size_t)tile * a.f_half * 4;
  for_layers(sh, 1, sh.n_layers(), [&](int l) {
    const int rows4 = sh.dims(l) * 4;
#pragma unroll
    for (int c = 0; c < ACT_C; ++c) {
      const int i = threadIdx.x + c * NTH;
      if (c * NTH < rows4 && i < rows4) r.v[l][c] = src[a.f_off[l] * 4 + i];
    }
  });
}
template <class S>
__device__ __forceinline__ void acts_commit(const S &sh, const ActRegs &r, float *H) {
  for_layers(sh, 1, sh.n_layers(), [&](int l) {
    const int rows4 = sh.dims(l) * 4;
    float *dst = H + sh.h_off(l) * TP;
#pragma unroll
    for (int c = 0; c < ACT_C; ++c) {
      const int i = threadIdx.x + c * NTH;
      if (c * NTH < rows4 && i < rows4) {
        const int o = (i >> 2) * TP + (i & 3) * 4;
        dst[o] = r.v[l][c][0], dst[o + 1] = r.v[l][c][1], dst[o + 2] = r.v[l][c][2], dst[o + 3] = r.v[l][c][3];
      }
    }
  });
}
template <class S>
__device__ __forceinline__ void acts_copy(const S &sh, const MlpArgs &a, int64_t tile, float *H) {   // no look-ahead
  const f32x4 *src = reinterpret_cast<const f32x4 *>(a.saved) + (size_t)tile * a.f_half * 4;
  for_layers(sh, 1, sh.n_layers(), [&](int l) {
    float *dst = H + sh.h_off(l) * TP;
    for (int i = threadIdx.x; i < sh.dims(l) * 4; i += NTH) {
      const f32x4 v = src[a.f_off[l] * 4 + i];
      const int o = (i >> 2) * TP + (i & 3) * 4;
      dst[o] = v[0], dst[o + 1] = v[1], dst[o + 2] = v[2], dst[o + 3] = v[3];
    }
  });
}
// LDS rows of H_l (l = 1..L-1) -> the transposed HBM block of one sample tile (forward: saved activations; backward with
// the split weight gradient: the H half of the factors)
template <class S>
__device__ __forceinline__ void acts_store(const S &sh, const MlpArgs &a, const float *H, f32x4 *dst_tile) {
  for_layers(sh, 1, sh.n_layers(), [&](int l) {
    const float *src = H + sh.h_off(l) * TP;
    f32x4 *hd = dst_tile + a.f_off[l] * 4;
    for (int i = threadIdx.x; i < sh.dims(l) * 4; i += NTH) {
      const int o = (i >> 2) * TP + (i & 3) * 4;
      hd[i] = (f32x4){src[o], src[o + 1], src[o + 2], src[o + 3]};
    }
  });
}

template <class S>
__host__ __device__ constexpr int issue_mark(int slot) {
  constexpr int nl = static_layers<S>();
  const int total = rounds_upto<S>(nl - 1);
  int m = rounds_upto<S>(nl > 1 ? 1 : 0);                   // prologue: layers 0 and 1
  if (nl < 4) return slot == 0 ? 0 : total;                 // too few barriers to spread anything: all in the prologue
  if (slot == 0) return 0;
  for (int j = 1; j < slot; ++j) m = (j == 3 || m + 4 > total) ? total : m + 4;
  return m > total ? total : m;
}


}  // namespace p2c_mlp
