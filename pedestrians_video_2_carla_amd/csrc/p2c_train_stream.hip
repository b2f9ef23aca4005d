// p2c_train_stream.hip -- the THROUGHPUT form of the fused train step's first launch (gfx950): two wavefronts per clip,
// four clips in flight per CU.
//
// What it replaces: the same reference path as p2c_train.hip (modules/flow/pose_lifting.py:121-144 -> LinearAE.forward,
// linear_ae.py:50-59 -> ProjectionModule, modules/layers/projection.py:73-123 -> transform_callable -> loss/loc_2d_3d.py:6-17
// -> backward), for batches of SEVERAL clips per CU. train_clip_kernel (p2c_train.hip) spends a whole workgroup of eight
// wavefronts and 156 KB of LDS on one clip at a time -- the latency form, right at one clip per CU (B = 256) and a queue of
// dependent 16-sample chains beyond it (B = 1024: four clips per CU back to back, 57 us). Here
//
//   * a workgroup is EIGHT wavefronts in FOUR pairs (waves 2p, 2p + 1: two different SIMDs); a pair walks its OWN clips start to
//     finish and meets only its partner (PairSync: two LDS counters) -- no workgroup barrier after the prologue, the four pairs
//     drift apart and one pair's MFMA phases fill the matrix pipes while another's pose head holds the VALUs;
//   * the packed weight image (84 KB) sits in LDS once per workgroup (LDS-DMA, one burst) and serves all four pairs;
//   * a pair's activations live in a private 18 KB LDS region, transposed ([feature row][16 frames], pitch 16 = the layout of
//     the factor blocks the weight-gradient launch reads: factor rows leave as straight 1 KB copies): H_0 .. H_5 ping-pong
//     through two small buffers, y^T / grad_y^T share the third; a layer's 16-row output tiles are split between the pair's
//     two waves (one pass over k per wave, up to GROUP accumulators), the ReLU masks of H_1 .. H_5 are 48 bits per lane in
//     registers (the same wave owns a tile of H_l as a forward output and as a dgrad m-tile);
//   * the pose head is the chain-lane arithmetic of p2c_pose_head_chain_dev.h with a different unit: eight lanes own a
//     (clip, FRAME), wave `half` of the pair owns frames 8 half .. 8 half + 7. The only couplings between frames become scans:
//     the cumulative rotation (projection.py:190-193) an exclusive scan of 3x3 products over the wave's eight frames (three
//     ds_bpermute rounds) + one hand-over of the first wave's total to the second through LDS; the backward's suffix sum of
//     torques an add scan + the second wave's total to the first. y never leaves LDS, grad_y is written over it and is the
//     dgrad chain's first operand: no y / grad_y traffic at any B.
//
// Leaves what train_clip_kernel leaves: the clip's factor block (H_0 .. H_5 | G_1 .. G_6, 34 KB) and its three loss sums.
// MLP arithmetic: the same fmaf chains in the same k order as p2c_mlp_dev.h (fp32 MFMA 16x16x4); pose head: the sums over
// time associate differently (scan instead of a running product) -- parity 1e-4 against the fp64 oracle, not bitwise
// against the latency form (tests/test_train_fused_gpu.py).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>

#include "../../include/p2c.h"
#include "p2c_mlp_dev.h"
#include "p2c_pose_head_dev.h"
#include "p2c_pose_head_chain_dev.h"
#include "p2c_train_dev.h"

namespace p2c_stream {

using namespace p2c_mlp;
using namespace p2c_train;
namespace ch = p2c::chain;
using ph::M3;
using ph::V3;

constexpr int SW = 8;                       // wavefronts per workgroup: two per SIMD
constexpr int PAIRS = 4;                    // ... in four pairs (2p, 2p + 1: two SIMDs): a pair shares an LDS region and a clip
constexpr int PT = 16;                      // LDS pitch of an activation row (floats) = frames of a clip
constexpr int ROWS_Y = 160;                 // y^T / grad_y^T (156 rows + k padding); H_0 before the first layer
constexpr int ROWS_A = 80, ROWS_B = 48;     // ping-pong buffers: A holds H_1, H_3, H_5 / G_5, G_3, G_1; B holds H_2, H_4 / G_4, G_2
constexpr int REGION = (ROWS_Y + ROWS_A + ROWS_B) * PT;
constexpr int TAB_LOC = 0, TAB_ROT = 4 * ph::J * 3, TAB_FLOATS = 4 * ph::J * 12;
constexpr int SCRATCH = 16;
constexpr int LDS_FLOATS = S::w_total() + TAB_FLOATS + SCRATCH + PAIRS * REGION;
static_assert(LDS_FLOATS * 4 <= 160 * 1024, "LDS budget");

__host__ __device__ constexpr int ksteps_fwd(int l) { return ((((S::dim_at(l) + 1 + 3) >> 2) + 3) & ~3); }
__host__ __device__ constexpr int ksteps_bwd(int l) { return ((((S::dim_at(l + 1) + 3) >> 2) + 3) & ~3); }
__host__ __device__ constexpr int ntiles_fwd(int l) { return (S::dim_at(l + 1) + 16) >> 4; }
__host__ __device__ constexpr int mtiles_bwd(int l) { return (S::dim_at(l) + 15) >> 4; }
// first ReLU-mask bit of H_l (l = 1 .. 5): four bits (the lane's four rows) per 16-row tile
__host__ __device__ constexpr int mask_bit0(int l) {
  int b = 0;
  for (int i = 1; i < l; ++i) b += 4 * ntiles_fwd(i - 1);
  return b;
}
static_assert(mask_bit0(NLAY - 1) + 4 * ntiles_fwd(NLAY - 2) <= 64, "ReLU masks fit 64 bits per lane");
// the buffers the layers meet
static_assert(ntiles_fwd(0) * 16 <= ROWS_A && ntiles_fwd(2) * 16 <= ROWS_A && ntiles_fwd(4) * 16 <= ROWS_A, "A rows");
static_assert(ntiles_fwd(1) * 16 <= ROWS_B && ntiles_fwd(3) * 16 <= ROWS_B, "B rows");
static_assert(ntiles_fwd(5) * 16 <= ROWS_Y && ksteps_bwd(5) * 4 <= ROWS_Y && ksteps_fwd(0) * 4 <= ROWS_Y, "y rows (H_0 before the first layer)");

#ifdef P2C_STREAM_TRACE   // developer build only (tools/streamtrace.py): shader-clock stamps of wave 0 of one workgroup
static __device__ unsigned long long g_strace[64];
#ifndef P2C_STREAM_TRACE_BLOCK
#define P2C_STREAM_TRACE_BLOCK 0
#endif
#define ST(i)                                                                            \
  do {                                                                                   \
    if (blockIdx.x == P2C_STREAM_TRACE_BLOCK && threadIdx.x == 0) {                      \
      g_strace[i] = __builtin_readcyclecounter();                                        \
      if ((i) == 0 || (i) == 63) g_strace[(i) == 0 ? 62 : 61] = wall_clock64();          \
    }                                                                                    \
  } while (0)
#else
#define ST(i)
#endif

// timing experiments only (WRONG results): 1 = no pose head, 2 = no LinearAE, 4 = no factor stores, 8 = every clip's factors
// to the pair's first block (bits combine)
#ifndef P2C_STREAM_EXPERIMENT
#define P2C_STREAM_EXPERIMENT 0
#endif
#ifndef P2C_STREAM_PRIO
#define P2C_STREAM_PRIO 1                   // s_setprio of a wave in its MLP phases (pose phases: 0): its sparse MFMA issue does not
#endif                                      // queue behind the VALU stream of the other pair on its SIMD (B = 8192: 221 -> 213 us)
constexpr int GROUP = 5;                   // output tiles per pass over k (accumulators in flight)

// A value the optimiser must take as new at this point: everything derived from it is recomputed here instead of being hoisted
// out of the clip loop and carried (spilled) across the other phases. A handful of integer instructions per phase buys ~50
// registers of live range.
__device__ __forceinline__ int fresh(int v) {
  asm volatile("" : "+v"(v));
  return v;
}

// first bone and number of bones of chain q (= chain::c_start / c_len, as arithmetic: a lane-indexed __constant__ table is a
// vector load, and the wait the compiler puts in front of its use drains every older store)
__device__ __forceinline__ int chain_start(int q) { return q < 5 ? 4 * q : (q == 5 ? 21 : (q == 6 ? 20 : 25)); }
__device__ __forceinline__ int chain_len(int q) { return q < 6 ? 4 : 1; }

// Rendezvous of the TWO wavefronts of a pair (no workgroup barrier inside the clip loop: the four pairs of a workgroup run free of
// each other). Each wave owns a counter in LDS: it publishes its own arrival number behind its LDS traffic and waits until its
// partner's counter has reached the same number. Both waves of a pair are resident in the same workgroup: the wait is bounded.
struct PairSync {
  int *mine, *other;
  int epoch;
  __device__ __forceinline__ void sync() {
    ++epoch;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __hip_atomic_store(mine, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    while (__hip_atomic_load(other, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < epoch) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
  }
};

struct WLane {
  int lane, c, g;        // c = lane & 15: frame (MFMA column), g = lane >> 4
};

// out^T[n][t] = act(sum_k Waug[n][k] in^T[k][t]) for NT consecutive 16-row output tiles; two accumulators in flight hide the
// 40-cycle dependent latency of v_mfma_f32_16x16x4_f32 behind its 32-cycle issue
template <int NT, int KSTEPS, bool RELU>
__device__ __forceinline__ void fwd_tiles(const WLane &L, const float *wl, const int ld, const int nt0, const float *in, float *out,
                                          uint64_t &mask, const int bit0) {
  f32x4 acc[NT];
  const float *ap[NT];
#pragma unroll
  for (int h = 0; h < NT; ++h) {
    acc[h] = (f32x4){0.f, 0.f, 0.f, 0.f};
    ap[h] = wl + ((nt0 + h) * 16 + L.c) * ld + L.g;
  }
  const float *bp = in + L.g * PT + L.c;
  float b0[4], a0[NT][4], b1[4], a1[NT][4];
  auto load = [&](float (&bv)[4], float (&av)[NT][4], int s) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      bv[u] = bp[(s + u) * 4 * PT];
#pragma unroll
      for (int h = 0; h < NT; ++h) av[h][u] = ap[h][(s + u) * 4];
    }
  };
  auto fma4 = [&](const float (&bv)[4], const float (&av)[NT][4]) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int h = 0; h < NT; ++h) acc[h] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[h][u], bv[u], acc[h], 0, 0, 0);
    }
  };
  load(b0, a0, 0);
#pragma unroll
  for (int s = 4;; s += 8) {     // ping-pong: the operands of k-group s+1 are in flight while the MFMAs of group s run
    if (s < KSTEPS) load(b1, a1, s);
    __builtin_amdgcn_sched_barrier(0);
    fma4(b0, a0);
    __builtin_amdgcn_sched_barrier(0);
    if (s >= KSTEPS) break;
    if (s + 4 < KSTEPS) load(b0, a0, s + 4);
    __builtin_amdgcn_sched_barrier(0);
    fma4(b1, a1);
    __builtin_amdgcn_sched_barrier(0);
    if (s + 4 >= KSTEPS) break;
  }
#pragma unroll
  for (int h = 0; h < NT; ++h) {
    f32x4 v = acc[h];
    const int nb = (nt0 + h) * 16 + 4 * L.g;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (RELU) {
        mask |= (uint64_t)(v[r] > 0.f ? 1u : 0u) << (bit0 + (nt0 + h) * 4 + r);
        v[r] = fmaxf(v[r], 0.f);
      }
      out[(nb + r) * PT + L.c] = v[r];     // the unit row of the image makes row n_out == 1, rows beyond it == 0
    }
  }
}
// The two wavefronts of a pair split a layer's output tiles: the first takes tiles [0, ceil(n / 2)), the second the rest (two
// SIMDs, two matrix pipes: see the kernel). The same split of H_l's tiles in the forward (as layer l-1's output) and in the
// dgrad (as layer l's m-tiles) keeps a lane's ReLU mask bits with the wave that needs them.
__host__ __device__ constexpr int split0(int n) { return (n + 1) / 2; }
template <int LL, int T0, int CNT>
__device__ __forceinline__ void fwd_range(const WLane &L, const float *img, const float *in, float *out, uint64_t &mask) {
  constexpr int KS_ = ksteps_fwd(LL);
  constexpr bool RELU = LL < NLAY - 1;
  const float *wl = img + S::w_off(LL);
  // up to GROUP tiles share one pass over k: one B operand read feeds GROUP MFMAs, and a pass has one start-up and one epilogue
  if constexpr (CNT > 0) {
    static_assert(CNT <= GROUP, "one pass per wave and layer");
    fwd_tiles<CNT, KS_, RELU>(L, wl, S::ld(LL), T0, in, out, mask, mask_bit0(LL + 1));
  }
}
template <int LL>
__device__ __forceinline__ void fwd_layer(const WLane &L, const int half, const float *img, const float *in, float *out, uint64_t &mask) {
  constexpr int NT_ = ntiles_fwd(LL), N0 = split0(NT_);
  if (half == 0) fwd_range<LL, 0, N0>(L, img, in, out, mask);
  else fwd_range<LL, N0, NT_ - N0>(L, img, in, out, mask);
}

// gout^T[m][t] = relu'(H[m][t]) * sum_k W[k][m] gin^T[k][t] for NT consecutive m-tiles (the mask bits stand in for H)
template <int NT, int KSTEPS>
__device__ __forceinline__ void dgrad_tiles(const WLane &L, const float *wl, const int ld, const int mt0, const int n_in,
                                            const float *gin, float *gout, const uint64_t mask, const int bit0) {
  f32x4 acc[NT];
  const float *ap[NT];
#pragma unroll
  for (int h = 0; h < NT; ++h) {
    acc[h] = (f32x4){0.f, 0.f, 0.f, 0.f};
    ap[h] = wl + L.g * ld + (mt0 + h) * 16 + L.c;
  }
  const float *bp = gin + L.g * PT + L.c;
  float b0[4], a0[NT][4], b1[4], a1[NT][4];
  auto load = [&](float (&bv)[4], float (&av)[NT][4], int s) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      bv[u] = bp[(s + u) * 4 * PT];
#pragma unroll
      for (int h = 0; h < NT; ++h) av[h][u] = ap[h][(s + u) * 4 * ld];
    }
  };
  auto fma4 = [&](const float (&bv)[4], const float (&av)[NT][4]) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int h = 0; h < NT; ++h) acc[h] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[h][u], bv[u], acc[h], 0, 0, 0);
    }
  };
  load(b0, a0, 0);
#pragma unroll
  for (int s = 4;; s += 8) {
    if (s < KSTEPS) load(b1, a1, s);
    __builtin_amdgcn_sched_barrier(0);
    fma4(b0, a0);
    __builtin_amdgcn_sched_barrier(0);
    if (s >= KSTEPS) break;
    if (s + 4 < KSTEPS) load(b0, a0, s + 4);
    __builtin_amdgcn_sched_barrier(0);
    fma4(b1, a1);
    __builtin_amdgcn_sched_barrier(0);
    if (s + 4 >= KSTEPS) break;
  }
#pragma unroll
  for (int h = 0; h < NT; ++h) {
    const int mb = (mt0 + h) * 16 + 4 * L.g;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool on = (mb + r < n_in) && ((mask >> (bit0 + (mt0 + h) * 4 + r)) & 1u);
      gout[(mb + r) * PT + L.c] = on ? acc[h][r] : 0.f;
    }
  }
}
template <int LL, int T0, int CNT>
__device__ __forceinline__ void dgrad_range(const WLane &L, const float *img, const float *gin, float *gout, const uint64_t mask) {
  constexpr int KS_ = ksteps_bwd(LL);
  const float *wl = img + S::w_off(LL);
  if constexpr (CNT > 0) {
    static_assert(CNT <= GROUP, "one pass per wave and layer");
    dgrad_tiles<CNT, KS_>(L, wl, S::ld(LL), T0, S::dims(LL), gin, gout, mask, mask_bit0(LL));
  }
}
template <int LL>   // G_LL from G_{LL+1}, LL = 5 .. 1
__device__ __forceinline__ void dgrad_layer(const WLane &L, const int half, const float *img, const float *gin, float *gout, const uint64_t mask) {
  constexpr int MT_ = mtiles_bwd(LL), M0 = split0(MT_);
  static_assert(LL == 0 || mtiles_bwd(LL) == ntiles_fwd(LL - 1) || LL < 1, "H_l: same tiles in the forward and in the dgrad");
  if (half == 0) dgrad_range<LL, 0, M0>(L, img, gin, gout, mask);
  else dgrad_range<LL, M0, MT_ - M0>(L, img, gin, gout, mask);
}

// `rows` rows of a region (64 contiguous bytes each) -> the clip's factor block: straight 16-byte copies, 1 KB per instruction.
// Two halves: the LDS reads go out before the next layer starts, the stores follow it (the rows wait in registers, no exposed
// LDS latency, and the region may be overwritten in between).
template <int ROWS>
struct RowRegs {
  f32x4 v[(ROWS * 4 + 63) / 64];
};
// the first wavefront's share of `rows` factor rows: whole 16-row (1 KB) pieces, about half of them
__host__ __device__ constexpr int rows0(int rows) { return ((rows / 16 + 1) / 2) * 16 < rows ? ((rows / 16 + 1) / 2) * 16 : rows; }
template <int ROWS>
struct HalfRows {          // registers for either share
  static constexpr int R0 = rows0(ROWS), R1 = ROWS - rows0(ROWS), RM = R0 > R1 ? R0 : R1;
  RowRegs<(RM > 0 ? RM : 1)> r;
};
template <int ROWS>
__device__ __forceinline__ void half_read(const float *src, const int lane, const int half, HalfRows<ROWS> &h);
template <int ROWS>
__device__ __forceinline__ void half_store(const HalfRows<ROWS> &h, float *dst, const int lane, const int half);
template <int ROWS>
__device__ __forceinline__ void rows_read(const float *src, const int lane, RowRegs<ROWS> &r) {
  const f32x4 *s4 = reinterpret_cast<const f32x4 *>(src);
#pragma unroll
  for (int i = 0; i < (ROWS * 4 + 63) / 64; ++i) {
    const int p = i * 64 + lane;
    r.v[i] = s4[((i + 1) * 64 <= ROWS * 4 || p < ROWS * 4) ? p : 0];
  }
}
template <int ROWS>
__device__ __forceinline__ void rows_store(const RowRegs<ROWS> &r, float *dst, const int lane) {
  f32x4 *d4 = reinterpret_cast<f32x4 *>(dst);
#pragma unroll
  for (int i = 0; i < (ROWS * 4 + 63) / 64; ++i) {
    const int p = i * 64 + lane;
    // An asm store: hipcc makes every VALU write to a register a pending store of ITS OWN has read wait for that store's
    // completion (vmcnt) -- with 34 KB of factors per clip in flight the MLP wave drained its queue a dozen times per clip.
    // The hardware needs two wait states, inside the string (cdna_hip_programming.md, inline-asm rules: stores).
    if (!(P2C_STREAM_EXPERIMENT & 4) && ((i + 1) * 64 <= ROWS * 4 || p < ROWS * 4))
      asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(d4 + p), "v"(r.v[i]) : "memory");
  }
}

template <int ROWS>
__device__ __forceinline__ void half_read(const float *src, const int lane, const int half, HalfRows<ROWS> &h) {
  constexpr int R0 = HalfRows<ROWS>::R0, R1 = HalfRows<ROWS>::R1, RM = HalfRows<ROWS>::RM;
  const f32x4 *s4 = reinterpret_cast<const f32x4 *>(src) + (half ? R0 * 4 : 0);
  const int n4 = (half ? R1 : R0) * 4;
#pragma unroll
  for (int i = 0; i < (RM * 4 + 63) / 64; ++i) {
    const int p = i * 64 + lane;
    h.r.v[i] = s4[p < n4 ? p : 0];
  }
}
template <int ROWS>
__device__ __forceinline__ void half_store(const HalfRows<ROWS> &h, float *dst, const int lane, const int half) {
  constexpr int R0 = HalfRows<ROWS>::R0, R1 = HalfRows<ROWS>::R1, RM = HalfRows<ROWS>::RM;
  f32x4 *d4 = reinterpret_cast<f32x4 *>(dst) + (half ? R0 * 4 : 0);
  const int n4 = (half ? R1 : R0) * 4;
#pragma unroll
  for (int i = 0; i < (RM * 4 + 63) / 64; ++i) {
    const int p = i * 64 + lane;
    if (!(P2C_STREAM_EXPERIMENT & 4) && p < n4) asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(d4 + p), "v"(h.r.v[i]) : "memory");
  }
}

// the x tile of one clip: T x 52 contiguous floats; rows beyond T read as zero
struct XRegs {
  f32x4 v[4];
};
__device__ __forceinline__ void x_issue(const float *x, const int64_t clip, const int B, const int T, const int lane, XRegs &r) {
  const f32x4 *p4 = reinterpret_cast<const f32x4 *>(x + (clip < B ? clip : 0) * T * 52);
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int i = lane + 64 * u;
    r.v[u] = (clip < B && i < 13 * T) ? p4[i] : (f32x4){0.f, 0.f, 0.f, 0.f};
  }
}
// LDS float offset (row k = feature, column t = frame) of the first element of each of the lane's float4 pieces: a property
// of the lane alone, worked out once per kernel (a division by 52 per piece)
struct XOffs {
  int o[4];
};
__device__ __forceinline__ XOffs x_offsets(const int lane) {
  XOffs r;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int e = 4 * (lane + 64 * u), t = e / 52, k = e - t * 52;
    r.o[u] = (lane + 64 * u < 13 * 16) ? k * PT + t : -1;
  }
  return r;
}
__device__ __forceinline__ void x_commit(const XRegs &r, const XOffs &xo, float *Y, const int lane) {
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    if (u < 3 || xo.o[u] >= 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j) Y[xo.o[u] + j * PT] = r.v[u][j];
    }
  }
  // the constant-one row behind the inputs and the zero rows the k loop reads up to its rounding
#pragma unroll
  for (int i = 0; i < (ksteps_fwd(0) * 4 - 52) * PT / 64; ++i) Y[52 * PT + i * 64 + lane] = (i == 0 && lane < PT) ? 1.f : 0.f;
}

typedef __attribute__((address_space(3))) void *lds_void_ptr;
constexpr int IMG_CHUNKS = ((S::w_total() >> 2) + 63) / 64;          // 1 KB pieces (64 lanes x 16 B)
constexpr int IMG_CHUNKS_EARLY = (S::w_off(NLAY - 1) * 4 + 1023) / 1024;   // ... that hold layers 0 .. L-2 (and the head of the last layer)

// targets of one frame for the lane's four bones + the clip's skeleton type: issued at the TOP of an iteration (in front of the
// forward's factor stores: vmcnt retires in issue order, a load behind 26 stores waits for all of them), consumed by the pose head
struct PoseIn {
  ch::FrameIn4 in;
  int st;
};
__device__ __forceinline__ void pose_inputs(const p2c_pose_head_desc &d, const int clip, const int lane_, const int half, PoseIn &pi) {
  constexpr int NS = ch::NS;
  const int lane = fresh(lane_);
  const int T = d.T, t = 8 * half + (lane >> 3), start = chain_start(lane & 7);
  pi.st = d.skel_type[clip];
  const size_t c2 = (size_t)T * ph::J * 2 * 4, c3 = (size_t)T * ph::J * 3 * 4;          // bytes per clip
  const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<void *>(reinterpret_cast<uintptr_t>(d.gt2d) + (size_t)clip * c2), 0, d.gt2d ? (int)c2 : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t r3 = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<void *>(reinterpret_cast<uintptr_t>(d.gt3d) + (size_t)clip * c3), 0, d.gt3d ? (int)c3 : 0, 0x00020000);
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    const int j = start + k < ph::J ? start + k : ph::J - 1;      // (steps a lane does not own: any bone, masked later)
    const int jf = t * ph::J + j;                                  // frames beyond T: past the records, read as zero
    const ph::fb_f32x2 a = __builtin_bit_cast(ph::fb_f32x2, __builtin_amdgcn_raw_buffer_load_b64(r2, jf * 8, 0, 0));
    const ph::fb_f32x3 b = __builtin_bit_cast(ph::fb_f32x3, __builtin_amdgcn_raw_buffer_load_b96(r3, jf * 12, 0, 0));
    pi.in.g2[k][0] = a[0], pi.in.g2[k][1] = a[1];
    pi.in.g3[k][0] = b[0], pi.in.g3[k][1] = b[1], pi.in.g3[k][2] = b[2];
  }
}

__device__ __forceinline__ void x_landed(XRegs &r) {
#pragma unroll
  for (int u = 0; u < 4; ++u) asm volatile("" : "+v"(r.v[u]));
}
// the loaded values are pinned HERE: the wait the compiler emits for them covers loads only when no store has been issued yet
__device__ __forceinline__ void pose_inputs_landed(PoseIn &pi) {
#pragma unroll
  for (int k = 0; k < ch::NS; ++k) {
    asm volatile("" : "+v"(pi.in.g2[k][0]), "+v"(pi.in.g2[k][1]), "+v"(pi.in.g3[k][0]), "+v"(pi.in.g3[k][1]), "+v"(pi.in.g3[k][2]));
  }
  asm volatile("" : "+v"(pi.st));
}

// ---- the pose head of ONE clip by the two wavefronts of a pair ------------------------------------------------------------------
// lane = (segment = lane >> 3, chain = lane & 7: up to four consecutive bones); wavefront `half` of the pair owns frames
// 8 half + segment: ONE frame per eight-lane unit. Per-frame work is chain::fk_local / fk_base / head4 / subtree4 unchanged (their
// cross-lane moves stay inside the eight lanes of a unit). The couplings between frames are scans over the segments (ds_bpermute:
// lane -/+ 8, 16, 32) plus one hand-over between the two wavefronts through LDS: the first half's product of changes to the
// second, the second half's torque sums to the first. The rendezvous are the pair's own (PairSync).
constexpr int XCH_ROT = 0, XCH_TAU = 8 * ch::NS * 9, XCH_LOSS = XCH_TAU + 8 * ch::NS * 3, XCH_FLOATS = XCH_LOSS + 4;
static_assert(XCH_FLOATS <= ROWS_A * PT, "the hand-over scratch lives in buffer A (idle during the pose head)");

template <int KIND>
__device__ __forceinline__ void pose_phase(const p2c_pose_head_desc &d, const bool active, const int clip, const int lane_, const int half,
                                           const PoseIn &pin, float *Y, float *xch, const float *tab, const float coef2, const float coef3,
                                           PairSync &ps) {
  using K = ph::KindTraits<KIND>;
  constexpr int NS = ch::NS;
  const int T = d.T;
  const int lane = fresh(lane_);
  ch::Lane L;
  L.lane = lane, L.slot = lane >> 3, L.chain = lane & 7, L.clip = clip, L.clip_ok = true;
  L.start = chain_start(L.chain);
  L.trunk = L.chain == 0, L.head = L.chain == 2, L.leg = (L.chain == 4 || L.chain == 5), L.toe = L.chain >= 6, L.on_hips = L.chain >= 4;
#pragma unroll
  for (int k = 0; k < NS; ++k) L.valid[k] = k < chain_len(L.chain);
  const int t = 8 * half + L.slot;                  // this unit's frame
  const bool frame_ok = t < T;
  // cross-segment moves: ds_bpermute with the byte addresses of lane -/+ 8, 16, 32
  const int up_addr[3] = {(lane - 8) * 4, (lane - 16) * 4, (lane - 32) * 4}, dn_addr[3] = {(lane + 8) * 4, (lane + 16) * 4, (lane + 32) * 4};
  auto bperm = [](int addr, float v) { return __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(v))); };

  const ch::FrameIn4 &in = pin.in;
  M3 c[NS], X[NS];                                  // the frame's changes (rows b1, b2, b3); their inclusive products over time
  float gs_n1[NS], gs_n2[NS], gs_d[NS];             // what the pull-back needs besides c: |a1|, |u2|, b1 . a2
  bool gs_ok[NS];
  ST(20);
  if (active) {
    // ---- the unit's rotations: y^T rows (bone, i) at column t ------------------------------------------------------------------
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      const int j = L.start + k < ph::J ? L.start + k : ph::J - 1;
      float y6[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) y6[i] = Y[(j * 6 + i) * PT + (t & 15)];
      if (!frame_ok) y6[0] = 1.f, y6[1] = 0.f, y6[2] = 0.f, y6[3] = 0.f, y6[4] = 1.f, y6[5] = 0.f;   // beyond the clip: identity change
      ph::SixD s;
      c[k] = ph::rot6d_fwd(y6, s);
      gs_n1[k] = s.n1, gs_n2[k] = s.n2, gs_d[k] = s.d, gs_ok[k] = s.c1 && s.c2;
      X[k] = c[k];
    }
    ST(21);
    if (K::SCAN) {
      // rel_rot[t] = change[t] rel_rot[t-1], rel_rot[-1] = the clip's reference pose (projection.py:190-193; data/carla/reference.py
      // tables, staged in LDS): frame 0's element of the scan is change[0] x reference, so the inclusive products ARE rel_rot -- no
      // exclusive shift, no product with the reference per frame. Only the first wavefront holds frame 0.
      if (half == 0) {
        const int st = __builtin_amdgcn_readfirstlane(pin.st) & 3;
#pragma unroll
        for (int k = 0; k < NS; ++k) {
          const int j = L.start + k < ph::J ? L.start + k : ph::J - 1;
          const int row = st * ph::J + j;
          M3 Rref;
#pragma unroll
          for (int i = 0; i < 9; ++i) Rref.m[i] = tab[TAB_ROT + row * 9 + i];
          X[k] = ch::sel(lane < 8, ph::mul(c[k], Rref), c[k]);
        }
      }
      // inclusive scan over the wave's eight frames (the four bones of a round travel together)
      ST(43);
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        if (r == 1) ST(44);
        if (r == 2) ST(45);
        M3 Q[NS];
#pragma unroll
        for (int k = 0; k < NS; ++k)
#pragma unroll
          for (int i = 0; i < 9; ++i) Q[k].m[i] = bperm(up_addr[r], X[k].m[i]);
        const bool has = lane >= (8 << r);
#pragma unroll
        for (int k = 0; k < NS; ++k) X[k] = ch::sel(has, ph::mul(X[k], Q[k]), X[k]);
      }
      if (half == 0 && lane >= 56) {               // rel_rot[7]: the second wavefront's frames continue from it
#pragma unroll
        for (int k = 0; k < NS; ++k)
#pragma unroll
          for (int i = 0; i < 9; ++i) xch[XCH_ROT + (L.chain * NS + k) * 9 + i] = X[k].m[i];
      }
    }
  }
  ST(22);
  ps.sync();                                       // ---- hand-over 1: the first half's rel_rot[7] is in LDS ----
  ch::Acc acc{0.f, 0.f, 0.f};
  V3 taup[NS], later[NS], gb1[NS], gb2[NS], gb3[NS];   // rows of the change: what the pull-back needs of it
  M3 R[NS];                                        // rel_rot of the frame
  if (active) {
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      gb1[k] = ph::v3(c[k].m[0], c[k].m[1], c[k].m[2]), gb2[k] = ph::v3(c[k].m[3], c[k].m[4], c[k].m[5]);
      gb3[k] = ph::v3(c[k].m[6], c[k].m[7], c[k].m[8]);
    }
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      R[k] = X[k];                                 // (kinds without a scan: the change itself)
      if (K::SCAN && half != 0) {
        M3 TA;
#pragma unroll
        for (int i = 0; i < 9; ++i) TA.m[i] = xch[XCH_ROT + (L.chain * NS + k) * 9 + i];
        R[k] = ph::mul(X[k], TA);
      }
    }
    ST(23);
    // ---- forward + backward of the frame down to the parent-frame torques -----------------------------------------------------
    V3 l[NS];
    {
      const int st = __builtin_amdgcn_readfirstlane(pin.st) & 3;
#pragma unroll
      for (int k = 0; k < NS; ++k) {
        const int j = L.start + k < ph::J ? L.start + k : ph::J - 1;
        const int row = st * ph::J + j;
        l[k] = ph::v3(tab[TAB_LOC + row * 3], tab[TAB_LOC + row * 3 + 1], tab[TAB_LOC + row * 3 + 2]);
        l[k] = ch::sel(L.valid[k], l[k], ph::v3(0.f, 0.f, 0.f));
      }
    }
    M3 Al[NS], Ap3, BA;
    V3 xl[NS], BX;
    ch::fk_local(L, R, l, Al, xl, Ap3);
    ch::fk_base(L, Al, xl, BA, BX);
    V3 x[NS], F[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) x[k] = ch::sel(L.valid[k], ph::vmul(xl[k], BA) + BX, ph::v3(0.f, 0.f, 0.f));
    ST(24);
    ch::head4<true, true>(d, L, t, x, in, acc, coef2, coef3, F);
    ST(25);
    V3 FX[NS], SubF[NS], SubX[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) FX[k] = ph::cross(F[k], x[k]);
    ch::subtree4(L, F, SubF);
    ch::subtree4(L, FX, SubX);
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      const V3 tau = SubX[k] - ph::cross(SubF[k], x[k]);
      V3 tp = ph::vmulT(tau, BA);                  // tau A_parent^T, A_parent = A'_parent-in-chain A_base
      if (k > 0) tp = ph::vmulT(tp, (k == 3) ? Ap3 : Al[k - 1]);
      taup[k] = tp;
    }
    ST(26);
    // ---- suffix sums over time of the torques: inclusive add scan over the wave's frames, the second half's total to LDS ----------
    if (K::SCAN) {
      V3 X[NS];
#pragma unroll
      for (int k = 0; k < NS; ++k) X[k] = taup[k];
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        V3 Q[NS];
#pragma unroll
        for (int k = 0; k < NS; ++k) Q[k] = ph::v3(bperm(dn_addr[r], X[k].x), bperm(dn_addr[r], X[k].y), bperm(dn_addr[r], X[k].z));
        const bool has = lane + (8 << r) < 64;
#pragma unroll
        for (int k = 0; k < NS; ++k) X[k] = ch::sel(has, X[k] + Q[k], X[k]);
      }
      if (half != 0 && lane < 8) {
#pragma unroll
        for (int k = 0; k < NS; ++k) {
          float *q = xch + XCH_TAU + (L.chain * NS + k) * 3;
          q[0] = X[k].x, q[1] = X[k].y, q[2] = X[k].z;
        }
      }
#pragma unroll
      for (int k = 0; k < NS; ++k) later[k] = X[k];       // S_t of this wave's frames (inclusive)
    }
    {   // this half's loss sums
      const float s2 = ph::wave_sum(acc.sum2), c2 = ph::wave_sum(acc.cnt2), s3 = ph::wave_sum(acc.sum3);
      acc.sum2 = s2, acc.cnt2 = c2, acc.sum3 = s3;
      if (half != 0 && lane == 0) xch[XCH_LOSS] = s2, xch[XCH_LOSS + 1] = c2, xch[XCH_LOSS + 2] = s3;
    }
  }
  ST(27);
  ps.sync();                                       // ---- hand-over 2: the second half's torque sums (and loss sums) are in LDS ----
  if (active) {
    if (half == 0 && lane == 0) {
      // (an asm store like the factor rows': a store hipcc knows about makes the next write to its data registers -- the ds_read
      // at the top of the next clip -- wait for vmcnt(0), i.e. for every factor store of this clip)
      float *pp = d.partials + (size_t)clip * 4;
      const f32x4 pv = {acc.sum2 + xch[XCH_LOSS], acc.cnt2 + xch[XCH_LOSS + 1], acc.sum3 + xch[XCH_LOSS + 2], 0.f};
      asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(pp), "v"(pv) : "memory");
    }
    // ---- pull-back through Gram-Schmidt (closed form, see pose_head_chain_bwd), grad_y^T over y^T ------------------------------
    // g = S rel_rot[t-1]^T enters only through its components in the frame (b1, b2, b3) = the rows of the change c, and
    // rel_rot[t-1] = c^T rel_rot[t]: (g . b_i) = (S rel_rot[t]^T c . b_i) = (S rel_rot[t]^T)_i -- rel_rot[t-1] is never formed
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      const V3 b1 = K::SCAN ? ph::cross(gb2[k], gb3[k]) : gb1[k], b2 = gb2[k], b3 = gb3[k];
      float al, be, ga;
      V3 vv = taup[k];
      if (K::SCAN) {
        V3 Ssum = later[k];
        if (half == 0) {
          const float *q = xch + XCH_TAU + (L.chain * NS + k) * 3;
          Ssum = Ssum + ph::v3(q[0], q[1], q[2]);
        }
        vv = ph::vmulT(Ssum, R[k]);
        al = vv.x, be = vv.y, ga = vv.z;
      } else {
        al = ph::dot(vv, b1), be = ph::dot(vv, b2), ga = ph::dot(vv, b3);
      }
      float gy6[6];
      {
        const float r1 = ph::frcp(gs_n1[k]), r2 = ph::frcp(gs_n2[k]);
        const float k3 = (be + al * gs_d[k] * r2) * r1, k2 = -ga * r1, k5 = -al * r2;
        gy6[0] = fmaf(k3, b3.x, k2 * b2.x), gy6[1] = fmaf(k3, b3.y, k2 * b2.y), gy6[2] = fmaf(k3, b3.z, k2 * b2.z);
        gy6[3] = k5 * b3.x, gy6[4] = k5 * b3.y, gy6[5] = k5 * b3.z;
      }
      const int j = L.start + k < ph::J ? L.start + k : ph::J - 1;
      if (__any(!gs_ok[k])) {         // (rare, wave-uniform) a norm sits on the 1e-12 clamp: generic chain rule through Gram-Schmidt
        const M3 cc = M3{{b1.x, b1.y, b1.z, b2.x, b2.y, b2.z, b3.x, b3.y, b3.z}};
        const V3 g = K::SCAN ? ph::vmul(vv, cc) : vv;                    // S rel_rot[t-1]^T = (S rel_rot[t]^T) c
        ph::SixD s;
        s.a2 = ph::v3(Y[(j * 6 + 3) * PT + (t & 15)], Y[(j * 6 + 4) * PT + (t & 15)], Y[(j * 6 + 5) * PT + (t & 15)]);   // (y is still there)
        s.b1 = b1, s.b2 = b2, s.n1 = gs_n1[k], s.n2 = gs_n2[k], s.d = gs_d[k];
        s.c1 = gs_n1[k] > 1e-12f, s.c2 = gs_n2[k] > 1e-12f;              // (the clamped norms equal the clamp where it applied)
        M3 G;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const V3 ci = ph::v3(cc.m[i * 3], cc.m[i * 3 + 1], cc.m[i * 3 + 2]);
          const V3 h = ph::cross(ci, g) * 0.5f;
          G.m[i * 3] = h.x, G.m[i * 3 + 1] = h.y, G.m[i * 3 + 2] = h.z;
        }
        float slow[6];
        ph::rot6d_bwd(s, G, slow);
#pragma unroll
        for (int i = 0; i < 6; ++i) gy6[i] = gs_ok[k] ? gy6[i] : slow[i];
      }
      if (L.valid[k]) {
#pragma unroll
        for (int i = 0; i < 6; ++i) Y[((L.start + k) * 6 + i) * PT + (t & 15)] = frame_ok ? gy6[i] : 0.f;
      }
    }
    // padding rows of G_L (the dgrad k loop reads them)
    if (half == 0 && lane < (ksteps_bwd(NLAY - 1) * 4 - S::dims(NLAY)) * PT) Y[S::dims(NLAY) * PT + lane] = 0.f;
  }
  ST(28);
  ps.sync();                                       // ---- grad_y^T is complete ----
}

template <int KIND>
__global__ __launch_bounds__(64 * SW) void train_stream_kernel(const p2c_pose_head_desc d, const ph::GradLosses gl, const ClipArgs m) {
  extern __shared__ float lds[];
  const int lane0 = threadIdx.x & 63;
  WLane L;
  L.lane = lane0, L.c = L.lane & 15, L.g = L.lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // Waves 2p and 2p + 1 share a clip and sit on DIFFERENT SIMDs (a workgroup's waves go to the SIMDs cyclically, w and w + 4 to the
  // same one): the tile split of the MLP layers runs on two matrix pipes, and every SIMD hosts one wave of pair p and one of pair
  // p + 2. The pairs run FREE of each other (PairSync, no workgroup barrier in the loop): 250 -> 221 us at B = 8192 against the
  // lockstep version (every rendezvous then waited for the slowest of eight waves).
#ifndef P2C_STREAM_PAIRING
#define P2C_STREAM_PAIRING 0
#endif
  const int pair = P2C_STREAM_PAIRING ? (wave & 3) : (wave >> 1), half = P2C_STREAM_PAIRING ? (wave >> 2) : (wave & 1);
  // (A/B, no gain either way: alternating the larger share of an odd tile split between pairs p and p + 2, which share their two
  // SIMDs -- 200 + 200 instead of 240 + 160 MFMAs per SIMD in layer 5's dgrad: 220 vs 212 us at B = 8192 when always on; on
  // only while every pair has a single clip and the pairs stay in step, B = 1024: 34.5 vs 34.7 us, inside the noise)
  const int role = half;
  float *img = lds;
  float *tab = lds + S::w_total();
  float *scratch = tab + TAB_FLOATS;
  float *Y = scratch + SCRATCH + pair * REGION, *A = Y + ROWS_Y * PT, *Bb = A + ROWS_A * PT;
  ST(0);
  const int T = d.T;
  const int64_t stride = (int64_t)gridDim.x * PAIRS;
  // pair p of workgroup b walks clips b + gridDim (p + 4 i): clip mod 8 = workgroup mod 8 when the grid is a multiple of 8 (the
  // XCD whose L2 the factors stay in).
  int64_t clip = (int64_t)blockIdx.x + (int64_t)gridDim.x * pair;
  const int n_iter = clip < d.B ? (int)(((int64_t)d.B - clip + stride - 1) / stride) : 0;   // this pair's own clips
  // ---- prologue: first x tile, weight image by LDS-DMA (one burst), tables, pair counts -----------------------------------------
  // The x tiles are the SECOND wavefront's job: its vector-memory queue holds loads only, so a wait for a tile never has to
  // drain factor stores (vmcnt retires in issue order, and the compiler's waits in front of loaded values are vmcnt(0)).
  XRegs xr;
  if (half != 0) x_issue(m.x, clip, d.B, T, L.lane, xr);
  // the loss weights (= ph::loss_coefs_n, its loads taken out: behind the barrier they were one more memory round trip)
  const bool any_gl = gl.p[0] || gl.p[1] || gl.p[2];
  const float gl_u0 = gl.p[0] ? *gl.p[0] : 0.f, gl_u1 = gl.p[1] ? *gl.p[1] : 0.f, gl_u2 = gl.p[2] ? *gl.p[2] : 0.f;
  // The weight image arrives in two parts: layers 0 .. 4 (32 KB) before the loop, layer 5 (52 KB, two thirds of the burst) while the
  // first clip's layers 0 .. 4 run. The late pieces are issued LAST in the prologue, from an asm statement: hipcc tracks an LDS-DMA
  // builtin as a pending write to ALL of LDS and puts s_waitcnt vmcnt(0) in front of the next ds_read, whatever it reads. Nothing
  // between their issue and the first layer-5 forward waits on the vector-memory counter (the first clip's targets are loaded
  // and pinned in front of them, the loss coefficients are worked out in front of them, factor stores are never waited for);
  // the wait + workgroup barrier in front of that layer is explicit.
  {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(m.w_image), 0, S::w_total() * 4, 0x00020000);
    const unsigned base = (unsigned)(uintptr_t)(lds_void_ptr)img;
#pragma unroll
    for (int i = 0; i < (IMG_CHUNKS_EARLY + SW - 1) / SW; ++i) {
      const int ck = wave + i * SW;                                  // (wave-uniform)
      if (ck < IMG_CHUNKS_EARLY)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_ptr)(uintptr_t)(base + ck * 1024), 16, (ck * 64 + L.lane) * 16, 0, 0, 0);
    }
  }
  // every load of the prologue is issued before the first one is waited for: ONE memory round trip (a table entry stored to LDS,
  // a count added, a target address formed from a loaded value -- each is a round trip of its own when it sits between the issues)
  PoseIn pin;                                      // targets of the pair's CURRENT clip (the next clip's are requested during the dgrad)
  pin.st = 0;
  pose_inputs(d, n_iter > 0 ? (int)clip : 0, lane0, half, pin);
  constexpr int TAB_ROUNDS = (TAB_FLOATS + 64 * SW - 1) / (64 * SW);
  float tabv[TAB_ROUNDS];
#pragma unroll
  for (int u = 0; u < TAB_ROUNDS; ++u) {
    const int i = (int)threadIdx.x + u * 64 * SW, ic = i < TAB_FLOATS ? i : 0;
    const float *src = ic < TAB_ROT ? d.ref_rel_loc + ic : d.ref_rel_rot + (ic - TAB_ROT);
    tabv[u] = *src;
  }
  {
    float cnt = 0.f;                               // small integers held in floats: exact in any order
    for (int i0 = 0; i0 < d.B; i0 += 8 * 64 * SW) {   // eight loads in flight per thread (a dependent add per load is a round trip each)
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + u * 64 * SW + (int)threadIdx.x;
        v[u] = m.counts[i < d.B ? i : 0];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) cnt += (i0 + u * 64 * SW + (int)threadIdx.x < d.B) ? v[u] : 0.f;
    }
    cnt = ph::wave_sum(cnt);
    if (L.lane == 0) scratch[wave] = cnt;
  }
#pragma unroll
  for (int u = 0; u < TAB_ROUNDS; ++u) {
    const int i = (int)threadIdx.x + u * 64 * SW;
    if (i < TAB_FLOATS) tab[i] = tabv[u];
  }
  if (blockIdx.x == 0 && (int)threadIdx.x < m.n_counters) m.counters[threadIdx.x] = 0;   // arrival tickets of train_wgrad_kernel
  if (threadIdx.x < SW) reinterpret_cast<int *>(scratch + 8)[threadIdx.x] = 0;            // the waves' rendezvous counters
  if (half != 0) x_commit(xr, x_offsets(L.lane), Y, L.lane);       // H_0 of the pair's first clip
  ST(40);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  pose_inputs_landed(pin);
  __syncthreads();
  ST(41);
  float coef2 = 0.f, coef3 = 0.f;
  {
    float n2 = 0.f;
    for (int w = 0; w < SW; ++w) n2 += scratch[w];
    const float n3 = ph::n3_elems(d), g2 = gl_u0 + gl_u2, g3 = gl_u1 + gl_u2;
    if (any_gl) {
      coef2 = (d.gt2d && n2 > 0.f) ? g2 / n2 : 0.f;
      coef3 = (d.gt3d && n3 > 0.f) ? 2.f * g3 / n3 : 0.f;
    }
  }
  ST(42);
  // whatever the prologue loaded has landed, and hipcc knows (the builtin is a real s_waitcnt to its wait-count pass, an asm string
  // is not): a load left pending on some path would come back as a vmcnt(0) in front of the loop's first ds_read
  __builtin_amdgcn_s_waitcnt(0x0F70);              // vmcnt(0)
  {
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    const uintptr_t wi = reinterpret_cast<uintptr_t>(m.w_image);
    i32x4 rs;                                      // = the resource above: base, no stride, bytes, raw 32-bit data format
    rs.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)wi), rs.y = __builtin_amdgcn_readfirstlane((int)((wi >> 32) & 0xffffu));
    rs.z = S::w_total() * 4, rs.w = 0x00020000;
    const unsigned base = (unsigned)(uintptr_t)(lds_void_ptr)img;
    constexpr int total4 = S::w_total() >> 2;
#pragma unroll
    for (int i = 0; i < (IMG_CHUNKS - IMG_CHUNKS_EARLY + SW - 1) / SW; ++i) {
      const int ck = IMG_CHUNKS_EARLY + wave + i * SW;               // (wave-uniform)
      if (ck < IMG_CHUNKS && ck * 64 + L.lane < total4) {            // the last piece is partial: its tail lanes stay out
        const unsigned dst = __builtin_amdgcn_readfirstlane(base + ck * 1024);
        const int voff = (ck * 64 + L.lane) * 16;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(dst), "v"(voff), "s"(rs) : "memory");
      }
    }
  }

  ST(1);
  PairSync ps;
  ps.mine = reinterpret_cast<int *>(scratch + 8) + wave, ps.other = reinterpret_cast<int *>(scratch + 8) + (wave ^ (P2C_STREAM_PAIRING ? 4 : 1)), ps.epoch = 0;
#ifndef P2C_STREAM_STAGGER
#define P2C_STREAM_STAGGER 0        // x 8 k cycles of head start for pairs 0, 1 (A/B timing: 0 / 3 / 6 / 12 -> 221 / 227 / 234 / 246 us at
#endif                              // B = 8192: the steady state does not care how the pairs of a SIMD are phased; the delay is just lost)
  if (P2C_STREAM_STAGGER > 0 && pair >= 2 && n_iter >= 2)
    for (int i = 0; i < P2C_STREAM_STAGGER; ++i) __builtin_amdgcn_s_sleep(127);
  for (int it = 0; it < n_iter; ++it, clip += stride) {
    ST(2);
    const bool active = true;
    float *fdst = m.factors + (size_t)((active && !(P2C_STREAM_EXPERIMENT & 8)) ? clip : blockIdx.x * PAIRS + pair) * F_ROWS * 16;
    uint64_t mask = 0;
    const bool mlp = active && !(P2C_STREAM_EXPERIMENT & 2);
    {
      // ---- LinearAE forward, the pair's two wavefronts side by side (one barrier per layer); every H_l leaves for the factor
      // block as soon as it exists ---------------------------------------------------------------------------------------------------
      WLane L;
      L.lane = fresh(lane0), L.c = L.lane & 15, L.g = L.lane >> 4;
      ST(3);
#if P2C_STREAM_PRIO
      __builtin_amdgcn_s_setprio(P2C_STREAM_PRIO);
#endif
      HalfRows<S::dims(0)> h0;
      if (mlp) {
        half_read(Y, L.lane, half, h0);
        fwd_layer<0>(L, role, img, Y, A, mask);
        half_store(h0, fdst + f_h_off(0) * 16, L.lane, half);
      }
      ps.sync();     
      ST(4);
      HalfRows<S::dims(1)> h1;
      if (mlp) {
        half_read(A, L.lane, half, h1);
        fwd_layer<1>(L, role, img, A, Bb, mask);
        half_store(h1, fdst + f_h_off(1) * 16, L.lane, half);
      }
      ps.sync();     
      ST(5);
      HalfRows<S::dims(2)> h2;
      if (mlp) {
        half_read(Bb, L.lane, half, h2);
        fwd_layer<2>(L, role, img, Bb, A, mask);
        half_store(h2, fdst + f_h_off(2) * 16, L.lane, half);
      }
      ps.sync();     
      ST(6);
      HalfRows<S::dims(3)> h3;
      if (mlp) {
        half_read(A, L.lane, half, h3);
        fwd_layer<3>(L, role, img, A, Bb, mask);
        half_store(h3, fdst + f_h_off(3) * 16, L.lane, half);
      }
      ps.sync();     
      ST(7);
      HalfRows<S::dims(4)> h4;
      if (mlp) {
        half_read(Bb, L.lane, half, h4);
        fwd_layer<4>(L, role, img, Bb, A, mask);
        half_store(h4, fdst + f_h_off(4) * 16, L.lane, half);
      }
      ps.sync();     
      ST(8);
      if (it == 0) {                               // layer 5's part of the image: every wave's pieces have landed (once per kernel)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
      }
      HalfRows<S::dims(5)> h5;
      if (mlp) {
        half_read(A, L.lane, half, h5);
        fwd_layer<5>(L, role, img, A, Y, mask);
        half_store(h5, fdst + f_h_off(5) * 16, L.lane, half);
      }
      ST(9);
    }
    ps.sync();                                      // ---- y^T is complete (and buffer A is free: the pose head's hand-over scratch) ----
    // ---- pose head forward + backward by both wavefronts of the pair: y^T -> grad_y^T in place --------------------------------------
#if P2C_STREAM_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    pose_phase<KIND>(d, active && !(P2C_STREAM_EXPERIMENT & 1), (int)clip, lane0, half, pin, Y, A, tab, coef2, coef3, ps);
    ST(32);
    {
      WLane L;
      L.lane = fresh(lane0), L.c = L.lane & 15, L.g = L.lane >> 4;
#if P2C_STREAM_PRIO
      __builtin_amdgcn_s_setprio(P2C_STREAM_PRIO);
#endif
      // the pair's next clip: its x tile and targets are requested here and pinned behind the first dgrad layer -- the one point of
      // the loop where this wave's vector-memory queue holds nothing recent (vmcnt retires in issue order: a load waited for behind
      // factor stores waits for all of them)
      if (half != 0) x_issue(m.x, clip + stride, d.B, T, L.lane, xr);
      pose_inputs(d, (int)(clip + stride < d.B ? clip + stride : clip), lane0, half, pin);
      // ---- dgrad chain, side by side; every G_l leaves as soon as it exists ----------------------------------------------------------
      HalfRows<S::dims(6)> g6;
      if (mlp) {
        half_read(Y, L.lane, half, g6);
        dgrad_layer<5>(L, role, img, Y, A, mask);
        pose_inputs_landed(pin);
        if (half != 0) x_landed(xr);
        half_store(g6, fdst + f_g_off(6) * 16, L.lane, half);
      }
      ps.sync();                                    // ---- grad_y^T has been read: the y rows are free for the next clip's H_0 ----
      ST(33);
      HalfRows<S::dims(5)> g5;
      if (mlp) {
        half_read(A, L.lane, half, g5);
        dgrad_layer<4>(L, role, img, A, Bb, mask);
        half_store(g5, fdst + f_g_off(5) * 16, L.lane, half);
      }
      if (half != 0 && clip + stride < d.B) x_commit(xr, x_offsets(L.lane), Y, L.lane);   // (its share of this layer is the small one)
      ps.sync();     
      ST(34);
      HalfRows<S::dims(4)> g4;
      if (mlp) {
        half_read(Bb, L.lane, half, g4);
        dgrad_layer<3>(L, role, img, Bb, A, mask);
        half_store(g4, fdst + f_g_off(4) * 16, L.lane, half);
      }
      ps.sync();     
      ST(35);
      HalfRows<S::dims(3)> g3;
      if (mlp) {
        half_read(A, L.lane, half, g3);
        dgrad_layer<2>(L, role, img, A, Bb, mask);
        half_store(g3, fdst + f_g_off(3) * 16, L.lane, half);
      }
      ps.sync();     
      ST(36);
      HalfRows<S::dims(2)> g2;
      if (mlp) {
        half_read(Bb, L.lane, half, g2);
        dgrad_layer<1>(L, role, img, Bb, A, mask);
        half_store(g2, fdst + f_g_off(2) * 16, L.lane, half);
      }
      ps.sync();     
      HalfRows<S::dims(1)> g1;
      if (mlp) {
        half_read(A, L.lane, half, g1);
        half_store(g1, fdst + f_g_off(1) * 16, L.lane, half);
      }
      ST(37);
    }
    ps.sync();                                      // ---- the next clip's H_0 is in place, buffer A has been read ----
    ST(63);
  }
  if (n_iter == 0) {                               // (a pair without a clip: the workgroup's second barrier counts every wave)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
}

}  // namespace p2c_stream

using namespace p2c_stream;

#ifdef P2C_STREAM_TRACE
extern "C" P2C_API int p2c_debug_stream_trace(unsigned long long *out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(p2c_stream::g_strace), sizeof(unsigned long long) * 64);
}
#endif

bool p2c_internal_train_stream_supported(const p2c_pose_head_desc &d) {
  if (d.kind != P2C_KIND_POSE_CHANGES_6D && d.kind != P2C_KIND_RELATIVE_ROT_6D) return false;
  if (d.dloc || d.drot || d.gt_rot) return false;
  if (d.T < 1 || d.T > 16) return false;
  if (d.transform != P2C_TRANSFORM_NONE && (d.n_hips != 1 || d.n_neck != 1 || d.hips_idx[0] != ch::HIPS || d.neck_idx[0] != ch::NECK))
    return false;
  if (d.gt2d && (d.gt2d_joints != P2C_JOINTS || d.gt2d_channels != 2)) return false;
  if (d.gt3d && d.gt3d_joints != P2C_JOINTS) return false;
  for (int j = 0; j < P2C_JOINTS; ++j)
    if ((d.gt2d && d.gmap2d[j] != j) || (d.gt3d && d.gmap3d[j] != j)) return false;
  return true;
}

int p2c_internal_train_stream_launch(const p2c_pose_head_desc &d, const p2c::GradLosses &gl, const ClipArgs &m, hipStream_t stream) {
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void *)train_stream_kernel<P2C_KIND_POSE_CHANGES_6D>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)train_stream_kernel<P2C_KIND_RELATIVE_ROT_6D>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  // every CU gets a workgroup as soon as there are 256 clips: pair p of workgroup b walks clips b + grid (p + 4 i), so with fewer
  // than four clips per CU the later pairs of every workgroup stay idle (B = 512: two pairs per CU, one per SIMD pair, instead of
  // four pairs on half the CUs)
  constexpr int kCUs = 256;
  const dim3 grid((unsigned)(d.B < kCUs ? d.B : kCUs));
  const size_t lds_bytes = (size_t)LDS_FLOATS * sizeof(float);
  if (d.kind == P2C_KIND_POSE_CHANGES_6D)
    hipLaunchKernelGGL(train_stream_kernel<P2C_KIND_POSE_CHANGES_6D>, grid, dim3(64 * SW), lds_bytes, stream, d, gl, m);
  else
    hipLaunchKernelGGL(train_stream_kernel<P2C_KIND_RELATIVE_ROT_6D>, grid, dim3(64 * SW), lds_bytes, stream, d, gl, m);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}
