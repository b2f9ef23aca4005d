// p2c_mlp.hip -- fused small-MLP (LinearAE) forward / backward for gfx950 on fp32 MFMA (v_mfma_f32_16x16x4_f32).
//
// The reference's LinearAE (modules/movements/linear_ae/linear_ae.py:25-59) is a per-frame MLP
// 52 -> 26 -> 13 -> 6 -> O/4 -> O/2 -> O with ReLU between (17 530 parameters for O = 156). As ATen ops that is ~50
// launches per train step (6 GEMMs with K <= 78, bias/ReLU/bias-grad kernels), each a few microseconds of pure launch
// latency: at the benchmark's B = 256 they are 85 % of the step. Here the whole stack is ONE launch forward and ONE
// backward (+ a small deterministic reduction of the per-workgroup weight-gradient partials).
//
// Structure ("cooperative 16-sample tile"):
//   * a workgroup of eight wavefronts walks over tiles of 16 frames ("samples"). Activations live TRANSPOSED in LDS,
//     H^T[n][sample]: the sample index sits on lane&15 for the MFMA B operand (B[k][col]: lane = col + 16*k) *and* for
//     the C/D tile (col = lane&15, row = 4*(lane>>4)+reg), so layer l+1 reads what layer l wrote with plain
//     ds_read_b32 -- no transpose anywhere. The 16-row output tiles of a layer are dealt round-robin to the eight waves
//     (one barrier per layer): the dependent-MFMA chain a single wave would walk is what bounds small batches;
//   * ALL weights are staged once per workgroup into a zero-padded LDS image [pad16(n_out)][pitch] with the bias in
//     column n_in and the activations carrying a constant-one row n_in: the inner loops are select-free
//     (2 ds_read + 1 MFMA per k-step), for W (forward) and W^T (dgrad) alike. Pitch == 2 (mod 4) floats puts the 16 rows
//     of an A fragment on 16 distinct banks. (Reading weights per k-step from L2 instead makes all 256 CUs request the
//     same cache lines in lock step -- measured 4x slower.)
//   * exact fp32: the MFMA is bit-for-bit an fmaf chain in k order (cdna_hip_programming.md §3), no bf16 anywhere;
//   * backward recomputes the activations (no HBM round trip), runs the dgrad chain, then the waves split the 16x16
//     tiles of dW_aug = G^T [H | 1] and keep them in MFMA accumulators across the persistent tile loop; per-workgroup
//     partials are reduced in fixed order (bitwise reproducible, no atomics).
// Regime-specific variants of the same arithmetic (all bit-identical where they overlap, selected by the row count):
//   * about one sample tile per CU: split weight gradient -- the backward leaves factors, mlp_wgrad_kernel contracts them
//     with an XCD-local K split, mlp_reduce_small_kernel adds the 8 partials and applies AdamW (split_wgrad());
//   * two or more tiles per workgroup: the forward saves its hidden activations, the backward loads them instead of
//     recomputing (save_activations());
//   * four or more tiles per CU: wave-per-tile forward without workgroup barriers (mlp_fwd_wave_kernel, wave_forward()).
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <stdint.h>
#include <type_traits>

#include "../../include/p2c.h"
#include "p2c_adam_math.h"

#include "p2c_mlp_dev.h"

namespace p2c_mlp {

// Pack kernel (once per forward): the zero-padded image of every [W_l | b_l] -- rows 0..rows_l-1, pitch ld_l, bias in
// column n_in, and a unit row n_out that copies the constant-one input row to the output (so that the next layer finds
// its ones row without any select in the epilogue) -- laid out exactly as the workgroups want it in LDS.
__global__ __launch_bounds__(256) void mlp_pack_kernel(const MlpArgs a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.w_total) return;
  int l = 0;
  while (l + 1 < a.n_layers && i >= a.w_off[l + 1]) ++l;
  const int n_in = a.dims[l], n_out = a.dims[l + 1], ld = a.ld[l];
  const int j = i - a.w_off[l];
  const int n = j / ld, k = j - n * ld;
  float v = 0.f;
  if (n < n_out) {
    if (k < n_in) v = a.W[l][n * n_in + k];
    else if (k == n_in) v = a.b[l][n];
  } else if (n == n_out && k == n_in) {
    v = 1.f;
  }
  a.w_image[i] = v;
}

// ---- forward ---------------------------------------------------------------------------------------------------------
// LDS: [weight images | H_0 .. H_{L-1}]   (the input rows of layer l are H_l, its output H_{l+1})
template <class S, int P = P2C_PREC_F32>
__global__ __launch_bounds__(64 * WAVES) void mlp_fwd_kernel(const MlpArgs a) {
  extern __shared__ float lds[];
  const S sh(a);
  Lane L;
  L.lane = threadIdx.x & 63, L.c = L.lane & 15, L.g = L.lane >> 4;
  L.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // in an SGPR: tile offsets run on the scalar unit
  const int nl = sh.n_layers();
  const int total4 = sh.w_total() >> 2;
  TR(0, 0);
  float *H = lds + sh.w_total();
  const int64_t n_tiles = (a.N + TS - 1) / TS;
  TileRegs xr;
  ImageRegs wr;
  tile_issue(a.x, (int64_t)blockIdx.x * TS, a.N, sh.dims(0), a.vec_x != 0, xr);   // first tile, then the image behind it
  if constexpr (S::kStatic) stage_issue(a.w_image, total4, wr, 0, 0, issue_mark<S>(1));
  else stage_issue(a.w_image, total4, wr);
  init_rows(H, sh.h_off(0) + sh.dims(0), sh.h_off(0) + k_rows(sh.dims(0)), sh.h_off(0) + sh.dims(0));
  if constexpr (!S::kStatic) {
    stage_commit(total4, wr, lds);
    stage_rest(a.w_image, total4, wr, lds);
  }
  TR(0, 1);
  // the first tile of a workgroup is peeled (compile-time flag): only there the image registers are live
  auto one_tile = [&](int64_t tile, auto first_c) {
    constexpr bool first = decltype(first_c)::value;
    const int64_t row0 = tile * TS, row = row0 + L.c;
    const bool row_ok = row < a.N;
    TR(0, 2);
    tile_commit(sh.dims(0), a.vec_x != 0, xr, H + sh.h_off(0) * TP);
    tile_issue(a.x, (tile + gridDim.x) * TS, a.N, sh.dims(0), a.vec_x != 0, xr);    // prefetch the next tile of this block
    for_layers(sh, 0, nl, [&](int l) {
      if constexpr (S::kStatic) {   // the image rounds this layer reads, as late as possible
        if constexpr (first) stage_commit(total4, wr, lds, 0, l == 0 ? 0 : rounds_upto<S>(l - 1), rounds_upto<S>(l));
      }
      lds_barrier();
      if constexpr (S::kStatic && first) stage_issue(a.w_image, total4, wr, 0, issue_mark<S>(l + 1), issue_mark<S>(l + 2));
      TR(0, 3 + l);
      const bool last = (l == nl - 1);
      layer_forward<P>(L, lds + sh.w_off(l), sh.ld(l), sh.dims(l), sh.dims(l + 1), !last, H + sh.h_off(l) * TP,
                    last ? nullptr : H + sh.h_off(l + 1) * TP, last ? a.y + row * sh.dims(l + 1) : nullptr, row_ok,
                    a.vec_y != 0);
    });
    TR(0, 12);
    // saved activations for the backward: every H_l is complete (the last layer's barrier came after H_{L-1} was written)
    // and stays untouched until the next tile's layer 0 has passed its barrier
    if (a.saved) acts_store(sh, a, H, reinterpret_cast<f32x4 *>(a.saved) + (size_t)tile * a.f_half * 4);
    // the next tile's x rows overwrite H_0 only after every wave has passed layer 0's barrier chain: the barrier of
    // layer 1 (or, for a single layer, the one below) orders them
    if (nl == 1) lds_barrier();
  };
  if ((int64_t)blockIdx.x < n_tiles) one_tile(blockIdx.x, std::true_type{});
  for (int64_t tile = (int64_t)blockIdx.x + gridDim.x; tile < n_tiles; tile += gridDim.x) one_tile(tile, std::false_type{});
  TR(0, 39);
}

// ---- forward, many sample tiles (wave-per-tile) -----------------------------------------------------------------------
// The cooperative kernel above splits ONE 16-sample tile over eight waves: right for a handful of tiles per CU, where the
// dependent MFMA chain of a tile is the critical path, but it spends a barrier per layer and leaves most waves idle in
// the narrow layers (fp32 MFMA utilisation ~35 %). With many tiles per CU the throughput form is the opposite one: every
// wave walks its OWN tile through all layers -- no workgroup barrier at all after the weight image is staged, activations
// ping-pong between two wave-private LDS buffers, two output tiles of a layer in flight per wave so that the MFMA pipe
// issues back to back. Four waves per workgroup: 84 KB image + 4 x 12 KB of activations (six waves fit the LDS but run slower).
// fp32 MFMA utilisation 35 % -> 50 % at 131 072 frames; what is left is the latency at the head of every output-tile pair with
// one wave per SIMD.
#ifndef P2C_FW_WAVES
#define P2C_FW_WAVES 4
#endif
constexpr int FW_WAVES = P2C_FW_WAVES;
constexpr int FW_XU = (TS * (MAXW - 1) + 63) / 64;   // x-tile floats per lane (static shapes use dims(0) * 16 / 64 of them)
template <class S, int P = P2C_PREC_F32>
__global__ __launch_bounds__(64 * FW_WAVES) void mlp_fwd_wave_kernel(const MlpArgs a) {
  extern __shared__ float lds[];
  const S sh(a);
  Lane L;
  L.lane = threadIdx.x & 63, L.c = L.lane & 15, L.g = L.lane >> 4, L.wave = 0;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nl = sh.n_layers(), n0 = sh.dims(0);
  {
    f32x4 *dst = reinterpret_cast<f32x4 *>(lds);
    const f32x4 *src = reinterpret_cast<const f32x4 *>(a.w_image);
    const int total4 = sh.w_total() >> 2;
    constexpr int NT4 = 64 * FW_WAVES, BATCH = 8;             // eight 16-byte loads of a thread in flight per round
    for (int i0 = threadIdx.x; i0 < total4; i0 += NT4 * BATCH) {
      f32x4 v[BATCH];
#pragma unroll
      for (int u = 0; u < BATCH; ++u) v[u] = (i0 + u * NT4 < total4) ? src[i0 + u * NT4] : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int u = 0; u < BATCH; ++u)
        if (i0 + u * NT4 < total4) dst[i0 + u * NT4] = v[u];
    }
  }
  float *bufA = lds + sh.w_total() + wave * (a.fw_rows_a + a.fw_rows_b) * TP, *bufB = bufA + a.fw_rows_a * TP;
  for (int i = L.lane; i < (a.fw_rows_a + a.fw_rows_b) * TP; i += 64) bufA[i] = 0.f;
  __syncthreads();                                           // the only workgroup barrier: the image is in place
  const int64_t n_tiles = (a.N + TS - 1) / TS;
  const int per = TS * n0;                                   // floats of one x tile
  const int64_t tstep = (int64_t)gridDim.x * FW_WAVES;
  // the next tile's x rows wait in registers while this one is computed (one wave per SIMD: nothing else hides HBM latency)
  float xr[FW_XU];
  auto x_issue = [&](int64_t tile) {
    const int64_t row0 = tile * TS, left = a.N - row0;
    const int valid = left <= 0 ? 0 : (int)(left < TS ? left : TS) * n0;
    const float *xp = a.x + row0 * n0;
#pragma unroll
    for (int u = 0; u < FW_XU; ++u) {
      const int e = L.lane + 64 * u;
      if (64 * u < per) xr[u] = e < valid ? xp[e] : 0.f;
    }
  };
  x_issue((int64_t)blockIdx.x * FW_WAVES + wave);
  for (int64_t tile = (int64_t)blockIdx.x * FW_WAVES + wave; tile < n_tiles; tile += tstep) {
    const int64_t row0 = tile * TS, row = row0 + L.c;
    const bool row_ok = row < a.N;
#pragma unroll
    for (int u = 0; u < FW_XU; ++u) {                         // transposed LDS write of the x tile
      const int e = L.lane + 64 * u;
      if (64 * u < per && e < per) {
        const int sidx = e / n0, k = e - sidx * n0;
        bufA[k * TP + sidx] = xr[u];
      }
    }
    x_issue(tile + tstep);
    for (int i = n0 * TP + L.lane; i < k_rows(n0) * TP; i += 64) bufA[i] = (i / TP == n0) ? 1.f : 0.f;   // ones row, k rounding
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for_layers(sh, 0, nl, [&](int l) {
      const bool last = (l == nl - 1);
      const float *in = (l & 1) ? bufB : bufA;
      float *out = (l & 1) ? bufA : bufB;
      const int n_in = sh.dims(l), n_out = sh.dims(l + 1);
      const int ksteps = (((n_in + 1 + 3) >> 2) + 3) & ~3, ntiles = (n_out + 16) >> 4;
      const float *wl = lds + sh.w_off(l);
      float *y_row = last ? a.y + row * n_out : nullptr;
      for (int nt = 0; nt < ntiles; nt += 2) {
        if (nt + 1 < ntiles)
          layer_forward_nt<2, 1, P>(L, wl, sh.ld(l), ksteps, nt, n_out, !last, in, last ? nullptr : out, y_row, row_ok, a.vec_y != 0);
        else
          layer_forward_nt<1, 1, P>(L, wl, sh.ld(l), ksteps, nt, n_out, !last, in, last ? nullptr : out, y_row, row_ok, a.vec_y != 0);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (!last && a.saved) {                                // saved activations of this tile (see acts_store)
        f32x4 *hd = reinterpret_cast<f32x4 *>(a.saved) + ((size_t)tile * a.f_half + a.f_off[l + 1]) * 4;
        for (int i = L.lane; i < n_out * 4; i += 64) {
          const int o = (i >> 2) * TP + (i & 3) * 4;
          hd[i] = (f32x4){out[o], out[o + 1], out[o + 2], out[o + 3]};
        }
      }
    });
  }
}

// ---- backward --------------------------------------------------------------------------------------------------------
// LDS: [weight images | H_0 .. H_{L-1} | G_1 .. G_L (G_L = gy tile)]
template <class S, bool FACTORS = false, bool SAVED = false, int P = P2C_PREC_F32>
__global__ __launch_bounds__(64 * WAVES) void mlp_bwd_kernel(const MlpArgs a) {
  extern __shared__ float lds[];
  const S sh(a);
  Lane L;
  L.lane = threadIdx.x & 63, L.c = L.lane & 15, L.g = L.lane >> 4;
  L.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // in an SGPR: tile offsets run on the scalar unit
  const int nl = sh.n_layers();
  const int total4 = sh.w_total() >> 2;
  TR(1, 0);
  float *H = lds + sh.w_total();
  float *G = H + (sh.h_off(nl) - sh.h_off(1)) * TP;          // G_l lives at row h_off(l) of this base (l = 1..L)
  TileRegs xr, gr;
  ImageRegs wr;
  tile_issue(a.x, (int64_t)blockIdx.x * TS, a.N, sh.dims(0), a.vec_x != 0, xr);   // needed first; the image in layer
  if constexpr (S::kStatic && !SAVED) stage_issue(a.w_image, total4, wr, 0, 0, issue_mark<S>(1));   // order behind it; gy (first
  else stage_issue(a.w_image, total4, wr);                                                // read by the dgrad chain) last
  if constexpr (static_layers<S>() < 4 || SAVED)
    tile_issue(a.gy, (int64_t)blockIdx.x * TS, a.N, sh.dims(nl), a.vec_gy != 0, gr);
  ActRegs ar;
  if constexpr (SAVED) {
    if constexpr (S::kStatic) acts_issue(sh, a, blockIdx.x, (a.N + TS - 1) / TS, ar);
    // rows the saved activations never overwrite: the constant-one row behind each H_l and the zero rows of the k rounding
    for_layers(sh, 1, nl, [&](int l) {
      init_rows(H, sh.h_off(l) + sh.dims(l), sh.h_off(l) + act_rows_of(sh.dims(l)), sh.h_off(l) + sh.dims(l));
    });
  }
  init_rows(H, sh.h_off(0) + sh.dims(0), sh.h_off(0) + k_rows(sh.dims(0)), sh.h_off(0) + sh.dims(0));
  init_rows(G, sh.h_off(nl) + sh.dims(nl), sh.h_off(nl) + pad16(sh.dims(nl)), -1);
  if constexpr (!S::kStatic) {
    stage_commit(total4, wr, lds);
    stage_rest(a.w_image, total4, wr, lds);
  }
  const int lane_off = L.c * TP + L.g;
  TR(1, 1);

  f32x4 acc[MAX_SLOTS];
#pragma unroll
  for (int i = 0; i < MAX_SLOTS; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int64_t n_tiles = (a.N + TS - 1) / TS;
  f32x4 *part = reinterpret_cast<f32x4 *>(a.partials) + (size_t)blockIdx.x * a.n_tiles_w * 64;
  auto one_tile = [&](int64_t tile, auto first_c) {   // first tile peeled: only there the image registers are live
    constexpr bool first = decltype(first_c)::value;
    const int64_t row0 = tile * TS;
    lds_barrier();                         // previous tile's dW phase has consumed H and G
    tile_commit(sh.dims(0), a.vec_x != 0, xr, H + sh.h_off(0) * TP);
    tile_issue(a.x, row0 + (int64_t)gridDim.x * TS, a.N, sh.dims(0), a.vec_x != 0, xr);   // prefetch this block's next tile
    if constexpr (SAVED) {
      // ---- phase F replaced: H_1 .. H_{L-1} as the forward left them (bit-identical to a recomputation)
      TR(1, 2);
      if constexpr (S::kStatic) {
        acts_commit(sh, ar, H);
        acts_issue(sh, a, tile + gridDim.x, n_tiles, ar);
        if constexpr (first) stage_commit(total4, wr, lds);
      } else {
        acts_copy(sh, a, tile, H);
      }
    } else {
      // ---- phase F: activations H_1 .. H_{L-1}
      TR(1, 2);
      for_layers(sh, 0, nl - 1, [&](int l) {
        if constexpr (S::kStatic) {   // the image rounds this layer reads, as late as possible
          if constexpr (first) stage_commit(total4, wr, lds, 0, l == 0 ? 0 : rounds_upto<S>(l - 1), rounds_upto<S>(l));
        }
        lds_barrier();
        if constexpr (S::kStatic && first) {
          stage_issue(a.w_image, total4, wr, 0, issue_mark<S>(l + 1), issue_mark<S>(l + 2));
          if (static_layers<S>() >= 4 && l == 2) tile_issue(a.gy, row0, a.N, sh.dims(nl), a.vec_gy != 0, gr);   // behind the image
        }
        TR(1, 3 + l);
        layer_forward<P>(L, lds + sh.w_off(l), sh.ld(l), sh.dims(l), sh.dims(l + 1), true, H + sh.h_off(l) * TP,
                      H + sh.h_off(l + 1) * TP, nullptr, false, false);
      });
      // the last layer's image (first use: the head of the dgrad chain) and the gy tile arrive behind the recomputation
      if constexpr (S::kStatic) {
        if constexpr (first) stage_commit(total4, wr, lds, 0, nl >= 2 ? rounds_upto<S>(nl - 2) : 0, STAGE_U);
      }
    }
    tile_commit(sh.dims(nl), a.vec_gy != 0, gr, G + sh.h_off(nl) * TP);
    tile_issue(a.gy, row0 + (int64_t)gridDim.x * TS, a.N, sh.dims(nl), a.vec_gy != 0, gr);
    // ---- phase D: G_l = relu'(H_l) .* (W_l^T G_{l+1}), l = L-1 .. 1
    for_layers_down(sh, nl - 1, 1, [&](int l) {
      lds_barrier();
      TR(1, 12 + l);
      layer_dgrad<P>(L, lds + sh.w_off(l), sh.ld(l), sh.dims(l), sh.dims(l + 1), G + sh.h_off(l + 1) * TP, H + sh.h_off(l) * TP,
                  G + sh.h_off(l) * TP);
    });
    lds_barrier();
    TR(1, 22);
    if constexpr (FACTORS) {
      // ---- split weight gradient: the tile's activations and output gradients go out as they sit in LDS (transposed,
      // 16 samples = 64 B per row); mlp_wgrad_kernel contracts them over ALL samples. H_0 = x and G_L = gy are in HBM.
      f32x4 *fdst = reinterpret_cast<f32x4 *>(a.factors) + (size_t)tile * a.f_rows * 4;
      for_layers(sh, 1, nl, [&](int l) {
        const int rows4 = sh.dims(l) * 4;
        const float *hsrc = H + sh.h_off(l) * TP, *gsrc = G + sh.h_off(l) * TP;
        f32x4 *hd = fdst + a.f_off[l] * 4, *gd = fdst + (a.f_half + a.f_off[l]) * 4;
        for (int i = threadIdx.x; i < rows4; i += NTH) {
          const int o = (i >> 2) * TP + (i & 3) * 4;
          hd[i] = (f32x4){hsrc[o], hsrc[o + 1], hsrc[o + 2], hsrc[o + 3]};
          gd[i] = (f32x4){gsrc[o], gsrc[o + 1], gsrc[o + 2], gsrc[o + 3]};
        }
      });
    } else {
    // ---- phase W: dW_aug_l[n][m] += sum_s G_{l+1}^T[n][s] * H_l^T_aug[m][s]; tile t = slot * WAVES + wave.
    // Branch-free: slots past the last tile alias tile 0 and are never written out. Samples beyond N carry G = 0.
    // On the workgroup's last sample tile every finished slot goes straight out as this workgroup's partial gradient
    // tile (tile-major in MFMA C layout: partials[block][tile][lane][4], one coalesced 16-byte store per lane;
    // mlp_reduce_kernel maps them to the parameter tensors): the 80 KB of stores drain behind the remaining MFMAs.
    const bool last_tile = tile + gridDim.x >= n_tiles;
#pragma unroll
    for (int slot = 0; slot < MAX_SLOTS; ++slot) {
      const int t = slot * WAVES + L.wave;
      const float *gp = H + a.tab[2 * t] + lane_off;       // A[n][k = sample]   (scalar loads from the kernel arguments)
      const float *hp = H + a.tab[2 * t + 1] + lane_off;   // B[k = sample][m]
      float av[4], bv[4];
#pragma unroll
      for (int s = 0; s < TS / 4; ++s) av[s] = gp[4 * s], bv[s] = hp[4 * s];
      static_assert(TS == 16, "one group of four k-steps per sample tile");
      acc[slot] = mfma_k16<P>(av, bv, acc[slot]);
      if (last_tile && t < a.n_tiles_w) __builtin_nontemporal_store(acc[slot], &part[t * 64 + L.lane]);   // read once, by another kernel
    }
    }
  };
  if ((int64_t)blockIdx.x < n_tiles) one_tile(blockIdx.x, std::true_type{});
  for (int64_t tile = (int64_t)blockIdx.x + gridDim.x; tile < n_tiles; tile += gridDim.x) one_tile(tile, std::false_type{});
  TR(1, 23);
  if (!FACTORS && (int64_t)blockIdx.x >= n_tiles) {   // empty batch: this workgroup saw no sample tile, its partial is zero
#pragma unroll
    for (int slot = 0; slot < MAX_SLOTS; ++slot) {
      const int t = slot * WAVES + L.wave;
      if (t < a.n_tiles_w) part[t * 64 + L.lane] = acc[slot];
    }
  }
  TR(1, 39);
}

// ---- split weight gradient (small batches: about one sample tile per CU) ---------------------------------------------
// With one 16-sample tile per workgroup the fused wgrad phase leaves 256 x 80 KB of partial dW tiles (20 MB written, 20 MB
// re-read by the reduction) for 70 KB of gradient. Here the backward leaves its FACTORS instead (20 KB per sample tile) and
// this kernel contracts them over all samples: workgroup (t, q) owns dW tile t and every KS-th group of WAVES sample
// tiles; wave w walks sample tiles q * WAVES + w, + KS * WAVES, ... with one accumulator, the eight waves are added in
// wave order through LDS, and the KS partial tiles per dW tile go to mlp_reduce_kernel (n_blocks = KS) in the usual
// tile-major layout. Fixed order everywhere: bitwise reproducible. Lane (r, k) of the MFMA A / B operand holds samples
// 4k .. 4k+3 of the tile as one float4 and feeds component i to MFMA step i -- the contraction runs over the 16 samples in
// the order (4k + i), the same bijection on both operands.
constexpr int WG_UNROLL = 4;
#ifndef P2C_WGRAD_WAVES
#define P2C_WGRAD_WAVES 8
#endif
constexpr int WGW = P2C_WGRAD_WAVES;   // waves per workgroup of the contraction
// `o` / `coefs_out` (optimizer in backward): the double-precision bias corrections of this step are worked out here, by
// one thread, while the contraction runs -- the reduction that follows only loads them.
template <int P = P2C_PREC_F32>
__global__ __launch_bounds__(64 * WGW) void mlp_wgrad_kernel(const MlpArgs a, int n_stiles, int ks, const p2c_adamw_desc o,
                                                               p2c_optim::Coefs *coefs_out) {
  __shared__ f32x4 red[WGW][64];
  const int lane = threadIdx.x & 63, r = lane & 15, k = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // consecutive workgroups land on consecutive XCDs: q = blockIdx % ks (ks = 8 XCDs) makes XCD q read only the sample
  // tiles st = q (mod 8) -- the ones the backward's workgroups ON THE SAME XCD just wrote, still in its L2 -- instead of
  // every XCD pulling all factors through its own L2 (measured: 11.4 -> see DESIGN.md)
  const int q = blockIdx.x % ks, t = blockIdx.x / ks;
  const TileRef tr = locate_tile(a.dims, t);
  const int nl = a.n_layers, n_in = a.dims[tr.l], n_out = a.dims[tr.l + 1];
  const int n = tr.ntile * 16 + r, m = tr.mtile * 16 + r;
  const bool a_ok = n < n_out, b_ok = m < n_in, b_one = m == n_in;
  const bool a_gy = tr.l + 1 == nl, b_x = tr.l == 0;
  // row of this lane inside a sample tile's factor block (unused for the gy / x operands)
  const size_t a_row = (size_t)(a.f_half + a.f_off[a_gy ? 1 : tr.l + 1] + n) * 16 + 4 * k;
  const size_t b_row = (size_t)(a.f_off[b_x ? 1 : tr.l] + m) * 16 + 4 * k;
  const size_t f_tile = (size_t)a.f_rows * 16;
  const int nL = a.dims[nl], n0 = a.dims[0];
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f}, ones = {1.f, 1.f, 1.f, 1.f};
  auto load_a = [&](int st) -> f32x4 {
    if (!a_ok) return zero;
    if (!a_gy) return *reinterpret_cast<const f32x4 *>(a.factors + (size_t)st * f_tile + a_row);
    f32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t s = (int64_t)st * TS + 4 * k + i;
      v[i] = s < a.N ? a.gy[s * nL + n] : 0.f;
    }
    return v;
  };
  auto load_b = [&](int st) -> f32x4 {
    if (b_one) return ones;                                   // the bias column (G is zero for samples beyond N)
    if (!b_ok) return zero;
    if (!b_x) return *reinterpret_cast<const f32x4 *>(a.factors + (size_t)st * f_tile + b_row);
    f32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t s = (int64_t)st * TS + 4 * k + i;
      v[i] = s < a.N ? a.x[s * n0 + m] : 0.f;
    }
    return v;
  };
  const int step = ks * WGW;
  int st = q + ks * wave;
  for (; st + (WG_UNROLL - 1) * step < n_stiles; st += WG_UNROLL * step) {   // WG_UNROLL sample tiles in flight
    f32x4 av[WG_UNROLL], bv[WG_UNROLL];
#pragma unroll
    for (int u = 0; u < WG_UNROLL; ++u) av[u] = load_a(st + u * step), bv[u] = load_b(st + u * step);
#pragma unroll
    for (int u = 0; u < WG_UNROLL; ++u) {
      const float a4[4] = {av[u][0], av[u][1], av[u][2], av[u][3]}, b4[4] = {bv[u][0], bv[u][1], bv[u][2], bv[u][3]};
      acc = mfma_k16<P>(a4, b4, acc);
    }
  }
  for (; st < n_stiles; st += step) {
    const f32x4 av = load_a(st), bv = load_b(st);
    const float a4[4] = {av[0], av[1], av[2], av[3]}, b4[4] = {bv[0], bv[1], bv[2], bv[3]};
    acc = mfma_k16<P>(a4, b4, acc);
  }
  // (after this wave's loads and MFMAs are in the pipes: the fp64 arithmetic of one thread hides behind them)
  if (coefs_out && blockIdx.x == 0 && threadIdx.x == 64 * WGW - 1) *coefs_out = p2c_optim::coefs(o, *o.step + 1.f);
  red[wave][lane] = acc;
  __syncthreads();
  if (wave == 0) {
    f32x4 s = red[0][lane];
#pragma unroll
    for (int w = 1; w < WGW; ++w) s += red[w][lane];
    reinterpret_cast<f32x4 *>(a.partials)[((size_t)q * a.n_tiles_w + t) * 64 + lane] = s;
  }
}

// grad = sum over workgroups of their partial tiles, in a fixed order (bitwise reproducible), scattered to the per-layer
// gradient tensors. One workgroup per dW tile: 64 lanes x 16 groups; group q adds workgroups q, q+16, ...
constexpr int RG = 16;    // groups of workgroup partials added in parallel
constexpr int RL = 16;    // lanes of a tile per reducing workgroup: 4 workgroups per tile -> every CU pulls partials
// ADAM: the optimizer step rides on the reduction (single-GPU training: no all-reduce sits between the two) -- the thread
// that holds a finished gradient applies AdamW to its parameter in the flat buffers and refreshes the packed weight image;
// the last workgroup to finish publishes the new step count. Same formula as p2c_optim::adamw_kernel (p2c_adam_math.h).
template <bool ADAM>
__global__ __launch_bounds__(RL * RG) void mlp_reduce_kernel(const MlpArgs a, int n_blocks, const p2c_adamw_desc o,
                                                          const p2c_optim::Coefs *coefs_in) {
  __shared__ f32x4 red[RG][RL];
  const int t = blockIdx.x / (64 / RL), li = threadIdx.x % RL, q = threadIdx.x / RL;
  const int lane = (blockIdx.x % (64 / RL)) * RL + li;        // lane of the MFMA C tile this thread reduces
  float step = 0.f;
  __shared__ p2c_optim::Coefs sc;
  if (ADAM) {
    step = *o.step + 1.f;                                     // read before this workgroup draws its completion ticket
    if (threadIdx.x == RL * RG - 1) sc = coefs_in ? *coefs_in : p2c_optim::coefs(o, step);   // overlaps the partial loads
  }
  // where this thread's four gradients go (threads of group 0 finish the job); with ADAM their parameter and moments are
  // requested now, so that their latency hides behind the partial-tile loads
  const TileRef tr = locate_tile(a.dims, t);
  const int n_in = a.dims[tr.l], n_out = a.dims[tr.l + 1];
  const int m = tr.mtile * 16 + (lane & 15);
  float *gp[4];
  float pv[4], mv[4], vv[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int n = tr.ntile * 16 + 4 * (lane >> 4) + r;
    gp[r] = (q == 0 && n < n_out && m <= n_in) ? ((m < n_in) ? a.gW[tr.l] + n * n_in + m : a.gb[tr.l] + n) : nullptr;
    pv[r] = mv[r] = vv[r] = 0.f;
    if (ADAM && gp[r]) {
      const ptrdiff_t off = gp[r] - o.grad;                    // same offset in every flat buffer
      pv[r] = o.param[off], mv[r] = o.exp_avg[off], vv[r] = o.exp_avg_sq[off];
    }
  }
  const size_t stride = (size_t)a.n_tiles_w * 64;
  const f32x4 *p = reinterpret_cast<const f32x4 *>(a.partials) + (size_t)t * 64 + lane;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  int w = q;
  for (; w + 7 * RG < n_blocks; w += 8 * RG) {     // eight loads in flight, added in workgroup order
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(&p[(size_t)(w + u * RG) * stride]);
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; w < n_blocks; w += RG) s += __builtin_nontemporal_load(&p[(size_t)w * stride]);
  red[q][li] = s;
  __syncthreads();
  if (q == 0) {
#pragma unroll
    for (int i = 1; i < RG; ++i) s += red[i][li];
    p2c_optim::Coefs c;
    if (ADAM) c = sc;                                         // written before the barrier above
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (gp[r]) {
        *gp[r] = (ADAM && o.zero_grad) ? 0.f : s[r];   // zero_grad: the optimizer leaves the gradient buffer zeroed
        if (ADAM) {
          const ptrdiff_t off = gp[r] - o.grad;
          if (o.adamw) p2c_optim::update<true>(c, pv[r], s[r], mv[r], vv[r]);
          else p2c_optim::update<false>(c, pv[r], s[r], mv[r], vv[r]);
          o.param[off] = pv[r], o.exp_avg[off] = mv[r], o.exp_avg_sq[off] = vv[r];
          const int n = tr.ntile * 16 + 4 * (lane >> 4) + r;
          if (a.w_image) a.w_image[a.w_off[tr.l] + n * a.ld[tr.l] + m] = pv[r];    // bias sits in column n_in == m
        }
      }
    }
  }
  if (ADAM) {
    __syncthreads();
    if (threadIdx.x == 0 && atomicAdd(o.ticket, 1) == (int)gridDim.x - 1) {
      *o.step = step;
      *o.ticket = 0;
    }
  }
}

// The same job for a handful of partials per tile (split weight gradient: NB = 8): one thread per (tile, lane) adds them in
// order -- no LDS, no barrier in front of the update.
template <bool ADAM, int NB>
__global__ __launch_bounds__(256) void mlp_reduce_small_kernel(const MlpArgs a, const p2c_adamw_desc o,
                                                               const p2c_optim::Coefs *coefs_in) {
  const int idx = blockIdx.x * 256 + threadIdx.x, t = idx >> 6, lane = idx & 63;
  const bool live = t < a.n_tiles_w;
  float step = 0.f;
  if (ADAM) step = *o.step + 1.f;
  if (live) {
    const size_t stride = (size_t)a.n_tiles_w * 64;
    const f32x4 *p = reinterpret_cast<const f32x4 *>(a.partials) + (size_t)t * 64 + lane;
    f32x4 v[NB];
#pragma unroll
    for (int w = 0; w < NB; ++w) v[w] = __builtin_nontemporal_load(&p[(size_t)w * stride]);
    const TileRef tr = locate_tile(a.dims, t);
    const int n_in = a.dims[tr.l], n_out = a.dims[tr.l + 1];
    const int m = tr.mtile * 16 + (lane & 15);
    float *gp[4];
    float pv[4], mv[4], vv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = tr.ntile * 16 + 4 * (lane >> 4) + r;
      gp[r] = (n < n_out && m <= n_in) ? ((m < n_in) ? a.gW[tr.l] + n * n_in + m : a.gb[tr.l] + n) : nullptr;
      pv[r] = mv[r] = vv[r] = 0.f;
      if (ADAM && gp[r]) {
        const ptrdiff_t off = gp[r] - o.grad;
        pv[r] = o.param[off], mv[r] = o.exp_avg[off], vv[r] = o.exp_avg_sq[off];
      }
    }
    p2c_optim::Coefs c;
    if (ADAM) c = *coefs_in;
    f32x4 s = v[0];
#pragma unroll
    for (int w = 1; w < NB; ++w) s += v[w];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (!gp[r]) continue;
      *gp[r] = (ADAM && o.zero_grad) ? 0.f : s[r];
      if (ADAM) {
        const ptrdiff_t off = gp[r] - o.grad;
        if (o.adamw) p2c_optim::update<true>(c, pv[r], s[r], mv[r], vv[r]);
        else p2c_optim::update<false>(c, pv[r], s[r], mv[r], vv[r]);
        o.param[off] = pv[r], o.exp_avg[off] = mv[r], o.exp_avg_sq[off] = vv[r];
        const int n = tr.ntile * 16 + 4 * (lane >> 4) + r;
        if (a.w_image) a.w_image[a.w_off[tr.l] + n * a.ld[tr.l] + m] = pv[r];
      }
    }
  }
  if (ADAM) {
    __syncthreads();
    if (threadIdx.x == 0 && atomicAdd(o.ticket, 1) == (int)gridDim.x - 1) {
      *o.step = step;
      *o.ticket = 0;
    }
  }
}

}  // namespace p2c_mlp

using namespace p2c_mlp;

static inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

static int fill(MlpArgs &a, const p2c_mlp_desc *d) {
  if (!d || !d->x) return P2C_E_NULL;
  if (d->n_layers < 1 || d->n_layers > P2C_MLP_MAX_LAYERS || d->N < 0) return P2C_E_SHAPE;
  a = MlpArgs{};
  a.n_layers = d->n_layers;
  a.N = d->N;
  a.x = d->x, a.y = d->y, a.gy = d->gy, a.partials = d->partials, a.w_image = d->w_image, a.saved = d->saved;
  int rows = 0, tiles = 0, params = 0, wtot = 0;
  for (int l = 0; l <= d->n_layers; ++l) {
    if (d->dims[l] < 1 || d->dims[l] > MAXW - 1) return P2C_E_SHAPE;
    a.dims[l] = d->dims[l];
    a.h_off[l] = rows;
    rows += act_rows_of(a.dims[l]);            // room for the ones row / the 4-step k rounding past pad16
  }
  for (int l = 0; l < d->n_layers; ++l) {
    if (!d->W[l] || !d->b[l]) return P2C_E_NULL;
    a.W[l] = d->W[l], a.b[l] = d->b[l], a.gW[l] = d->gW[l], a.gb[l] = d->gb[l];
    tiles += ((a.dims[l + 1] + 15) / 16) * ((a.dims[l] + 1 + 15) / 16);
    params += a.dims[l + 1] * (a.dims[l] + 1);
    a.ld[l] = ld_of(a.dims[l]);
    a.w_off[l] = wtot;
    // rows: forward tiles cover 0..n_out (unit row), dgrad k-steps cover up to pad16(n_out)
    wtot += img_rows_of(a.dims[l + 1]) * a.ld[l];
  }
  a.act_rows = rows;
  for (int l = 1, r = 0; l <= d->n_layers; ++l) {
    a.f_off[l] = r;
    if (l < d->n_layers) r += a.dims[l];
    a.f_half = r;
  }
  a.f_rows = 2 * a.f_half;
  for (int l = 0; l < d->n_layers; ++l) {            // wave-per-tile forward: H_l lives in buffer l & 1
    int32_t &r = (l & 1) ? a.fw_rows_b : a.fw_rows_a;
    if (act_rows_of(a.dims[l]) > r) r = act_rows_of(a.dims[l]);
  }
  a.n_tiles_w = tiles;
  a.n_params = params;
  a.w_total = (wtot + 3) & ~3;
  if (tiles <= MAX_SLOTS * WAVES) {   // slots past the last tile alias tile 0 (computed, never written out)
    const int nl = d->n_layers;
    int t = 0;
    for (int l = 0; l < nl; ++l) {
      const int ntl = (a.dims[l + 1] + 15) / 16, mtl = (a.dims[l] + 1 + 15) / 16;
      for (int nt = 0; nt < ntl; ++nt)
        for (int mt = 0; mt < mtl; ++mt, ++t) {
          a.tab[2 * t] = (a.h_off[nl] - a.h_off[1] + a.h_off[l + 1] + nt * 16) * TP;   // G rows live behind the H area
          a.tab[2 * t + 1] = (a.h_off[l] + mt * 16) * TP;
        }
    }
    for (; t < MAX_SLOTS * WAVES; ++t) a.tab[2 * t] = a.tab[0], a.tab[2 * t + 1] = a.tab[1];
  }
  const int n0 = a.dims[0], nL = a.dims[d->n_layers];
  a.vec_x = (n0 % 4 == 0) && aligned16(a.x);
  a.vec_y = (nL % 4 == 0) && aligned16(a.y);
  a.vec_gy = (nL % 4 == 0) && aligned16(a.gy);
  return 0;
}

static size_t lds_fwd(const MlpArgs &a) { return ((size_t)a.w_total + (size_t)a.h_off[a.n_layers] * TP) * sizeof(float); }
static size_t lds_bwd(const MlpArgs &a) {
  return ((size_t)a.w_total + (size_t)(a.h_off[a.n_layers] + a.act_rows - a.h_off[1]) * TP) * sizeof(float);
}

// Kernel selection: the LinearAE shapes of the reference get the instantiations with compile-time geometry; every other
// MLP (and P2C_MLP_GENERIC=1, for tests) the generic ones. Same algorithm, same arithmetic order, same results.
typedef void (*mlp_kernel_t)(const MlpArgs);
static bool force_generic() {
  static int v = -1;
  if (v < 0) {
    const char *e = getenv("P2C_MLP_GENERIC");
    v = (e && atoi(e)) ? 1 : 0;
  }
  return v == 1;
}
template <class S>
static void allow_big_lds_for() {
  (void)hipFuncSetAttribute((const void *)mlp_bwd_kernel<S, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  (void)hipFuncSetAttribute((const void *)mlp_bwd_kernel<S, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  (void)hipFuncSetAttribute((const void *)mlp_bwd_kernel<S, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  (void)hipFuncSetAttribute((const void *)mlp_fwd_kernel<S>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}
static void allow_big_lds() {
  static bool done = false;
  if (done) return;
  allow_big_lds_for<DynShape>();
  allow_big_lds_for<LinearAE156>();
  allow_big_lds_for<LinearAE78>();
  allow_big_lds_for<LinearAE52>();
  done = true;
}
static int max_blocks();
static size_t lds_fwd_wave(const MlpArgs &a) {
  return ((size_t)a.w_total + (size_t)FW_WAVES * (a.fw_rows_a + a.fw_rows_b) * TP) * sizeof(float);
}
// Forward strategy: wave-per-tile once every wave of the grid has a tile of its own, i.e. four sample tiles per CU (measured,
// forward at N = 8 192 / 16 384 / 32 768 / 65 536 / 131 072 frames: cooperative 13.8 / 21.0 / 34.4 / 58.8 / 108 us,
// wave-per-tile 17.2 / 19.2 / 27.7 / 42.8 / 75 us). P2C_MLP_FWD=wave|coop overrides.
static bool wave_forward(const MlpArgs &a) {
  static int mode = -1;
  if (mode < 0) {
    const char *e = getenv("P2C_MLP_FWD");
    mode = !e ? 0 : (e[0] == 'w' ? 1 : 2);
  }
  if (a.N < 1 || lds_fwd_wave(a) > 160 * 1024 || mode == 2) return false;
  if (mode == 1) return true;
  return (a.N + TS - 1) / TS >= (int64_t)FW_WAVES * max_blocks();
}
template <class S, int P = P2C_PREC_F32>
static mlp_kernel_t pick_wave_of() {
  static bool done = false;
  if (!done) {
    (void)hipFuncSetAttribute((const void *)mlp_fwd_wave_kernel<S, P>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    done = true;
  }
  return mlp_fwd_wave_kernel<S, P>;
}
// reduced-precision arms: LinearAE with the 6-D rotation output only (BASELINE.json configs[1]); nullptr = not available
static bool reduced_ok(const MlpArgs &a) { return !force_generic() && LinearAE156::matches(a); }
static mlp_kernel_t pick_wave(const MlpArgs &a, int prec) {
  if (prec == P2C_PREC_BF16) return reduced_ok(a) ? pick_wave_of<LinearAE156, P2C_PREC_BF16>() : nullptr;
  if (prec == P2C_PREC_BF16X3) return reduced_ok(a) ? pick_wave_of<LinearAE156, P2C_PREC_BF16X3>() : nullptr;
  if (!force_generic()) {
    if (LinearAE156::matches(a)) return pick_wave_of<LinearAE156>();
    if (LinearAE78::matches(a)) return pick_wave_of<LinearAE78>();
    if (LinearAE52::matches(a)) return pick_wave_of<LinearAE52>();
  }
  return pick_wave_of<DynShape>();
}
template <class S, int P = P2C_PREC_F32>
static mlp_kernel_t pick_of(bool bwd, bool factors, bool saved) {
  if constexpr (P != P2C_PREC_F32) {
    static bool done = false;
    if (!done) {
      (void)hipFuncSetAttribute((const void *)mlp_bwd_kernel<S, false, false, P>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)hipFuncSetAttribute((const void *)mlp_bwd_kernel<S, true, false, P>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)hipFuncSetAttribute((const void *)mlp_bwd_kernel<S, false, true, P>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)hipFuncSetAttribute((const void *)mlp_fwd_kernel<S, P>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      done = true;
    }
  }
  if (!bwd) return mlp_fwd_kernel<S, P>;
  if (factors) return mlp_bwd_kernel<S, true, false, P>;
  return saved ? mlp_bwd_kernel<S, false, true, P> : mlp_bwd_kernel<S, false, false, P>;
}
static mlp_kernel_t pick(const MlpArgs &a, int prec, bool bwd, bool factors = false, bool saved = false) {
  if (prec == P2C_PREC_BF16) return reduced_ok(a) ? pick_of<LinearAE156, P2C_PREC_BF16>(bwd, factors, saved) : nullptr;
  if (prec == P2C_PREC_BF16X3) return reduced_ok(a) ? pick_of<LinearAE156, P2C_PREC_BF16X3>(bwd, factors, saved) : nullptr;
  if (!force_generic()) {
    if (LinearAE156::matches(a)) return pick_of<LinearAE156>(bwd, factors, saved);
    if (LinearAE78::matches(a)) return pick_of<LinearAE78>(bwd, factors, saved);
    if (LinearAE52::matches(a)) return pick_of<LinearAE52>(bwd, factors, saved);
  }
  return pick_of<DynShape>(bwd, factors, saved);
}

static int max_blocks() {
  static int v = -1;
  if (v < 0) {
    const char *e = getenv("P2C_MLP_MAX_BLOCKS");
    v = e ? atoi(e) : 256;
  }
  return v;
}
static inline int n_blocks(int64_t N) {
  int64_t n_tiles = (N + TS - 1) / TS;
  const int cap = max_blocks();
  return (int)(n_tiles < cap ? (n_tiles < 1 ? 1 : n_tiles) : cap);   // persistent: one workgroup per CU
}

// Weight-gradient strategy. The fused phase (accumulators in registers across the persistent tile loop, one partial per
// workgroup) costs 2 x 80 KB of HBM traffic per WORKGROUP; the split path costs 20 KB of factors per SAMPLE TILE plus one
// more launch. Split pays where the fused path's partials peak relative to its compute: about one sample tile per CU
// (measured, train step at B = 128 / 256 / 512 / 768 clips of 16 frames: fused 47.0 / 51.8 / 67.2 / 82.3 us, split
// 47.1 / 49.8 / 71.4 / 93.0 us). P2C_MLP_WGRAD=fused|split overrides (tests, measurements).
constexpr int WGRAD_KS = 8;      // = XCDs: see mlp_wgrad_kernel
static bool split_wgrad(int64_t N) {
  static int mode = -1;
  if (mode < 0) {
    const char *e = getenv("P2C_MLP_WGRAD");
    mode = !e ? 0 : (e[0] == 's' ? 1 : (e[0] == 'f' ? 2 : 0));
  }
  const int64_t n_stiles = (N + TS - 1) / TS;
  if (n_stiles < 1) return false;
  if (mode == 1) return true;
  if (mode == 2) return false;
  const int64_t cus = max_blocks();
  return 4 * n_stiles > 3 * cus && 2 * n_stiles <= 3 * cus;      // (0.75, 1.5] sample tiles per workgroup slot
}
static inline int64_t split_floats(const MlpArgs &a) {
  const int64_t n_stiles = (a.N + TS - 1) / TS;
  return (int64_t)WGRAD_KS * a.n_tiles_w * 256 + n_stiles * a.f_rows * 16 + 16;     // + this step's optimizer coefficients
}

#ifdef P2C_MLP_TRACE
extern "C" P2C_API int p2c_debug_mlp_trace(unsigned long long *out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(p2c_mlp::g_trace), sizeof(unsigned long long) * 80);
}
#endif

extern "C" int64_t p2c_mlp_image_floats(const p2c_mlp_desc *d) {
  MlpArgs a;
  if (fill(a, d)) return 0;
  return a.w_total;
}

extern "C" int p2c_mlp_pack(const p2c_mlp_desc *d, void *stream_) {
  MlpArgs a;
  int rc = fill(a, d);
  if (rc) return rc;
  if (!a.w_image) return P2C_E_NULL;
  hipLaunchKernelGGL(mlp_pack_kernel, dim3((a.w_total + 255) / 256), dim3(256), 0, (hipStream_t)stream_, a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

extern "C" int64_t p2c_mlp_image_index(const p2c_mlp_desc *d, int32_t *index, int64_t capacity) {
  if (!d || !index) return P2C_E_NULL;
  if (d->n_layers < 1 || d->n_layers > P2C_MLP_MAX_LAYERS) return P2C_E_SHAPE;
  int64_t n = 0;
  int off = 0;
  for (int l = 0; l < d->n_layers; ++l) {
    const int n_in = d->dims[l], n_out = d->dims[l + 1];
    if (n_in < 1 || n_out < 1 || n_in > MAXW - 1 || n_out > MAXW - 1) return P2C_E_SHAPE;
    const int ld = ld_of(n_in);
    if (n + (int64_t)n_out * (n_in + 1) > capacity) return P2C_E_SHAPE;
    for (int r = 0; r < n_out; ++r)
      for (int k = 0; k < n_in; ++k) index[n++] = off + r * ld + k;
    for (int r = 0; r < n_out; ++r) index[n++] = off + r * ld + n_in;
    off += img_rows_of(n_out) * ld;
  }
  return n;
}

// Saved activations pay once a workgroup walks two or more sample tiles: the recomputation is ~1/3 of a tile's backward,
// the 10 KB per tile of extra traffic each way hides behind the persistent loop's look-ahead. (With one tile per workgroup
// the recomputation is what hides the weight-image staging, and the split weight gradient leaves its own factors.)
// P2C_MLP_SAVE=0|1 overrides.
static bool save_activations(const MlpArgs &a) {
  static int mode = -1;
  if (mode < 0) {
    const char *e = getenv("P2C_MLP_SAVE");
    mode = !e ? 0 : (atoi(e) ? 1 : 2);
  }
  if (a.n_layers < 2 || a.N < 1 || split_wgrad(a.N) || mode == 2) return false;
  if (mode == 1) return true;
  return (a.N + TS - 1) / TS >= 2 * (int64_t)n_blocks(a.N);
}
extern "C" int64_t p2c_mlp_saved_floats(const p2c_mlp_desc *d) {
  MlpArgs a;
  if (fill(a, d) || !save_activations(a)) return 0;
  return ((a.N + TS - 1) / TS) * (int64_t)a.f_half * 16;
}

extern "C" int64_t p2c_mlp_workspace_floats(const p2c_mlp_desc *d) {
  MlpArgs a;
  if (fill(a, d)) return 0;
  const int64_t fused = (int64_t)n_blocks(a.N) * a.n_tiles_w * 256;
  return split_wgrad(a.N) && split_floats(a) > fused ? split_floats(a) : fused;
}

extern "C" int p2c_mlp_fwd(const p2c_mlp_desc *d, void *stream_) {
  MlpArgs a;
  int rc = fill(a, d);
  if (rc) return rc;
  if (!a.y || !a.w_image) return P2C_E_NULL;
  if (a.N == 0) return 0;
  if (a.saved && !save_activations(a)) a.saved = nullptr;      // same rule on both sides of the autograd edge
  const size_t lds = lds_fwd(a);
  if (lds > 160 * 1024) return P2C_E_SHAPE;
  allow_big_lds();
  if (!d->skip_pack)
    hipLaunchKernelGGL(mlp_pack_kernel, dim3((a.w_total + 255) / 256), dim3(256), 0, (hipStream_t)stream_, a);
  const int prec = d->precision;
  if (prec < P2C_PREC_F32 || prec > P2C_PREC_BF16X3 || (prec != P2C_PREC_F32 && !reduced_ok(a))) return P2C_E_ENUM;
  if (wave_forward(a)) {
    const int64_t groups = ((a.N + TS - 1) / TS + FW_WAVES - 1) / FW_WAVES;
    const int cap = max_blocks();
    hipLaunchKernelGGL(pick_wave(a, prec), dim3((unsigned)(groups < cap ? groups : cap)), dim3(64 * FW_WAVES), lds_fwd_wave(a),
                       (hipStream_t)stream_, a);
  } else {
    hipLaunchKernelGGL(pick(a, prec, false), dim3(n_blocks(a.N)), dim3(64 * WAVES), lds, (hipStream_t)stream_, a);
  }
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

extern "C" int p2c_mlp_bwd(const p2c_mlp_desc *d, void *stream_) {
  MlpArgs a;
  int rc = fill(a, d);
  if (rc) return rc;
  if (!a.gy || !a.partials || !a.w_image) return P2C_E_NULL;
  for (int l = 0; l < a.n_layers; ++l)
    if (!a.gW[l] || !a.gb[l]) return P2C_E_NULL;
  if (a.n_tiles_w > MAX_SLOTS * WAVES) return P2C_E_SHAPE;
  const int prec = d->precision;
  if (prec < P2C_PREC_F32 || prec > P2C_PREC_BF16X3 || (prec != P2C_PREC_F32 && !reduced_ok(a))) return P2C_E_ENUM;
  const size_t lds = lds_bwd(a);
  if (lds > 160 * 1024) return P2C_E_SHAPE;
  int blocks = n_blocks(a.N);
  p2c_optim::Coefs *coefs = nullptr;
  allow_big_lds();
  if (d->fused_adamw) {                                         // everything about the optimizer BEFORE the first launch
    const p2c_adamw_desc &o = *d->fused_adamw;
    if (!o.param || !o.grad || !o.exp_avg || !o.exp_avg_sq || !o.step || !o.ticket || !o.hyper) return P2C_E_NULL;
    for (int l = 0; l < a.n_layers; ++l) {                      // every gradient tensor must be a view of the optimizer's grad
      const float *lo = o.grad, *hi = o.grad + o.n;
      if (a.gW[l] < lo || a.gW[l] + (size_t)a.dims[l + 1] * a.dims[l] > hi || a.gb[l] < lo || a.gb[l] + a.dims[l + 1] > hi)
        return P2C_E_INDEX;
    }
    if ((int64_t)a.n_params != o.n) return P2C_E_SHAPE;         // the MLP must be ALL the optimizer optimises (step counter)
  }
  if (split_wgrad(a.N)) {
    const int n_stiles = (int)((a.N + TS - 1) / TS);
    a.factors = a.partials + (size_t)WGRAD_KS * a.n_tiles_w * 256;      // [KS partial tiles | factors]
    hipLaunchKernelGGL(pick(a, prec, true, true), dim3(blocks), dim3(64 * WAVES), lds, (hipStream_t)stream_, a);
    coefs = reinterpret_cast<p2c_optim::Coefs *>(a.factors + (size_t)n_stiles * a.f_rows * 16);
    const auto wgrad = prec == P2C_PREC_BF16 ? mlp_wgrad_kernel<P2C_PREC_BF16>
                       : (prec == P2C_PREC_BF16X3 ? mlp_wgrad_kernel<P2C_PREC_BF16X3> : mlp_wgrad_kernel<P2C_PREC_F32>);
    hipLaunchKernelGGL(wgrad, dim3(a.n_tiles_w * WGRAD_KS), dim3(64 * WGW), 0, (hipStream_t)stream_, a,
                       n_stiles, WGRAD_KS, d->fused_adamw ? *d->fused_adamw : p2c_adamw_desc{},
                       d->fused_adamw ? coefs : nullptr);
    blocks = WGRAD_KS;                                                   // what the reduction adds up
  } else {
    const bool saved = a.saved && save_activations(a);
    hipLaunchKernelGGL(pick(a, prec, true, false, saved), dim3(blocks), dim3(64 * WAVES), lds, (hipStream_t)stream_, a);
  }
  const dim3 rgrid(a.n_tiles_w * (64 / RL)), rblock(RL * RG), sgrid((a.n_tiles_w * 64 + 255) / 256);
  if (d->fused_adamw) {
    const p2c_adamw_desc o = *d->fused_adamw;                 // (validated above, before the first launch)
    if (coefs)
      hipLaunchKernelGGL((mlp_reduce_small_kernel<true, WGRAD_KS>), sgrid, dim3(256), 0, (hipStream_t)stream_, a, o, coefs);
    else
      hipLaunchKernelGGL(mlp_reduce_kernel<true>, rgrid, rblock, 0, (hipStream_t)stream_, a, blocks, o, coefs);
  } else {
    if (blocks == WGRAD_KS && split_wgrad(a.N))
      hipLaunchKernelGGL((mlp_reduce_small_kernel<false, WGRAD_KS>), sgrid, dim3(256), 0, (hipStream_t)stream_, a,
                         p2c_adamw_desc{}, nullptr);
    else
      hipLaunchKernelGGL(mlp_reduce_kernel<false>, rgrid, rblock, 0, (hipStream_t)stream_, a, blocks, p2c_adamw_desc{}, nullptr);
  }
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}
