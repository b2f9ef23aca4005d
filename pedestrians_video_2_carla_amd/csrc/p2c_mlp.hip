// p2c_mlp.hip -- fused small-MLP (LinearAE) forward / backward for gfx950 on fp32 MFMA (v_mfma_f32_16x16x4_f32).
//
// The reference's LinearAE (modules/movements/linear_ae/linear_ae.py:25-59) is a per-frame MLP
// 52 -> 26 -> 13 -> 6 -> O/4 -> O/2 -> O with ReLU between (17 530 parameters for O = 156). As ATen ops that is ~50
// launches per train step (6 GEMMs with K <= 78, bias/ReLU/bias-grad kernels), each a few microseconds of pure launch
// latency: at the benchmark's B = 256 they are 85 % of the step. Here the whole stack is ONE launch forward and ONE
// backward (+ a small deterministic reduction of the per-workgroup weight-gradient partials).
//
// Structure ("cooperative 16-sample tile"):
//   * a workgroup of four wavefronts walks over tiles of 16 frames ("samples"). Activations live TRANSPOSED in LDS,
//     H^T[n][sample]: the sample index sits on lane&15 for the MFMA B operand (B[k][col]: lane = col + 16*k) *and* for
//     the C/D tile (col = lane&15, row = 4*(lane>>4)+reg), so layer l+1 reads what layer l wrote with plain
//     ds_read_b32 -- no transpose anywhere. The 16-row output tiles of a layer are dealt round-robin to the four waves
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
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/p2c.h"

namespace p2c_mlp {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int MAXW = 160;        // widest layer (padded to 16)
constexpr int TS = 16;           // samples per tile
constexpr int WAVES = 4;         // waves per workgroup
constexpr int MAX_SLOTS = 24;    // dW tiles per wave held in accumulators
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
  int32_t act_rows;              // rows of the H area (the G area has the same layout)
  float *w_image;                // packed, zero-padded weight images in HBM (w_total floats), written by mlp_pack_kernel
};

__device__ __forceinline__ int pad16(int n) { return (n + 15) & ~15; }

// Pack kernel (once per forward): the zero-padded image of every [W_l | b_l] -- rows 0..rows_l-1, pitch ld_l, bias in
// column n_in -- laid out exactly as the workgroups want it in LDS.
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
  }
  a.w_image[i] = v;
}

// Workgroup copy of the packed image HBM -> LDS: 16-byte loads, eight in flight per thread; every workgroup starts at
// a different offset so that the 256 CUs do not ask the L2 for the same line at the same moment.
__device__ __forceinline__ void stage_image(const MlpArgs &a, float *dst) {
  const int total4 = a.w_total >> 2, nth = blockDim.x;
  const f32x4 *src = reinterpret_cast<const f32x4 *>(a.w_image);
  f32x4 *d4 = reinterpret_cast<f32x4 *>(dst);
  const int rot = (int)((blockIdx.x * 2654435761u) % (unsigned)total4);
  constexpr int U = 8;
  for (int i0 = threadIdx.x; i0 < total4; i0 += U * nth) {
    f32x4 v[U];
    int idx[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int i = i0 + u * nth + rot;
      i -= (i >= total4) ? total4 : 0;
      idx[u] = (i0 + u * nth < total4) ? i : 0;
      v[u] = src[idx[u]];
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (i0 + u * nth < total4) d4[idx[u]] = v[u];
  }
}

struct Lane {
  int lane, c, g, wave;   // c = lane & 15 (sample / column), g = lane >> 4
};

// x tile (16 samples x n0 features, contiguous rows in HBM) -> H_0^T[k][sample] with the ones row and zero padding
__device__ __forceinline__ void load_x_tile(const MlpArgs &a, int64_t row0, float *h0) {
  const int n0 = a.dims[0], rows = pad16(n0) + 16;
  for (int i = threadIdx.x; i < rows * TS; i += blockDim.x) {
    const int k = i >> 4, s = i & 15;      // consecutive threads -> consecutive samples of one feature (conflict-free)
    float v = 0.f;
    if (k < n0) {
      const int64_t r = row0 + s;
      v = (r < a.N) ? a.x[r * n0 + k] : 0.f;
    } else if (k == n0) {
      v = 1.f;
    }
    h0[i] = v;
  }
}

// out^T[n][s] = act( sum_k Waug[n][k] in^T_aug[k][s] ) for the output tiles owned by this wave (nt = wave, wave+4, ..)
// in: LDS rows [k][16] incl. ones row; out: LDS rows (ones row n_out written as 1, padding rows as 0) and/or HBM rows y.
__device__ __forceinline__ void layer_forward(const Lane &L, const float *wl, int ld, int n_in, int n_out, bool relu,
                                              const float *in, float *out, float *y_row, bool row_ok, int y_stride) {
  const int ksteps = (((n_in + 1 + 3) >> 2) + 3) & ~3;   // multiple of 4: image and activations are zero beyond n_in
  const int ntiles = (n_out + 16) >> 4;    // tiles covering rows 0..n_out (the ones row included)
  for (int nt = L.wave; nt < ntiles; nt += 3 * WAVES) {
    // up to three tiles of this wave per pass: independent accumulators hide the 40-cycle MFMA latency
    const int nt1 = nt + WAVES, nt2 = nt + 2 * WAVES;
    const bool two = nt1 < ntiles, three = nt2 < ntiles;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
    const float *a0p = wl + (nt * 16 + L.c) * ld + L.g;
    const float *a1p = wl + ((two ? nt1 : nt) * 16 + L.c) * ld + L.g;
    const float *a2p = wl + ((three ? nt2 : nt) * 16 + L.c) * ld + L.g;
    const float *bp = in + L.g * TS + L.c;
    for (int s = 0; s < ksteps; s += 4) {
      float bv[4], a0[4], a1[4], a2[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        bv[u] = bp[(s + u) * 4 * TS];
        a0[u] = a0p[(s + u) * 4], a1[u] = a1p[(s + u) * 4], a2[u] = a2p[(s + u) * 4];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[u], bv[u], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[u], bv[u], acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[u], bv[u], acc2, 0, 0, 0);
      }
    }
#pragma unroll
    for (int h = 0; h < 3; ++h) {
      if ((h == 1 && !two) || (h == 2 && !three)) break;
      f32x4 acc = (h == 0) ? acc0 : ((h == 1) ? acc1 : acc2);
      const int nb = (nt + h * WAVES) * 16 + 4 * L.g;   // first of this lane's 4 output rows
      if (relu) {
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = fmaxf(acc[r], 0.f);
      }
      if (out) {
#pragma unroll
        for (int r = 0; r < 4; ++r) out[(nb + r) * TS + L.c] = (nb + r == n_out) ? 1.f : acc[r];
      }
      if (y_row && row_ok) {
        if (nb + 3 < n_out && (y_stride & 3) == 0) {   // 16-byte aligned only when the row pitch is a multiple of 4
          *reinterpret_cast<f32x4 *>(y_row + nb) = acc;
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (nb + r < n_out) y_row[nb + r] = acc[r];
        }
      }
    }
  }
}

// gout^T[m][s] = (H^T[m][s] > 0 && m < n_in) * sum_k W[k][m] gin^T[k][s]; the m-tiles are dealt to the waves
__device__ __forceinline__ void layer_dgrad(const Lane &L, const float *wl, int ld, int n_in, int n_out, const float *gin,
                                            const float *Hprev, float *gout) {
  const int ksteps = (((n_out + 3) >> 2) + 3) & ~3;      // multiple of 4: image rows and G rows are zero beyond n_out
  const int mtiles = (n_in + 15) >> 4;
  for (int mt = L.wave; mt < mtiles; mt += 2 * WAVES) {
    const int mt1 = mt + WAVES;
    const bool two = mt1 < mtiles;
    f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};
    const float *a0p = wl + L.g * ld + mt * 16 + L.c;
    const float *a1p = wl + L.g * ld + (two ? mt1 : mt) * 16 + L.c;
    const float *bp = gin + L.g * TS + L.c;
    for (int s = 0; s < ksteps; s += 4) {
      float bv[4], a0[4], a1[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        bv[u] = bp[(s + u) * 4 * TS];
        a0[u] = a0p[(s + u) * 4 * ld], a1[u] = a1p[(s + u) * 4 * ld];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[u], bv[u], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[u], bv[u], c1, 0, 0, 0);
      }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (h == 1 && !two) break;
      f32x4 cc = h ? c1 : c0;
      const int mb = (h ? mt1 : mt) * 16 + 4 * L.g;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float hv = Hprev[(mb + r) * TS + L.c];
        gout[(mb + r) * TS + L.c] = (mb + r < n_in && hv > 0.f) ? cc[r] : 0.f;
      }
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

// ---- forward ---------------------------------------------------------------------------------------------------------
// LDS: [weight images | H area: H_0 .. H_{L-1}]   (the input rows of layer l are H_l, its output H_{l+1})
__global__ __launch_bounds__(256) void mlp_fwd_kernel(const MlpArgs a) {
  extern __shared__ float lds[];
  Lane L;
  L.lane = threadIdx.x & 63, L.c = L.lane & 15, L.g = L.lane >> 4, L.wave = threadIdx.x >> 6;
  stage_image(a, lds);
  float *H = lds + a.w_total;
  const int64_t n_tiles = (a.N + TS - 1) / TS;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t row0 = tile * TS, row = row0 + L.c;
    const bool row_ok = row < a.N;
    __syncthreads();                                        // previous tile fully consumed (and weights staged)
    load_x_tile(a, row0, H + a.h_off[0] * TS);
    for (int l = 0; l < a.n_layers; ++l) {
      __syncthreads();
      const bool last = (l == a.n_layers - 1);
      layer_forward(L, lds + a.w_off[l], a.ld[l], a.dims[l], a.dims[l + 1], !last, H + a.h_off[l] * TS,
                    last ? nullptr : H + a.h_off[l + 1] * TS, last ? a.y + row * a.dims[l + 1] : nullptr, row_ok,
                    a.dims[l + 1]);
    }
  }
}

// ---- backward --------------------------------------------------------------------------------------------------------
// LDS: [weight images | H area (H_0 .. H_{L-1}) | G area (G_1 .. G_L, G_l at the offset of H_l; G_L = gy tile)]
__global__ __launch_bounds__(256) void mlp_bwd_kernel(const MlpArgs a) {
  extern __shared__ float lds[];
  Lane L;
  L.lane = threadIdx.x & 63, L.c = L.lane & 15, L.g = L.lane >> 4, L.wave = threadIdx.x >> 6;
  const int nl = a.n_layers;
  stage_image(a, lds);
  float *H = lds + a.w_total, *G = H + a.act_rows * TS;

  f32x4 acc[MAX_SLOTS];
#pragma unroll
  for (int i = 0; i < MAX_SLOTS; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int64_t n_tiles = (a.N + TS - 1) / TS;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t row0 = tile * TS;
    __syncthreads();
    load_x_tile(a, row0, H + a.h_off[0] * TS);
    {   // gy tile -> G_L^T[n][sample], zero padded (rows >= n_out and samples beyond N)
      const int n_out = a.dims[nl], rows = pad16(n_out) + 16;
      float *gl = G + a.h_off[nl] * TS;
      for (int i = threadIdx.x; i < rows * TS; i += blockDim.x) {
        const int k = i >> 4, s = i & 15;
        const int64_t r = row0 + s;
        gl[i] = (k < n_out && r < a.N) ? a.gy[r * n_out + k] : 0.f;
      }
    }
    // ---- phase F: activations H_1 .. H_{L-1}
    for (int l = 0; l < nl - 1; ++l) {
      __syncthreads();
      layer_forward(L, lds + a.w_off[l], a.ld[l], a.dims[l], a.dims[l + 1], true, H + a.h_off[l] * TS,
                    H + a.h_off[l + 1] * TS, nullptr, false, 0);
    }
    // ---- phase D: G_l = relu'(H_l) .* (W_l^T G_{l+1}), l = L-1 .. 1
    for (int l = nl - 1; l >= 1; --l) {
      __syncthreads();
      layer_dgrad(L, lds + a.w_off[l], a.ld[l], a.dims[l], a.dims[l + 1], G + a.h_off[l + 1] * TS, H + a.h_off[l] * TS,
                  G + a.h_off[l] * TS);
    }
    __syncthreads();
    // ---- phase W: dW_aug_l[n][m] += sum_s G_{l+1}^T[n][s] * H_l^T_aug[m][s]; tile t belongs to wave (t & 3), slot (t >> 2)
    // (rows of samples beyond N carry G = 0, so they add nothing)
#pragma unroll
    for (int slot = 0; slot < MAX_SLOTS; ++slot) {
      const int t = slot * WAVES + L.wave;
      if (t < a.n_tiles_w) {
        const TileRef tr = locate_tile(a.dims, t);
        const float *gp = G + (a.h_off[tr.l + 1] + tr.ntile * 16 + L.c) * TS + L.g;   // A[n][k = sample]
        const float *hp = H + (a.h_off[tr.l] + tr.mtile * 16 + L.c) * TS + L.g;       // B[k = sample][m]
        f32x4 c = acc[slot];
#pragma unroll
        for (int s = 0; s < TS / 4; ++s) c = __builtin_amdgcn_mfma_f32_16x16x4f32(gp[4 * s], hp[4 * s], c, 0, 0, 0);
        acc[slot] = c;
      }
    }
  }
  // ---- per-workgroup partial gradients: partials[block][param], parameter order = (W_0, b_0, W_1, b_1, ...)
  float *part = a.partials + (size_t)blockIdx.x * a.n_params;
#pragma unroll
  for (int slot = 0; slot < MAX_SLOTS; ++slot) {
    const int t = slot * WAVES + L.wave;
    if (t < a.n_tiles_w) {
      const TileRef tr = locate_tile(a.dims, t);
      const int n_in = a.dims[tr.l], n_out = a.dims[tr.l + 1];
      const int m = tr.mtile * 16 + L.c;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = tr.ntile * 16 + 4 * L.g + r;
        if (n < n_out) {
          if (m < n_in) part[tr.base + n * n_in + m] = acc[slot][r];
          else if (m == n_in) part[tr.base + n_out * n_in + n] = acc[slot][r];
        }
      }
    }
  }
}

// grad[i] = sum over workgroups of partials[w][i], fixed order; scattered to the per-layer gradient tensors
__global__ __launch_bounds__(256) void mlp_reduce_kernel(const MlpArgs a, int n_blocks) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n_params) return;
  float s8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int w = 0;
  for (; w + 8 <= n_blocks; w += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) s8[u] += a.partials[(size_t)(w + u) * a.n_params + i];
  }
  for (; w < n_blocks; ++w) s8[0] += a.partials[(size_t)w * a.n_params + i];
  float s = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
  int base = 0;
  for (int l = 0; l < a.n_layers; ++l) {
    const int nw = a.dims[l + 1] * a.dims[l], nb = a.dims[l + 1];
    if (i < base + nw) { a.gW[l][i - base] = s; return; }
    if (i < base + nw + nb) { a.gb[l][i - base - nw] = s; return; }
    base += nw + nb;
  }
}

}  // namespace p2c_mlp

using namespace p2c_mlp;

static int fill(MlpArgs &a, const p2c_mlp_desc *d) {
  if (!d || !d->x) return P2C_E_NULL;
  if (d->n_layers < 1 || d->n_layers > P2C_MLP_MAX_LAYERS || d->N < 0) return P2C_E_SHAPE;
  a = MlpArgs{};
  a.n_layers = d->n_layers;
  a.N = d->N;
  a.x = d->x, a.y = d->y, a.gy = d->gy, a.partials = d->partials, a.w_image = d->w_image;
  int rows = 0, tiles = 0, params = 0, wtot = 0;
  for (int l = 0; l <= d->n_layers; ++l) {
    if (d->dims[l] < 1 || d->dims[l] > MAXW - 1) return P2C_E_SHAPE;
    a.dims[l] = d->dims[l];
    a.h_off[l] = rows;
    rows += ((a.dims[l] + 15) & ~15) + 16;     // room for the ones row / the 4-step k rounding past pad16
  }
  for (int l = 0; l < d->n_layers; ++l) {
    if (!d->W[l] || !d->b[l]) return P2C_E_NULL;
    a.W[l] = d->W[l], a.b[l] = d->b[l], a.gW[l] = d->gW[l], a.gb[l] = d->gb[l];
    tiles += ((a.dims[l + 1] + 15) / 16) * ((a.dims[l] + 1 + 15) / 16);
    params += a.dims[l + 1] * (a.dims[l] + 1);
    int ld = ((((a.dims[l] + 1 + 3) >> 2) + 3) & ~3) * 4;      // k extent incl. the bias column, in 4-step blocks
    int p16 = (a.dims[l] + 15) & ~15;                          // dgrad reads columns up to pad16(n_in)
    if (ld < p16) ld = p16;
    while ((ld & 3) != 2) ++ld;                                // pitch == 2 (mod 4)
    a.ld[l] = ld;
    a.w_off[l] = wtot;
    // rows: forward tiles cover 0..n_out (ones row), dgrad k-steps cover up to 4*ceil(n_out/4)
    wtot += (((a.dims[l + 1] + 16) & ~15)) * ld;
  }
  a.act_rows = rows;
  a.n_tiles_w = tiles;
  a.n_params = params;
  a.w_total = (wtot + 3) & ~3;
  return 0;
}

static void allow_big_lds() {
  static bool done = false;
  if (done) return;
  (void)hipFuncSetAttribute((const void *)mlp_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  (void)hipFuncSetAttribute((const void *)mlp_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  done = true;
}

static inline int n_blocks(int64_t N) {
  int64_t n_tiles = (N + TS - 1) / TS;
  return (int)(n_tiles < 256 ? (n_tiles < 1 ? 1 : n_tiles) : 256);   // persistent: one workgroup per CU
}

extern "C" int64_t p2c_mlp_image_floats(const p2c_mlp_desc *d) {
  MlpArgs a;
  if (fill(a, d)) return 0;
  return a.w_total;
}

extern "C" int64_t p2c_mlp_workspace_floats(const p2c_mlp_desc *d) {
  MlpArgs a;
  if (fill(a, d)) return 0;
  return (int64_t)n_blocks(a.N) * a.n_params;
}

extern "C" int p2c_mlp_fwd(const p2c_mlp_desc *d, void *stream_) {
  MlpArgs a;
  int rc = fill(a, d);
  if (rc) return rc;
  if (!a.y || !a.w_image) return P2C_E_NULL;
  if (a.N == 0) return 0;
  size_t lds = ((size_t)a.w_total + (size_t)a.act_rows * TS) * sizeof(float);
  if (lds > 160 * 1024) return P2C_E_SHAPE;
  allow_big_lds();
  hipLaunchKernelGGL(mlp_pack_kernel, dim3((a.w_total + 255) / 256), dim3(256), 0, (hipStream_t)stream_, a);
  hipLaunchKernelGGL(mlp_fwd_kernel, dim3(n_blocks(a.N)), dim3(64 * WAVES), lds, (hipStream_t)stream_, a);
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
  size_t lds = ((size_t)a.w_total + 2 * (size_t)a.act_rows * TS) * sizeof(float);
  if (lds > 160 * 1024) return P2C_E_SHAPE;
  const int blocks = n_blocks(a.N);
  allow_big_lds();
  hipLaunchKernelGGL(mlp_bwd_kernel, dim3(blocks), dim3(64 * WAVES), lds, (hipStream_t)stream_, a);
  hipLaunchKernelGGL(mlp_reduce_kernel, dim3((a.n_params + 255) / 256), dim3(256), 0, (hipStream_t)stream_, a, blocks);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}
