// p2c_mlp.hip -- fused small-MLP (LinearAE) forward / backward for gfx950 on fp32 MFMA (v_mfma_f32_16x16x4_f32).
//
// The reference's LinearAE (modules/movements/linear_ae/linear_ae.py:25-59) is a per-frame MLP
// 52 -> 26 -> 13 -> 6 -> O/4 -> O/2 -> O with ReLU between (17 530 parameters for O = 156). As ATen ops that is ~50
// launches per train step (6 GEMMs with K <= 78, bias/ReLU/bias-grad kernels), each a few microseconds of pure launch
// latency: at the benchmark's B = 256 they are 85 % of the step. Here the whole stack is ONE launch forward and ONE
// backward (+ a 69-block deterministic reduction of the per-workgroup weight-gradient partials):
//   * a wavefront owns 16 frames ("samples"); activations live TRANSPOSED in LDS, H^T[n][sample], so the sample index
//     sits on lane&15 for the MFMA B operand (B[k][col] : lane = col + 16*k) *and* for the C/D tile
//     (col = lane&15, row = 4*(lane>>4)+reg): layer l+1 reads what layer l wrote with plain ds_read_b32, no transpose;
//   * weights are the A operand (A[row][k] : lane = row + 16*k), read straight from L2 (70 KB, shared by every wave);
//     the bias rides along as one extra K column against a constant-one activation row;
//   * exact fp32: the MFMA is bit-for-bit an fmaf chain in k order (cdna_hip_programming.md §3), no bf16 anywhere;
//   * backward recomputes the activations (no HBM round trip), runs the dgrad chain per wave, then the four waves of
//     a workgroup split the 16x16 tiles of dW = G^T H over their 64 samples and keep them in MFMA accumulators across
//     a persistent loop over row tiles; partials are reduced in fixed order (bitwise reproducible, no atomics).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/p2c.h"

namespace p2c_mlp {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int MAXW = 160;        // widest layer (padded to 16)
constexpr int TS = 16;           // samples per wave tile
constexpr int WAVES = 4;         // waves per workgroup
constexpr int MAX_SLOTS = 24;    // dW tiles per wave held in accumulators
constexpr int ROWS_FWD = 2 * MAXW;   // ping-pong activation buffers (rows of 16 floats) per wave, forward kernel

__device__ __forceinline__ int pad16(int n) { return (n + 15) & ~15; }

struct Lane {
  int lane, c, g;   // c = lane & 15 (sample / column), g = lane >> 4
};

// A operand of the forward product: W_aug[n][k], k < n_in weights, k == n_in bias, zero padding elsewhere
__device__ __forceinline__ float w_aug(const float *W, const float *b, int n_in, int n_out, int n, int k) {
  if (n >= n_out) return 0.f;
  if (k < n_in) return W[n * n_in + k];
  return (k == n_in) ? b[n] : 0.f;
}

// One dense layer on a 16-sample tile: out^T[n][s] = act( sum_k W_aug[n][k] * in^T_aug[k][s] ).
// in: LDS rows [k][16] (or global x when x_rows != nullptr), out: LDS rows [n][16] (post-activation).
__device__ __forceinline__ void layer_forward(const Lane &L, const float *W, const float *b, int n_in, int n_out, bool relu,
                                              const float *in_lds, const float *x_rows, bool row_ok, float *out_lds,
                                              float *y_row, int y_stride) {
  const int ksteps = (n_in + 1 + 3) >> 2;
  const int ntiles = (n_out + 15) >> 4;
  for (int nt = 0; nt < ntiles; nt += 2) {
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    const int n0 = nt * 16 + L.c, n1 = n0 + 16;
    for (int s = 0; s < ksteps; ++s) {
      const int k = 4 * s + L.g;
      float bv;
      if (k < n_in) bv = x_rows ? (row_ok ? x_rows[k] : 0.f) : in_lds[k * TS + L.c];
      else bv = (k == n_in) ? 1.f : 0.f;
      float a0 = w_aug(W, b, n_in, n_out, n0, k);
      float a1 = w_aug(W, b, n_in, n_out, n1, k);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bv, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bv, acc1, 0, 0, 0);
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      f32x4 acc = h ? acc1 : acc0;
      const int nb = (nt + h) * 16 + 4 * L.g;   // first of this lane's 4 output rows
      if (nb >= pad16(n_out)) continue;
      if (relu) {
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = fmaxf(acc[r], 0.f);
      }
      if (out_lds) {
#pragma unroll
        for (int r = 0; r < 4; ++r) out_lds[(nb + r) * TS + L.c] = acc[r];
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

struct MlpArgs {
  int32_t n_layers;
  int32_t dims[P2C_MLP_MAX_LAYERS + 1];
  const float *W[P2C_MLP_MAX_LAYERS];
  const float *b[P2C_MLP_MAX_LAYERS];
  float *gW[P2C_MLP_MAX_LAYERS];
  float *gb[P2C_MLP_MAX_LAYERS];
  const float *x;
  float *y;
  const float *gy;
  float *partials;
  int64_t N;
  int32_t n_params, n_tiles_w;          // total parameters; total 16x16 dW tiles
  int32_t lds_off[P2C_MLP_MAX_LAYERS];  // row offset of H_l^T (l = 1..L-1) inside a wave's LDS region
  int32_t lds_rows;                     // rows of one of the two (H / G) halves
};

__global__ __launch_bounds__(256) void mlp_fwd_kernel(const MlpArgs a) {
  extern __shared__ float lds[];
  Lane L;
  L.lane = threadIdx.x & 63, L.c = L.lane & 15, L.g = L.lane >> 4;
  const int wave = threadIdx.x >> 6;
  float *buf0 = lds + wave * ROWS_FWD * TS, *buf1 = buf0 + MAXW * TS;
  const int64_t n_tiles = (a.N + TS - 1) / TS;
  for (int64_t tile = (int64_t)blockIdx.x * WAVES + wave; tile < n_tiles; tile += (int64_t)gridDim.x * WAVES) {
    const int64_t row = tile * TS + L.c;
    const bool row_ok = row < a.N;
    const float *x_row = a.x + row * a.dims[0];
    float *in = nullptr, *out = buf0;
    for (int l = 0; l < a.n_layers; ++l) {
      const bool last = (l == a.n_layers - 1);
      layer_forward(L, a.W[l], a.b[l], a.dims[l], a.dims[l + 1], !last, in, l == 0 ? x_row : nullptr, row_ok,
                    last ? nullptr : out, last ? a.y + row * a.dims[l + 1] : nullptr, a.dims[l + 1]);
      in = out;
      out = (out == buf0) ? buf1 : buf0;
    }
  }
}

// position of global dW tile t: layer, n-tile (output neurons), m-tile (input neurons + bias column), parameter base
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

// ---- backward ------------------------------------------------------------------------------------------------------
// LDS per wave: H region (post-ReLU activations of layers 1..L-1) then G region (their gradients), same row offsets.
__global__ __launch_bounds__(256) void mlp_bwd_kernel(const MlpArgs a) {
  extern __shared__ float lds[];
  Lane L;
  L.lane = threadIdx.x & 63, L.c = L.lane & 15, L.g = L.lane >> 4;
  const int wave = threadIdx.x >> 6;
  const int nl = a.n_layers;
  const int region = 2 * a.lds_rows * TS;                 // floats per wave
  float *Hreg = lds + wave * region, *Greg = Hreg + a.lds_rows * TS;

  f32x4 acc[MAX_SLOTS];
#pragma unroll
  for (int i = 0; i < MAX_SLOTS; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int64_t n_super = (a.N + TS * WAVES - 1) / (TS * WAVES);
  for (int64_t st = blockIdx.x; st < n_super; st += gridDim.x) {
    const int64_t row0 = st * TS * WAVES;                 // first sample of the workgroup's 64
    const int64_t row = row0 + wave * TS + L.c;
    const bool row_ok = row < a.N;
    // ---- phase F: recompute the activations of this wave's 16 samples -> H_1 .. H_{L-1} in LDS
    {
      const float *x_row = a.x + row * a.dims[0];
      const float *in = nullptr;
      for (int l = 0; l < nl - 1; ++l) {
        float *out = Hreg + a.lds_off[l + 1] * TS;
        layer_forward(L, a.W[l], a.b[l], a.dims[l], a.dims[l + 1], true, in, l == 0 ? x_row : nullptr, row_ok, out,
                      nullptr, 0);
        in = out;
      }
    }
    // ---- phase D: G_{l-1}^T = relu'(H_{l-1}) .* (W_l^T G_l^T), l = L .. 2   (G_L = gy read from HBM)
    for (int l = nl - 1; l >= 1; --l) {
      const int n_in = a.dims[l], n_out = a.dims[l + 1];  // W_l is (n_out x n_in); result has n_in rows
      const float *W = a.W[l];
      const float *gin = (l == nl - 1) ? nullptr : Greg + a.lds_off[l + 1] * TS;
      const float *gy_row = a.gy + row * n_out;
      const float *Hprev = Hreg + a.lds_off[l] * TS;
      float *gout = Greg + a.lds_off[l] * TS;
      const int ksteps = (n_out + 3) >> 2;
      const int mtiles = (n_in + 15) >> 4;
      for (int mt = 0; mt < mtiles; mt += 2) {
        f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};
        const int m0 = mt * 16 + L.c, m1 = m0 + 16;
        for (int s = 0; s < ksteps; ++s) {
          const int k = 4 * s + L.g;
          float bv = 0.f;
          if (k < n_out) bv = gin ? gin[k * TS + L.c] : (row_ok ? gy_row[k] : 0.f);
          float a0 = (k < n_out && m0 < n_in) ? W[k * n_in + m0] : 0.f;
          float a1 = (k < n_out && m1 < n_in) ? W[k * n_in + m1] : 0.f;
          c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bv, c0, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bv, c1, 0, 0, 0);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          f32x4 cc = h ? c1 : c0;
          const int mb = (mt + h) * 16 + 4 * L.g;
          if (mb >= pad16(n_in)) continue;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float hv = Hprev[(mb + r) * TS + L.c];
            gout[(mb + r) * TS + L.c] = (hv > 0.f) ? cc[r] : 0.f;
          }
        }
      }
    }
    __syncthreads();
    // ---- phase W: dW_l[n][m] += sum over the workgroup's 64 samples of G_l^T[n][s] * H_{l-1}^T_aug[m][s]
    // tile t of the global tile list belongs to wave (t & 3), accumulator slot (t >> 2)
#pragma unroll
    for (int slot = 0; slot < MAX_SLOTS; ++slot) {
      const int t = slot * WAVES + wave;
      if (t < a.n_tiles_w) {
        const TileRef tr = locate_tile(a.dims, t);
        const int l = tr.l;
        const int n_in = a.dims[l], n_out = a.dims[l + 1];
        const int n = tr.ntile * 16 + L.c;          // A row  (output neuron)
        const int m = tr.mtile * 16 + L.c;          // B col  (input neuron, n_in = the bias column)
        const bool g_from_hbm = (l == nl - 1), h_from_hbm = (l == 0);
        f32x4 c = acc[slot];
        for (int s = 0; s < TS * WAVES / 4; ++s) {      // 16 k-steps of 4 samples
          const int sample = 4 * s + L.g;               // 0..63 inside the workgroup
          const int wsrc = sample >> 4, si = sample & 15;
          const int64_t srow = row0 + sample;
          const bool ok = srow < a.N;
          const float *Hs = lds + wsrc * region, *Gs = Hs + a.lds_rows * TS;
          float av = 0.f, bv = 0.f;
          if (ok) {
            if (n < n_out) av = g_from_hbm ? a.gy[srow * n_out + n] : Gs[(a.lds_off[l + 1] + n) * TS + si];
            if (m < n_in) bv = h_from_hbm ? a.x[srow * n_in + m] : Hs[(a.lds_off[l] + m) * TS + si];
            else if (m == n_in) bv = 1.f;
          }
          c = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, c, 0, 0, 0);
        }
        acc[slot] = c;
      }
    }
    __syncthreads();
  }
  // ---- per-workgroup partial gradients: partials[block][param], parameter order = (W_0, b_0, W_1, b_1, ...)
  float *part = a.partials + (size_t)blockIdx.x * a.n_params;
#pragma unroll
  for (int slot = 0; slot < MAX_SLOTS; ++slot) {
    const int t = slot * WAVES + wave;
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
  float s = 0.f;
  for (int w = 0; w < n_blocks; ++w) s += a.partials[(size_t)w * a.n_params + i];
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
  a.x = d->x, a.y = d->y, a.gy = d->gy, a.partials = d->partials;
  int rows = 0, tiles = 0, params = 0;
  for (int l = 0; l <= d->n_layers; ++l) {
    if (d->dims[l] < 1 || d->dims[l] > MAXW - 1) return P2C_E_SHAPE;
    a.dims[l] = d->dims[l];
  }
  for (int l = 0; l < d->n_layers; ++l) {
    if (!d->W[l] || !d->b[l]) return P2C_E_NULL;
    a.W[l] = d->W[l], a.b[l] = d->b[l], a.gW[l] = d->gW[l], a.gb[l] = d->gb[l];
    a.lds_off[l] = rows;                       // H_l for l >= 1 (slot 0 unused: H_0 = x stays in HBM)
    if (l >= 1) rows += (a.dims[l] + 15) & ~15;
    tiles += ((a.dims[l + 1] + 15) / 16) * ((a.dims[l] + 1 + 15) / 16);
    params += a.dims[l + 1] * (a.dims[l] + 1);
  }
  a.lds_rows = rows > 0 ? rows : 16;
  a.n_tiles_w = tiles;
  a.n_params = params;
  return 0;
}

static void allow_big_lds() {
  static bool done = false;
  if (done) return;
  (void)hipFuncSetAttribute((const void *)mlp_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  (void)hipFuncSetAttribute((const void *)mlp_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  done = true;
}

static inline int bwd_blocks(int64_t N) {
  int64_t n_super = (N + TS * WAVES - 1) / (TS * WAVES);
  return (int)(n_super < 256 ? (n_super < 1 ? 1 : n_super) : 256);
}

extern "C" int64_t p2c_mlp_workspace_floats(const p2c_mlp_desc *d) {
  MlpArgs a;
  if (fill(a, d)) return 0;
  return (int64_t)bwd_blocks(a.N) * a.n_params;
}

extern "C" int p2c_mlp_fwd(const p2c_mlp_desc *d, void *stream_) {
  MlpArgs a;
  int rc = fill(a, d);
  if (rc) return rc;
  if (!a.y) return P2C_E_NULL;
  if (a.N == 0) return 0;
  int64_t n_tiles = (a.N + TS - 1) / TS;
  int64_t blocks = (n_tiles + WAVES - 1) / WAVES;
  if (blocks > 1024) blocks = 1024;
  size_t lds = (size_t)WAVES * ROWS_FWD * TS * sizeof(float);
  allow_big_lds();
  hipLaunchKernelGGL(mlp_fwd_kernel, dim3((unsigned)blocks), dim3(64 * WAVES), lds, (hipStream_t)stream_, a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

extern "C" int p2c_mlp_bwd(const p2c_mlp_desc *d, void *stream_) {
  MlpArgs a;
  int rc = fill(a, d);
  if (rc) return rc;
  if (!a.gy || !a.partials) return P2C_E_NULL;
  for (int l = 0; l < a.n_layers; ++l)
    if (!a.gW[l] || !a.gb[l]) return P2C_E_NULL;
  if (a.n_tiles_w > MAX_SLOTS * WAVES) return P2C_E_SHAPE;
  size_t lds = (size_t)WAVES * 2 * a.lds_rows * TS * sizeof(float);
  if (lds > 160 * 1024) return P2C_E_SHAPE;
  const int blocks = bwd_blocks(a.N);
  allow_big_lds();
  hipLaunchKernelGGL(mlp_bwd_kernel, dim3(blocks), dim3(64 * WAVES), lds, (hipStream_t)stream_, a);
  hipLaunchKernelGGL(mlp_reduce_kernel, dim3((a.n_params + 255) / 256), dim3(256), 0, (hipStream_t)stream_, a, blocks);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}
