// p2c_aux.hip -- the stand-alone pieces of the hot path for gfx950: normaliser (K4), masked 2-D MSE (K3),
// zero-filled joint remap (K5). All HBM-bound streaming kernels; one frame of <= 32 joints per 32-lane group
// (<= 64 joints: one frame per wavefront), statistics by wave shuffles, deterministic two-stage loss reduction.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

#include "../../include/p2c.h"

namespace p2c_aux {

constexpr int MAXJ = 64;

struct NormArgs {
  const float *x;
  const float *grad_out;
  float *out;      // fwd: normalised ; bwd: grad_x
  float *shift;
  float *scale;
  int64_t N;
  int32_t Jn, C, dim, transform, n_hips, n_neck;
  int32_t hips_idx[2], neck_idx[2];
  float near_zero;
};

template <int G>
__device__ __forceinline__ float gsum(float v) {
#pragma unroll
  for (int d = G / 2; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}
template <int G>
__device__ __forceinline__ float gmin(float v) {
#pragma unroll
  for (int d = G / 2; d >= 1; d >>= 1) v = fminf(v, __shfl_xor(v, d, 64));
  return v;
}
template <int G>
__device__ __forceinline__ float gmax(float v) {
#pragma unroll
  for (int d = G / 2; d >= 1; d >>= 1) v = fmaxf(v, __shfl_xor(v, d, 64));
  return v;
}

// Normalizer.__call__ (transforms/pose/normalization/normalizer.py:20-41) with the extractors of the same package:
// hips_neck_extractor.py:6-13, bbox_extractor.py:6-18 (+ utils/tensors.py:12-26), hips_neck_bbox_fallback_extractor.py:20-40.
// G lanes per frame, lane = joint. BWD: gradient of sum(out * grad_out) wrt x.
template <int G, bool BWD>
__global__ __launch_bounds__(256) void normalize_kernel(const NormArgs a) {
  const int lane = threadIdx.x & 63;
  const int j = lane & (G - 1);
  const int base = lane & ~(G - 1);
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t n = wave * (64 / G) + (lane / G);
  const bool active = (j < a.Jn) && (n < a.N);
  const int dim = a.dim, C = a.C, tr = a.transform;

  float p[3] = {0.f, 0.f, 0.f};  // the first `dim` channels
  float extra[2] = {0.f, 0.f};   // channels dim..C-1 (at most 2 are carried: C <= dim + 2)
  const float *px = a.x + (n * a.Jn + j) * C;
  if (active) {
    for (int c = 0; c < dim; ++c) p[c] = px[c];
    for (int c = dim; c < C && c < dim + 2; ++c) extra[c - dim] = px[c];
  }
  // ---- shift / scale ----
  float s[3] = {0.f, 0.f, 0.f}, k[3] = {0.f, 0.f, 0.f};
  float scale = 1.f, hn_scale = 1.f, bb_scale = 1.f;
  bool use_bb = false, missing = false;
  float mn[2] = {0.f, 0.f}, mx[2] = {0.f, 0.f};
  if (tr != P2C_TRANSFORM_BBOX) {
    for (int c = 0; c < dim; ++c) {
      float h = __shfl(p[c], base + a.hips_idx[0], 64);
      if (a.n_hips == 2) h = 0.5f * (h + __shfl(p[c], base + a.hips_idx[1], 64));
      float q = __shfl(p[c], base + a.neck_idx[0], 64);
      if (a.n_neck == 2) q = 0.5f * (q + __shfl(p[c], base + a.neck_idx[1], 64));
      s[c] = h, k[c] = q;
    }
    float acc = 0.f;
    for (int c = 0; c < dim; ++c) acc = fmaf(k[c] - s[c], k[c] - s[c], acc);
    hn_scale = sqrtf(acc);
    scale = hn_scale;
  }
  bool need_bb = (tr == P2C_TRANSFORM_BBOX);
  if (tr == P2C_TRANSFORM_HIPS_NECK_BBOX) {
    bool mh = true, mk = true;
    for (int c = 0; c < dim; ++c) mh = mh && (s[c] < a.near_zero), mk = mk && (k[c] < a.near_zero);
    use_bb = mh || mk;
    need_bb = use_bb;
  }
  if (__any(need_bb)) {  // dim == 2 guaranteed by the host wrapper
    missing = !active || ((p[0] < a.near_zero) && (p[1] < a.near_zero));
    const float inf = __builtin_inff();
    mn[0] = gmin<G>(missing ? inf : p[0]), mn[1] = gmin<G>(missing ? inf : p[1]);
    mx[0] = gmax<G>(missing ? -inf : p[0]), mx[1] = gmax<G>(missing ? -inf : p[1]);
    float cu = 0.5f * (mn[0] + mx[0]), cv = 0.5f * (mn[1] + mx[1]);
    float dx = cu - cu, dy = fminf(mn[1], mx[1]) - cv;
    bb_scale = sqrtf(fmaf(dx, dx, dy * dy));
    if (tr == P2C_TRANSFORM_BBOX) {
      s[0] = cu, s[1] = cv, scale = bb_scale;
    } else if (use_bb) {
      scale = bb_scale * 0.5748f;
    }
  }
  float o[3];
  bool fin[3];
  for (int c = 0; c < dim; ++c) {
    float v = (p[c] - s[c]) / scale;
    fin[c] = isfinite(v);
    o[c] = fin[c] ? v : 0.f;
  }
  float e0 = isfinite(extra[0]) ? extra[0] : 0.f, e1 = isfinite(extra[1]) ? extra[1] : 0.f;
  bool keep = true;
  if (dim == 2 && C > 2) {
    keep = e0 >= a.near_zero;  // normalizer.py:35-37
    if (!keep) o[0] = 0.f, o[1] = 0.f;
  }
  if (!BWD) {
    if (active) {
      float *po = a.out + (n * a.Jn + j) * C;
      for (int c = 0; c < dim; ++c) po[c] = o[c];
      if (C > dim) po[dim] = e0;
      if (C > dim + 1) po[dim + 1] = e1;
      for (int c = dim + 2; c < C; ++c) {  // rarely: copy further channels through nan_to_zero
        float v = px[c];
        po[c] = isfinite(v) ? v : 0.f;
      }
    }
    if (j == 0 && n < a.N) {
      if (a.shift)
        for (int c = 0; c < dim; ++c) a.shift[n * dim + c] = s[c];
      if (a.scale) a.scale[n] = scale;
    }
    return;
  }
  // ---- backward ----
  float g[3] = {0.f, 0.f, 0.f};
  float gextra[2] = {0.f, 0.f};
  if (active) {
    const float *pg = a.grad_out + (n * a.Jn + j) * C;
    for (int c = 0; c < dim; ++c) g[c] = (keep && fin[c]) ? pg[c] : 0.f;
    if (C > dim) gextra[0] = isfinite(extra[0]) ? pg[dim] : 0.f;
    if (C > dim + 1) gextra[1] = isfinite(extra[1]) ? pg[dim + 1] : 0.f;
  }
  float inv = 1.f / scale;
  bool ok = isfinite(inv) && scale != 0.f;
  float gp[3], A[3] = {0.f, 0.f, 0.f};
  float Cs_l = 0.f;
  for (int c = 0; c < dim; ++c) {
    gp[c] = ok ? g[c] * inv : 0.f;
    Cs_l = fmaf(gp[c], o[c], Cs_l);
  }
  for (int c = 0; c < dim; ++c) A[c] = gsum<G>(gp[c]);
  float g_scale = -gsum<G>(Cs_l);
  float gs[3] = {-A[0], -A[1], -A[2]};
  float g_bbs = 0.f;
  if (tr == P2C_TRANSFORM_BBOX) {
    g_bbs = g_scale;
  } else if (use_bb) {
    g_bbs = g_scale * 0.5748f;
  } else {
    float r = (hn_scale > 0.f) ? g_scale / hn_scale : 0.f;
    float kn = 1.f / (float)a.n_neck;
    bool is_neck = (j == a.neck_idx[0]) || (a.n_neck == 2 && j == a.neck_idx[1]);
    for (int c = 0; c < dim; ++c) {
      float gk = r * (k[c] - s[c]);
      gs[c] -= gk;
      if (is_neck) gp[c] += gk * kn;
    }
  }
  if (tr != P2C_TRANSFORM_BBOX) {
    float hn = 1.f / (float)a.n_hips;
    bool is_hips = (j == a.hips_idx[0]) || (a.n_hips == 2 && j == a.hips_idx[1]);
    if (is_hips)
      for (int c = 0; c < dim; ++c) gp[c] += gs[c] * hn;
  }
  if (tr == P2C_TRANSFORM_BBOX || __any(use_bb)) {
    float g_mn[2] = {0.f, 0.f}, g_mx[2] = {0.f, 0.f};
    if (tr == P2C_TRANSFORM_BBOX) {
      g_mn[0] += 0.5f * gs[0], g_mx[0] += 0.5f * gs[0], g_mn[1] += 0.5f * gs[1], g_mx[1] += 0.5f * gs[1];
    }
    if (tr == P2C_TRANSFORM_BBOX || use_bb) {
      float dy = fminf(mn[1], mx[1]) - 0.5f * (mn[1] + mx[1]);
      float g_dy = (bb_scale > 0.f) ? g_bbs * dy / bb_scale : 0.f;
      g_mn[1] += 0.5f * g_dy;
      g_mx[1] -= 0.5f * g_dy;
    }
    unsigned long long grp = (G == 64) ? ~0ull : (0xffffffffull << base);
    for (int c = 0; c < 2; ++c) {
      unsigned long long b = __ballot(!missing && p[c] == mn[c]) & grp;
      if (b && lane == __ffsll((long long)b) - 1) gp[c] += g_mn[c];
      b = __ballot(!missing && p[c] == mx[c]) & grp;
      if (b && lane == __ffsll((long long)b) - 1) gp[c] += g_mx[c];
    }
  }
  if (active) {
    float *po = a.out + (n * a.Jn + j) * C;
    for (int c = 0; c < dim; ++c) po[c] = gp[c];
    if (C > dim) po[dim] = gextra[0];
    if (C > dim + 1) po[dim + 1] = gextra[1];
    for (int c = dim + 2; c < C; ++c) {
      float v = px[c];
      po[c] = isfinite(v) ? a.grad_out[(n * a.Jn + j) * C + c] : 0.f;
    }
  }
}

// ---- masked 2-D MSE (loss/base_pose_loss.py:36-66, loss/loc_2d.py:69-89, utils/tensors.py:29-40) ------------------
struct Loss2dArgs {
  const float *pred, *gt;
  int64_t N;
  int32_t Jp, Cp, Jg, Cg, K, hips_col, mask;
  int32_t pidx[MAXJ], gidx[MAXJ];
};

__global__ __launch_bounds__(256) void loss2d_fwd_kernel(const Loss2dArgs a, float *partials) {
  const int64_t total = a.N * a.K;
  float s = 0.f, c = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t n = i / a.K;
    int k = (int)(i - n * a.K);
    const float *p = a.pred + (n * a.Jp + a.pidx[k]) * a.Cp;
    const float *g = a.gt + (n * a.Jg + a.gidx[k]) * a.Cg;
    float g0 = g[0], g1 = g[1];
    bool m = !a.mask || (k == a.hips_col) || (g0 != 0.f && g1 != 0.f);
    if (m) {
      float e0 = p[0] - g0, e1 = p[1] - g1;
      s += fmaf(e0, e0, e1 * e1);
      c += 1.f;
    }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64), c += __shfl_xor(c, d, 64);
  if ((threadIdx.x & 63) == 0) {
    int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    partials[wave * 2 + 0] = s, partials[wave * 2 + 1] = c;
  }
}

__global__ __launch_bounds__(256) void loss2d_finalize(const float *partials, int n_waves, float *loss_sums, float *loss) {
  __shared__ double sh[2][256];
  double a = 0.0, b = 0.0;
  for (int i = threadIdx.x; i < n_waves; i += 256) a += (double)partials[i * 2], b += (double)partials[i * 2 + 1];
  sh[0][threadIdx.x] = a, sh[1][threadIdx.x] = b;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) sh[0][threadIdx.x] += sh[0][threadIdx.x + s], sh[1][threadIdx.x] += sh[1][threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    loss_sums[0] = (float)sh[0][0], loss_sums[1] = (float)sh[1][0];
    loss[0] = (float)(sh[0][0] / (2.0 * sh[1][0]));
  }
}

// grad_pred is zero-filled by the same launch: one thread per (frame, pred joint)
__global__ __launch_bounds__(256) void loss2d_bwd_kernel(const Loss2dArgs a, const float *loss_sums, const float *grad_loss,
                                                         float *grad_pred) {
  const int64_t total = a.N * a.Jp;
  const float n2 = loss_sums[1];
  const float coef = (n2 > 0.f) ? grad_loss[0] / n2 : 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t n = i / a.Jp;
    int jp = (int)(i - n * a.Jp);
    int k = -1;  // position of this predicted joint in the common-joint list
    for (int q = 0; q < a.K; ++q)
      if (a.pidx[q] == jp) k = q;
    float g0v = 0.f, g1v = 0.f;
    float *gp = grad_pred + i * a.Cp;
    if (k >= 0) {
      const float *p = a.pred + i * a.Cp;
      const float *g = a.gt + (n * a.Jg + a.gidx[k]) * a.Cg;
      float g0 = g[0], g1 = g[1];
      bool m = !a.mask || (k == a.hips_col) || (g0 != 0.f && g1 != 0.f);
      if (m) g0v = coef * (p[0] - g0), g1v = coef * (p[1] - g1);
    }
    gp[0] = g0v, gp[1] = g1v;
    for (int c = 2; c < a.Cp; ++c) gp[c] = 0.f;
  }
}

// ---- zero-filled joint remap (data/base/base_dataset.py:156-167) ------------------------------------------------------
struct RemapArgs {
  const float *src;
  float *dst;
  int64_t N;
  int32_t Jsrc, Jdst, C;
  int32_t inv[MAXJ];  // per destination joint: source joint or -1
};
__global__ __launch_bounds__(256) void remap_kernel(const RemapArgs a) {
  const int64_t total = a.N * a.Jdst * a.C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t nj = i / a.C;
    int c = (int)(i - nj * a.C);
    int64_t n = nj / a.Jdst;
    int jd = (int)(nj - n * a.Jdst);
    int js = a.inv[jd];
    a.dst[i] = (js >= 0) ? a.src[(n * a.Jsrc + js) * a.C + c] : 0.f;
  }
}

static inline unsigned stream_grid(int64_t items) {
  int64_t blocks = (items + 255) / 256;
  const int64_t cap = 256 * 8;  // 8 workgroups per CU, grid-stride beyond
  return (unsigned)(blocks < 1 ? 1 : (blocks > cap ? cap : blocks));
}

}  // namespace p2c_aux

using namespace p2c_aux;

static int norm_launch(bool bwd, const float *x, const float *grad_out, float *out, float *shift, float *scale, int64_t N,
                       int32_t Jn, int32_t C, int32_t dim, int32_t transform, int32_t n_hips, const int32_t *hips,
                       int32_t n_neck, const int32_t *neck, float near_zero, void *stream_) {
  if (!x || !out || (bwd && !grad_out)) return P2C_E_NULL;
  if (N < 0 || Jn < 1 || Jn > MAXJ || (dim != 2 && dim != 3) || C < dim) return P2C_E_SHAPE;
  if (transform < P2C_TRANSFORM_HIPS_NECK || transform > P2C_TRANSFORM_HIPS_NECK_BBOX) return P2C_E_ENUM;
  if (dim == 3 && transform != P2C_TRANSFORM_HIPS_NECK) return P2C_E_ENUM;  // bbox extractors are 2-D only
  NormArgs a{};
  a.x = x, a.grad_out = grad_out, a.out = out, a.shift = shift, a.scale = scale;
  a.N = N, a.Jn = Jn, a.C = C, a.dim = dim, a.transform = transform, a.near_zero = near_zero;
  a.n_hips = 1, a.n_neck = 1;
  if (transform != P2C_TRANSFORM_BBOX) {
    if (!hips || !neck || n_hips < 1 || n_hips > 2 || n_neck < 1 || n_neck > 2) return P2C_E_INDEX;
    a.n_hips = n_hips, a.n_neck = n_neck;
    for (int i = 0; i < n_hips; ++i) {
      if (hips[i] < 0 || hips[i] >= Jn) return P2C_E_INDEX;
      a.hips_idx[i] = hips[i];
    }
    for (int i = 0; i < n_neck; ++i) {
      if (neck[i] < 0 || neck[i] >= Jn) return P2C_E_INDEX;
      a.neck_idx[i] = neck[i];
    }
  }
  if (N == 0) return 0;
  hipStream_t stream = (hipStream_t)stream_;
  const int G = (Jn <= 32) ? 32 : 64;
  int64_t waves = (N + (64 / G) - 1) / (64 / G);
  dim3 block(256), grid((unsigned)((waves + 3) / 4));
  if (G == 32) {
    if (bwd) hipLaunchKernelGGL((normalize_kernel<32, true>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((normalize_kernel<32, false>), grid, block, 0, stream, a);
  } else {
    if (bwd) hipLaunchKernelGGL((normalize_kernel<64, true>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((normalize_kernel<64, false>), grid, block, 0, stream, a);
  }
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

extern "C" int p2c_normalize_fwd(const float *x, float *out, float *shift, float *scale, int64_t N, int32_t Jn, int32_t C,
                                 int32_t dim, int32_t transform, int32_t n_hips, const int32_t *host_hips_idx,
                                 int32_t n_neck, const int32_t *host_neck_idx, float near_zero, void *stream) {
  return norm_launch(false, x, nullptr, out, shift, scale, N, Jn, C, dim, transform, n_hips, host_hips_idx, n_neck,
                     host_neck_idx, near_zero, stream);
}

extern "C" int p2c_normalize_bwd(const float *x, const float *grad_out, float *grad_x, int64_t N, int32_t Jn, int32_t C,
                                 int32_t dim, int32_t transform, int32_t n_hips, const int32_t *host_hips_idx,
                                 int32_t n_neck, const int32_t *host_neck_idx, float near_zero, void *stream) {
  return norm_launch(true, x, grad_out, grad_x, nullptr, nullptr, N, Jn, C, dim, transform, n_hips, host_hips_idx, n_neck,
                     host_neck_idx, near_zero, stream);
}

static int loss2d_args(Loss2dArgs &a, const float *pred, const float *gt, int64_t N, int32_t Jp, int32_t Cp, int32_t Jg,
                       int32_t Cg, int32_t K, const int32_t *pidx, const int32_t *gidx, int32_t hips_col, int32_t mask) {
  if (!pred || !gt || !pidx || !gidx) return P2C_E_NULL;
  if (N < 0 || Jp < 1 || Jg < 1 || Cp < 2 || Cg < 2 || K < 1 || K > MAXJ || Jp > MAXJ * 4) return P2C_E_SHAPE;
  if (hips_col < -1 || hips_col >= K) return P2C_E_INDEX;
  a.pred = pred, a.gt = gt, a.N = N, a.Jp = Jp, a.Cp = Cp, a.Jg = Jg, a.Cg = Cg, a.K = K, a.hips_col = hips_col, a.mask = mask;
  for (int k = 0; k < K; ++k) {
    if (pidx[k] < 0 || pidx[k] >= Jp || gidx[k] < 0 || gidx[k] >= Jg) return P2C_E_INDEX;
    a.pidx[k] = pidx[k], a.gidx[k] = gidx[k];
  }
  return 0;
}

extern "C" int64_t p2c_loss2d_workspace_floats(int64_t N) {
  (void)N;
  return (int64_t)256 * 8 * 4 * 2;  // waves of the largest grid x 2 floats
}

extern "C" int p2c_loss2d_fwd(const float *pred, const float *gt, int64_t N, int32_t Jp, int32_t Cp, int32_t Jg, int32_t Cg,
                              int32_t K, const int32_t *host_pred_idx, const int32_t *host_gt_idx, int32_t hips_col,
                              int32_t mask_missing_joints, float *partials, float *loss_sums, float *loss, void *stream_) {
  Loss2dArgs a{};
  int rc = loss2d_args(a, pred, gt, N, Jp, Cp, Jg, Cg, K, host_pred_idx, host_gt_idx, hips_col, mask_missing_joints);
  if (rc) return rc;
  if (!partials || !loss_sums || !loss) return P2C_E_NULL;
  hipStream_t stream = (hipStream_t)stream_;
  unsigned grid = stream_grid(N * K);
  hipLaunchKernelGGL(loss2d_fwd_kernel, dim3(grid), dim3(256), 0, stream, a, partials);
  hipLaunchKernelGGL(loss2d_finalize, dim3(1), dim3(256), 0, stream, (const float *)partials, (int)(grid * 4), loss_sums, loss);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

extern "C" int p2c_loss2d_bwd(const float *pred, const float *gt, int64_t N, int32_t Jp, int32_t Cp, int32_t Jg, int32_t Cg,
                              int32_t K, const int32_t *host_pred_idx, const int32_t *host_gt_idx, int32_t hips_col,
                              int32_t mask_missing_joints, const float *loss_sums, const float *grad_loss,
                              float *grad_pred, void *stream_) {
  Loss2dArgs a{};
  int rc = loss2d_args(a, pred, gt, N, Jp, Cp, Jg, Cg, K, host_pred_idx, host_gt_idx, hips_col, mask_missing_joints);
  if (rc) return rc;
  if (!loss_sums || !grad_loss || !grad_pred) return P2C_E_NULL;
  hipStream_t stream = (hipStream_t)stream_;
  hipLaunchKernelGGL(loss2d_bwd_kernel, dim3(stream_grid(N * Jp)), dim3(256), 0, stream, a, loss_sums, grad_loss, grad_pred);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

extern "C" int p2c_remap_nodes(const float *src, float *dst, int64_t N, int32_t Jsrc, int32_t Jdst, int32_t C, int32_t K,
                               const int32_t *host_src_idx, const int32_t *host_dst_idx, void *stream_) {
  if (!src || !dst || (K > 0 && (!host_src_idx || !host_dst_idx))) return P2C_E_NULL;
  if (N < 0 || Jsrc < 1 || Jdst < 1 || Jdst > MAXJ || C < 1 || K < 0 || K > MAXJ) return P2C_E_SHAPE;
  RemapArgs a{};
  a.src = src, a.dst = dst, a.N = N, a.Jsrc = Jsrc, a.Jdst = Jdst, a.C = C;
  for (int j = 0; j < MAXJ; ++j) a.inv[j] = -1;
  for (int k = 0; k < K; ++k) {
    if (host_src_idx[k] < 0 || host_src_idx[k] >= Jsrc || host_dst_idx[k] < 0 || host_dst_idx[k] >= Jdst) return P2C_E_INDEX;
    a.inv[host_dst_idx[k]] = host_src_idx[k];
  }
  if (N == 0) return 0;
  hipLaunchKernelGGL(remap_kernel, dim3(stream_grid(N * Jdst * C)), dim3(256), 0, (hipStream_t)stream_, a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

// ---- grouped copy: the tensors of a new batch into the static buffers of a captured step, ONE launch ------------------------
// A HIP-graph step reads fixed addresses, so every new batch (frames, the target tensors, skel_type: 12 tensors, 14 MB at
// B = 256) is copied into them first. As framework copies that is one launch per tensor (41 us of the 77 us a fresh-batch
// step took at B = 256, against 33 us for the step itself); here the (src, dst, bytes) table rides in the kernel arguments
// and workgroups are dealt out in proportion to the sizes.
namespace p2c_copy {
constexpr int MAXN = 24;
struct Args {
  const void *src[MAXN];
  void *dst[MAXN];
  int64_t bytes[MAXN];
  int32_t first[MAXN + 1];      // first workgroup of tensor i
  int32_t n;
};
__global__ __launch_bounds__(256) void copy_group_kernel(const Args a) {
  int i = 0;
  while (i + 1 < a.n && (int)blockIdx.x >= a.first[i + 1]) ++i;
  const int nb = a.first[i + 1] - a.first[i], b = blockIdx.x - a.first[i];
  const int64_t bytes = a.bytes[i];
  const char *s = static_cast<const char *>(a.src[i]);
  char *d = static_cast<char *>(a.dst[i]);
  if (((reinterpret_cast<uintptr_t>(s) | reinterpret_cast<uintptr_t>(d)) & 15) == 0) {
    const int64_t n16 = bytes >> 4;
    const float4 *s4 = reinterpret_cast<const float4 *>(s);
    float4 *d4 = reinterpret_cast<float4 *>(d);
    for (int64_t k = (int64_t)b * 256 + threadIdx.x; k < n16; k += (int64_t)nb * 256) d4[k] = s4[k];
    for (int64_t k = (n16 << 4) + (int64_t)b * 256 + threadIdx.x; k < bytes; k += (int64_t)nb * 256) d[k] = s[k];
  } else {
    for (int64_t k = (int64_t)b * 256 + threadIdx.x; k < bytes; k += (int64_t)nb * 256) d[k] = s[k];
  }
}
}  // namespace p2c_copy

extern "C" int p2c_copy_group(const void *const *src, void *const *dst, const int64_t *bytes, int32_t n, void *stream_) {
  using namespace p2c_copy;
  if (!src || !dst || !bytes) return P2C_E_NULL;
  if (n < 0 || n > MAXN) return P2C_E_SHAPE;
  Args a{};
  a.n = 0, a.first[0] = 0;
  for (int i = 0; i < n; ++i) {
    if (bytes[i] < 0) return P2C_E_SHAPE;
    if (bytes[i] == 0) continue;
    if (!src[i] || !dst[i]) return P2C_E_NULL;
    const int k = a.n++;
    a.src[k] = src[i], a.dst[k] = dst[i], a.bytes[k] = bytes[i];
    int64_t nb = (bytes[i] + 16383) / 16384;                    // 16 KB per workgroup pass, at most 256 workgroups per tensor
    if (nb > 256) nb = 256;
    a.first[k + 1] = a.first[k] + (int32_t)nb;
  }
  if (a.n == 0) return 0;
  hipLaunchKernelGGL(copy_group_kernel, dim3((unsigned)a.first[a.n]), dim3(256), 0, (hipStream_t)stream_, a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

// ---- inspection: what a captured HIP graph holds --------------------------------------------------------------------------
// The trainer replays a captured step by launching its recorded C-ABI call directly when the capture turned out to be
// nothing but that call's kernels (a graph launch costs ~5 us of start-up per replay; two direct launches ~2 us of gap).
// This is how it checks: total nodes and kernel nodes of the graph.
extern "C" int p2c_graph_node_counts(void *graph, int32_t *n_total, int32_t *n_kernel) {
  if (!graph || !n_total || !n_kernel) return P2C_E_NULL;
  size_t n = 0;
  hipError_t e = hipGraphGetNodes((hipGraph_t)graph, nullptr, &n);
  if (e != hipSuccess) return (int)e;
  *n_total = (int32_t)n, *n_kernel = 0;
  if (n == 0) return 0;
  hipGraphNode_t *nodes = new hipGraphNode_t[n];
  e = hipGraphGetNodes((hipGraph_t)graph, nodes, &n);
  for (size_t i = 0; e == hipSuccess && i < n; ++i) {
    hipGraphNodeType t;
    e = hipGraphNodeGetType(nodes[i], &t);
    if (e == hipSuccess && t == hipGraphNodeTypeKernel) ++*n_kernel;
  }
  delete[] nodes;
  return e == hipSuccess ? 0 : (int)e;
}


// ---- testing aid: fill the LDS of every CU with NaN bit patterns ----------------------------------------------------------------
// LDS keeps what the last workgroup on the CU left in it. A kernel that reads LDS it never wrote works or fails by that
// accident (tests/test_pose_head_gpu.py: a stale NaN behind the chain-lane backward's rotation image surfaced as an intermittent
// NaN in grad_y). After this launch every uninitialised LDS read returns NaN.
namespace p2c_aux {
__global__ __launch_bounds__(1024) void poison_lds_kernel(int n_floats) {
  extern __shared__ float poison[];
  for (int i = threadIdx.x; i < n_floats; i += blockDim.x) poison[i] = __builtin_nanf("");
  __syncthreads();
  if (poison[(threadIdx.x * 37) % n_floats] == 0.f) __builtin_trap();      // (keeps the stores alive)
}
}  // namespace p2c_aux

extern "C" int p2c_debug_poison_lds(void *stream) {
  const int bytes = 160 * 1024;
  hipError_t e = hipFuncSetAttribute((const void *)p2c_aux::poison_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(p2c_aux::poison_lds_kernel, dim3(256 * 8), dim3(1024), bytes, (hipStream_t)stream, bytes / 4);
  e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}


// ---- learned weighted mean over F frame tokens: out (B, C) = sum_f w[f] x[b, f, c] + bias ---------------------------------------
// PoseTransformer's weighted_mean = Conv1d(F, 1, kernel 1) (bound at modules/movements/pose_former/pose_former.py:62-76) as the
// streaming pass it is: a thread owns four channels of one sample and walks the F rows (16-byte loads, F <= 64 weights in
// registers). Its backward is element-wise (d x) plus two K12 contractions (d w, d bias): ops.FrameMeanFunction.
namespace p2c_aux {
__global__ __launch_bounds__(256) void frame_mean_kernel(const float *x, const float *w, const float *bias, float *out, int64_t B, int F,
                                                         int C) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  const int c4 = C >> 2;
  const float b0 = bias ? bias[0] : 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < B * c4; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = i / c4;
    const int c = (int)(i - b * c4) * 4;
    const float *p = x + (b * F) * C + c;
    f4 acc = {b0, b0, b0, b0};
    for (int f = 0; f < F; ++f) acc += *reinterpret_cast<const f4 *>(p + (int64_t)f * C) * w[f];
    *reinterpret_cast<f4 *>(out + b * C + c) = acc;
  }
}
}  // namespace p2c_aux

extern "C" int p2c_frame_mean_fwd(const float *x, const float *w, const float *bias, float *out, int64_t B, int32_t F, int32_t C,
                                  void *stream) {
  if (!x || !w || !out) return P2C_E_NULL;
  if (B < 0 || F < 1 || C < 4 || (C & 3) || ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) & 15)) return P2C_E_SHAPE;
  if (B == 0) return 0;
  const int64_t n = B * (C >> 2);
  const unsigned grid = (unsigned)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
  hipLaunchKernelGGL(p2c_aux::frame_mean_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, w, bias, out, B, F, C);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}
