// p2c_eval.hip -- validation metrics on device: MPJPE, MRPE, PCK as streaming reductions (gfx950).
//
// The reference computes them with torchmetrics objects on gathered tensors (metrics/mpjpe.py:29-45, mrpe.py:38-76,
// pck.py:66-98): fancy indexing, boolean-mask gathers (dynamic shapes -> host syncs), norms, means. Here each metric
// update is one pass over the tensors the materialising pose head (K6) already wrote, plus a one-workgroup fixed-order
// reduction that ADDS into a persistent device state (sum, count) -- no host sync until compute().
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

#include "../../include/p2c.h"

namespace p2c_eval {

constexpr int JP = P2C_JOINTS;   // prediction joints (CARLA skeleton)

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

// ---- MPJPE + MRPE ------------------------------------------------------------------------------------------------------
struct Pose3dArgs {
  const float *pred, *gt;          // (B,T,26,3), (B,T,Jg,3)
  const float *wpred, *wgt;        // absolute world locations (B,T,3) or NULL (MRPE skipped)
  float *partials;                 // (n_waves, 2)
  int32_t B, T, Jg, n_common;
  int32_t gmap[JP];                // gt joint of prediction joint j, or -1
  int32_t pred_hips[2], n_pred_hips, gt_hips[2], n_gt_hips;
};

// a 32-lane group per clip (two clips per wavefront); lane = prediction joint; frames walked in order
__global__ __launch_bounds__(256) void pose3d_kernel(const Pose3dArgs a) {
  const int lane = threadIdx.x & 63, j = lane & 31;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t clip = wave * 2 + (lane >> 5);
  const bool clip_ok = clip < a.B;
  const int gm = (j < JP) ? a.gmap[j] : -1;
  float dsum = 0.f, rsum = 0.f;
  for (int t = 0; t < a.T; ++t) {
    const size_t frame = (size_t)clip * a.T + t;
    if (clip_ok && gm >= 0) {                                     // mpjpe.py:36-41: norm of the difference per common joint
      const float *p = a.pred + (frame * JP + j) * 3, *q = a.gt + (frame * a.Jg + gm) * 3;
      const float dx = p[0] - q[0], dy = p[1] - q[1], dz = p[2] - q[2];
      dsum += sqrtf(dx * dx + dy * dy + dz * dz);
    }
    if (clip_ok && j == 0 && a.wpred) {                           // mrpe.py:58-70: hips (mean of the hips points) + world
      float e[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        float hp = 0.f, hg = 0.f;
        for (int i = 0; i < a.n_pred_hips; ++i) hp += a.pred[(frame * JP + a.pred_hips[i]) * 3 + k];
        for (int i = 0; i < a.n_gt_hips; ++i) hg += a.gt[(frame * a.Jg + a.gt_hips[i]) * 3 + k];
        hp /= (float)a.n_pred_hips, hg /= (float)a.n_gt_hips;
        e[k] = (a.wpred[frame * 3 + k] + hp) - (a.wgt[frame * 3 + k] + hg);
      }
      rsum += sqrtf(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
    }
  }
  // per clip: mean over joints and frames (mpjpe.py:40-41), mean over frames (mrpe.py:68-71); then sum over clips
  float m = gsum<32>(dsum) / (float)(a.T * a.n_common);
  float r = gsum<32>(rsum) / (float)a.T;
  if (!clip_ok) m = 0.f, r = 0.f;
  const float m2 = m + __shfl_xor(m, 32, 64), r2 = r + __shfl_xor(r, 32, 64);
  if (lane == 0) a.partials[wave * 2] = m2, a.partials[wave * 2 + 1] = r2;
}

// state += (sum of partial column k, count) in fixed order
__global__ __launch_bounds__(256) void accumulate_kernel(const float *partials, int n, int cols, double *state,
                                                         double count0, double count1) {
  __shared__ double sh[256];
  for (int k = 0; k < cols; ++k) {
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += (double)partials[(size_t)i * cols + k];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
      if ((int)threadIdx.x < st) sh[threadIdx.x] += sh[threadIdx.x + st];
      __syncthreads();
    }
    if (threadIdx.x == 0) state[2 * k] += sh[0], state[2 * k + 1] += (k == 0 ? count0 : count1);
    __syncthreads();
  }
}

// ---- PCK ---------------------------------------------------------------------------------------------------------------
struct PckArgs {
  const float *pred, *gt, *mask_src;   // (N,Jp,Cp), (N,Jg,Cg), (N,Jg,Cg) [projection_2d of the targets, for the mask]
  float *partials;                     // (n_waves, 2): correct, total
  int64_t N;
  int32_t Jp, Cp, Jg, Cg;
  int32_t pmap[64];                    // prediction joint paired with gt joint i, or -1
  int32_t mask_missing, hips_joint;    // gt joint that is never masked (-1: none)
  int32_t norm_mode;                   // 0 = bounding-box diagonal of the gt frame, 1 = |neck - hips| of the gt frame
  int32_t hips_idx[2], n_hips, neck_idx[2], n_neck;
  float threshold, near_zero;
};

template <int G>
__global__ __launch_bounds__(256) void pck_kernel(const PckArgs a) {
  const int lane = threadIdx.x & 63, i = lane & (G - 1);
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t n = wave * (64 / G) + lane / G;
  const bool frame_ok = n < a.N, joint_ok = frame_ok && i < a.Jg;
  float gx = 0.f, gy = 0.f;
  if (joint_ok) {
    const float *q = a.gt + (n * a.Jg + i) * a.Cg;
    gx = q[0], gy = q[1];
  }
  // normalisation distance of the frame (pck.py:55-64)
  float norm;
  if (a.norm_mode == 0) {                                       // utils/tensors.py:12-26 over ALL gt joints
    const bool missing = !joint_ok || (gx < a.near_zero && gy < a.near_zero);
    const float inf = __builtin_inff();
    const float x0 = gmin<G>(missing ? inf : gx), y0 = gmin<G>(missing ? inf : gy);
    const float x1 = gmax<G>(missing ? -inf : gx), y1 = gmax<G>(missing ? -inf : gy);
    const float dx = x1 - x0, dy = y1 - y0;
    norm = sqrtf(dx * dx + dy * dy);
  } else {                                                      // hips_neck_extractor.py:6-13, extractor.py:27-28
    float hx = 0.f, hy = 0.f, kx = 0.f, ky = 0.f;
    const int base = lane & ~(G - 1);
    for (int k = 0; k < a.n_hips; ++k) hx += __shfl(gx, base + a.hips_idx[k], 64), hy += __shfl(gy, base + a.hips_idx[k], 64);
    for (int k = 0; k < a.n_neck; ++k) kx += __shfl(gx, base + a.neck_idx[k], 64), ky += __shfl(gy, base + a.neck_idx[k], 64);
    hx /= (float)a.n_hips, hy /= (float)a.n_hips, kx /= (float)a.n_neck, ky /= (float)a.n_neck;
    norm = sqrtf((kx - hx) * (kx - hx) + (ky - hy) * (ky - hy));
  }
  const bool frame_masked = norm < a.near_zero;                 // pck.py:82-83 (NaN compares false, as in torch)
  if (frame_masked) norm = 1.f;
  // joint mask (tensors.py:29-40 on targets['projection_2d'], pck.py:67-74)
  const int pj = joint_ok ? a.pmap[i] : -1;
  bool counted = pj >= 0 && !frame_masked;
  if (counted && a.mask_missing) {
    const float *m = a.mask_src + (n * a.Jg + i) * a.Cg;
    counted = (m[0] != 0.f && m[1] != 0.f) || i == a.hips_joint;
  }
  float correct = 0.f;
  if (counted) {
    const float *p = a.pred + (n * a.Jp + pj) * a.Cp;
    const float dx = (p[0] - gx) / norm, dy = (p[1] - gy) / norm;
    correct = (sqrtf(dx * dx + dy * dy) < a.threshold) ? 1.f : 0.f;
  }
  const float c = gsum<64>(correct), tot = gsum<64>(counted ? 1.f : 0.f);
  if (lane == 0) a.partials[wave * 2] = c, a.partials[wave * 2 + 1] = tot;
}

}  // namespace p2c_eval

using namespace p2c_eval;

extern "C" int64_t p2c_eval_workspace_floats(int64_t units) {
  if (units <= 0) return 0;
  return 2 * (units + 8);                         // two floats per wavefront; at most one wavefront per unit (+ tail)
}

extern "C" int p2c_eval_pose3d(const float *pred, const float *gt, int32_t B, int32_t T, int32_t Jg, const int32_t *gmap,
                               const int32_t *pred_hips, int32_t n_pred_hips, const int32_t *gt_hips, int32_t n_gt_hips,
                               const float *world_pred, const float *world_gt, float *partials, double *state,
                               void *stream_) {
  if (!pred || !gt || !gmap || !partials || !state) return P2C_E_NULL;
  if (B < 0 || T < 1 || Jg < 1 || n_pred_hips < 1 || n_pred_hips > 2 || n_gt_hips < 1 || n_gt_hips > 2) return P2C_E_SHAPE;
  if ((world_pred == nullptr) != (world_gt == nullptr)) return P2C_E_NULL;
  if (B == 0) return 0;
  Pose3dArgs a{};
  a.pred = pred, a.gt = gt, a.wpred = world_pred, a.wgt = world_gt, a.partials = partials;
  a.B = B, a.T = T, a.Jg = Jg;
  for (int j = 0; j < JP; ++j) {
    if (gmap[j] < -1 || gmap[j] >= Jg) return P2C_E_INDEX;
    a.gmap[j] = gmap[j];
    a.n_common += gmap[j] >= 0;
  }
  if (a.n_common == 0) return P2C_E_SHAPE;
  for (int i = 0; i < n_pred_hips; ++i) {
    if (pred_hips[i] < 0 || pred_hips[i] >= JP) return P2C_E_INDEX;
    a.pred_hips[i] = pred_hips[i];
  }
  for (int i = 0; i < n_gt_hips; ++i) {
    if (gt_hips[i] < 0 || gt_hips[i] >= Jg) return P2C_E_INDEX;
    a.gt_hips[i] = gt_hips[i];
  }
  a.n_pred_hips = n_pred_hips, a.n_gt_hips = n_gt_hips;
  hipStream_t stream = (hipStream_t)stream_;
  const int waves = (B + 1) / 2, blocks = (waves + 3) / 4;
  hipLaunchKernelGGL(pose3d_kernel, dim3(blocks), dim3(256), 0, stream, a);
  hipLaunchKernelGGL(accumulate_kernel, dim3(1), dim3(256), 0, stream, (const float *)partials, blocks * 4, 2, state,
                     (double)B, world_pred ? (double)B : 0.0);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

extern "C" int p2c_eval_pck(const float *pred, const float *gt, const float *mask_src, int64_t N, int32_t Jp, int32_t Cp,
                            int32_t Jg, int32_t Cg, const int32_t *pmap, int32_t mask_missing, int32_t hips_joint,
                            int32_t norm_mode, const int32_t *hips_idx, int32_t n_hips, const int32_t *neck_idx,
                            int32_t n_neck, float threshold, float near_zero, float *partials, double *state, void *stream_) {
  if (!pred || !gt || !pmap || !partials || !state) return P2C_E_NULL;
  if (N < 0 || Jp < 1 || Jg < 1 || Jg > 64 || Cp < 2 || Cg < 2 || norm_mode < 0 || norm_mode > 1) return P2C_E_SHAPE;
  if (N == 0) return 0;
  PckArgs a{};
  a.pred = pred, a.gt = gt, a.mask_src = mask_src ? mask_src : gt, a.partials = partials;
  a.N = N, a.Jp = Jp, a.Cp = Cp, a.Jg = Jg, a.Cg = Cg;
  for (int i = 0; i < Jg; ++i) {
    if (pmap[i] < -1 || pmap[i] >= Jp) return P2C_E_INDEX;
    a.pmap[i] = pmap[i];
  }
  for (int i = Jg; i < 64; ++i) a.pmap[i] = -1;
  a.mask_missing = mask_missing ? 1 : 0, a.hips_joint = hips_joint, a.norm_mode = norm_mode;
  if (norm_mode == 1) {
    if (!hips_idx || !neck_idx || n_hips < 1 || n_hips > 2 || n_neck < 1 || n_neck > 2) return P2C_E_INDEX;
    for (int i = 0; i < n_hips; ++i) a.hips_idx[i] = hips_idx[i];
    for (int i = 0; i < n_neck; ++i) a.neck_idx[i] = neck_idx[i];
    a.n_hips = n_hips, a.n_neck = n_neck;
  }
  a.threshold = threshold, a.near_zero = near_zero;
  hipStream_t stream = (hipStream_t)stream_;
  const int G = Jg <= 32 ? 32 : 64;
  const int64_t waves = (N + (64 / G) - 1) / (64 / G);
  const int blocks = (int)((waves + 3) / 4);
  if (G == 32) hipLaunchKernelGGL(pck_kernel<32>, dim3(blocks), dim3(256), 0, stream, a);
  else hipLaunchKernelGGL(pck_kernel<64>, dim3(blocks), dim3(256), 0, stream, a);
  // column 0 = correct -> state[0], column 1 = total -> state[2]; the "count" slots state[1], state[3] stay untouched
  hipLaunchKernelGGL(accumulate_kernel, dim3(1), dim3(256), 0, stream, (const float *)partials, blocks * 4, 2, state, 0.0, 0.0);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}
