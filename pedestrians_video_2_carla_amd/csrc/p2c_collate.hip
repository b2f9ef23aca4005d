// p2c_collate.hip -- K11: the dataset-side input pipeline of the reference as ONE streaming launch over a batch of clips
// (gfx950). The reference runs this chain per clip, on the CPU, inside DataLoader workers (BaseDataset.__getitem__,
// data/base/base_dataset.py:206-234): ~60 small tensor ops per clip, 32 worker processes to keep up with a GPU step
// of 0.4 s. With the train step at ~50 us that pipeline is what bounds a real run.
//
//   augmentation   transforms/pose/augmentation/augment_pose.py:43-76, random_flip.py:39-76, random_rotation.py:34-68
//   deform         data/base/mixins/dataset/projection_2d_mixin.py:137-171   (noise + per-joint missing mask)
//   normalise x2   projection_2d_mixin.py:209-232 -> transforms/pose/normalization/normalizer.py:20-41 with the
//                  hips_neck / bbox / hips_neck_bbox extractors (model input AND the projection_2d_transformed target)
//   confidence     data/base/mixins/dataset/confidence_mixin.py:13-20
//   node map       data/base/base_dataset.py:156-190 (_get_common_tensor, zero-filled scatter to the model's skeleton)
//
// One G-lane group (G = 32 or 64) owns one frame; lane = joint (data joint on the way in, model-input joint on the way
// out). Bounding boxes, hips / neck points and the joint permutations are wave shuffles; nothing is staged in LDS.
// HBM traffic is exactly the boundary tensors: the raw pose and the random draws in, six small tensors out.
// The random numbers are inputs (the host wrapper draws them on the device): the kernel is a pure function, so the
// reference, the oracle (oracle/collate.py) and this kernel can be compared on identical draws.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

#include "../../include/p2c.h"

namespace p2c_collate {

constexpr int MAXJ = 64;
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x3 __attribute__((ext_vector_type(3)));
// the (x, y) pairs and (x, y, conf) triples move as one 8- / 12-byte access per lane (4-byte aligned: torch tensors)
__device__ __forceinline__ f32x2 ld2(const float *p) { return *reinterpret_cast<const f32x2 __attribute__((aligned(4))) *>(p); }
__device__ __forceinline__ void st2(float *p, float a, float b) {
  *reinterpret_cast<f32x2 __attribute__((aligned(4))) *>(p) = (f32x2){a, b};
}

struct Args {
  const float *raw, *rotation, *bboxes, *clip_size, *noise, *miss_u;
  const uint8_t *is_flipped;
  float *frames, *t_projection_2d, *t_deformed, *t_transformed, *shift, *scale, *bboxes_out;
  int64_t frames_total;
  int32_t T, Jd, Ji, C, Cf, transform, n_hips, n_neck;
  int32_t hips_idx[2], neck_idx[2];
  float near_zero;
  int8_t perm[MAXJ];       // RandomFlip: new joint j takes the value of joint perm[j]
  int8_t inv[MAXJ];        // node map: model-input joint i takes data joint inv[i], -1 = zero
  float miss_prob[MAXJ];
};

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
__device__ __forceinline__ float finite_or_zero(float v) { return isfinite(v) ? v : 0.f; }   // nan_to_zero

// Normalizer.__call__ (normalizer.py:20-41), dim = 2, for the frame of this lane group: (x, y[, conf]) -> normalised
// (x, y), conf through nan_to_zero; shift / scale of the frame. Same arithmetic as p2c_aux::normalize_kernel.
template <int G>
__device__ __forceinline__ void normalise(const Args &a, int base, bool active, float x, float y, float conf,
                                          float &ox, float &oy, float &oconf, float (&s)[2], float &scale) {
  const int tr = a.transform;
  float k[2] = {0.f, 0.f};
  s[0] = s[1] = 0.f;
  scale = 1.f;
  const float p[2] = {x, y};
  if (tr != P2C_TRANSFORM_BBOX) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      float h = __shfl(p[c], base + a.hips_idx[0], 64);
      if (a.n_hips == 2) h = 0.5f * (h + __shfl(p[c], base + a.hips_idx[1], 64));
      float q = __shfl(p[c], base + a.neck_idx[0], 64);
      if (a.n_neck == 2) q = 0.5f * (q + __shfl(p[c], base + a.neck_idx[1], 64));
      s[c] = h, k[c] = q;
    }
    scale = sqrtf(fmaf(k[1] - s[1], k[1] - s[1], (k[0] - s[0]) * (k[0] - s[0])));
  }
  bool use_bb = false;
  if (tr == P2C_TRANSFORM_HIPS_NECK_BBOX)
    use_bb = (s[0] < a.near_zero && s[1] < a.near_zero) || (k[0] < a.near_zero && k[1] < a.near_zero);
  if (__any(tr == P2C_TRANSFORM_BBOX || use_bb)) {
    const bool missing = !active || (x < a.near_zero && y < a.near_zero);
    const float inf = __builtin_inff();
    const float mn0 = gmin<G>(missing ? inf : x), mn1 = gmin<G>(missing ? inf : y);
    const float mx0 = gmax<G>(missing ? -inf : x), mx1 = gmax<G>(missing ? -inf : y);
    const float cu = 0.5f * (mn0 + mx0), cv = 0.5f * (mn1 + mx1);
    const float dx = cu - cu, dy = fminf(mn1, mx1) - cv;
    const float bb_scale = sqrtf(fmaf(dx, dx, dy * dy));
    if (tr == P2C_TRANSFORM_BBOX) s[0] = cu, s[1] = cv, scale = bb_scale;
    else if (use_bb) scale = bb_scale * 0.5748f;
  }
  ox = finite_or_zero((x - s[0]) / scale), oy = finite_or_zero((y - s[1]) / scale);
  oconf = finite_or_zero(conf);
  if (a.C > 2 && !(oconf >= a.near_zero)) ox = 0.f, oy = 0.f;     // normalizer.py:35-37
}

template <int G>
__global__ __launch_bounds__(256) void collate_kernel(const Args a) {
  const int lane = threadIdx.x & 63;
  const int j = lane & (G - 1);
  const int base = lane & ~(G - 1);
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t f = wave * (64 / G) + (lane / G);           // frame index over N * T
  const bool live = f < a.frames_total;
  const int64_t fc = live ? f : 0;
  const int64_t n = fc / a.T;
  const bool active = live && j < a.Jd;
  const int C = a.C;

  // every HBM read of the frame goes out before the first use: one memory latency per wave, not one per stage
  float x = 0.f, y = 0.f, conf = 0.f, nx = 0.f, ny = 0.f, mu = 2.f;
  float bx[4] = {0.f, 0.f, 0.f, 0.f}, cw = 0.f, ch = 0.f, deg = 0.f;
  const bool flip = a.is_flipped && a.is_flipped[n] != 0;
  const bool rotate = a.rotation != nullptr;
  if (active) {
    const float *p = a.raw + (fc * a.Jd + j) * C;
    if (C > 2) {
      const f32x3 v = *reinterpret_cast<const f32x3 __attribute__((aligned(4))) *>(p);
      x = v[0], y = v[1], conf = v[2];
    } else {
      const f32x2 v = ld2(p);
      x = v[0], y = v[1];
    }
    if (a.noise) {
      const f32x2 q = ld2(a.noise + (fc * a.Jd + j) * 2);
      nx = q[0], ny = q[1];
    }
    if (a.miss_u) mu = a.miss_u[fc * a.Jd + j];
  }
  if (a.bboxes) {
    const float *b = a.bboxes + fc * 4;
    bx[0] = b[0], bx[1] = b[1], bx[2] = b[2], bx[3] = b[3];
  }
  if (a.clip_size) cw = a.clip_size[2 * n], ch = a.clip_size[2 * n + 1];
  if (rotate) deg = a.rotation[n];
  const int lj = j & (MAXJ - 1);
  const float mp = a.miss_prob[lj];               // per-lane reads of the kernel-argument tables, also up front
  const int perm_j = a.perm[lj], inv_j = a.inv[lj];

  // ---- AugmentPose: boxes and centres (augment_pose.py:57-60) ------------------------------------------------------
  float lo[2] = {0.f, 0.f}, hi[2] = {0.f, 0.f}, ctr[2] = {0.f, 0.f};
  if (a.is_flipped || rotate) {
    if (a.bboxes) {
      lo[0] = bx[0], lo[1] = bx[1], hi[0] = bx[2], hi[1] = bx[3];
    } else {                                                   // get_bboxes(pose), utils/tensors.py:12-26
      const bool missing = !active || (x < a.near_zero && y < a.near_zero);
      const float inf = __builtin_inff();
      lo[0] = gmin<G>(missing ? inf : x), lo[1] = gmin<G>(missing ? inf : y);
      hi[0] = gmax<G>(missing ? -inf : x), hi[1] = gmax<G>(missing ? -inf : y);
    }
    ctr[0] = (lo[0] + hi[0]) * 0.5f, ctr[1] = (lo[1] + hi[1]) * 0.5f;
  }
  // ---- RandomFlip (random_flip.py:39-76) -----------------------------------------------------------------------------
  if (__any(flip)) {
    const bool gone = (x == 0.f) || (y == 0.f) || (C > 2 && conf == 0.f);    // remembered before the permutation
    const int src = base + (active ? perm_j : j);
    const float qx = __shfl(x, src, 64), qy = __shfl(y, src, 64), qc = __shfl(conf, src, 64);
    if (flip) {
      float fx = (qx - ctr[0]) * -1.f;
      const bool sized = a.clip_size && cw != 0.f && ch != 0.f;
      if (sized) {
        const float half = cw / 2.f;
        const float l2 = (lo[0] - half) * -1.f + half, h2 = (hi[0] - half) * -1.f + half;
        lo[0] = h2, hi[0] = l2;
        ctr[0] = (lo[0] + hi[0]) * 0.5f;
      }
      fx += ctr[0];
      x = gone ? 0.f : fx, y = gone ? 0.f : qy, conf = gone ? 0.f : qc;
    }
  }
  // ---- RandomRotation (random_rotation.py:34-68) -----------------------------------------------------------------------
  if (rotate) {
    const bool gone = (x == 0.f) || (y == 0.f) || (C > 2 && conf == 0.f);
    const float rad = deg * 0.017453292519943295f;
    float sn, cs;
    sincosf(rad, &sn, &cs);                  // one range reduction for both
    const float dx = x - ctr[0], dy = y - ctr[1];
    const float rx = fmaf(dy, sn, dx * cs) + ctr[0], ry = fmaf(dy, cs, dx * -sn) + ctr[1];
    x = gone ? 0.f : rx, y = gone ? 0.f : ry, conf = gone ? 0.f : conf;
    // the four corners of the box about the same centre; the new box is their hull
    float nlo[2] = {__builtin_inff(), __builtin_inff()}, nhi[2] = {-__builtin_inff(), -__builtin_inff()};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float bx = ((q & 1) ? hi[0] : lo[0]) - ctr[0], by = ((q & 2) ? hi[1] : lo[1]) - ctr[1];
      const float cx = fmaf(by, sn, bx * cs) + ctr[0], cy = fmaf(by, cs, bx * -sn) + ctr[1];
      nlo[0] = fminf(nlo[0], cx), nlo[1] = fminf(nlo[1], cy), nhi[0] = fmaxf(nhi[0], cx), nhi[1] = fmaxf(nhi[1], cy);
    }
    lo[0] = nlo[0], lo[1] = nlo[1], hi[0] = nhi[0], hi[1] = nhi[1];
  }
  if (a.bboxes_out && live && j == 0) {
    float *b = a.bboxes_out + f * 4;
    b[0] = lo[0], b[1] = lo[1], b[2] = hi[0], b[3] = hi[1];
  }
  // ---- apply_deform (projection_2d_mixin.py:137-171) ---------------------------------------------------------------------
  float dx = x, dy = y;
  if (active) {
    if (a.noise) dx += nx, dy += ny;
    if (a.miss_u && mu < mp) dx = 0.f, dy = 0.f;
  }
  // ---- apply_transform twice: model input (deformed) and target (augmented); shift / scale of the second call ----------
  float ix = dx, iy = dy, ic = conf, tx = x, ty = y, s[2] = {0.f, 0.f}, scale = 1.f;
  if (a.transform != P2C_TRANSFORM_NONE) {
    float s0[2], sc0, tc;
    normalise<G>(a, base, active, dx, dy, conf, ix, iy, ic, s0, sc0);
    normalise<G>(a, base, active, x, y, conf, tx, ty, tc, s, scale);
    if (a.shift && live && j == 0) a.shift[f * 2] = s[0], a.shift[f * 2 + 1] = s[1];
    if (a.scale && live && j == 0) a.scale[f] = scale;
  }
  // ---- node map: lane = model-input joint --------------------------------------------------------------------------------
  const int srcj = (j < a.Ji) ? inv_j : -1;
  const int from = base + (srcj < 0 ? 0 : srcj);
  const float o_ix = __shfl(ix, from, 64), o_iy = __shfl(iy, from, 64), o_ic = __shfl(ic, from, 64);
  const float o_x = __shfl(x, from, 64), o_y = __shfl(y, from, 64);
  const float o_dx = __shfl(dx, from, 64), o_dy = __shfl(dy, from, 64);
  const float o_tx = __shfl(tx, from, 64), o_ty = __shfl(ty, from, 64);
  if (!live || j >= a.Ji) return;
  const bool has = srcj >= 0;
  const int64_t o = f * a.Ji + j;
  float *fr = a.frames + o * a.Cf;
  if (a.Cf > 2) {
    *reinterpret_cast<f32x3 __attribute__((aligned(4))) *>(fr) =
        (f32x3){has ? o_ix : 0.f, has ? o_iy : 0.f, has ? o_ic : 0.f};
  } else {
    st2(fr, has ? o_ix : 0.f, has ? o_iy : 0.f);
  }
  if (a.t_projection_2d) st2(a.t_projection_2d + 2 * o, has ? o_x : 0.f, has ? o_y : 0.f);
  if (a.t_deformed) st2(a.t_deformed + 2 * o, has ? o_dx : 0.f, has ? o_dy : 0.f);
  if (a.t_transformed) st2(a.t_transformed + 2 * o, has ? o_tx : 0.f, has ? o_ty : 0.f);
}

}  // namespace p2c_collate

extern "C" int p2c_collate_fwd(const p2c_collate_desc *d, void *stream_) {
  using namespace p2c_collate;
  if (!d) return P2C_E_NULL;
  if (d->N == 0) return 0;                                      // empty batch: nothing to read or write
  if (!d->raw || !d->frames) return P2C_E_NULL;
  if (d->N < 0 || d->T < 1 || d->Jd < 1 || d->Jd > MAXJ || d->Ji < 1 || d->Ji > MAXJ || (d->C != 2 && d->C != 3))
    return P2C_E_SHAPE;
  if (d->transform < P2C_TRANSFORM_NONE || d->transform > P2C_TRANSFORM_HIPS_NECK_BBOX) return P2C_E_ENUM;
  if (d->return_confidence && d->C != 3) return P2C_E_SHAPE;     // confidence_mixin.py:17-18 raises for a 2-channel pose
  if (d->rotation_deg && !d->bboxes && d->C != 2) return P2C_E_SHAPE;   // random_rotation.py:50: centres of a 3-channel box
  if (d->is_flipped && !d->flip_perm) return P2C_E_NULL;
  if (d->miss_u && !d->miss_prob) return P2C_E_NULL;
  if ((d->t_transformed || d->shift || d->scale) && d->transform == P2C_TRANSFORM_NONE) return P2C_E_ENUM;
  if (d->bboxes_out && !d->bboxes) return P2C_E_NULL;
  Args a{};
  a.raw = d->raw, a.rotation = d->rotation_deg, a.bboxes = d->bboxes, a.clip_size = d->clip_size, a.noise = d->noise;
  a.miss_u = d->miss_u, a.is_flipped = d->is_flipped, a.frames = d->frames, a.t_projection_2d = d->t_projection_2d;
  a.t_deformed = d->t_deformed, a.t_transformed = d->t_transformed, a.shift = d->shift, a.scale = d->scale;
  a.bboxes_out = d->bboxes_out;
  a.frames_total = d->N * d->T, a.T = d->T, a.Jd = d->Jd, a.Ji = d->Ji, a.C = d->C;
  a.Cf = d->return_confidence ? d->C : 2;
  a.transform = d->transform, a.near_zero = d->near_zero;
  if (d->transform != P2C_TRANSFORM_NONE && d->transform != P2C_TRANSFORM_BBOX) {
    if (d->n_hips < 1 || d->n_hips > 2 || d->n_neck < 1 || d->n_neck > 2) return P2C_E_SHAPE;
    for (int i = 0; i < d->n_hips; ++i)
      if (d->hips_idx[i] < 0 || d->hips_idx[i] >= d->Jd) return P2C_E_INDEX;
    for (int i = 0; i < d->n_neck; ++i)
      if (d->neck_idx[i] < 0 || d->neck_idx[i] >= d->Jd) return P2C_E_INDEX;
  }
  a.n_hips = d->n_hips, a.n_neck = d->n_neck;
  for (int i = 0; i < 2; ++i) a.hips_idx[i] = d->hips_idx[i], a.neck_idx[i] = d->neck_idx[i];
  for (int j = 0; j < MAXJ; ++j) a.perm[j] = (int8_t)j, a.inv[j] = -1, a.miss_prob[j] = 0.f;
  if (d->flip_perm)
    for (int j = 0; j < d->Jd; ++j) {
      if (d->flip_perm[j] < 0 || d->flip_perm[j] >= d->Jd) return P2C_E_INDEX;
      a.perm[j] = (int8_t)d->flip_perm[j];
    }
  if (d->miss_prob)
    for (int j = 0; j < d->Jd; ++j) a.miss_prob[j] = d->miss_prob[j];
  if (d->K > 0) {
    if (!d->src_idx || !d->dst_idx) return P2C_E_NULL;
    for (int k = 0; k < d->K; ++k) {
      if (d->src_idx[k] < 0 || d->src_idx[k] >= d->Jd || d->dst_idx[k] < 0 || d->dst_idx[k] >= d->Ji) return P2C_E_INDEX;
      a.inv[d->dst_idx[k]] = (int8_t)d->src_idx[k];
    }
  } else {
    if (d->Ji != d->Jd) return P2C_E_SHAPE;
    for (int j = 0; j < d->Jd; ++j) a.inv[j] = (int8_t)j;
  }
  if (a.frames_total == 0) return 0;
  const int G = (d->Jd <= 32 && d->Ji <= 32) ? 32 : 64;
  const int64_t waves = (a.frames_total + (64 / G) - 1) / (64 / G);
  const dim3 block(256), grid((unsigned)((waves + 3) / 4));
  if (G == 32) hipLaunchKernelGGL(collate_kernel<32>, grid, block, 0, (hipStream_t)stream_, a);
  else hipLaunchKernelGGL(collate_kernel<64>, grid, block, 0, (hipStream_t)stream_, a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}
