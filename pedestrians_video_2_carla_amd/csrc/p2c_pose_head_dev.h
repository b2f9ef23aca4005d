// p2c_pose_head_dev.h -- device functions of the fused pose head (shared by p2c_pose_head.hip and p2c_train.hip).
// See p2c_pose_head.hip for the work decomposition; reference citations are per function.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

#include "../../include/p2c.h"

namespace p2c {

constexpr int J = P2C_JOINTS;
constexpr int GROUP = 32;
constexpr int IDL = 26;  // identity lane inside a group

// parent lane inside the group (data/carla/files/structure.yaml DFS order); root and idle lanes -> identity lane
static __constant__ int c_parent[GROUP] = {IDL, 0, 1, 2, 3, 4, 5, 6, 3, 8, 9, 9, 3, 12, 13, 14, 1, 16, 17, 18, 19, 1, 21, 22, 23, 24,
                                    26, 27, 28, 29, 30, 31};
// last lane of the subtree rooted at each joint
static __constant__ int c_subtree_end[GROUP] = {25, 25, 15, 15, 7, 7, 7, 7, 11, 11, 10, 11, 15, 15, 15, 15, 20, 20, 20, 20, 20,
                                         25, 25, 25, 25, 25, 26, 27, 28, 29, 30, 31};
// Tree walk schedule of the forward kinematics (pointer doubling, 3 rounds):
//   round 1: joints whose parent is the previous lane ("interior" joints of a limb chain) absorb it via DPP row_shr:1;
//   round 2: every joint composes with the transform of its first not-yet-absorbed ancestor c_anc_r2[j];
//   round 3: likewise with c_anc_r3[j]; afterwards every path reaches the root (checked on the host).
constexpr unsigned kInteriorMask = 0x3dee6feu;
static __constant__ int c_anc_r2[GROUP] = {26, 26, 0, 1, 2, 3, 4, 5, 3, 3, 8, 9, 3, 3, 12, 13, 1, 1, 16, 17, 18, 1, 1, 21, 22, 23,
                                    26, 27, 28, 29, 30, 31};
static __constant__ int c_anc_r3[GROUP] = {26, 26, 26, 26, 0, 1, 2, 3, 1, 1, 3, 3, 1, 1, 3, 3, 26, 26, 1, 1, 16, 26, 26, 1, 1, 21,
                                    26, 27, 28, 29, 30, 31};

struct V3 {
  float x, y, z;
};
struct M3 {
  float m[9];  // row-major
};

__device__ __forceinline__ V3 v3(float x, float y, float z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ float dot(V3 a, V3 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, a.z * b.z)); }
__device__ __forceinline__ V3 cross(V3 a, V3 b) {
  return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ M3 identity() { return M3{{1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f}}; }
__device__ __forceinline__ M3 zero3() { return M3{{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}}; }
// C = A @ B
__device__ __forceinline__ M3 mul(const M3 &a, const M3 &b) {
  M3 c;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int k = 0; k < 3; ++k)
      c.m[i * 3 + k] = fmaf(a.m[i * 3 + 0], b.m[0 + k], fmaf(a.m[i * 3 + 1], b.m[3 + k], a.m[i * 3 + 2] * b.m[6 + k]));
  return c;
}
// C = A^T @ B
__device__ __forceinline__ M3 mulTN(const M3 &a, const M3 &b) {
  M3 c;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int k = 0; k < 3; ++k)
      c.m[i * 3 + k] = fmaf(a.m[0 + i], b.m[0 + k], fmaf(a.m[3 + i], b.m[3 + k], a.m[6 + i] * b.m[6 + k]));
  return c;
}
// C = A @ B^T
__device__ __forceinline__ M3 mulNT(const M3 &a, const M3 &b) {
  M3 c;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int k = 0; k < 3; ++k)
      c.m[i * 3 + k] = fmaf(a.m[i * 3 + 0], b.m[k * 3 + 0], fmaf(a.m[i * 3 + 1], b.m[k * 3 + 1], a.m[i * 3 + 2] * b.m[k * 3 + 2]));
  return c;
}
// row vector times matrix
__device__ __forceinline__ V3 vmul(V3 v, const M3 &a) {
  return v3(fmaf(v.x, a.m[0], fmaf(v.y, a.m[3], v.z * a.m[6])), fmaf(v.x, a.m[1], fmaf(v.y, a.m[4], v.z * a.m[7])),
            fmaf(v.x, a.m[2], fmaf(v.y, a.m[5], v.z * a.m[8])));
}
// row vector times matrix transposed
__device__ __forceinline__ V3 vmulT(V3 v, const M3 &a) {
  return v3(fmaf(v.x, a.m[0], fmaf(v.y, a.m[1], v.z * a.m[2])), fmaf(v.x, a.m[3], fmaf(v.y, a.m[4], v.z * a.m[5])),
            fmaf(v.x, a.m[6], fmaf(v.y, a.m[7], v.z * a.m[8])));
}
__device__ __forceinline__ M3 add(const M3 &a, const M3 &b) {
  M3 c;
#pragma unroll
  for (int i = 0; i < 9; ++i) c.m[i] = a.m[i] + b.m[i];
  return c;
}

// ---- cross-lane helpers (64-wide wavefront, two 32-lane groups) ----------------------------------------------------
__device__ __forceinline__ float shfl(float v, int src_lane) { return __shfl(v, src_lane, 64); }
__device__ __forceinline__ V3 shfl(V3 v, int s) { return v3(shfl(v.x, s), shfl(v.y, s), shfl(v.z, s)); }
__device__ __forceinline__ M3 shfl(const M3 &a, int s) {
  M3 c;
#pragma unroll
  for (int i = 0; i < 9; ++i) c.m[i] = shfl(a.m[i], s);
  return c;
}
// ---- DPP (data-parallel primitives): cross-lane moves executed by the VALU itself, no LDS round trip ---------------
// dpp_ctrl encodings (GFX9): row_shr:n = 0x110+n, row_bcast15 = 0x142. bound_ctrl=true: lanes without a source read 0.
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp0(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, true));
}
// inclusive prefix sum over the lanes of a 32-lane group (lane order = DFS order of the joints): four shifts inside the
// 16-lane rows, then lane 15 of the even rows is added to the odd rows (row_mask 0b1010)
__device__ __forceinline__ float group_prefix(float v) {
  v += dpp0<0x111>(v);
  v += dpp0<0x112>(v);
  v += dpp0<0x114>(v);
  v += dpp0<0x118>(v);
  v += dpp0<0x142, 0xA>(v);
  return v;
}
// value of lane `j` of this lane's own group, for a wave-uniform j (v_readlane x2 + select)
__device__ __forceinline__ float group_bcast(float v, int j, bool upper) {
  float lo = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), j));
  float hi = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), j + 32));
  return upper ? hi : lo;
}
__device__ __forceinline__ float group_sum(float v, bool upper) { return group_bcast(group_prefix(v), 31, upper); }
__device__ __forceinline__ float group_min(float v) {
#pragma unroll
  for (int d = 16; d >= 1; d >>= 1) v = fminf(v, __shfl_xor(v, d, 64));
  return v;
}
__device__ __forceinline__ float group_max(float v) {
#pragma unroll
  for (int d = 16; d >= 1; d >>= 1) v = fmaxf(v, __shfl_xor(v, d, 64));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}

// 1-ulp hardware reciprocal / square root (v_rcp_f32, v_sqrt_f32): the IEEE-exact division and sqrt sequences cost ~10
// VALU each and the parity budget is 1e-4 relative
__device__ __forceinline__ float frcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fsqrt(float x) { return __builtin_amdgcn_sqrtf(x); }

__device__ __forceinline__ float nan_to_zero(float v) { return (isfinite(v)) ? v : 0.f; }  // utils/tensors.py:43-53

// ---- 6-D rotation -> matrix (pytorch3d 0.6.0 rotation_6d_to_matrix; movements/movements.py:105-118) -------------
struct SixD {
  V3 a1, a2, b1, b2;
  float n1, n2, d;  // clamped norms, b1.a2
  bool c1, c2;      // norm above the 1e-12 clamp (gradient flows through the norm)
};
__device__ __forceinline__ M3 rot6d_fwd(const float *y6, SixD &s) {
  s.a1 = v3(y6[0], y6[1], y6[2]);
  s.a2 = v3(y6[3], y6[4], y6[5]);
  float n1 = fsqrt(dot(s.a1, s.a1));
  s.c1 = n1 > 1e-12f;
  s.n1 = fmaxf(n1, 1e-12f);
  s.b1 = s.a1 * frcp(s.n1);
  s.d = dot(s.b1, s.a2);
  V3 u2 = s.a2 - s.b1 * s.d;
  float n2 = fsqrt(dot(u2, u2));
  s.c2 = n2 > 1e-12f;
  s.n2 = fmaxf(n2, 1e-12f);
  s.b2 = u2 * frcp(s.n2);
  V3 b3 = cross(s.b1, s.b2);
  return M3{{s.b1.x, s.b1.y, s.b1.z, s.b2.x, s.b2.y, s.b2.z, b3.x, b3.y, b3.z}};
}
// gradient wrt the 6 inputs given the gradient wrt the matrix rows
__device__ __forceinline__ void rot6d_bwd(const SixD &s, const M3 &g, float *gy6) {
  V3 g1 = v3(g.m[0], g.m[1], g.m[2]), g2 = v3(g.m[3], g.m[4], g.m[5]), g3 = v3(g.m[6], g.m[7], g.m[8]);
  V3 gb1 = g1 + cross(s.b2, g3);
  V3 gb2 = g2 + cross(g3, s.b1);
  float r2 = frcp(s.n2);
  V3 gu2 = s.c2 ? (gb2 - s.b2 * dot(s.b2, gb2)) * r2 : gb2 * r2;
  float k = dot(gu2, s.b1);
  V3 ga2 = gu2 - s.b1 * k;
  gb1 = gb1 - gu2 * s.d - s.a2 * k;
  float r1 = frcp(s.n1);
  V3 ga1 = s.c1 ? (gb1 - s.b1 * dot(s.b1, gb1)) * r1 : gb1 * r1;
  gy6[0] = ga1.x, gy6[1] = ga1.y, gy6[2] = ga1.z, gy6[3] = ga2.x, gy6[4] = ga2.y, gy6[5] = ga2.z;
}

// ---- per-lane constants ------------------------------------------------------------------------------------------------
struct LaneCtx {
  int lane, j, base, clip;
  bool active;  // real joint of a real clip
  int anc0, anc1, anc2, sub_end;  // parent lane; round-2 / round-3 ancestors of the FK tree walk
  bool interior;                  // parent is the previous lane
  int gm2, gm3;
  bool never_masked;
  bool has2, has3;  // this lane's joint takes part in loc_2d / loc_3d
};

__device__ __forceinline__ void fill_lane(LaneCtx &L, const p2c_pose_head_desc &d) {
  L.anc0 = c_parent[L.j];
  L.anc1 = c_anc_r2[L.j];
  L.anc2 = c_anc_r3[L.j];
  L.interior = (kInteriorMask >> L.j) & 1u;
  L.sub_end = c_subtree_end[L.j];
  L.gm2 = (L.j < J) ? d.gmap2d[L.j] : -1;
  L.gm3 = (L.j < J) ? d.gmap3d[L.j] : -1;
  L.never_masked = (L.j == d.hips_lane);
  L.has2 = L.active && d.gt2d && L.gm2 >= 0;
  L.has3 = L.active && d.gt3d && L.gm3 >= 0;
}

// clip-sequential kernels: a wavefront owns two clips (one per 32-lane group) and walks their frames in order
__device__ __forceinline__ LaneCtx make_lane(const p2c_pose_head_desc &d) {
  LaneCtx L;
  L.lane = threadIdx.x & 63;
  L.j = L.lane & 31;
  L.base = L.lane & 32;
  int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  L.clip = wave * 2 + (L.lane >> 5);
  L.active = (L.j < J) && (L.clip < d.B);
  fill_lane(L, d);
  return L;
}

// time-parallel kernels: a workgroup owns ONE clip, every 32-lane group one of its frames (t = 2 * wave + group)
__device__ __forceinline__ LaneCtx make_lane_tp(const p2c_pose_head_desc &d, int &t) {
  LaneCtx L;
  L.lane = threadIdx.x & 63;
  L.j = L.lane & 31;
  L.base = L.lane & 32;
  t = (int)(threadIdx.x >> 6) * 2 + (L.lane >> 5);
  L.clip = blockIdx.x;
  L.active = (L.j < J) && (t < d.T);
  fill_lane(L, d);
  return L;
}

// Forward kinematics by pointer doubling over the group (walker_control/p3d_pose.py:116-184):
//   abs_rot[j] = rel_rot[j] @ abs_rot[parent], abs_loc[j] = rel_loc[j] @ abs_rot[parent] + abs_loc[parent].
template <bool NEED_ROT = true>   // NEED_ROT = false: only the absolute locations are wanted (lean forward): the last
                                 // round's 3x3 product is skipped
__device__ __forceinline__ void fk_doubling(const LaneCtx &L, M3 &R, V3 &l) {
  {
    // round 1 without LDS: DPP row_shr:1 hands every lane the transform of lane-1
    M3 Ra;
#pragma unroll
    for (int i = 0; i < 9; ++i) Ra.m[i] = dpp0<0x111>(R.m[i]);
    V3 la = v3(dpp0<0x111>(l.x), dpp0<0x111>(l.y), dpp0<0x111>(l.z));
    V3 l1 = vmul(l, Ra) + la;
    M3 R1 = mul(R, Ra);
    if (L.interior) {
      l = l1;
      R = R1;
    }
  }
  {
    M3 Ra = shfl(R, L.base + L.anc1);
    V3 la = shfl(l, L.base + L.anc1);
    l = vmul(l, Ra) + la;
    R = mul(R, Ra);
  }
  {
    M3 Ra = shfl(R, L.base + L.anc2);
    V3 la = shfl(l, L.base + L.anc2);
    l = vmul(l, Ra) + la;
    if (NEED_ROT) R = mul(R, Ra);
  }
}

// ---- projection + normaliser + losses for one frame -----------------------------------------------------------------
// Forward restates walker_control/p3d_pose_projection.py:115-152 (camera :37-69), normalizer.py:20-41 with the
// extractors of transforms/pose/normalization/, loss/loc_2d.py:69-89 + base_pose_loss.py:36-66, loss/loc_3d.py:12-40.
// With BWD it also returns d(total)/d(abs_loc) of this lane's joint.
struct World {
  M3 rot;
  V3 loc;
  bool on;
};

struct HeadAcc {
  float sum2, cnt2, sum3;
};

enum { MODE_FWD = 0, MODE_FWD_MATERIALIZE = 1, MODE_BWD = 2, MODE_TRAIN = 3 };   // TRAIN = BWD that also sums the losses

template <int MODE>
__device__ __forceinline__ V3 frame_head(const p2c_pose_head_desc &d, const LaneCtx &L, int t, V3 x, const World &W,
                                         HeadAcc &acc, float coef2, float coef3, const float *g_abs_ext,
                                         const float *g_projt_ext, const float *gt2v, const float *gt3v) {
  constexpr bool BWD = (MODE == MODE_BWD || MODE == MODE_TRAIN);
  constexpr bool SUMS = !BWD || MODE == MODE_TRAIN;
  constexpr bool MAT = (MODE == MODE_FWD_MATERIALIZE);
  const bool in_slice = (t >= d.t0) && (t < d.t1);
  const size_t frame = (size_t)L.clip * d.T + t;

  // ---- projection --------------------------------------------------------------------------------------------------
  V3 w = v3(x.y, -x.x, x.z);  // x @ p3d_2_world
  V3 p = W.on ? vmul(w, W.rot) + W.loc : w;
  float Z = d.cam_dist - p.x;
  float invZ = frcp(Z);
  float u = d.cam_cx - d.cam_f * p.y * invZ;
  float v = d.cam_cy + d.cam_f * (p.z + d.cam_elev) * invZ;
  if (!L.active) { u = 0.f; v = 0.f; invZ = 0.f; }

  if (MAT) {
    if (L.active && d.out_projection_2d) {
      float *o = d.out_projection_2d + (frame * J + L.j) * 3;
      o[0] = u, o[1] = v, o[2] = invZ;
    }
    if (L.active && d.out_absolute_pose_loc) {
      float *o = d.out_absolute_pose_loc + (frame * J + L.j) * 3;
      o[0] = x.x, o[1] = x.y, o[2] = x.z;
    }
  }

  // ---- normaliser ---------------------------------------------------------------------------------------------------
  const int tr = d.transform;
  float su = 0.f, sv = 0.f, scale = 1.f;         // shift, scale
  float hu = 0.f, hv = 0.f, ku = 0.f, kv = 0.f;  // hips / neck points
  float hn_scale = 1.f, bb_scale = 1.f;
  bool use_bb_scale = false;
  float minu = 0.f, maxu = 0.f, minv = 0.f, maxv = 0.f;
  bool missing = false;
  if (tr != P2C_TRANSFORM_NONE) {
    if (tr != P2C_TRANSFORM_BBOX) {  // hips_neck_extractor.py:6-13 (mean over the point tuple)
      const bool up = L.base != 0;
      hu = group_bcast(u, d.hips_idx[0], up);
      hv = group_bcast(v, d.hips_idx[0], up);
      if (d.n_hips == 2) {
        hu = 0.5f * (hu + group_bcast(u, d.hips_idx[1], up));
        hv = 0.5f * (hv + group_bcast(v, d.hips_idx[1], up));
      }
      ku = group_bcast(u, d.neck_idx[0], up);
      kv = group_bcast(v, d.neck_idx[0], up);
      if (d.n_neck == 2) {
        ku = 0.5f * (ku + group_bcast(u, d.neck_idx[1], up));
        kv = 0.5f * (kv + group_bcast(v, d.neck_idx[1], up));
      }
      float du = ku - hu, dv = kv - hv;
      hn_scale = fsqrt(fmaf(du, du, dv * dv));  // extractor.py:27-28
      su = hu, sv = hv, scale = hn_scale;
    }
    bool need_bb = (tr == P2C_TRANSFORM_BBOX);
    if (tr == P2C_TRANSFORM_HIPS_NECK_BBOX) {  // hips_neck_bbox_fallback_extractor.py:25,33
      bool mh = (hu < d.near_zero) && (hv < d.near_zero);
      bool mk = (ku < d.near_zero) && (kv < d.near_zero);
      use_bb_scale = mh || mk;
      need_bb = use_bb_scale;
    }
    if (__any(need_bb)) {  // utils/tensors.py:12-26, bbox_extractor.py:6-18
      missing = !L.active || ((u < d.near_zero) && (v < d.near_zero)) || (L.j >= J);
      const float inf = __builtin_inff();
      minu = group_min(missing ? inf : u);
      minv = group_min(missing ? inf : v);
      maxu = group_max(missing ? -inf : u);
      maxv = group_max(missing ? -inf : v);
      float cu = 0.5f * (minu + maxu), cv = 0.5f * (minv + maxv);
      float top_v = fminf(minv, maxv);
      float dx = cu - cu, dy = top_v - cv;  // literal: inf - inf = nan when every joint is missing
      bb_scale = fsqrt(fmaf(dx, dx, dy * dy));
      if (tr == P2C_TRANSFORM_BBOX) {
        su = cu, sv = cv, scale = bb_scale;
      } else if (use_bb_scale) {
        scale = bb_scale * 0.5748f;  // :18,:34-38 ; the shift fallback (:26-31) is a no-op in the reference
      }
    }
  }
  float nu = u, nv = v, wch = invZ;
  bool fin_u = true, fin_v = true, keep = true;
  float inv_scale = 1.f;
  if (tr != P2C_TRANSFORM_NONE) {
    inv_scale = frcp(scale);
    nu = (u - su) * inv_scale;  // normalizer.py:24-25
    nv = (v - sv) * inv_scale;
    fin_u = isfinite(nu), fin_v = isfinite(nv);
    nu = fin_u ? nu : 0.f;  // :30
    nv = fin_v ? nv : 0.f;
    wch = nan_to_zero(invZ);
    keep = wch >= d.near_zero;  // :35-37 third channel (1/depth) acts as the confidence
    if (!keep) { nu = 0.f; nv = 0.f; }
  }
  if (MAT && tr != P2C_TRANSFORM_NONE && in_slice && L.clip < d.B) {
    if (L.active && d.out_projection_2d_transformed) {
      float *o = d.out_projection_2d_transformed + (frame * J + L.j) * 3;
      o[0] = nu, o[1] = nv, o[2] = wch;
    }
    if (L.j == 0) {
      if (d.out_shift) d.out_shift[frame * 2 + 0] = su, d.out_shift[frame * 2 + 1] = sv;
      if (d.out_scale) d.out_scale[frame] = scale;
    }
  }

  // ---- losses -------------------------------------------------------------------------------------------------------
  V3 gx = v3(0.f, 0.f, 0.f);
  float dnu = 0.f, dnv = 0.f;  // d total / d normalised (u, v)
  if (in_slice) {
    if (L.has2) {
      float g0 = gt2v[0], g1 = gt2v[1];
      bool m = !d.mask_missing_joints || L.never_masked || ((g0 != 0.f) && (g1 != 0.f));  // tensors.py:29-40
      if (m) {
        float e0 = nu - g0, e1 = nv - g1;
        if (SUMS) {
          acc.sum2 += fmaf(e0, e0, e1 * e1);
          acc.cnt2 += 1.f;
        }
        if (BWD) {
          dnu = coef2 * e0;
          dnv = coef2 * e1;
        }
      }
    }
    if (L.has3) {
      float e0 = x.x - gt3v[0], e1 = x.y - gt3v[1], e2 = x.z - gt3v[2];
      if (SUMS) acc.sum3 += fmaf(e0, e0, fmaf(e1, e1, e2 * e2));
      if (BWD) gx = v3(coef3 * e0, coef3 * e1, coef3 * e2);
    }
  }
  if (!BWD) return gx;

  // ================================================ backward ==========================================================
  if (L.active && g_abs_ext) {
    const float *g = g_abs_ext + (frame * J + L.j) * 3;
    gx = gx + v3(g[0], g[1], g[2]);
  }
  if (L.active && g_projt_ext && in_slice) {
    const float *g = g_projt_ext + (frame * J + L.j) * 3;
    dnu += g[0], dnv += g[1];
  }
  float gu, gv;
  if (tr == P2C_TRANSFORM_NONE) {
    gu = dnu, gv = dnv;
  } else {
    // where(keep) and nan_to_num pass the gradient only through kept, finite entries
    if (!keep || !fin_u) dnu = 0.f;
    if (!keep || !fin_v) dnv = 0.f;
    bool ok = isfinite(inv_scale) && (scale != 0.f);
    gu = ok ? dnu * inv_scale : 0.f;
    gv = ok ? dnv * inv_scale : 0.f;
    float Au = group_sum(gu, L.base != 0), Av = group_sum(gv, L.base != 0);                 // -d/d shift
    float Cs = group_sum(fmaf(gu, nu, gv * nv), L.base != 0);                  // -d/d scale  (n = (p - shift)/scale)
    float g_scale = -Cs;
    float gsu = -Au, gsv = -Av;                                   // gradient wrt the shift point
    float g_bbs = 0.f;
    if (tr == P2C_TRANSFORM_BBOX) {
      g_bbs = g_scale;
    } else if (use_bb_scale) {
      g_bbs = g_scale * 0.5748f;
    } else {
      // scale = |neck - hips| (torch.linalg.norm backward; zero norm -> zero gradient)
      float r = (hn_scale > 0.f) ? g_scale * frcp(hn_scale) : 0.f;
      float gku = r * (ku - hu), gkv = r * (kv - hv);
      gsu -= gku, gsv -= gkv;
      float kn = 1.f / (float)d.n_neck;
      if (L.j == d.neck_idx[0] || (d.n_neck == 2 && L.j == d.neck_idx[1])) gu += gku * kn, gv += gkv * kn;
    }
    if (tr != P2C_TRANSFORM_BBOX) {
      float hn = 1.f / (float)d.n_hips;
      if (L.j == d.hips_idx[0] || (d.n_hips == 2 && L.j == d.hips_idx[1])) gu += gsu * hn, gv += gsv * hn;
    }
    if (tr == P2C_TRANSFORM_BBOX || __any(use_bb_scale)) {
      // min / max pick the first joint holding the extreme value (torch.min/max(dim) backward)
      float g_minu = 0.f, g_maxu = 0.f, g_minv = 0.f, g_maxv = 0.f;
      if (tr == P2C_TRANSFORM_BBOX) {  // shift = centre of the box
        g_minu += 0.5f * gsu, g_maxu += 0.5f * gsu, g_minv += 0.5f * gsv, g_maxv += 0.5f * gsv;
      }
      if (tr == P2C_TRANSFORM_BBOX || use_bb_scale) {
        float dy = fminf(minv, maxv) - 0.5f * (minv + maxv);
        float g_dy = (bb_scale > 0.f) ? g_bbs * dy * frcp(bb_scale) : 0.f;
        g_minv += 0.5f * g_dy;   // top_v = minv (+g_dy), centre (-g_dy/2 each)
        g_maxv -= 0.5f * g_dy;
      }
      unsigned long long grp = 0xffffffffull << L.base;
      unsigned long long b;
      b = __ballot(!missing && u == minu) & grp;
      if (b && L.lane == __ffsll((long long)b) - 1) gu += g_minu;
      b = __ballot(!missing && u == maxu) & grp;
      if (b && L.lane == __ffsll((long long)b) - 1) gu += g_maxu;
      b = __ballot(!missing && v == minv) & grp;
      if (b && L.lane == __ffsll((long long)b) - 1) gv += g_minv;
      b = __ballot(!missing && v == maxv) & grp;
      if (b && L.lane == __ffsll((long long)b) - 1) gv += g_maxv;
    }
  }
  // projection backward
  float fz = d.cam_f * invZ;
  float gb = -fz * gu;
  float gc = fz * gv;
  float gZ = (d.cam_f * p.y * gu - d.cam_f * (p.z + d.cam_elev) * gv) * invZ * invZ;
  V3 gp = v3(-gZ, gb, gc);
  V3 gw = W.on ? vmulT(gp, W.rot) : gp;
  if (L.active) gx = gx + v3(-gw.y, gw.x, gw.z);
  return gx;
}

// world transform scan step (utils/world.py:16-63): rot[t] = rot[t-1] @ drot[t], loc[t] = loc[t-1] + dloc[t]
__device__ __forceinline__ void world_step(const p2c_pose_head_desc &d, const LaneCtx &L, int t, World &W) {
  if (!W.on || L.clip >= d.B) return;
  size_t frame = (size_t)L.clip * d.T + t;
  if (d.drot) {
    M3 dr;
#pragma unroll
    for (int i = 0; i < 9; ++i) dr.m[i] = d.drot[frame * 9 + i];
    W.rot = d.world_absolute ? dr : mul(W.rot, dr);
  }
  if (d.dloc) {
    V3 dl = v3(d.dloc[frame * 3 + 0], d.dloc[frame * 3 + 1], d.dloc[frame * 3 + 2]);
    W.loc = d.world_absolute ? dl : W.loc + dl;
  }
}
__device__ __forceinline__ void world_store(const p2c_pose_head_desc &d, const LaneCtx &L, int t, const World &W) {
  if (L.j != 0 || L.clip >= d.B) return;
  size_t frame = (size_t)L.clip * d.T + t;
  if (d.out_world_loc) {
    d.out_world_loc[frame * 3 + 0] = W.loc.x, d.out_world_loc[frame * 3 + 1] = W.loc.y, d.out_world_loc[frame * 3 + 2] = W.loc.z;
  }
  if (d.out_world_rot) {
#pragma unroll
    for (int i = 0; i < 9; ++i) d.out_world_rot[frame * 9 + i] = W.rot.m[i];
  }
}

// loss scaling: loc_2d = S2 / (2 N2) -> d/dn = (n - g) / N2 ;  loc_3d = S3 / N3 -> d/dx = 2 (x - g) / N3.
// grad_losses = upstream gradients of (loc_2d, loc_3d, loc_2d_3d); loc_2d_3d = loc_2d + loc_3d (loss/loc_2d_3d.py:15).
// Each upstream gradient is its own (nullable) device scalar: autograd hands the gradient of the one loss that was
// used straight through, without a scatter into a 3-vector.
struct GradLosses {
  const float *p[3];
};
__device__ __forceinline__ void loss_coefs_n(const p2c_pose_head_desc &d, const GradLosses &gl, float n2, float n3,
                                             float &coef2, float &coef3) {
  if (!gl.p[0] && !gl.p[1] && !gl.p[2]) return;
  const float u0 = gl.p[0] ? *gl.p[0] : 0.f, u1 = gl.p[1] ? *gl.p[1] : 0.f, u2 = gl.p[2] ? *gl.p[2] : 0.f;
  float g2 = u0 + u2, g3 = u1 + u2;
  coef2 = (d.gt2d && n2 > 0.f) ? g2 / n2 : 0.f;
  coef3 = (d.gt3d && n3 > 0.f) ? 2.f * g3 / n3 : 0.f;
}
__device__ __forceinline__ void loss_coefs(const p2c_pose_head_desc &d, const GradLosses &gl, float &coef2, float &coef3) {
  if (!gl.p[0] && !gl.p[1] && !gl.p[2]) return;
  float n2 = d.loss_sums[1], n3 = d.loss_sums[3];
  const float u0 = gl.p[0] ? *gl.p[0] : 0.f, u1 = gl.p[1] ? *gl.p[1] : 0.f, u2 = gl.p[2] ? *gl.p[2] : 0.f;
  float g2 = u0 + u2, g3 = u1 + u2;
  coef2 = (d.gt2d && n2 > 0.f) ? g2 / n2 : 0.f;
  coef3 = (d.gt3d && n3 > 0.f) ? 2.f * g3 / n3 : 0.f;
}

// Torque of an upstream gradient G on the ABSOLUTE ROTATION of this lane's joint (rot_3d-type losses, loss/rot_3d.py:9-37):
// a virtual world rotation d_theta of the subtree turns A_m into A_m (I + [d_theta]x), so dL = d_theta . t_m with
// t_m = (P_zy - P_yz, P_xz - P_zx, P_yx - P_xy), P = A_m^T G_m; its subtree sum joins the location torque
// (tools/proto_rot_bwd.py: 1e-15 against autograd).
__device__ __forceinline__ V3 rotation_torque(const float *g_rot_ext, size_t joint_frame, const M3 &A, bool active) {
  if (!active) return v3(0.f, 0.f, 0.f);
  M3 G;
  const float *p = g_rot_ext + joint_frame * 9;
#pragma unroll
  for (int i = 0; i < 9; ++i) G.m[i] = p[i];
  const M3 P = mulTN(A, G);
  return v3(P.m[7] - P.m[5], P.m[2] - P.m[6], P.m[3] - P.m[1]);
}

// rot_3d fused (p2c_pose_head_desc.gt_rot): difference between this lane's absolute rotation and its target, or false when the
// joint / frame takes no part (not a common joint, outside the eval slice, idle lane)
__device__ __forceinline__ bool rot_difference(const p2c_pose_head_desc &d, const LaneCtx &L, int t, const M3 &A, M3 &D) {
  if (!(L.active && L.gm3 >= 0 && t >= d.t0 && t < d.t1 && t < d.T)) return false;
  const float *g = d.gt_rot + ((((size_t)L.clip * d.T + t) * d.gt3d_joints) + L.gm3) * 9;
#pragma unroll
  for (int i = 0; i < 9; ++i) D.m[i] = A.m[i] - g[i];
  return true;
}
__device__ __forceinline__ float rot_loss_term(const p2c_pose_head_desc &d, const LaneCtx &L, int t, const M3 &A) {
  M3 D;
  if (!rot_difference(d, L, t, A, D)) return 0.f;
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 9; ++i) s = fmaf(D.m[i], D.m[i], s);
  return s;
}
// its torque: G = coef (A - gt) in place of an upstream gradient tensor (rotation_torque above)
__device__ __forceinline__ V3 rot_loss_torque(const p2c_pose_head_desc &d, const LaneCtx &L, int t, const M3 &A, float coef) {
  M3 D;
  if (!rot_difference(d, L, t, A, D)) return v3(0.f, 0.f, 0.f);
  const M3 P = mulTN(A, D);
  return v3(P.m[7] - P.m[5], P.m[2] - P.m[6], P.m[3] - P.m[1]) * coef;
}
__device__ __forceinline__ float rot_coef(const p2c_pose_head_desc &d) {       // d rot_3d / d A = 2 (A - gt) / n
  if (!d.gt_rot || !d.grad_loss_rot) return 0.f;
  const float n = d.loss_sums[5];
  return n > 0.f ? 2.f * *d.grad_loss_rot / n : 0.f;
}

template <int KIND>
struct KindTraits {
  static constexpr bool SIXD = (KIND == P2C_KIND_POSE_CHANGES_6D || KIND == P2C_KIND_RELATIVE_ROT_6D);
  static constexpr bool SCAN = (KIND == P2C_KIND_POSE_CHANGES_6D || KIND == P2C_KIND_POSE_CHANGES_MAT);
  static constexpr int NY = SIXD ? 6 : 9;
};

template <int NY>
__device__ __forceinline__ void load_y(const float *y, size_t idx, float *dst) {
  const float *p = y + idx * NY;
  if (NY == 6) {
    const float2 *q = reinterpret_cast<const float2 *>(p);
    float2 a = q[0], b = q[1], c = q[2];
    dst[0] = a.x, dst[1] = a.y, dst[2] = b.x, dst[3] = b.y, dst[4] = c.x, dst[5] = c.y;
  } else {
#pragma unroll
    for (int i = 0; i < NY; ++i) dst[i] = p[i];
  }
}
// everything one lane reads from HBM for one frame; loaded one frame AHEAD of its use (software prefetch: the loads
// of frame t+1 are in flight while frame t is computed)
template <int NY>
struct FrameIn {
  float y[NY];
  float g2[2];
  float g3[3];
};
// Per-lane read cursors, computed once and advanced by one frame (a constant stride) per iteration. The frame inputs go
// through BUFFER loads whose range check stands in for `if (active)` / `if (has2)` / `if (has3)`: a lane without the input
// addresses past num_records and reads zeros. Guarded loads put exec-masked branches into the frame loop, and at their joins
// the compiler's waitcnt pass falls back to vmcnt(0) -- the loads of frame t + 1, issued one line earlier as a software
// prefetch, were waited for before frame t was computed (measured: see DESIGN section 5, large-batch pose head). The descriptors
// are per wave (base = the wave's first clip, records = the one or two clips the wave owns): offsets stay small for any B.
typedef unsigned int fb_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int fb_u32x3 __attribute__((ext_vector_type(3)));
typedef unsigned int fb_u32x4 __attribute__((ext_vector_type(4)));
constexpr int FRAME_OOB = 0x7fffff00;
struct FramePtrs {
  __amdgpu_buffer_rsrc_t y, g2, g3;
  int oy, o2, o3;      // byte offset of this lane's current frame (FRAME_OOB for a lane without that input)
  int sy, s2, s3;      // bytes per frame (0 for such a lane)
  float idle;          // 1 for lanes that hold the identity rotation
};
template <int NY>
__device__ __forceinline__ FramePtrs frame_ptrs(const p2c_pose_head_desc &d, const LaneCtx &L, int t) {
  FramePtrs p;
  const int clip0 = __builtin_amdgcn_readfirstlane(L.clip);          // lane 0's clip = the wave's first
  const int avail = clip0 < d.B ? (d.B - clip0 < 2 ? d.B - clip0 : 2) : 0;
  const int fy = J * NY * 4, f2 = d.gt2d_joints * d.gt2d_channels * 4, f3 = d.gt3d_joints * 3 * 4;   // bytes per frame
  const int cy = d.T * fy, c2 = d.T * f2, c3 = d.T * f3;                                                // bytes per clip
  auto rsrc = [&](const float *base, int clip_bytes) {
    const uintptr_t q = reinterpret_cast<uintptr_t>(base) + (size_t)clip0 * clip_bytes;
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(q), 0, base ? avail * clip_bytes : 0, 0x00020000);
  };
  p.y = rsrc(d.y, cy), p.g2 = rsrc(d.gt2d, c2), p.g3 = rsrc(d.gt3d, c3);
  const int lc = L.clip - clip0;
  p.oy = L.active ? lc * cy + (t * J + L.j) * NY * 4 : FRAME_OOB, p.sy = L.active ? fy : 0;
  p.o2 = L.has2 ? lc * c2 + (t * d.gt2d_joints + L.gm2) * d.gt2d_channels * 4 : FRAME_OOB, p.s2 = L.has2 ? f2 : 0;
  p.o3 = L.has3 ? lc * c3 + (t * d.gt3d_joints + L.gm3) * 12 : FRAME_OOB, p.s3 = L.has3 ? f3 : 0;
  p.idle = L.active ? 0.f : 1.f;
  return p;
}
// Idle lanes (joint slots 26..31, clips beyond the batch) read zeros; where a frame's y is CONSUMED they become the identity
// rotation (added at the use, not at the load: an add right behind the load would wait for the prefetch it has just issued).
template <int NY>
__device__ __forceinline__ void rotation_input(const FrameIn<NY> &f, const FramePtrs &p, float (&y)[NY]) {
#pragma unroll
  for (int i = 0; i < NY; ++i) y[i] = f.y[i];
  if (NY == 6) y[0] += p.idle, y[4] += p.idle;
  if (NY == 9) y[0] += p.idle, y[4] += p.idle, y[8] += p.idle;
}
typedef float fb_f32x2 __attribute__((ext_vector_type(2)));
typedef float fb_f32x3 __attribute__((ext_vector_type(3)));
typedef float fb_f32x4 __attribute__((ext_vector_type(4)));
// (the loaded vectors are re-typed as WHOLE vectors: element-wise bit casts of the integer vector make this compiler keep only
// the first dword of the load)
template <int NY, int DIR>
__device__ __forceinline__ void load_frame(const LaneCtx &L, FramePtrs &p, FrameIn<NY> &f) {
  (void)L;
  if constexpr (NY == 6) {
    const fb_f32x4 a = __builtin_bit_cast(fb_f32x4, __builtin_amdgcn_raw_buffer_load_b128(p.y, p.oy, 0, 0));
    const fb_f32x2 b = __builtin_bit_cast(fb_f32x2, __builtin_amdgcn_raw_buffer_load_b64(p.y, p.oy + 16, 0, 0));
    f.y[0] = a[0], f.y[1] = a[1], f.y[2] = a[2], f.y[3] = a[3], f.y[4] = b[0], f.y[5] = b[1];
  } else if constexpr (NY == 9) {
    const fb_f32x4 a = __builtin_bit_cast(fb_f32x4, __builtin_amdgcn_raw_buffer_load_b128(p.y, p.oy, 0, 0));
    const fb_f32x4 b = __builtin_bit_cast(fb_f32x4, __builtin_amdgcn_raw_buffer_load_b128(p.y, p.oy + 16, 0, 0));
    const float c = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(p.y, p.oy + 32, 0, 0));
    f.y[0] = a[0], f.y[1] = a[1], f.y[2] = a[2], f.y[3] = a[3];
    f.y[4] = b[0], f.y[5] = b[1], f.y[6] = b[2], f.y[7] = b[3], f.y[8] = c;
  } else {
    static_assert(NY == 3, "frame inputs are 6-D rotations, 3x3 matrices or 3-D locations");
    const fb_f32x3 a = __builtin_bit_cast(fb_f32x3, __builtin_amdgcn_raw_buffer_load_b96(p.y, p.oy, 0, 0));
    f.y[0] = a[0], f.y[1] = a[1], f.y[2] = a[2];
  }
  const fb_f32x2 g2 = __builtin_bit_cast(fb_f32x2, __builtin_amdgcn_raw_buffer_load_b64(p.g2, p.o2, 0, 0));
  const fb_f32x3 g3 = __builtin_bit_cast(fb_f32x3, __builtin_amdgcn_raw_buffer_load_b96(p.g3, p.o3, 0, 0));
  f.g2[0] = g2[0], f.g2[1] = g2[1];
  f.g3[0] = g3[0], f.g3[1] = g3[1], f.g3[2] = g3[2];
  p.oy += DIR * p.sy, p.o2 += DIR * p.s2, p.o3 += DIR * p.s3;
}

__device__ __forceinline__ void store_m3(float *base, size_t idx, const M3 &a) {
  float *p = base + idx * 9;
#pragma unroll
  for (int i = 0; i < 9; ++i) p[i] = a.m[i];
}


__device__ __forceinline__ World world_at(const p2c_pose_head_desc &d, const LaneCtx &L, int t) {
  World W;
  W.on = (d.dloc != nullptr) || (d.drot != nullptr);
  W.rot = identity();
  W.loc = v3(0.f, 0.f, 0.f);
  if (W.on) {
    const int tt = t < d.T ? t : d.T - 1;
    if (d.world_absolute) world_step(d, L, tt, W);
    else
      for (int i = 0; i <= tt; ++i) world_step(d, L, i, W);
  }
  return W;
}

// inclusive scan over the frames of the clip: returns P_t = c_t c_{t-1} ... c_0 (planes: 2 x [frames][32][9] floats)
__device__ __forceinline__ M3 scan_time(M3 P, int t, int j, int T, float *planes, int plane_floats) {
  int cur = 0;
  for (int off = 1; off < T; off <<= 1) {
    float *mine = planes + cur * plane_floats + (t * GROUP + j) * 9;
#pragma unroll
    for (int i = 0; i < 9; ++i) mine[i] = P.m[i];
    __syncthreads();
    if (t >= off) {
      const float *q = planes + cur * plane_floats + ((t - off) * GROUP + j) * 9;
      M3 Q;
#pragma unroll
      for (int i = 0; i < 9; ++i) Q.m[i] = q[i];
      P = mul(P, Q);
    }
    cur ^= 1;
  }
  return P;
}

// Deferred loss finalize (p2c_pose_head_desc.defer_loss_finalize, time-parallel kernels: one partial per clip).
// Every workgroup of the backward needs the number of unmasked 2-D pairs: a sum of small integers held in floats -- exact
// in any order. Workgroup 0 also does what loss_finalize does (fp64 accumulators, fixed order) and publishes the losses.
__device__ __forceinline__ float n3_elems(const p2c_pose_head_desc &d) {
  return (float)((double)d.B * (double)(d.t1 - d.t0) * (double)d.n_common3d * 3.0);
}
__device__ __forceinline__ float deferred_count(const p2c_pose_head_desc &d, float *sh, int slot = 1) {   // sh: >= 16 floats
  float c = 0.f;
  for (int i = threadIdx.x; i < d.B; i += blockDim.x) c += d.partials[i * 4 + slot];
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) c += __shfl_xor(c, o, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = c;
  __syncthreads();
  float n2 = 0.f;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) n2 += sh[w];
  return n2;
}
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ void deferred_finalize(const p2c_pose_head_desc &d, double *sh) {   // sh: >= 48 doubles
  double a = 0.0, b = 0.0, c = 0.0;
  for (int i = threadIdx.x; i < d.B; i += blockDim.x) {
    a += (double)d.partials[i * 4 + 0];
    b += (double)d.partials[i * 4 + 1];
    c += (double)d.partials[i * 4 + 2];
  }
  a = wave_sum_f64(a), b = wave_sum_f64(b), c = wave_sum_f64(c);
  const int wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
  if ((threadIdx.x & 63) == 0) sh[wave] = a, sh[16 + wave] = b, sh[32 + wave] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s2 = 0.0, n2 = 0.0, s3 = 0.0;
    for (int w = 0; w < n_waves; ++w) s2 += sh[w], n2 += sh[16 + w], s3 += sh[32 + w];
    const float n3 = n3_elems(d);
    d.loss_sums[0] = (float)s2, d.loss_sums[1] = (float)n2, d.loss_sums[2] = (float)s3, d.loss_sums[3] = n3;
    const float nan = __builtin_nanf("");
    const float l2 = d.gt2d ? (float)(s2 / (2.0 * n2)) : nan, l3 = d.gt3d ? (float)(s3 / (double)n3) : nan;
    d.losses[0] = l2, d.losses[1] = l3, d.losses[2] = l2 + l3;
  }
}

}  // namespace p2c
