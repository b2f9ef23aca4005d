// p2c_pose_head_chain_dev.h -- device functions of the chain-lane pose head (a lane owns a chain of up to four consecutive
// bones, eight lanes own one (clip, frame) unit): shared by p2c_pose_head_chain.hip (a wavefront walks eight clips frame by
// frame) and p2c_train_stream.hip (a wavefront owns ONE clip: eight time segments of two frames). See p2c_pose_head_chain.hip
// for the mapping and the reference citations.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/p2c.h"
#include "p2c_pose_head_dev.h"

namespace p2c {
namespace chain {

constexpr int NS = 4;                    // bones per lane
constexpr int CLIPS = 8;                 // clips per wavefront
constexpr int Y_ROW = J * 6 * 4;         // bytes of one frame of y (6-D), gt2d (2 channels), gt3d
constexpr int G2_ROW = J * 2 * 4;
constexpr int G3_ROW = J * 3 * 4;
constexpr int N_DMA_Y = 5, N_DMA_2 = 2, N_DMA_3 = 4;      // LDS-DMA instructions per frame: 16-, 16-, 12-byte pieces
// (a 12-byte LDS-DMA piece lands at lane * 16 like a 16-byte one, its fourth dword untouched -- measured, tools/exp/dma12.hip:
// the gt3d image has one joint per 16 bytes)
// The last instruction of each group covers fewer than 64 pieces: its surplus lanes are switched off (EXEC), so the three
// images are exactly as long as their rows: 4992 + 1664 + 3328 = 9984 bytes per wavefront -- sixteen wavefronts per CU.
constexpr int LDS_Y = 0, LDS_G2 = CLIPS * Y_ROW, LDS_G3 = LDS_G2 + CLIPS * G2_ROW, LDS_WAVE = LDS_G3 + CLIPS * J * 16;
constexpr int OOB = 0x7fffff00;
constexpr int HIPS = 1, NECK = 8;        // HipsNeckExtractor(CARLA_SKELETON)

static __constant__ int c_start[8] = {0, 4, 8, 12, 16, 21, 20, 25};
static __constant__ int c_len[8] = {4, 4, 4, 4, 4, 4, 1, 1};

struct Lane {
  int lane, slot, chain, clip, start;
  bool clip_ok, trunk, head, on_hips, toe, leg;
  bool valid[NS];
};

__device__ __forceinline__ Lane make_lane(const p2c_pose_head_desc &d) {
  Lane L;
  L.lane = threadIdx.x & 63;
  L.slot = L.lane >> 3;
  L.chain = L.lane & 7;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  L.clip = wave * CLIPS + L.slot;
  L.clip_ok = L.clip < d.B;
  L.start = c_start[L.chain];
  L.trunk = L.chain == 0;
  L.head = L.chain == 2;
  L.leg = L.chain == 4 || L.chain == 5;
  L.toe = L.chain >= 6;
  L.on_hips = L.chain >= 4;                                  // legs and toe ends hang on the hips, the rest on spine01
#pragma unroll
  for (int k = 0; k < NS; ++k) L.valid[k] = L.clip_ok && k < c_len[L.chain];
  return L;
}

// ---- cross-lane ------------------------------------------------------------------------------------------------------
// DPP moves inside the 8 lanes of a clip: quad_perm [1,0,3,2] (lane ^ 1), quad_perm [2,3,0,1] (lane ^ 2),
// row_half_mirror (lane -> 7 - lane inside each half row)
template <int CTRL>
__device__ __forceinline__ float dppm(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
constexpr int DPP_X1 = 0xB1, DPP_X2 = 0x4E, DPP_HM = 0x141;
constexpr int DPP_Q0 = 0x00, DPP_Q2 = 0xAA, DPP_SHR2 = 0x112, DPP_SHR4 = 0x114, DPP_SHL2 = 0x102;   // quad_perm [0,0,0,0] / [2,2,2,2], row_shr / row_shl
// `src` moved by CTRL into the lanes of the second quad of every clip (lanes 4-7 and 12-15 of a row: bank_mask 0b1010); the
// other lanes keep `old`
template <int CTRL>
__device__ __forceinline__ float dpp_quad1(float old, float src) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(src), CTRL, 0xF, 0xA, false));
}
// lane (clip, 0)'s value of `for_upper` in lanes 0-3 of the clip and its value of `for_lower` in lanes 4-7: three VALU moves,
// no LDS round trip (a ds_bpermute exchange costs no VALU slot but ~100+ cycles of latency at two or three waves per SIMD)
__device__ __forceinline__ float from_trunk(float for_upper, float for_lower) {
  return dpp_quad1<DPP_SHR4>(dppm<DPP_Q0>(for_upper), dppm<DPP_Q0>(for_lower));
}
// lane (clip, SRC)'s value in all eight lanes of the clip (SRC = 0 or 2: a lane of the first quad)
template <int QUAD_BCAST>
__device__ __forceinline__ float first_quad_to_all(float v) {
  const float q = dppm<QUAD_BCAST>(v);
  return dpp_quad1<DPP_SHR4>(q, q);
}
__device__ __forceinline__ float quad_sum(float v) {
  v += dppm<DPP_X1>(v);
  v += dppm<DPP_X2>(v);
  return v;
}
__device__ __forceinline__ float clip_sum(float v) {          // sum over the 8 lanes of the clip, in every lane
  v = quad_sum(v);
  return v + dppm<DPP_HM>(v);
}
__device__ __forceinline__ float clip_min(float v) {
  v = fminf(v, dppm<DPP_X1>(v));
  v = fminf(v, dppm<DPP_X2>(v));
  return fminf(v, dppm<DPP_HM>(v));
}
__device__ __forceinline__ float clip_max(float v) {
  v = fmaxf(v, dppm<DPP_X1>(v));
  v = fmaxf(v, dppm<DPP_X2>(v));
  return fmaxf(v, dppm<DPP_HM>(v));
}
__device__ __forceinline__ M3 sel(bool c, const M3 &a, const M3 &b) {
  M3 r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.m[i] = c ? a.m[i] : b.m[i];
  return r;
}
__device__ __forceinline__ V3 sel(bool c, V3 a, V3 b) { return v3(c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z); }

// Wavefronts that share a SIMD start together and run the same program: left alone they reach their waits (LDS reads at the
// top of a frame, the staged rows) together and the SIMD idles. Each wave sleeps its slot number x P2C_CHAIN_STAGGER x 64
// cycles once, at the start (HW_REG_HW_ID bits 3:0 = wave slot on the SIMD): a fraction of a frame apart, and they stay apart.
#ifndef P2C_CHAIN_STAGGER
#define P2C_CHAIN_STAGGER 0
#endif
__device__ __forceinline__ void stagger() {
#if P2C_CHAIN_STAGGER > 0
  const int slot = __builtin_amdgcn_s_getreg((3 << 11) | 4) & 7;      // s_getreg_b32 hwreg(HW_REG_HW_ID, 0, 4)
  for (int i = 0; i < slot; ++i) __builtin_amdgcn_s_sleep(P2C_CHAIN_STAGGER);
#endif
}

struct FrameIn4 {
  float y[NS][6];
  float g2[NS][2];
  float g3[NS][3];
};
// ---- one frame -------------------------------------------------------------------------------------------------------
struct Acc {
  float sum2, cnt2, sum3;
};
struct FrameOut {      // backward only
  V3 F[NS];            // d total / d abs_loc of the lane's bones
};

// chain-local forward kinematics (p3d_pose.py:116-184 restricted to the lane's bones, the chain's first parent = identity)
__device__ __forceinline__ void fk_local(const Lane &L, const M3 (&R)[NS], const V3 (&l)[NS], M3 (&Al)[NS], V3 (&xl)[NS], M3 &Ap3) {
  Al[0] = R[0];
  xl[0] = l[0];
#pragma unroll
  for (int k = 1; k < NS; ++k) {
    // the second eye (bone 11) hangs on the head (bone 9 = step 1), like the first
    if (k == 3) Ap3 = sel(L.head, Al[1], Al[2]);
    const M3 &Ap = (k == 3) ? Ap3 : Al[k - 1];
    const V3 xp = (k == 3) ? sel(L.head, xl[1], xl[2]) : xl[k - 1];
    xl[k] = vmul(l[k], Ap) + xp;
    Al[k] = mul(R[k], Ap);
  }
}
// the transform the chain hangs on: hips / spine01 from the trunk lane; toe ends: their leg's end composed with the hips
__device__ __forceinline__ void fk_base(const Lane &L, const M3 (&Al)[NS], const V3 (&xl)[NS], M3 &BA, V3 &BX) {
  // upper chains (lanes 1-3) take the trunk's spine01 = step 3, legs and toe ends (lanes 4-7) its hips = step 1
#pragma unroll
  for (int i = 0; i < 9; ++i) BA.m[i] = from_trunk(Al[3].m[i], Al[1].m[i]);
  BX = v3(from_trunk(xl[3].x, xl[1].x), from_trunk(xl[3].y, xl[1].y), from_trunk(xl[3].z, xl[1].z));
  // a toe end hangs on the END of its leg (lane - 2, step 3), itself on the hips
  M3 EA;
#pragma unroll
  for (int i = 0; i < 9; ++i) EA.m[i] = dppm<DPP_SHR2>(Al[3].m[i]);
  const V3 EX = v3(dppm<DPP_SHR2>(xl[3].x), dppm<DPP_SHR2>(xl[3].y), dppm<DPP_SHR2>(xl[3].z));
  const M3 CA = mul(EA, BA);
  const V3 CX = vmul(EX, BA) + BX;
  BA = sel(L.toe, CA, BA);
  BX = sel(L.toe, CX, BX);
  BA = sel(L.trunk, identity(), BA);
  BX = sel(L.trunk, v3(0.f, 0.f, 0.f), BX);
}

// Projection (walker_control/p3d_pose_projection.py:115-152, identity world), normaliser (normalizer.py:20-41 + the
// extractors of transforms/pose/normalization/), loc_2d (loss/loc_2d.py:69-89, base_pose_loss.py:36-66) and loc_3d
// (loss/loc_3d.py:12-40) for the lane's four bones; with BWD also d total / d abs_loc. Mirrors frame_head() of
// p2c_pose_head_dev.h with the per-clip quantities computed once per lane.
template <bool BWD, bool SUMS = !BWD>       // SUMS: the loss sums are accumulated (forward; the train step's backward too)
__device__ __forceinline__ void head4(const p2c_pose_head_desc &d, const Lane &L, int t, const V3 (&x)[NS], const FrameIn4 &in,
                                      Acc &acc, float coef2, float coef3, V3 (&F)[NS]) {
  const bool in_slice = (t >= d.t0) && (t < d.t1);
  const bool has2 = d.gt2d != nullptr, has3 = d.gt3d != nullptr;
  const int tr = d.transform;
  // Pixel coordinates are kept CENTRED on the principal point (u = cx + up, v = cy + vp): every difference the normaliser
  // forms (point - shift, neck - hips, box extents) is then a difference of numbers of the size of the body in pixels, not of
  // ~400-pixel numbers -- in clips whose projected hips-neck distance is a fraction of a pixel the reference's own fp32
  // arithmetic loses 3-4 digits there (tests/test_pose_head_gpu.py: fp32 oracle vs fp64), this form does not. Absolute
  // coordinates appear only where the reference tests them against near_zero, and for transform = none.
  float up[NS], vp[NS], iz[NS];
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    const float Z = d.cam_dist - x[k].y;                // w = (x1, -x0, x2); Z = dist - w.x
    iz[k] = frcp(Z);
    const float fi = d.cam_f * iz[k];
    up[k] = x[k].x * fi;
    vp[k] = (x[k].z + d.cam_elev) * fi;
  }
  // ---- per-clip normaliser statistics ---------------------------------------------------------------------------------
  float su = 0.f, sv = 0.f, scale = 1.f, hu = 0.f, hv = 0.f, ku = 0.f, kv = 0.f, hn_scale = 1.f, bb_scale = 1.f;
  float minu = 0.f, maxu = 0.f, minv = 0.f, maxv = 0.f;
  bool use_bb = false, did_bb = false;
  bool missing[NS] = {false, false, false, false};
  if (tr != P2C_TRANSFORM_NONE) {
    if (tr != P2C_TRANSFORM_BBOX) {                     // hips_neck_extractor.py:6-13
      hu = first_quad_to_all<DPP_Q0>(up[1]), hv = first_quad_to_all<DPP_Q0>(vp[1]);      // hips: step 1 of lane 0
      ku = first_quad_to_all<DPP_Q2>(up[0]), kv = first_quad_to_all<DPP_Q2>(vp[0]);      // neck: step 0 of lane 2
      const float du = ku - hu, dv = kv - hv;
      hn_scale = fsqrt(fmaf(du, du, dv * dv));          // extractor.py:27-28
      su = hu, sv = hv, scale = hn_scale;
    }
    bool need_bb = (tr == P2C_TRANSFORM_BBOX);
    if (tr == P2C_TRANSFORM_HIPS_NECK_BBOX) {           // hips_neck_bbox_fallback_extractor.py:25,33
      const bool mh = (hu + d.cam_cx < d.near_zero) && (hv + d.cam_cy < d.near_zero);
      const bool mk = (ku + d.cam_cx < d.near_zero) && (kv + d.cam_cy < d.near_zero);
      use_bb = mh || mk;
      need_bb = use_bb;
    }
    if (__any(need_bb)) {                               // utils/tensors.py:12-26, bbox_extractor.py:6-18
      did_bb = true;
      const float inf = __builtin_inff();
      float a = inf, b = inf, c = -inf, e = -inf;
#pragma unroll
      for (int k = 0; k < NS; ++k) {
        missing[k] = !L.valid[k] || ((up[k] + d.cam_cx < d.near_zero) && (vp[k] + d.cam_cy < d.near_zero));
        a = fminf(a, missing[k] ? inf : up[k]), b = fminf(b, missing[k] ? inf : vp[k]);
        c = fmaxf(c, missing[k] ? -inf : up[k]), e = fmaxf(e, missing[k] ? -inf : vp[k]);
      }
      minu = clip_min(a), minv = clip_min(b), maxu = clip_max(c), maxv = clip_max(e);
      const float cu = 0.5f * (minu + maxu), cv = 0.5f * (minv + maxv);
      const float top_v = fminf(minv, maxv);
      const float dx = cu - cu, dy = top_v - cv;        // literal: inf - inf = nan when every joint is missing
      bb_scale = fsqrt(fmaf(dx, dx, dy * dy));
      if (tr == P2C_TRANSFORM_BBOX) {
        su = cu, sv = cv, scale = bb_scale;
      } else if (use_bb) {
        scale = bb_scale * 0.5748f;                     // :18,:34-38; the shift fallback (:26-31) is a no-op in the reference
      }
    }
  }
  const float inv_scale = (tr != P2C_TRANSFORM_NONE) ? frcp(scale) : 1.f;
  // ---- per bone: normalise, losses, and (BWD) the gradient wrt the normalised point ----------------------------------------
  float nu[NS], nv[NS], gu[NS], gv[NS];
  bool pass_u[NS], pass_v[NS];
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    nu[k] = up[k] + d.cam_cx, nv[k] = vp[k] + d.cam_cy;
    pass_u[k] = pass_v[k] = true;
    if (tr != P2C_TRANSFORM_NONE) {
      nu[k] = (up[k] - su) * inv_scale;                 // normalizer.py:24-25 (shift and point both centred)
      nv[k] = (vp[k] - sv) * inv_scale;
      const bool fu = isfinite(nu[k]), fv = isfinite(nv[k]);
      nu[k] = fu ? nu[k] : 0.f;                         // :30
      nv[k] = fv ? nv[k] : 0.f;
      const bool keep = nan_to_zero(iz[k]) >= d.near_zero;   // :35-37: the third channel (1/depth) acts as the confidence
      if (!keep) nu[k] = 0.f, nv[k] = 0.f;
      pass_u[k] = keep && fu, pass_v[k] = keep && fv;
    }
    float dnu = 0.f, dnv = 0.f;
    V3 gx = v3(0.f, 0.f, 0.f);
    if (in_slice && L.valid[k]) {
      if (has2) {
        const float g0 = in.g2[k][0], g1 = in.g2[k][1];
        const bool m = !d.mask_missing_joints || (L.start + k == d.hips_lane) || ((g0 != 0.f) && (g1 != 0.f));   // tensors.py:29-40
        if (m) {
          const float e0 = nu[k] - g0, e1 = nv[k] - g1;
          if (SUMS) acc.sum2 += fmaf(e0, e0, e1 * e1), acc.cnt2 += 1.f;
          if (BWD) dnu = coef2 * e0, dnv = coef2 * e1;
        }
      }
      if (has3) {
        const float e0 = x[k].x - in.g3[k][0], e1 = x[k].y - in.g3[k][1], e2 = x[k].z - in.g3[k][2];
        if (SUMS) acc.sum3 += fmaf(e0, e0, fmaf(e1, e1, e2 * e2));
        if (BWD) gx = v3(coef3 * e0, coef3 * e1, coef3 * e2);
      }
    }
    if (BWD) {
      F[k] = gx;
      gu[k] = dnu, gv[k] = dnv;
    }
  }
  if (!BWD) return;

  // ================================================ backward ============================================================
  if (tr != P2C_TRANSFORM_NONE) {
    const bool ok = isfinite(inv_scale) && (scale != 0.f);
    float a = 0.f, b = 0.f, c = 0.f;
#pragma unroll
    for (int k = 0; k < NS; ++k) {      // where(keep) and nan_to_num pass the gradient only through kept, finite entries
      gu[k] = (ok && pass_u[k]) ? gu[k] * inv_scale : 0.f;
      gv[k] = (ok && pass_v[k]) ? gv[k] * inv_scale : 0.f;
      a += gu[k], b += gv[k];
      c += fmaf(gu[k], nu[k], gv[k] * nv[k]);
    }
    const float Au = clip_sum(a), Av = clip_sum(b), Cs = clip_sum(c);     // -d/d shift, -d/d scale (n = (p - shift) / scale)
    const float g_scale = -Cs;
    float gsu = -Au, gsv = -Av, g_bbs = 0.f;
    if (tr == P2C_TRANSFORM_BBOX) {
      g_bbs = g_scale;
    } else if (use_bb) {
      g_bbs = g_scale * 0.5748f;
    } else {                            // scale = |neck - hips| (torch.linalg.norm backward; zero norm -> zero gradient)
      const float r = (hn_scale > 0.f) ? g_scale * frcp(hn_scale) : 0.f;
      const float gku = r * (ku - hu), gkv = r * (kv - hv);
      gsu -= gku, gsv -= gkv;
      gu[0] += L.head ? gku : 0.f, gv[0] += L.head ? gkv : 0.f;          // the neck is step 0 of the head lane
    }
    if (tr != P2C_TRANSFORM_BBOX) gu[1] += L.trunk ? gsu : 0.f, gv[1] += L.trunk ? gsv : 0.f;   // the hips: step 1 of the trunk
    if (did_bb && (tr == P2C_TRANSFORM_BBOX || __any(use_bb))) {
      // min / max pick the first joint holding the extreme value (torch.min/max(dim) backward)
      float g_minu = 0.f, g_maxu = 0.f, g_minv = 0.f, g_maxv = 0.f;
      if (tr == P2C_TRANSFORM_BBOX) g_minu += 0.5f * gsu, g_maxu += 0.5f * gsu, g_minv += 0.5f * gsv, g_maxv += 0.5f * gsv;
      if (tr == P2C_TRANSFORM_BBOX || use_bb) {
        const float dy = fminf(minv, maxv) - 0.5f * (minv + maxv);
        const float g_dy = (bb_scale > 0.f) ? g_bbs * dy * frcp(bb_scale) : 0.f;
        g_minv += 0.5f * g_dy;          // top_v = minv (+g_dy), centre (-g_dy / 2 each)
        g_maxv -= 0.5f * g_dy;
      }
      auto route = [&](const float (&val)[NS], float extreme, float g, float (&dst)[NS]) {
        float first = 99.f;
#pragma unroll
        for (int k = 0; k < NS; ++k) first = fminf(first, (!missing[k] && val[k] == extreme) ? (float)(L.start + k) : 99.f);
        first = clip_min(first);
#pragma unroll
        for (int k = 0; k < NS; ++k) dst[k] += (L.valid[k] && (float)(L.start + k) == first) ? g : 0.f;
      };
      route(up, minu, g_minu, gu);
      route(up, maxu, g_maxu, gu);
      route(vp, minv, g_minv, gv);
      route(vp, maxv, g_maxv, gv);
    }
  }
  // projection backward: u = cx + f x0 / Z, v = cy + f (x2 + elev) / Z, Z = dist - x1
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    const float fz = d.cam_f * iz[k];
    const float gy1 = fz * iz[k] * fmaf(x[k].x, gu[k], (x[k].z + d.cam_elev) * gv[k]);
    if (L.valid[k]) F[k] = F[k] + v3(fz * gu[k], gy1, fz * gv[k]);
  }
}


// Subtree sums of a per-bone quantity (children -> parent accumulation of the FK gradients): suffix sums inside the chain, the
// chain totals to the trunk lane (two quad sums), the toe ends to their legs.
__device__ __forceinline__ void subtree4(const Lane &L, const V3 (&f)[NS], V3 (&sub)[NS]) {
  // the toe end of a leg (zero elsewhere) hangs on the leg's last bone: it joins every suffix sum of the leg
  const V3 toe = sel(L.leg, v3(dppm<DPP_SHL2>(f[0].x), dppm<DPP_SHL2>(f[0].y), dppm<DPP_SHL2>(f[0].z)), v3(0.f, 0.f, 0.f));   // lane + 2
  sub[3] = f[3] + toe;
  const V3 t23 = f[2] + sub[3];
  sub[2] = sel(L.head, f[2], t23);               // the two eyes are siblings
  sub[1] = f[1] + t23;
  sub[0] = f[0] + sub[1];
  // trunk: spine01 / spine carry the arms and the head, hips / root also the legs (their totals include the toe ends)
  const V3 others = sel(L.trunk || L.toe, v3(0.f, 0.f, 0.f), sub[0]);
  const V3 q = v3(quad_sum(others.x), quad_sum(others.y), quad_sum(others.z));          // lanes 0-3: upper chains; 4-7: legs + toes
  const V3 legs = v3(dppm<DPP_HM>(q.x), dppm<DPP_HM>(q.y), dppm<DPP_HM>(q.z));          // lane 0 reads lane 7's quad sum
  const V3 up = sel(L.trunk, q, v3(0.f, 0.f, 0.f)), all = sel(L.trunk, q + legs, v3(0.f, 0.f, 0.f));
  sub[3] = sub[3] + up, sub[2] = sub[2] + up;
  sub[1] = sub[1] + all, sub[0] = sub[0] + all;
}
}  // namespace chain
}  // namespace p2c
