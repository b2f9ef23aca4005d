// p2c_pose_head.hip -- fused pose head (forward + recompute-backward) for gfx950 / CDNA4.
//
// Work decomposition ("joint-lane" mapping):
//   * one 64-lane wavefront owns TWO clips; each clip gets a 32-lane group, lanes 0..25 = the 26 CARLA bones in
//     DFS order, lanes 26..31 carry the identity transform (they are the "no ancestor" target of the tree walks).
//   * every lane walks its joint through the T frames of the clip sequentially, so the cumulative rotation
//     rel_rot[t] = change[t] @ rel_rot[t-1] (reference projection.py:190-193) lives in 9 registers and costs one 3x3
//     product per frame; the global loads of y[b,t,:,:] are 26 lanes x 24 B = 624 contiguous bytes per clip-frame.
//   * forward kinematics over the bone tree is a parallel tree-prefix "product" of affine maps (R, l) with
//     (R1,l1)o(R2,l2) = (R1 R2, l1 R2 + l2): three pointer-doubling rounds (depth of the skeleton <= 8) of
//     ds_bpermute exchanges -- no LDS storage, no barrier, any T.
//   * hips / neck (and bbox) statistics of the normaliser and the loss sums are wave-level reductions.
//   * backward recomputes the forward per frame while running the frames in REVERSE; rel_rot[t-1] = change[t]^T
//     rel_rot[t] (changes are rotations), children->parent accumulation of FK gradients becomes two prefix sums over
//     the DFS-ordered lanes (subtree(j) is the contiguous lane range [j, end(j)]).
//
// Reference semantics restated here are cited per function (paths relative to src/pedestrians_video_2_carla/).
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <stdint.h>
#include <math.h>

#include "../../include/p2c.h"

#include "p2c_pose_head_dev.h"

namespace p2c {

// =====================================================================================================================
// forward, rotation kinds (pose_changes / relative_rot; projection.py:144-195)
// =====================================================================================================================
template <int KIND, bool MAT>
__global__ __launch_bounds__(256) void pose_head_rot_fwd(const p2c_pose_head_desc d) {
  using K = KindTraits<KIND>;
  const LaneCtx L = make_lane(d);
  const int T = d.T;

  V3 l = v3(0.f, 0.f, 0.f);  // reference relative location of this joint
  M3 R = identity();         // running relative rotation
  if (L.active) {
    int st = d.skel_type[L.clip];
    const float *pl = d.ref_rel_loc + ((size_t)st * J + L.j) * 3;
    l = v3(pl[0], pl[1], pl[2]);
    if (K::SCAN) {
      const float *pr = d.ref_rel_rot + ((size_t)st * J + L.j) * 9;
#pragma unroll
      for (int i = 0; i < 9; ++i) R.m[i] = pr[i];
    }
  }
  World W;
  W.on = (d.dloc != nullptr) || (d.drot != nullptr);
  W.rot = identity();
  W.loc = v3(0.f, 0.f, 0.f);
  HeadAcc acc{0.f, 0.f, 0.f};
  float rot_acc = 0.f;

  // two register sets for the frame inputs, used alternately: the loads of frame t+1 are in flight while frame t is
  // computed, and no register-to-register copies are needed at the end of an iteration
  FrameIn<K::NY> fa, fb;
  FramePtrs ptrs = frame_ptrs<K::NY>(d, L, 0);
  auto step = [&](const FrameIn<K::NY> &cur, int t) {
    const size_t jf = ((size_t)L.clip * T + t) * J + L.j;
    M3 c;
    float yin[K::NY];
    rotation_input<K::NY>(cur, ptrs, yin);
    if (K::SIXD) {
      SixD s;
      c = rot6d_fwd(yin, s);
    } else {
#pragma unroll
      for (int i = 0; i < 9; ++i) c.m[i] = yin[i];
    }
    R = K::SCAN ? mul(c, R) : c;  // p3d_pose.py:98-114
    if (MAT && L.active) {
      if (d.out_pose_changes && K::SCAN) store_m3(d.out_pose_changes, jf, c);
      if (d.out_relative_pose_rot) store_m3(d.out_relative_pose_rot, jf, R);
      if (d.out_relative_pose_loc) {
        float *o = d.out_relative_pose_loc + jf * 3;
        o[0] = l.x, o[1] = l.y, o[2] = l.z;
      }
    }
    M3 A = R;
    V3 x = l;
    if (d.gt_rot) {                      // rot_3d fused: the complete absolute rotation is needed also on the lean path
      fk_doubling<true>(L, A, x);
      rot_acc += rot_loss_term(d, L, t, A);
    } else {
      fk_doubling<MAT>(L, A, x);
    }
    if (MAT && L.active && d.out_absolute_pose_rot) store_m3(d.out_absolute_pose_rot, jf, A);
    world_step(d, L, t, W);
    if (MAT && W.on) world_store(d, L, t, W);
    frame_head<MAT ? MODE_FWD_MATERIALIZE : MODE_FWD>(d, L, t, x, W, acc, 0.f, 0.f, nullptr, nullptr, cur.g2, cur.g3);
  };
  // (the prefetch is unconditional: past the last frame a buffer load reads the next clip's rows or, out of range, zeros --
  // never used; a guard around it is a join at which the wait for THIS frame's loads would also cover the prefetch)
  load_frame<K::NY, 1>(L, ptrs, fa);
  for (int t = 0; t < T; t += 2) {
    load_frame<K::NY, 1>(L, ptrs, fb);
    step(fa, t);
    if (t + 1 < T) {
      load_frame<K::NY, 1>(L, ptrs, fa);
      step(fb, t + 1);
    }
  }
  if (L.active && K::SCAN && d.final_rel_rot) store_m3(d.final_rel_rot, (size_t)L.clip * J + L.j, R);

  float s2 = wave_sum(acc.sum2), c2 = wave_sum(acc.cnt2), s3 = wave_sum(acc.sum3);
  const float sr = d.gt_rot ? wave_sum(rot_acc) : 0.f;
  if (L.lane == 0) {
    size_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    float *p = d.partials + wave * 4;
    p[0] = s2, p[1] = c2, p[2] = s3, p[3] = sr;
  }
}

// =====================================================================================================================
// backward, rotation kinds: frames in reverse, forward recomputed per frame
// =====================================================================================================================
template <int KIND>
__global__ __launch_bounds__(256) void pose_head_rot_bwd(const p2c_pose_head_desc d, const GradLosses grad_losses,
                                                         const float *g_abs_ext, const float *g_projt_ext,
                                                         float *grad_y) {
  using K = KindTraits<KIND>;
  const LaneCtx L = make_lane(d);
  const int T = d.T;

  V3 l = v3(0.f, 0.f, 0.f);
  M3 Rref = identity(), R = identity();
  if (L.active) {
    int st = d.skel_type[L.clip];
    const float *pl = d.ref_rel_loc + ((size_t)st * J + L.j) * 3;
    l = v3(pl[0], pl[1], pl[2]);
    if (K::SCAN) {
      const float *pr = d.ref_rel_rot + ((size_t)st * J + L.j) * 9;
      const float *pf = d.final_rel_rot + ((size_t)L.clip * J + L.j) * 9;
#pragma unroll
      for (int i = 0; i < 9; ++i) Rref.m[i] = pr[i], R.m[i] = pf[i];
    }
  }
  float coef2 = 0.f, coef3 = 0.f;
  loss_coefs(d, grad_losses, coef2, coef3);

  World W;
  W.on = (d.dloc != nullptr) || (d.drot != nullptr);
  W.rot = identity();
  W.loc = v3(0.f, 0.f, 0.f);
  if (W.on) {  // world state at the last frame
    if (d.world_absolute) world_step(d, L, T - 1, W);
    else
      for (int t = 0; t < T; ++t) world_step(d, L, t, W);
  }

  HeadAcc acc{0.f, 0.f, 0.f};
  M3 carry = zero3();  // change[t+1]^T @ dL/d rel_rot[t+1]

  FrameIn<K::NY> cur, nxt;
  FramePtrs ptrs = frame_ptrs<K::NY>(d, L, T - 1);
  load_frame<K::NY, -1>(L, ptrs, cur);
  for (int t = T - 1; t >= 0; --t) {
    const size_t jf = ((size_t)L.clip * T + t) * J + L.j;
    load_frame<K::NY, -1>(L, ptrs, nxt);       // (unconditional: before frame 0 the offsets wrap out of range -> zeros, unused)
    M3 c;
    SixD s;
    float yin[K::NY];
    rotation_input<K::NY>(cur, ptrs, yin);
    if (K::SIXD) {
      c = rot6d_fwd(yin, s);
    } else {
#pragma unroll
      for (int i = 0; i < 9; ++i) c.m[i] = yin[i];
    }
    if (!K::SCAN) R = c;
    // ---- forward of this frame ----
    M3 A = R;
    V3 x = l;
    fk_doubling(L, A, x);
    V3 F = frame_head<MODE_BWD>(d, L, t, x, W, acc, coef2, coef3, g_abs_ext, g_projt_ext, cur.g2, cur.g3);
    // ---- FK backward: subtree sums via prefix sums over the DFS-ordered lanes ----
    // SubF[j] = sum of F over subtree(j);  Z[j] = sum over strict descendants m of r_m^T (x) SubF[m],
    // r_m = x_m - x_parent(m);  dL/dA_j = A_j Z_j ;  dL/d rel_rot_j = dL/dA_j @ A_parent^T
    V3 P = v3(group_prefix(F.x), group_prefix(F.y), group_prefix(F.z));
    V3 Pe = shfl(P, L.base + L.sub_end);
    V3 SubF = Pe - (P - F);
    V3 xp = shfl(x, L.base + L.anc0);
    M3 Ap = shfl(A, L.base + L.anc0);
    V3 r = x - xp;
    M3 Y = M3{{r.x * SubF.x, r.x * SubF.y, r.x * SubF.z, r.y * SubF.x, r.y * SubF.y, r.y * SubF.z, r.z * SubF.x,
               r.z * SubF.y, r.z * SubF.z}};
    M3 Zm;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      float py = group_prefix(Y.m[i]);
      Zm.m[i] = shfl(py, L.base + L.sub_end) - py;
    }
    M3 GR = mulNT(mul(A, Zm), Ap);
    // ---- reverse scan over time ----
    M3 gc;
    if (K::SCAN) {
      GR = add(GR, carry);
      M3 Rprev = (t > 0) ? mulTN(c, R) : Rref;  // change is a rotation: rel_rot[t-1] = change^T rel_rot[t]
      gc = mulNT(GR, Rprev);
      carry = mulTN(c, GR);
      R = Rprev;
    } else {
      gc = GR;
    }
    if (L.active) {
      float *g = grad_y + jf * K::NY;
      if (K::SIXD) {
        float gy6[6];
        rot6d_bwd(s, gc, gy6);
        float2 *q = reinterpret_cast<float2 *>(g);
        q[0] = make_float2(gy6[0], gy6[1]), q[1] = make_float2(gy6[2], gy6[3]), q[2] = make_float2(gy6[4], gy6[5]);
      } else {
#pragma unroll
        for (int i = 0; i < 9; ++i) g[i] = gc.m[i];
      }
    }
    // ---- world state of the previous frame (inverse step; drot is a rotation) ----
    if (W.on && t > 0 && d.world_absolute) {
      world_step(d, L, t - 1, W);
    } else if (W.on && t > 0 && L.clip < d.B) {
      size_t frame = (size_t)L.clip * T + t;
      if (d.drot) {
        M3 dr;
#pragma unroll
        for (int i = 0; i < 9; ++i) dr.m[i] = d.drot[frame * 9 + i];
        W.rot = mulNT(W.rot, dr);
      }
      if (d.dloc) W.loc = W.loc - v3(d.dloc[frame * 3 + 0], d.dloc[frame * 3 + 1], d.dloc[frame * 3 + 2]);
    }
    cur = nxt;
  }
}

// =====================================================================================================================
// backward for the 6-D kinds in the tangent space of SO(3) ("rigid body" form; derivation + fp64 check against autograd
// in tools/proto_bwd_math.py). With F_m = dL/d abs_loc_m:
//   world torque about joint j      tau_j  = sum_{m in subtree(j)} F_m x (x_m - x_j) = Sub(F x x)_j - Sub(F)_j x x_j
//   in the right-tangent of rel_rot tau'_j = tau_j A_parent^T = tau_j A_j^T R_j            (A_j = R_j A_parent)
//   rel_rot[t] = change[t] rel_rot[t-1]: a right-perturbation eps of change[s] turns every rel_rot[t], t >= s, by the
//   SAME vector eps rel_rot[s-1]  ->  g_s = (sum_{t>=s} tau'_t) rel_rot[s-1]^T : the time scan is a plain suffix sum
//   6-D pull-back of the tangent gradient g (G = 1/2 rows(c) x g), closed form in the Gram-Schmidt basis (b1,b2,b3):
//   d/da1 = ((g.b2 + (g.b1) d/n2) b3 - (g.b3) b2) / n1 ,  d/da2 = -(g.b1)/n2 b3
// Two prefix sums of 3-vectors replace the twelve of the matrix form; nothing 3x3 is carried between frames.
// =====================================================================================================================
#ifndef P2C_BWD_WAVES
#define P2C_BWD_WAVES 1
#endif
template <int KIND>
__global__ __launch_bounds__(256, P2C_BWD_WAVES) void pose_head_rot_bwd_tangent(const p2c_pose_head_desc d, const GradLosses grad_losses,
                                                                 const float *g_abs_ext, const float *g_projt_ext,
                                                                 const float *g_rot_ext, float *grad_y) {
  using K = KindTraits<KIND>;
  static_assert(K::SIXD, "tangent-space backward is for the 6-D kinds");
  const LaneCtx L = make_lane(d);
  const int T = d.T;

  V3 l = v3(0.f, 0.f, 0.f);
  M3 Rref = identity(), R = identity();
  if (L.active) {
    int st = d.skel_type[L.clip];
    const float *pl = d.ref_rel_loc + ((size_t)st * J + L.j) * 3;
    l = v3(pl[0], pl[1], pl[2]);
    if (K::SCAN) {
      const float *pr = d.ref_rel_rot + ((size_t)st * J + L.j) * 9;
      const float *pf = d.final_rel_rot + ((size_t)L.clip * J + L.j) * 9;
#pragma unroll
      for (int i = 0; i < 9; ++i) Rref.m[i] = pr[i], R.m[i] = pf[i];
    }
  }
  float coef2 = 0.f, coef3 = 0.f;
  loss_coefs(d, grad_losses, coef2, coef3);
  const float coefr = rot_coef(d);

  World W;
  W.on = (d.dloc != nullptr) || (d.drot != nullptr);
  W.rot = identity();
  W.loc = v3(0.f, 0.f, 0.f);
  if (W.on) {
    if (d.world_absolute) world_step(d, L, T - 1, W);
    else
      for (int t = 0; t < T; ++t) world_step(d, L, t, W);
  }

  HeadAcc acc{0.f, 0.f, 0.f};
  V3 S = v3(0.f, 0.f, 0.f);  // suffix sum over time of the parent-frame torques

  FrameIn<6> cur, nxt;
  FramePtrs ptrs = frame_ptrs<6>(d, L, T - 1);
  load_frame<6, -1>(L, ptrs, cur);
  float *gy = L.active ? grad_y + (((size_t)L.clip * T + (T - 1)) * J + L.j) * 6 : nullptr;
  for (int t = T - 1; t >= 0; --t) {
    load_frame<6, -1>(L, ptrs, nxt);           // (unconditional: before frame 0 the offsets wrap out of range -> zeros, unused)
    SixD s;
    float yin[6];
    rotation_input<6>(cur, ptrs, yin);
    M3 c = rot6d_fwd(yin, s);
    if (!K::SCAN) R = c;
    M3 A = R;
    V3 x = l;
    fk_doubling(L, A, x);
    V3 F = frame_head<MODE_BWD>(d, L, t, x, W, acc, coef2, coef3, g_abs_ext, g_projt_ext, cur.g2, cur.g3);
    // subtree sums of F and F x x through prefix sums over the DFS-ordered lanes
    V3 FX = cross(F, x);
    V3 PF = v3(group_prefix(F.x), group_prefix(F.y), group_prefix(F.z));
    V3 PX = v3(group_prefix(FX.x), group_prefix(FX.y), group_prefix(FX.z));
    V3 SubF = shfl(PF, L.base + L.sub_end) - (PF - F);
    V3 SubX = shfl(PX, L.base + L.sub_end) - (PX - FX);
    V3 tau = SubX - cross(SubF, x);
    if (g_rot_ext || coefr != 0.f) {
      V3 tm = g_rot_ext ? rotation_torque(g_rot_ext, ((size_t)L.clip * T + t) * J + L.j, A, L.active) : v3(0.f, 0.f, 0.f);
      if (coefr != 0.f) tm = tm + rot_loss_torque(d, L, t, A, coefr);          // rot_3d fused: G = coef (A - gt)
      const V3 PT = v3(group_prefix(tm.x), group_prefix(tm.y), group_prefix(tm.z));
      tau = tau + (shfl(PT, L.base + L.sub_end) - (PT - tm));
    }
    V3 taup = vmul(vmulT(tau, A), R);  // tau A^T R
    V3 g;
    if (K::SCAN) {
      S = S + taup;
      M3 Rprev = (t > 0) ? mulTN(c, R) : Rref;  // change is a rotation: rel_rot[t-1] = change^T rel_rot[t]
      g = vmulT(S, Rprev);
      R = Rprev;
    } else {
      g = taup;
    }
    if (L.active) {
      float gy6[6];
      if (s.c1 && s.c2) {
        V3 b3 = v3(c.m[6], c.m[7], c.m[8]);
        float al = dot(g, s.b1), be = dot(g, s.b2), ga = dot(g, b3);
        float r1 = frcp(s.n1), r2 = frcp(s.n2);
        float k3 = (be + al * s.d * r2) * r1, k2 = -ga * r1, k5 = -al * r2;
        gy6[0] = fmaf(k3, b3.x, k2 * s.b2.x), gy6[1] = fmaf(k3, b3.y, k2 * s.b2.y), gy6[2] = fmaf(k3, b3.z, k2 * s.b2.z);
        gy6[3] = k5 * b3.x, gy6[4] = k5 * b3.y, gy6[5] = k5 * b3.z;
      } else {  // a norm sits on the 1e-12 clamp: generic chain rule through the Gram-Schmidt steps
        M3 G;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          V3 ci = v3(c.m[i * 3], c.m[i * 3 + 1], c.m[i * 3 + 2]);
          V3 h = cross(ci, g) * 0.5f;
          G.m[i * 3] = h.x, G.m[i * 3 + 1] = h.y, G.m[i * 3 + 2] = h.z;
        }
        rot6d_bwd(s, G, gy6);
      }
      float2 *q = reinterpret_cast<float2 *>(gy);
      q[0] = make_float2(gy6[0], gy6[1]), q[1] = make_float2(gy6[2], gy6[3]), q[2] = make_float2(gy6[4], gy6[5]);
      gy -= J * 6;
    }
    if (W.on && t > 0 && d.world_absolute) {
      world_step(d, L, t - 1, W);
    } else if (W.on && t > 0 && L.clip < d.B) {
      size_t frame = (size_t)L.clip * T + t;
      if (d.drot) {
        M3 dr;
#pragma unroll
        for (int i = 0; i < 9; ++i) dr.m[i] = d.drot[frame * 9 + i];
        W.rot = mulNT(W.rot, dr);
      }
      if (d.dloc) W.loc = W.loc - v3(d.dloc[frame * 3 + 0], d.dloc[frame * 3 + 1], d.dloc[frame * 3 + 2]);
    }
    cur = nxt;
  }
}

// =====================================================================================================================
// time-parallel variants for small batches (6-D kinds, lean outputs)
// ---------------------------------------------------------------------------------------------------------------------
// The clip-sequential kernels above give a whole clip to half a wavefront: with B = 256 that is 128 wavefronts on a
// 1024-SIMD chip, each walking T frames one after the other. Here a workgroup takes ONE clip and every 32-lane group one
// frame; the only coupling between frames -- rel_rot[t] = change[t] @ rel_rot[t-1] (p3d_pose.py:98-114) -- becomes a
// log2(T)-round inclusive scan of 3x3 products through LDS, and the backward's suffix sum of torques a second pass
// through LDS. B * T / 2 wavefronts, each doing one frame's work.
// =====================================================================================================================

template <int KIND>
__global__ __launch_bounds__(1024) void pose_head_rot_fwd_tp(const p2c_pose_head_desc d) {
  using K = KindTraits<KIND>;
  extern __shared__ float tp_lds[];
  int t;
  const LaneCtx L = make_lane_tp(d, t);
  const int T = d.T, n_waves = blockDim.x >> 6, wave = threadIdx.x >> 6;
  const int plane = n_waves * 2 * GROUP * 9;

  V3 l = v3(0.f, 0.f, 0.f);
  M3 Rref = identity();
  if (L.j < J) {
    int st = d.skel_type[L.clip];
    const float *pl = d.ref_rel_loc + ((size_t)st * J + L.j) * 3;
    l = v3(pl[0], pl[1], pl[2]);
    if (K::SCAN) {
      const float *pr = d.ref_rel_rot + ((size_t)st * J + L.j) * 9;
#pragma unroll
      for (int i = 0; i < 9; ++i) Rref.m[i] = pr[i];
    }
  }
  FrameIn<K::NY> cur;
  FramePtrs ptrs = frame_ptrs<K::NY>(d, L, t);
  load_frame<K::NY, 1>(L, ptrs, cur);
  M3 c;
  float yin[K::NY];
  rotation_input<K::NY>(cur, ptrs, yin);
  if (K::SIXD) {
    SixD s;
    c = rot6d_fwd(yin, s);
  } else {
#pragma unroll
    for (int i = 0; i < 9; ++i) c.m[i] = yin[i];
  }
  M3 R = c;
  if (K::SCAN) R = mul(scan_time(c, t, L.j, T, tp_lds, plane), Rref);
  if (L.active && K::SCAN && d.final_rel_rot && t == T - 1) store_m3(d.final_rel_rot, (size_t)L.clip * J + L.j, R);
  M3 A = R;
  V3 x = l;
  fk_doubling(L, A, x);
  const World W = world_at(d, L, t);
  HeadAcc acc{0.f, 0.f, 0.f};
  frame_head<MODE_FWD>(d, L, t, x, W, acc, 0.f, 0.f, nullptr, nullptr, cur.g2, cur.g3);

  // one partial per clip, waves added in frame order
  float s2 = wave_sum(acc.sum2), c2 = wave_sum(acc.cnt2), s3 = wave_sum(acc.sum3);
  const float sr = d.gt_rot ? wave_sum(rot_loss_term(d, L, t, A)) : 0.f;       // rot_3d fused
  float *red = tp_lds + 2 * plane;
  __syncthreads();
  if (L.lane == 0) red[wave * 4 + 0] = s2, red[wave * 4 + 1] = c2, red[wave * 4 + 2] = s3, red[wave * 4 + 3] = sr;
  __syncthreads();
  if (threadIdx.x == 0) {
    float a = 0.f, b = 0.f, cc = 0.f, rr = 0.f;
    for (int w = 0; w < n_waves; ++w) a += red[w * 4], b += red[w * 4 + 1], cc += red[w * 4 + 2], rr += red[w * 4 + 3];
    float *p = d.partials + (size_t)L.clip * 4;
    p[0] = a, p[1] = b, p[2] = cc, p[3] = rr;
  }
}

// defer_loss_finalize == 2 ("train"): the forward call does not run the pose head at all -- the backward kernel recomputes
// it anyway -- but only counts, per clip, the gt pairs the 2-D loss will not mask (a property of the targets alone:
// utils/tensors.py:29-40) into slot 3 of the clip's partial; the backward kernel then computes losses AND gradients.
__global__ __launch_bounds__(1024) void pose_head_count_tp(const p2c_pose_head_desc d) {
  __shared__ float sh[16];
  int t;
  const LaneCtx L = make_lane_tp(d, t);
  float c = 0.f;
  if (t >= d.t0 && t < d.t1 && L.has2) {
    const float *g = d.gt2d + (((size_t)L.clip * d.T + t) * d.gt2d_joints + L.gm2) * d.gt2d_channels;
    c = (!d.mask_missing_joints || L.never_masked || ((g[0] != 0.f) && (g[1] != 0.f))) ? 1.f : 0.f;
  }
  c = wave_sum(c);
  if (L.lane == 0) sh[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    float n = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) n += sh[w];
    d.partials[(size_t)L.clip * 4 + 3] = n;
  }
}


template <int KIND, bool TRAIN = false>
__global__ __launch_bounds__(1024) void pose_head_rot_bwd_tangent_tp(const p2c_pose_head_desc d, const GradLosses grad_losses,
                                                                     const float *g_abs_ext, const float *g_projt_ext,
                                                                     const float *g_rot_ext, float *grad_y) {
  using K = KindTraits<KIND>;
  static_assert(K::SIXD, "tangent-space backward is for the 6-D kinds");
  extern __shared__ float tp_lds[];
  int t;
  const LaneCtx L = make_lane_tp(d, t);
  const int T = d.T, n_waves = blockDim.x >> 6;
  const int plane = n_waves * 2 * GROUP * 9;

  V3 l = v3(0.f, 0.f, 0.f);
  M3 Rref = identity();
  if (L.j < J) {
    int st = d.skel_type[L.clip];
    const float *pl = d.ref_rel_loc + ((size_t)st * J + L.j) * 3;
    l = v3(pl[0], pl[1], pl[2]);
    if (K::SCAN) {
      const float *pr = d.ref_rel_rot + ((size_t)st * J + L.j) * 9;
#pragma unroll
      for (int i = 0; i < 9; ++i) Rref.m[i] = pr[i];
    }
  }
  float coef2 = 0.f, coef3 = 0.f;
  FrameIn<6> cur;
  FramePtrs ptrs = frame_ptrs<6>(d, L, t);
  load_frame<6, 1>(L, ptrs, cur);
  __shared__ float cnt_sh[16];
  if (TRAIN) {                          // the forward only counted the unmasked pairs (slot 3 of every clip's partial)
    loss_coefs_n(d, grad_losses, deferred_count(d, cnt_sh, 3), n3_elems(d), coef2, coef3);
  } else if (d.defer_loss_finalize) {   // the forward skipped loss_finalize: count from the per-clip partials
    __shared__ double fin_sh[48];
    loss_coefs_n(d, grad_losses, deferred_count(d, cnt_sh), n3_elems(d), coef2, coef3);
    if (blockIdx.x == 0) deferred_finalize(d, fin_sh);
  } else {
    loss_coefs(d, grad_losses, coef2, coef3);
  }
  SixD s;
  float yin6[6];
  rotation_input<6>(cur, ptrs, yin6);
  const M3 c = rot6d_fwd(yin6, s);
  M3 R = c;
  if (K::SCAN) R = mul(scan_time(c, t, L.j, T, tp_lds, plane), Rref);
  M3 A = R;
  V3 x = l;
  fk_doubling(L, A, x);
  const World W = world_at(d, L, t);
  HeadAcc acc{0.f, 0.f, 0.f};
  V3 F = frame_head<TRAIN ? MODE_TRAIN : MODE_BWD>(d, L, t, x, W, acc, coef2, coef3, g_abs_ext, g_projt_ext, cur.g2, cur.g3);
  if (TRAIN) {                          // this clip's loss sums (same reduction as pose_head_rot_fwd_tp)
    const float s2 = wave_sum(acc.sum2), c2 = wave_sum(acc.cnt2), s3 = wave_sum(acc.sum3);
    __shared__ float red_sh[16 * 3];
    if (L.lane == 0) red_sh[(threadIdx.x >> 6) * 3 + 0] = s2, red_sh[(threadIdx.x >> 6) * 3 + 1] = c2, red_sh[(threadIdx.x >> 6) * 3 + 2] = s3;
    __syncthreads();
    if (threadIdx.x == 0) {
      float a = 0.f, b = 0.f, cc = 0.f;
      for (int w = 0; w < n_waves; ++w) a += red_sh[w * 3], b += red_sh[w * 3 + 1], cc += red_sh[w * 3 + 2];
      float *p = d.partials + (size_t)L.clip * 4;
      p[0] = a, p[1] = b, p[2] = cc;
    }
  }
  // subtree sums of F and F x x through prefix sums over the DFS-ordered lanes
  V3 FX = cross(F, x);
  V3 PF = v3(group_prefix(F.x), group_prefix(F.y), group_prefix(F.z));
  V3 PX = v3(group_prefix(FX.x), group_prefix(FX.y), group_prefix(FX.z));
  V3 SubF = shfl(PF, L.base + L.sub_end) - (PF - F);
  V3 SubX = shfl(PX, L.base + L.sub_end) - (PX - FX);
  V3 tau = SubX - cross(SubF, x);
  const float coefr = TRAIN ? 0.f : rot_coef(d);
  if (g_rot_ext || coefr != 0.f) {
    V3 tm = g_rot_ext ? rotation_torque(g_rot_ext, ((size_t)L.clip * T + (t < T ? t : 0)) * J + L.j, A, L.active) : v3(0.f, 0.f, 0.f);
    if (coefr != 0.f) tm = tm + rot_loss_torque(d, L, t, A, coefr);            // rot_3d fused: G = coef (A - gt)
    const V3 PT = v3(group_prefix(tm.x), group_prefix(tm.y), group_prefix(tm.z));
    tau = tau + (shfl(PT, L.base + L.sub_end) - (PT - tm));
  }
  V3 taup = vmul(vmulT(tau, A), R);  // tau A^T R
  V3 g = taup;
  if (K::SCAN) {
    // S_t = sum over t' >= t of the parent-frame torques, added from the last frame down (the clip-sequential order)
    float *sb = tp_lds;
    __syncthreads();   // every group is done reading the scan planes
    sb[(t * GROUP + L.j) * 3 + 0] = taup.x, sb[(t * GROUP + L.j) * 3 + 1] = taup.y, sb[(t * GROUP + L.j) * 3 + 2] = taup.z;
    __syncthreads();
    V3 S = v3(0.f, 0.f, 0.f);
    for (int tt = T - 1; tt >= t; --tt) {
      const float *q = sb + (tt * GROUP + L.j) * 3;
      S = S + v3(q[0], q[1], q[2]);
    }
    const M3 Rprev = (t > 0) ? mulTN(c, R) : Rref;  // change is a rotation: rel_rot[t-1] = change^T rel_rot[t]
    g = vmulT(S, Rprev);
  }
  if (L.active) {
    float gy6[6];
    if (s.c1 && s.c2) {
      V3 b3 = v3(c.m[6], c.m[7], c.m[8]);
      float al = dot(g, s.b1), be = dot(g, s.b2), ga = dot(g, b3);
      float r1 = frcp(s.n1), r2 = frcp(s.n2);
      float k3 = (be + al * s.d * r2) * r1, k2 = -ga * r1, k5 = -al * r2;
      gy6[0] = fmaf(k3, b3.x, k2 * s.b2.x), gy6[1] = fmaf(k3, b3.y, k2 * s.b2.y), gy6[2] = fmaf(k3, b3.z, k2 * s.b2.z);
      gy6[3] = k5 * b3.x, gy6[4] = k5 * b3.y, gy6[5] = k5 * b3.z;
    } else {  // a norm sits on the 1e-12 clamp: generic chain rule through the Gram-Schmidt steps
      M3 G;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        V3 ci = v3(c.m[i * 3], c.m[i * 3 + 1], c.m[i * 3 + 2]);
        V3 h = cross(ci, g) * 0.5f;
        G.m[i * 3] = h.x, G.m[i * 3 + 1] = h.y, G.m[i * 3 + 2] = h.z;
      }
      rot6d_bwd(s, G, gy6);
    }
    float2 *q = reinterpret_cast<float2 *>(grad_y + (((size_t)L.clip * T + t) * J + L.j) * 6);
    q[0] = make_float2(gy6[0], gy6[1]), q[1] = make_float2(gy6[2], gy6[3]), q[2] = make_float2(gy6[4], gy6[5]);
  }
}

#include "p2c_pose_head_pk.inc"

// =====================================================================================================================
// absolute_loc kind (projection.py:125-136 + reference_skeletons_denormalizer.py:67-91)
//   x^ = nan_to_zero((y - y[hips]) / |y[neck] - y[hips]|) ;  abs_loc = x^ * s_ref + h_ref
// =====================================================================================================================
template <int MODE>
__global__ __launch_bounds__(256) void pose_head_absloc(const p2c_pose_head_desc d, const GradLosses grad_losses,
                                                        const float *g_abs_ext, const float *g_projt_ext,
                                                        float *grad_y) {
  constexpr bool BWD = (MODE == MODE_BWD);
  const LaneCtx L = make_lane(d);
  const int T = d.T;
  constexpr int HIPS = 1, NECK = 8;  // HipsNeckExtractor(CARLA_SKELETON), reference_skeletons_denormalizer.py:37
  V3 href = v3(0.f, 0.f, 0.f);
  float sref = 1.f;
  if (L.clip < d.B) {
    int st = d.skel_type[L.clip];
    href = v3(d.ref_hn_shift[st * 3 + 0], d.ref_hn_shift[st * 3 + 1], d.ref_hn_shift[st * 3 + 2]);
    sref = d.ref_hn_scale[st];
  }
  float coef2 = 0.f, coef3 = 0.f;
  if (BWD) loss_coefs(d, grad_losses, coef2, coef3);
  World W;
  W.on = (d.dloc != nullptr) || (d.drot != nullptr);
  W.rot = identity();
  W.loc = v3(0.f, 0.f, 0.f);
  HeadAcc acc{0.f, 0.f, 0.f};

  FrameIn<3> cur, nxt;
  FramePtrs ptrs = frame_ptrs<3>(d, L, 0);
  load_frame<3, 1>(L, ptrs, cur);
  for (int t = 0; t < T; ++t) {
    const size_t jf = ((size_t)L.clip * T + t) * J + L.j;
    load_frame<3, 1>(L, ptrs, nxt);            // (unconditional prefetch: past the last frame the values are never used)
    V3 yin = v3(cur.y[0], cur.y[1], cur.y[2]);
    const bool up = L.base != 0;
    V3 h = v3(group_bcast(yin.x, HIPS, up), group_bcast(yin.y, HIPS, up), group_bcast(yin.z, HIPS, up));
    V3 k = v3(group_bcast(yin.x, NECK, up), group_bcast(yin.y, NECK, up), group_bcast(yin.z, NECK, up));
    V3 dk = k - h;
    float sc = fsqrt(dot(dk, dk));
    float inv = frcp(sc);
    V3 q = yin - h;
    V3 xn = v3(q.x * inv, q.y * inv, q.z * inv);
    bool f0 = isfinite(xn.x), f1 = isfinite(xn.y), f2 = isfinite(xn.z);
    xn = v3(f0 ? xn.x : 0.f, f1 ? xn.y : 0.f, f2 ? xn.z : 0.f);
    V3 x = xn * sref + href;
    world_step(d, L, t, W);
    if (MODE == MODE_FWD_MATERIALIZE && W.on) world_store(d, L, t, W);
    V3 gx = frame_head<MODE>(d, L, t, x, W, acc, coef2, coef3, g_abs_ext, g_projt_ext, cur.g2, cur.g3);
    if (BWD) {
      bool ok = isfinite(inv) && sc != 0.f;
      V3 gn = v3(f0 && ok ? gx.x * sref * inv : 0.f, f1 && ok ? gx.y * sref * inv : 0.f,
                 f2 && ok ? gx.z * sref * inv : 0.f);  // d/dq
      V3 S = v3(group_sum(gn.x, L.base != 0), group_sum(gn.y, L.base != 0), group_sum(gn.z, L.base != 0));
      float Cs = group_sum(gn.x * xn.x + gn.y * xn.y + gn.z * xn.z, L.base != 0);
      float rr = (sc > 0.f) ? -Cs * inv : 0.f;   // d/d sc  times 1/sc
      V3 gk = dk * rr;
      V3 gh = v3(-S.x, -S.y, -S.z) - gk;
      V3 gy = gn;
      if (L.j == HIPS) gy = gy + gh;
      if (L.j == NECK) gy = gy + gk;
      if (L.active) {
        float *g = grad_y + jf * 3;
        g[0] = gy.x, g[1] = gy.y, g[2] = gy.z;
      }
    }
    cur = nxt;
  }
  if (!BWD) {
    float s2 = wave_sum(acc.sum2), c2 = wave_sum(acc.cnt2), s3 = wave_sum(acc.sum3);
    if (L.lane == 0) {
      size_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
      float *p = d.partials + wave * 4;
      p[0] = s2, p[1] = c2, p[2] = s3, p[3] = 0.f;
    }
  }
}

// =====================================================================================================================
// deterministic second stage of the loss reduction (fixed order, fp64 accumulators)
// =====================================================================================================================
__global__ __launch_bounds__(256) void loss_finalize(const float *partials, int n_waves, float n3_elems, int has2d,
                                                      int has3d, float *loss_sums, float *losses, float nrot_elems) {
  // One workgroup, fixed order: thread i adds partials i, i + 256, ... (fp64), then a tree over the threads (p2c_train.hip's
  // finalize_losses keeps the same order: bit-identical losses). Every partial is one 16-byte load, eight of them in flight per
  // thread: with dword loads in a rolled loop this launch took 15 us for the 8192 partials of a 65 536-clip batch -- 6 % of
  // the kernel whose sums it adds.
  __shared__ double sh[4][256];
  const float4 *p4 = reinterpret_cast<const float4 *>(partials);
  double a = 0.0, b = 0.0, c = 0.0, r = 0.0;       // r: slot 3 of a partial = the rot_3d sum when gt_rot is set (nrot_elems > 0)
  int i = threadIdx.x;
  for (; i + 7 * 256 < n_waves; i += 8 * 256) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = p4[i + u * 256];
#pragma unroll
    for (int u = 0; u < 8; ++u) a += (double)v[u].x, b += (double)v[u].y, c += (double)v[u].z, r += (double)v[u].w;
  }
  for (; i < n_waves; i += 256) {
    const float4 v = p4[i];
    a += (double)v.x, b += (double)v.y, c += (double)v.z, r += (double)v.w;
  }
  sh[0][threadIdx.x] = a, sh[1][threadIdx.x] = b, sh[2][threadIdx.x] = c, sh[3][threadIdx.x] = r;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      sh[0][threadIdx.x] += sh[0][threadIdx.x + s];
      sh[1][threadIdx.x] += sh[1][threadIdx.x + s];
      sh[2][threadIdx.x] += sh[2][threadIdx.x + s];
      sh[3][threadIdx.x] += sh[3][threadIdx.x + s];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0 && nrot_elems > 0.f) {
    loss_sums[4] = (float)sh[3][0], loss_sums[5] = nrot_elems;
    losses[3] = (float)(sh[3][0] / (double)nrot_elems);
  }
  if (threadIdx.x == 0) {
    double s2 = sh[0][0], n2 = sh[1][0], s3 = sh[2][0];
    loss_sums[0] = (float)s2, loss_sums[1] = (float)n2, loss_sums[2] = (float)s3, loss_sums[3] = n3_elems;
    const float nan = __builtin_nanf("");
    float l2 = has2d ? (float)(s2 / (2.0 * n2)) : nan;  // MSELoss(mean) over the unmasked (x, y) pairs
    float l3 = has3d ? (float)(s3 / (double)n3_elems) : nan;
    losses[0] = l2, losses[1] = l3, losses[2] = l2 + l3;
  }
}

}  // namespace p2c

// =====================================================================================================================
// C ABI
// =====================================================================================================================
using namespace p2c;

static constexpr int kBlock = 256;  // four independent wavefronts (two clips each) per workgroup; no barriers

static int validate(const p2c_pose_head_desc *d) {
  if (!d || !d->y || !d->skel_type || !d->partials || !d->loss_sums || !d->losses) return P2C_E_NULL;
  if (d->B <= 0 || d->T <= 0 || d->t0 < 0 || d->t1 > d->T || d->t0 > d->t1) return P2C_E_SHAPE;
  if (d->kind < 0 || d->kind > P2C_KIND_ABSOLUTE_LOC) return P2C_E_ENUM;
  if (d->transform < 0 || d->transform > P2C_TRANSFORM_HIPS_NECK_BBOX) return P2C_E_ENUM;
  if (d->kind == P2C_KIND_ABSOLUTE_LOC) {
    if (!d->ref_hn_shift || !d->ref_hn_scale) return P2C_E_NULL;
  } else {
    if (!d->ref_rel_loc || !d->ref_rel_rot) return P2C_E_NULL;
  }
  if (d->n_hips < 1 || d->n_hips > 2 || d->n_neck < 1 || d->n_neck > 2) return P2C_E_INDEX;
  for (int i = 0; i < d->n_hips; ++i)
    if (d->hips_idx[i] < 0 || d->hips_idx[i] >= P2C_JOINTS) return P2C_E_INDEX;
  for (int i = 0; i < d->n_neck; ++i)
    if (d->neck_idx[i] < 0 || d->neck_idx[i] >= P2C_JOINTS) return P2C_E_INDEX;
  if (d->hips_lane < -1 || d->hips_lane >= P2C_JOINTS) return P2C_E_INDEX;
  int n2 = 0, n3 = 0;
  for (int j = 0; j < P2C_JOINTS; ++j) {
    if (d->gt2d && (d->gmap2d[j] < -1 || d->gmap2d[j] >= d->gt2d_joints)) return P2C_E_INDEX;
    if (d->gt3d && (d->gmap3d[j] < -1 || d->gmap3d[j] >= d->gt3d_joints)) return P2C_E_INDEX;
    n2 += d->gmap2d[j] >= 0;
    n3 += d->gmap3d[j] >= 0;
  }
  if (d->gt2d && (d->gt2d_channels < 2 || n2 != d->n_common2d)) return P2C_E_SHAPE;
  if (d->gt3d && n3 != d->n_common3d) return P2C_E_SHAPE;
  if (d->gt_rot) {                                   // rot_3d fused: 6-D kinds, the joint map of the 3-D targets
    if (d->kind != P2C_KIND_POSE_CHANGES_6D && d->kind != P2C_KIND_RELATIVE_ROT_6D) return P2C_E_ENUM;
    if (n3 != d->n_common3d || d->gt3d_joints < 1) return P2C_E_SHAPE;
    for (int j = 0; j < P2C_JOINTS; ++j)
      if (d->gmap3d[j] < -1 || d->gmap3d[j] >= d->gt3d_joints) return P2C_E_INDEX;
  }
  return 0;
}

int p2c_internal_validate_pose_head(const p2c_pose_head_desc *d) { return validate(d); }   // for p2c_train.hip

static inline unsigned grid_for(int B) {
  int waves = (B + 1) / 2;
  int waves_per_block = kBlock / 64;
  return (unsigned)((waves + waves_per_block - 1) / waves_per_block);
}

extern "C" const char *p2c_version(void) { return "p2c-hip 0.1.0 gfx950"; }

// Batches up to this many clips take the time-parallel kernels (6-D kinds, lean outputs, T <= 32); larger ones have
// enough clips to fill the chip with the cheaper clip-sequential walk. P2C_TP_MAX_B overrides (tuning / tests).
static int g_tp_max_b = -1;
static int tp_max_b() {
  if (g_tp_max_b < 0) {
    const char *e = getenv("P2C_TP_MAX_B");
    g_tp_max_b = e ? atoi(e) : 2048;
  }
  return g_tp_max_b;
}
extern "C" int p2c_pose_head_set_time_parallel_max_batch(int32_t max_b) {
  const int prev = tp_max_b();
  if (max_b >= 0) g_tp_max_b = max_b;
  return prev;
}
static inline bool use_tp(const p2c_pose_head_desc &d) {
  return (d.kind == P2C_KIND_POSE_CHANGES_6D || d.kind == P2C_KIND_RELATIVE_ROT_6D) && d.T <= 32 && d.B <= tp_max_b();
}
// EXPERIMENTAL, off by default: the packed-fp32 kernels (two clips per lane, p2c_pose_head_pk.inc) for large batches of the
// training configuration (6-D kinds, lean outputs, no world motion, no external gradients). They halve the FP instruction
// count but not the cross-lane / register-pairing moves and run at 2-3 waves per SIMD: measured 370 vs 324 us (forward)
// and 667 vs 538 us (backward) at B = 65 536, i.e. slower than the scalar kernels. P2C_PK_MIN_B /
// p2c_pose_head_set_packed_min_batch enable them from a batch size on (parity-tested in tests/test_pose_head_gpu.py).
static int g_pk_min_b = -1;
static int pk_min_b() {
  if (g_pk_min_b < 0) {
    const char *e = getenv("P2C_PK_MIN_B");
    g_pk_min_b = e ? atoi(e) : (1 << 30);
  }
  return g_pk_min_b;
}
extern "C" int p2c_pose_head_set_packed_min_batch(int32_t min_b) {
  const int prev = pk_min_b();
  if (min_b >= 0) g_pk_min_b = min_b;
  return prev;
}
static inline bool use_pk(const p2c_pose_head_desc &d) {
  return (d.kind == P2C_KIND_POSE_CHANGES_6D || d.kind == P2C_KIND_RELATIVE_ROT_6D) && !d.dloc && !d.drot && !d.gt_rot &&
         d.B >= pk_min_b() && !use_tp(d);
}
// chain-lane kernels for large batches (p2c_pose_head_chain.hip)
bool p2c_internal_chain_supported(const p2c_pose_head_desc &d);
unsigned p2c_internal_chain_waves(int B);
int p2c_internal_chain_fwd(const p2c_pose_head_desc &d, hipStream_t stream);
int p2c_internal_chain_bwd(const p2c_pose_head_desc &d, const GradLosses &gl, float *grad_y, hipStream_t stream);
static inline bool use_chain(const p2c_pose_head_desc &d) {
  return !use_tp(d) && !use_pk(d) && !d.gt_rot && p2c_internal_chain_supported(d);
}

static inline unsigned grid_pk(int B) {
  const int waves = (B + 3) / 4;
  return (unsigned)((waves + kBlock / 64 - 1) / (kBlock / 64));
}
static inline float n3_elems_host(const p2c_pose_head_desc &d) {
  return (float)((double)d.B * (double)(d.t1 - d.t0) * (double)d.n_common3d * 3.0);
}
static inline float nrot_elems_host(const p2c_pose_head_desc &d) {
  return d.gt_rot ? (float)((double)d.B * (double)(d.t1 - d.t0) * (double)d.n_common3d * 9.0) : 0.f;
}
static inline unsigned tp_threads(int T) { return 64u * (unsigned)((T + 1) / 2); }
static inline size_t tp_lds_bytes(int T) {
  const size_t waves = (size_t)(T + 1) / 2;
  return (2 * waves * 2 * GROUP * 9 + waves * 4) * sizeof(float);
}

extern "C" int64_t p2c_pose_head_workspace_floats(int32_t B) {
  if (B <= 0) return 0;
  const int64_t seq = (int64_t)grid_for(B) * (kBlock / 64) * 4, tp = (int64_t)B * 4;   // per-wave / per-clip partials
  return seq > tp ? seq : tp;
}

// which: bit 0 = the pose-head kernel, bit 1 = the one-workgroup loss reduction behind it (p2c_pose_head_fwd_launch: measurement)
static int pose_head_fwd_impl(const p2c_pose_head_desc *desc, void *stream_, int which) {
  int rc = validate(desc);
  if (rc) return rc;
  p2c_pose_head_desc d = *desc;
  if (d.gt_rot) d.defer_loss_finalize = 0;            // rot_3d fused: the forward kernel runs and its sums are finalized here
  hipStream_t stream = (hipStream_t)stream_;
  dim3 grid(grid_for(d.B)), block(kBlock);
  const bool mat = d.out_pose_changes || d.out_projection_2d || d.out_projection_2d_transformed || d.out_shift ||
                   d.out_scale || d.out_relative_pose_loc || d.out_relative_pose_rot || d.out_absolute_pose_loc ||
                   d.out_absolute_pose_rot || d.out_world_loc || d.out_world_rot;
  if ((d.kind == P2C_KIND_POSE_CHANGES_6D || d.kind == P2C_KIND_POSE_CHANGES_MAT) && !d.final_rel_rot) return P2C_E_NULL;
  const bool tp = !mat && use_tp(d);
  const bool pkd = !mat && use_pk(d);
  const bool sixd = d.kind == P2C_KIND_POSE_CHANGES_6D || d.kind == P2C_KIND_RELATIVE_ROT_6D;
  if (d.defer_loss_finalize == 2 && tp && !pkd && sixd) {     // "train": the backward call does all of it
    hipLaunchKernelGGL(pose_head_count_tp, dim3((unsigned)d.B), dim3(tp_threads(d.T)), 0, stream, d);
    hipError_t e0 = hipGetLastError();
    return e0 == hipSuccess ? 0 : (int)e0;
  }
  const dim3 tp_grid((unsigned)d.B), tp_block(tp_threads(d.T)), pk_grid(grid_pk(d.B));
  if (!mat && use_chain(d)) {            // large batch of the training configuration: eight clips per wavefront
    rc = (which & 1) ? p2c_internal_chain_fwd(d, stream) : 0;
    if (rc) return rc;
    if (!(which & 2)) return 0;
    hipLaunchKernelGGL(loss_finalize, dim3(1), dim3(256), 0, stream, (const float *)d.partials, (int)p2c_internal_chain_waves(d.B),
                       n3_elems_host(d), d.gt2d ? 1 : 0, d.gt3d ? 1 : 0, d.loss_sums, d.losses, 0.f);
    hipError_t ec = hipGetLastError();
    return ec == hipSuccess ? 0 : (int)ec;
  }
#define P2C_LAUNCH_ROT_FWD(KIND)                                                                                \
  if (mat) hipLaunchKernelGGL((pose_head_rot_fwd<KIND, true>), grid, block, 0, stream, d);                      \
  else if (tp) hipLaunchKernelGGL((pose_head_rot_fwd_tp<KIND>), tp_grid, tp_block, tp_lds_bytes(d.T), stream, d); \
  else hipLaunchKernelGGL((pose_head_rot_fwd<KIND, false>), grid, block, 0, stream, d)
  if (which & 1) switch (d.kind) {
    case P2C_KIND_POSE_CHANGES_6D:
      if (pkd) hipLaunchKernelGGL((pk::pose_head_rot_fwd_pk<P2C_KIND_POSE_CHANGES_6D>), pk_grid, block, 0, stream, d);
      else { P2C_LAUNCH_ROT_FWD(P2C_KIND_POSE_CHANGES_6D); }
      break;
    case P2C_KIND_POSE_CHANGES_MAT: P2C_LAUNCH_ROT_FWD(P2C_KIND_POSE_CHANGES_MAT); break;
    case P2C_KIND_RELATIVE_ROT_6D:
      if (pkd) hipLaunchKernelGGL((pk::pose_head_rot_fwd_pk<P2C_KIND_RELATIVE_ROT_6D>), pk_grid, block, 0, stream, d);
      else { P2C_LAUNCH_ROT_FWD(P2C_KIND_RELATIVE_ROT_6D); }
      break;
    case P2C_KIND_RELATIVE_ROT_MAT: P2C_LAUNCH_ROT_FWD(P2C_KIND_RELATIVE_ROT_MAT); break;
    default:
      if (mat)
        hipLaunchKernelGGL((pose_head_absloc<MODE_FWD_MATERIALIZE>), grid, block, 0, stream, d, GradLosses{{nullptr, nullptr, nullptr}},
                           (const float *)nullptr, (const float *)nullptr, (float *)nullptr);
      else
        hipLaunchKernelGGL((pose_head_absloc<MODE_FWD>), grid, block, 0, stream, d, GradLosses{{nullptr, nullptr, nullptr}},
                           (const float *)nullptr, (const float *)nullptr, (float *)nullptr);
  }
#undef P2C_LAUNCH_ROT_FWD
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  if (!(which & 2)) return 0;
  if (d.defer_loss_finalize && tp && !mat && !pkd && (d.kind == P2C_KIND_POSE_CHANGES_6D || d.kind == P2C_KIND_RELATIVE_ROT_6D))
    return 0;                            // p2c_pose_head_bwd's time-parallel kernel finishes the reduction
  int n_waves = tp ? d.B : (int)((pkd ? pk_grid.x : grid.x) * (kBlock / 64));
  float n3 = (float)((double)d.B * (double)(d.t1 - d.t0) * (double)d.n_common3d * 3.0);
  hipLaunchKernelGGL(loss_finalize, dim3(1), dim3(256), 0, stream, (const float *)d.partials, n_waves, n3,
                     d.gt2d ? 1 : 0, d.gt3d ? 1 : 0, d.loss_sums, d.losses, nrot_elems_host(d));
  e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

extern "C" int p2c_pose_head_fwd(const p2c_pose_head_desc *desc, void *stream_) { return pose_head_fwd_impl(desc, stream_, 3); }
extern "C" int p2c_pose_head_fwd_launch(const p2c_pose_head_desc *desc, int32_t which, void *stream_) {
  if (which < 1 || which > 3) return P2C_E_ENUM;
  return pose_head_fwd_impl(desc, stream_, which);
}

extern "C" int p2c_pose_head_bwd(const p2c_pose_head_desc *desc, const float *const grad_losses_[3],
                                 const float *grad_absolute_pose_loc, const float *grad_projection_2d_transformed,
                                 const float *grad_absolute_pose_rot, float *grad_y, void *stream_) {
  int rc = validate(desc);
  if (rc) return rc;
  if (!grad_y) return P2C_E_NULL;
  p2c_pose_head_desc d = *desc;
  // the forward skipped its finalize launch only for the lean time-parallel 6-D kernels (same rule as p2c_pose_head_fwd)
  if (!(use_tp(d) && !use_pk(d) && (d.kind == P2C_KIND_POSE_CHANGES_6D || d.kind == P2C_KIND_RELATIVE_ROT_6D)) || d.gt_rot)
    d.defer_loss_finalize = 0;
  hipStream_t stream = (hipStream_t)stream_;
  dim3 grid(grid_for(d.B)), block(kBlock);
  GradLosses grad_losses{{nullptr, nullptr, nullptr}};
  if (grad_losses_)
    for (int i = 0; i < 3; ++i) grad_losses.p[i] = grad_losses_[i];
  const float *ga = grad_absolute_pose_loc, *gp = grad_projection_2d_transformed, *gr = grad_absolute_pose_rot;
  if (gr && d.kind != P2C_KIND_POSE_CHANGES_6D && d.kind != P2C_KIND_RELATIVE_ROT_6D)
    return P2C_E_ENUM;      // rotation-loss gradients go through the tangent-space backward of the 6-D kinds
  if (!ga && !gp && !gr && use_chain(d)) {
    if (d.kind == P2C_KIND_POSE_CHANGES_6D && !d.final_rel_rot) return P2C_E_NULL;
    return p2c_internal_chain_bwd(d, grad_losses, grad_y, stream);
  }
  switch (d.kind) {
    case P2C_KIND_POSE_CHANGES_6D:
      if (!d.final_rel_rot) return P2C_E_NULL;
      if (use_pk(d) && !ga && !gp && !gr)
        hipLaunchKernelGGL(pk::pose_head_rot_bwd_tangent_pk<P2C_KIND_POSE_CHANGES_6D>, dim3(grid_pk(d.B)), block, 0, stream, d,
                           grad_losses, grad_y);
      else if (use_tp(d) && d.defer_loss_finalize == 2)
        hipLaunchKernelGGL((pose_head_rot_bwd_tangent_tp<P2C_KIND_POSE_CHANGES_6D, true>), dim3((unsigned)d.B),
                           dim3(tp_threads(d.T)), tp_lds_bytes(d.T), stream, d, grad_losses, ga, gp, gr, grad_y);
      else if (use_tp(d))
        hipLaunchKernelGGL(pose_head_rot_bwd_tangent_tp<P2C_KIND_POSE_CHANGES_6D>, dim3((unsigned)d.B), dim3(tp_threads(d.T)),
                           tp_lds_bytes(d.T), stream, d, grad_losses, ga, gp, gr, grad_y);
      else
        hipLaunchKernelGGL(pose_head_rot_bwd_tangent<P2C_KIND_POSE_CHANGES_6D>, grid, block, 0, stream, d, grad_losses, ga, gp, gr, grad_y);
      break;
    case P2C_KIND_POSE_CHANGES_MAT:
      if (!d.final_rel_rot) return P2C_E_NULL;
      hipLaunchKernelGGL(pose_head_rot_bwd<P2C_KIND_POSE_CHANGES_MAT>, grid, block, 0, stream, d, grad_losses, ga, gp, grad_y);
      break;
    case P2C_KIND_RELATIVE_ROT_6D:
      if (use_pk(d) && !ga && !gp && !gr)
        hipLaunchKernelGGL(pk::pose_head_rot_bwd_tangent_pk<P2C_KIND_RELATIVE_ROT_6D>, dim3(grid_pk(d.B)), block, 0, stream, d,
                           grad_losses, grad_y);
      else if (use_tp(d) && d.defer_loss_finalize == 2)
        hipLaunchKernelGGL((pose_head_rot_bwd_tangent_tp<P2C_KIND_RELATIVE_ROT_6D, true>), dim3((unsigned)d.B),
                           dim3(tp_threads(d.T)), tp_lds_bytes(d.T), stream, d, grad_losses, ga, gp, gr, grad_y);
      else if (use_tp(d))
        hipLaunchKernelGGL(pose_head_rot_bwd_tangent_tp<P2C_KIND_RELATIVE_ROT_6D>, dim3((unsigned)d.B), dim3(tp_threads(d.T)),
                           tp_lds_bytes(d.T), stream, d, grad_losses, ga, gp, gr, grad_y);
      else
        hipLaunchKernelGGL(pose_head_rot_bwd_tangent<P2C_KIND_RELATIVE_ROT_6D>, grid, block, 0, stream, d, grad_losses, ga, gp, gr, grad_y);
      break;
    case P2C_KIND_RELATIVE_ROT_MAT:
      hipLaunchKernelGGL(pose_head_rot_bwd<P2C_KIND_RELATIVE_ROT_MAT>, grid, block, 0, stream, d, grad_losses, ga, gp, grad_y);
      break;
    default:
      hipLaunchKernelGGL((pose_head_absloc<MODE_BWD>), grid, block, 0, stream, d, grad_losses, ga, gp, grad_y);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  if (d.defer_loss_finalize == 2) {      // the kernel above left one (sum_sq_2d, n, sum_sq_3d) per clip
    hipLaunchKernelGGL(loss_finalize, dim3(1), dim3(256), 0, stream, (const float *)d.partials, d.B, n3_elems_host(d),
                       d.gt2d ? 1 : 0, d.gt3d ? 1 : 0, d.loss_sums, d.losses, 0.f);
    e = hipGetLastError();
  }
  return e == hipSuccess ? 0 : (int)e;
}
