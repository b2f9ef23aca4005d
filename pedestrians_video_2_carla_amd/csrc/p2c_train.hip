// p2c_train.hip -- the small-batch train step of LitPoseLiftingFlow(LinearAE, 6-D output) in TWO launches (gfx950).
//
// What it replaces (reference, paths relative to src/pedestrians_video_2_carla/): modules/flow/base.py:397-410 (_step) ->
// modules/flow/pose_lifting.py:121-144 (_inner_step: LinearAE.forward, linear_ae.py:50-59 -> ProjectionModule ->
// transform_callable) -> loss/loc_2d_3d.py:6-17 -> backward -> AdamW (flow/base_model.py:156-158).
//
// The separate kernels of this library run that step at B = 256 as seven dependent launches (count, mlp_fwd, pose-head
// backward, loss_finalize, mlp_bwd, mlp_wgrad, mlp_reduce) whose sum is launch latency, and move the model output y, its
// gradient and the MLP's factors through HBM in between. Here one clip of T <= 16 frames IS one 16-sample MFMA tile:
//
//   train_clip_kernel   one workgroup (8 waves) per clip: LinearAE forward of the clip's 16 frames on fp32 MFMA (activations
//                       transposed in LDS, the cooperative-tile code of p2c_mlp_dev.h) -> y^T stays in LDS -> the
//                       time-parallel pose head (one 32-lane group per frame, lane = joint: 6-D -> R, cumulative-rotation
//                       scan through LDS, FK by pointer doubling, projection, normaliser, loc_2d + loc_3d) and its
//                       tangent-space backward (p2c_pose_head_dev.h) -> grad_y^T written back into the same LDS rows ->
//                       the dgrad chain. Leaves: the per-clip loss sums and the clip's weight-gradient FACTORS
//                       (H_0..H_5, G_1..G_6 transposed, 532 rows x 16 samples = 34 KB).
//   train_wgrad_kernel  one workgroup (16 waves) per 16x16 tile of dW_aug = G^T [H | 1]: contracts the factors over ALL
//                       clips in a fixed order, applies AdamW to the tile's parameters in place (p2c_adam_math.h) and
//                       refreshes the packed weight image; one extra workgroup finishes the loss reduction (fp64, fixed
//                       order). No atomics on data, bitwise reproducible.
//
// Arithmetic and summation orders are those of the separate kernels (mlp_fwd / pose_head_rot_bwd_tangent_tp<TRAIN> /
// mlp_bwd<FACTORS> / mlp_wgrad + mlp_reduce_small / loss_finalize): at a batch where those run the split weight gradient
// the two paths produce bit-identical losses, gradients and parameters (tests/test_train_fused_gpu.py).
// The number of unmasked 2-D target pairs (the loc_2d denominator, utils/tensors.py:29-40) is a property of the targets
// alone: p2c_count_target_pairs computes it when a batch is staged, not per step.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <stdint.h>

#include "../../include/p2c.h"
#include "p2c_adam_math.h"
#include "p2c_mlp_dev.h"
#include "p2c_pose_head_dev.h"
#include "p2c_train_dev.h"

int p2c_internal_validate_pose_head(const p2c_pose_head_desc *d);   // p2c_pose_head.hip

namespace p2c_train {

constexpr int T_MAX = TS;                    // one clip = one sample tile
constexpr int PLANE = T_MAX * ph::GROUP * 9; // one plane of the cumulative-rotation scan (floats)
constexpr int H_ROWS = S::h_off(NLAY);       // rows of H_0 .. H_{L-1}
constexpr int G_ROWS = S::act_rows() - S::h_off(1);   // rows of G_1 .. G_L
constexpr int ACT_FLOATS = (H_ROWS + G_ROWS) * TP;
static_assert((S::h_off(NLAY) - S::h_off(1)) * TP >= PLANE, "scan plane 1 aliases the rows of G_1 .. G_{L-1}");
static_assert(S::dim_at(NLAY) == ph::J * 6, "the last layer is the (26, 6) rotation output");
constexpr int SCRATCH = 64;
constexpr int LDS_FLOATS = S::w_total() + ACT_FLOATS + PLANE + SCRATCH;
static_assert(LDS_FLOATS * 4 <= 160 * 1024, "LDS budget");


#ifdef P2C_TRAIN_TRACE   // developer build only (tools/traintrace.py): shader-clock stamps of one workgroup per kernel
static __device__ unsigned long long g_ttrace[2][40];
#ifndef P2C_TRAIN_TRACE_BLOCK
#define P2C_TRAIN_TRACE_BLOCK 0
#endif
#define TT(k, i)                                                                         \
  do {                                                                                   \
    if (blockIdx.x == P2C_TRAIN_TRACE_BLOCK && threadIdx.x == 0) {                       \
      g_ttrace[k][i] = __builtin_readcyclecounter();                                     \
      if ((i) == 0 || (i) == 39) g_ttrace[k][(i) == 0 ? 38 : 37] = wall_clock64();      \
    }                                                                                    \
  } while (0)
#ifndef P2C_TRAIN_TRACE_TILE
#define P2C_TRAIN_TRACE_TILE 60
#endif
// second kernel: stamps stay in registers and are written out by the LAST ARRIVER of one tile (the long path)
#define TB_DECL unsigned long long tb_[40] = {0}; unsigned long long tbw0_ = wall_clock64()
#define TB(i) tb_[i] = __builtin_readcyclecounter()
#define TB_DUMP(cond)                                                      \
  do {                                                                     \
    if ((cond) && threadIdx.x == 0) {                                      \
      for (int i_ = 0; i_ < 40; ++i_) g_ttrace[1][i_] = tb_[i_];           \
      g_ttrace[1][38] = tbw0_, g_ttrace[1][37] = wall_clock64();           \
    }                                                                      \
  } while (0)
#else
#define TT(k, i)
#define TB_DECL
#define TB(i)
#define TB_DUMP(cond)
#endif

// geometry of the packed weight image for a layer index known only at run time
__device__ __forceinline__ int image_w_off(int l) {
  int r = 0;
#pragma unroll
  for (int i = 0; i < NLAY; ++i)
    if (i == l) r = S::w_off(i);
  return r;
}
__device__ __forceinline__ int image_ld(int l) {
  int r = 0;
#pragma unroll
  for (int i = 0; i < NLAY; ++i)
    if (i == l) r = S::ld(i);
  return r;
}

// inclusive scan over the frames of the clip, P_t = c_t c_{t-1} ... c_0 (the order of p2c::scan_time), ping-pong planes
__device__ __forceinline__ ph::M3 scan_time2(ph::M3 P, int t, int j, int T, float *p0, float *p1) {
  float *cur = p0, *oth = p1;
  for (int off = 1; off < T; off <<= 1) {
    float *mine = cur + (t * ph::GROUP + j) * 9;
#pragma unroll
    for (int i = 0; i < 9; ++i) mine[i] = P.m[i];
    lds_barrier();
    if (t >= off) {
      const float *q = cur + ((t - off) * ph::GROUP + j) * 9;
      ph::M3 Q;
#pragma unroll
      for (int i = 0; i < 9; ++i) Q.m[i] = q[i];
      P = ph::mul(P, Q);
    }
    float *tmp = cur;
    cur = oth, oth = tmp;
  }
  return P;
}


// ---- the weight image by LDS-DMA (round 3 experiment, OFF: build with EXTRA=-DP2C_TRAIN_DMA_IMAGE=1) ------------------------
// The 70 KB image is a straight copy HBM -> LDS; staged through registers (stage_issue / stage_commit, K8) it arrives at
// ~11 B/clk per CU. Variant: every wave issues ALL its 1 KB pieces at once (buffer_load ... lds: no registers held, no LDS
// store instructions) and a layer waits for the pieces it reads with s_waitcnt vmcnt(n) (vector-memory operations retire in
// order). Correct (bitwise tests pass) but SLOWER: train_clip_kernel 19.0 vs 18.1 us at B = 256, equal at B = 1024. The ~25
// ordinary loads a wave issues behind its pieces (targets, tables, counts) cannot retire before them and make every counted
// wait cover the whole image, so the per-layer pipelining of the staged rounds is lost; issuing the pieces after those loads
// would put them behind the skel_type -> table address chain.
#ifndef P2C_TRAIN_DMA_IMAGE
#define P2C_TRAIN_DMA_IMAGE 0
#endif
typedef __attribute__((address_space(3))) void *lds_void_ptr;
constexpr int IMG_CHUNKS = ((S::w_total() >> 2) + 63) / 64;          // 1 KB pieces (64 lanes x 16 B)
__device__ __forceinline__ void image_dma_issue(const float *w_image, float *lds, int wave, int lane) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(w_image), 0, S::w_total() * 4, 0x00020000);
  const unsigned base = (unsigned)(uintptr_t)(lds_void_ptr)lds;
  constexpr int total4 = S::w_total() >> 2;
#pragma unroll
  for (int i = 0; i < (IMG_CHUNKS + WAVES - 1) / WAVES; ++i) {
    const int c = wave + i * WAVES;                                  // (wave-uniform)
    if (c < IMG_CHUNKS && c * 64 + lane < total4)                    // the last piece is partial: its tail lanes stay out (the LDS
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_ptr)(uintptr_t)(base + c * 1024), 16, (c * 64 + lane) * 16, 0, 0, 0);   // behind the image is live)
  }
}
// this wave's pieces of layers 0..l have landed (pieces are issued in ascending order, wave w owns pieces w, w + WAVES, ...)
__device__ __forceinline__ void image_dma_wait(int l, int wave) {
  const int end4 = (l + 1 >= NLAY) ? (S::w_total() >> 2) : ((S::w_off(l + 1) + 3) >> 2);
  const int bound = (end4 + 63) / 64;                                // pieces [0, bound) hold layers 0..l
  const int mine = wave < IMG_CHUNKS ? (IMG_CHUNKS - wave + WAVES - 1) / WAVES : 0;
  const int before = wave < bound ? (bound - wave + WAVES - 1) / WAVES : 0;
  switch (mine - before) {                                           // my pieces that may still be on their way
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
  }
}

template <int KIND>
__global__ __launch_bounds__(64 * WAVES) void train_clip_kernel(const p2c_pose_head_desc d, const ph::GradLosses gl,
                                                                const ClipArgs m) {
  using K = ph::KindTraits<KIND>;
  static_assert(K::SIXD, "the fused step is for the 6-D kinds");
  extern __shared__ float lds[];
  const S sh;
  Lane L;
  L.lane = threadIdx.x & 63, L.c = L.lane & 15, L.g = L.lane >> 4;
  L.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int nl = NLAY;
  [[maybe_unused]] constexpr int total4 = S::w_total() >> 2;
  float *H = lds + S::w_total();
  float *G = H + (S::h_off(nl) - S::h_off(1)) * TP;        // G_l lives at row h_off(l) of this base (l = 1..L)
  float *Y = G + S::h_off(nl) * TP;                        // y^T, later grad_y^T (= G_L)
  float *plane0 = H + ACT_FLOATS, *plane1 = G + S::h_off(1) * TP;
  float *scratch = plane0 + PLANE;
  const int T = d.T;
  TT(0, 0);

  // ---- clip-independent set-up --------------------------------------------------------------------------------------------------
  // Persistent over clips: workgroup b walks clips b, b + gridDim.x, ... (grid = min(B, CUs)); the 84 KB weight image is staged
  // ONCE, during the first clip's forward. The first clip's x tile is requested first, then the image rounds.
  TileRegs xr;
  tile_issue(m.x, (int64_t)blockIdx.x * T, (int64_t)blockIdx.x * T + T, S::dims(0), true, xr);
#if P2C_TRAIN_DMA_IMAGE
  image_dma_issue(m.w_image, lds, L.wave, L.lane);
#else
  ImageRegs wr;
  stage_issue(m.w_image, total4, wr, 0, 0, issue_mark<S>(1));
#endif
  // lane context of the pose head (= p2c::make_lane_tp) -- with identity joint maps (CARLA targets for a CARLA model, the
  // usual case) nothing in the prologue depends on a loaded value: every load is issued back to back. (The general maps
  // are kernel arguments indexed by lane, i.e. vector loads whose result the target addresses wait for.)
  ph::LaneCtx PL;
  PL.lane = threadIdx.x & 63, PL.j = PL.lane & 31, PL.base = PL.lane & 32, PL.clip = blockIdx.x;
  const int t = (int)(threadIdx.x >> 6) * 2 + (PL.lane >> 5);
  PL.active = (PL.j < ph::J) && (t < T);
  if (m.identity_maps) {
    PL.anc0 = ph::c_parent[PL.j], PL.anc1 = ph::c_anc_r2[PL.j], PL.anc2 = ph::c_anc_r3[PL.j];
    PL.interior = (ph::kInteriorMask >> PL.j) & 1u;
    PL.sub_end = ph::c_subtree_end[PL.j];
    PL.gm2 = PL.gm3 = (PL.j < ph::J) ? PL.j : -1;
    PL.never_masked = (PL.j == d.hips_lane);
    PL.has2 = PL.active && d.gt2d;
    PL.has3 = PL.active && d.gt3d;
  } else {
    ph::fill_lane(PL, d);
  }
  const int jc = PL.j < ph::J ? PL.j : 0, tc = t < T ? t : T - 1;
  // pair counts: issued now, consumed after the first clip's MLP forward (no add here: an add would wait for every load above)
  const float c0raw = m.counts[(int)threadIdx.x < d.B ? threadIdx.x : 0];
  const float c1raw = m.counts[(int)threadIdx.x + NTH < d.B ? threadIdx.x + NTH : 0];
  if (blockIdx.x == 0 && (int)threadIdx.x < m.n_counters) m.counters[threadIdx.x] = 0;   // arrival tickets of train_wgrad_kernel
  init_rows(H, S::h_off(0) + S::dims(0), S::h_off(0) + k_rows(S::dims(0)), S::h_off(0) + S::dims(0));
  float coef2 = 0.f, coef3 = 0.f;             // loss coefficients: the same for every clip, worked out in the first one
  float *red = scratch + 16;

  auto one_clip = [&](const int clip, auto first_c) {
  constexpr bool first = decltype(first_c)::value;
  PL.clip = clip;
  if constexpr (!first) lds_barrier();        // the previous clip's factor store / loss sums have consumed H, G and scratch
  // Every lane loads UNCONDITIONALLY from a clamped (always valid) address and selects afterwards: a load inside a divergent
  // branch makes the compiler wait for it (and for every older load) at the branch's end.
  const int st = d.skel_type[clip];
  float lraw[3], rraw[9];
  {
    const float *pl = d.ref_rel_loc + ((size_t)st * ph::J + jc) * 3;
    lraw[0] = pl[0], lraw[1] = pl[1], lraw[2] = pl[2];
    const float *pr = d.ref_rel_rot + ((size_t)st * ph::J + jc) * 9;
#pragma unroll
    for (int i = 0; i < 9; ++i) rraw[i] = K::SCAN ? pr[i] : 0.f;
  }
  float g2raw[2], g3raw[3];
  {
    const size_t frame = (size_t)clip * T + tc;
    const float *b2 = d.gt2d ? d.gt2d : d.ref_rel_loc, *b3 = d.gt3d ? d.gt3d : d.ref_rel_loc;   // any readable address
    const float *p2 = b2 + (PL.has2 ? (frame * d.gt2d_joints + PL.gm2) * d.gt2d_channels : 0);
    const float *p3 = b3 + (PL.has3 ? (frame * d.gt3d_joints + PL.gm3) * 3 : 0);
    g2raw[0] = p2[0], g2raw[1] = p2[1];
    g3raw[0] = p3[0], g3raw[1] = p3[1], g3raw[2] = p3[2];
  }
  // ---- LinearAE forward of the clip's frames (mlp_fwd_kernel's first-tile schedule); the last layer writes y^T to LDS ----
  TT(0, 1);
  tile_commit(S::dims(0), true, xr, H + S::h_off(0) * TP);
  {   // the next clip of this workgroup: its x tile waits in registers (rows beyond the batch read as zero, nothing is loaded)
    const int64_t nxt = (int64_t)clip + gridDim.x;
    tile_issue(m.x, nxt * T, nxt < d.B ? nxt * T + T : 0, S::dims(0), true, xr);
  }
  for_layers(sh, 0, nl, [&](int ll) {
#if P2C_TRAIN_DMA_IMAGE
    if constexpr (first) image_dma_wait(ll, L.wave);
    lds_barrier();
    TT(0, 2 + ll);
#else
    if constexpr (first) stage_commit(total4, wr, lds, 0, ll == 0 ? 0 : rounds_upto<S>(ll - 1), rounds_upto<S>(ll));
    lds_barrier();
    TT(0, 2 + ll);
    if constexpr (first) stage_issue(m.w_image, total4, wr, 0, issue_mark<S>(ll + 1), issue_mark<S>(ll + 2));
#endif
    const bool last = (ll == nl - 1);
    layer_forward(L, lds + S::w_off(ll), S::ld(ll), S::dims(ll), S::dims(ll + 1), !last, H + S::h_off(ll) * TP,
                  last ? Y : H + S::h_off(ll + 1) * TP, nullptr, false, false);
  });
  lds_barrier();
  TT(0, 8);

  // ---- pose head of frame t (this 32-lane group), forward + backward: pose_head_rot_bwd_tangent_tp<KIND, TRAIN> ----
  float y6[6] = {1.f, 0.f, 0.f, 0.f, 1.f, 0.f};              // identity rotation for idle lanes
  if (PL.active) {
#pragma unroll
    for (int i = 0; i < 6; ++i) y6[i] = Y[(PL.j * 6 + i) * TP + t];
  }
  const ph::V3 l = (PL.j < ph::J) ? ph::v3(lraw[0], lraw[1], lraw[2]) : ph::v3(0.f, 0.f, 0.f);
  ph::M3 Rref = ph::identity();
  if (K::SCAN && PL.j < ph::J) {
#pragma unroll
    for (int i = 0; i < 9; ++i) Rref.m[i] = rraw[i];
  }
  const float g2[2] = {PL.has2 ? g2raw[0] : 0.f, PL.has2 ? g2raw[1] : 0.f};
  const float g3[3] = {PL.has3 ? g3raw[0] : 0.f, PL.has3 ? g3raw[1] : 0.f, PL.has3 ? g3raw[2] : 0.f};
  if constexpr (first) {
    const float c0 = (int)threadIdx.x < d.B ? c0raw : 0.f, c1 = (int)threadIdx.x + NTH < d.B ? c1raw : 0.f;
    float cnt = c0 + c1;                      // small integers held in floats: exact in any order
    for (int i = threadIdx.x + 2 * NTH; i < d.B; i += NTH) cnt += m.counts[i];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
    if (L.lane == 0) scratch[L.wave] = cnt;
  }
  ph::SixD s;
  const ph::M3 c = ph::rot6d_fwd(y6, s);
  ph::M3 R = c;
  if (K::SCAN) R = ph::mul(scan_time2(c, t, PL.j, T, plane0, plane1), Rref);     // (its barriers also publish scratch[])
  else lds_barrier();
  TT(0, 9);
  if constexpr (first) {
    float n2 = 0.f;
    for (int w = 0; w < WAVES; ++w) n2 += scratch[w];
    ph::loss_coefs_n(d, gl, n2, ph::n3_elems(d), coef2, coef3);
  }
  ph::M3 A = R;
  ph::V3 x = l;
  ph::fk_doubling(PL, A, x);
  const ph::World W = ph::world_at(d, PL, t);
  ph::HeadAcc acc{0.f, 0.f, 0.f};
  ph::V3 F = ph::frame_head<ph::MODE_TRAIN>(d, PL, t, x, W, acc, coef2, coef3, nullptr, nullptr, g2, g3);
  TT(0, 10);
  {   // this clip's loss sums: per wave now, added in frame order at the very end (the barriers below publish red[])
    const float s2 = ph::wave_sum(acc.sum2), c2 = ph::wave_sum(acc.cnt2), s3 = ph::wave_sum(acc.sum3);
    if (L.lane == 0) red[L.wave * 3 + 0] = s2, red[L.wave * 3 + 1] = c2, red[L.wave * 3 + 2] = s3;
  }
  TT(0, 11);
  ph::V3 FX = ph::cross(F, x);
  ph::V3 PF = ph::v3(ph::group_prefix(F.x), ph::group_prefix(F.y), ph::group_prefix(F.z));
  ph::V3 PX = ph::v3(ph::group_prefix(FX.x), ph::group_prefix(FX.y), ph::group_prefix(FX.z));
  ph::V3 SubF = ph::shfl(PF, PL.base + PL.sub_end) - (PF - F);
  ph::V3 SubX = ph::shfl(PX, PL.base + PL.sub_end) - (PX - FX);
  ph::V3 tau = SubX - ph::cross(SubF, x);
  ph::V3 taup = ph::vmul(ph::vmulT(tau, A), R);   // tau A^T R
  ph::V3 g = taup;
  if (K::SCAN) {
    float *sb = plane0;
    lds_barrier();    // every group is done reading the scan planes
    sb[(t * ph::GROUP + PL.j) * 3 + 0] = taup.x, sb[(t * ph::GROUP + PL.j) * 3 + 1] = taup.y, sb[(t * ph::GROUP + PL.j) * 3 + 2] = taup.z;
    lds_barrier();
    ph::V3 Ssum = ph::v3(0.f, 0.f, 0.f);
    for (int tt = T - 1; tt >= t; --tt) {
      const float *q = sb + (tt * ph::GROUP + PL.j) * 3;
      Ssum = Ssum + ph::v3(q[0], q[1], q[2]);
    }
    const ph::M3 Rprev = (t > 0) ? ph::mulTN(c, R) : Rref;
    g = ph::vmulT(Ssum, Rprev);
  }
  TT(0, 12);
  if (PL.j < ph::J) {
    float gy6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};             // frames beyond T: no gradient
    if (PL.active) {
      if (s.c1 && s.c2) {
        ph::V3 b3 = ph::v3(c.m[6], c.m[7], c.m[8]);
        float al = ph::dot(g, s.b1), be = ph::dot(g, s.b2), ga = ph::dot(g, b3);
        float r1 = ph::frcp(s.n1), r2 = ph::frcp(s.n2);
        float k3 = (be + al * s.d * r2) * r1, k2 = -ga * r1, k5 = -al * r2;
        gy6[0] = fmaf(k3, b3.x, k2 * s.b2.x), gy6[1] = fmaf(k3, b3.y, k2 * s.b2.y), gy6[2] = fmaf(k3, b3.z, k2 * s.b2.z);
        gy6[3] = k5 * b3.x, gy6[4] = k5 * b3.y, gy6[5] = k5 * b3.z;
      } else {
        ph::M3 Gm;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          ph::V3 ci = ph::v3(c.m[i * 3], c.m[i * 3 + 1], c.m[i * 3 + 2]);
          ph::V3 h = ph::cross(ci, g) * 0.5f;
          Gm.m[i * 3] = h.x, Gm.m[i * 3 + 1] = h.y, Gm.m[i * 3 + 2] = h.z;
        }
        ph::rot6d_bwd(s, Gm, gy6);
      }
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) Y[(PL.j * 6 + i) * TP + t] = gy6[i];   // every y^T read happened before the scan's barriers
  } else if (PL.j == ph::J) {                                          // padding rows of G_L (the dgrad k loop reads them)
    for (int r = S::dims(nl); r < pad16(S::dims(nl)); ++r) Y[r * TP + t] = 0.f;
  }

  // ---- dgrad chain G_l = relu'(H_l) .* (W_l^T G_{l+1}), l = L-1 .. 1 ----
  TT(0, 13);
  for_layers_down(sh, nl - 1, 1, [&](int ll) {
    lds_barrier();
    TT(0, 14 + ll);
    layer_dgrad(L, lds + S::w_off(ll), S::ld(ll), S::dims(ll), S::dims(ll + 1), G + S::h_off(ll + 1) * TP, H + S::h_off(ll) * TP,
                G + S::h_off(ll) * TP);
  });
  lds_barrier();
  TT(0, 20);

  // ---- the clip's factors, as they sit in LDS (transposed, 16 samples = 64 B per row) ----
  f32x4 *fdst = reinterpret_cast<f32x4 *>(m.factors) + (size_t)clip * F_ROWS * 4;
  auto put = [&](const float *src, f32x4 *dst, int rows) {
    for (int i = threadIdx.x; i < rows * 4; i += NTH) {
      const int o = (i >> 2) * TP + (i & 3) * 4;
      dst[i] = (f32x4){src[o], src[o + 1], src[o + 2], src[o + 3]};
    }
  };
  for_layers(sh, 0, nl, [&](int ll) { put(H + S::h_off(ll) * TP, fdst + f_h_off(ll) * 4, S::dims(ll)); });
  for_layers(sh, 0, nl, [&](int ll) { put(G + S::h_off(ll + 1) * TP, fdst + f_g_off(ll + 1) * 4, S::dims(ll + 1)); });
  if (threadIdx.x == 0) {
    float a = 0.f, b = 0.f, cc = 0.f;
    for (int w = 0; w < WAVES; ++w) a += red[w * 3], b += red[w * 3 + 1], cc += red[w * 3 + 2];
    float *pp = d.partials + (size_t)clip * 4;
    pp[0] = a, pp[1] = b, pp[2] = cc, pp[3] = 0.f;
  }
  TT(0, 39);
  };   // one_clip

  if ((int)blockIdx.x < d.B) one_clip((int)blockIdx.x, std::true_type{});
  for (int clip = (int)blockIdx.x + (int)gridDim.x; clip < d.B; clip += (int)gridDim.x) one_clip(clip, std::false_type{});
}

// ---- second launch: weight gradient over all clips + optimizer + loss reduction ------------------------------------------
// Split-K over ALL CUs with an in-launch combine. Workgroup (t, q) -- q = blockIdx % 8, consecutive workgroups land on
// consecutive XCDs -- owns dW tile t and the clips st = q (mod 8): the ones train_clip_kernel's workgroups on the SAME XCD
// wrote a moment ago (speed only: plain stores keep the lines in that XCD's L2). Wave w walks clips q + 8 w, + 64, ... with
// one accumulator, the eight waves are added in w order through LDS (= mlp_wgrad_kernel). The slice's partial tile is
// PUBLISHED with write-through (sc1) stores; the workgroup whose ticket on the tile's counter is the last one reads the
// eight slices back with sc1 loads, adds them in q order (fixed: not arrival order -> bitwise reproducible and identical
// to mlp_reduce_small_kernel), applies AdamW and refreshes the packed image. Nobody waits on anybody: no spin, no
// residency assumption (cdna_hip_programming.md Guideline 16 / MI355X_MICROARCH.md "Valid forms", row 1: sc1 stores
// drained by the storing wave -> agent-scope atomic add by one lane -> the last adder loads sc1 after its add returned).
// The counters are zeroed by train_clip_kernel (previous launch on the stream) and re-armed by the last arriver.
// One extra workgroup runs loss_finalize's reduction.
constexpr int WG_WAVES = 8, KS = 8;
struct WgradArgs {
  const float *factors;
  int32_t n_stiles, n_tiles_w;
  float *gW[NLAY], *gb[NLAY];
  float *w_image;
  float *slices;                // (KS, n_tiles_w, 64, 4) partial tiles
  int32_t *counters;            // (n_tiles_w) arrival tickets, zero at launch
  const float *loss_partials;   // (B, 4) from train_clip_kernel
  float n3_elems;
  int32_t has2d, has3d;
  float *loss_sums, *losses;
};

__device__ __forceinline__ void finalize_losses(const WgradArgs &a, double (*sh)[256]) {   // = p2c::loss_finalize
  double s0 = 0.0, s1 = 0.0, s2 = 0.0;
  // one 16-byte load per clip, eight in flight per thread, added in index order (the sums are the same doubles as a plain loop's:
  // same terms, same order per thread). With one dependent round trip per clip this one workgroup was the long pole of the
  // launch from a few thousand clips on (B = 8192: 20 us).
  const f32x4 *lp = reinterpret_cast<const f32x4 *>(a.loss_partials);
  for (int i0 = threadIdx.x; i0 < a.n_stiles; i0 += 8 * 256) {
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + u * 256;
      v[u] = lp[i < a.n_stiles ? i : 0];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (i0 + u * 256 < a.n_stiles) s0 += (double)v[u][0], s1 += (double)v[u][1], s2 += (double)v[u][2];
  }
  sh[0][threadIdx.x] = s0, sh[1][threadIdx.x] = s1, sh[2][threadIdx.x] = s2;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      sh[0][threadIdx.x] += sh[0][threadIdx.x + s];
      sh[1][threadIdx.x] += sh[1][threadIdx.x + s];
      sh[2][threadIdx.x] += sh[2][threadIdx.x + s];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double q2 = sh[0][0], n2 = sh[1][0], q3 = sh[2][0];
    a.loss_sums[0] = (float)q2, a.loss_sums[1] = (float)n2, a.loss_sums[2] = (float)q3, a.loss_sums[3] = a.n3_elems;
    const float nan = __builtin_nanf("");
    const float l2 = a.has2d ? (float)(q2 / (2.0 * n2)) : nan;
    const float l3 = a.has3d ? (float)(q3 / (double)a.n3_elems) : nan;
    a.losses[0] = l2, a.losses[1] = l3, a.losses[2] = l2 + l3;
  }
}

template <bool ADAM>
__global__ __launch_bounds__(64 * WG_WAVES) void train_wgrad_kernel(const WgradArgs a, const p2c_adamw_desc o) {
  __shared__ f32x4 red[WG_WAVES][64];
  __shared__ p2c_optim::Coefs sc;
  __shared__ int sc_ready;
  if (threadIdx.x == 0) sc_ready = 0;        // (published by the barrier behind the contraction)
  if ((int)blockIdx.x == a.n_tiles_w * KS) {   // the loss reduction rides on this launch (256 of the 512 threads)
    __shared__ double fin[3][256];
    if (threadIdx.x >= 256) return;            // whole waves leave: the barriers below count the remaining four
    finalize_losses(a, fin);
    return;
  }
  TB_DECL;
  TB(0);
  const int lane = threadIdx.x & 63, r = lane & 15, k = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q = blockIdx.x % KS, t = blockIdx.x / KS;
  int32_t dd[NLAY + 1];
#pragma unroll
  for (int i = 0; i <= NLAY; ++i) dd[i] = S::dim_at(i);
  const TileRef tr = locate_tile(dd, t);
  const int n_in = dd[tr.l], n_out = dd[tr.l + 1];
  const int n = tr.ntile * 16 + r, mm = tr.mtile * 16 + r;
  const bool a_ok = n < n_out, b_ok = mm < n_in, b_one = mm == n_in;
  int g_row = F_HALF, h_row = 0;                               // first factor row of G_{l+1} / H_l
  for (int i = 1; i <= tr.l; ++i) g_row += dd[i];
  for (int i = 0; i < tr.l; ++i) h_row += dd[i];
  const size_t a_off = (size_t)(g_row + n) * 16 + 4 * k, b_off = (size_t)(h_row + mm) * 16 + 4 * k;
  const size_t f_tile = (size_t)F_ROWS * 16;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f}, ones = {1.f, 1.f, 1.f, 1.f};
  float step = 0.f;
  if (ADAM) step = *o.step + 1.f;            // read at kernel start: before any workgroup can publish the new count

  auto load_a = [&](int st) -> f32x4 {
    return a_ok ? *reinterpret_cast<const f32x4 *>(a.factors + (size_t)st * f_tile + a_off) : zero;
  };
  auto load_b = [&](int st) -> f32x4 {
    return b_one ? ones : (b_ok ? *reinterpret_cast<const f32x4 *>(a.factors + (size_t)st * f_tile + b_off) : zero);
  };
  f32x4 acc = zero;
  constexpr int UNROLL = 4;
  const int stride = KS * WG_WAVES;
  int st = q + KS * wave;
  for (; st + (UNROLL - 1) * stride < a.n_stiles; st += UNROLL * stride) {   // UNROLL clips in flight
    f32x4 av[UNROLL], bv[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) av[u] = load_a(st + u * stride), bv[u] = load_b(st + u * stride);
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][i], bv[u][i], acc, 0, 0, 0);
    }
  }
  for (; st < a.n_stiles; st += stride) {
    const f32x4 av = load_a(st), bv = load_b(st);
#pragma unroll
    for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[i], acc, 0, 0, 0);
  }
  TB(1);
  red[wave][lane] = acc;
  __syncthreads();
  if (wave == 1 && ADAM) {                   // the fp64 bias corrections of this step: off wave 0's critical path
    if (lane == 0) {
      sc = p2c_optim::coefs(o, step);
      __hip_atomic_store(&sc_ready, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    return;
  }
  if (wave != 0) return;
  f32x4 s = red[0][lane];
#pragma unroll
  for (int w = 1; w < WG_WAVES; ++w) s += red[w][lane];
  // what only the tile's last arriver will need is requested now by every slice (a few KB of L2 hits each): the latency
  // hides behind the publish / ticket round trips. Nobody else writes these words during this launch.
  const int m = tr.mtile * 16 + (lane & 15);
  float *gp[4];
  float pv[4], mv[4], vv[4];
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int nn = tr.ntile * 16 + 4 * (lane >> 4) + rr;
    gp[rr] = (nn < n_out && m <= n_in) ? ((m < n_in) ? a.gW[tr.l] + nn * n_in + m : a.gb[tr.l] + nn) : nullptr;
    pv[rr] = mv[rr] = vv[rr] = 0.f;
    if (ADAM && gp[rr]) {
      const ptrdiff_t off = gp[rr] - o.grad;
      pv[rr] = o.param[off], mv[rr] = o.exp_avg[off], vv[rr] = o.exp_avg_sq[off];
    }
  }
  // ---- publish this slice's partial tile (16-byte write-through stores), draw the tile's ticket ----
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.slices, 0, KS * a.n_tiles_w * 1024, 0x00020000);
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const int my_off = (q * a.n_tiles_w + t) * 1024 + lane * 16;
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, s), rs, my_off, 0, 16);      // aux 16 = sc1
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // the storing wave drains before it signals
  int ticket = 0;
  if (lane == 0) ticket = __hip_atomic_fetch_add(a.counters + t, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  ticket = __builtin_amdgcn_readfirstlane(ticket);                 // (the returned value is used: the add has completed)
  TB(2);
  if (ticket != KS - 1) return;
  // ---- last arriver of tile t: every slice is published. Sum in q order, update (= mlp_reduce_small_kernel) ----
  f32x4 v[KS];
#pragma unroll
  for (int qq = 0; qq < KS; ++qq)                                  // sc1 loads: past this CU's L1
    v[qq] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (qq * a.n_tiles_w + t) * 1024 + lane * 16, 0, 16));
  s = v[0];
#pragma unroll
  for (int qq = 1; qq < KS; ++qq) s += v[qq];
  TB(3);
  p2c_optim::Coefs c;
  if (ADAM) {                                // wave 1 of THIS workgroup wrote them long ago (it waits for nobody)
    while (__hip_atomic_load(&sc_ready, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) __builtin_amdgcn_s_sleep(1);
    c = sc;
  }
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    if (!gp[rr]) continue;
    *gp[rr] = (ADAM && o.zero_grad) ? 0.f : s[rr];   // zero_grad: the optimizer leaves the gradient buffer zeroed
    if (ADAM) {
      const ptrdiff_t off = gp[rr] - o.grad;
      if (o.adamw) p2c_optim::update<true>(c, pv[rr], s[rr], mv[rr], vv[rr]);
      else p2c_optim::update<false>(c, pv[rr], s[rr], mv[rr], vv[rr]);
      o.param[off] = pv[rr], o.exp_avg[off] = mv[rr], o.exp_avg_sq[off] = vv[rr];
      const int nn = tr.ntile * 16 + 4 * (lane >> 4) + rr;
      if (a.w_image) a.w_image[image_w_off(tr.l) + nn * image_ld(tr.l) + m] = pv[rr];    // bias sits in column n_in == m
    }
  }
  TB(4);
  if (lane == 0) {
    __hip_atomic_store(a.counters + t, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-armed for the next launch
    if (ADAM && atomicAdd(o.ticket, 1) == a.n_tiles_w - 1) {   // the last tile to finish publishes the new step count
      *o.step = step;
      *o.ticket = 0;
    }
  }
  TB(39);
  TB_DUMP(t == P2C_TRAIN_TRACE_TILE);
}

// ---- second launch, throughput form (large batches): stream every clip's factor block ONCE ------------------------------------
// train_wgrad_kernel gives every dW tile its own workgroups: each reads the two 1 KB operand blocks of its tile from every
// clip -- 158 KB per clip in all for a 34 KB factor block. That is the right trade at a few hundred clips (632 workgroups,
// every CU busy, one launch incl. the optimizer); at thousands of clips the launch is bound by that redundant operand
// traffic (B = 8192: 114 us for 8.7 GFLOP). Here a persistent workgroup owns ALL 79 tiles for a strided slice of the clips:
// its eight waves split the tiles in dense blocks of (G row-tiles) x (H row-tiles) -- waves 0-4 two of layer 5's ten
// n-tiles against its five m-tiles each (ten accumulators from seven 16-byte loads per lane and clip), wave 5 / 6 layer 4
// (+ layer 3), wave 7 layers 0-2 -- so a clip's factors cross HBM once, the accumulators stay in registers over the whole
// slice, and a workgroup leaves one 79 KB partial. wgrad_reduce_kernel adds the partials in workgroup order (fixed: bitwise
// reproducible), applies AdamW, refreshes the weight image and finishes the losses.
constexpr int WS_BLOCKS_MAX = 256;
// Every factor row enters the CU ONCE: the eight waves copy a clip's 34 KB block into an LDS slot by LDS-DMA (buffer_load ... lds,
// 1 KB pieces dealt round-robin), three slots in a ring, and take their operand fragments from there (ds_read_b128: a fragment's
// 16 rows x 16 bytes are contiguous per sample quarter). Loaded straight into registers (first version) the H rows of a layer were
// fetched by every wave that owns a tile column of it -- 61 KB per clip through a CU's vector-memory path (about 11 B/clk) for a
// 34 KB block: 77 us at B = 8192, the same at any prefetch depth and twice that on half the CUs.
constexpr int WS_SLOT_PIECES = (F_ROWS * 64 + 1023) / 1024, WS_SLOT_FLOATS = WS_SLOT_PIECES * 256, WS_SLOTS = 3;   // (four slots: 76 vs 74 us at B = 8192)
static_assert(WS_SLOTS * WS_SLOT_FLOATS * 4 <= 160 * 1024, "LDS budget of the factor ring");
typedef __attribute__((address_space(3))) void *ws_lds_ptr;

struct RowBlock {          // one dense block: n-tiles [n0, n0 + NA) x m-tiles [0, NB) of layer l; its first dW tile index
  int l, n0, tile0;
};
template <int NA, int NB>
struct BlockState {
  int a_off[NA], b_off[NB];                 // float offsets of the lane's fragments inside a slot
  bool a_ok[NA], b_ok[NB], b_one[NB];
  f32x4 acc[NA * NB];
  int tile0, mtiles;
  __device__ __forceinline__ void init(const RowBlock rb, const int lane) {
    const int r = lane & 15, k = lane >> 4;
    int32_t dd[NLAY + 1];
#pragma unroll
    for (int i = 0; i <= NLAY; ++i) dd[i] = S::dim_at(i);
    const int n_in = dd[rb.l], n_out = dd[rb.l + 1];
    int g_row = F_HALF, h_row = 0;                               // first factor row of G_{l+1} / H_l
    for (int i = 1; i <= rb.l; ++i) g_row += dd[i];
    for (int i = 0; i < rb.l; ++i) h_row += dd[i];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int n = (rb.n0 + i) * 16 + r;
      a_ok[i] = n < n_out, a_off[i] = (g_row + (a_ok[i] ? n : 0)) * 16 + 4 * k;
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int mm = j * 16 + r;
      b_ok[j] = mm < n_in, b_one[j] = mm == n_in, b_off[j] = (h_row + (b_ok[j] ? mm : 0)) * 16 + 4 * k;
    }
#pragma unroll
    for (int i = 0; i < NA * NB; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    tile0 = rb.tile0, mtiles = (n_in + 1 + 15) >> 4;
  }
  __device__ __forceinline__ void fma(const float *slot) {
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f}, ones = {1.f, 1.f, 1.f, 1.f};
    f32x4 av[NA], bv[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) av[i] = *reinterpret_cast<const f32x4 *>(slot + a_off[i]);
#pragma unroll
    for (int j = 0; j < NB; ++j) bv[j] = *reinterpret_cast<const f32x4 *>(slot + b_off[j]);
#pragma unroll
    for (int i = 0; i < NA; ++i) av[i] = a_ok[i] ? av[i] : zero;
#pragma unroll
    for (int j = 0; j < NB; ++j) bv[j] = b_one[j] ? ones : (b_ok[j] ? bv[j] : zero);
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[i * NB + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i][u], bv[j][u], acc[i * NB + j], 0, 0, 0);
  }
  // the block's partial tiles, in MFMA C layout (one 16-byte store per lane and tile)
  __device__ __forceinline__ void store(const WgradArgs &a, const int lane) const {
    // tile-major: the partials of one tile from all workgroups are contiguous (gridDim KB): the reduction streams them
    f32x4 *out = reinterpret_cast<f32x4 *>(a.slices) + ((size_t)tile0 * gridDim.x + blockIdx.x) * 64 + lane;
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j) out[(size_t)(i * mtiles + j) * gridDim.x * 64] = acc[i * NB + j];
  }
};

// the clips of this workgroup through the LDS ring; compute(slot) is the wave's own tile work on one resident clip
template <class F>
__device__ __forceinline__ void ws_clip_loop(const WgradArgs &a, float *ring, const int first, const int step, const int end, const int wave,
                                             const int lane, F &&compute) {
  const int my_pieces = (WS_SLOT_PIECES - wave + WG_WAVES - 1) / WG_WAVES;        // pieces wave, wave + 8, ...
  auto issue = [&](int clip, int slot) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(a.factors) + (size_t)clip * F_ROWS * 16, 0, F_ROWS * 64, 0x00020000);
    const unsigned base = (unsigned)(uintptr_t)(ws_lds_ptr)(ring + slot * WS_SLOT_FLOATS);
#pragma unroll
    for (int i = 0; i < (WS_SLOT_PIECES + WG_WAVES - 1) / WG_WAVES; ++i) {
      const int pc = wave + i * WG_WAVES;                            // (wave-uniform; the tail of the last piece reads past the records: zeros)
      if (pc < WS_SLOT_PIECES)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (ws_lds_ptr)(uintptr_t)(base + pc * 1024), 16, (pc * 64 + lane) * 16, 0, 0, 0);
    }
  };
#pragma unroll
  for (int s = 0; s < WS_SLOTS - 1; ++s)
    if (first + s * step < end) issue(first + s * step, s);
  int slot = 0;
  for (int clip = first; clip < end; clip += step) {
    // this wave's pieces of THIS clip have landed once only the later clips' are outstanding (vector-memory operations retire in
    // issue order: my_pieces per clip still in flight); then the barrier: everybody's have
    int later = 0;                                  // clips already requested behind this one (at most WS_SLOTS - 2)
#pragma unroll
    for (int s = 1; s <= WS_SLOTS - 2; ++s) later += (clip + s * step < end) ? 1 : 0;
    switch (later * my_pieces) {
      case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
      case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
      case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
      case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
      case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    compute(ring + slot * WS_SLOT_FLOATS);
    // the slot WS_SLOTS - 1 ahead (= the one behind) was read in the previous iteration: every wave finished with it before it
    // passed the barrier above
    const int ahead = slot >= 1 ? slot - 1 : WS_SLOTS - 1;
    if (clip + (WS_SLOTS - 1) * step < end) issue(clip + (WS_SLOTS - 1) * step, ahead);
    slot = slot + 1 == WS_SLOTS ? 0 : slot + 1;
  }
}
static_assert(WS_SLOTS >= 3 && WS_SLOTS <= 5, "the counted waits above: up to three clips behind the current one");
static_assert((WS_SLOT_PIECES + WG_WAVES - 1) / WG_WAVES == 5 && WS_SLOT_PIECES / WG_WAVES == 4, "the counted waits above: four or five pieces per wave");

__global__ __launch_bounds__(64 * WG_WAVES) void wgrad_stream_kernel(const WgradArgs a) {
  extern __shared__ float ws_ring[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int first = blockIdx.x, step = gridDim.x, end = a.n_stiles;      // clips b, b + grid, ...
  // tile index of (layer l, n-tile n, m-tile 0): layers 0..5 hold 8, 2, 1, 3, 15, 50 tiles
  constexpr int T0 = 0, T1 = 8, T2 = 10, T3 = 11, T4 = 14, T5 = 29;
  if (wave < 5) {
    BlockState<2, 5> b0;
    b0.init(RowBlock{5, 2 * wave, T5 + 2 * wave * 5}, lane);
    ws_clip_loop(a, ws_ring, first, step, end, wave, lane, [&](const float *slot) { b0.fma(slot); });
    b0.store(a, lane);
  } else if (wave == 5) {
    BlockState<3, 3> b0;
    b0.init(RowBlock{4, 0, T4}, lane);
    ws_clip_loop(a, ws_ring, first, step, end, wave, lane, [&](const float *slot) { b0.fma(slot); });
    b0.store(a, lane);
  } else if (wave == 6) {
    BlockState<2, 3> b0;
    BlockState<3, 1> b1;
    b0.init(RowBlock{4, 3, T4 + 9}, lane), b1.init(RowBlock{3, 0, T3}, lane);
    ws_clip_loop(a, ws_ring, first, step, end, wave, lane, [&](const float *slot) { b0.fma(slot), b1.fma(slot); });
    b0.store(a, lane), b1.store(a, lane);
  } else {
    BlockState<2, 4> b0;
    BlockState<1, 2> b1;
    BlockState<1, 1> b2;
    b0.init(RowBlock{0, 0, T0}, lane), b1.init(RowBlock{1, 0, T1}, lane), b2.init(RowBlock{2, 0, T2}, lane);
    ws_clip_loop(a, ws_ring, first, step, end, wave, lane, [&](const float *slot) { b0.fma(slot), b1.fma(slot), b2.fma(slot); });
    b0.store(a, lane), b1.store(a, lane), b2.store(a, lane);
  }
}

// grad = sum over the stream kernel's workgroups of their partial tiles, in workgroup order; the optimizer step and the packed
// image ride on it (= mlp_reduce_kernel of p2c_mlp.hip on this launch's arguments); the extra workgroup finishes the losses
constexpr int RED_G = 32, RED_L = 8;     // groups of partials added in parallel (256 partials: one round of eight loads per thread); lanes of a tile per workgroup
template <bool ADAM>
__global__ __launch_bounds__(RED_L * RED_G) void wgrad_reduce_kernel(const WgradArgs a, const int n_blocks, const p2c_adamw_desc o) {
  __shared__ f32x4 red[RED_G][RED_L];
  __shared__ p2c_optim::Coefs sc;
  if ((int)blockIdx.x == a.n_tiles_w * (64 / RED_L)) {
    __shared__ double fin[3][256];
    finalize_losses(a, fin);
    return;
  }
  const int t = blockIdx.x / (64 / RED_L), li = threadIdx.x % RED_L, q = threadIdx.x / RED_L;
  const int lane = (blockIdx.x % (64 / RED_L)) * RED_L + li;
  float step = 0.f;
  if (ADAM) {
    step = *o.step + 1.f;
    if (threadIdx.x == RED_L * RED_G - 1) sc = p2c_optim::coefs(o, step);
  }
  int32_t dd[NLAY + 1];
#pragma unroll
  for (int i = 0; i <= NLAY; ++i) dd[i] = S::dim_at(i);
  const TileRef tr = locate_tile(dd, t);
  const int n_in = dd[tr.l], n_out = dd[tr.l + 1];
  const int m = tr.mtile * 16 + (lane & 15);
  float *gp[4];
  float pv[4], mv[4], vv[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int n = tr.ntile * 16 + 4 * (lane >> 4) + r;
    gp[r] = (q == 0 && n < n_out && m <= n_in) ? ((m < n_in) ? a.gW[tr.l] + n * n_in + m : a.gb[tr.l] + n) : nullptr;
    pv[r] = mv[r] = vv[r] = 0.f;
    if (ADAM && gp[r]) {
      const ptrdiff_t off = gp[r] - o.grad;
      pv[r] = o.param[off], mv[r] = o.exp_avg[off], vv[r] = o.exp_avg_sq[off];
    }
  }
  const size_t stride = 64;                              // tile-major partials: workgroup w's copy of tile t at (t n_blocks + w) KB
  const f32x4 *p = reinterpret_cast<const f32x4 *>(a.slices) + (size_t)t * n_blocks * 64 + lane;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  int w = q;
  for (; w + 7 * RED_G < n_blocks; w += 8 * RED_G) {     // eight loads in flight, added in workgroup order
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(&p[(size_t)(w + u * RED_G) * stride]);
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; w < n_blocks; w += RED_G) s += __builtin_nontemporal_load(&p[(size_t)w * stride]);
  red[q][li] = s;
  __syncthreads();
  if (q == 0) {
#pragma unroll
    for (int i = 1; i < RED_G; ++i) s += red[i][li];
    p2c_optim::Coefs cf;
    if (ADAM) cf = sc;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (!gp[r]) continue;
      *gp[r] = (ADAM && o.zero_grad) ? 0.f : s[r];
      if (ADAM) {
        const ptrdiff_t off = gp[r] - o.grad;
        if (o.adamw) p2c_optim::update<true>(cf, pv[r], s[r], mv[r], vv[r]);
        else p2c_optim::update<false>(cf, pv[r], s[r], mv[r], vv[r]);
        o.param[off] = pv[r], o.exp_avg[off] = mv[r], o.exp_avg_sq[off] = vv[r];
        const int n = tr.ntile * 16 + 4 * (lane >> 4) + r;
        if (a.w_image) a.w_image[image_w_off(tr.l) + n * image_ld(tr.l) + m] = pv[r];
      }
    }
  }
  if (ADAM) {
    __syncthreads();
    if (threadIdx.x == 0 && atomicAdd(o.ticket, 1) == a.n_tiles_w * (64 / RED_L) - 1) {
      *o.step = step;
      *o.ticket = 0;
    }
  }
}

// per-clip count of the 2-D target pairs the loss will not mask (utils/tensors.py:29-40, loss/loc_2d.py:69-89): a property
// of the targets alone, computed once per batch
__global__ __launch_bounds__(256) void count_pairs_kernel(const p2c_pose_head_desc d, float *counts) {
  __shared__ float sh[4];
  const int clip = blockIdx.x;
  float c = 0.f;
  for (int i = threadIdx.x; i < d.T * ph::J; i += 256) {
    const int t = i / ph::J, j = i - t * ph::J;
    const int gm = d.gmap2d[j];
    if (t < d.t0 || t >= d.t1 || gm < 0 || !d.gt2d) continue;
    const float *g = d.gt2d + (((size_t)clip * d.T + t) * d.gt2d_joints + gm) * d.gt2d_channels;
    c += (!d.mask_missing_joints || j == d.hips_lane || ((g[0] != 0.f) && (g[1] != 0.f))) ? 1.f : 0.f;
  }
  c = ph::wave_sum(c);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) counts[clip] = sh[0] + sh[1] + sh[2] + sh[3];
}

}  // namespace p2c_train

using namespace p2c_train;

static bool shape_is_linear_ae156(const p2c_mlp_desc &m) {
  if (m.n_layers != NLAY) return false;
  for (int l = 0; l <= NLAY; ++l)
    if (m.dims[l] != S::dim_at(l)) return false;
  return true;
}

#ifdef P2C_TRAIN_TRACE
extern "C" P2C_API int p2c_debug_train_trace(unsigned long long *out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(p2c_train::g_ttrace), sizeof(unsigned long long) * 80);
}
#endif

extern "C" int p2c_train_step_supported(const p2c_train_step_desc *d) {
  if (!d) return 0;
  const p2c_pose_head_desc &h = d->head;
  if (h.kind != P2C_KIND_POSE_CHANGES_6D && h.kind != P2C_KIND_RELATIVE_ROT_6D) return 0;
  if (h.T < 1 || h.T > T_MAX || h.B < 1) return 0;
  if (!shape_is_linear_ae156(d->mlp) || d->mlp.N != (int64_t)h.B * h.T) return 0;
  return 1;
}

// batches from which the first launch takes its throughput form (P2C_STREAM_MIN_B; p2c_train_step_set_stream_min_batch)
static int g_stream_min_b = -1;
static int stream_min_b() {
  if (g_stream_min_b < 0) {
    const char *e = getenv("P2C_STREAM_MIN_B");
    g_stream_min_b = e ? atoi(e) : 257;       // measured (tools/step_sweep.py, whole step): 320 clips 35.5 vs 42.1 us, 512: 37.7 vs 45.8,
                                              // 768: 47.6 vs 60.1, 1024: 52.7 vs 76.0 -- past one clip per CU the pair form wins at once
  }
  return g_stream_min_b;
}
// batches from which the second launch streams the factors once (P2C_WGRAD_STREAM_MIN_B)
static int g_wgrad_stream_min_b = -1;
static int wgrad_stream_min_b() {
  if (g_wgrad_stream_min_b < 0) {
    const char *e = getenv("P2C_WGRAD_STREAM_MIN_B");
    g_wgrad_stream_min_b = e ? atoi(e) : 3072;   // measured: 2048 clips 34 vs 31 us, 4096 clips 45 vs 56 us, 8192 clips 86 vs 114 us
  }
  return g_wgrad_stream_min_b;
}
extern "C" P2C_API int p2c_train_step_set_wgrad_stream_min_batch(int32_t min_b) {
  const int prev = wgrad_stream_min_b();
  if (min_b >= 0) g_wgrad_stream_min_b = min_b;
  return prev;
}
extern "C" P2C_API int p2c_train_step_set_stream_min_batch(int32_t min_b) {
  const int prev = stream_min_b();
  if (min_b >= 0) g_stream_min_b = min_b;
  return prev;
}

static constexpr int kClipBlocks = 256;   // CUs of an MI355X: one 156 KB-LDS workgroup each
static constexpr int kTilesW = 79;    // 16x16 tiles of the augmented weight gradients of LinearAE156 (checked below)
// workspace: [factors (B, F_ROWS, 16) | slices (KS, tiles, 256) | counters (tiles, padded to 128 ints)]
extern "C" int64_t p2c_train_step_workspace_floats(const p2c_train_step_desc *d) {
  if (!p2c_train_step_supported(d)) return 0;
  return (int64_t)d->head.B * F_ROWS * 16 + (int64_t)WS_BLOCKS_MAX * kTilesW * 256 + 128;   // [factors | partial tiles | tickets]
}

extern "C" int p2c_count_target_pairs(const p2c_pose_head_desc *desc, float *counts, void *stream_) {
  if (!desc || !counts) return P2C_E_NULL;
  p2c_pose_head_desc d = *desc;
  if (!d.y) d.y = reinterpret_cast<const float *>(counts);        // not read: only the target-side fields matter here
  float dummy = 0.f;
  if (!d.partials) d.partials = &dummy;
  if (!d.loss_sums) d.loss_sums = &dummy;
  if (!d.losses) d.losses = &dummy;
  int rc = p2c_internal_validate_pose_head(&d);
  if (rc) return rc;
  hipLaunchKernelGGL(count_pairs_kernel, dim3((unsigned)d.B), dim3(256), 0, (hipStream_t)stream_, d, counts);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

extern "C" int p2c_train_step(const p2c_train_step_desc *desc, const float *const grad_losses_[3], void *stream_) {
  return p2c_train_step_launch(desc, grad_losses_, 3, stream_);
}

extern "C" int p2c_train_step_launch(const p2c_train_step_desc *desc, const float *const grad_losses_[3], int32_t which,
                                     void *stream_) {
  if (!desc) return P2C_E_NULL;
  if ((which & 3) == 0) return P2C_E_ENUM;
  if (!p2c_train_step_supported(desc)) return P2C_E_SHAPE;
  p2c_pose_head_desc d = desc->head;
  const p2c_mlp_desc &m = desc->mlp;
  if (!m.x || !m.w_image || !m.partials || !desc->pair_counts) return P2C_E_NULL;
  d.y = m.x;                                                      // (validation wants a model output; the kernel never reads it)
  if (d.out_pose_changes || d.out_projection_2d || d.out_projection_2d_transformed || d.out_shift || d.out_scale ||
      d.out_relative_pose_loc || d.out_relative_pose_rot || d.out_absolute_pose_loc || d.out_absolute_pose_rot ||
      d.out_world_loc || d.out_world_rot)
    return P2C_E_ENUM;                                            // lean outputs only
  if (d.gt_rot) return P2C_E_ENUM;                                // rot_3d runs through p2c_pose_head_fwd / _bwd
  int rc = p2c_internal_validate_pose_head(&d);
  if (rc) return rc;
  if ((reinterpret_cast<uintptr_t>(m.x) & 15) || (reinterpret_cast<uintptr_t>(m.w_image) & 15) ||
      (reinterpret_cast<uintptr_t>(m.partials) & 15))
    return P2C_E_SHAPE;
  for (int l = 0; l < NLAY; ++l)
    if (!m.gW[l] || !m.gb[l] || !m.W[l] || !m.b[l]) return P2C_E_NULL;
  const bool adam = m.fused_adamw != nullptr;
  p2c_adamw_desc o{};
  if (adam) {
    o = *m.fused_adamw;
    if (!o.param || !o.grad || !o.exp_avg || !o.exp_avg_sq || !o.step || !o.ticket || !o.hyper) return P2C_E_NULL;
    int64_t n_params = 0;
    for (int l = 0; l < NLAY; ++l) {
      const float *lo = o.grad, *hi = o.grad + o.n;
      if (m.gW[l] < lo || m.gW[l] + (size_t)m.dims[l + 1] * m.dims[l] > hi || m.gb[l] < lo || m.gb[l] + m.dims[l + 1] > hi)
        return P2C_E_INDEX;
      n_params += (int64_t)m.dims[l + 1] * (m.dims[l] + 1);
    }
    if (n_params != o.n) return P2C_E_SHAPE;                      // the MLP must be ALL the optimizer optimises (step counter)
  }
  hipStream_t stream = (hipStream_t)stream_;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void *)train_clip_kernel<P2C_KIND_POSE_CHANGES_6D>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)train_clip_kernel<P2C_KIND_RELATIVE_ROT_6D>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  if (!m.skip_pack && (which & 1)) {
    rc = p2c_mlp_pack(&m, stream_);
    if (rc) return rc;
  }
  ph::GradLosses gl{{nullptr, nullptr, nullptr}};
  if (grad_losses_)
    for (int i = 0; i < 3; ++i) gl.p[i] = grad_losses_[i];
  int tiles = 0;
  for (int l = 0; l < NLAY; ++l) tiles += ((m.dims[l + 1] + 15) / 16) * ((m.dims[l] + 1 + 15) / 16);
  if (tiles != kTilesW) return P2C_E_SHAPE;
  float *slices = m.partials + (size_t)d.B * F_ROWS * 16;
  int32_t *counters = reinterpret_cast<int32_t *>(slices + (size_t)WS_BLOCKS_MAX * kTilesW * 256);
  int identity = 1;
  for (int j = 0; j < P2C_JOINTS; ++j) identity &= (d.gmap2d[j] == j) && (d.gmap3d[j] == j);
  ClipArgs ca{m.x, m.w_image, desc->pair_counts, m.partials, counters, tiles, identity};
  const size_t lds_a = (size_t)LDS_FLOATS * sizeof(float);
  hipError_t e = hipSuccess;
  const dim3 grid_a((unsigned)(d.B < kClipBlocks ? d.B : kClipBlocks));   // persistent: one workgroup per CU walks its clips
  if ((which & 1) && d.B >= stream_min_b() && p2c_internal_train_stream_supported(d)) {
    // several clips per CU: the throughput form (one wavefront per clip, p2c_train_stream.hip)
    rc = p2c_internal_train_stream_launch(d, gl, ca, stream);
    if (rc) return rc;
  } else if (which & 1) {
    if (d.kind == P2C_KIND_POSE_CHANGES_6D)
      hipLaunchKernelGGL(train_clip_kernel<P2C_KIND_POSE_CHANGES_6D>, grid_a, dim3(64 * WAVES), lds_a, stream, d, gl, ca);
    else
      hipLaunchKernelGGL(train_clip_kernel<P2C_KIND_RELATIVE_ROT_6D>, grid_a, dim3(64 * WAVES), lds_a, stream, d, gl, ca);
    e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
  }
  if (!(which & 2)) return 0;

  WgradArgs wa{};
  wa.factors = m.partials, wa.n_stiles = d.B, wa.n_tiles_w = tiles;
  for (int l = 0; l < NLAY; ++l) wa.gW[l] = m.gW[l], wa.gb[l] = m.gb[l];
  wa.w_image = adam ? m.w_image : nullptr;
  wa.slices = slices, wa.counters = counters;
  wa.loss_partials = d.partials;
  wa.n3_elems = (float)((double)d.B * (double)(d.t1 - d.t0) * (double)d.n_common3d * 3.0);
  wa.has2d = d.gt2d ? 1 : 0, wa.has3d = d.gt3d ? 1 : 0;
  wa.loss_sums = d.loss_sums, wa.losses = d.losses;
  if (d.B >= wgrad_stream_min_b()) {
    // thousands of clips: every factor block crosses HBM once (wgrad_stream_kernel), the per-workgroup partials are added by a third launch
    int blocks = d.B / 8;                                         // at least eight clips per workgroup; a multiple of 8 (XCD affinity)
    static const int max_blocks = getenv("P2C_WGRAD_STREAM_BLOCKS") ? atoi(getenv("P2C_WGRAD_STREAM_BLOCKS")) : WS_BLOCKS_MAX;   // (A/B timing)
    const int cap = max_blocks < WS_BLOCKS_MAX ? (max_blocks < 8 ? 8 : max_blocks) : WS_BLOCKS_MAX;
    blocks = blocks > cap ? cap : (blocks & ~7);
    if (blocks < 8) blocks = 8;
    static bool ws_attr = false;
    if (!ws_attr) {
      (void)hipFuncSetAttribute((const void *)wgrad_stream_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, WS_SLOTS * WS_SLOT_FLOATS * 4);
      (void)hipGetLastError();
      ws_attr = true;
    }
    hipLaunchKernelGGL(wgrad_stream_kernel, dim3((unsigned)blocks), dim3(64 * WG_WAVES), WS_SLOTS * WS_SLOT_FLOATS * 4, stream, wa);
    e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    const dim3 grid_r((unsigned)(tiles * (64 / RED_L) + 1));
    if (adam) hipLaunchKernelGGL(wgrad_reduce_kernel<true>, grid_r, dim3(RED_L * RED_G), 0, stream, wa, blocks, o);
    else hipLaunchKernelGGL(wgrad_reduce_kernel<false>, grid_r, dim3(RED_L * RED_G), 0, stream, wa, blocks, o);
    e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
  }
  const dim3 grid_b((unsigned)(tiles * KS + 1));
  if (adam) hipLaunchKernelGGL(train_wgrad_kernel<true>, grid_b, dim3(64 * WG_WAVES), 0, stream, wa, o);
  else hipLaunchKernelGGL(train_wgrad_kernel<false>, grid_b, dim3(64 * WG_WAVES), 0, stream, wa, o);
  e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}
