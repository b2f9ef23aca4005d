// p2c_pose_head_chain.hip -- large-batch pose head (6-D kinds, lean outputs, identity world): the "chain-lane" mapping.
//
// The joint-lane kernels of p2c_pose_head.hip give every bone its own lane: 26 of 32 lanes work, forward kinematics is
// three pointer-doubling rounds (three 3x3 compositions per bone instead of one) and every frame pays ~40 cross-lane
// moves. At large B those kernels are VALU-issue-bound with exactly algorithmic HBM traffic (a wave64
// VALU instruction holds its SIMD for 4 cycles; forward 290, backward 588 instructions per wave-frame = 2 clip-frames).
//
// Here a lane owns a CHAIN of up to four consecutive bones of the DFS-ordered tree (data/carla/files/structure.yaml) and
// eight lanes own a clip, so a wavefront walks the frames of EIGHT clips with all 64 lanes busy:
//
//     lane & 7   chain                     bones (DFS index)       attached to
//        0       trunk                      0  1  2  3              --
//        1       left arm                   4  5  6  7              spine01 (3)
//        2       neck / head / eyes         8  9 10 11 (11 -> 9)    spine01 (3)
//        3       right arm                 12 13 14 15              spine01 (3)
//        4       right leg                 16 17 18 19              hips (1)
//        5       left leg                  21 22 23 24              hips (1)
//        6       right toe end             20                       right toe (19, end of lane 4)
//        7       left toe end              25                       left toe (24, end of lane 5)
//
//   * forward kinematics (walker_control/p3d_pose.py:116-184) is sequential INSIDE a lane -- one composition per bone, in
//     registers, relative to the chain's first parent -- followed by ONE exchange per frame: every lane fetches the
//     transform its chain hangs on (hips or spine01 from the trunk lane; the toe-end lanes also their leg's end) with DPP
//     moves inside the clip's eight lanes (quad broadcast + row_shr: VALU, no LDS round trip) and applies it to its four
//     locations;
//   * normaliser statistics (hips / neck points, the bbox fallback) are worked out once per lane-frame, not per bone; sums
//     over a clip's bones are an in-lane add over four bones plus a 3-step DPP butterfly over the 8 lanes;
//   * the backward's subtree sums (children -> parent accumulation of FK gradients) are suffix sums inside the lane plus
//     the chain totals handed to the trunk lane (two quad sums);
//   * frame inputs arrive by LDS-DMA (buffer_load ... lds): whole 624 / 208 / 312-byte rows of the eight clips, coalesced,
//     into a wave-private LDS image, from which every lane reads its 96 + 32 + 48 bytes; no workgroup barrier anywhere.
//
// Same arithmetic as the joint-lane kernels up to the association of the 3x3 products (FK composes chain-locally, then
// with the base); parity: tests/test_pose_head_gpu.py runs every large-batch case on both mappings.
// Scope (else the joint-lane kernels run): 6-D kinds, no materialised outputs, no world motion, no external gradients,
// targets in the CARLA joint layout with 2 channels, hips / neck = joints 1 / 8 (HipsNeckExtractor(CARLA_SKELETON)).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>

#include "../../include/p2c.h"

#include "p2c_pose_head_dev.h"

#ifndef P2C_CHAIN_DMA_AUX
#define P2C_CHAIN_DMA_AUX 0              // cache policy bits of the LDS-DMA loads (2 = nt: streamed once)
#endif
// minimum wavefronts per SIMD the register allocation is held to (A/B builds: make EXTRA=-DP2C_CHAIN_..._WAVES=n)
#ifndef P2C_CHAIN_BWD_WAVES
#define P2C_CHAIN_BWD_WAVES 1
#endif
#ifndef P2C_CHAIN_FWD_WAVES
#define P2C_CHAIN_FWD_WAVES 1
#endif

#include "p2c_pose_head_chain_dev.h"

namespace p2c {
namespace chain {

// ---- frame inputs: LDS-DMA of whole rows, then per-lane reads --------------------------------------------------------
struct Stage {
  __amdgpu_buffer_rsrc_t y, g2, g3;
  int vy[N_DMA_Y], v2[N_DMA_2], v3[N_DMA_3];     // per-lane source offsets of the DMA pieces (frame 0)
  int ry, r2, r3;                                // per-lane LDS byte addresses of this lane's bones
  int Bexp;
};
typedef __attribute__((address_space(3))) void *lds_ptr;

__device__ __forceinline__ Stage make_stage(const p2c_pose_head_desc &d, const Lane &L, unsigned lds_wave) {
  Stage s;
  const int clip0 = __builtin_amdgcn_readfirstlane(L.clip);
  const int avail = clip0 < d.B ? (d.B - clip0 < CLIPS ? d.B - clip0 : CLIPS) : 0;
#ifdef P2C_CHAIN_EXPERIMENT_FRAME_MAJOR    // timing experiment only (wrong results): inputs addressed as if laid out (T, B, J, .)
  const int cy = Y_ROW, c2 = G2_ROW, c3 = G3_ROW;
  auto rsrc = [&](const float *base, int clip_bytes) {
    const uintptr_t q = reinterpret_cast<uintptr_t>(base) + (size_t)clip0 * clip_bytes;
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(q), 0, base ? 0x7ffffff0 : 0, 0x00020000);
  };
#else
  const int cy = d.T * Y_ROW, c2 = d.T * G2_ROW, c3 = d.T * G3_ROW;
  auto rsrc = [&](const float *base, int clip_bytes) {
    const uintptr_t q = reinterpret_cast<uintptr_t>(base) + (size_t)clip0 * clip_bytes;
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(q), 0, base ? avail * clip_bytes : 0, 0x00020000);
  };
#endif
  s.y = rsrc(d.y, cy), s.g2 = rsrc(d.gt2d, c2), s.g3 = rsrc(d.gt3d, c3);
#pragma unroll
  for (int i = 0; i < N_DMA_Y; ++i) {
    const int p = i * 64 + L.lane;
    s.vy[i] = p < CLIPS * 39 ? (p / 39) * cy + (p % 39) * 16 : OOB;
  }
#pragma unroll
  for (int i = 0; i < N_DMA_2; ++i) {
    const int p = i * 64 + L.lane;
    s.v2[i] = p < CLIPS * 13 ? (p / 13) * c2 + (p % 13) * 16 : OOB;
  }
#pragma unroll
  for (int i = 0; i < N_DMA_3; ++i) {
    const int p = i * 64 + L.lane;
    s.v3[i] = p < CLIPS * 26 ? (p / 26) * c3 + (p % 26) * 12 : OOB;
  }
  s.Bexp = d.B;
  s.ry = lds_wave + LDS_Y + L.slot * Y_ROW + L.start * 24;
  s.r2 = lds_wave + LDS_G2 + L.slot * G2_ROW + L.start * 8;
  s.r3 = lds_wave + LDS_G3 + (L.slot * J + L.start) * 16;
  return s;
}
// rows of frame t of the wave's eight clips -> the wave's LDS image (asynchronous; counted on vmcnt)
__device__ __forceinline__ void stage_issue(const Stage &s, unsigned lds_wave, int lane, int t) {
#ifdef P2C_CHAIN_EXPERIMENT_SAME_ROWS      // timing experiment only (wrong results): every frame re-reads frame 0 -> cache hits
  t = 0;
#endif
#ifdef P2C_CHAIN_EXPERIMENT_FRAME_MAJOR
  const int oy = t * Y_ROW * s.Bexp, o2 = t * G2_ROW * s.Bexp, o3 = t * G3_ROW * s.Bexp;
#else
  const int oy = t * Y_ROW, o2 = t * G2_ROW, o3 = t * G3_ROW;
#endif
#pragma unroll
  for (int i = 0; i < N_DMA_Y; ++i)
    if (i * 64 + 64 <= CLIPS * 39 || i * 64 + lane < CLIPS * 39)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(s.y, (lds_ptr)(uintptr_t)(lds_wave + LDS_Y + i * 1024), 16, s.vy[i], oy, 0, P2C_CHAIN_DMA_AUX);
#if defined(P2C_CHAIN_EXPERIMENT_NO_COMPUTE) && P2C_CHAIN_EXPERIMENT_NO_COMPUTE == 2
  return;                                  // (experiment: the y stream alone)
#endif
#pragma unroll
  for (int i = 0; i < N_DMA_2; ++i)
    if (i * 64 + 64 <= CLIPS * 13 || i * 64 + lane < CLIPS * 13)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(s.g2, (lds_ptr)(uintptr_t)(lds_wave + LDS_G2 + i * 1024), 16, s.v2[i], o2, 0, P2C_CHAIN_DMA_AUX);
#pragma unroll
  for (int i = 0; i < N_DMA_3; ++i)
    if (i * 64 + 64 <= CLIPS * 26 || i * 64 + lane < CLIPS * 26)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(s.g3, (lds_ptr)(uintptr_t)(lds_wave + LDS_G3 + i * 1024), 12, s.v3[i], o3, 0, P2C_CHAIN_DMA_AUX);
}
__device__ __forceinline__ void stage_read(const Stage &s, FrameIn4 &f) {
  typedef __attribute__((address_space(3))) const fb_f32x2 *lds_f2;
  typedef __attribute__((address_space(3))) const fb_f32x4 *lds_f4;
  lds_f2 py = (lds_f2)(uintptr_t)s.ry, p2 = (lds_f2)(uintptr_t)s.r2;
  lds_f4 p3 = (lds_f4)(uintptr_t)s.r3;
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    const fb_f32x2 a = py[k * 3 + 0], b = py[k * 3 + 1], c = py[k * 3 + 2];
    f.y[k][0] = a[0], f.y[k][1] = a[1], f.y[k][2] = b[0], f.y[k][3] = b[1], f.y[k][4] = c[0], f.y[k][5] = c[1];
    const fb_f32x2 g = p2[k];
    f.g2[k][0] = g[0], f.g2[k][1] = g[1];
    const fb_f32x4 h = p3[k];                           // (the fourth dword is padding)
    f.g3[k][0] = h[0], f.g3[k][1] = h[1], f.g3[k][2] = h[2];
  }
}

// The reference skeletons (data/carla/reference.py: 4 types x 26 bones x (3 + 9) floats = 4992 bytes) sit in LDS behind the four
// staging images, copied once per workgroup with coalesced loads: read per lane straight from global memory they were 48
// scattered dword loads per lane and wavefront -- about as many address-coalescer cycles per CU as a tenth of the kernel's
// streamed rows (the memory path alone ran 248 us against 195 us for the same requests without them, tools/exp/readbw.hip).
constexpr int TAB_LOC = 0, TAB_ROT = 4 * J * 3, TAB_FLOATS = 4 * J * 12, TAB_BYTES = TAB_FLOATS * 4;
typedef __attribute__((address_space(3))) float *lds_fp;
__device__ __forceinline__ void stage_tables(const p2c_pose_head_desc &d, unsigned lds_tab) {
  lds_fp t = (lds_fp)(uintptr_t)lds_tab;
  for (int i = threadIdx.x; i < TAB_FLOATS; i += blockDim.x)
    t[i] = i < TAB_ROT ? d.ref_rel_loc[i] : d.ref_rel_rot[i - TAB_ROT];
  __syncthreads();
}
template <int KIND>
__device__ __forceinline__ void load_reference(const p2c_pose_head_desc &d, const Lane &L, unsigned lds_tab, V3 (&l)[NS], M3 (&R)[NS]) {
  using K = KindTraits<KIND>;
  const int st = L.clip_ok ? d.skel_type[L.clip] : 0;
  lds_fp t = (lds_fp)(uintptr_t)lds_tab;
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    const int j = L.start + k < J ? L.start + k : J - 1;            // (steps a lane does not own: any bone, masked below)
    const int row = (st & 3) * J + j;
    l[k] = v3(t[TAB_LOC + row * 3], t[TAB_LOC + row * 3 + 1], t[TAB_LOC + row * 3 + 2]);
    l[k] = sel(L.valid[k], l[k], v3(0.f, 0.f, 0.f));
    R[k] = identity();
    if (K::SCAN) {
#pragma unroll
      for (int i = 0; i < 9; ++i) R[k].m[i] = t[TAB_ROT + row * 9 + i];
    }
  }
}

// final_rel_rot (B, 26, 3, 3) -- the one activation the forward hands to the backward (rel_rot of the last frame: the reverse
// scan starts there) -- crosses HBM in whole 1 KB wave accesses: the eight clips of a wavefront are 7488 contiguous bytes,
// transposed through the wave's staging image (idle before the first and after the last frame). Written lane by lane (144
// bytes per lane) the 61 MB hand-over of B = 65 536 cost as much as a quarter of the streamed rows.
constexpr int ROT_BYTES = CLIPS * J * 36, N_ROT = (ROT_BYTES + 1023) / 1024;
constexpr int LDS_WAVE_BWD = LDS_WAVE + CLIPS * Y_ROW;      // the backward adds an image of the frame's grad_y rows (below)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rot_rsrc(const p2c_pose_head_desc &d, const Lane &L) {
  const int clip0 = __builtin_amdgcn_readfirstlane(L.clip);
  const int avail = clip0 < d.B ? (d.B - clip0 < CLIPS ? d.B - clip0 : CLIPS) : 0;
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(reinterpret_cast<uintptr_t>(d.final_rel_rot) + (size_t)clip0 * J * 36), 0,
                                           d.final_rel_rot ? avail * J * 36 : 0, 0x00020000);
}
__device__ __forceinline__ void store_rotations(const p2c_pose_head_desc &d, const Lane &L, unsigned lds_wave, const M3 (&R)[NS]) {
  lds_fp img = (lds_fp)(uintptr_t)lds_wave;
  const int at = (L.slot * J + L.start) * 9;
#pragma unroll
  for (int k = 0; k < NS; ++k)
    if (k == 0 || !L.toe) {
#pragma unroll
      for (int i = 0; i < 9; ++i) img[at + k * 9 + i] = R[k].m[i];
    }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const __amdgpu_buffer_rsrc_t rsrc = rot_rsrc(d, L);
  typedef __attribute__((address_space(3))) const fb_f32x4 *lds_f4;
#pragma unroll
  for (int i = 0; i < N_ROT; ++i) {
    const int p = i * 64 + L.lane;
    if (i * 64 + 64 <= ROT_BYTES / 16 || p < ROT_BYTES / 16) {
      const fb_f32x4 w = ((lds_f4)(uintptr_t)lds_wave)[p];
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(fb_u32x4, w), rsrc, p * 16, 0, 0);
    }
  }
}
__device__ __forceinline__ void load_rotations(const p2c_pose_head_desc &d, const Lane &L, unsigned lds_wave, M3 (&R)[NS]) {
  const __amdgpu_buffer_rsrc_t rsrc = rot_rsrc(d, L);
#pragma unroll
  for (int i = 0; i < N_ROT; ++i)
    if (i * 64 + 64 <= ROT_BYTES / 16 || i * 64 + L.lane < ROT_BYTES / 16)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr)(uintptr_t)(lds_wave + i * 1024), 16, (i * 64 + L.lane) * 16, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  lds_fp img = (lds_fp)(uintptr_t)lds_wave;
  const int at = (L.slot * J + L.start) * 9;
#pragma unroll
  for (int k = 0; k < NS; ++k)
#pragma unroll
    for (int i = 0; i < 9; ++i) R[k].m[i] = img[at + k * 9 + i];
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  // Steps a lane does not own read the next bones' values -- and, for the left toe end of a wave's last clip, LDS BEHIND the 7 488
  // bytes that were fetched: whatever an earlier kernel left there. Those steps carry no force, but 0 x NaN is NaN: a stale NaN
  // travelled through the chain-local FK into F x x of the lane's OWN bone (intermittent NaN in grad_y of bone 25). Identity there.
#pragma unroll
  for (int k = 0; k < NS; ++k) R[k] = sel(L.valid[k], R[k], identity());
}

// =====================================================================================================================
// forward
// =====================================================================================================================
template <int KIND>
__global__ __launch_bounds__(256, P2C_CHAIN_FWD_WAVES) void pose_head_chain_fwd(const p2c_pose_head_desc d) {
  using K = KindTraits<KIND>;
  extern __shared__ float4 chain_lds[];
  const Lane L = make_lane(d);
  const unsigned lds_wave = (unsigned)(uintptr_t)(lds_ptr)chain_lds + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) * LDS_WAVE;
  const Stage S = make_stage(d, L, lds_wave);
  const int T = d.T;
  V3 l[NS];
  M3 R[NS];
  const unsigned lds_tab = (unsigned)(uintptr_t)(lds_ptr)chain_lds + 4 * LDS_WAVE;
#ifndef P2C_CHAIN_EXPERIMENT_BARE
  stage_tables(d, lds_tab);
#endif
  stagger();
  stage_issue(S, lds_wave, L.lane, 0);               // frame 0's rows are on their way while the per-clip constants load
#ifndef P2C_CHAIN_EXPERIMENT_BARE
  load_reference<KIND>(d, L, lds_tab, l, R);
#else
  for (int k = 0; k < NS; ++k) l[k] = v3(0.f, 0.f, 0.f), R[k] = identity();
#endif
  Acc acc{0.f, 0.f, 0.f};
  for (int t = 0; t < T; ++t) {
    // The rows of this frame were requested one iteration ago. hipcc puts its own wait for an LDS-DMA in front of the first
    // read only -- before the loop, not on the back edge (seen in the ISA: the loop header started with the ds_reads) -- so
    // the wait is explicit.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    FrameIn4 in;
#if defined(P2C_CHAIN_EXPERIMENT_NO_COMPUTE) && P2C_CHAIN_EXPERIMENT_NO_COMPUTE == 3
    acc.sum2 += *(__attribute__((address_space(3))) const float *)(uintptr_t)S.ry;      // (experiment: one LDS read per frame)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (t + 1 < T) stage_issue(S, lds_wave, L.lane, t + 1);
    continue;
#endif
    stage_read(S, in);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // every read has left LDS before the next frame's rows arrive
    if (t + 1 < T) stage_issue(S, lds_wave, L.lane, t + 1);
#ifdef P2C_CHAIN_EXPERIMENT_NO_COMPUTE     // timing experiment only: the memory path alone
#pragma unroll
    for (int k = 0; k < NS; ++k) acc.sum2 += in.y[k][0] + in.y[k][5] + in.g2[k][1] + in.g3[k][2];
    continue;
#endif
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      SixD s;
      const M3 c = rot6d_fwd(in.y[k], s);
      R[k] = K::SCAN ? mul(c, R[k]) : c;             // p3d_pose.py:98-114
    }
    M3 Al[NS], Ap3;
    V3 xl[NS];
    fk_local(L, R, l, Al, xl, Ap3);
    M3 BA;
    V3 BX;
    fk_base(L, Al, xl, BA, BX);
    V3 x[NS], F[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) x[k] = vmul(xl[k], BA) + BX;
    head4<false>(d, L, t, x, in, acc, 0.f, 0.f, F);
  }
#ifndef P2C_CHAIN_EXPERIMENT_BARE
  if (K::SCAN && d.final_rel_rot) store_rotations(d, L, lds_wave, R);       // (every staged row has been read: the image is free)
#endif
  const float s2 = wave_sum(acc.sum2), c2 = wave_sum(acc.cnt2), s3 = wave_sum(acc.sum3);
  if (L.lane == 0) {
    const size_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    float *p = d.partials + wave * 4;
    p[0] = s2, p[1] = c2, p[2] = s3, p[3] = 0.f;
  }
}

// =====================================================================================================================
// backward (tangent-space form, see pose_head_rot_bwd_tangent in p2c_pose_head.hip): frames in reverse, forward recomputed
// =====================================================================================================================
template <int KIND>
__global__ __launch_bounds__(256, P2C_CHAIN_BWD_WAVES) void pose_head_chain_bwd(const p2c_pose_head_desc d, const GradLosses grad_losses, float *grad_y) {
  using K = KindTraits<KIND>;
  extern __shared__ float4 chain_lds[];
  const Lane L = make_lane(d);
  const unsigned lds_wave = (unsigned)(uintptr_t)(lds_ptr)chain_lds + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) * LDS_WAVE_BWD;
  const Stage S = make_stage(d, L, lds_wave);
  const int T = d.T;
  V3 l[NS];
  M3 R[NS];
  const unsigned lds_tab = (unsigned)(uintptr_t)(lds_ptr)chain_lds + 4 * LDS_WAVE_BWD;
  stage_tables(d, lds_tab);
  stagger();
  load_reference<KIND>(d, L, lds_tab, l, R);         // (R: overwritten below; the reference rotations are re-read at frame 0)
#pragma unroll
  for (int k = 0; k < NS; ++k) R[k] = identity();
  if (K::SCAN) load_rotations(d, L, lds_wave, R);    // through the staging image, before the first frame's rows go there
  stage_issue(S, lds_wave, L.lane, T - 1);
  float coef2 = 0.f, coef3 = 0.f;
  loss_coefs(d, grad_losses, coef2, coef3);
  Acc acc{0.f, 0.f, 0.f};
  V3 Ssum[NS];       // suffix sums over time of the parent-frame torques
#pragma unroll
  for (int k = 0; k < NS; ++k) Ssum[k] = v3(0.f, 0.f, 0.f);

  // grad_y rows leave through buffer stores, bone by bone
  const int clip0 = __builtin_amdgcn_readfirstlane(L.clip);
  const int avail = clip0 < d.B ? (d.B - clip0 < CLIPS ? d.B - clip0 : CLIPS) : 0;
  const int cy = T * Y_ROW;
  const __amdgpu_buffer_rsrc_t gy_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<void *>(reinterpret_cast<uintptr_t>(grad_y) + (size_t)clip0 * cy), 0, avail * cy, 0x00020000);
  const unsigned gy_img = lds_wave + LDS_WAVE + L.slot * Y_ROW + L.start * 24;      // this lane's bones in the wave's grad_y image

  // (frame 0 is peeled: there rel_rot[t-1] is the reference pose, read from the table, and nothing is staged behind it)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the rows of the last frame (see the forward)
  auto frame = [&](const int t, auto first) {
    constexpr bool FIRST = decltype(first)::value;
    FrameIn4 in;
    stage_read(S, in);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (!FIRST) stage_issue(S, lds_wave, L.lane, t - 1);
    M3 c[NS];
    SixD s6[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      c[k] = rot6d_fwd(in.y[k], s6[k]);
      if (!K::SCAN) R[k] = c[k];
    }
    M3 Al[NS], Ap3;
    V3 xl[NS];
    fk_local(L, R, l, Al, xl, Ap3);
    M3 BA;
    V3 BX;
    fk_base(L, Al, xl, BA, BX);
    V3 x[NS], F[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) x[k] = sel(L.valid[k], vmul(xl[k], BA) + BX, v3(0.f, 0.f, 0.f));   // (unowned steps: a finite point)
    head4<true>(d, L, t, x, in, acc, coef2, coef3, F);

    // ---- subtree sums of F and F x x: suffix sums inside the chain, chain totals to the trunk, toe ends to their legs ----
    V3 FX[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) FX[k] = cross(F[k], x[k]);       // (F is zero for the steps a lane does not own)
    V3 SubF[NS], SubX[NS];
    subtree4(L, F, SubF);
    subtree4(L, FX, SubX);
    // ---- torque about each bone, into the tangent of its relative rotation, suffix sum over time, 6-D pull-back ---------
    M3 Rref0[NS];
    if (K::SCAN && FIRST) {
      V3 unused[NS];
      load_reference<KIND>(d, L, lds_tab, unused, Rref0);
    }
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      const V3 tau = SubX[k] - cross(SubF[k], x[k]);
      V3 taup = vmulT(tau, BA);                      // tau A_parent^T, A_parent = A'_parent-in-chain A_base
      if (k > 0) taup = vmulT(taup, (k == 3) ? Ap3 : Al[k - 1]);
      V3 g = taup;
      if (K::SCAN) {
        Ssum[k] = Ssum[k] + taup;
        // change is a rotation: rel_rot[t-1] = change^T rel_rot[t]; before frame 0 it is the reference pose, exactly
        const M3 Rprev = FIRST ? Rref0[k] : mulTN(c[k], R[k]);
        g = vmulT(Ssum[k], Rprev);
        if (!FIRST) R[k] = Rprev;
      }
      const SixD &s = s6[k];
      float gy6[6];
      {
        const V3 b3 = v3(c[k].m[6], c[k].m[7], c[k].m[8]);
        const float al = dot(g, s.b1), be = dot(g, s.b2), ga = dot(g, b3);
        const float r1 = frcp(s.n1), r2 = frcp(s.n2);
        const float k3 = (be + al * s.d * r2) * r1, k2 = -ga * r1, k5 = -al * r2;
        gy6[0] = fmaf(k3, b3.x, k2 * s.b2.x), gy6[1] = fmaf(k3, b3.y, k2 * s.b2.y), gy6[2] = fmaf(k3, b3.z, k2 * s.b2.z);
        gy6[3] = k5 * b3.x, gy6[4] = k5 * b3.y, gy6[5] = k5 * b3.z;
      }
      if (__any(!(s.c1 && s.c2))) {   // (rare, wave-uniform) a norm sits on the 1e-12 clamp: generic chain rule through Gram-Schmidt
        M3 G;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const V3 ci = v3(c[k].m[i * 3], c[k].m[i * 3 + 1], c[k].m[i * 3 + 2]);
          const V3 h = cross(ci, g) * 0.5f;
          G.m[i * 3] = h.x, G.m[i * 3 + 1] = h.y, G.m[i * 3 + 2] = h.z;
        }
        float slow[6];
        rot6d_bwd(s, G, slow);
#pragma unroll
        for (int i = 0; i < 6; ++i) gy6[i] = (s.c1 && s.c2) ? gy6[i] : slow[i];
      }
      // the bone's 24 bytes go to the wave's image of this frame's grad_y rows (a toe-end lane owns step 0 only)
      if (k == 0 || !L.toe) {
        typedef __attribute__((address_space(3))) fb_f32x2 *lds_w2;
        lds_w2 q = (lds_w2)(uintptr_t)(gy_img + k * 24);
        q[0] = fb_f32x2{gy6[0], gy6[1]}, q[1] = fb_f32x2{gy6[2], gy6[3]}, q[2] = fb_f32x2{gy6[4], gy6[5]};
      }
    }
    // ... and leave as whole rows: five 1 KB stores per wavefront and frame, at the offsets the y rows were fetched from.
    // (Bone by bone -- a 16- and an 8-byte store each -- the same bytes were 64 partial lines per store instruction.)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < N_DMA_Y; ++i)
      if (i * 64 + 64 <= CLIPS * 39 || i * 64 + L.lane < CLIPS * 39) {
        typedef __attribute__((address_space(3))) const fb_f32x4 *lds_f4;
        const fb_f32x4 w = ((lds_f4)(uintptr_t)(lds_wave + LDS_WAVE))[i * 64 + L.lane];
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(fb_u32x4, w), gy_rsrc, S.vy[i], t * Y_ROW, 0);
      }
    // The next frame's rows must have landed before its reads; this frame's five stores need not have. vmcnt retires in
    // issue order (loads, stores and LDS-DMA alike on gfx9) and the eleven DMA requests were issued before the stores.
    asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
  };
  for (int t = T - 1; t > 0; --t) frame(t, std::false_type{});
  frame(0, std::true_type{});
}

}  // namespace chain
}  // namespace p2c

// =====================================================================================================================
// dispatch hooks for p2c_pose_head.hip
// =====================================================================================================================
using namespace p2c;

static int g_chain_min_b = -1;
static int chain_min_b() {
  if (g_chain_min_b < 0) {
    const char *e = getenv("P2C_CHAIN_MIN_B");
    g_chain_min_b = e ? atoi(e) : 8192;      // measured (tools/kbench.py): 8 clips per wavefront fill the chip from ~8k clips on
  }
  return g_chain_min_b;
}
extern "C" P2C_API int p2c_pose_head_set_chain_min_batch(int32_t min_b) {
  const int prev = chain_min_b();
  if (min_b >= 0) g_chain_min_b = min_b;
  return prev;
}

// true when the chain-lane kernels implement this descriptor (the caller has already ruled out materialised outputs and, for
// the backward, external gradients; small batches go to the time-parallel kernels before this is asked)
bool p2c_internal_chain_supported(const p2c_pose_head_desc &d) {
  if (d.kind != P2C_KIND_POSE_CHANGES_6D && d.kind != P2C_KIND_RELATIVE_ROT_6D) return false;
  if (d.dloc || d.drot || d.B < chain_min_b()) return false;
  if (d.transform != P2C_TRANSFORM_NONE && (d.n_hips != 1 || d.n_neck != 1 || d.hips_idx[0] != chain::HIPS || d.neck_idx[0] != chain::NECK))
    return false;
  if (d.gt2d && (d.gt2d_joints != P2C_JOINTS || d.gt2d_channels != 2)) return false;
  if (d.gt3d && d.gt3d_joints != P2C_JOINTS) return false;
  for (int j = 0; j < P2C_JOINTS; ++j)
    if ((d.gt2d && d.gmap2d[j] != j) || (d.gt3d && d.gmap3d[j] != j)) return false;
  if ((long long)d.T * chain::Y_ROW * chain::CLIPS >= 0x7fff0000ll) return false;       // 32-bit buffer offsets
  return true;
}
unsigned p2c_internal_chain_waves(int B) { return (unsigned)((B + chain::CLIPS - 1) / chain::CLIPS); }

static inline dim3 chain_grid(int B) { return dim3((p2c_internal_chain_waves(B) + 3) / 4); }
static constexpr size_t kChainLds = 4 * chain::LDS_WAVE + chain::TAB_BYTES, kChainLdsBwd = 4 * chain::LDS_WAVE_BWD + chain::TAB_BYTES;

int p2c_internal_chain_fwd(const p2c_pose_head_desc &d, hipStream_t stream) {
  if (d.kind == P2C_KIND_POSE_CHANGES_6D)
    hipLaunchKernelGGL(chain::pose_head_chain_fwd<P2C_KIND_POSE_CHANGES_6D>, chain_grid(d.B), dim3(256), kChainLds, stream, d);
  else
    hipLaunchKernelGGL(chain::pose_head_chain_fwd<P2C_KIND_RELATIVE_ROT_6D>, chain_grid(d.B), dim3(256), kChainLds, stream, d);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}
int p2c_internal_chain_bwd(const p2c_pose_head_desc &d, const GradLosses &gl, float *grad_y, hipStream_t stream) {
  if (d.kind == P2C_KIND_POSE_CHANGES_6D)
    hipLaunchKernelGGL(chain::pose_head_chain_bwd<P2C_KIND_POSE_CHANGES_6D>, chain_grid(d.B), dim3(256), kChainLdsBwd, stream, d, gl, grad_y);
  else
    hipLaunchKernelGGL(chain::pose_head_chain_bwd<P2C_KIND_RELATIVE_ROT_6D>, chain_grid(d.B), dim3(256), kChainLdsBwd, stream, d, gl, grad_y);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}
