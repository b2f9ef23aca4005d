// p2c_train_dev.h -- geometry shared by the two forms of the fused train step's first launch: train_clip_kernel (p2c_train.hip:
// one workgroup of eight wavefronts per clip, the latency form for about one clip per CU) and train_stream_kernel
// (p2c_train_stream.hip: a pair of wavefronts per clip, four pairs per workgroup: the throughput form for several clips per CU). Both leave the same factor
// blocks and per-clip loss sums for train_wgrad_kernel.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/p2c.h"
#include "p2c_mlp_dev.h"
#include "p2c_pose_head_dev.h"

namespace p2c_train {

using namespace p2c_mlp;
namespace ph = p2c;

using S = LinearAE156;                       // 52 -> 26 -> 13 -> 6 -> 39 -> 78 -> 156 (linear_ae.py:25-40, 6-D output)
constexpr int NLAY = S::NLAY;
// factor block of one clip: rows [H_0 .. H_{L-1} | G_1 .. G_L], 16 samples (64 B) per row
__host__ __device__ constexpr int f_h_off(int l) {
  int r = 0;
  for (int i = 0; i < l; ++i) r += S::dim_at(i);
  return r;
}
constexpr int F_HALF = f_h_off(NLAY);
__host__ __device__ constexpr int f_g_off(int l) {
  int r = F_HALF;
  for (int i = 1; i < l; ++i) r += S::dim_at(i);
  return r;
}
constexpr int F_ROWS = f_g_off(NLAY) + S::dim_at(NLAY);

struct ClipArgs {
  const float *x;          // (B*T, 52) model input
  const float *w_image;    // packed weight image (p2c_mlp_pack layout), current
  const float *counts;     // (B) unmasked 2-D target pairs per clip
  float *factors;          // (B, F_ROWS, 16)
  int32_t *counters;       // arrival tickets of the second launch: zeroed here
  int32_t n_counters;
  int32_t identity_maps;   // gmap2d[j] == gmap3d[j] == j for every joint (host-checked)
};

}  // namespace p2c_train

// p2c_train_stream.hip: the throughput form of the first launch. *_supported: the descriptor is one the kernel implements
// (identity joint maps, CARLA targets with two channels, hips / neck = joints 1 / 8, no world motion).
bool p2c_internal_train_stream_supported(const p2c_pose_head_desc &d);
int p2c_internal_train_stream_launch(const p2c_pose_head_desc &d, const p2c::GradLosses &gl, const p2c_train::ClipArgs &m, hipStream_t stream);
