// p2c_lstm.hip -- K7b: the recurrent half of an LSTM layer as ONE launch per direction of time (gfx950, fp32 MFMA).
//
// Seq2Seq(Embeddings) of the reference (modules/movements/seq2seq/seq2seq.py:21-94,245-349) runs nn.LSTM(hidden 64, 2
// layers) over the clip in the encoder and -- frame by frame, T times -- in the decoder. Through the framework RNN path
// that is ~1 700 launches of 3-5 us per train step at B = 512 (per-time-step tensor ops). The layer splits into
//   (a) the input projection  gx[t] = x[t] W_ih^T + b_ih + b_hh  for all t at once: a plain dense GEMM, left to the
//       library (rocBLAS fp32-MFMA kernels), as are its weight / bias / input gradients;
//   (b) the recurrence  gates[t] = gx[t] + h[t-1] W_hh^T,  c[t] = f c[t-1] + i g,  h[t] = o tanh(c[t])
//       (gate order i, f, g, o as torch.nn.LSTM): T dependent steps of a (16 x H) x (H x 4H) product plus pointwise
//       math -- this file.
// Mapping: a workgroup owns 16 sequences; wave w owns hidden units [16w, 16w+16) and holds the four gate tiles of W_hh
// for them as MFMA A fragments IN REGISTERS for the whole sequence (4 x H/4 VGPRs); h[t-1] lives transposed in LDS
// (the MFMA B operand), the accumulators start from gx[t], and because lane (c, g) of every gate tile holds the same
// (sample c, units 16w+4g..+3) the cell update happens in registers with no exchange. One barrier per time step.
// Backward: same mapping with W_hh^T fragments; it produces d gates (which is also d gx: the library GEMMs turn it into
// the weight gradients  dW_hh = sum_t dgates[t]^T h[t-1],  dW_ih, db  outside) and carries dh, dc in registers.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/p2c.h"

namespace p2c_lstm {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int TS = 16;    // sequences per workgroup
constexpr int TP = 17;    // LDS pitch of a transposed row [unit][sequence]

struct Args {
  const float *gx;      // (T, B, 4H) input projection incl. both biases      [bwd: unused]
  const float *h0, *c0; // (B, H) or NULL (= zeros)
  const float *w_hh;    // (4H, H)
  float *out;           // (T, B, H)
  float *hT, *cT;       // (B, H) or NULL
  float *acts;          // (T, B, 4H) activated gates i, f, g, o   (saved for the backward)
  float *cs;            // (T, B, H)  cell states                   (saved for the backward)
  const float *g_out;   // (T, B, H) or NULL      } backward inputs
  const float *g_hT, *g_cT;   // (B, H) or NULL   }
  float *g_gx;          // (T, B, 4H)             } backward outputs
  float *g_h0, *g_c0;   // (B, H) or NULL         }
  int32_t T, B, H;
};

// v_exp_f32 + v_rcp_f32 (1 ulp each): the IEEE division sequence would triple the cost of the cell update
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return 2.f * __builtin_amdgcn_rcpf(1.f + __expf(-2.f * x)) - 1.f; }

__device__ __forceinline__ f32x4 load4(const float *p, bool ok) {
  return ok ? *reinterpret_cast<const f32x4 *>(p) : (f32x4){0.f, 0.f, 0.f, 0.f};
}

// ---- forward -----------------------------------------------------------------------------------------------------------
// W_hh (4H x H floats) -> LDS with coalesced 16-byte loads (a strided per-lane gather from global costs ~10 us per
// launch, which is most of a single-step decoder call); pitch H + 1 keeps the fragment pick-up conflict-free
template <int H>
__device__ __forceinline__ void stage_w(const float *w_hh, float *wl) {
  constexpr int N4 = 4 * H * H / 4;
  const f32x4 *src = reinterpret_cast<const f32x4 *>(w_hh);
  for (int i = threadIdx.x; i < N4; i += blockDim.x) {
    const f32x4 v = src[i];
    const int e = i * 4, row = e / H, col = e - row * H;
    float *p = wl + row * (H + 1) + col;
    p[0] = v[0], p[1] = v[1], p[2] = v[2], p[3] = v[3];
  }
  __syncthreads();
}

template <int H>
__global__ __launch_bounds__(64 * (H / 16)) void lstm_rec_fwd_kernel(const Args a) {
  constexpr int KS = H / 4;                       // k-steps of the recurrent product
  extern __shared__ float dyn_lds[];              // [4H][H+1] staging image of W_hh, then unused
  __shared__ float hbuf[2][H * TP];               // h[t-1]^T, double buffered
  const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.x * TS + c;              // this lane's sequence
  const bool ok = b < a.B;
  const int u0 = w * 16 + 4 * g;                  // first of this lane's four hidden units
  const int B = a.B, T = a.T;

  stage_w<H>(a.w_hh, dyn_lds);
  float frag[4][KS];                              // A fragments: gate q, rows 16w + (lane & 15), k = 4 ks + g
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) frag[q][ks] = dyn_lds[(q * H + w * 16 + c) * (H + 1) + 4 * ks + g];

  f32x4 cst = a.c0 ? load4(a.c0 + (size_t)b * H + u0, ok) : (f32x4){0.f, 0.f, 0.f, 0.f};
  {
    f32x4 h = a.h0 ? load4(a.h0 + (size_t)b * H + u0, ok) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) hbuf[0][(u0 + r) * TP + c] = h[r];
  }
  __syncthreads();
  int cur = 0;
  f32x4 hlast = {0.f, 0.f, 0.f, 0.f};
  f32x4 nxt[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) nxt[q] = load4(a.gx + ((size_t)0 * B + b) * 4 * H + q * H + u0, ok);
  for (int t = 0; t < T; ++t) {
    f32x4 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = nxt[q];
    if (t + 1 < T) {                              // the next step's input projection is in flight during this step
#pragma unroll
      for (int q = 0; q < 4; ++q) nxt[q] = load4(a.gx + ((size_t)(t + 1) * B + b) * 4 * H + q * H + u0, ok);
    }
    const float *hb = hbuf[cur] + g * TP + c;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const float bv = hb[ks * 4 * TP];
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(frag[q][ks], bv, acc[q], 0, 0, 0);
    }
    f32x4 ai, af, ag, ao, h;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      ai[r] = sigmoidf_(acc[0][r]), af[r] = sigmoidf_(acc[1][r]), ag[r] = tanhf_(acc[2][r]), ao[r] = sigmoidf_(acc[3][r]);
      cst[r] = af[r] * cst[r] + ai[r] * ag[r];
      h[r] = ao[r] * tanhf_(cst[r]);
    }
    hlast = h;
    if (ok) {
      const size_t row = (size_t)t * B + b;
      if (a.acts) {
        float *p = a.acts + row * 4 * H + u0;
        *reinterpret_cast<f32x4 *>(p) = ai, *reinterpret_cast<f32x4 *>(p + H) = af;
        *reinterpret_cast<f32x4 *>(p + 2 * H) = ag, *reinterpret_cast<f32x4 *>(p + 3 * H) = ao;
      }
      if (a.cs) *reinterpret_cast<f32x4 *>(a.cs + row * H + u0) = cst;
      *reinterpret_cast<f32x4 *>(a.out + row * H + u0) = h;
    }
    float *hn = hbuf[cur ^ 1];
#pragma unroll
    for (int r = 0; r < 4; ++r) hn[(u0 + r) * TP + c] = h[r];
    __syncthreads();
    cur ^= 1;
  }
  if (ok) {
    if (a.hT) *reinterpret_cast<f32x4 *>(a.hT + (size_t)b * H + u0) = (T > 0) ? hlast : (a.h0 ? load4(a.h0 + (size_t)b * H + u0, true) : hlast);
    if (a.cT) *reinterpret_cast<f32x4 *>(a.cT + (size_t)b * H + u0) = cst;
  }
}

// ---- backward ----------------------------------------------------------------------------------------------------------
template <int H>
__global__ __launch_bounds__(64 * (H / 16)) void lstm_rec_bwd_kernel(const Args a) {
  constexpr int KS = H;                           // 4H gate rows / 4
  extern __shared__ float dyn_lds[];              // [4H][H+1] staging image of W_hh, then d gates^T, double buffered
  float *dg0 = dyn_lds, *dg1 = dyn_lds + 4 * H * TP;   // (2 x 4H x TP <= 4H x (H+1) for H >= 33; sized on the host)
  const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.x * TS + c;
  const bool ok = b < a.B;
  const int u0 = w * 16 + 4 * g;
  const int B = a.B, T = a.T;

  stage_w<H>(a.w_hh, dyn_lds);
  float frag[KS];                                 // A fragments of W_hh^T: rows = units 16w + (lane & 15), k = gate row 4 ks + g
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) frag[ks] = dyn_lds[(4 * ks + g) * (H + 1) + w * 16 + c];
  __syncthreads();                                // the staging image is dead: its space becomes the d-gates buffers

  f32x4 dh = a.g_hT ? load4(a.g_hT + (size_t)b * H + u0, ok) : (f32x4){0.f, 0.f, 0.f, 0.f};
  f32x4 dc = a.g_cT ? load4(a.g_cT + (size_t)b * H + u0, ok) : (f32x4){0.f, 0.f, 0.f, 0.f};
  int cur = 0;
  for (int t = T - 1; t >= 0; --t) {
    const size_t row = (size_t)t * B + b;
    const float *pa = a.acts + row * 4 * H + u0;
    const f32x4 ai = load4(pa, ok), af = load4(pa + H, ok), ag = load4(pa + 2 * H, ok), ao = load4(pa + 3 * H, ok);
    const f32x4 ct = load4(a.cs + row * H + u0, ok);
    const f32x4 cp = (t > 0) ? load4(a.cs + (row - B) * H + u0, ok)
                             : (a.c0 ? load4(a.c0 + (size_t)b * H + u0, ok) : (f32x4){0.f, 0.f, 0.f, 0.f});
    const f32x4 go = a.g_out ? load4(a.g_out + row * H + u0, ok) : (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 pi, pf, pg, po;                          // gradients of the pre-activation gates
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float dht = go[r] + dh[r];
      const float tc = tanhf_(ct[r]);
      const float dct = dc[r] + dht * ao[r] * (1.f - tc * tc);
      po[r] = dht * tc * ao[r] * (1.f - ao[r]);
      pi[r] = dct * ag[r] * ai[r] * (1.f - ai[r]);
      pf[r] = dct * cp[r] * af[r] * (1.f - af[r]);
      pg[r] = dct * ai[r] * (1.f - ag[r] * ag[r]);
      dc[r] = dct * af[r];
    }
    if (ok) {
      float *p = a.g_gx + row * 4 * H + u0;
      *reinterpret_cast<f32x4 *>(p) = pi, *reinterpret_cast<f32x4 *>(p + H) = pf;
      *reinterpret_cast<f32x4 *>(p + 2 * H) = pg, *reinterpret_cast<f32x4 *>(p + 3 * H) = po;
    }
    float *d = cur ? dg1 : dg0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      d[(u0 + r) * TP + c] = pi[r], d[(H + u0 + r) * TP + c] = pf[r];
      d[(2 * H + u0 + r) * TP + c] = pg[r], d[(3 * H + u0 + r) * TP + c] = po[r];
    }
    __syncthreads();
    // dh[t-1] = W_hh^T d gates: two accumulators (even / odd k-steps) halve the dependent MFMA chain
    f32x4 e0 = {0.f, 0.f, 0.f, 0.f}, e1 = {0.f, 0.f, 0.f, 0.f};
    const float *db = d + g * TP + c;
#pragma unroll
    for (int ks = 0; ks < KS; ks += 2) {
      e0 = __builtin_amdgcn_mfma_f32_16x16x4f32(frag[ks], db[ks * 4 * TP], e0, 0, 0, 0);
      e1 = __builtin_amdgcn_mfma_f32_16x16x4f32(frag[ks + 1], db[(ks + 1) * 4 * TP], e1, 0, 0, 0);
    }
    dh = e0 + e1;
    cur ^= 1;                                      // the other buffer was last read two steps ago: one barrier per step
  }
  if (ok) {
    if (a.g_h0) *reinterpret_cast<f32x4 *>(a.g_h0 + (size_t)b * H + u0) = dh;
    if (a.g_c0) *reinterpret_cast<f32x4 *>(a.g_c0 + (size_t)b * H + u0) = dc;
  }
}

}  // namespace p2c_lstm

using namespace p2c_lstm;

static int check(const p2c_lstm_desc *d, Args &a) {
  if (!d || !d->w_hh) return P2C_E_NULL;
  if (d->T < 0 || d->B < 0) return P2C_E_SHAPE;
  if (d->H != 16 && d->H != 32 && d->H != 48 && d->H != 64) return P2C_E_SHAPE;
  a = Args{};
  a.gx = d->gx, a.h0 = d->h0, a.c0 = d->c0, a.w_hh = d->w_hh, a.out = d->out, a.hT = d->hT, a.cT = d->cT;
  a.acts = d->acts, a.cs = d->cs, a.g_out = d->g_out, a.g_hT = d->g_hT, a.g_cT = d->g_cT, a.g_gx = d->g_gx;
  a.g_h0 = d->g_h0, a.g_c0 = d->g_c0, a.T = d->T, a.B = d->B, a.H = d->H;
  return 0;
}

static size_t lds_bytes(int H, bool bwd) {
  size_t image = (size_t)4 * H * (H + 1), dgates = (size_t)2 * 4 * H * TP;
  return sizeof(float) * ((bwd && dgates > image) ? dgates : image);
}
template <int H>
static void allow_lds() {
  static bool done = false;
  if (done) return;
  (void)hipFuncSetAttribute((const void *)lstm_rec_fwd_kernel<H>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  (void)hipFuncSetAttribute((const void *)lstm_rec_bwd_kernel<H>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  done = true;
}
#define P2C_LSTM_DISPATCH(KERNEL, BWD)                                                                                     \
  switch (a.H) {                                                                                                           \
    case 16: allow_lds<16>(); hipLaunchKernelGGL(KERNEL<16>, grid, dim3(64), lds_bytes(16, BWD), (hipStream_t)stream, a); break;   \
    case 32: allow_lds<32>(); hipLaunchKernelGGL(KERNEL<32>, grid, dim3(128), lds_bytes(32, BWD), (hipStream_t)stream, a); break;  \
    case 48: allow_lds<48>(); hipLaunchKernelGGL(KERNEL<48>, grid, dim3(192), lds_bytes(48, BWD), (hipStream_t)stream, a); break;  \
    default: allow_lds<64>(); hipLaunchKernelGGL(KERNEL<64>, grid, dim3(256), lds_bytes(64, BWD), (hipStream_t)stream, a);         \
  }

extern "C" int p2c_lstm_rec_fwd(const p2c_lstm_desc *d, void *stream) {
  Args a;
  int rc = check(d, a);
  if (rc) return rc;
  if (!a.gx || !a.out) return P2C_E_NULL;
  if (a.B == 0) return 0;
  const dim3 grid((unsigned)((a.B + TS - 1) / TS));
  P2C_LSTM_DISPATCH(lstm_rec_fwd_kernel, false)
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

extern "C" int p2c_lstm_rec_bwd(const p2c_lstm_desc *d, void *stream) {
  Args a;
  int rc = check(d, a);
  if (rc) return rc;
  if (!a.acts || !a.cs || !a.g_gx) return P2C_E_NULL;
  if (a.B == 0) return 0;
  const dim3 grid((unsigned)((a.B + TS - 1) / TS));
  P2C_LSTM_DISPATCH(lstm_rec_bwd_kernel, true)
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}
