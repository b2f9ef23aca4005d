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
#include "p2c_rec_dev.h"

namespace p2c_lstm {

using namespace p2c_rec;
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
  const float *bias_a, *bias_b;   // (4H) or NULL: added to gx as it is read (b_ih, b_hh: the projection GEMM then runs bias-free)
  float *g_gx_bt;       // (B, T, 4H) or NULL: second copy of g_gx, batch-first (pairs with a batch-first layer input)
  int32_t T, B, H, gx_bt;         // gx_bt: gx is laid out (B, T, 4H)
  // inter-layer dropout on this layer's output, drawn in the kernels (p2c_rec_dev.h): fwd also writes out_drop = out * mask (the
  // next layer's input; `out` stays the raw h the weight gradients need), bwd takes g_out as the gradient of out_drop
  float *out_drop;      // (T, B, H) or NULL
  DropRng rng;
};

__device__ __forceinline__ f32x4 load4(const float *p, bool ok) {
  return ok ? *reinterpret_cast<const f32x4 *>(p) : (f32x4){0.f, 0.f, 0.f, 0.f};
}

// ---- forward -----------------------------------------------------------------------------------------------------------
// W_hh (4H x H floats) -> LDS with coalesced 16-byte loads (a strided per-lane gather from global costs ~10 us per
// launch, which is most of a single-step decoder call); pitch H + 1 keeps the fragment pick-up conflict-free
// Hidden sizes above 64 (the reference's own `hidden_size: 128` configs): 4H x (H + 1) floats no longer fit LDS (264 KB at 128),
// so the image is staged in CHUNKS of whole gates -- two chunks of 2H rows -- and a lane picks up the fragments whose rows the
// chunk in LDS holds (a compile-time test in the 16-sequence kernels, a per-lane one in the 4-sequence kernels).
template <int H>
struct Chunks {
  static constexpr int N = H > 64 ? 2 : 1, ROWS = 4 * H / N;
};
template <int H>
__device__ __forceinline__ void stage_w(const float *w_hh, float *wl, const int chunk = 0) {
  constexpr int N4 = Chunks<H>::ROWS * H / 4;
  const f32x4 *src = reinterpret_cast<const f32x4 *>(w_hh) + (size_t)chunk * N4;
  constexpr int SB = 8;                           // loads in flight per thread: the 64 KB image is two round trips, not 16
  const int nt = blockDim.x;
  if (chunk > 0) __syncthreads();                 // every fragment of the previous chunk has been picked up
  for (int i0 = threadIdx.x; i0 < N4; i0 += nt * SB) {
    f32x4 v[SB];
#pragma unroll
    for (int r = 0; r < SB; ++r) v[r] = (i0 + r * nt < N4) ? src[i0 + r * nt] : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < SB; ++r) {
      const int i = i0 + r * nt;
      if (i >= N4) continue;
      const int e = i * 4, row = e / H, col = e - row * H;
      float *p = wl + row * (H + 1) + col;
      p[0] = v[r][0], p[1] = v[r][1], p[2] = v[r][2], p[3] = v[r][3];
    }
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
  DropRng rng = a.rng;
  const bool hashed = rng.state != nullptr && a.out_drop != nullptr;
  if (hashed) drop_begin(rng, false);
  const int off4 = (b * 4 * H + u0) * 4, off1 = (b * H + u0) * 4;   // byte offsets inside one step's rows

  float frag[4][KS];                              // A fragments: gate q, rows 16w + (lane & 15), k = 4 ks + g
#pragma unroll
  for (int ch = 0; ch < Chunks<H>::N; ++ch) {
    stage_w<H>(a.w_hh, dyn_lds, ch);
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (q * H / Chunks<H>::ROWS == ch) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) frag[q][ks] = dyn_lds[(q * H - ch * Chunks<H>::ROWS + w * 16 + c) * (H + 1) + 4 * ks + g];
      }
  }

  f32x4 cst = a.c0 ? load4(a.c0 + (size_t)b * H + u0, ok) : (f32x4){0.f, 0.f, 0.f, 0.f};
  {
    f32x4 h = a.h0 ? load4(a.h0 + (size_t)b * H + u0, ok) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) hbuf[0][(u0 + r) * TP + c] = h[r];
  }
  f32x4 nxt[4], bsum[4];
  const int offx = a.gx_bt ? (b * T * 4 * H + u0) * 4 : off4;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
#pragma unroll
    for (int r = 0; r < 4; ++r)                    // (scalar reads: a bias inside a flat parameter buffer is only 4-byte aligned)
      bsum[q][r] = (a.bias_a ? a.bias_a[q * H + u0 + r] : 0.f) + (a.bias_b ? a.bias_b[q * H + u0 + r] : 0.f);
  }
  {
    const __amdgpu_buffer_rsrc_t rg = a.gx_bt ? bt_rows(a.gx, 0, B, T, 4 * H) : step_rows(a.gx, 0, B, 4 * H);
#pragma unroll
    for (int q = 0; q < 4; ++q) nxt[q] = bload4(rg, offx + q * H * 4);
  }
  __syncthreads();
  if (hashed) drop_keys(rng, false);
#pragma unroll
  for (int q = 0; q < 4; ++q) pin(nxt[q]);        // resident on entry: the loop header then carries no pending loads
  int cur = 0;
  f32x4 hlast = {0.f, 0.f, 0.f, 0.f};
  for (int t = 0; t < T; ++t) {
    f32x4 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = nxt[q] + bsum[q];
    {   // the next step's input projection is in flight during this step (the last step re-reads its own rows: no branch)
      const int tn = (t + 1 < T) ? t + 1 : t;
      const __amdgpu_buffer_rsrc_t rg = a.gx_bt ? bt_rows(a.gx, tn, B, T, 4 * H) : step_rows(a.gx, tn, B, 4 * H);
#pragma unroll
      for (int q = 0; q < 4; ++q) nxt[q] = bload4(rg, offx + q * H * 4);
    }
    const float *hb = hbuf[cur] + g * TP + c;
    float bv[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) bv[ks] = hb[ks * 4 * TP];
    // gates i and g first: their activations run on the VALU while the matrix pipe works through f and o
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(frag[0][ks], bv[ks], acc[0], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(frag[2][ks], bv[ks], acc[2], 0, 0, 0);
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(frag[1][ks], bv[ks], acc[1], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(frag[3][ks], bv[ks], acc[3], 0, 0, 0);
    }
    f32x4 ai, af, ag, ao, h, ig;
#pragma unroll
    for (int r = 0; r < 4; ++r) ai[r] = sigmoidf_(acc[0][r]), ag[r] = tanhf_(acc[2][r]), ig[r] = ai[r] * ag[r];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      af[r] = sigmoidf_(acc[1][r]), ao[r] = sigmoidf_(acc[3][r]);
      cst[r] = af[r] * cst[r] + ig[r];
      h[r] = ao[r] * tanhf_(cst[r]);
    }
    hlast = h;
    float *hn = hbuf[cur ^ 1];
#pragma unroll
    for (int r = 0; r < 4; ++r) hn[(u0 + r) * TP + c] = h[r];
#pragma unroll
    for (int q = 0; q < 4; ++q) pin(nxt[q]);
    {
      const __amdgpu_buffer_rsrc_t ra = step_rows(a.acts, t, B, 4 * H), rc = step_rows(a.cs, t, B, H), ro = step_rows(a.out, t, B, H);
      bstore4(ra, off4, ai), bstore4(ra, off4 + H * 4, af), bstore4(ra, off4 + 2 * H * 4, ag), bstore4(ra, off4 + 3 * H * 4, ao);
      bstore4(rc, off1, cst);
      bstore4(ro, off1, h);
      {   // (the store in either case -- a NULL tensor has no records -- and ALU only under the branch: a store under a branch makes
          // every counted wait behind the merge a full drain)
        f32x4 m = {1.f, 1.f, 1.f, 1.f};
        if (hashed) m = drop_value4(rng, (uint32_t)((t * B + b) * H + u0));
        bstore4(step_rows(a.out_drop, t, B, H), off1, h * m);
      }
    }
    lds_barrier();
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
  DropRng rng = a.rng;
  const bool hashed = rng.state != nullptr;
  if (hashed) drop_begin(rng, true);
  const int off4 = (b * 4 * H + u0) * 4, off1 = (b * H + u0) * 4;

  float frag[KS];                                 // A fragments of W_hh^T: rows = units 16w + (lane & 15), k = gate row 4 ks + g
#pragma unroll
  for (int ch = 0; ch < Chunks<H>::N; ++ch) {
    stage_w<H>(a.w_hh, dyn_lds, ch);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
      if (4 * ks / Chunks<H>::ROWS == ch) frag[ks] = dyn_lds[(4 * ks + g - ch * Chunks<H>::ROWS) * (H + 1) + w * 16 + c];
  }
  __syncthreads();                                // the staging image is dead: its space becomes the d-gates buffers
  if (hashed) drop_keys(rng, true);

  f32x4 dh = a.g_hT ? load4(a.g_hT + (size_t)b * H + u0, ok) : (f32x4){0.f, 0.f, 0.f, 0.f};
  f32x4 dc = a.g_cT ? load4(a.g_cT + (size_t)b * H + u0, ok) : (f32x4){0.f, 0.f, 0.f, 0.f};
  // The saved rows of step t - 1 are requested at the top of step t and pinned at its end: their latency hides behind the
  // whole step (cell math + 64 MFMAs).
  struct Saved { f32x4 ai, af, ag, ao, cp, go, m; };
  auto fetch = [&](int t, Saved &s) {             // t >= 0
    const __amdgpu_buffer_rsrc_t ra = step_rows(a.acts, t, B, 4 * H), rg = step_rows(a.g_out, t, B, H);
    const __amdgpu_buffer_rsrc_t rc = (t > 0) ? step_rows(a.cs, t - 1, B, H) : step_rows(a.c0, 0, B, H);
    s.ai = bload4(ra, off4), s.af = bload4(ra, off4 + H * 4), s.ag = bload4(ra, off4 + 2 * H * 4), s.ao = bload4(ra, off4 + 3 * H * 4);
    s.cp = bload4(rc, off1);
    s.go = bload4(rg, off1);
    s.m = (f32x4){1.f, 1.f, 1.f, 1.f};             // (applied where go is used: a product here would wait for the load)
    if (hashed) s.m = drop_value4(rng, (uint32_t)((t * B + b) * H + u0));
  };
  Saved nx = {};
  f32x4 ct = {0.f, 0.f, 0.f, 0.f};
  if (T > 0) {
    fetch(T - 1, nx);
    ct = bload4(step_rows(a.cs, T - 1, B, H), off1);
  }
  pin(nx.ai), pin(nx.af), pin(nx.ag), pin(nx.ao), pin(nx.cp), pin(nx.go), pin(ct);
  int cur = 0;
  for (int t = T - 1; t >= 0; --t) {
    const f32x4 ai = nx.ai, af = nx.af, ag = nx.ag, ao = nx.ao, cp = nx.cp, go = nx.go * nx.m;
    fetch(t > 0 ? t - 1 : 0, nx);                  // (the last step re-reads its own rows: no branch in the body)
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
    ct = cp;                                       // c[t-1] is the next step's cell state
    float *d = cur ? dg1 : dg0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      d[(u0 + r) * TP + c] = pi[r], d[(H + u0 + r) * TP + c] = pf[r];
      d[(2 * H + u0 + r) * TP + c] = pg[r], d[(3 * H + u0 + r) * TP + c] = po[r];
    }
    {
      const __amdgpu_buffer_rsrc_t rx = step_rows(a.g_gx, t, B, 4 * H);
      bstore4(rx, off4, pi), bstore4(rx, off4 + H * 4, pf), bstore4(rx, off4 + 2 * H * 4, pg), bstore4(rx, off4 + 3 * H * 4, po);
      const __amdgpu_buffer_rsrc_t rb = bt_rows(a.g_gx_bt, t, B, T, 4 * H);
      const int ob = (b * T * 4 * H + u0) * 4;
      bstore4(rb, ob, pi), bstore4(rb, ob + H * 4, pf), bstore4(rb, ob + 2 * H * 4, pg), bstore4(rb, ob + 3 * H * 4, po);
    }
    lds_barrier();
    // dh[t-1] = W_hh^T d gates: two accumulators (even / odd k-steps) halve the dependent MFMA chain
    f32x4 e0 = {0.f, 0.f, 0.f, 0.f}, e1 = {0.f, 0.f, 0.f, 0.f};
    const float *db = d + g * TP + c;
#pragma unroll
    for (int ks = 0; ks < KS; ks += 2) {
      e0 = __builtin_amdgcn_mfma_f32_16x16x4f32(frag[ks], db[ks * 4 * TP], e0, 0, 0, 0);
      e1 = __builtin_amdgcn_mfma_f32_16x16x4f32(frag[ks + 1], db[(ks + 1) * 4 * TP], e1, 0, 0, 0);
    }
    dh = e0 + e1;
    pin(nx.ai), pin(nx.af), pin(nx.ag), pin(nx.ao), pin(nx.cp), pin(nx.go);
    cur ^= 1;                                      // the other buffer was last read two steps ago: one barrier per step
  }
  if (ok) {
    if (a.g_h0) *reinterpret_cast<f32x4 *>(a.g_h0 + (size_t)b * H + u0) = dh;
    if (a.g_c0) *reinterpret_cast<f32x4 *>(a.g_c0 + (size_t)b * H + u0) = dc;
  }
}

// ---- narrow variants: 4 sequences per workgroup ------------------------------------------------------------------------
// At the batch sizes the flows train with (B = 512: 32 workgroups of 16 sequences) the 16 x 16 tile leaves 7/8 of the CUs
// idle while each busy SIMD queues 64 MFMAs of 32 cycles per time step. v_mfma_f32_4x4x1_16B_f32 runs 16 independent
// 4 x 4 x 1 outer products per instruction at the same FLOP rate (8 cycles): with the 16 blocks = 16 hidden units, the
// 4 rows of a block = the unit's four gates and the 4 columns = 4 SEQUENCES (the B operand, shared by all blocks), a
// workgroup needs only 4 sequences -- 4x the workgroups, a quarter of the MFMA cycles and of the cell math per step each.
// Lane (block k, column s) receives the four gate pre-activations of (unit 16w + k, sequence s): the cell update is one
// element per lane, again without any exchange. Used for B <= 4096 (above that the wide kernels fill the chip and stage
// W_hh four times less often).
constexpr int NS = 4;

__device__ __forceinline__ f32x4 mfma4(float a_, float b_, f32x4 c_) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a_, b_, c_, 0, 0, 0); }

template <int H>
__global__ __launch_bounds__(64 * (H / 16)) void lstm_rec_fwd_narrow_kernel(const Args a) {
  constexpr int HP = H + 4;                       // LDS pitch of one sequence's h row (16-byte aligned)
  extern __shared__ float dyn_lds[];              // [4H][H+1] staging image of W_hh, then unused
  __shared__ __attribute__((aligned(16))) float hs[2][NS][HP];   // h[t-1], double buffered
  const int lane = threadIdx.x & 63, s = lane & 3;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int u = w * 16 + (lane >> 2);             // this lane's hidden unit (its MFMA block)
  const int b = blockIdx.x * NS + s;              // this lane's sequence (its MFMA column)
  const bool ok = b < a.B;
  const int B = a.B, T = a.T;
  DropRng rng = a.rng;
  const bool hashed = rng.state != nullptr && a.out_drop != nullptr;
  if (hashed) drop_begin(rng, false);
  const int offg = (b * 4 * H + u) * 4, offh = (b * H + u) * 4;   // byte offsets inside one step's rows

  float frag[H];                                  // A operand: row (lane & 3) = gate, of this lane's block = unit; k = 0..H-1
#pragma unroll
  for (int ch = 0; ch < Chunks<H>::N; ++ch) {
    stage_w<H>(a.w_hh, dyn_lds, ch);
    const int row = s * H + u - ch * Chunks<H>::ROWS;          // this lane's gate row inside the chunk, if it is there
    if (row >= 0 && row < Chunks<H>::ROWS) {
#pragma unroll
      for (int k = 0; k < H; ++k) frag[k] = dyn_lds[row * (H + 1) + k];
    }
  }

  float cst = (a.c0 && ok) ? a.c0[(size_t)b * H + u] : 0.f;
  hs[0][s][u] = (a.h0 && ok) ? a.h0[(size_t)b * H + u] : 0.f;
  f32x4 nxt, bsum;
  const int offx = a.gx_bt ? (b * T * 4 * H + u) * 4 : offg;
#pragma unroll
  for (int q = 0; q < 4; ++q) bsum[q] = (a.bias_a ? a.bias_a[q * H + u] : 0.f) + (a.bias_b ? a.bias_b[q * H + u] : 0.f);
  {
    const __amdgpu_buffer_rsrc_t rg = a.gx_bt ? bt_rows(a.gx, 0, B, T, 4 * H) : step_rows(a.gx, 0, B, 4 * H);
#pragma unroll
    for (int q = 0; q < 4; ++q) nxt[q] = bload1(rg, offx + q * H * 4);
  }
  __syncthreads();
  if (hashed) drop_keys(rng, false);
  pin(nxt);
  int cur = 0;
  float hlast = 0.f;
  for (int t = 0; t < T; ++t) {
    f32x4 acc0 = nxt + bsum, acc1 = {0.f, 0.f, 0.f, 0.f}, acc2 = acc1, acc3 = acc1;
    {
      const int tn = (t + 1 < T) ? t + 1 : t;
      const __amdgpu_buffer_rsrc_t rg = a.gx_bt ? bt_rows(a.gx, tn, B, T, 4 * H) : step_rows(a.gx, tn, B, 4 * H);
#pragma unroll
      for (int q = 0; q < 4; ++q) nxt[q] = bload1(rg, offx + q * H * 4);
    }
    const f32x4 *hp = reinterpret_cast<const f32x4 *>(hs[cur][s]);
#pragma unroll
    for (int k4 = 0; k4 < H / 4; ++k4) {
      const f32x4 hv = hp[k4];
      acc0 = mfma4(frag[4 * k4], hv[0], acc0), acc1 = mfma4(frag[4 * k4 + 1], hv[1], acc1);
      acc2 = mfma4(frag[4 * k4 + 2], hv[2], acc2), acc3 = mfma4(frag[4 * k4 + 3], hv[3], acc3);
    }
    const f32x4 acc = (acc0 + acc1) + (acc2 + acc3);
    const float ai = sigmoidf_(acc[0]), af = sigmoidf_(acc[1]), ag = tanhf_(acc[2]), ao = sigmoidf_(acc[3]);
    cst = af * cst + ai * ag;
    const float h = ao * tanhf_(cst);
    hlast = h;
    hs[cur ^ 1][s][u] = h;
    pin(nxt);
    {
      const __amdgpu_buffer_rsrc_t ra = step_rows(a.acts, t, B, 4 * H), rc = step_rows(a.cs, t, B, H), ro = step_rows(a.out, t, B, H);
      bstore1(ra, offg, ai), bstore1(ra, offg + H * 4, af), bstore1(ra, offg + 2 * H * 4, ag), bstore1(ra, offg + 3 * H * 4, ao);
      bstore1(rc, offh, cst);
      bstore1(ro, offh, h);
      {
        float m = 1.f;
        if (hashed) m = drop_value(rng, (uint32_t)((t * B + b) * H + u));
        bstore1(step_rows(a.out_drop, t, B, H), offh, h * m);
      }
    }
    lds_barrier();
    cur ^= 1;
  }
  if (ok) {
    if (a.hT) a.hT[(size_t)b * H + u] = (T > 0) ? hlast : (a.h0 ? a.h0[(size_t)b * H + u] : 0.f);
    if (a.cT) a.cT[(size_t)b * H + u] = cst;
  }
}

// Backward, narrow: dh[t-1] = W_hh^T d gates has the hidden units as output rows and the 4H gate rows as K. A wave owns
// units [16w, 16w+16) as 4 groups of 4; its 16 MFMA blocks are (unit group ug) x (K quarter q = gate q's H rows): H
// instructions per step, then the four quarters are summed across lanes (two row-rotate DPP adds). For the cell math lane
// (ug, q, s) takes the element (unit 16w + 4ug + q, sequence s), i.e. register q of the reduced tile.
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  const int r = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false);
  return v + __builtin_bit_cast(float, r);
}

template <int H>
__global__ __launch_bounds__(64 * (H / 16)) void lstm_rec_bwd_narrow_kernel(const Args a) {
  constexpr int HQ = H + 4, GP = 4 * HQ;          // LDS pitches: gate j of a sequence at j * HQ (the 16 (sequence, quarter) rows a
                                                  // wave reads then start in different banks), sequences GP apart
  extern __shared__ float dyn_lds[];              // [4H][H+1] staging image of W_hh
  __shared__ __attribute__((aligned(16))) float dgs[2][NS][GP];
  const int lane = threadIdx.x & 63, s = lane & 3, blk = lane >> 2, ug = blk >> 2, q = blk & 3;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int u = w * 16 + 4 * ug + q;              // this lane's element for the cell math
  const int b = blockIdx.x * NS + s;
  const bool ok = b < a.B;
  const int B = a.B, T = a.T;
  DropRng rng = a.rng;
  const bool hashed = rng.state != nullptr;
  if (hashed) drop_begin(rng, true);
  const int offg = (b * 4 * H + u) * 4, offh = (b * H + u) * 4;

  float frag[H];                                  // A operand: row (lane & 3) -> unit 16w + 4ug + (lane & 3); k -> gate row q H + k
#pragma unroll
  for (int ch = 0; ch < Chunks<H>::N; ++ch) {
    stage_w<H>(a.w_hh, dyn_lds, ch);
    const int row0 = q * H - ch * Chunks<H>::ROWS;             // the lane's quarter of the gate rows inside the chunk, if it is there
    if (row0 >= 0 && row0 < Chunks<H>::ROWS) {
#pragma unroll
      for (int k = 0; k < H; ++k) frag[k] = dyn_lds[(row0 + k) * (H + 1) + w * 16 + 4 * ug + s];
    }
  }

  float dh = (a.g_hT && ok) ? a.g_hT[(size_t)b * H + u] : 0.f;
  float dc = (a.g_cT && ok) ? a.g_cT[(size_t)b * H + u] : 0.f;
  struct Saved { f32x4 act; float cp, go, m; };
  auto fetch = [&](int t, Saved &sv) {
    const __amdgpu_buffer_rsrc_t ra = step_rows(a.acts, t, B, 4 * H), rg = step_rows(a.g_out, t, B, H);
    const __amdgpu_buffer_rsrc_t rc = (t > 0) ? step_rows(a.cs, t - 1, B, H) : step_rows(a.c0, 0, B, H);
#pragma unroll
    for (int j = 0; j < 4; ++j) sv.act[j] = bload1(ra, offg + j * H * 4);
    sv.cp = bload1(rc, offh);
    sv.go = bload1(rg, offh);
    sv.m = 1.f;
    if (hashed) sv.m = drop_value(rng, (uint32_t)((t * B + b) * H + u));
  };
  if (hashed) drop_keys(rng, true);                     // (behind the W_hh staging above)
  Saved nx = {};
  float ct = 0.f;
  if (T > 0) {
    fetch(T - 1, nx);
    ct = bload1(step_rows(a.cs, T - 1, B, H), offh);
  }
  __syncthreads();
  pin(nx.act);
  asm volatile("" : "+v"(nx.cp), "+v"(nx.go), "+v"(ct));
  int cur = 0;
  for (int t = T - 1; t >= 0; --t) {
    const float ai = nx.act[0], af = nx.act[1], ag = nx.act[2], ao = nx.act[3], cp = nx.cp, go = nx.go * nx.m;
    fetch(t > 0 ? t - 1 : 0, nx);
    const float dht = go + dh;
    const float tc = tanhf_(ct);
    const float dct = dc + dht * ao * (1.f - tc * tc);
    const float po = dht * tc * ao * (1.f - ao);
    const float pi = dct * ag * ai * (1.f - ai);
    const float pf = dct * cp * af * (1.f - af);
    const float pg = dct * ai * (1.f - ag * ag);
    dc = dct * af;
    ct = cp;
    float *d = dgs[cur][s];
    d[u] = pi, d[HQ + u] = pf, d[2 * HQ + u] = pg, d[3 * HQ + u] = po;
    {
      const __amdgpu_buffer_rsrc_t rx = step_rows(a.g_gx, t, B, 4 * H);
      bstore1(rx, offg, pi), bstore1(rx, offg + H * 4, pf), bstore1(rx, offg + 2 * H * 4, pg), bstore1(rx, offg + 3 * H * 4, po);
      const __amdgpu_buffer_rsrc_t rb = bt_rows(a.g_gx_bt, t, B, T, 4 * H);
      const int ob = (b * T * 4 * H + u) * 4;
      bstore1(rb, ob, pi), bstore1(rb, ob + H * 4, pf), bstore1(rb, ob + 2 * H * 4, pg), bstore1(rb, ob + 3 * H * 4, po);
    }
    lds_barrier();
    f32x4 e0 = {0.f, 0.f, 0.f, 0.f}, e1 = e0, e2 = e0, e3 = e0;
    const f32x4 *dp = reinterpret_cast<const f32x4 *>(d + q * HQ);
#pragma unroll
    for (int k4 = 0; k4 < H / 4; ++k4) {
      const f32x4 dv = dp[k4];
      e0 = mfma4(frag[4 * k4], dv[0], e0), e1 = mfma4(frag[4 * k4 + 1], dv[1], e1);
      e2 = mfma4(frag[4 * k4 + 2], dv[2], e2), e3 = mfma4(frag[4 * k4 + 3], dv[3], e3);
    }
    f32x4 e = (e0 + e1) + (e2 + e3);
#pragma unroll
    for (int i = 0; i < 4; ++i) e[i] = dpp_add<0x128>(dpp_add<0x124>(e[i]));   // + row_ror:4, then + row_ror:8: all four quarters
    dh = (q == 0) ? e[0] : (q == 1) ? e[1] : (q == 2) ? e[2] : e[3];
    pin(nx.act);
    asm volatile("" : "+v"(nx.cp), "+v"(nx.go));
    cur ^= 1;                                      // the other buffer was last read two steps ago: one barrier per step
  }
  if (ok) {
    if (a.g_h0) a.g_h0[(size_t)b * H + u] = dh;
    if (a.g_c0) a.g_c0[(size_t)b * H + u] = dc;
  }
}

}  // namespace p2c_lstm

using namespace p2c_lstm;

static int check(const p2c_lstm_desc *d, Args &a) {
  if (!d || !d->w_hh) return P2C_E_NULL;
  if (d->T < 0 || d->B < 0 || d->B > (1 << 20)) return P2C_E_SHAPE;   // buffer offsets: (B + 16) * 4H * 4 bytes < 2^31
  if (d->H != 16 && d->H != 32 && d->H != 48 && d->H != 64 && d->H != 96 && d->H != 128) return P2C_E_SHAPE;
  a = Args{};
  a.gx = d->gx, a.h0 = d->h0, a.c0 = d->c0, a.w_hh = d->w_hh, a.out = d->out, a.hT = d->hT, a.cT = d->cT;
  a.acts = d->acts, a.cs = d->cs, a.g_out = d->g_out, a.g_hT = d->g_hT, a.g_cT = d->g_cT, a.g_gx = d->g_gx;
  a.g_h0 = d->g_h0, a.g_c0 = d->g_c0, a.T = d->T, a.B = d->B, a.H = d->H;
  a.bias_a = d->bias_a, a.bias_b = d->bias_b, a.g_gx_bt = d->g_gx_bt, a.gx_bt = d->gx_bt;
  if ((a.gx_bt || a.g_gx_bt) && (int64_t)a.B * a.T * 4 * a.H * 4 >= (int64_t)1 << 31) return P2C_E_SHAPE;
  if (d->drop_state) {
    if (!(d->drop_p >= 0.f && d->drop_p < 1.f)) return P2C_E_SHAPE;
    a.out_drop = d->out_drop;
    a.rng.state = d->drop_state, a.rng.site = d->drop_site, a.rng.scale = 1.f / (1.f - d->drop_p);
    a.rng.thresh = (uint32_t)((double)d->drop_p * 4294967296.0);
  }
  return 0;
}

static size_t lds_bytes(int H, bool bwd) {
  size_t image = (size_t)(H > 64 ? 2 : 4) * H * (H + 1), dgates = (size_t)2 * 4 * H * TP;     // (one chunk of the staging image)
  return sizeof(float) * ((bwd && dgates > image) ? dgates : image);
}
template <int H>
static void allow_lds() {
  static bool done = false;
  if (done) return;
  // exactly what each kernel asks for: static + dynamic LDS must stay within 160 KB or the attribute call fails (and leaves its
  // error behind for the next hipGetLastError)
  const int f = (int)lds_bytes(H, false), bw = (int)lds_bytes(H, true);
  (void)hipFuncSetAttribute((const void *)lstm_rec_fwd_kernel<H>, hipFuncAttributeMaxDynamicSharedMemorySize, f);
  (void)hipFuncSetAttribute((const void *)lstm_rec_bwd_kernel<H>, hipFuncAttributeMaxDynamicSharedMemorySize, bw);
  (void)hipFuncSetAttribute((const void *)lstm_rec_fwd_narrow_kernel<H>, hipFuncAttributeMaxDynamicSharedMemorySize, f);
  (void)hipFuncSetAttribute((const void *)lstm_rec_bwd_narrow_kernel<H>, hipFuncAttributeMaxDynamicSharedMemorySize, f);
  (void)hipGetLastError();
  done = true;
}
#define P2C_LSTM_DISPATCH(KERNEL, BWD)                                                                                     \
  switch (a.H) {                                                                                                           \
    case 16: allow_lds<16>(); hipLaunchKernelGGL(KERNEL<16>, grid, dim3(64), lds_bytes(16, BWD), (hipStream_t)stream, a); break;   \
    case 32: allow_lds<32>(); hipLaunchKernelGGL(KERNEL<32>, grid, dim3(128), lds_bytes(32, BWD), (hipStream_t)stream, a); break;  \
    case 48: allow_lds<48>(); hipLaunchKernelGGL(KERNEL<48>, grid, dim3(192), lds_bytes(48, BWD), (hipStream_t)stream, a); break;  \
    case 96: allow_lds<96>(); hipLaunchKernelGGL(KERNEL<96>, grid, dim3(384), lds_bytes(96, BWD), (hipStream_t)stream, a); break;  \
    case 128: allow_lds<128>(); hipLaunchKernelGGL(KERNEL<128>, grid, dim3(512), lds_bytes(128, BWD), (hipStream_t)stream, a); break;  \
    default: allow_lds<64>(); hipLaunchKernelGGL(KERNEL<64>, grid, dim3(256), lds_bytes(64, BWD), (hipStream_t)stream, a);         \
  }

// 4 sequences per workgroup up to B = 4096, 16 above; P2C_REC_TILE=wide|narrow forces one (tests run both)
#include <stdlib.h>
#include <string.h>
static bool use_narrow(int B) {
  const char *e = getenv("P2C_REC_TILE");
  if (e && !strcmp(e, "wide")) return false;
  if (e && !strcmp(e, "narrow")) return true;
  return B <= 4096;
}

extern "C" int p2c_lstm_rec_fwd(const p2c_lstm_desc *d, void *stream) {
  Args a;
  int rc = check(d, a);
  if (rc) return rc;
  if (!a.gx || !a.out) return P2C_E_NULL;
  if (a.B == 0) return 0;
  if (use_narrow(a.B)) {
    const dim3 grid((unsigned)((a.B + NS - 1) / NS));
    P2C_LSTM_DISPATCH(lstm_rec_fwd_narrow_kernel, false)
  } else {
    const dim3 grid((unsigned)((a.B + TS - 1) / TS));
    P2C_LSTM_DISPATCH(lstm_rec_fwd_kernel, false)
  }
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

extern "C" int p2c_lstm_rec_bwd(const p2c_lstm_desc *d, void *stream) {
  Args a;
  int rc = check(d, a);
  if (rc) return rc;
  if (!a.acts || !a.cs || !a.g_gx) return P2C_E_NULL;
  if (a.B == 0) return 0;
  if (use_narrow(a.B)) {
    const dim3 grid((unsigned)((a.B + NS - 1) / NS));
    P2C_LSTM_DISPATCH(lstm_rec_bwd_narrow_kernel, false)
  } else {
    const dim3 grid((unsigned)((a.B + TS - 1) / TS));
    P2C_LSTM_DISPATCH(lstm_rec_bwd_kernel, true)
  }
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}
