// p2c_s2s.hip -- K7c: the whole decoder loop of Seq2Seq(Embeddings) as ONE launch forward and ONE backward (gfx950).
//
// Reference: modules/movements/seq2seq/seq2seq.py:245-349 -- for t in range(T): out_t = fc(LSTM_2layers(step_in, (hidden,
// cell))), step_in = out_t, where (hidden, cell) are the ENCODER's for every frame (the decoder state is not carried,
// seq2seq.py:272-288). With the state frozen, the recurrent terms are per-clip constants
//     k_l = b_ih_l + b_hh_l + W_hh_l hidden_l      (computed by the caller: two small library GEMMs, autograd on top)
// and the decoder is the same 3-stage map applied T times to its own output:
//     gates0 = x_t W_ih0^T + k0 ;  c = f c_enc0 + i g ;  h0 = o tanh(c)   [* dropout mask in training]
//     gates1 = h0  W_ih1^T + k1 ;  c = f c_enc1 + i g ;  h1 = o tanh(c)
//     x_{t+1} = out_t = h1 W_fc^T + b_fc
// As framework ops that is T x (2 LSTM launches + projections + adds) forward and about twice that backward, with every
// weight gradient accumulated T times. Here: a workgroup owns 16 clips, 4 waves; wave w owns hidden units / output
// features [16w, 16w+16); W_ih0, W_ih1, W_fc live as MFMA A fragments IN REGISTERS for all T steps (O/4*4 + 64 + 16
// VGPRs), k_l / c_enc_l / b_fc too; x_t, h0, h1 pass through LDS transposed (the B operands); three barriers per step.
// Backward walks t = T-1..0 with the transposed fragments, carries d x_{t+1} in registers, and writes d gates0, d gates1
// and d out_total once: the weight gradients are then six dense library GEMMs / reductions over all (t, b) at once
// (dW_ih0 = dgates0^T x_prev, dW_ih1 = dgates1^T h0, dW_fc = dout^T h1, dk_l = sum_t dgates_l, db_fc = sum dout).
// H = 64, two layers, O <= 64 (pose_2d: 52); everything else takes the per-step path (ops.lstm_layer).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/p2c.h"
#include "p2c_rec_dev.h"

namespace p2c_s2s {

using namespace p2c_rec;
constexpr int TS = 16, TP = 17, H = 64, G4 = 4 * H, OMAX = 64;

struct Args {
  const float *k0, *c0, *k1, *c1;        // (B,4H), (B,H), (B,4H), (B,H)
  const float *w_ih0, *w_ih1, *w_fc, *b_fc;   // (4H,O), (4H,H), (O,H), (O)
  const float *x0;                       // (B,O) or NULL = zeros (<sos>)
  const float *drop;                     // (T,B,H) multiplicative dropout mask on the layer-0 output, or NULL
  DropRng rng;                           // ... or drawn here (rng.state != NULL; `drop` wins when both are given)
  float *out;                            // (T,B,O)
  float *acts0, *acts1, *h0d, *h1;       // saved for the backward: (T,B,4H) x2, (T,B,H) x2
  const float *g_out;                    // (T,B,O)
  float *g_gates0, *g_gates1, *g_outtot; // (T,B,4H) x2, (T,B,O)
  float *g_c0, *g_c1;                    // (B,H) x2
  // frame-invariant terms formed by the kernels themselves (hid0 != NULL): k_l = ba_l + bb_l + hid_l W_hh_l^T
  const float *hid0, *hid1, *w_hh0, *w_hh1, *b0a, *b0b, *b1a, *b1b;
  float *kw0, *kw1;                      // (B,4H) scratch for k_l (16-clip tiling only)
  float *out_bt;                         // (B,T,O) second copy of out in the model's output layout, or NULL
  float *g_k0, *g_k1, *g_hid0, *g_hid1;  // (B,4H) = sum_t d gates_l, (B,H) = g_k_l W_hh_l; or NULL
  int32_t T, B, O, g_out_bt;             // g_out_bt: g_out is laid out (B,T,O)
  // teacher forcing (seq2seq.py:283-288): where force[t][b] != 0 the frame's output -- and so the next input -- IS target[t][b]
  const float *force, *target;           // (T,B) 0 / 1, (T,B,O); or both NULL
};


__device__ __forceinline__ f32x4 load4(const float *p, bool ok) {
  return ok ? *reinterpret_cast<const f32x4 *>(p) : (f32x4){0.f, 0.f, 0.f, 0.f};
}
__device__ __forceinline__ f32x4 zero4() { return (f32x4){0.f, 0.f, 0.f, 0.f}; }

// rows x cols row-major matrix -> LDS image [rows][cols + 1] (coalesced 16-byte loads when cols % 4 == 0)
__device__ __forceinline__ void stage(const float *src, int rows, int cols, float *img) {
  __syncthreads();                                   // previous users of the image region are done
  const int n = rows * cols;
  if ((cols & 3) == 0) {
    const f32x4 *s4 = reinterpret_cast<const f32x4 *>(src);
    constexpr int SB = 8;                            // loads in flight per thread: a 64 KB image is two round trips, not 16
    const int n4 = n >> 2, nt = blockDim.x;
    for (int i0 = threadIdx.x; i0 < n4; i0 += nt * SB) {
      f32x4 v[SB];
#pragma unroll
      for (int r = 0; r < SB; ++r) v[r] = (i0 + r * nt < n4) ? s4[i0 + r * nt] : zero4();
#pragma unroll
      for (int r = 0; r < SB; ++r) {
        const int i = i0 + r * nt;
        if (i >= n4) continue;
        const int e = i * 4, rr = e / cols, cc = e - rr * cols;
        float *p = img + rr * (cols + 1) + cc;
        p[0] = v[r][0], p[1] = v[r][1], p[2] = v[r][2], p[3] = v[r][3];
      }
    }
  } else {
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
      const int r = i / cols, cc = i - r * cols;
      img[r * (cols + 1) + cc] = src[i];
    }
  }
  __syncthreads();
}

__device__ __forceinline__ void cell_fwd(const f32x4 (&acc)[4], const f32x4 &c_enc, f32x4 &ai, f32x4 &af, f32x4 &ag, f32x4 &ao,
                                         f32x4 &h) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    ai[r] = sigmoidf_(acc[0][r]), af[r] = sigmoidf_(acc[1][r]), ag[r] = tanhf_(acc[2][r]), ao[r] = sigmoidf_(acc[3][r]);
    h[r] = ao[r] * tanhf_(af[r] * c_enc[r] + ai[r] * ag[r]);
  }
}
// gradient of the pre-activation gates given dh (the cell state is the frozen encoder state: nothing is carried in time)
__device__ __forceinline__ void cell_bwd(const f32x4 &dh, const f32x4 &ai, const f32x4 &af, const f32x4 &ag, const f32x4 &ao,
                                         const f32x4 &c_enc, f32x4 &pi, f32x4 &pf, f32x4 &pg, f32x4 &po, f32x4 &dc_acc) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float tc = tanhf_(af[r] * c_enc[r] + ai[r] * ag[r]);
    const float dct = dh[r] * ao[r] * (1.f - tc * tc);
    po[r] = dh[r] * tc * ao[r] * (1.f - ao[r]);
    pi[r] = dct * ag[r] * ai[r] * (1.f - ai[r]);
    pf[r] = dct * c_enc[r] * af[r] * (1.f - af[r]);
    pg[r] = dct * ai[r] * (1.f - ag[r] * ag[r]);
    dc_acc[r] += dct * af[r];
  }
}
__device__ __forceinline__ void store_gates(__amdgpu_buffer_rsrc_t rs, int off4, const f32x4 &a, const f32x4 &b, const f32x4 &c,
                                            const f32x4 &d) {
  bstore4(rs, off4, a), bstore4(rs, off4 + H * 4, b), bstore4(rs, off4 + 2 * H * 4, c), bstore4(rs, off4 + 3 * H * 4, d);
}

// ---- forward -----------------------------------------------------------------------------------------------------------
// KS0 = k-steps of the layer-0 product = ceil(O / 4) rounded to the instantiated values (13: pose_2d's 52 features; 16):
// a compile-time bound keeps the k-loop free of the uniform guards that cut the MFMA stream into 4-instruction blocks.
template <int KS0>
__global__ __launch_bounds__(256) void decoder_fwd_kernel(const Args a) {
  DropRng rng = a.rng;                              // (its state words are requested here, behind nothing; the keys are formed behind the staging)
  const bool hashed = !a.drop && rng.state != nullptr;
  if (hashed) drop_begin(rng, false);
  extern __shared__ float img[];                    // staging image of one weight matrix at a time
  __shared__ float xT[OMAX * TP], h0T[H * TP], h1T[H * TP];
  const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.x * TS + c;
  const bool ok = b < a.B;
  const int u0 = w * 16 + 4 * g;
  const int O = a.O, B = a.B, T = a.T;
  const int off4 = (b * G4 + u0) * 4, off1 = (b * H + u0) * 4;

  float fa0[4][KS0], fa1[4][H / 4], ffc[H / 4];
  stage(a.w_ih0, G4, O, img);
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int ks = 0; ks < KS0; ++ks) {
      const int k = 4 * ks + g;
      fa0[q][ks] = (k < O) ? img[(q * H + w * 16 + c) * (O + 1) + k] : 0.f;
    }
  stage(a.w_ih1, G4, H, img);
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int ks = 0; ks < H / 4; ++ks) fa1[q][ks] = img[(q * H + w * 16 + c) * (H + 1) + 4 * ks + g];
  stage(a.w_fc, O, H, img);
#pragma unroll
  for (int ks = 0; ks < H / 4; ++ks) ffc[ks] = (w * 16 + c < O) ? img[(w * 16 + c) * (H + 1) + 4 * ks + g] : 0.f;

  f32x4 k0r[4], k1r[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    k0r[q] = load4(a.k0 + (size_t)b * G4 + q * H + u0, ok);
    k1r[q] = load4(a.k1 + (size_t)b * G4 + q * H + u0, ok);
  }
  const f32x4 c0r = load4(a.c0 + (size_t)b * H + u0, ok), c1r = load4(a.c1 + (size_t)b * H + u0, ok);
  f32x4 bfc;
  int offo[4], offb[4];                              // byte offsets of this lane's four output features (OOB beyond O)
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    bfc[r] = (u0 + r < O) ? a.b_fc[u0 + r] : 0.f;
    offo[r] = (u0 + r < O) ? (b * O + u0 + r) * 4 : OOB;
    offb[r] = (u0 + r < O) ? (b * T * O + u0 + r) * 4 : OOB;
  }
  {   // x_0 (rows >= O stay zero: they only ever meet zero weight fragments)
#pragma unroll
    for (int r = 0; r < 4; ++r) xT[(u0 + r) * TP + c] = (a.x0 && ok && u0 + r < O) ? a.x0[(size_t)b * O + u0 + r] : 0.f;
  }
  if (hashed) drop_keys(rng, false);
  // this step's dropout mask: a row of the caller's mask tensor, or drawn from the element index (t, b, unit)
  // This step's dropout mask: a row of the caller's mask tensor (requested in either case -- a NULL tensor has no records and costs
  // nothing -- so that both forms have the same loads outstanding) AND, drawn from the element index, the hashed one; the choice is
  // a select where the mask is used. (Written into the LOADED value under a branch, the hash's first instruction re-used the load's
  // destination register and the compiler put s_waitcnt vmcnt(0) in front of it: every load of the step's prefetch drained at the
  // top of the step, + 0.4 us per step.)
  auto mask_of = [&](const int t) -> f32x4 { return bload4(step_rows(a.drop, t, B, H), off1); };
  auto hash_of = [&](const int t) -> f32x4 {
    f32x4 m = {1.f, 1.f, 1.f, 1.f};
    if (hashed) m = drop_value4(rng, (uint32_t)((t * B + b) * H + u0));
    return m;
  };
  f32x4 maskh = hash_of(0);
  f32x4 mask = mask_of(0);
  __syncthreads();
  pin(mask);
  const bool has_drop = a.drop != nullptr || hashed;

  for (int t = 0; t < T; ++t) {
    f32x4 acc[4], ai, af, ag, ao, h;
    const f32x4 m = hashed ? maskh : mask;           // this step's dropout mask was requested / drawn one step ahead
    mask = mask_of((t + 1 < T) ? t + 1 : t), maskh = hash_of((t + 1 < T) ? t + 1 : t);
    // teacher forcing: flag and target features of this step, requested now, used behind the fc product (NULL: zeros)
    const float forced = bload1(step_rows(a.force, t, B, 1), b * 4);
    f32x4 tgt;
    {
      const __amdgpu_buffer_rsrc_t rtg = step_rows(a.target, t, B, O);
#pragma unroll
      for (int r = 0; r < 4; ++r) tgt[r] = bload1(rtg, offo[r]);
    }
    // ---- layer 0
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = k0r[q];
#pragma unroll
    for (int ks = 0; ks < KS0; ++ks) {
      const float bv = xT[(4 * ks + g) * TP + c];
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa0[q][ks], bv, acc[q], 0, 0, 0);
    }
    cell_fwd(acc, c0r, ai, af, ag, ao, h);
    if (has_drop) h *= m;
#pragma unroll
    for (int r = 0; r < 4; ++r) h0T[(u0 + r) * TP + c] = h[r];
    pin(mask);                                       // before the stores: the wait covers the one load only
    store_gates(step_rows(a.acts0, t, B, G4), off4, ai, af, ag, ao);
    bstore4(step_rows(a.h0d, t, B, H), off1, h);
    lds_barrier();
    // ---- layer 1
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = k1r[q];
#pragma unroll
    for (int ks = 0; ks < H / 4; ++ks) {
      const float bv = h0T[(4 * ks + g) * TP + c];
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa1[q][ks], bv, acc[q], 0, 0, 0);
    }
    cell_fwd(acc, c1r, ai, af, ag, ao, h);
#pragma unroll
    for (int r = 0; r < 4; ++r) h1T[(u0 + r) * TP + c] = h[r];
    store_gates(step_rows(a.acts1, t, B, G4), off4, ai, af, ag, ao);
    bstore4(step_rows(a.h1, t, B, H), off1, h);
    lds_barrier();
    // ---- fc: this wave's 16 output features (two accumulators halve the dependent chain)
    f32x4 o = bfc, o2 = zero4();
#pragma unroll
    for (int ks = 0; ks < H / 4; ks += 2) {
      o = __builtin_amdgcn_mfma_f32_16x16x4f32(ffc[ks], h1T[(4 * ks + g) * TP + c], o, 0, 0, 0);
      o2 = __builtin_amdgcn_mfma_f32_16x16x4f32(ffc[ks + 1], h1T[(4 * ks + 4 + g) * TP + c], o2, 0, 0, 0);
    }
    o += o2;
    if (forced != 0.f) o = tgt;
    const __amdgpu_buffer_rsrc_t ro = step_rows(a.out, t, B, O), rb = bt_rows(a.out_bt, t, B, T, O);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      xT[(u0 + r) * TP + c] = o[r];                // next step's input (features >= O are exactly zero)
      bstore1(ro, offo[r], o[r]);
      bstore1(rb, offb[r], o[r]);
    }
    lds_barrier();
  }
}

// ---- backward ----------------------------------------------------------------------------------------------------------
template <int KS0>
__global__ __launch_bounds__(256) void decoder_bwd_kernel(const Args a) {
  DropRng rng = a.rng;                              // (its state words are requested here, behind nothing; the keys are formed behind the staging)
  const bool hashed = !a.drop && rng.state != nullptr;
  if (hashed) drop_begin(rng, true);
  extern __shared__ float img[];
  __shared__ float doT[OMAX * TP], dg1T[G4 * TP], dg0T[G4 * TP];
  const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.x * TS + c;
  const bool ok = b < a.B;
  const int u0 = w * 16 + 4 * g;
  const int O = a.O, B = a.B, T = a.T;
  const int off4 = (b * G4 + u0) * 4, off1 = (b * H + u0) * 4;

  // transposed fragments: rows = this wave's 16 hidden units (or output features), k = the contracted index
  float ffcT[KS0], f1T[G4 / 4], f0T[G4 / 4];
  stage(a.w_fc, O, H, img);                          // dh1 = W_fc^T dout:  A[unit][k = o] = W_fc[o][unit]
#pragma unroll
  for (int ks = 0; ks < KS0; ++ks) {
    const int k = 4 * ks + g;
    ffcT[ks] = (k < O) ? img[k * (H + 1) + w * 16 + c] : 0.f;
  }
  stage(a.w_ih1, G4, H, img);                        // dh0 = W_ih1^T dgates1:  A[unit][k = gate row] = W_ih1[k][unit]
#pragma unroll
  for (int ks = 0; ks < G4 / 4; ++ks) f1T[ks] = img[(4 * ks + g) * (H + 1) + w * 16 + c];
  stage(a.w_ih0, G4, O, img);                        // dx = W_ih0^T dgates0:  A[o][k = gate row] = W_ih0[k][o]
#pragma unroll
  for (int ks = 0; ks < G4 / 4; ++ks) f0T[ks] = (w * 16 + c < O) ? img[(4 * ks + g) * (O + 1) + w * 16 + c] : 0.f;

  const f32x4 c0r = load4(a.c0 + (size_t)b * H + u0, ok), c1r = load4(a.c1 + (size_t)b * H + u0, ok);
  f32x4 dc0 = zero4(), dc1 = zero4(), dx = zero4();
  int offo[4], offi[4];                              // d out_total rows (T,B,O); g_out rows in the layout the caller has
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    offo[r] = (u0 + r < O) ? (b * O + u0 + r) * 4 : OOB;
    offi[r] = (u0 + r < O) ? ((a.g_out_bt ? b * T * O : b * O) + u0 + r) * 4 : OOB;
  }
  if (hashed) drop_keys(rng, true);
  const bool has_drop = a.drop != nullptr || hashed;

  // The rows a step reads (loss gradient, saved gates of both layers, dropout mask) are requested at the top of the previous
  // step and pinned at its end (a round trip to rows of a fresh 8 MB tensor measured longer than one 64-MFMA chain):
  // requested at their use, each of the three groups would put that round trip on the critical path of the step.
  struct Saved { f32x4 go, a1[4], a0[4], m, mh; float forced; };
  auto fetch = [&](int t, Saved &s) {
    s.forced = bload1(step_rows(a.force, t, B, 1), b * 4);
    const __amdgpu_buffer_rsrc_t rg = a.g_out_bt ? bt_rows(a.g_out, t, B, T, O) : step_rows(a.g_out, t, B, O);
    const __amdgpu_buffer_rsrc_t r1 = step_rows(a.acts1, t, B, G4), r0 = step_rows(a.acts0, t, B, G4);
#pragma unroll
    for (int r = 0; r < 4; ++r) s.go[r] = bload1(rg, offi[r]);
#pragma unroll
    for (int q = 0; q < 4; ++q) s.a1[q] = bload4(r1, off4 + q * H * 4), s.a0[q] = bload4(r0, off4 + q * H * 4);
    s.m = bload4(step_rows(a.drop, t, B, H), off1);
    s.mh = (f32x4){1.f, 1.f, 1.f, 1.f};              // (a value of its own, chosen at the use: see the forward)
    if (hashed) s.mh = drop_value4(rng, (uint32_t)((t * B + b) * H + u0));
  };
  auto pin_all = [&](Saved &s) {
    pin(s.go), pin(s.m);
    asm volatile("" : "+v"(s.forced));
#pragma unroll
    for (int q = 0; q < 4; ++q) pin(s.a1[q]), pin(s.a0[q]);
  };
  Saved nx = {};
  if (T > 0) fetch(T - 1, nx);
  pin_all(nx);

  for (int t = T - 1; t >= 0; --t) {
    const Saved sv = nx;
    fetch(t > 0 ? t - 1 : 0, nx);                    // (the last step re-reads its own rows: no branch in the body)
    // ---- d out_t (loss + the next step's input gradient), this wave's 16 output features
    f32x4 dout = dx + sv.go;                         // (features >= O: zero fragments gave dx = 0, the OOB load gave 0)
    if (sv.forced != 0.f) dout = zero4();            // a forced frame is the target: no gradient reaches the decoder through it
    const __amdgpu_buffer_rsrc_t rt = step_rows(a.g_outtot, t, B, O);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      doT[(u0 + r) * TP + c] = dout[r];
      bstore1(rt, offo[r], dout[r]);
    }
    lds_barrier();
    // ---- fc backward: dh1 for this wave's 16 hidden units
    f32x4 dh = zero4();
#pragma unroll
    for (int ks = 0; ks < KS0; ++ks) dh = __builtin_amdgcn_mfma_f32_16x16x4f32(ffcT[ks], doT[(4 * ks + g) * TP + c], dh, 0, 0, 0);
    f32x4 pi, pf, pg, po;
    cell_bwd(dh, sv.a1[0], sv.a1[1], sv.a1[2], sv.a1[3], c1r, pi, pf, pg, po, dc1);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      dg1T[(u0 + r) * TP + c] = pi[r], dg1T[(H + u0 + r) * TP + c] = pf[r];
      dg1T[(2 * H + u0 + r) * TP + c] = pg[r], dg1T[(3 * H + u0 + r) * TP + c] = po[r];
    }
    store_gates(step_rows(a.g_gates1, t, B, G4), off4, pi, pf, pg, po);
    lds_barrier();
    // ---- layer-1 input gradient: dh0 (two accumulators halve the dependent chain)
    f32x4 e0 = zero4(), e1 = zero4();
#pragma unroll
    for (int ks = 0; ks < G4 / 4; ks += 2) {
      e0 = __builtin_amdgcn_mfma_f32_16x16x4f32(f1T[ks], dg1T[(4 * ks + g) * TP + c], e0, 0, 0, 0);
      e1 = __builtin_amdgcn_mfma_f32_16x16x4f32(f1T[ks + 1], dg1T[(4 * ks + 4 + g) * TP + c], e1, 0, 0, 0);
    }
    dh = e0 + e1;
    if (has_drop) dh *= hashed ? sv.mh : sv.m;
    cell_bwd(dh, sv.a0[0], sv.a0[1], sv.a0[2], sv.a0[3], c0r, pi, pf, pg, po, dc0);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      dg0T[(u0 + r) * TP + c] = pi[r], dg0T[(H + u0 + r) * TP + c] = pf[r];
      dg0T[(2 * H + u0 + r) * TP + c] = pg[r], dg0T[(3 * H + u0 + r) * TP + c] = po[r];
    }
    store_gates(step_rows(a.g_gates0, t, B, G4), off4, pi, pf, pg, po);
    lds_barrier();
    // ---- layer-0 input gradient = gradient of the previous step's output
    e0 = zero4(), e1 = zero4();
#pragma unroll
    for (int ks = 0; ks < G4 / 4; ks += 2) {
      e0 = __builtin_amdgcn_mfma_f32_16x16x4f32(f0T[ks], dg0T[(4 * ks + g) * TP + c], e0, 0, 0, 0);
      e1 = __builtin_amdgcn_mfma_f32_16x16x4f32(f0T[ks + 1], dg0T[(4 * ks + 4 + g) * TP + c], e1, 0, 0, 0);
    }
    dx = e0 + e1;
    pin_all(nx);
  }
  if (ok) {
    *reinterpret_cast<f32x4 *>(a.g_c0 + (size_t)b * H + u0) = dc0;
    *reinterpret_cast<f32x4 *>(a.g_c1 + (size_t)b * H + u0) = dc1;
  }
}

// ---- narrow variants: 4 clips per workgroup (v_mfma_f32_4x4x1_16B_f32) -----------------------------------------------------
// Same reasoning as p2c_lstm.hip's narrow kernels: at B = 512 the 16-clip tiling runs 32 workgroups, each queueing ~140
// 32-cycle MFMAs per time step on its SIMDs. With 16 blocks of 4 x 4 x 1 per instruction a workgroup owns 4 clips (the 4
// columns, shared by all blocks) and a wave 16 hidden units / output features (the blocks):
//   gate products (forward): block = unit, the 4 rows = its gates i, f, g, o  -> lane (unit, clip) holds all four;
//   transposed products (fc, and every product of the backward): block = (row group of 4, quarter of K); the quarters are
//     summed across lanes with two row-rotate DPP adds and lane (group ug, quarter q) keeps row 4 ug + q -- so in every
//     stage lane (block k, column s) owns element (16 w + k, clip s).
constexpr int NS = 4, HP = H + 4, GP = 4 * HP, OP = OMAX + 4;   // LDS pitches (16-byte aligned): a clip's h row, its d-gates row
// (gate j at j * HP: with the clip pitch 4 * HP = 16 banks mod 64 the 16 (clip, quarter) rows a wave reads start 4 banks apart)

__device__ __forceinline__ f32x4 mfma4(float a_, float b_, f32x4 c_) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a_, b_, c_, 0, 0, 0); }
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  const int r = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false);
  return v + __builtin_bit_cast(float, r);
}
// init + sum_k fr[k] * brow[k] on every block, four interleaved accumulators; brow = this lane's clip row in LDS
template <int N>
__device__ __forceinline__ f32x4 block_product(const float (&fr)[N], const float *brow, f32x4 init) {
  f32x4 e0 = init, e1 = zero4(), e2 = e1, e3 = e1;
  const f32x4 *bp = reinterpret_cast<const f32x4 *>(brow);
#pragma unroll
  for (int k4 = 0; k4 < N / 4; ++k4) {
    const f32x4 v = bp[k4];
    e0 = mfma4(fr[4 * k4], v[0], e0), e1 = mfma4(fr[4 * k4 + 1], v[1], e1);
    e2 = mfma4(fr[4 * k4 + 2], v[2], e2), e3 = mfma4(fr[4 * k4 + 3], v[3], e3);
  }
  return (e0 + e1) + (e2 + e3);
}
// the K-quarter form: blocks (ug, q) hold partial sums of rows 4 ug .. 4 ug + 3; returns row 4 ug + q of the full sum
template <int N>
__device__ __forceinline__ float quarter_product(const float (&fr)[N], const float *brow, int q) {
  f32x4 e = block_product<N>(fr, brow, zero4());
#pragma unroll
  for (int i = 0; i < 4; ++i) e[i] = dpp_add<0x128>(dpp_add<0x124>(e[i]));   // + row_ror:4, then + row_ror:8
  return (q == 0) ? e[0] : (q == 1) ? e[1] : (q == 2) ? e[2] : e[3];
}
__device__ __forceinline__ void cell_fwd1(const f32x4 &acc, float c_enc, f32x4 &act, float &h) {
  act[0] = sigmoidf_(acc[0]), act[1] = sigmoidf_(acc[1]), act[2] = tanhf_(acc[2]), act[3] = sigmoidf_(acc[3]);
  h = act[3] * tanhf_(act[1] * c_enc + act[0] * act[2]);
}
__device__ __forceinline__ f32x4 cell_bwd1(float dh, const f32x4 &act, float c_enc, float &dc_acc) {
  const float ai = act[0], af = act[1], ag = act[2], ao = act[3];
  const float tc = tanhf_(af * c_enc + ai * ag);
  const float dct = dh * ao * (1.f - tc * tc);
  f32x4 p;
  p[3] = dh * tc * ao * (1.f - ao);
  p[0] = dct * ag * ai * (1.f - ai);
  p[1] = dct * c_enc * af * (1.f - af);
  p[2] = dct * ai * (1.f - ag * ag);
  dc_acc += dct * af;
  return p;
}
__device__ __forceinline__ void store_gates1(__amdgpu_buffer_rsrc_t rs, int offg, const f32x4 &v) {
  bstore1(rs, offg, v[0]), bstore1(rs, offg + H * 4, v[1]), bstore1(rs, offg + 2 * H * 4, v[2]), bstore1(rs, offg + 3 * H * 4, v[3]);
}
__device__ __forceinline__ f32x4 load_gates1(__amdgpu_buffer_rsrc_t rs, int offg) {
  f32x4 v;
  v[0] = bload1(rs, offg), v[1] = bload1(rs, offg + H * 4), v[2] = bload1(rs, offg + 2 * H * 4), v[3] = bload1(rs, offg + 3 * H * 4);
  return v;
}

template <int KS0>
__global__ __launch_bounds__(256) void decoder_fwd_narrow_kernel(const Args a) {
  DropRng rng = a.rng;                              // (its state words are requested here, behind nothing; the keys are formed behind the staging)
  const bool hashed = !a.drop && rng.state != nullptr;
  if (hashed) drop_begin(rng, false);
  constexpr int K0 = 4 * KS0;
  extern __shared__ float img[];                    // staging image of one weight matrix at a time
  __shared__ __attribute__((aligned(16))) float xs[NS][OP], h0s[NS][HP], h1s[NS][HP];
  const int lane = threadIdx.x & 63, s = lane & 3, blk = lane >> 2, q = blk & 3;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int u = w * 16 + blk;                       // this lane's hidden unit AND output feature
  const int b = blockIdx.x * NS + s;
  const bool ok = b < a.B;
  const int O = a.O, B = a.B, T = a.T;
  const int offg = (b * G4 + u) * 4, offh = (b * H + u) * 4, offo = (u < O) ? (b * O + u) * 4 : OOB;
  const int offb = (u < O) ? (b * a.T * O + u) * 4 : OOB;

  f32x4 k0r, k1r;
  if (a.hid0) {   // k_l = ba_l + bb_l + hid_l W_hh_l^T: two gate products over K = H before the loop's fragments are loaded
    float tmp[H];
    h0s[s][u] = ok ? a.hid0[(size_t)b * H + u] : 0.f, h1s[s][u] = ok ? a.hid1[(size_t)b * H + u] : 0.f;
    stage(a.w_hh0, G4, H, img);
#pragma unroll
    for (int k = 0; k < H; ++k) tmp[k] = img[(s * H + u) * (H + 1) + k];
    f32x4 init;
#pragma unroll
    for (int j = 0; j < 4; ++j) init[j] = (a.b0a ? a.b0a[j * H + u] : 0.f) + (a.b0b ? a.b0b[j * H + u] : 0.f);
    k0r = block_product<H>(tmp, h0s[s], init);
    stage(a.w_hh1, G4, H, img);
#pragma unroll
    for (int k = 0; k < H; ++k) tmp[k] = img[(s * H + u) * (H + 1) + k];
#pragma unroll
    for (int j = 0; j < 4; ++j) init[j] = (a.b1a ? a.b1a[j * H + u] : 0.f) + (a.b1b ? a.b1b[j * H + u] : 0.f);
    k1r = block_product<H>(tmp, h1s[s], init);
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      k0r[j] = ok ? a.k0[(size_t)b * G4 + j * H + u] : 0.f;
      k1r[j] = ok ? a.k1[(size_t)b * G4 + j * H + u] : 0.f;
    }
  }

  float fa0[K0], fa1[H], ffc[16];
  stage(a.w_ih0, G4, O, img);                       // gate products: row (lane & 3) = gate of block = unit u
#pragma unroll
  for (int k = 0; k < K0; ++k) fa0[k] = (k < O) ? img[(s * H + u) * (O + 1) + k] : 0.f;
  stage(a.w_ih1, G4, H, img);
#pragma unroll
  for (int k = 0; k < H; ++k) fa1[k] = img[(s * H + u) * (H + 1) + k];
  stage(a.w_fc, O, H, img);                         // fc: blocks (feature group, K quarter q): row (lane & 3) -> feature 16w + 4 (blk >> 2) + (lane & 3)
  {
    const int f = w * 16 + 4 * (blk >> 2) + s;
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) ffc[kk] = (f < O) ? img[f * (H + 1) + q * 16 + kk] : 0.f;
  }
  const float c0r = ok ? a.c0[(size_t)b * H + u] : 0.f, c1r = ok ? a.c1[(size_t)b * H + u] : 0.f;
  const float bfc = (u < O) ? a.b_fc[u] : 0.f;
  xs[s][u] = (a.x0 && ok && u < O) ? a.x0[(size_t)b * O + u] : 0.f;    // features >= O stay zero
  if (hashed) drop_keys(rng, false);
  auto mask_of = [&](const int t) -> float { return bload1(step_rows(a.drop, t, B, H), offh); };    // (see decoder_fwd_kernel)
  auto hash_of = [&](const int t) -> float {
    float m = 1.f;
    if (hashed) m = drop_value(rng, (uint32_t)((t * B + b) * H + u));
    return m;
  };
  float maskh = hash_of(0);
  float mask = mask_of(0);
  __syncthreads();
  asm volatile("" : "+v"(mask));
  const bool has_drop = a.drop != nullptr || hashed;

  for (int t = 0; t < T; ++t) {
    const float m = hashed ? maskh : mask;
    mask = mask_of((t + 1 < T) ? t + 1 : t), maskh = hash_of((t + 1 < T) ? t + 1 : t);
    const float forced = bload1(step_rows(a.force, t, B, 1), b * 4);     // teacher forcing (NULL tensors read as zero)
    const float tgt = bload1(step_rows(a.target, t, B, O), offo);
    f32x4 act;
    float h;
    // ---- layer 0
    cell_fwd1(block_product<K0>(fa0, xs[s], k0r), c0r, act, h);
    if (has_drop) h *= m;
    h0s[s][u] = h;
    asm volatile("" : "+v"(mask));                   // resident before the stores: the wait covers the one load only
    store_gates1(step_rows(a.acts0, t, B, G4), offg, act);
    bstore1(step_rows(a.h0d, t, B, H), offh, h);
    lds_barrier();
    // ---- layer 1
    cell_fwd1(block_product<H>(fa1, h0s[s], k1r), c1r, act, h);
    h1s[s][u] = h;
    store_gates1(step_rows(a.acts1, t, B, G4), offg, act);
    bstore1(step_rows(a.h1, t, B, H), offh, h);
    lds_barrier();
    // ---- fc
    float o = quarter_product<16>(ffc, h1s[s] + q * 16, q) + bfc;
    if (forced != 0.f) o = tgt;                      // (features >= O: the out-of-range load gave zero)
    xs[s][u] = o;                                    // next step's input (features >= O are exactly zero)
    bstore1(step_rows(a.out, t, B, O), offo, o);
    bstore1(bt_rows(a.out_bt, t, B, T, O), offb, o);
    lds_barrier();
  }
}

template <int KS0>
__global__ __launch_bounds__(256) void decoder_bwd_narrow_kernel(const Args a) {
  DropRng rng = a.rng;                              // (its state words are requested here, behind nothing; the keys are formed behind the staging)
  const bool hashed = !a.drop && rng.state != nullptr;
  if (hashed) drop_begin(rng, true);
  extern __shared__ float img[];
  __shared__ __attribute__((aligned(16))) float dos[NS][OP], dg1s[NS][GP], dg0s[NS][GP];
  const int lane = threadIdx.x & 63, s = lane & 3, blk = lane >> 2, q = blk & 3;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int u = w * 16 + blk;                       // this lane's hidden unit AND output feature
  const int ra_ = w * 16 + 4 * (blk >> 2) + s;      // the A-operand row of this lane (a unit or an output feature)
  const int b = blockIdx.x * NS + s;
  const bool ok = b < a.B;
  const int O = a.O, B = a.B, T = a.T;
  const int offg = (b * G4 + u) * 4, offh = (b * H + u) * 4, offo = (u < O) ? (b * O + u) * 4 : OOB;
  const int offi = (u < O) ? ((a.g_out_bt ? b * a.T * O : b * O) + u) * 4 : OOB;

  float ffcT[16], f1T[H], f0T[H];
  stage(a.w_fc, O, H, img);                          // dh1 = W_fc^T dout: rows = units, K quarter q = features 16 q .. 16 q + 15
#pragma unroll
  for (int kk = 0; kk < 16; ++kk) ffcT[kk] = (q * 16 + kk < O) ? img[(q * 16 + kk) * (H + 1) + ra_] : 0.f;
  stage(a.w_ih1, G4, H, img);                        // dh0 = W_ih1^T dgates1: rows = units, K quarter q = gate q's H rows
#pragma unroll
  for (int k = 0; k < H; ++k) f1T[k] = img[(q * H + k) * (H + 1) + ra_];
  stage(a.w_ih0, G4, O, img);                        // dx = W_ih0^T dgates0: rows = output features
#pragma unroll
  for (int k = 0; k < H; ++k) f0T[k] = (ra_ < O) ? img[(q * H + k) * (O + 1) + ra_] : 0.f;

  const float c0r = ok ? a.c0[(size_t)b * H + u] : 0.f, c1r = ok ? a.c1[(size_t)b * H + u] : 0.f;
  float dc0 = 0.f, dc1 = 0.f, dx = 0.f;
  f32x4 dk0 = zero4(), dk1 = zero4();                // sum_t d gates_l = the gradient of k_l
  if (hashed) drop_keys(rng, true);
  const bool has_drop = a.drop != nullptr || hashed;

  struct Saved { f32x4 a1, a0; float go, m, mh, forced; };
  auto fetch = [&](int t, Saved &sv) {
    sv.forced = bload1(step_rows(a.force, t, B, 1), b * 4);
    sv.go = bload1(a.g_out_bt ? bt_rows(a.g_out, t, B, T, O) : step_rows(a.g_out, t, B, O), offi);
    sv.a1 = load_gates1(step_rows(a.acts1, t, B, G4), offg);
    sv.a0 = load_gates1(step_rows(a.acts0, t, B, G4), offg);
    sv.m = bload1(step_rows(a.drop, t, B, H), offh);
    sv.mh = 1.f;
    if (hashed) sv.mh = drop_value(rng, (uint32_t)((t * B + b) * H + u));
  };
  auto pin_all = [&](Saved &sv) {
    pin(sv.a1), pin(sv.a0);
    asm volatile("" : "+v"(sv.go), "+v"(sv.m), "+v"(sv.forced));
  };
  Saved nx = {};
  if (T > 0) fetch(T - 1, nx);
  __syncthreads();
  pin_all(nx);

  for (int t = T - 1; t >= 0; --t) {
    const Saved sv = nx;
    fetch(t > 0 ? t - 1 : 0, nx);                    // (the last step re-reads its own rows: no branch in the body)
    // ---- d out_t (loss + the next step's input gradient)
    const float dout = sv.forced != 0.f ? 0.f : dx + sv.go;   // features >= O: zero fragments gave dx = 0, the OOB load 0;
                                                              // a forced frame is the target: no gradient through it
    dos[s][u] = dout;
    bstore1(step_rows(a.g_outtot, t, B, O), offo, dout);
    lds_barrier();
    // ---- fc backward, layer-1 cell
    float dh = quarter_product<16>(ffcT, dos[s] + q * 16, q);
    f32x4 p = cell_bwd1(dh, sv.a1, c1r, dc1);
    dk1 += p;
    dg1s[s][u] = p[0], dg1s[s][HP + u] = p[1], dg1s[s][2 * HP + u] = p[2], dg1s[s][3 * HP + u] = p[3];
    store_gates1(step_rows(a.g_gates1, t, B, G4), offg, p);
    lds_barrier();
    // ---- layer-1 input gradient, layer-0 cell
    dh = quarter_product<H>(f1T, dg1s[s] + q * HP, q);
    if (has_drop) dh *= hashed ? sv.mh : sv.m;
    p = cell_bwd1(dh, sv.a0, c0r, dc0);
    dk0 += p;
    dg0s[s][u] = p[0], dg0s[s][HP + u] = p[1], dg0s[s][2 * HP + u] = p[2], dg0s[s][3 * HP + u] = p[3];
    store_gates1(step_rows(a.g_gates0, t, B, G4), offg, p);
    lds_barrier();
    // ---- layer-0 input gradient = gradient of the previous step's output
    dx = quarter_product<H>(f0T, dg0s[s] + q * HP, q);
    pin_all(nx);
  }
  if (ok) {
    a.g_c0[(size_t)b * H + u] = dc0;
    a.g_c1[(size_t)b * H + u] = dc1;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (a.g_k0) a.g_k0[(size_t)b * G4 + j * H + u] = dk0[j];
      if (a.g_k1) a.g_k1[(size_t)b * G4 + j * H + u] = dk1[j];
    }
  }
  if (a.g_hid0) {   // d hid_l = g_k_l W_hh_l: the transposed (K-quarter) product once more, on the summed d gates
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) dg0s[s][j * HP + u] = dk0[j], dg1s[s][j * HP + u] = dk1[j];
    stage(a.w_hh0, G4, H, img);
#pragma unroll
    for (int k = 0; k < H; ++k) f1T[k] = img[(q * H + k) * (H + 1) + ra_];
    const float gh0 = quarter_product<H>(f1T, dg0s[s] + q * HP, q);
    stage(a.w_hh1, G4, H, img);
#pragma unroll
    for (int k = 0; k < H; ++k) f1T[k] = img[(q * H + k) * (H + 1) + ra_];
    const float gh1 = quarter_product<H>(f1T, dg1s[s] + q * HP, q);
    if (ok) a.g_hid0[(size_t)b * H + u] = gh0, a.g_hid1[(size_t)b * H + u] = gh1;
  }
}

// ---- helpers of the 16-clip tiling for the in-library k_l / d k_l / d hid_l (B > 4096: three more launches are noise there)
__global__ void decoder_k_kernel(const float *hid, const float *w_hh, const float *ba, const float *bb, float *k, int B) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)B * G4) return;
  const int64_t b = i / G4;
  const int row = (int)(i - b * G4);
  float s = (ba ? ba[row] : 0.f) + (bb ? bb[row] : 0.f);
  for (int kk = 0; kk < H; ++kk) s = fmaf(hid[b * H + kk], w_hh[row * H + kk], s);
  k[i] = s;
}
__global__ void decoder_gk_kernel(const float *gg, float *gk, int T, int B) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, n = (int64_t)B * G4;
  if (i >= n) return;
  float s = 0.f;
  for (int t = 0; t < T; ++t) s += gg[(int64_t)t * n + i];
  gk[i] = s;
}
__global__ void decoder_ghid_kernel(const float *gk, const float *w_hh, float *ghid, int B) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)B * H) return;
  const int64_t b = i / H;
  const int u = (int)(i - b * H);
  float s = 0.f;
  for (int row = 0; row < G4; ++row) s = fmaf(gk[b * G4 + row], w_hh[row * H + u], s);
  ghid[i] = s;
}

}  // namespace p2c_s2s

using namespace p2c_s2s;

static int fill(Args &a, const p2c_decoder_desc *d) {
  if (!d || !d->c0 || !d->c1 || !d->w_ih0 || !d->w_ih1 || !d->w_fc || !d->b_fc) return P2C_E_NULL;
  if (d->hid0 ? (!d->hid1 || !d->w_hh0 || !d->w_hh1) : (!d->k0 || !d->k1)) return P2C_E_NULL;
  if ((d->g_hid0 || d->g_hid1) && (!d->g_hid0 || !d->g_hid1 || !d->w_hh0 || !d->w_hh1)) return P2C_E_NULL;
  if (d->T < 0 || d->B < 0 || d->B > (1 << 20) || d->H != H || d->O < 1 || d->O > OMAX) return P2C_E_SHAPE;
  if ((d->out_bt || d->g_out_bt) && (int64_t)d->B * d->T * d->O * 4 >= (int64_t)1 << 31) return P2C_E_SHAPE;
  a = Args{};
  a.k0 = d->k0, a.c0 = d->c0, a.k1 = d->k1, a.c1 = d->c1, a.w_ih0 = d->w_ih0, a.w_ih1 = d->w_ih1, a.w_fc = d->w_fc;
  a.b_fc = d->b_fc, a.x0 = d->x0, a.drop = d->drop, a.out = d->out, a.acts0 = d->acts0, a.acts1 = d->acts1, a.h0d = d->h0d;
  a.h1 = d->h1, a.g_out = d->g_out, a.g_gates0 = d->g_gates0, a.g_gates1 = d->g_gates1, a.g_outtot = d->g_outtot;
  a.g_c0 = d->g_c0, a.g_c1 = d->g_c1, a.T = d->T, a.B = d->B, a.O = d->O;
  a.hid0 = d->hid0, a.hid1 = d->hid1, a.w_hh0 = d->w_hh0, a.w_hh1 = d->w_hh1, a.b0a = d->b0a, a.b0b = d->b0b, a.b1a = d->b1a;
  a.b1b = d->b1b, a.kw0 = d->kw0, a.kw1 = d->kw1, a.out_bt = d->out_bt, a.g_k0 = d->g_k0, a.g_k1 = d->g_k1;
  a.g_hid0 = d->g_hid0, a.g_hid1 = d->g_hid1, a.g_out_bt = d->g_out_bt;
  if ((d->force != nullptr) != (d->target != nullptr)) return P2C_E_NULL;
  a.force = d->force, a.target = d->target;
  if (d->drop_state) {
    if (!(d->drop_p >= 0.f && d->drop_p < 1.f)) return P2C_E_SHAPE;
    a.rng.state = d->drop_state, a.rng.site = d->drop_site, a.rng.scale = 1.f / (1.f - d->drop_p);
    a.rng.thresh = (uint32_t)((double)d->drop_p * 4294967296.0);
  }
  return 0;
}

// 4 clips per workgroup up to B = 4096, 16 above; P2C_REC_TILE=wide|narrow forces one (tests run both)
static bool use_narrow(int B) {
  const char *e = getenv("P2C_REC_TILE");
  if (e && !strcmp(e, "wide")) return false;
  if (e && !strcmp(e, "narrow")) return true;
  return B <= 4096;
}

static void allow_lds() {
  static bool done = false;
  if (done) return;
  (void)hipFuncSetAttribute((const void *)decoder_fwd_kernel<13>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  (void)hipFuncSetAttribute((const void *)decoder_fwd_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  (void)hipFuncSetAttribute((const void *)decoder_bwd_kernel<13>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  (void)hipFuncSetAttribute((const void *)decoder_bwd_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  (void)hipFuncSetAttribute((const void *)decoder_fwd_narrow_kernel<13>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  (void)hipFuncSetAttribute((const void *)decoder_fwd_narrow_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  (void)hipFuncSetAttribute((const void *)decoder_bwd_narrow_kernel<13>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  (void)hipFuncSetAttribute((const void *)decoder_bwd_narrow_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  done = true;
}
static size_t image_bytes() { return sizeof(float) * (size_t)G4 * (H + 1); }   // the largest staged matrix (4H x H, O <= H)

extern "C" int p2c_decoder_fwd(const p2c_decoder_desc *d, void *stream) {
  Args a;
  int rc = fill(a, d);
  if (rc) return rc;
  if (!a.out || !a.acts0 || !a.acts1 || !a.h0d || !a.h1) return P2C_E_NULL;
  if (a.B == 0 || a.T == 0) return 0;
  allow_lds();
  if (use_narrow(a.B)) {
    const dim3 grid((unsigned)((a.B + NS - 1) / NS));
    if (a.O <= 52) hipLaunchKernelGGL(decoder_fwd_narrow_kernel<13>, grid, dim3(256), image_bytes(), (hipStream_t)stream, a);
    else hipLaunchKernelGGL(decoder_fwd_narrow_kernel<16>, grid, dim3(256), image_bytes(), (hipStream_t)stream, a);
  } else {
    if (a.hid0) {                                  // k_l through two helper launches into the caller's scratch
      if (!a.kw0 || !a.kw1) return P2C_E_NULL;
      const unsigned nb = (unsigned)(((int64_t)a.B * G4 + 255) / 256);
      hipLaunchKernelGGL(decoder_k_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, a.hid0, a.w_hh0, a.b0a, a.b0b, a.kw0, a.B);
      hipLaunchKernelGGL(decoder_k_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, a.hid1, a.w_hh1, a.b1a, a.b1b, a.kw1, a.B);
      a.k0 = a.kw0, a.k1 = a.kw1;
    }
    const dim3 grid((unsigned)((a.B + TS - 1) / TS));
    if (a.O <= 52) hipLaunchKernelGGL(decoder_fwd_kernel<13>, grid, dim3(256), image_bytes(), (hipStream_t)stream, a);
    else hipLaunchKernelGGL(decoder_fwd_kernel<16>, grid, dim3(256), image_bytes(), (hipStream_t)stream, a);
  }
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

extern "C" int p2c_decoder_bwd(const p2c_decoder_desc *d, void *stream) {
  Args a;
  int rc = fill(a, d);
  if (rc) return rc;
  if (!a.g_out || !a.acts0 || !a.acts1 || !a.g_gates0 || !a.g_gates1 || !a.g_outtot || !a.g_c0 || !a.g_c1) return P2C_E_NULL;
  if (a.B == 0) return 0;
  allow_lds();
  if (use_narrow(a.B)) {
    const dim3 grid((unsigned)((a.B + NS - 1) / NS));
    if (a.O <= 52) hipLaunchKernelGGL(decoder_bwd_narrow_kernel<13>, grid, dim3(256), image_bytes(), (hipStream_t)stream, a);
    else hipLaunchKernelGGL(decoder_bwd_narrow_kernel<16>, grid, dim3(256), image_bytes(), (hipStream_t)stream, a);
  } else {
    const dim3 grid((unsigned)((a.B + TS - 1) / TS));
    if (a.O <= 52) hipLaunchKernelGGL(decoder_bwd_kernel<13>, grid, dim3(256), image_bytes(), (hipStream_t)stream, a);
    else hipLaunchKernelGGL(decoder_bwd_kernel<16>, grid, dim3(256), image_bytes(), (hipStream_t)stream, a);
    if (a.g_hid0 && (!a.g_k0 || !a.g_k1)) return P2C_E_NULL;      // (d hid_l is formed from the stored d k_l here)
    const unsigned nk = (unsigned)(((int64_t)a.B * G4 + 255) / 256), nh = (unsigned)(((int64_t)a.B * H + 255) / 256);
    if (a.g_k0) hipLaunchKernelGGL(decoder_gk_kernel, dim3(nk), dim3(256), 0, (hipStream_t)stream, a.g_gates0, a.g_k0, a.T, a.B);
    if (a.g_k1) hipLaunchKernelGGL(decoder_gk_kernel, dim3(nk), dim3(256), 0, (hipStream_t)stream, a.g_gates1, a.g_k1, a.T, a.B);
    if (a.g_hid0) {
      hipLaunchKernelGGL(decoder_ghid_kernel, dim3(nh), dim3(256), 0, (hipStream_t)stream, a.g_k0, a.w_hh0, a.g_hid0, a.B);
      hipLaunchKernelGGL(decoder_ghid_kernel, dim3(nh), dim3(256), 0, (hipStream_t)stream, a.g_k1, a.w_hh1, a.g_hid1, a.B);
    }
  }
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}
