// p2c_rec_dev.h -- device helpers shared by the time-loop kernels (K7b p2c_lstm.hip, K7c p2c_s2s.hip).
//
// A recurrence step is ~1.3 us of MFMA + transcendental work; one exposed memory round trip per step doubles it. Two rules
// keep the loop body free of them:
//  (1) the waves exchange data through LDS only, so the barrier fences the LOCAL address space (lds_barrier) --
//      __syncthreads() on gfx9 carries s_waitcnt vmcnt(0), i.e. it waits for the write acknowledgement of the step's own
//      saved-row stores;
//  (2) global rows go through BUFFER instructions with hardware range checking instead of `if (ok)` branches: lanes of
//      sequences beyond B address past num_records (loads return 0, stores are dropped), a NULL tensor becomes a
//      zero-record descriptor. The body is then straight-line code and the compiler's waitcnt pass can count
//      (vmcnt(N) for "the loads issued before the last N stores") instead of falling back to vmcnt(0) at every join.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace p2c_rec {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// One time step's rows of a (T, B, row_floats) tensor as a raw buffer: base + t * B * row_floats, B * row_floats * 4 bytes
// of records (0 for a NULL tensor). Lane offsets are b * row_bytes + column bytes: b >= B is out of range by construction.
// The host bounds B so that every offset a lane can form stays below 2^31.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t step_rows(const float *base, int t, int B, int row_floats) {
  const uintptr_t p = reinterpret_cast<uintptr_t>(base) + (size_t)t * B * row_floats * 4;   // (NULL: no records, never used)
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(p), 0, base ? B * row_floats * 4 : 0, 0x00020000);
}
// rows of step t of a batch-first (B, T, row_floats) tensor: lane offset (b T row_floats + column) * 4; b >= B is out of range
// by construction (records end with row (B-1, t)). The host bounds B T row_floats * 4 below 2^31.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t bt_rows(const float *base, int t, int B, int T, int row_floats) {
  const uintptr_t p = reinterpret_cast<uintptr_t>(base) + (size_t)t * row_floats * 4;
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(p), 0, base ? ((B - 1) * T + 1) * row_floats * 4 : 0, 0x00020000);
}
__device__ __forceinline__ f32x4 bload4(__amdgpu_buffer_rsrc_t r, int byte_off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0));
}
__device__ __forceinline__ float bload1(__amdgpu_buffer_rsrc_t r, int byte_off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 0));
}
__device__ __forceinline__ void bstore4(__amdgpu_buffer_rsrc_t r, int byte_off, f32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, byte_off, 0, 0);
}
__device__ __forceinline__ void bstore1(__amdgpu_buffer_rsrc_t r, int byte_off, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, v), r, byte_off, 0, 0);
}
constexpr int OOB = 0x7fffff00;          // a byte offset no descriptor of these kernels reaches (host: records < 2^30)

// ---- dropout masks drawn INSIDE the time-loop kernels (nn.LSTM's inter-layer dropout, seq2seq.py:36-58 / 110-115) -------------------
// mask(e) = keep(e) / (1 - p), keep(e) = hash(seed, step, site, e) >= p 2^32 for element e of the (T, B, H) tensor: a stateless
// function of the element index, so the forward and the backward of a step draw the SAME mask without a mask tensor, a
// generator launch or the framework's RNG-state fills. `state` (device, 4 x int32, owned by the module that owns the dropout
// site): {seed_lo, seed_hi, step, next}. The forward reads `step` and leaves next = step + 1; the backward reads next - 1 and
// leaves step = next -- no kernel reads a word another workgroup of the same launch may have rewritten, and a replayed HIP
// graph advances the stream by itself. (A training-mode forward WITHOUT a backward draws the same masks again next time.)
struct DropRng {
  int32_t *state;        // NULL: no hashed mask
  uint32_t thresh;       // p 2^32
  float scale;           // 1 / (1 - p)
  int32_t site;          // distinguishes the dropout sites that share a state
  uint32_t k0, k1;       // per-launch keys (drop_begin: the raw seed words; drop_keys: the keys)
  uint32_t thresh_step;  // the step the launch draws for (drop_begin)
};
__device__ __forceinline__ uint32_t mix32(uint32_t x) {      // "lowbias32" integer finaliser: full avalanche
  x ^= x >> 16, x *= 0x7feb352du;
  x ^= x >> 15, x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}
// Two halves: the three state words are REQUESTED at the top of a kernel (drop_begin) and turned into the launch's keys behind its
// weight staging (drop_keys), where one thread of the launch also leaves the word the NEXT launch of the stream reads -- from the
// registers it already has: a load + dependent store of its own at the top of the kernel held workgroup 0 back by a memory round
// trip, i.e. the whole launch (decoder: + 2 ... 3 us).
__device__ __forceinline__ void drop_begin(DropRng &r, const bool backward) {
  if (!r.state) return;
  r.k0 = (uint32_t)r.state[0], r.k1 = (uint32_t)r.state[1];
  r.thresh_step = backward ? (uint32_t)r.state[3] - 1u : (uint32_t)r.state[2];
}
__device__ __forceinline__ void drop_keys(DropRng &r, const bool backward) {
  if (!r.state) return;
  const uint32_t s0 = r.k0, s1 = r.k1, step = r.thresh_step;
  r.k0 = mix32(s0 ^ (step * 0x9E3779B9u) ^ ((uint32_t)(r.site + 1) * 0x632BE59Bu));   // (the site in BOTH keys: with k1 alone two sites' hashes
                                                                                       // differ by a constant xor -- their masks are dependent)
  r.k1 = mix32(s1 + step + 0x85EBCA6Bu * (uint32_t)(r.site + 1));
  // forward: next = step + 1 (the backward reads it); backward: step = next (the next forward reads it). No launch reads the word
  // it writes, so the moment of the store does not matter.
  if (blockIdx.x == 0 && threadIdx.x == 0) r.state[backward ? 2 : 3] = (int32_t)(step + 1u);
}
__device__ __forceinline__ float drop_value(const DropRng &r, const uint32_t e) {
  // one Fibonacci multiply spreads the element index, the launch key shifts it, one avalanche finaliser: three 32-bit multiplies
  // on the dependent chain of a time step (two finalisers in a row were four, and ~200 cycles of latency per step)
  return (mix32(e * 0x9E3779B1u + r.k0) ^ r.k1) >= r.thresh ? r.scale : 0.f;
}
__device__ __forceinline__ f32x4 drop_value4(const DropRng &r, const uint32_t e) {
  return (f32x4){drop_value(r, e), drop_value(r, e + 1), drop_value(r, e + 2), drop_value(r, e + 3)};
}

// Values loaded one step ahead are pinned (made resident) BEFORE the step's stores are issued: the wait the compiler
// inserts for them then covers the loads only -- placed after the stores, vmcnt would also count the stores' round trip.
__device__ __forceinline__ void pin(f32x4 &v) { asm volatile("" : "+v"(v)); }

// v_exp_f32 + v_rcp_f32 (1 ulp each): the IEEE division sequence would triple the cost of the cell update
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return 2.f * __builtin_amdgcn_rcpf(1.f + __expf(-2.f * x)) - 1.f; }

}  // namespace p2c_rec
