// Fused co-attention kernels (coattn_fused.hip): entry points used by api.hip.
#pragma once
#include "common.h"

int fused_supported(int B, int N, int T, int d, int L);
// everything after the projections (P_v, P_q already in `saved`)
int fused_attention_forward(int B, int N, int T, int d, int L, const float* V, const float* const* Q,
                            const coattn_params* p, float* v_out, float* q_out, float* saved, float* ws,
                            hipStream_t s);
int fused_backward_supported(int B, int N, int T, int d, int L);
int fused_backward(int B, int N, int T, int d, int L, const float* V, const float* const* Q, const coattn_params* p,
                   const float* saved, const float* gv, const float* gq, float* dV, float* const* dQ,
                   const coattn_param_grads* pg, int accumulate, float* ws, hipStream_t s, int bf16_proj);

// ---- shared by the fused forward / backward translation units -------------------------------
constexpr int kTS = 7;       // k-steps of 4 over T: T <= 28
constexpr int kTRows = 28;   // rows of C kept in LDS
constexpr int kSlotRows = 32; // rows of a cross-wave reduction slot (both 16-row MFMA tiles)

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));      // MFMA 16x16x32 bf16 operand: 8 k-values per lane
typedef __bf16 bfv8 __attribute__((ext_vector_type(8)));

// buffer resource over [ptr, ptr + bytes): out-of-range loads return 0, stores are dropped
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* ptr, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(ptr), 0, bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ void buf_store4(f32x4 v, __amdgpu_buffer_rsrc_t r, int voff, int soff) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff, soff, 0);
}
__device__ __forceinline__ float buf_load1(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}

// XCD-aware block -> (b, l): blocks i and i+8 share an XCD (round-robin dispatch), so give the
// L levels of one sample consecutive slots on one XCD.  Speed only; any mapping is correct.
__device__ __forceinline__ bool block_to_pair(int bid, int B, int L, int& b, int& l) {
  const int x = bid & 7, slot = bid >> 3;
  b = (slot / L) * 8 + x;
  l = slot % L;
  return b < B;
}

// arguments of the fused forward kernel (coattn_fused.hip)
struct FwdArgs {
  const float* V;        // [B][d][N]
  const float* Q[8];     // L x [B][T][d]
  const float* Pv;       // [B][N][d]
  const float* Pq;       // [L][B][T][d]
  const float* wv; const float* cv; const float* wq; const float* cq;
  float* C;              // [L][B][T][N]
  float* av;             // [L][B][N]
  float* aq;             // [L][B][T]
  float* Hq;             // [L][B][T][d]
  float* q_out;          // [L][B][d]
  unsigned long long* stamps;   // diagnostic builds only
  int B, N, T, d, L;
};

inline size_t fal64(size_t n) { return (n + 63) & ~(size_t)63; }

struct SavedOff {
  size_t Pv, Pq, C, av, aq, Hq, total;
};
inline SavedOff saved_off(int B, int N, int T, int d, int L) {   // the one layout of `saved`
  SavedOff p;
  size_t o = 0;
  p.Pv = o; o += fal64((size_t)B * N * d);
  p.Pq = o; o += fal64((size_t)L * B * T * d);
  p.C = o;  o += fal64((size_t)L * B * T * N);
  p.av = o; o += fal64((size_t)L * B * N);
  p.aq = o; o += fal64((size_t)L * B * T);
  p.Hq = o; o += fal64((size_t)L * B * T * d);
  p.total = o;
  return p;
}

// workspace of the fused backward (floats)
struct FusedBwdOff {
  size_t dsv, dZq, dPq, dPv, dA, dwv_part, dbv_part, dbq_part, dwq_part, dcs_part, part, total;
};
inline FusedBwdOff fused_bwd_off(int B, int N, int T, int d, int L) {
  FusedBwdOff p;
  size_t o = 0;
  p.dsv = o; o += fal64((size_t)L * B * N);
  p.dZq = o; o += fal64((size_t)L * B * T * d);
  p.dPq = o; o += fal64((size_t)L * B * T * d);
  p.dPv = o; o += fal64((size_t)L * B * N * d);
  p.dA = o;  o += fal64((size_t)L * B * T * N);
  p.dwv_part = o; o += fal64((size_t)L * B * d);
  p.dbv_part = o; o += fal64((size_t)L * B * d);
  p.dbq_part = o; o += fal64((size_t)L * B * d);
  p.dwq_part = o; o += fal64((size_t)L * B * d);
  p.dcs_part = o; o += fal64((size_t)L * B * 2);
  // shared scratch: split-K partials of the weight-gradient GEMMs (<= 32 x d x d) and, before them, the da_v
  // partials of bwd_dav_kernel ([B][d/64][3][N]), which outgrow the former at large B
  const size_t part_gemm = (size_t)32 * d * d, part_dav = (size_t)B * (d / 64) * 3 * N;
  p.part = o; o += fal64(part_gemm > part_dav ? part_gemm : part_dav);
  p.total = o;
  return p;
}
size_t fused_bwd_ws_floats(int B, int N, int T, int d, int L);
