// Fused co-attention kernels for gfx950 (fp32, exact-f32 MFMA 16x16x4).
//
// Forward ("affinity + softmax + reduce", model.py:377-392 after the projections):
//   coattn_attn_fwd_kernel : one workgroup per (sample b, level l); NW = d/128 waves.
//     phase 1  A = Q V^T, K (=d) split over the waves: each wave streams its 128 channel rows
//              of V [d][N] straight from HBM into MFMA B operands (every V element is used by
//              exactly one wave, so no LDS staging) and its 128-column slice of Q as A operands;
//              partial [T x N] tiles are summed through LDS in a fixed tree order, C = tanh(A)
//              lands in LDS (and in `saved` for backward).
//     phase 2  loop over 16-row tiles of P_v [N][d] (read once): the tile is the B operand of
//              H_q += C . P_v (contraction over N) and then the accumulator of
//              H_v = tanh(P_v + C^T P_q) (contraction over T); the wave owns a 128-wide slice of
//              d, keeps its P_q slice in registers, and folds H_v into score partials
//              s_v[n] += H_v[n][:] . w_v without ever writing H_v.
//     phase 3  cross-wave score reduction, un-masked row softmax over N and over T
//              (model.py:387-388) by wave shuffles, q = a_q^T Q, H_q saved for backward.
//   attend_v_kernel        : v_l = a_{v,l}^T V for all levels with ONE more pass over V.
//
// Register/LDS budget at d=512 (NW=4, 256 threads): <= 256 VGPRs and < 80 KB LDS per workgroup,
// so two workgroups share a CU (480 workgroups for B=160, L=3 on 256 CUs); block ids are mapped
// so that the L levels of one sample run on the same XCD (shared L2 for V and P_v).
#include "fused.h"

#include <stdlib.h>

namespace {




// c += a . b over 32 k with fp32 accuracy: the six partial products down to relative order 2^-16
// (each bf16 x bf16 product is exact in the fp32 accumulator), smallest terms first
__device__ __forceinline__ f32x4 mfma_x3(const bf16x8 (&a)[3], const bf16x8 (&b)[3], f32x4 c) {
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[2], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[0], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[1], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[1], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[0], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], c, 0, 0, 0);
  return c;
}

// X3: phase 1 (A = Q V^T) on the bf16 MFMA with the exact 3-way split of both operands (fp32-accurate,
// 6/16 of the f32-MFMA time), channels split over the waves + cross-wave sum; 0 = v_mfma_f32_16x16x4_f32.
template <int NT, int NW, int X3>
__global__ __launch_bounds__(NW * 64, 2) void coattn_attn_fwd_kernel(const FwdArgs a) {
  constexpr int NPAD = 16 * NT;
  constexpr int LD = NPAD + 4;                       // row stride of the LDS [t][n] images
  constexpr int NSLOT = (NW / 2 > 2) ? NW / 2 : 2;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* slots = lds;                                // NSLOT x kTRows x LD   (phase 1 reduction)
  float* Cbuf = lds + NSLOT * kSlotRows * LD;           // kTRows x LD
  float* svpart = slots;                             // NW x NPAD             (aliases, phase 2+)
  float* sqpart = slots + NW * NPAD;                 // NW x 32
  float* aqs = sqpart + NW * 32;                     // 32

  int b, l;
  if (!block_to_pair(blockIdx.x, a.B, a.L, b, l)) return;
  CA_STAMP(0);
  const int N = a.N, T = a.T, d = a.d;
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, q4 = lane >> 4;           // MFMA column / k-quad (also C/D row quad)
  const float* Qp = a.Q[l] + (size_t)b * T * d;
  const float* Vp = a.V + (size_t)b * a.v_sB;
  const float* Pvp = a.Pv + (size_t)b * N * d;
  const float* Pqp = a.Pq + ((size_t)l * a.B + b) * T * d;
  const size_t pair = (size_t)l * a.B + b;
  // buffer resources over exactly this sample's tensors: rows beyond T / N read as 0
  const __amdgpu_buffer_rsrc_t rs_q = make_rsrc(Qp, (unsigned)T * d * 4u);
  const __amdgpu_buffer_rsrc_t rs_v = make_rsrc(Vp, (unsigned)d * N * 4u);
  const __amdgpu_buffer_rsrc_t rs_pv = make_rsrc(Pvp, (unsigned)N * d * 4u);

  // A wave owns the 128-channel slices (sl * NW + w), sl = 0 .. d / (128 NW) - 1 (one slice at d = 512).
  const int nsl = d / (128 * NW);
  // phase-2 operands (a 128-channel slice of P_q as MFMA B operands, w_v): the first slice is loaded at
  // the end of phase 1 so that the loads fly under the cross-wave reduction
  float pq[kTS][8];                                  // B operand P_q[t = 4s + q4][dsl + 16c + j]
  float wvr[8];
  const __amdgpu_buffer_rsrc_t rs_pq = make_rsrc(Pqp, (unsigned)T * d * 4u);     // rows >= T read 0
  auto load_slice_operands = [&](int dsl) {
#pragma unroll
    for (int s = 0; s < kTS; ++s)
#pragma unroll
      for (int c = 0; c < 8; ++c) pq[s][c] = buf_load1(rs_pq, ((4 * s + q4) * d + j) * 4 + 64 * c, dsl * 4);
#pragma unroll
    for (int c = 0; c < 8; ++c) wvr[c] = a.wv[dsl + 16 * c + j];
  };

  // ------------------------------------------------------------------ phase 1: A = Q V^T
  {
    f32x4 acc[2][NT];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[tt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (X3) {
      // bf16 MFMA 16x16x32: lane (row/col = j, k-group q4) holds 8 consecutive k = channels 8*q4 .. 8*q4+7
      // of a 32-channel step.  A operand: Q[t = 16tt + j][k]; B operand: V[k][n = 16 tile + j].
      constexpr int RING = 4;                        // V tiles in flight (8 dwords each)
      const int q_voff0 = (j * d + 8 * q4) * 4, q_voff1 = ((16 + j) * d + 8 * q4) * 4;
      const int v_voff = (8 * q4 * N + j) * 4;
      f32x8 vr[RING];                                // ring over the location tiles of the current step
      f32x8 vn[RING - 1];                            // first tiles of the next step, loaded ahead
      f32x8 qr[2];
      bf16x8 qa[2][3];
      // g enumerates the 32-channel steps of this wave: 4 per 128-channel slice
      const int G = 4 * nsl;
      auto chan0 = [&](int g) { return ((g >> 2) * NW + w) * 128 + 32 * (g & 3); };
      auto load_q = [&](int k0) {
        const f32x4 a0 = buf_load4(rs_q, q_voff0, k0 * 4), a1 = buf_load4(rs_q, q_voff0 + 16, k0 * 4);
        const f32x4 b0 = buf_load4(rs_q, q_voff1, k0 * 4), b1 = buf_load4(rs_q, q_voff1 + 16, k0 * 4);
        qr[0] = f32x8{a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
        qr[1] = f32x8{b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
      };
      auto load_v = [&](int k0, int t, f32x8& dst) {
#pragma unroll
        for (int i = 0; i < 8; ++i)                  // cols >= N: finite junk, zeroed when C is finalised
          dst[i] = buf_load1(rs_v, v_voff + 64 * t, (k0 + i) * N * 4);
      };
      load_q(chan0(0));
#pragma unroll
      for (int t = 0; t < RING - 1; ++t) load_v(chan0(0), t, vr[t]);
#pragma unroll 1
      for (int g = 0; g < G; ++g) {
        const int k0 = chan0(g), k1 = chan0(g + 1);
        const bool more = g + 1 < G;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          if (t + RING - 1 < NT) load_v(k0, t + RING - 1, vr[(t + RING - 1) % RING]);
          else if (more) load_v(k1, t + RING - 1 - NT, vn[t + RING - 1 - NT]);
          if (t == 0) {
            split3(qr[0], qa[0]);
            split3(qr[1], qa[1]);
            if (more) load_q(k1);
          }
          __builtin_amdgcn_sched_barrier(0);         // keep the prefetch ahead of this step's work
          bf16x8 vb[3];
          split3(vr[t % RING], vb);
          acc[0][t] = mfma_x3(qa[0], vb, acc[0][t]);
          acc[1][t] = mfma_x3(qa[1], vb, acc[1][t]);
        }
#pragma unroll
        for (int t = 0; t < RING - 1; ++t) vr[t] = vn[t];
      }
    } else {
    constexpr int RING = 4;                          // V operand ring: 3 k-steps in flight
    float vb[RING][NT];
    f32x4 qa[2][2];
    int ks = w * 128;                                // first channel of the current slice
    // A operand (Q): lane (row t = 16tt + j, quad q4) holds Q[t][k0 + 4*q4 + s], s = 0..3
    // B operand (V): lane (col n = 16tile + j, quad q4) holds V[k0 + 4*q4 + s][n]
    const int q_voff0 = (j * d + 4 * q4) * 4, q_voff1 = ((16 + j) * d + 4 * q4) * 4;
    const int v_voff = (4 * q4 * N + j) * 4;
    auto load_q = [&](int kb, f32x4(&dst)[2]) {
      dst[0] = buf_load4(rs_q, q_voff0, (ks + 16 * kb) * 4);
      dst[1] = buf_load4(rs_q, q_voff1, (ks + 16 * kb) * 4);
    };
    auto load_v = [&](int u, float(&dst)[NT]) {
      const int soff = (ks + 16 * (u >> 2) + (u & 3)) * N * 4;
#pragma unroll
      for (int t = 0; t < NT; ++t) dst[t] = buf_load1(rs_v, v_voff + 64 * t, soff);   // cols >= N: finite junk,
                                                                                       // zeroed when C is finalised
    };
    for (int sl = 0; sl < nsl; ++sl) {
    ks = (sl * NW + w) * 128;
    load_q(0, qa[0]);
#pragma unroll
    for (int u = 0; u < RING - 1; ++u) load_v(u, vb[u]);
#pragma unroll
    for (int u = 0; u < 32; ++u) {                   // 32 k-steps of 4 = one 128-channel slice
      if (u + RING - 1 < 32) load_v(u + RING - 1, vb[(u + RING - 1) % RING]);
      if ((u & 3) == 1 && (u >> 2) + 1 < 8) load_q((u >> 2) + 1, qa[((u >> 2) + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);             // keep the prefetch ahead of this step's MFMAs
      const int s = u & 3, qb = (u >> 2) & 1;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        acc[0][t] = mfma16(qa[qb][0][s], vb[u % RING][t], acc[0][t]);
        acc[1][t] = mfma16(qa[qb][1][s], vb[u % RING][t], acc[1][t]);
      }
    }
    }
    }
    CA_STAMP(1);
    load_slice_operands(w * 128);
    // cross-wave sum in a fixed tree order through LDS; C/D layout: col = j, row = 4*q4 + r
    auto put = [&](float* slot) {
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            slot[(16 * tt + 4 * q4 + r) * LD + 16 * t + j] = acc[tt][t][r];
          }
    };
    auto add = [&](const float* slot) {
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            acc[tt][t][r] += slot[(16 * tt + 4 * q4 + r) * LD + 16 * t + j];
          }
    };
#pragma unroll
    for (int stride = 1; stride < NW / 2; stride <<= 1) {
      const int m = 2 * stride - 1;
      if (stride > 1) __syncthreads();
      if ((w & m) == stride) put(slots + (w / (2 * stride)) * kSlotRows * LD);
      __syncthreads();
      if ((w & m) == 0) add(slots + (w / (2 * stride)) * kSlotRows * LD);
    }
    if (NW > 2) __syncthreads();
    if (w == NW / 2) put(slots);
    if (w == 0) put(slots + kSlotRows * LD);
    __syncthreads();
    // C = tanh(sum) by all threads; rows >= T are tanh(0) = 0 (their Q rows read as 0)
    float* Cg = a.C + pair * (size_t)T * N;
    constexpr int RSTEP = NW * 64 / 16;              // rows covered per sweep: 16 lanes per row
    for (int row = tid >> 4; row < kTRows; row += RSTEP) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int col = 16 * t + (tid & 15);
        float c = tanh_fast(slots[row * LD + col] + slots[kSlotRows * LD + row * LD + col]);
        c = (col < N) ? c : 0.f;                     // padded columns carry junk from phase 1
        Cbuf[row * LD + col] = c;
        if (row < T && col < N) Cg[(size_t)row * N + col] = c;
      }
    }
    __syncthreads();
    CA_STAMP(2);
    CA_STAMP_CYC(6);
  }

  // ------------------------------------------------------------------ phase 2: H_v scores, H_q
  float sqacc[2][4];                                 // s_q partials of this wave, summed over its slices
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int r = 0; r < 4; ++r) sqacc[tt][r] = 0.f;
  for (int sl = 0; sl < nsl; ++sl) {
  const int dsl = (sl * NW + w) * 128;
  if (sl > 0) load_slice_operands(dsl);
  f32x4 accq[2][8];
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int c = 0; c < 8; ++c) accq[tt][c] = f32x4{0.f, 0.f, 0.f, 0.f};

  // Tile loop, software pipelined over quarter tiles (2 channel tiles = 8 VGPRs each, ring of 4):
  // the MFMAs of quarter u are interleaved with the tanh / score VALU work of quarter u-1, and the
  // loads of quarter u+2 are issued two steps ahead (rows beyond N read 0 through the buffer rule).
  const int ntiles = (N + 15) >> 4;
  {
    f32x4 ring[4][2];
    f32x4 ca[2];
    float ct[kTS];
    float sv[4] = {0.f, 0.f, 0.f, 0.f};
    auto load_q = [&](int tile, int qc, f32x4(&dst)[2]) {        // pv[c][r] = P_v[16 tile + 4 q4 + r][dsl + 16(2qc + c) + j]
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int voff = ((4 * q4 + r) * d + j) * 4;
#pragma unroll
        for (int c = 0; c < 2; ++c) dst[c][r] = buf_load1(rs_pv, voff + 64 * (2 * qc + c), (16 * tile * d + dsl) * 4);
      }
    };
    auto load_a = [&](int tile) {                                // MFMA A operands of this location tile
      const int nb = 16 * tile;
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        const int t = min(16 * tt + j, kTRows - 1);
        ca[tt] = *reinterpret_cast<const f32x4*>(&Cbuf[t * LD + nb + 4 * q4]);
      }
#pragma unroll
      for (int s = 0; s < kTS; ++s) ct[s] = Cbuf[(4 * s + q4) * LD + nb + j];
    };
    // MFMAs of one quarter: accq += C . P_v (B = the tile), then the tile accumulates C^T P_q
    auto mfma_q = [&](f32x4(&pv)[2], const int qc) {
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int c = 0; c < 2; ++c) accq[tt][2 * qc + c] = mfma16(ca[tt][s], pv[c][s], accq[tt][2 * qc + c]);
#pragma unroll
      for (int s = 0; s < kTS; ++s)
#pragma unroll
        for (int c = 0; c < 2; ++c) pv[c] = mfma16(ct[s], pq[s][2 * qc + c], pv[c]);
    };
    auto valu_q = [&](const f32x4(&pv)[2], const int qc) {       // s_v[n] += tanh(H_v[n][d]) w_v[d]
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) sv[r] = fmaf(tanh_fast(pv[c][r]), wvr[2 * qc + c], sv[r]);
    };
    auto flush_sv = [&](int tile) {                              // 16-lane row sums -> per-wave score partials
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float t = row16_sum(sv[r]);
        float* dst = &svpart[w * NPAD + 16 * tile + 4 * q4 + r];
        if (j == 0) *dst = (sl > 0) ? *dst + t : t;             // accumulate over this wave's channel slices
        sv[r] = 0.f;
      }
    };
    // 30 MFMAs of one quarter interleaved with the ~50 VALU ops of the previous one
#define COATTN_INTERLEAVE()                                        \
  _Pragma("unroll") for (int g_ = 0; g_ < 28; ++g_) {              \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);             \
    __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);             \
  }                                                                \
  __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
    load_q(0, 0, ring[0]);
    load_q(0, 1, ring[1]);
    for (int tile = 0; tile < ntiles; ++tile) {
      // step 0: quarter 0 of this tile; VALU of quarter 3 of the previous tile
      load_q(tile, 2, ring[2]);
      load_a(tile);
      __builtin_amdgcn_sched_barrier(0);
      mfma_q(ring[0], 0);
      if (tile > 0) valu_q(ring[3], 3);
      COATTN_INTERLEAVE();
      __builtin_amdgcn_sched_barrier(0);
      if (tile > 0) flush_sv(tile - 1);
      // step 1
      load_q(tile, 3, ring[3]);
      __builtin_amdgcn_sched_barrier(0);
      mfma_q(ring[1], 1);
      valu_q(ring[0], 0);
      COATTN_INTERLEAVE();
      __builtin_amdgcn_sched_barrier(0);
      // step 2 (next tile's quarter 0 starts loading; beyond N it reads zeros)
      load_q(tile + 1, 0, ring[0]);
      __builtin_amdgcn_sched_barrier(0);
      mfma_q(ring[2], 2);
      valu_q(ring[1], 1);
      COATTN_INTERLEAVE();
      __builtin_amdgcn_sched_barrier(0);
      // step 3
      load_q(tile + 1, 1, ring[1]);
      __builtin_amdgcn_sched_barrier(0);
      mfma_q(ring[3], 3);
      valu_q(ring[2], 2);
      COATTN_INTERLEAVE();
      __builtin_amdgcn_sched_barrier(0);
    }
    if (ntiles > 0) {
      valu_q(ring[3], 3);
      flush_sv(ntiles - 1);
    }
#undef COATTN_INTERLEAVE
  }
  CA_STAMP_CYC(7);                                   // shader cycles around the tile loop (with stamp 6)
  CA_STAMP(3);

  // ------------------------------------------------------------------ phase 3
  // H_q epilogue: hq = tanh(P_q + acc); saved for backward; s_q partials.  Branch-free: rows t >= T
  // fall outside the per-sample buffers (loads give 0, stores are dropped), all loads issued first.
  {
    const __amdgpu_buffer_rsrc_t rs_hq = make_rsrc(a.Hq + pair * (size_t)T * d, (unsigned)T * d * 4u);
    float wqr[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) wqr[c] = a.wq[dsl + 16 * c + j];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      float pqv[4][8];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 8; ++c)
          pqv[r][c] = buf_load1(rs_pq, ((16 * tt + 4 * q4 + r) * d + j) * 4 + 64 * c, dsl * 4);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const float h = tanh_fast(accq[tt][c][r] + pqv[r][c]);
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, h), rs_hq,
                                                ((16 * tt + 4 * q4 + r) * d + j) * 4 + 64 * c, dsl * 4, 0);
          acc = fmaf(h, wqr[c], acc);
        }
        sqacc[tt][r] += row16_sum(acc);
      }
    }
  }
  }   // channel slices
  CA_STAMP(4);
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (j == 0) sqpart[w * 32 + 16 * tt + 4 * q4 + r] = sqacc[tt][r];
  __syncthreads();
  if (w == 0) {
    // a_v = softmax_n(s_v + c_v): N <= 16*NT <= 256 -> <= 4 values per lane
    constexpr int PER = (NPAD + 63) / 64;
    float sc[PER];
    float m = -INFINITY;
    const float cv = a.cv[0];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int n = lane + 64 * k;
      float s = -INFINITY;
      if (n < N) {
        s = cv;
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) s += svpart[ww * NPAD + n];
      }
      sc[k] = s;
      m = fmaxf(m, s);
    }
    m = wave_max(m);
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      sc[k] = (lane + 64 * k < N) ? expf(sc[k] - m) : 0.f;
      sum += sc[k];
    }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    float* avg = a.av + pair * (size_t)N;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int n = lane + 64 * k;
      if (n < N) avg[n] = sc[k] * inv;
    }
    // a_q = softmax_t(s_q + c_q), un-masked over all T positions (model.py:388)
    float s = -INFINITY;
    if (lane < T) {
      s = a.cq[0];
#pragma unroll
      for (int ww = 0; ww < NW; ++ww) s += sqpart[ww * 32 + lane];
    }
    const float mq = wave_max(s);
    const float e = (lane < T) ? expf(s - mq) : 0.f;
    const float se = wave_sum(e);
    const float aqv = e / se;
    if (lane < 32) aqs[lane] = aqv;                  // zeros beyond T
    if (lane < T) a.aq[pair * (size_t)T + lane] = aqv;
  }
  __syncthreads();
  // q = sum_t a_q[t] Q[t][:]   (model.py:392): all kTRows row loads in flight at once (rows >= T read 0)
  for (int dd = tid; dd < d; dd += NW * 64) {
    float x[kTRows];
#pragma unroll
    for (int t = 0; t < kTRows; ++t) x[t] = buf_load1(rs_q, (t * d + dd) * 4, 0);
    float acc = 0.f;
#pragma unroll
    for (int t = 0; t < kTRows; ++t) acc = fmaf(aqs[t], x[t], acc);
    a.q_out[pair * (size_t)d + dd] = acc;
  }
  CA_STAMP(5);
}

// v_l[b][k] = sum_n a_v[l][b][n] V[b][k][n]   (model.py:391), all L levels in one pass over V.
// grid (d/64, B); 256 threads: 16 lanes per channel row, 16 rows per sweep.
template <int NT>
__global__ __launch_bounds__(256) void attend_v_kernel(const float* V, long v_sB, const float* av, float* v_out, int B,
                                                       int N, int d, int L) {
  const int b = blockIdx.y, k0 = blockIdx.x * 64;
  const int tid = threadIdx.x, j = tid & 15, rsub = tid >> 4;      // rsub 0..15
  float aw[3][NT];
#pragma unroll
  for (int l = 0; l < 3; ++l)
#pragma unroll
    for (int m = 0; m < NT; ++m) {
      const int n = j + 16 * m;
      aw[l][m] = (n < N && l < L) ? av[((size_t)l * B + b) * N + n] : 0.f;
    }
  const float* Vb = V + (size_t)b * v_sB;
  for (int it = 0; it < 4; ++it) {
    const int k = k0 + 16 * it + rsub;
    const float* vr = Vb + (size_t)k * N;
    float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < NT; ++m) {
      const int n = j + 16 * m;
      const float x = (n < N) ? vr[n] : 0.f;
#pragma unroll
      for (int l = 0; l < 3; ++l) acc[l] = fmaf(x, aw[l][m], acc[l]);
    }
#pragma unroll
    for (int l = 0; l < 3; ++l) {
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) acc[l] += __shfl_xor(acc[l], o, 64);
      if (j == 0 && l < L) v_out[((size_t)l * B + b) * d + k] = acc[l];
    }
  }
}

template <int NT, int NW, int X3>
int launch_fwd(const FwdArgs& a, hipStream_t s) {
  constexpr int LD = 16 * NT + 4;
  constexpr int NSLOT = (NW / 2 > 2) ? NW / 2 : 2;
  const size_t lds = (size_t)(NSLOT * kSlotRows + kTRows) * LD * sizeof(float);
  static DeviceOnce once;                            // the attribute is per device
  CA_TRY(once.run([&] {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(coattn_attn_fwd_kernel<NT, NW, X3>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }, "coattn_attn_fwd"));
  const int groups = (a.B + 7) / 8;
  dim3 grid(groups * a.L * 8), block(NW * 64);
  hipLaunchKernelGGL((coattn_attn_fwd_kernel<NT, NW, X3>), grid, block, lds, s, a);
  CA_CHECK_LAUNCH("coattn_attn_fwd");
  return 0;
}

}  // namespace

int fused_supported(int B, int N, int T, int d, int L) {
  (void)B;
  if (d <= 0 || d % 256 != 0 || d > 4096) return 0;   // NW = 4 waves when d % 512 == 0, else 2; slices of 128 channels
  if (T > kTRows || N > 208 || L > 3) return 0;
  return 1;
}

int fused_attention_forward(int B, int N, int T, int d, int L, const float* V, const VLayout& vl,
                            const float* const* Q, const coattn_params* p, float* v_out, float* q_out, float* saved,
                            float* ws, hipStream_t s) {
  CA_CHECK_ARG(fused_supported(B, N, T, d, L), "fused forward: unsupported shape");
  const bool lm = v_is_lm(vl, N, d);
  CA_CHECK_ARG(lm || v_is_cm(vl, N, d), "fused forward: image features must be channel-major [B,d,N] or location-major [B,N,d]");
  const SavedOff so = saved_off(B, N, T, d, L);
  FwdArgs a;
  a.V = V; a.v_sB = vl.sB; a.lm = lm ? 1 : 0;
  for (int l = 0; l < 8; ++l) a.Q[l] = l < L ? Q[l] : nullptr;
  a.Pv = saved + so.Pv; a.Pq = saved + so.Pq;
  a.wv = (const float*)p->w_v; a.cv = (const float*)p->c_v; a.wq = (const float*)p->w_q; a.cq = (const float*)p->c_q;
  a.C = saved + so.C; a.av = saved + so.av; a.aq = saved + so.aq; a.Hq = saved + so.Hq;
  a.q_out = q_out;
  a.stamps = COATTN_STAMPS ? reinterpret_cast<unsigned long long*>(ws) : nullptr;
  a.B = B; a.N = N; a.T = T; a.d = d; a.L = L;
  const bool small_n = N <= 64;
  // COATTN_FWD_K (developer switch): 32 = the bf16-split kernel on the 32x32x16 MFMA (coattn_fwd32.hip) for both
  // layouts, 16 = the 16x16 kernel of this file (channel-major only).  Default: 32 for location-major features,
  // 16 for channel-major ones.
  static const int kenv = [] { const char* e = getenv("COATTN_FWD_K"); return e ? atoi(e) : 0; }();
  // COATTN_FWD_X3=0: phase 1 of the 16x16 kernel on the f32 MFMA (developer switch for ablations; default 1: bf16 split)
  static const int x3 = [] { const char* e = getenv("COATTN_FWD_X3"); return (e && e[0] == '0') ? 0 : 1; }();
  if (lm || kenv == 32) {
    CA_TRY(fused32_forward(a, s));
  } else if (d % 512 == 0) {
    if (x3) CA_TRY(small_n ? (launch_fwd<4, 4, 1>(a, s)) : (launch_fwd<13, 4, 1>(a, s)));
    else CA_TRY(small_n ? (launch_fwd<4, 4, 0>(a, s)) : (launch_fwd<13, 4, 0>(a, s)));
  } else {
    if (x3) CA_TRY(small_n ? (launch_fwd<4, 2, 1>(a, s)) : (launch_fwd<13, 2, 1>(a, s)));
    else CA_TRY(small_n ? (launch_fwd<4, 2, 0>(a, s)) : (launch_fwd<13, 2, 0>(a, s)));
  }
  if (lm) return launch_attend_v_lm(V, vl.sB, a.av, v_out, B, N, d, L, s);
  dim3 grid(d / 64, B);
  if (small_n)
    hipLaunchKernelGGL(attend_v_kernel<4>, grid, dim3(256), 0, s, V, vl.sB, a.av, v_out, B, N, d, L);
  else
    hipLaunchKernelGGL(attend_v_kernel<13>, grid, dim3(256), 0, s, V, vl.sB, a.av, v_out, B, N, d, L);
  CA_CHECK_LAUNCH("attend_v");
  return 0;
}
