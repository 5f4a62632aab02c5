// Tile-pipelined fused co-attention forward for gfx950 (fp32, exact-f32 MFMA 16x16x4).
//
// Same math and the same outputs as coattn_fused.hip's forward kernel (model.py:377-392 after the
// projections), different schedule.  That kernel runs "all of A = Q V^T, then all of the P_v tile loop":
// every workgroup of the launch streams V at the same time (HBM-bound, MFMA idle) and then grinds
// through the MFMA-bound tile loop at the same time (HBM idle).  Here the affinity slice of location
// tile i+1 is computed INSIDE the tile loop, one tile ahead of its use:
//
//   per 16-location tile i (one workgroup per (sample, level), wave w owns channels [128w, 128w+128)):
//     a) partial A[:, tile i+1] over the wave's 128 channels: Q operands from an LDS image of Q,
//        V operands streamed from HBM into a 16-register ring that is refilled as it is consumed;
//     b) the P_v tile i is the B operand of H_q += C . P_v and then the accumulator of
//        H_v = tanh(P_v + C^T P_q), folded into score partials (as in coattn_fused.hip);
//     c) the four partial A tiles meet in LDS (double-buffered, one barrier per tile); every wave sums
//        them in the fixed order (w0 + w1) + (w2 + w3), applies tanh and keeps C[:, tile i+1] in
//        registers in the two operand layouts the next iteration needs.  C never lives in LDS.
//
// V, P_v streaming and the MFMA work are spread evenly over the kernel's lifetime, and the two
// workgroups of a CU no longer meet in the same phase.  Supported: d = 128 NW (NW = 4 or 2 waves, one
// 128-channel slice per wave), T <= 26 (LDS: Q image + partial slots, two workgroups per CU), N <= 208.
#include "fused.h"

#include <stdlib.h>

#ifndef COATTN_STAMPS
#define COATTN_STAMPS 0
#endif
#if COATTN_STAMPS
#define CA_STAMP(k)                                                                                   \
  do {                                                                                                \
    if (threadIdx.x == 0) a.stamps[(size_t)blockIdx.x * 64 + (k)] = __builtin_amdgcn_s_memrealtime();  \
  } while (0)
#else
#define CA_STAMP(k)
#endif

namespace {

constexpr int kSLD = 20;      // row stride of a partial-A slot (conflict-free b128 row reads, b32 column reads)
constexpr int kSRows = 28;    // rows of a slot (rows >= T are zero: their Q rows are)
constexpr int kMaxT2 = 26;
constexpr int kDead = 0x7ffffff0;   // buffer offset beyond every resource: loads return 0, no memory traffic

// the lane id, recomputed where it is used: keeps the address registers of the once-per-tile LDS exchange
// from staying live (and getting spilled) across the MFMA steps -- a spill reload is a VMEM load whose
// s_waitcnt vmcnt(0) would drain the whole prefetch ring
__device__ __forceinline__ int fresh_lane() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}

template <int NW>
__global__ __launch_bounds__(NW * 64, 2) void coattn_attn_fwd2_kernel(const FwdArgs a) {
  constexpr int d = 128 * NW;
  constexpr int QLD = d + 4;                         // row stride of the LDS image of Q
  extern __shared__ __attribute__((aligned(16))) float lds[];
  int b, l;
  if (!block_to_pair(blockIdx.x, a.B, a.L, b, l)) return;
  CA_STAMP(0);
  const int N = a.N, T = a.T;
  const int ntiles = (N + 15) >> 4, npad = ntiles * 16;
  float* Qs = lds;                                   // (T + 1) x QLD; row T is all zeros
  float* slots = Qs + (T + 1) * QLD;                 // 2 x NW x kSRows x kSLD
  float* svpart = slots + 2 * NW * kSRows * kSLD;    // NW x npad
  float* sqpart = svpart + NW * npad;                // NW x 32
  float* aqs = sqpart + NW * 32;                     // 32
  float* wvs = aqs + 32;                             // d: w_v (read per quarter step; keeps 8 registers free)

  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, q4 = lane >> 4;           // MFMA column / k-quad (also C/D row quad)
  const float* Qp = a.Q[l] + (size_t)b * T * d;
  const float* Vp = a.V + (size_t)b * d * N;
  const float* Pvp = a.Pv + (size_t)b * N * d;
  const float* Pqp = a.Pq + ((size_t)l * a.B + b) * T * d;
  const size_t pair = (size_t)l * a.B + b;
  // buffer resources over exactly this sample's tensors: rows beyond T / N read as 0, stores are dropped
  const __amdgpu_buffer_rsrc_t rs_q = make_rsrc(Qp, (unsigned)T * d * 4u);
  const __amdgpu_buffer_rsrc_t rs_v = make_rsrc(Vp, (unsigned)d * N * 4u);
  const __amdgpu_buffer_rsrc_t rs_pv = make_rsrc(Pvp, (unsigned)N * d * 4u);
  const __amdgpu_buffer_rsrc_t rs_pq = make_rsrc(Pqp, (unsigned)T * d * 4u);
  const __amdgpu_buffer_rsrc_t rs_c = make_rsrc(a.C + pair * (size_t)T * N, (unsigned)T * N * 4u);
  const int dsl = w * 128;                           // this wave's channel slice

  // ---- prologue: Q -> LDS (rows T .. kSRows-1 of the sweep read 0 through the buffer rule; row T is kept
  //      as the zero row), V operands of tile 0, P_v quarters of tile 0, the slice's P_q / w_v operands
  float vreg[16];                                    // B operands of A = Q V^T, half a location tile at a time:
                                                     // k-step u lives in vreg[u & 15] = V[dsl + 16(u>>2) + (u&3) + 4 q4][16 tile + j]
  const int v_voff = (4 * q4 * N + j) * 4;
  // cols >= N: finite junk, zeroed when C is finalised; voff = kDead: out of range, returns 0 without traffic
  auto load_v = [&](int u, int tile, int voff) {
    vreg[u & 15] = buf_load1(rs_v, voff, ((dsl + 16 * (u >> 2) + (u & 3)) * N + 16 * tile) * 4);
  };
#pragma unroll
  for (int u = 0; u < 16; ++u) load_v(u, 0, v_voff);
  {
    constexpr int SWEEP = (kMaxT2 + 2) * (d / 4) / (NW * 64);          // float4 per thread: 14
    f32x4 x[SWEEP];
#pragma unroll
    for (int i = 0; i < SWEEP; ++i) x[i] = buf_load4(rs_q, (tid + i * NW * 64) * 16, 0);
#pragma unroll
    for (int i = 0; i < SWEEP; ++i) {
      const int idx = tid + i * NW * 64, row = idx / (d / 4), c4 = idx % (d / 4);
      if (row <= T) *reinterpret_cast<f32x4*>(&Qs[row * QLD + 4 * c4]) = x[i];
    }
  }
  float pq[kTS][8];                                  // B operand P_q[t = 4s + q4][dsl + 16c + j]
#pragma unroll
  for (int s = 0; s < kTS; ++s)
#pragma unroll
    for (int c = 0; c < 8; ++c) pq[s][c] = buf_load1(rs_pq, ((4 * s + q4) * d + j) * 4 + 64 * c, dsl * 4);
  for (int i = tid; i < d; i += NW * 64) wvs[i] = a.wv[i];

  f32x4 ring[2][2];                                  // P_v quarter tiles (2 channel tiles each), double-buffered
  auto load_q = [&](int tile, int qc, f32x4(&dst)[2]) {   // pv[c][r] = P_v[16 tile + 4 q4 + r][dsl + 16(2qc + c) + j]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int voff = ((4 * q4 + r) * d + j) * 4;
#pragma unroll
      for (int c = 0; c < 2; ++c) dst[c][r] = buf_load1(rs_pv, voff + 64 * (2 * qc + c), (16 * tile * d + dsl) * 4);
    }
  };
  load_q(0, 0, ring[0]);
  __syncthreads();                                   // Q image complete
  CA_STAMP(1);

  // ---- partial affinity of one location tile over this wave's channels
  const int qrow0 = min(j, T) * QLD + dsl + 4 * q4, qrow1 = min(16 + j, T) * QLD + dsl + 4 * q4;
  f32x4 acc2[2];
  // k-steps [8g, 8g+8) of tile `cur`; each consumed register is refilled with the k-step 16 further on
  // (of `cur` for g < 2, of the tile after it for g >= 2; voff_next = kDead when there is none), so that
  // every load has two steps of the tile loop to land
  auto apart = [&](int g, int cur, int voff_cur, int voff_next) {
#pragma unroll
    for (int kb = 2 * g; kb < 2 * g + 2; ++kb) {
      const f32x4 q0 = *reinterpret_cast<const f32x4*>(&Qs[qrow0 + 16 * kb]);
      const f32x4 q1 = *reinterpret_cast<const f32x4*>(&Qs[qrow1 + 16 * kb]);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        acc2[0] = mfma16(q0[s], vreg[(4 * kb + s) & 15], acc2[0]);
        acc2[1] = mfma16(q1[s], vreg[(4 * kb + s) & 15], acc2[1]);
      }
    }
#pragma unroll
    for (int u = 8 * g; u < 8 * g + 8; ++u) {
      if (g < 2) load_v(u + 16, cur, voff_cur);
      else load_v(u - 16, cur + 1, voff_next);
    }
  };
  auto put_partial = [&](int buf) {                  // C/D layout: col = j, row = 4 q4 + r
    const int ln = fresh_lane(), j = ln & 15, q4 = ln >> 4;
    float* slot = slots + (buf * NW + w) * kSRows * kSLD;
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * tt + 4 * q4 + r;
        if (row < kSRows) slot[row * kSLD + j] = acc2[tt][r];
      }
  };
  // C[:, tile] = tanh(sum of the partials) in the two MFMA A-operand layouts of the tile loop
  f32x4 ca[2];                                       // ca[tt][e] = C[16 tt + j][16 tile + 4 q4 + e]   (H_q += C . P_v)
  float ct[kTS];                                     // ct[s]     = C[4 s + q4][16 tile + j]           (H_v += C^T P_q)
  auto finish_c = [&](int buf, int tile) {
    const int ln = fresh_lane(), j = ln & 15, q4 = ln >> 4;
    const float* sb = slots + buf * NW * kSRows * kSLD;
    const int nb = 16 * tile;
    // branch-free, in three batches (tt = 0, tt = 1, the transposed layout) of independent chains: all LDS
    // reads of a batch, then its sums in the fixed order (w0 + w1) + (w2 + w3), then its tanh; padded
    // columns (>= N) carry finite junk from the V rows and are cleared by a 0/1 factor
    float ma[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) ma[e] = (nb + 4 * q4 + e < N) ? 1.f : 0.f;
    const float mt = (nb + j < N) ? 1.f : 0.f;
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      const int row = min(16 * tt + j, kSRows - 1);  // rows >= kSRows only feed accumulator rows that are never stored
      f32x4 xa[NW];
#pragma unroll
      for (int ww = 0; ww < NW; ++ww)
        xa[ww] = *reinterpret_cast<const f32x4*>(&sb[ww * kSRows * kSLD + row * kSLD + 4 * q4]);
      f32x4 x = xa[0] + xa[1];
      if (NW > 2) x += xa[2] + xa[3];
#pragma unroll
      for (int e = 0; e < 4; ++e) ca[tt][e] = tanh_fast(x[e]) * ma[e];
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {                    // s = 0..3, then 4..6
      float xt[4][NW];
#pragma unroll
      for (int s = 4 * h; s < (h ? kTS : 4); ++s)
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) xt[s - 4 * h][ww] = sb[ww * kSRows * kSLD + (4 * s + q4) * kSLD + j];
#pragma unroll
      for (int s = 4 * h; s < (h ? kTS : 4); ++s) {
        float y = xt[s - 4 * h][0] + xt[s - 4 * h][1];
        if (NW > 2) y += xt[s - 4 * h][2] + xt[s - 4 * h][3];
        ct[s] = tanh_fast(y) * mt;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // C saved for backward: waves 0 and 1 store the row halves tt = 0, 1 (rows >= T fall outside the buffer,
    // columns >= N and the other waves are sent there)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int col = nb + 4 * q4 + e;
      const float cvv = (w == 0) ? ca[0][e] : ca[1][e];
      const int off = (col < N && w < 2) ? ((16 * w + j) * N + col) * 4 : kDead;
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, cvv), rs_c, off, 0, 0);
    }
  };

  // tile 0 up front
  acc2[0] = acc2[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int g = 0; g < 4; ++g) apart(g, 0, v_voff, ntiles > 1 ? v_voff : kDead);
  put_partial(0);
  __syncthreads();
  finish_c(0, 0);
  CA_STAMP(2);

  // ---- tile loop
  float sqacc[2][4];
  f32x4 accq[2][8];
#pragma unroll
  for (int tt = 0; tt < 2; ++tt) {
#pragma unroll
    for (int r = 0; r < 4; ++r) sqacc[tt][r] = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) accq[tt][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  {
    float sv[4] = {0.f, 0.f, 0.f, 0.f};
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
      for (int c = 0; c < 2; ++c) {
        const float wvc = wvs[dsl + 16 * (2 * qc + c) + j];
#pragma unroll
        for (int r = 0; r < 4; ++r) sv[r] = fmaf(tanh_fast(pv[c][r]), wvc, sv[r]);
      }
    };
    auto flush_sv = [&](int tile) {                              // 16-lane row sums -> per-wave score partials
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float t = row16_sum(sv[r]);
        if (j == 0) svpart[w * npad + 16 * tile + 4 * q4 + r] = t;
        sv[r] = 0.f;
      }
    };
    // 46 MFMAs of one step (30 of the P_v quarter, 16 of the next tile's affinity) interleaved with the
    // ~50 VALU ops of the previous quarter
#define COATTN_INTERLEAVE()                                        \
  __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);               \
  __builtin_amdgcn_sched_group_barrier(0x008, 30, 0);              \
  _Pragma("unroll") for (int g_ = 0; g_ < 16; ++g_) {              \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);             \
    __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);             \
  }                                                                \
  __builtin_amdgcn_sched_group_barrier(0x020, 8, 0);
    for (int tile = 0; tile < ntiles; ++tile) {
      // this iteration computes tile+1's affinity (junk from a dead ring when there is no such tile)
      const int cvoff = (tile + 1 < ntiles) ? v_voff : kDead, rvoff = (tile + 2 < ntiles) ? v_voff : kDead;
      acc2[0] = acc2[1] = f32x4{0.f, 0.f, 0.f, 0.f};
      CA_STAMP(8 + 4 * tile);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        // quarter k: its P_v registers were loaded one step ago; the next quarter starts loading now
        if (k < 3) load_q(tile, k + 1, ring[(k + 1) & 1]);
        else load_q(tile + 1, 0, ring[0]);           // beyond N it reads zeros
        __builtin_amdgcn_sched_barrier(0);
        mfma_q(ring[k & 1], k);
        apart(k, tile + 1, cvoff, rvoff);
        valu_q(ring[k & 1], k);
        COATTN_INTERLEAVE();
        __builtin_amdgcn_sched_barrier(0);
      }
      flush_sv(tile);
      CA_STAMP(9 + 4 * tile);
      if (tile + 1 < ntiles) put_partial((tile + 1) & 1);
      __syncthreads();
      CA_STAMP(10 + 4 * tile);
      if (tile + 1 < ntiles) finish_c((tile + 1) & 1, tile + 1);
      CA_STAMP(11 + 4 * tile);
    }

#undef COATTN_INTERLEAVE
  }
  CA_STAMP(3);

  // ---- H_q epilogue: hq = tanh(P_q + acc); saved for backward; s_q partials.  Branch-free: rows t >= T
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
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (j == 0) sqpart[w * 32 + 16 * tt + 4 * q4 + r] = sqacc[tt][r];
  __syncthreads();
  CA_STAMP(4);
  if (w == 0) {
    // a_v = softmax_n(s_v + c_v): N <= 208 -> <= 4 values per lane
    constexpr int PER = 4;
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
        for (int ww = 0; ww < NW; ++ww) s += svpart[ww * npad + n];
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
  // q = sum_t a_q[t] Q[t][:]   (model.py:392) from the LDS image of Q
  for (int dd = tid; dd < d; dd += NW * 64) {
    float acc = 0.f;
    for (int t = 0; t < T; ++t) acc = fmaf(aqs[t], Qs[t * QLD + dd], acc);
    a.q_out[pair * (size_t)d + dd] = acc;
  }
  CA_STAMP(5);
}

size_t lds_bytes(int NW, int N, int T) {
  const int npad = ((N + 15) / 16) * 16;
  return (size_t)((T + 1) * (128 * NW + 4) + 2 * NW * kSRows * kSLD + NW * npad + NW * 32 + 32 + 128 * NW) *
         sizeof(float);
}

template <int NW>
int launch(const FwdArgs& a, hipStream_t s) {
  const size_t lds = lds_bytes(NW, a.N, a.T);
  static size_t attr_set = 0;
  if (lds > attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(coattn_attn_fwd2_kernel<NW>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = lds;
  }
  const int groups = (a.B + 7) / 8;
  dim3 grid(groups * a.L * 8), block(NW * 64);
  hipLaunchKernelGGL((coattn_attn_fwd2_kernel<NW>), grid, block, lds, s, a);
  CA_CHECK_LAUNCH("coattn_attn_fwd2");
  return 0;
}

}  // namespace

int fused2_supported(int B, int N, int T, int d, int L) {
  (void)B;
  if (d != 512 && d != 256) return 0;
  if (T > kMaxT2 || T < 1 || N > 208 || N < 1 || L > 3) return 0;
  return lds_bytes(d / 128, N, T) <= 80 * 1024;      // two workgroups per CU
}

int fused2_launch(const FwdArgs& a, hipStream_t s) {
  CA_CHECK_ARG(fused2_supported(a.B, a.N, a.T, a.d, a.L), "tile-pipelined forward: unsupported shape");
  return a.d == 512 ? launch<4>(a, s) : launch<2>(a, s);
}
