// Fused backward of the parallel co-attention (hand-derived; SURVEY.md section 8, checked
// against autograd of the reference by the oracle).  fp32 results; bwd_dq_kernel (channel-major V) uses the exact-f32
// MFMA 16x16x4, the big kernels live in coattn_bwd32.hip on the bf16 MFMA.
//
// H_v [N,d] is never stored: both big kernels recompute it from P_v, P_q and C (saved).
//
//   bwd_pre_kernel  (per sample, all levels)  one pass over V: da_v = V gv partials (the image side's softmax backward
//                   ds_v is finished in the prologues of the two big kernels); da_q = Q gq -> ds_q, dc_q partials.
//                   (dZ_q = ds_q (x) w_q (.) (1 - H_q^2) is not stored: the two big kernels form it from the saved H_q
//                   where they used to read it -- the same bytes for them, one read of H_q and one write of dZ_q
//                   less for this kernel -- and bwd_nat32_kernel sums the dw_q partial.)
//   bwd_dc32_kernel (coattn_bwd32.hip; per sample x level, orientation [d][n], bf16 MFMA with the exact 3-way
//                   split): H_v^T tile = P_v^T + P_q^T C -> dZ_v^T, and dC += P_q dZ_v^T + dZ_q P_v^T with the
//                   dZ_v^T / P_v^T fragments as MFMA B operands (contraction over d); cross-wave sum through
//                   LDS, dA = dC (.) (1 - C^2).
//   bwd_nat32_kernel (coattn_bwd32.hip; per sample x level, orientation [n][d], the loop of forward phase 2, on
//                   the bf16 MFMA with the exact 3-way split): H_v tile -> dZ_v tile, which is the B operand of
//                   dP_q += C dZ_v (contraction over N) and then the accumulator of dP_v = dZ_v + C^T dZ_q
//                   (stored as it lies); dw_v, db_v, db_q partials.
//   then MFMA GEMMs: dQ = a_q (x) gq + dA V^T + dP_q W_q;  dV = sum_l (a_v (x) gv + Q^T dA) +
//   (sum_l dP_v) W_v (skipped when the image features need no gradient);  dW_v, dW_q, biases.
#include "fused.h"
#include <stdlib.h>



namespace {

// partial da_v over 64 channel rows: part[b][kc][l][n] = sum_{k in chunk kc} V[b][k][n] gv[l][b][k].
// grid (d/64, B); thread <-> location n (rows of V are contiguous in n: fully coalesced).
__device__ __forceinline__ void dav_cm_block(const float* V, long v_sB, const float* gv, float* part, int B, int N, int d,
                                             int L, int kc, int b, int nkc, float (*g)[64]) {
  const int n = threadIdx.x;
  if (threadIdx.x < 192) {
    const int l = threadIdx.x >> 6, k = threadIdx.x & 63;
    g[l][k] = (l < L) ? gv[((size_t)l * B + b) * d + kc * 64 + k] : 0.f;
  }
  __syncthreads();
  if (n >= N) return;
  const float* vp = V + (size_t)b * v_sB + (size_t)kc * 64 * N + n;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll 16
  for (int k = 0; k < 64; ++k) {
    const float x = vp[(size_t)k * N];
    a0 = fmaf(x, g[0][k], a0);
    a1 = fmaf(x, g[1][k], a1);
    a2 = fmaf(x, g[2][k], a2);
  }
  float* o = part + (((size_t)b * nkc + kc) * 3) * N + n;
  o[0] = a0; o[N] = a1; o[2 * (size_t)N] = a2;
}

// location-major V [N][d]: da_v[l][n] = V[n][:] . gv[l][:], whole rows -> part[b][0][l][n] (one "chunk").
// grid (ceil(N / 16), B); a wave takes 4 rows, lanes along the channels (float4), three wave sums per row.
__device__ __forceinline__ void dav_lm_block(const float* __restrict__ V, long v_sB, const float* __restrict__ gv, float* __restrict__ part,
                                             int B, int N, int d, int L, int bx, int b) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const float* Vb = V + (size_t)b * v_sB;
  const int n0 = bx * 16 + 4 * w;
  if (n0 >= N) return;
  // the wave's four rows are requested together (rows past N: row N - 1 again, its sums dropped), the level vectors are
  // read once per 256-channel sweep, the twelve wave sums run as independent chains
  float a[4][3] = {};
  const float* rowp[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) rowp[i] = Vb + (size_t)min(n0 + i, N - 1) * d;
  for (int k = 4 * lane; k < d; k += 256) {
    f32x4 x[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) x[i] = *reinterpret_cast<const f32x4*>(rowp[i] + k);
    const f32x4 g0 = *reinterpret_cast<const f32x4*>(gv + ((size_t)0 * B + b) * d + k);
    const f32x4 g1 = L > 1 ? *reinterpret_cast<const f32x4*>(gv + ((size_t)1 * B + b) * d + k) : f32x4{0.f, 0.f, 0.f, 0.f};
    const f32x4 g2 = L > 2 ? *reinterpret_cast<const f32x4*>(gv + ((size_t)2 * B + b) * d + k) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        a[i][0] = fmaf(x[i][e], g0[e], a[i][0]);
        a[i][1] = fmaf(x[i][e], g1[e], a[i][1]);
        a[i][2] = fmaf(x[i][e], g2[e], a[i][2]);
      }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int l = 0; l < 3; ++l) a[i][l] = wave_sum(a[i][l]);
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (n0 + i >= N) break;
      float* o = part + (size_t)b * 3 * N + n0 + i;
      o[0] = a[i][0]; o[N] = a[i][1]; o[2 * (size_t)N] = a[i][2];
    }
  }
}

struct PreArgs {
  const float* dav_part;                 // [B][nkc][3][N] partial da_v (bwd_dav_kernel: nkc = d/64; _lm: nkc = 1)
  int nkc;
  const float* Q[8];
  const float* gq;                       // [L][B][d]
  const float* av; const float* aq;      // saved
  // the da_v blocks ride along in the same launch (blocks past L*B): dav_gx blocks per sample
  const float* V; long v_sB; const float* gv; float* dav_out; int dav_lm, dav_gx;
  float* dsq;                            // [L*B][32] ds_q, zeros for t >= T
  float* dcs_part;                       // [2][L*B]
  int B, N, T, d, L;
  TnDyn dyn; TnDynPlan* plan_out;        // plan_out != NULL: ONE more workgroup (the launch's last) evaluates the split-K plan of the
                                         // weight gradients over the live question rows (fused.h tn_dyn_plan) and leaves it there
};

// Blocks [0, L*B): one workgroup (256 threads) per (sample, level), question side -- softmax backward of a_q
// (da_q = Q gq -> ds_q, dc_q partial).  Blocks past L*B: the da_v partials (one pass over V for all levels, independent
// of the question side: they fill the idle half of the chip).
// (Round 4, measured and not kept while this kernel also swept H_q into dZ_q: requesting a wave's rows together -- all 7 Q
//  rows of the da_q dot products, all 14 / 2 x 7 H_q rows of the sweep, the 4 V rows of a da_v wave -- costs registers the
//  da_v blocks of the same launch pay for with occupancy: 60 -> 102 / 128 VGPRs, 27.7 -> 32.6 / 37.3 us at N = 196,
//  22.7 -> 22.3 / 21.6 at N = 49.)
__global__ __launch_bounds__(256) void bwd_pre_kernel(const PreArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  if (a.plan_out && (int)blockIdx.x == a.L * a.B + a.dav_gx * a.B) {
    if (threadIdx.x < 64) {                          // one full wave
      const TnDynPlan pl = tn_dyn_plan(a.dyn);
      if (threadIdx.x == 0) *a.plan_out = pl;
    }
    return;
  }
  const int d = a.d, T = a.T, B = a.B;
  if ((int)blockIdx.x >= a.L * B) {
    const int id = (int)blockIdx.x - a.L * B;
    if (a.dav_lm) dav_lm_block(a.V, a.v_sB, a.gv, a.dav_out, B, a.N, d, a.L, id % a.dav_gx, id / a.dav_gx);
    else dav_cm_block(a.V, a.v_sB, a.gv, a.dav_out, B, a.N, d, a.L, id % a.dav_gx, id / a.dav_gx, a.dav_gx,
                      reinterpret_cast<float(*)[64]>(lds));
    return;
  }
  float* daq = lds;                       // 32
  const int pairi = blockIdx.x, l = pairi / B, b = pairi - l * B;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const size_t pair = (size_t)pairi;
  // ---- question side: da_q[t] = Q[t] . gq, one wave per row
  const float* Qp = a.Q[l] + (size_t)b * T * d;
  const float* gqp = a.gq + pair * d;
  for (int t0 = w; t0 < T; t0 += 16) {               // four of the wave's rows (t0, t0 + 4, ...) at a time, requested together
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k = 4 * lane; k < d; k += 256) {
      f32x4 x[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) x[i] = *reinterpret_cast<const f32x4*>(Qp + (size_t)min(t0 + 4 * i, T - 1) * d + k);
      const f32x4 g = *reinterpret_cast<const f32x4*>(gqp + k);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] += x[i][0] * g[0] + x[i][1] * g[1] + x[i][2] * g[2] + x[i][3] * g[3];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = wave_sum(acc[i]);
    if (lane == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (t0 + 4 * i < T) daq[t0 + 4 * i] = acc[i];
    }
  }
  __syncthreads();
  if (w == 0) {
    const float aqv = (lane < T) ? a.aq[pair * T + lane] : 0.f;
    const float x = (lane < T) ? daq[lane] : 0.f;
    const float dq = wave_sum(aqv * x);
    const float sq = aqv * (x - dq);
    if (lane < 32) a.dsq[pair * 32 + lane] = sq;                     // (lanes >= T: zeros)
    const float tot_q = wave_sum(sq);
    if (lane == 0) a.dcs_part[(size_t)a.L * B + pair] = tot_q;       // [2][L*B]
  }
}


// dQ_l[b][t][k] = a_q,l[t] gq_l[k] + sum_n dA_l[t][n] V[b][k][n]   for all levels with one pass over V.
// grid (d/128, B); a wave owns 32 channels (two 16-wide MFMA column tiles); dA of the three levels is
// staged zero-padded in LDS and read as MFMA A operands (16 bytes = 4 k-steps per ds_read_b128).

// LM: V location-major [N][d] (dword loads, 64 contiguous bytes per 16 lanes); else channel-major [d][N].
template <int NT, bool ALIGNED, bool LM = false>
__global__ __launch_bounds__(256, 2) void bwd_dq_kernel(const DqArgs a) {
  constexpr int NPAD = 16 * NT;
  constexpr int LD = NPAD + 4;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* dAs = lds;                                  // 3 x kTRows x LD
  float* aqs = lds + 3 * kTRows * LD;                // 3 x 32
  const int b = blockIdx.y, N = a.N, T = a.T, d = a.d, L = a.L, B = a.B;
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, q4 = lane >> 4;
  // stage dA of the three levels, zero padded: one wave per row, lanes along n
  for (int rr = w; rr < 3 * kTRows; rr += 4) {
    const int l = rr / kTRows, row = rr - l * kTRows;
    const bool live = l < L && row < T;
    const float* src = a.dA + (((size_t)l * B + b) * T + row) * N;
    float* dst = dAs + (l * kTRows + row) * LD;
    if (ALIGNED) {
      for (int c4 = lane; c4 < NPAD / 4; c4 += 64) {
        const int col = 4 * c4;
        *reinterpret_cast<f32x4*>(dst + col) =
            (live && col < N) ? *reinterpret_cast<const f32x4*>(src + col) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    } else {
      for (int col = lane; col < NPAD; col += 64) dst[col] = (live && col < N) ? src[col] : 0.f;
    }
  }
  if (tid < 96) {
    const int l = tid >> 5, t = tid & 31;
    aqs[tid] = (l < L && t < T) ? a.aq[((size_t)l * B + b) * T + t] : 0.f;
  }
  __syncthreads();
  const int kb = blockIdx.x * 128 + 32 * w;
  const float* Vb = a.V + (size_t)b * a.v_sB;
  const __amdgpu_buffer_rsrc_t rs_v = make_rsrc(Vb, (unsigned)d * N * 4u);
  f32x4 acc[3][2][2];
#pragma unroll
  for (int l = 0; l < 3; ++l)
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int c = 0; c < 2; ++c) acc[l][tt][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  // B operand: lane (channel kb + 16c + j, quad q4) holds V[k][16g + 4*q4 + s], s = 0..3
  auto load_v = [&](int g, f32x4(&dst)[2]) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      if constexpr (LM) {                            // rows n >= N lie beyond the buffer: read 0
#pragma unroll
        for (int s = 0; s < 4; ++s) dst[c][s] = buf_load1(rs_v, ((4 * q4 + s) * d + 16 * c + j) * 4, (16 * g * d + kb) * 4);
        continue;
      }
      const int voff = ((16 * c + j) * N + 4 * q4) * 4;
      if (ALIGNED) {
        dst[c] = buf_load4(rs_v, voff, (kb * N + 16 * g) * 4);
      } else {
#pragma unroll
        for (int s = 0; s < 4; ++s) dst[c][s] = buf_load1(rs_v, voff + 4 * s, (kb * N + 16 * g) * 4);
      }
    }
  };
  f32x4 vb[2][2];
  load_v(0, vb[0]);
#pragma unroll
  for (int g = 0; g < NT; ++g) {
    if (g + 1 < NT) load_v(g + 1, vb[(g + 1) & 1]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int l = 0; l < 3; ++l)
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        const int t = min(16 * tt + j, kTRows - 1);
        const f32x4 a4 = *reinterpret_cast<const f32x4*>(&dAs[(l * kTRows + t) * LD + 16 * g + 4 * q4]);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int c = 0; c < 2; ++c) acc[l][tt][c] = mfma16(a4[s], vb[g & 1][c][s], acc[l][tt][c]);
      }
  }
  // epilogue: + a_q (x) gq ; rows t >= T fall outside the per-sample dQ buffer (stores dropped)
#pragma unroll
  for (int l = 0; l < 3; ++l) {
    if (l < L) {
      const __amdgpu_buffer_rsrc_t rs_dq = make_rsrc(a.dQ[l] + (size_t)b * T * d, (unsigned)T * d * 4u);
      const float* gqp = a.gq + ((size_t)l * B + b) * d;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const float g = gqp[kb + 16 * c + j];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int t = 16 * tt + 4 * q4 + r;
            const float v = fmaf(aqs[l * 32 + t], g, acc[l][tt][c][r]);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs_dq, (t * d + j) * 4 + 64 * c,
                                                  kb * 4, 0);
          }
      }
    }
  }
}

template <typename K>
hipError_t set_lds(K kern, size_t bytes) {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

int launch_pre(const PreArgs& a, hipStream_t s) {
  const size_t lds = 768;                                           // the 3 x 64 g values of a channel-major da_v block; 32 da_q
  hipLaunchKernelGGL(bwd_pre_kernel, dim3(a.L * a.B + a.dav_gx * a.B + (a.plan_out ? 1 : 0)), dim3(256), lds, s, a);
  CA_CHECK_LAUNCH("bwd_pre");
  // (the image side's softmax backward -- ds_v from these partials -- happens in the prologues of bwd_dc32_kernel and
  //  bwd_nat32_kernel: fused.h softmax_bwd_v)
  return 0;
}

}  // namespace

size_t fused_bwd_ws_floats(int B, int N, int T, int d, int L) { return fused_bwd_off(B, N, T, d, L).total; }

int fused_backward_supported(int B, int N, int T, int d, int L) { return fused_supported(B, N, T, d, L); }

int fused_backward(int B, int N, int T, int d, int L, const float* V, const VLayout& vl, const float* const* Q,
                   const coattn_params* p, const float* saved, const float* gv, const float* gq, float* dV,
                   const VLayout& dvl, float* const* dQ, const coattn_param_grads* pg, int accumulate, float* ws,
                   hipStream_t s, int bf16_proj, int wgemm, int np, int live_rows) {
  // np: width of the fp32 mode's contractions (fused.h): 2 = hi + mid in the three fused kernels and in the GEMM launch
  // (dW_v, dW_q, dQ = dP_q W_q), 3 = the exact split everywhere; dV (general GEMM) is always exact
  np = (np == 2 && !bf16_proj) ? 2 : 3;
  CA_CHECK_ARG(fused_backward_supported(B, N, T, d, L), "fused backward: unsupported shape");
  const bool lm = v_is_lm(vl, N, d);
  CA_CHECK_ARG(lm || v_is_cm(vl, N, d), "fused backward: image features must be channel-major [B,d,N] or location-major [B,N,d]");
  // gradients of the two projections (d x d contractions): fp32 MFMA, or bf16 MFMA under COATTN_FLAG_BF16_PROJ
  auto gemm_proj = [&](const coattn_gemm_desc& g) { return bf16_proj ? launch_gemm_bf16in(g, s) : launch_gemm_f32(g, s); };
  const SavedOff so = saved_off(B, N, T, d, L);
  const FusedBwdOff wo = fused_bwd_off(B, N, T, d, L);
  const size_t BTd = (size_t)B * T * d, BTN = (size_t)B * T * N, BNd = (size_t)B * N * d, Bd = (size_t)B * d;
  const bool small_n = N <= 64;
  // 1. per-sample pre-pass: one launch for the question side and the da_v partials, a small one for the image side
  CA_CHECK_ARG(N <= 256, "fused backward: N > 256");
  PreArgs pa;
  pa.V = V; pa.v_sB = vl.sB; pa.gv = gv; pa.dav_out = ws + wo.part; pa.dav_lm = lm ? 1 : 0;
  pa.dav_gx = lm ? (N + 15) / 16 : d / 64;
  pa.dav_part = ws + wo.part;
  pa.nkc = lm ? 1 : d / 64;
  for (int l = 0; l < 8; ++l) pa.Q[l] = l < L ? Q[l] : nullptr;
  pa.gq = gq; pa.av = saved + so.av; pa.aq = saved + so.aq;
  pa.dsq = ws + wo.dsq; pa.dcs_part = ws + wo.dcs_part;
  pa.B = B; pa.N = N; pa.T = T; pa.d = d; pa.L = L;
  // (exact mode: the forward's bitmap of the live question rows is in `saved`; the plan of the weight gradients over those rows
  //  is a function of it and of shapes known here -- evaluated once, by an extra workgroup of this launch)
  static const int tn_budget_env = dev_env_int("COATTN_TN_PARTS", 0);   // developer switch
  const int tn_budget = tn_budget_env > 0 ? tn_budget_env : 32;
  TnDyn dyn_all = {};
  pa.dyn = dyn_all; pa.plan_out = nullptr;
  if (live_rows && L <= kDynLevels && (B * T + 31) / 32 <= kRowBitsMaxWords && tn_budget > L) {
    dyn_all.bits = reinterpret_cast<const unsigned*>(saved + so.rowbits);
    dyn_all.words = (B * T + 31) / 32; dyn_all.levels = L; dyn_all.P = tn_budget; dyn_all.K0 = B * N;
    dyn_all.plan = reinterpret_cast<const TnDynPlan*>(ws + wo.dynplan);
    pa.dyn = dyn_all; pa.plan_out = reinterpret_cast<TnDynPlan*>(ws + wo.dynplan);
  }
  CA_TRY(launch_pre(pa, s));
  prof_mark(s, "bwd_pre");
  // 2. the two recompute kernels
  BwdArgs ba;
  ba.Pv = saved + so.Pv; ba.Pq = saved + so.Pq; ba.C = saved + so.C;
  ba.Hq = saved + so.Hq; ba.dsq = ws + wo.dsq; ba.wq = (const float*)p->w_q; ba.dwq_part = ws + wo.dwq_part;
  ba.dav_part = ws + wo.part; ba.nkc = pa.nkc; ba.av = saved + so.av; ba.dcs_part = ws + wo.dcs_part;
  ba.wv = (const float*)p->w_v;
  ba.dPv = ws + wo.dPv; ba.dPq = ws + wo.dPq; ba.dA = ws + wo.dA; ba.dwv_part = ws + wo.dwv_part;
  ba.dbv_part = ws + wo.dbv_part; ba.dbq_part = ws + wo.dbq_part;
  ba.B = B; ba.N = N; ba.T = T; ba.d = d; ba.L = L;
  ba.bf16 = bf16_proj;
  ba.np = np;
  static const int ko_dpv = dev_env_int("COATTN_KO_DPV", 0);      // developer knock-outs (DEV builds; wrong results): what would
  static const int ko_sum3 = dev_env_int("COATTN_KO_SUM3", 0);    // ONE sum_l dP_v array save in bwd_nat32's stores / the GEMM's reads
  ba.ko_dpv = ko_dpv;
  ba.dp_bf16 = 0;
  CA_TRY(launch_bwd_dc32(ba, s));                    // dC, dA                       (coattn_bwd32.hip)
  prof_mark(s, "bwd_dc32");
  // which of the backward's GEMMs take the hand-scheduled kernels (decided here: when all three do, they share ONE
  // launch in step 5 -- weight gradients, the dQ projection's tiles and the small reductions)
  float* dPv = ws + wo.dPv;
  float* part = ws + wo.part;
  TnGemm tnv = {};
  tnv.A = dPv; tnv.a_ld = d; tnv.B = V; tnv.b_ld = (int)vl.sN; tnv.C = part; tnv.M = d; tnv.N = d; tnv.K = B * N; tnv.levels = 1;
  bool tn_v = false;
  tnv.bf16 = bf16_proj; tnv.np = np;
  if (wgemm && lm && vl.sB == (long)N * vl.sN && vl.sN < (1L << 24)) {
    tn_v = gemm_tn_supported(tnv) != 0;              // location-major rows, samples abutting
  } else if (wgemm && !lm && vl.sD < (1L << 24)) {
    tnv.b_ld = (int)vl.sD; tnv.b_kdiv = N; tnv.b_sdiv = vl.sB;          // channel-major, read in place
    tn_v = gemm_tn_supported(tnv) != 0;
  }
  TnGemm tnq = {};
  tnq.A = ws + wo.dPq; tnq.a_sl = (long)BTd; tnq.a_ld = d; tnq.b_ld = d; tnq.M = d; tnq.N = d; tnq.K = B * T; tnq.levels = L;
  for (int l = 0; l < L; ++l) tnq.b_ptrs[l] = Q[l];
  tnq.bf16 = bf16_proj; tnq.np = np;
  const bool tn_q = wgemm && gemm_tn_supported(tnq);       // levels as extra split-K parts (gemm_tn.hip)
  const bool dq32 = lm || (N % 4) == 0;              // the bf16 dA V kernel takes both layouts (channel-major: aligned rows)
  WGemm wdq = {};                                    // dQ_l = dP_q,l W_q against the W_q image the forward left in `saved`
  wdq.A = ws + wo.dPq; wdq.a_sz = (long)BTd; wdq.a_sm = d; wdq.Wf = saved + so.wqT;
  for (int l = 0; l < L; ++l) wdq.c_ptrs[l] = dQ[l];
  wdq.c_sm = d; wdq.M = B * T; wdq.N = d; wdq.K = d; wdq.batch = L;
  // (exactly the forward's test for writing that image, api.hip general_projections: same shape, and its A rows were Q_l)
  bool q_al = true;
  for (int l = 0; l < L; ++l) q_al = q_al && (((uintptr_t)Q[l]) & 15) == 0;
  wdq.bf16 = bf16_proj; wdq.np = np;
  const bool wdq_ok = wgemm && q_al && gemm_w_supported(wdq);
  // (a dQ projection on gemm_bf.hip -- 512-thread workgroups -- cannot ride in the weight-gradient launch)
  static const int no_combine = dev_env_int("COATTN_NO_COMBINE", 0);   // developer switch
  // COATTN_OWN_DQ=1 (developer switch; round 5, measured and NOT kept): the dQ projection as a launch of its own on the
  // persistent pipeline of gemm_h2.hip (bf16 pieces) between the weight gradients and the dA V kernel -- 98.9 + 28.5 us against
  // 123.2 combined at N = 196, 41.8 + 26.6 against 63.7 at N = 49: inside the weight-gradient launch its tiles fill the CUs
  // the last parts leave, which is worth as much as the faster kernel.
  static const int own_dq_env = dev_env_int("COATTN_OWN_DQ", 0);
  const bool own_dq = own_dq_env && dq32 && wdq_ok && tn_v && tn_q && !gemm_bf_supported(wdq) && gemm_h2_supported(wdq) && !no_combine;
  const bool combine = dq32 && wdq_ok && tn_v && tn_q && !gemm_bf_supported(wdq) && !no_combine && !own_dq;
  const bool late_dq = combine || own_dq;           // the dA V kernel (and, before it, the projection) run after the weight gradients
  // Reduced-precision mode with a frozen image encoder (no dV) and all three consumers of dP_v / dP_q on gemm_bf.hip:
  // bwd_nat32 stores both as bf16 -- the GEMMs would round them on their way in anyway -- halving what it writes and
  // what they fetch.
  {
    TnGemm tv = tnv, tq = tnq;
    WGemm wq = wdq;
    tv.a_bf16 = tq.a_bf16 = wq.a_bf16 = 1;
    tv.a_term = L == 3 ? (long)BNd : 0;
    static const int off = (dev_env_int("COATTN_DP_BF16", 1) == 0);   // developer switch
    if (!off && bf16_proj && !dV && L == 3 && d % 512 == 0 && dq32 && wdq_ok && tn_v && tn_q && gemm_bf_tn_supported(tv) &&
        gemm_bf_tn_supported(tq) && gemm_bf_supported(wq)) {
      ba.dp_bf16 = 1;
      tnv.a_bf16 = tnq.a_bf16 = wdq.a_bf16 = 1;
      tnq.a_sl = (long)BTd; wdq.a_sz = (long)BTd;   // (elements, as before)
    }
  }
  CA_TRY(launch_bwd_nat32(ba, s));                   // dP_q, dP_v, dw_v, db_v, db_q (coattn_bwd32.hip)
  prof_mark(s, "bwd_nat32");
  // 3. small parameter gradients from the per-(sample, level) partials (dw_v, db_v, db_q, dw_q, and dc_v, dc_q as
  //    whole-array sums): a few short workgroups -- riding along in the weight-gradient launch of step 5 when that
  //    is the hand-scheduled one, else a launch of their own
  TnReduce small = {};
  {
    const float* src[4] = {ws + wo.dwv_part, ws + wo.dbv_part, ws + wo.dbq_part, ws + wo.dwq_part};
    float* dst[4] = {(float*)pg->dw_v, (float*)pg->db_v, (float*)pg->db_q, (float*)pg->dw_q};
    for (int i = 0; i < 4; ++i) { small.src[i] = src[i]; small.dst[i] = dst[i]; }
    small.njobs = 4; small.nparts = L * B; small.n = d; small.accumulate = accumulate;
    small.sum_x[0] = ws + wo.dcs_part; small.sum_x[1] = ws + wo.dcs_part + (size_t)L * B;
    small.sum_out[0] = (float*)pg->dc_v; small.sum_out[1] = (float*)pg->dc_q; small.sum_n = (long)L * B;
  }
  auto small_reductions = [&]() -> int {
    return launch_reduce_jobs(small.src, small.dst, 4, L * B, d, accumulate, s, small.sum_x, small.sum_out, small.sum_n);
  };
  // dQ_l (+)= dP_q,l W_q for all levels in one launch (batch z = level, C through the pointer table)
  auto dq_projection = [&](bool onto_dq) -> int {
    // (W_q split once by the forward's launch -- the same shape test decided there, api.hip general_projections --
    //  and read as MFMA fragments, gemm_w.hip)
    if (!onto_dq && wdq_ok) return launch_gemm_wx(&wdq, 1, s);
    coattn_gemm_desc g = {};
    g.A = ws + wo.dPq; g.a_sz = (int64_t)BTd; g.a_sm = d; g.a_sk = 1;
    g.B = p->W_q; g.b_sk = d; g.b_sn = 1;
    for (int l = 0; l < L; ++l) { g.c_ptrs[l] = dQ[l]; if (onto_dq) g.cin_ptrs[l] = dQ[l]; }
    if (onto_dq) { g.cin_sm = d; g.cin_sn = 1; g.beta = 1.f; }
    g.c_sm = d; g.c_sn = 1;
    g.M = B * T; g.N = d; g.K = d; g.batch = L;
    return gemm_proj(g);
  };
  // 4. dQ_l = a_q (x) gq + dA V^T + dP_q W_q ;  dV = sum_l (a_v (x) gv + Q^T dA) + (sum_l dP_v) W_v
  //    The projection writes dQ first and the bf16 dA V kernel adds onto it (the GEMM is 24 us faster without an
  //    accumulate input); channel-major features with unaligned rows (N % 4 != 0): the exact-f32 kernel first, then
  //    the projection onto it.
  //    When the projection shares the weight-gradient launch (step 5), the dA V kernel runs after that launch.
  if (dq32 && !late_dq) {
    CA_TRY(dq_projection(false));
    prof_mark(s, "bwd_gemm_dq_projection");
  }
  // (red: the two weight gradients' partial sums, handed to the dQ kernel's launch when it is the bf16-MFMA one)
  struct RedJob { const float* part[2]; float* out[2]; int np[2]; long n; int acc; bool on; TnDyn dyn; } red = {};
  static const int red_in_dq = dev_env_int("COATTN_RED_IN_DQ", 1);   // developer switch
  auto run_dq = [&]() -> int {
    DqArgs da = {};
    da.accumulate = dq32 ? 1 : 0;
    if (red.on && dq32) {
      for (int i = 0; i < 2; ++i) { da.red_part[i] = red.part[i]; da.red_out[i] = red.out[i]; da.red_np[i] = red.np[i]; }
      da.red_n = red.n; da.red_acc = red.acc; da.red_jobs = 2; da.red_blocks = (int)((red.n / 4 + 255) / 256);
      da.red_dyn = red.dyn;
    }
    da.V = V; da.v_sB = vl.sB; da.dA = ws + wo.dA; da.aq = saved + so.aq; da.gq = gq;
    for (int l = 0; l < 8; ++l) da.dQ[l] = l < L ? dQ[l] : nullptr;
    da.B = B; da.N = N; da.T = T; da.d = d; da.L = L;
    da.bf16 = bf16_proj; da.np = np;
    const bool al = (N % 4) == 0;
    dim3 grid(d / 128, B), block(256);
    if (dq32) {
      CA_TRY(launch_bwd_dq32(da, lm ? 1 : 0, s));    // bf16 MFMA kernel (coattn_bwd32.hip)
    } else if (small_n) {
      const size_t lds = (size_t)(3 * kTRows * (64 + 4) + 96) * sizeof(float);
      if (lm && al) hipLaunchKernelGGL((bwd_dq_kernel<4, true, true>), grid, block, lds, s, da);
      else if (lm) hipLaunchKernelGGL((bwd_dq_kernel<4, false, true>), grid, block, lds, s, da);
      else if (al) hipLaunchKernelGGL((bwd_dq_kernel<4, true>), grid, block, lds, s, da);
      else hipLaunchKernelGGL((bwd_dq_kernel<4, false>), grid, block, lds, s, da);
    } else {
      const size_t lds = (size_t)(3 * kTRows * (208 + 4) + 96) * sizeof(float);
      static DeviceOnce once;
      CA_TRY(once.run([&] {
        hipError_t e = set_lds(bwd_dq_kernel<13, true>, lds);
        if (e == hipSuccess) e = set_lds(bwd_dq_kernel<13, false>, lds);
        if (e == hipSuccess) e = set_lds(bwd_dq_kernel<13, true, true>, lds);
        if (e == hipSuccess) e = set_lds(bwd_dq_kernel<13, false, true>, lds);
        return e;
      }, "bwd_dq"));
      if (lm && al) hipLaunchKernelGGL((bwd_dq_kernel<13, true, true>), grid, block, lds, s, da);
      else if (lm) hipLaunchKernelGGL((bwd_dq_kernel<13, false, true>), grid, block, lds, s, da);
      else if (al) hipLaunchKernelGGL((bwd_dq_kernel<13, true>), grid, block, lds, s, da);
      else hipLaunchKernelGGL((bwd_dq_kernel<13, false>), grid, block, lds, s, da);
    }
    CA_CHECK_LAUNCH("bwd_dq");
    prof_mark(s, "bwd_dq");
    return 0;
  };
  if (!late_dq) CA_TRY(run_dq());
  if (dV) {
    for (int l = 0; l < L; ++l) {
      const float* dA = ws + wo.dA + l * BTN;
      const float* av = saved + so.av + (size_t)l * B * N;
      CA_TRY(launch_rank1(av, gv + l * Bd, dV, B, N, d, dvl.sB, dvl.sN, dvl.sD, l > 0 ? 1 : 0, s));
      coattn_gemm_desc g = {};
      g.A = Q[l]; g.a_sz = (int64_t)T * d; g.a_sm = 1; g.a_sk = d;
      g.B = dA; g.b_sz = (int64_t)T * N; g.b_sk = N; g.b_sn = 1;
      g.Cin = dV; g.cin_sz = dvl.sB; g.cin_sm = dvl.sD; g.cin_sn = dvl.sN; g.beta = 1.f;
      g.C = dV; g.c_sz = dvl.sB; g.c_sm = dvl.sD; g.c_sn = dvl.sN;
      g.M = d; g.N = N; g.K = T; g.batch = B;
      CA_TRY(launch_gemm_f32(g, s));
    }
  }
  if (!dq32) CA_TRY(dq_projection(true));
  // sum dP_v over the levels in place into level 0 (one streaming pass for L = 3; folding the sum into
  // the weight-gradient GEMM's operand loads was measured slower: 302 vs 170 + 50 us)
  // (the frozen-encoder default needs no dV: the weight-gradient kernel then adds the three levels while staging them)
  const bool sum_in_gemm = tn_v && L == 3 && !dV;
  if (sum_in_gemm) {
    tnv.a_term = ko_sum3 ? 0 : (long)BNd;
  } else if (L == 3) {
    CA_TRY(launch_add3_inplace(dPv, dPv + BNd, dPv + 2 * BNd, (int64_t)BNd, s));
  } else {
    for (int l = 1; l < L; ++l) CA_TRY(launch_add_inplace(dPv, dPv + l * BNd, (int64_t)BNd, 1, s));
  }
  if (dV) {
    // dV[b][k][n] += sum_j W_v[j][k] dP_v[b][n][j]
    coattn_gemm_desc g = {};
    g.A = p->W_v; g.a_sm = 1; g.a_sk = d; g.a_sz = 0;
    g.B = dPv; g.b_sz = (int64_t)N * d; g.b_sk = 1; g.b_sn = d;
    g.Cin = dV; g.cin_sz = dvl.sB; g.cin_sm = dvl.sD; g.cin_sn = dvl.sN; g.beta = 1.f;
    g.C = dV; g.c_sz = dvl.sB; g.c_sm = dvl.sD; g.c_sn = dvl.sN;
    g.M = d; g.N = N; g.K = d; g.batch = B;
    CA_TRY(gemm_proj(g));
  }
  // 5. weight gradients
  if (tn_v && tn_q && gemm_bf_tn_supported(tnv) && gemm_bf_tn_supported(tnq)) {
    // reduced-precision mode at wide shapes (config 4): the single-product kernel of gemm_bf.hip, 256 x 256 tiles; the
    // parts of the two products share two rounds of workgroups in proportion to their contraction lengths
    const int ntiles = (d / 256) * (d / 256);
    int total = (bf_tn_rounds() * 256 + ntiles - 1) / ntiles;
    total = total < 2 ? 2 : (total > kMaxParts ? kMaxParts : total);
    const double kv = (double)B * N, kq = (double)L * B * T;
    int pv = (int)(total * kv / (kv + kq) + 0.5);
    pv = pv < 1 ? 1 : (pv > total - 1 ? total - 1 : pv);
    int spp[2], parts[2];
    parts[0] = gemm_bf_tn_plan(tnv, pv, &spp[0]);
    tnq.C = part + (size_t)parts[0] * d * d;
    parts[1] = gemm_bf_tn_plan(tnq, total - pv, &spp[1]);
    CA_CHECK_ARG(parts[0] + parts[1] <= kMaxParts, "fused backward: %d split-K parts exceed the workspace", parts[0] + parts[1]);
    CA_TRY(small_reductions());
    const TnGemm both[2] = {tnv, tnq};
    CA_TRY(launch_gemm_bf_tn(both, spp, parts, 2, s));
    prof_mark(s, "bwd_gemm_dw");
    if (combine) CA_TRY(run_dq());
    CA_TRY(launch_reduce_partials2(part, (float*)pg->dW_v, parts[0], tnq.C, (float*)pg->dW_q, parts[1], (int64_t)d * d,
                                   accumulate, s));
    prof_mark(s, "reduce_partials");
    return 0;
  }
  if (tn_v && tn_q) {
    // both weight gradients in one launch: 32 split-K parts (x 16 tiles = the 512 workgroup slots) shared in
    // proportion to the contraction lengths, so that all workgroups run about equally long
    const double kv = (double)B * N, kq = (double)L * B * T;
    const int budget = tn_budget;
    int pv = (int)((double)budget * kv / (kv + kq) + 0.5);
    pv = pv < 1 ? 1 : (pv > budget - 1 ? budget - 1 : pv);
    const int pq = (budget - pv) / L > 0 ? (budget - pv) / L * L : L;
    int ks[2], S[2];
    // two-piece width: 128 x 256 tiles on 512-thread workgroups (gemm_tn_wide.hip) -- the same split-K partition, pieces and
    // order of products, so the same bits as the 128 x 128 kernel
    const bool wide = gemm_tn_wide_supported(tnv) && gemm_tn_wide_supported(tnq) && (!combine || (wdq.N % 256 == 0 && !wdq.f16 && (wdq.np == 2) == (tnv.np == 2)));
    const bool red_al = (((int64_t)d * d) & 3) == 0 && ((((uintptr_t)part) | ((uintptr_t)pg->dW_v) | ((uintptr_t)pg->dW_q)) & 15) == 0;
    // The forward left the bitmap of the question rows that are not all zeros in `saved` (exact mode; api.hip rowbits_in_saved):
    // dW_q = sum dP_q^T Q then contracts over those rows only, and the launch shares its `budget` parts between dW_v and the
    // levels of dW_q ON THE DEVICE (fused.h TnDyn) -- the partial sums are added by the dQ kernel's extra workgroups, which
    // evaluate the same plan.  COATTN_DW_LIVE_ROWS=0 (developer switch): the host's static plan over all rows.
    static const int live_env = dev_env_int("COATTN_DW_LIVE_ROWS", 1);
    TnDyn dyn = {};
    const bool use_dyn = live_env && dyn_all.bits && wide && late_dq && dq32 && red_in_dq && red_al && tnq.K <= 8192 && !tnq.a_bf16 &&
                         (tnq.K + 31) / 32 <= kRowBitsMaxWords && budget > L && budget <= kMaxParts && L <= kDynLevels;
    if (use_dyn) dyn = dyn_all;
    const int parts_v = wide ? gemm_tn_wide_plan(tnv, pv, &ks[0], &S[0]) : gemm_tn_plan(tnv, pv, &ks[0], &S[0]);
    tnq.C = use_dyn ? part : part + (size_t)parts_v * d * d;
    const int parts_q = wide ? gemm_tn_wide_plan(tnq, pq, &ks[1], &S[1]) : gemm_tn_plan(tnq, pq, &ks[1], &S[1]);
    CA_CHECK_ARG(parts_v + parts_q <= kMaxParts, "fused backward: %d split-K parts exceed the workspace", parts_v + parts_q);
    const TnGemm both[2] = {tnv, tnq};
    if (wide) CA_TRY(launch_gemm_tn_wide(both, ks, S, 2, s, &small, combine ? &wdq : nullptr, use_dyn ? &dyn : nullptr));
    else CA_TRY(launch_gemm_tn(both, ks, S, 2, s, &small, combine ? &wdq : nullptr));
    prof_mark(s, combine ? "bwd_gemm" : "bwd_gemm_dw");
    if (own_dq) {
      CA_TRY(dq_projection(false));
      prof_mark(s, "bwd_gemm_dq_projection");
    }
    if (late_dq && dq32 && red_in_dq && red_al) {       // the partial sums ride in the dQ kernel's launch
      red.part[0] = part; red.out[0] = (float*)pg->dW_v; red.np[0] = parts_v;
      red.part[1] = tnq.C; red.out[1] = (float*)pg->dW_q; red.np[1] = parts_q;
      red.n = (long)d * d; red.acc = accumulate; red.on = true; red.dyn = dyn;
      return run_dq();
    }
    if (late_dq) CA_TRY(run_dq());
    CA_TRY(launch_reduce_partials2(part, (float*)pg->dW_v, parts_v, tnq.C, (float*)pg->dW_q, parts_q, (int64_t)d * d,
                                   accumulate, s));
    prof_mark(s, "reduce_partials");
    return 0;
  }
  tnq.C = part;
  CA_TRY(small_reductions());
  {
    // dW_v[j][k] = sum_{b,n} dP_v[b][n][j] V[b][k][n]
    coattn_gemm_desc g = {};
    int S;
    if (tn_v) {
      // both operands row-major over the B*N contraction rows: the hand-scheduled A^T B kernel (gemm_tn.hip)
      int ks;
      const int parts = gemm_tn_plan(tnv, 32, &ks, &S);
      CA_TRY(launch_gemm_tn(&tnv, &ks, &S, 1, s));
      CA_TRY(launch_reduce_partials(part, (float*)pg->dW_v, parts, (int64_t)d * d, accumulate, s));
    } else {
    if (lm && vl.sB == (long)N * d) {
      // location-major, samples abutting: one flat contraction over m = (b, n), split-K over the B*N rows
      const int K = B * N;
      int ks = (K + 31) / 32;
      ks = (ks + 15) / 16 * 16;
      S = (K + ks - 1) / ks;
      g.A = dPv; g.a_sm = 1; g.a_sk = d;
      g.B = V; g.b_sk = d; g.b_sn = 1;
      g.M = d; g.N = d; g.K = K; g.batch = S; g.ksplit = ks;
    } else {
      // inner index = sample, split into <= 32 groups
      const int G = (B + 31) / 32;
      S = (B + G - 1) / G;
      g.A = dPv; g.a_sm = 1; g.a_sk = d; g.a_si = (int64_t)N * d; g.a_sz = (int64_t)G * N * d;
      g.B = V; g.b_sk = vl.sN; g.b_sn = vl.sD; g.b_si = vl.sB; g.b_sz = (int64_t)G * vl.sB;
      g.M = d; g.N = d; g.K = N; g.batch = S; g.inner = G; g.inner_total = B;
    }
    g.C = part; g.c_sz = (int64_t)d * d; g.c_sm = d; g.c_sn = 1;
    CA_TRY(gemm_proj(g));
    CA_TRY(launch_reduce_partials(part, (float*)pg->dW_v, S, (int64_t)d * d, accumulate, s));
    }
  }
  {
    // dW_q[j][k] = sum_l sum_m dP_q,l[m][j] Q_l[m][k]: levels as the inner loop (B from the pointer
    // table), split-K over the B*T rows
    const int K = B * T;
    if (tn_q) {
      int ks, S;
      const int parts = gemm_tn_plan(tnq, 32, &ks, &S);
      CA_TRY(launch_gemm_tn(&tnq, &ks, &S, 1, s));
      return launch_reduce_partials(part, (float*)pg->dW_q, parts, (int64_t)d * d, accumulate, s);
    }
    int ks = (K + 31) / 32;
    ks = (ks + 15) / 16 * 16;
    const int S = (K + ks - 1) / ks;
    coattn_gemm_desc g = {};
    g.A = ws + wo.dPq; g.a_sm = 1; g.a_sk = d; g.a_si = (int64_t)BTd;
    for (int l = 0; l < L; ++l) g.b_ptrs[l] = Q[l];
    g.ptr_by_inner = 1; g.b_sk = d; g.b_sn = 1;
    g.C = part; g.c_sz = (int64_t)d * d; g.c_sm = d; g.c_sn = 1;
    g.M = d; g.N = d; g.K = K; g.batch = S; g.ksplit = ks; g.inner = L;
    CA_TRY(gemm_proj(g));
    CA_TRY(launch_reduce_partials(part, (float*)pg->dW_q, S, (int64_t)d * d, accumulate, s));
  }
  return 0;
}
