// Fused backward of the parallel co-attention (hand-derived; SURVEY.md section 8, checked
// against autograd of the reference by the oracle).  fp32, exact-f32 MFMA 16x16x4.
//
// H_v [N,d] is never stored: both big kernels recompute it from P_v, P_q and C (saved).
//
//   bwd_pre_kernel  (per sample, all levels)  one pass over V: da_v = V gv -> softmax backward
//                   ds_v; da_q = Q gq -> ds_q; dZ_q = ds_q (x) w_q (.) (1 - H_q^2); dw_q / dc partials.
//   bwd_dc_kernel   (per sample x level; the wave owns 128 channels, outer loop over 16-channel
//                   tiles, inner over 16-location tiles, orientation [d][n]):
//                   H_v^T tile = P_v^T + P_q^T C -> dZ_v^T -> dP_v^T = dZ_v^T + dZ_q^T C (stored),
//                   dw_v partials, and dC += P_q dZ_v^T + dZ_q P_v^T with the dZ_v^T / P_v^T
//                   accumulator registers reused directly as MFMA B operands (contraction over d);
//                   cross-wave tree sum through LDS, dA = dC (.) (1 - C^2).
//   bwd_dpq_kernel  (per sample x level, orientation [n][d], same loop as forward phase 2):
//                   H_v tile -> dZ_v tile, which is the B operand of dP_q += C dZ_v
//                   (contraction over N).
//   then MFMA GEMMs: dQ = a_q (x) gq + dA V^T + dP_q W_q;  dV = sum_l (a_v (x) gv + Q^T dA) +
//   (sum_l dP_v) W_v (skipped when the image features need no gradient);  dW_v, dW_q, biases.
#include "fused.h"

namespace {

struct PreArgs {
  const float* V; const float* Q[8];
  const float* gv; const float* gq;      // [L][B][d]
  const float* av; const float* aq;      // saved
  const float* Hq;                       // saved [L][B][T][d]
  const float* wq;
  float* dsv;                            // [L][B][N]
  float* dZq;                            // [L][B][T][d]
  float* dwq_part;                       // [B][d]
  float* dcs_part;                       // [B][2]
  int B, N, T, d, L;
};

template <int NT>
__global__ __launch_bounds__(256) void bwd_pre_kernel(const PreArgs a) {
  constexpr int NPAD = 16 * NT;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int d = a.d, N = a.N, T = a.T, L = a.L, B = a.B;
  float* gvs = lds;                       // 3 x d
  float* gqs = gvs + 3 * d;               // 3 x d
  float* red = gqs + 3 * d;               // 16 x 3 x NPAD
  float* dav = red + 16 * 3 * NPAD;       // 3 x NPAD
  float* daq = dav + 3 * NPAD;            // 3 x 32
  float* dsq = daq + 96;                  // 3 x 32
  float* dcs = dsq + 96;                  // 8
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  for (int e = tid; e < 3 * d; e += 256) {
    const int l = e / d, k = e - l * d;
    gvs[e] = (l < L) ? a.gv[((size_t)l * B + b) * d + k] : 0.f;
    gqs[e] = (l < L) ? a.gq[((size_t)l * B + b) * d + k] : 0.f;
  }
  if (tid < 8) dcs[tid] = 0.f;
  __syncthreads();
  // ---- da_v[l][n] = sum_k V[k][n] gv[l][k]: 16 lanes per channel row, 16 rows per sweep
  {
    const int j = tid & 15, rs = tid >> 4;
    float acc[3][NT];
#pragma unroll
    for (int l = 0; l < 3; ++l)
#pragma unroll
      for (int m = 0; m < NT; ++m) acc[l][m] = 0.f;
    const float* Vb = a.V + (size_t)b * d * N;
    for (int k = rs; k < d; k += 16) {
      const float* vr = Vb + (size_t)k * N;
      const float g0 = gvs[k], g1 = gvs[d + k], g2 = gvs[2 * d + k];
#pragma unroll
      for (int m = 0; m < NT; ++m) {
        const int n = j + 16 * m;
        const float x = (n < N) ? vr[n] : 0.f;
        acc[0][m] = fmaf(x, g0, acc[0][m]);
        acc[1][m] = fmaf(x, g1, acc[1][m]);
        acc[2][m] = fmaf(x, g2, acc[2][m]);
      }
    }
#pragma unroll
    for (int l = 0; l < 3; ++l)
#pragma unroll
      for (int m = 0; m < NT; ++m) red[(rs * 3 + l) * NPAD + j + 16 * m] = acc[l][m];
  }
  __syncthreads();
  for (int e = tid; e < 3 * NPAD; e += 256) {
    float s = 0.f;
#pragma unroll
    for (int rs = 0; rs < 16; ++rs) s += red[rs * 3 * NPAD + e];
    dav[e] = s;
  }
  // ---- da_q[l][t] = Q_l[t] . gq_l : one wave per (l,t)
  for (int idx = w; idx < L * T; idx += 4) {
    const int l = idx / T, t = idx - l * T;
    const float* qr = a.Q[l] + ((size_t)b * T + t) * d;
    float acc = 0.f;
    for (int k = lane; k < d; k += 64) acc = fmaf(qr[k], gqs[l * d + k], acc);
    acc = wave_sum(acc);
    if (lane == 0) daq[l * 32 + t] = acc;
  }
  __syncthreads();
  // ---- softmax backward: ds = a (.) (da - <a, da>)
  if (w < L) {
    const int l = w;
    const float* avp = a.av + ((size_t)l * B + b) * N;
    float dot = 0.f;
    for (int n = lane; n < N; n += 64) dot = fmaf(avp[n], dav[l * NPAD + n], dot);
    dot = wave_sum(dot);
    float tot = 0.f;
    for (int n = lane; n < N; n += 64) {
      const float v = avp[n] * (dav[l * NPAD + n] - dot);
      a.dsv[((size_t)l * B + b) * N + n] = v;
      tot += v;
    }
    tot = wave_sum(tot);
    const float aqv = (lane < T) ? a.aq[((size_t)l * B + b) * T + lane] : 0.f;
    const float x = (lane < T) ? daq[l * 32 + lane] : 0.f;
    const float dq = wave_sum(aqv * x);
    const float sq = aqv * (x - dq);
    if (lane < 32) dsq[l * 32 + lane] = (lane < T) ? sq : 0.f;
    const float totq = wave_sum(sq);
    if (lane == 0) { dcs[l] = tot; dcs[4 + l] = totq; }
  }
  __syncthreads();
  // ---- dZ_q = ds_q (x) w_q (.) (1 - H_q^2); dw_q partial of this sample (summed over levels)
  for (int dd = tid; dd < d; dd += 256) {
    const float wqv = a.wq[dd];
    float dw = 0.f;
    for (int l = 0; l < L; ++l)
      for (int t = 0; t < T; ++t) {
        const size_t o = (((size_t)l * B + b) * T + t) * d + dd;
        const float h = a.Hq[o];
        const float s = dsq[l * 32 + t];
        dw = fmaf(s, h, dw);
        a.dZq[o] = s * wqv * (1.0f - h * h);
      }
    a.dwq_part[(size_t)b * d + dd] = dw;
  }
  if (tid == 0) {
    a.dcs_part[(size_t)b * 2 + 0] = dcs[0] + dcs[1] + dcs[2];
    a.dcs_part[(size_t)b * 2 + 1] = dcs[4] + dcs[5] + dcs[6];
  }
}

struct BwdArgs {
  const float* Pv;        // [B][N][d]
  const float* Pq;        // [L][B][T][d]
  const float* C;         // [L][B][T][N]
  const float* dsv;       // [L][B][N]
  const float* dZq;       // [L][B][T][d]
  const float* wv;
  float* dPv;             // [L][B][N][d]
  float* dPq;             // [L][B][T][d]
  float* dA;              // [L][B][T][N]
  float* dwv_part;        // [L*B][d]
  int B, N, T, d, L;
};

// stage C (zero padded to kTRows x NPAD) and ds_v (zero padded) into LDS
template <int NPAD, int LD, int NTHREADS>
__device__ __forceinline__ void stage_c(const BwdArgs& a, size_t pair, float* Cbuf, float* dsvs, int tid) {
  const float* Cg = a.C + pair * (size_t)a.T * a.N;
  for (int e = tid; e < kTRows * NPAD; e += NTHREADS) {
    const int row = e / NPAD, col = e - row * NPAD;
    Cbuf[row * LD + col] = (row < a.T && col < a.N) ? Cg[(size_t)row * a.N + col] : 0.f;
  }
  const float* dg = a.dsv + pair * (size_t)a.N;
  for (int e = tid; e < NPAD; e += NTHREADS) dsvs[e] = (e < a.N) ? dg[e] : 0.f;
}

template <int NT, int NW>
__global__ __launch_bounds__(NW * 64, 2) void bwd_dc_kernel(const BwdArgs a) {
  constexpr int NPAD = 16 * NT;
  constexpr int LD = NPAD + 4;
  constexpr int NSLOT = (NW / 2 > 2) ? NW / 2 : 2;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* slots = lds;                                // NSLOT x kTRows x LD
  float* Cbuf = lds + NSLOT * kTRows * LD;           // kTRows x LD
  float* dsvs = Cbuf + kTRows * LD;                  // NPAD
  int b, l;
  if (!block_to_pair(blockIdx.x, a.B, a.L, b, l)) return;
  const int N = a.N, T = a.T, d = a.d;
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, q4 = lane >> 4;
  const size_t pair = (size_t)l * a.B + b;
  const float* Pvp = a.Pv + (size_t)b * N * d;
  const float* Pqp = a.Pq + pair * (size_t)T * d;
  const float* dZqp = a.dZq + pair * (size_t)T * d;
  float* dPvp = a.dPv + pair * (size_t)N * d;
  // rows n >= N fall outside these buffers: loads give 0, stores are dropped
  const __amdgpu_buffer_rsrc_t rs_pv = make_rsrc(Pvp, (unsigned)N * d * 4u);
  const __amdgpu_buffer_rsrc_t rs_dpv = make_rsrc(dPvp, (unsigned)N * d * 4u);
  const int voff = (j * d + 4 * q4) * 4;             // lane's row n = 16nt + j, channels db + 4q4..+3
  stage_c<NPAD, LD, NW * 64>(a, pair, Cbuf, dsvs, tid);
  __syncthreads();

  f32x4 acc[2][NT];
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[tt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int dt = 0; dt < 8; ++dt) {
    const int db = w * 128 + 16 * dt;
    // per-channel-tile operands
    float pqB[kTS], dzqB[kTS];                     // A[i = d = db + j][k = t = 4s + q4]
#pragma unroll
    for (int s = 0; s < kTS; ++s) {
      const int t = 4 * s + q4;
      pqB[s] = (t < T) ? Pqp[(size_t)t * d + db + j] : 0.f;
      dzqB[s] = (t < T) ? dZqp[(size_t)t * d + db + j] : 0.f;
    }
    f32x4 pqA[2], dzqA[2];                         // A[i = t = 16tt + j][k = d = db + 4*q4 + r]
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      const int t = 16 * tt + j;
      pqA[tt] = (t < T) ? *reinterpret_cast<const f32x4*>(Pqp + (size_t)t * d + db + 4 * q4) : zero4;
      dzqA[tt] = (t < T) ? *reinterpret_cast<const f32x4*>(dZqp + (size_t)t * d + db + 4 * q4) : zero4;
    }
    const f32x4 wv4 = *reinterpret_cast<const f32x4*>(a.wv + db + 4 * q4);
    f32x4 dwv4 = zero4;
    // transposed tiles, C/D layout: col = j <-> n, row = 4*q4 + r <-> channel db + 4*q4 + r
    auto load_pvT = [&](int nt) { return buf_load4(rs_pv, voff, (16 * nt * d + db) * 4); };
    f32x4 pvT_q[3];                                    // prefetch ring, two tiles ahead
    pvT_q[0] = load_pvT(0);
    if (NT > 1) pvT_q[1] = load_pvT(1);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      if (nt + 2 < NT) pvT_q[(nt + 2) % 3] = load_pvT(nt + 2);
      __builtin_amdgcn_sched_barrier(0);               // pin the prefetch ahead of this tile's math
      const f32x4 pvT = pvT_q[nt % 3];
      float cB[kTS];                                 // B[k = t][j = n]
#pragma unroll
      for (int s = 0; s < kTS; ++s) cB[s] = Cbuf[(4 * s + q4) * LD + 16 * nt + j];
      f32x4 hv = pvT;
#pragma unroll
      for (int s = 0; s < kTS; ++s) hv = mfma16(pqB[s], cB[s], hv);
      const float dsn = dsvs[16 * nt + j];
      f32x4 dzv;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float h = tanh_fast(hv[r]);
        dwv4[r] = fmaf(dsn, h, dwv4[r]);
        dzv[r] = dsn * wv4[r] * (1.0f - h * h);
      }
      // dP_v^T tile = dZ_v^T + dZ_q^T C
      f32x4 dpv = dzv;
#pragma unroll
      for (int s = 0; s < kTS; ++s) dpv = mfma16(dzqB[s], cB[s], dpv);
      buf_store4(dpv, rs_dpv, voff, (16 * nt * d + db) * 4);
      // dC[t][n] += sum_r P_q[t][db+4q4+r] dZ_v[n][..] + dZ_q[t][..] P_v[n][..]
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        acc[0][nt] = mfma16(pqA[0][r], dzv[r], acc[0][nt]);
        acc[1][nt] = mfma16(pqA[1][r], dzv[r], acc[1][nt]);
        acc[0][nt] = mfma16(dzqA[0][r], pvT[r], acc[0][nt]);
        acc[1][nt] = mfma16(dzqA[1][r], pvT[r], acc[1][nt]);
      }
      __builtin_amdgcn_sched_barrier(0);               // keep live ranges per tile (no cross-tile hoisting)
    }
    // dw_v[db + 4*q4 + r] partial of this (sample, level): sum over the 16 lanes (locations)
#pragma unroll
    for (int r = 0; r < 4; ++r) dwv4[r] = row16_sum(dwv4[r]);
    if (j == 0) *reinterpret_cast<f32x4*>(a.dwv_part + pair * (size_t)d + db + 4 * q4) = dwv4;
  }

  // cross-wave sum of dC in a fixed tree order, then dA = dC (1 - C^2)
  auto put = [&](float* slot) {
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * tt + 4 * q4 + r;
          if (row < kTRows) slot[row * LD + 16 * t + j] = acc[tt][t][r];
        }
  };
  auto add = [&](const float* slot) {
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * tt + 4 * q4 + r;
          if (row < kTRows) acc[tt][t][r] += slot[row * LD + 16 * t + j];
        }
  };
#pragma unroll
  for (int stride = 1; stride < NW / 2; stride <<= 1) {
    const int m = 2 * stride - 1;
    if (stride > 1) __syncthreads();
    if ((w & m) == stride) put(slots + (w / (2 * stride)) * kTRows * LD);
    __syncthreads();
    if ((w & m) == 0) add(slots + (w / (2 * stride)) * kTRows * LD);
  }
  if (NW > 2) __syncthreads();
  if (w == NW / 2) put(slots);
  if (w == 0) put(slots + kTRows * LD);
  __syncthreads();
  float* dAg = a.dA + pair * (size_t)T * N;
  for (int e = tid; e < T * NPAD; e += NW * 64) {
    const int row = e / NPAD, col = e - row * NPAD;
    if (col < N) {
      const float c = Cbuf[row * LD + col];
      dAg[(size_t)row * N + col] = (slots[row * LD + col] + slots[kTRows * LD + row * LD + col]) * (1.0f - c * c);
    }
  }
}

template <int NT, int NW>
__global__ __launch_bounds__(NW * 64, 2) void bwd_dpq_kernel(const BwdArgs a) {
  constexpr int NPAD = 16 * NT;
  constexpr int LD = NPAD + 4;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Cbuf = lds;                                 // kTRows x LD
  float* dsvs = Cbuf + kTRows * LD;                  // NPAD
  int b, l;
  if (!block_to_pair(blockIdx.x, a.B, a.L, b, l)) return;
  const int N = a.N, T = a.T, d = a.d;
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, q4 = lane >> 4;
  const size_t pair = (size_t)l * a.B + b;
  const float* Pvp = a.Pv + (size_t)b * N * d;
  const float* Pqp = a.Pq + pair * (size_t)T * d;
  const float* dZqp = a.dZq + pair * (size_t)T * d;
  stage_c<NPAD, LD, NW * 64>(a, pair, Cbuf, dsvs, tid);
  const int dsl = w * 128;
  float pq[kTS][8];
#pragma unroll
  for (int s = 0; s < kTS; ++s) {
    const int t = 4 * s + q4;
#pragma unroll
    for (int c = 0; c < 8; ++c) pq[s][c] = (t < T) ? Pqp[(size_t)t * d + dsl + 16 * c + j] : 0.f;
  }
  float wvr[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) wvr[c] = a.wv[dsl + 16 * c + j];
  f32x4 accq[2][8];
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int c = 0; c < 8; ++c) accq[tt][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();

  const int ntiles = (N + 15) >> 4;
  {
    const __amdgpu_buffer_rsrc_t rs_pv = make_rsrc(Pvp, (unsigned)N * d * 4u);
    f32x4 pvA[4], pvB[4];
    float sv[4] = {0.f, 0.f, 0.f, 0.f};
    load_pv_half<0>(rs_pv, d, dsl, 0, j, q4, pvA);
    for (int tile = 0; tile < ntiles; ++tile) {
      const int nb = 16 * tile;
      load_pv_half<1>(rs_pv, d, dsl, nb, j, q4, pvB);
      float dsn[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) dsn[r] = dsvs[nb + 4 * q4 + r];
      __builtin_amdgcn_sched_barrier(0);
      half_unit<true, 0, LD>(pvA, pq, wvr, accq, Cbuf, nb, j, q4, sv, dsn);
      load_pv_half<0>(rs_pv, d, dsl, nb + 16, j, q4, pvA);
      __builtin_amdgcn_sched_barrier(0);
      half_unit<true, 1, LD>(pvB, pq, wvr, accq, Cbuf, nb, j, q4, sv, dsn);
    }
  }
  // dP_q = dZ_q + acc
  float* dPqp = a.dPq + pair * (size_t)T * d;
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int t = 16 * tt + 4 * q4 + r;
      if (t < T) {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const size_t o = (size_t)t * d + dsl + 16 * c + j;
          dPqp[o] = accq[tt][c][r] + dZqp[o];
        }
      }
    }
}

template <typename K>
void set_lds(K kern, size_t bytes) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

template <int NT>
int launch_pre(const PreArgs& a, hipStream_t s) {
  const size_t lds = (size_t)(6 * a.d + 16 * 3 * 16 * NT + 3 * 16 * NT + 96 + 96 + 8) * sizeof(float);
  static bool once = false;
  if (!once) { set_lds(bwd_pre_kernel<NT>, lds); once = true; }
  hipLaunchKernelGGL(bwd_pre_kernel<NT>, dim3(a.B), dim3(256), lds, s, a);
  CA_CHECK_LAUNCH("bwd_pre");
  return 0;
}

template <int NT, int NW>
int launch_main(const BwdArgs& a, hipStream_t s) {
  constexpr int LD = 16 * NT + 4;
  constexpr int NSLOT = (NW / 2 > 2) ? NW / 2 : 2;
  const size_t lds_dc = (size_t)((NSLOT + 1) * kTRows * LD + 16 * NT) * sizeof(float);
  const size_t lds_pq = (size_t)(kTRows * LD + 16 * NT) * sizeof(float);
  static bool once = false;
  if (!once) {
    set_lds(bwd_dc_kernel<NT, NW>, lds_dc);
    set_lds(bwd_dpq_kernel<NT, NW>, lds_pq);
    once = true;
  }
  const int groups = (a.B + 7) / 8;
  dim3 grid(groups * a.L * 8), block(NW * 64);
  hipLaunchKernelGGL((bwd_dc_kernel<NT, NW>), grid, block, lds_dc, s, a);
  CA_CHECK_LAUNCH("bwd_dc");
  hipLaunchKernelGGL((bwd_dpq_kernel<NT, NW>), grid, block, lds_pq, s, a);
  CA_CHECK_LAUNCH("bwd_dpq");
  return 0;
}

}  // namespace

size_t fused_bwd_ws_floats(int B, int N, int T, int d, int L) { return fused_bwd_off(B, N, T, d, L).total; }

int fused_backward_supported(int B, int N, int T, int d, int L) { return fused_supported(B, N, T, d, L); }

int fused_backward(int B, int N, int T, int d, int L, const float* V, const float* const* Q, const coattn_params* p,
                   const float* saved, const float* gv, const float* gq, float* dV, float* const* dQ,
                   const coattn_param_grads* pg, int accumulate, float* ws, hipStream_t s) {
  CA_CHECK_ARG(fused_backward_supported(B, N, T, d, L), "fused backward: unsupported shape");
  const SavedOff so = saved_off(B, N, T, d, L);
  const FusedBwdOff wo = fused_bwd_off(B, N, T, d, L);
  const size_t BTd = (size_t)B * T * d, BTN = (size_t)B * T * N, BNd = (size_t)B * N * d, Bd = (size_t)B * d;
  const bool small_n = N <= 64;
  // 1. per-sample pre-pass
  PreArgs pa;
  pa.V = V;
  for (int l = 0; l < 8; ++l) pa.Q[l] = l < L ? Q[l] : nullptr;
  pa.gv = gv; pa.gq = gq; pa.av = saved + so.av; pa.aq = saved + so.aq; pa.Hq = saved + so.Hq;
  pa.wq = (const float*)p->w_q;
  pa.dsv = ws + wo.dsv; pa.dZq = ws + wo.dZq; pa.dwq_part = ws + wo.dwq_part; pa.dcs_part = ws + wo.dcs_part;
  pa.B = B; pa.N = N; pa.T = T; pa.d = d; pa.L = L;
  CA_TRY(small_n ? launch_pre<4>(pa, s) : launch_pre<13>(pa, s));
  // 2. the two recompute kernels
  BwdArgs ba;
  ba.Pv = saved + so.Pv; ba.Pq = saved + so.Pq; ba.C = saved + so.C; ba.dsv = ws + wo.dsv; ba.dZq = ws + wo.dZq;
  ba.wv = (const float*)p->w_v;
  ba.dPv = ws + wo.dPv; ba.dPq = ws + wo.dPq; ba.dA = ws + wo.dA; ba.dwv_part = ws + wo.dwv_part;
  ba.B = B; ba.N = N; ba.T = T; ba.d = d; ba.L = L;
  if (d == 512) {
    CA_TRY(small_n ? (launch_main<4, 4>(ba, s)) : (launch_main<13, 4>(ba, s)));
  } else {
    CA_TRY(small_n ? (launch_main<4, 2>(ba, s)) : (launch_main<13, 2>(ba, s)));
  }
  // 3. small parameter gradients from the partials
  CA_TRY(launch_reduce_partials(ws + wo.dwv_part, (float*)pg->dw_v, L * B, d, accumulate, s));
  CA_TRY(launch_reduce_partials(ws + wo.dwq_part, (float*)pg->dw_q, B, d, accumulate, s));
  {
    // dc_v, dc_q: column sums of dcs_part [B][2]
    int nch = 0;
    float* part = ws + wo.part;
    CA_TRY(launch_colsum_partial(nullptr, ws + wo.dcs_part, part, B, 2, B, &nch, s));
    CA_TRY(launch_reduce_partials(part, (float*)pg->dc_v, 1, 1, accumulate, s));
    CA_TRY(launch_reduce_partials(part + 1, (float*)pg->dc_q, 1, 1, accumulate, s));
  }
  // 4. dQ_l = a_q (x) gq + dA V^T + dP_q W_q ;  dV = sum_l (a_v (x) gv + Q^T dA) + (sum_l dP_v) W_v
  for (int l = 0; l < L; ++l) {
    const float* dA = ws + wo.dA + l * BTN;
    const float* aq = saved + so.aq + (size_t)l * B * T;
    const float* av = saved + so.av + (size_t)l * B * N;
    CA_TRY(launch_rank1(aq, gq + l * Bd, dQ[l], B, T, d, (int64_t)T * d, d, 1, 0, s));
    {
      coattn_gemm_desc g = {};
      g.A = dA; g.a_sz = (int64_t)T * N; g.a_sm = N; g.a_sk = 1;
      g.B = V; g.b_sz = (int64_t)d * N; g.b_sk = 1; g.b_sn = N;
      g.Cin = dQ[l]; g.cin_sz = (int64_t)T * d; g.cin_sm = d; g.cin_sn = 1; g.beta = 1.f;
      g.C = dQ[l]; g.c_sz = (int64_t)T * d; g.c_sm = d; g.c_sn = 1;
      g.M = T; g.N = d; g.K = N; g.batch = B;
      CA_TRY(launch_gemm_f32(g, s));
    }
    {
      coattn_gemm_desc g = {};
      g.A = ws + wo.dPq + l * BTd; g.a_sm = d; g.a_sk = 1;
      g.B = p->W_q; g.b_sk = d; g.b_sn = 1;
      g.Cin = dQ[l]; g.cin_sm = d; g.cin_sn = 1; g.beta = 1.f;
      g.C = dQ[l]; g.c_sm = d; g.c_sn = 1;
      g.M = B * T; g.N = d; g.K = d; g.batch = 1;
      CA_TRY(launch_gemm_f32(g, s));
    }
    if (dV) {
      CA_TRY(launch_rank1(av, gv + l * Bd, dV, B, N, d, (int64_t)d * N, 1, N, l > 0 ? 1 : 0, s));
      coattn_gemm_desc g = {};
      g.A = Q[l]; g.a_sz = (int64_t)T * d; g.a_sm = 1; g.a_sk = d;
      g.B = dA; g.b_sz = (int64_t)T * N; g.b_sk = N; g.b_sn = 1;
      g.Cin = dV; g.cin_sz = (int64_t)d * N; g.cin_sm = N; g.cin_sn = 1; g.beta = 1.f;
      g.C = dV; g.c_sz = (int64_t)d * N; g.c_sm = N; g.c_sn = 1;
      g.M = d; g.N = N; g.K = T; g.batch = B;
      CA_TRY(launch_gemm_f32(g, s));
    }
  }
  // sum dP_v over the levels (in place into level 0)
  float* dPv = ws + wo.dPv;
  for (int l = 1; l < L; ++l) CA_TRY(launch_add_inplace(dPv, dPv + l * BNd, (int64_t)BNd, 1, s));
  if (dV) {
    coattn_gemm_desc g = {};
    g.A = p->W_v; g.a_sm = 1; g.a_sk = d; g.a_sz = 0;
    g.B = dPv; g.b_sz = (int64_t)N * d; g.b_sk = 1; g.b_sn = d;
    g.Cin = dV; g.cin_sz = (int64_t)d * N; g.cin_sm = N; g.cin_sn = 1; g.beta = 1.f;
    g.C = dV; g.c_sz = (int64_t)d * N; g.c_sm = N; g.c_sn = 1;
    g.M = d; g.N = N; g.K = d; g.batch = B;
    CA_TRY(launch_gemm_f32(g, s));
  }
  // 5. weight gradients
  float* part = ws + wo.part;
  int nch = 0;
  {
    const int G = (B + 31) / 32;
    const int S = (B + G - 1) / G;
    coattn_gemm_desc g = {};
    g.A = dPv; g.a_sm = 1; g.a_sk = d; g.a_si = (int64_t)N * d; g.a_sz = (int64_t)G * N * d;
    g.B = V; g.b_sk = 1; g.b_sn = N; g.b_si = (int64_t)d * N; g.b_sz = (int64_t)G * d * N;
    g.C = part; g.c_sz = (int64_t)d * d; g.c_sm = d; g.c_sn = 1;
    g.M = d; g.N = d; g.K = N; g.batch = S; g.inner = G; g.inner_total = B;
    CA_TRY(launch_gemm_f32(g, s));
    CA_TRY(launch_reduce_partials(part, (float*)pg->dW_v, S, (int64_t)d * d, accumulate, s));
    const int rpc = (B * N + 255) / 256 > 32 ? (B * N + 255) / 256 : 32;
    CA_TRY(launch_colsum_partial(nullptr, dPv, part, B * N, d, rpc, &nch, s));
    CA_TRY(launch_reduce_partials(part, (float*)pg->db_v, nch, d, accumulate, s));
  }
  for (int l = 0; l < L; ++l) {
    const int K = B * T;
    int ks = (K + 31) / 32;
    ks = (ks + 15) / 16 * 16;
    const int S = (K + ks - 1) / ks;
    coattn_gemm_desc g = {};
    g.A = ws + wo.dPq + l * BTd; g.a_sm = 1; g.a_sk = d;
    g.B = Q[l]; g.b_sk = d; g.b_sn = 1;
    g.C = part; g.c_sz = (int64_t)d * d; g.c_sm = d; g.c_sn = 1;
    g.M = d; g.N = d; g.K = K; g.batch = S; g.ksplit = ks;
    CA_TRY(launch_gemm_f32(g, s));
    CA_TRY(launch_reduce_partials(part, (float*)pg->dW_q, S, (int64_t)d * d, (accumulate || l > 0) ? 1 : 0, s));
  }
  {
    const int R = L * B * T;
    const int rpc = (R + 255) / 256 > 32 ? (R + 255) / 256 : 32;
    CA_TRY(launch_colsum_partial(nullptr, ws + wo.dPq, part, R, d, rpc, &nch, s));
    CA_TRY(launch_reduce_partials(part, (float*)pg->db_q, nch, d, accumulate, s));
  }
  return 0;
}
