// Strided, batched fp32 GEMM on the gfx950 f32 MFMA (v_mfma_f32_32x32x2_f32).
//
// General-shape workhorse of the co-attention path: projections P_v = V W_v^T + b_v and
// P_q = Q W_q^T + b_q (model.py:380-384), the per-sample affinity / H_v / H_q contractions of
// the general-shape implementation, and every gradient GEMM of the backward.  Operands are
// addressed through element strides so that V is consumed in its physical channel-major
// [B,d,N] layout (model.py:215-217) without a transpose pass.
//
// Tile: BM x 128 per 256-thread workgroup (4 waves), BK = 16, operands staged through LDS as
// As[k][m] / Bs[k][n] so that an MFMA operand read is 32 consecutive floats per half wave
// (conflict-free ds_read_b32).  Numerics: exact fp32 fmaf chain in k order (MFMA f32).
#include "common.h"

namespace {

struct GemmK {
  const float* A; const float* B; const float* Cin; float* C;
  const float* bias_n; const float* bias_m;
  int M, N, K, inner, inner_total, ksplit, act;
  float beta;
  int a_mfast, b_nfast;
  long a_sm, a_sk, a_sz, a_si, a_mdiv, a_sdiv;
  long b_sk, b_sn, b_sz, b_si;
  long c_sm, c_sn, c_sz, c_mdiv, c_sdiv;
  long cin_sm, cin_sn, cin_sz, cin_mdiv, cin_sdiv;
};

__device__ __forceinline__ long row_off(long m, long sm, long mdiv, long sdiv) {
  return mdiv > 0 ? (m / mdiv) * sdiv + (m % mdiv) * sm : m * sm;
}

template <int BM>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmK g) {
  constexpr int BN = 128, BK = 16;
  constexpr int LDA = BM + 4, LDB = BN + 4;
  constexpr int WM = (BM >= 128) ? 2 : 1;  // waves along m
  constexpr int WN = 4 / WM;               // waves along n
  constexpr int TM = BM / WM / 32;         // 32x32 tiles per wave along m
  constexpr int TN = BN / WN / 32;
  __shared__ float As[BK * LDA];
  __shared__ float Bs[BK * LDB];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave / WN, wc = wave % WN;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN, z = blockIdx.z;

  int kbeg = 0, kend = g.K;
  if (g.ksplit > 0) {
    kbeg = z * g.ksplit;
    kend = min(g.K, kbeg + g.ksplit);
  }
  int ninner = g.inner;
  if (g.inner_total > 0) ninner = max(0, min(g.inner, g.inner_total - z * g.inner));

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  for (int ii = 0; ii < ninner; ++ii) {
    const float* Ab = g.A + (long)z * g.a_sz + (long)ii * g.a_si;
    const float* Bb = g.B + (long)z * g.b_sz + (long)ii * g.b_si;
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
#pragma unroll
      for (int i = 0; i < BM * BK / 256; ++i) {
        const int idx = tid + i * 256;
        int m, k;
        if (g.a_mfast) { m = idx % BM; k = idx / BM; } else { k = idx % BK; m = idx / BK; }
        const int gm = m0 + m, gk = k0 + k;
        float v = 0.f;
        if (gm < g.M && gk < kend) v = Ab[row_off(gm, g.a_sm, g.a_mdiv, g.a_sdiv) + (long)gk * g.a_sk];
        As[k * LDA + m] = v;
      }
#pragma unroll
      for (int i = 0; i < BN * BK / 256; ++i) {
        const int idx = tid + i * 256;
        int n, k;
        if (g.b_nfast) { n = idx % BN; k = idx / BN; } else { k = idx % BK; n = idx / BK; }
        const int gn = n0 + n, gk = k0 + k;
        float v = 0.f;
        if (gn < g.N && gk < kend) v = Bb[(long)gk * g.b_sk + (long)gn * g.b_sn];
        Bs[k * LDB + n] = v;
      }
      __syncthreads();
#pragma unroll
      for (int kk = 0; kk < BK; kk += 2) {
        float a[TM], b[TN];
        const int krow = kk + (lane >> 5);
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = As[krow * LDA + wr * (TM * 32) + i * 32 + (lane & 31)];
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = Bs[krow * LDB + wc * (TN * 32) + j * 32 + (lane & 31)];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
      }
      __syncthreads();
    }
  }

  // epilogue: C/D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  float* Cb = g.C + (long)z * g.c_sz;
  const float* Cinb = g.Cin ? g.Cin + (long)z * g.cin_sz : nullptr;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wc * (TN * 32) + j * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wr * (TM * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < g.M && col < g.N) {
          float v = acc[i][j][r];
          if (g.bias_n) v += g.bias_n[col];
          if (g.bias_m) v += g.bias_m[row];
          if (Cinb) v += g.beta * Cinb[row_off(row, g.cin_sm, g.cin_mdiv, g.cin_sdiv) + (long)col * g.cin_sn];
          if (g.act == 1) v = tanhf(v);
          Cb[row_off(row, g.c_sm, g.c_mdiv, g.c_sdiv) + (long)col * g.c_sn] = v;
        }
      }
    }
}

}  // namespace

int launch_gemm_f32(const coattn_gemm_desc& d, hipStream_t s) {
  CA_CHECK_ARG(d.A && d.B && d.C, "gemm: null operand");
  CA_CHECK_ARG(d.M > 0 && d.N > 0 && d.K > 0 && d.batch > 0, "gemm: bad shape M=%d N=%d K=%d batch=%d", d.M, d.N, d.K, d.batch);
  CA_CHECK_ARG(d.batch <= 65535, "gemm: batch %d exceeds grid.z", d.batch);
  GemmK g;
  g.A = (const float*)d.A; g.B = (const float*)d.B; g.Cin = (const float*)d.Cin; g.C = (float*)d.C;
  g.bias_n = (const float*)d.bias_n; g.bias_m = (const float*)d.bias_m;
  g.M = d.M; g.N = d.N; g.K = d.K; g.inner = d.inner > 0 ? d.inner : 1; g.inner_total = d.inner_total;
  g.ksplit = d.ksplit; g.act = d.act; g.beta = d.beta;
  g.a_sm = d.a_sm; g.a_sk = d.a_sk; g.a_sz = d.a_sz; g.a_si = d.a_si; g.a_mdiv = d.a_mdiv; g.a_sdiv = d.a_sdiv;
  g.b_sk = d.b_sk; g.b_sn = d.b_sn; g.b_sz = d.b_sz; g.b_si = d.b_si;
  g.c_sm = d.c_sm; g.c_sn = d.c_sn; g.c_sz = d.c_sz; g.c_mdiv = d.c_mdiv; g.c_sdiv = d.c_sdiv;
  g.cin_sm = d.cin_sm; g.cin_sn = d.cin_sn; g.cin_sz = d.cin_sz; g.cin_mdiv = d.cin_mdiv; g.cin_sdiv = d.cin_sdiv;
  // global-load thread mapping: run consecutive threads along the contiguous operand axis
  g.a_mfast = (d.a_sm == 1 && d.a_sk != 1) ? 1 : 0;
  g.b_nfast = (d.b_sn == 1) ? 1 : 0;
  const bool small_m = d.M <= 64;
  dim3 block(256);
  if (small_m) {
    dim3 grid((d.N + 127) / 128, (d.M + 31) / 32, d.batch);
    hipLaunchKernelGGL(gemm_f32_kernel<32>, grid, block, 0, s, g);
  } else {
    dim3 grid((d.N + 127) / 128, (d.M + 127) / 128, d.batch);
    hipLaunchKernelGGL(gemm_f32_kernel<128>, grid, block, 0, s, g);
  }
  CA_CHECK_LAUNCH("gemm_f32");
  return 0;
}
