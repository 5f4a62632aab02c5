// Strided, batched fp32 GEMM on the gfx950 f32 MFMA (v_mfma_f32_32x32x2_f32).
//
// Workhorse of the co-attention path outside the fused kernels: projections
// P_v = V W_v^T + b_v and P_q = Q W_q^T + b_q (model.py:380-384), the projection / weight
// gradients of the backward, and every contraction of the general-shape implementation.
// Operands are addressed through element strides so that V is consumed in its physical
// channel-major [B,d,N] layout (model.py:215-217) without a transpose pass.
//
// Tile BM x 128 per 256-thread workgroup (4 waves), BK = 16.  Software pipeline: the global
// loads of K-step s+1 are issued into registers before the MFMAs of step s and written to the
// other LDS buffer afterwards (one barrier per step).  Per-thread element offsets (including
// the optional row split m -> (m / mdiv, m % mdiv)) are hoisted out of the K loop as 32-bit
// offsets.  LDS images are As[k][m] / Bs[k][n]: an MFMA operand read is 32 consecutive floats
// per half wave (conflict-free ds_read_b32).  Numerics: exact fp32 fmaf chain in k order.
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
  constexpr int EA = BM * BK / 256;        // A elements per thread per K-step
  constexpr int EB = BN * BK / 256;
  __shared__ float As[2][BK * LDA];
  __shared__ float Bs[2][BK * LDB];

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
  const int ksteps = (kend - kbeg + BK - 1) / BK;
  const int nsteps = ninner * max(ksteps, 0);

  // hoisted per-thread element coordinates: tile-local (m,k) / (k,n), 32-bit global offsets
  int a_off[EA], a_lds[EA], a_k[EA];
  bool a_ok[EA];
#pragma unroll
  for (int i = 0; i < EA; ++i) {
    const int idx = tid + i * 256;
    int m, k;
    if (g.a_mfast) { m = idx % BM; k = idx / BM; } else { k = idx % BK; m = idx / BK; }
    a_k[i] = k;
    a_lds[i] = k * LDA + m;
    a_ok[i] = (m0 + m) < g.M;
    a_off[i] = a_ok[i] ? (int)(row_off(m0 + m, g.a_sm, g.a_mdiv, g.a_sdiv) + (long)k * g.a_sk) : 0;
  }
  int b_off[EB], b_lds[EB], b_k[EB];
  bool b_ok[EB];
#pragma unroll
  for (int i = 0; i < EB; ++i) {
    const int idx = tid + i * 256;
    int n, k;
    if (g.b_nfast) { n = idx % BN; k = idx / BN; } else { k = idx % BK; n = idx / BK; }
    b_k[i] = k;
    b_lds[i] = k * LDB + n;
    b_ok[i] = (n0 + n) < g.N;
    b_off[i] = b_ok[i] ? (int)((long)k * g.b_sk + (long)(n0 + n) * g.b_sn) : 0;
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float ra[EA], rb[EB];
  // step -> (inner index, k0)
  auto load_step = [&](int step) {
    const int ii = step / ksteps, k0 = kbeg + (step - ii * ksteps) * BK;
    const float* Ab = g.A + (long)z * g.a_sz + (long)ii * g.a_si + (long)k0 * g.a_sk;
    const float* Bb = g.B + (long)z * g.b_sz + (long)ii * g.b_si + (long)k0 * g.b_sk;
    if (k0 + BK <= kend) {
#pragma unroll
      for (int i = 0; i < EA; ++i) ra[i] = a_ok[i] ? Ab[a_off[i]] : 0.f;
#pragma unroll
      for (int i = 0; i < EB; ++i) rb[i] = b_ok[i] ? Bb[b_off[i]] : 0.f;
    } else {
      const int klim = kend - k0;
#pragma unroll
      for (int i = 0; i < EA; ++i) ra[i] = (a_ok[i] && a_k[i] < klim) ? Ab[a_off[i]] : 0.f;
#pragma unroll
      for (int i = 0; i < EB; ++i) rb[i] = (b_ok[i] && b_k[i] < klim) ? Bb[b_off[i]] : 0.f;
    }
  };
  auto store_step = [&](int buf) {
#pragma unroll
    for (int i = 0; i < EA; ++i) As[buf][a_lds[i]] = ra[i];
#pragma unroll
    for (int i = 0; i < EB; ++i) Bs[buf][b_lds[i]] = rb[i];
  };

  if (nsteps > 0) {
    load_step(0);
    store_step(0);
  }
  __syncthreads();
  for (int step = 0; step < nsteps; ++step) {
    const int buf = step & 1;
    if (step + 1 < nsteps) load_step(step + 1);
    const float* as = As[buf];
    const float* bs = Bs[buf];
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      float a[TM], b[TN];
      const int krow = kk + (lane >> 5);
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = as[krow * LDA + wr * (TM * 32) + i * 32 + (lane & 31)];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = bs[krow * LDB + wc * (TN * 32) + j * 32 + (lane & 31)];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (step + 1 < nsteps) store_step(buf ^ 1);
    __syncthreads();
  }

  // epilogue: C/D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  float* Cb = g.C + (long)z * g.c_sz;
  const float* Cinb = g.Cin ? g.Cin + (long)z * g.cin_sz : nullptr;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wc * (TN * 32) + j * 32 + (lane & 31);
      const float bn = (g.bias_n && col < g.N) ? g.bias_n[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wr * (TM * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < g.M && col < g.N) {
          float v = acc[i][j][r] + bn;
          if (g.bias_m) v += g.bias_m[row];
          if (Cinb) v += g.beta * Cinb[row_off(row, g.cin_sm, g.cin_mdiv, g.cin_sdiv) + (long)col * g.cin_sn];
          if (g.act == 1) v = tanhf(v);
          Cb[row_off(row, g.c_sm, g.c_mdiv, g.c_sdiv) + (long)col * g.c_sn] = v;
        }
      }
    }
}

}  // namespace

static long span(long n, long s) { return n > 0 ? (n - 1) * (s < 0 ? -s : s) : 0; }

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
  // the kernel keeps tile-relative element offsets in 32 bits
  const long a_rows = d.a_mdiv > 0 ? span((d.M + d.a_mdiv - 1) / d.a_mdiv + 1, d.a_sdiv) + span(d.a_mdiv, d.a_sm)
                                   : span(d.M, d.a_sm);
  CA_CHECK_ARG(a_rows + span(16, d.a_sk) < 2147483647L && span(d.N, d.b_sn) + span(16, d.b_sk) < 2147483647L,
               "gemm: operand extent exceeds 32-bit element offsets");
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
