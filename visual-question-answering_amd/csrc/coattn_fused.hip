// Fused co-attention forward (model.py:377-392 after the projections): dispatch.
//
// The affinity + softmax + reduce kernel itself lives in coattn_fwd32.hip (every contraction on the bf16 MFMA
// 32x32x16 with the exact 3-way split; channel-major and location-major image features).  This file holds the
// entry point, the shape limits, and the attended-image-feature pass of the channel-major layout:
//   attend_v_kernel : v_l = a_{v,l}^T V for all levels with ONE more pass over a channel-major V [d][N]
//   (the location-major twin is attend_v_lm_kernel in coattn_fwd32.hip).
#include "fused.h"
#include <stdlib.h>


namespace {

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

}  // namespace

int fused_supported(int B, int N, int T, int d, int L) {
  (void)B;
  if (d <= 0 || d % 256 != 0 || d > 4096) return 0;   // NW = 4 waves when d % 512 == 0, else 2; slices of 128 channels
  if (T > kTRows || N > 208 || L > 3) return 0;
  return 1;
}

int fused_attention_forward(int B, int N, int T, int d, int L, const float* V, const VLayout& vl,
                            const float* const* Q, const coattn_params* p, float* v_out, float* q_out, float* saved,
                            float* ws, hipStream_t s, int bf16, int np) {
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
  // Small grids (the 7 x 7 grid of 224 x 224 images: N = 49) on location-major features: the attended image feature
  // v_l = a_v^T V is computed by the affinity kernel's own workgroup -- its 100 KB of V come from L2, where phase 1 left
  // them, in less time than a second launch costs.  (At N = 196 the separate pass over V stays: DESIGN.md section 3.1.)
  static const int fuse_v_env = dev_env_int("COATTN_FUSE_V", 1);   // developer switch
  const bool fuse_v = lm && N <= 64 && d % 512 == 0 && fuse_v_env;
  a.v_out = fuse_v ? v_out : nullptr;
  a.stamps = COATTN_STAMPS ? reinterpret_cast<unsigned long long*>(ws) : nullptr;
  a.B = B; a.N = N; a.T = T; a.d = d; a.L = L;
  a.bf16 = bf16;
  a.np = ((np == 2 || np == 4) && !bf16) ? np : 3;      // 4: both phases on two FP16 pieces (coattn_fwd32.hip)
  CA_TRY(fused32_forward(a, s));
  prof_mark(s, "coattn_fwd32");
  if (fuse_v) return 0;
  if (lm) {
    CA_TRY(launch_attend_v_lm(V, vl.sB, a.av, v_out, B, N, d, L, s));
    prof_mark(s, "attend_v");
    return 0;
  }
  dim3 grid(d / 64, B);
  if (N <= 64)
    hipLaunchKernelGGL(attend_v_kernel<4>, grid, dim3(256), 0, s, V, vl.sB, a.av, v_out, B, N, d, L);
  else
    hipLaunchKernelGGL(attend_v_kernel<13>, grid, dim3(256), 0, s, V, vl.sB, a.av, v_out, B, N, d, L);
  CA_CHECK_LAUNCH("attend_v");
  prof_mark(s, "attend_v");
  return 0;
}
