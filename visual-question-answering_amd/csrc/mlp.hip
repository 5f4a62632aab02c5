// MLPClassifier (reference model.py:400-434) and the CrossEntropyLoss of the train step (main.py:214) on
// gfx950: the consumer of the co-attention outputs.
//
//   h_w = tanh(W_w (q_w + v_w) + b_w)
//   h_p = tanh(W_p [q_p + v_p | h_w] + b_p)
//   h_s = tanh(W_s [q_s + v_s | h_p] + b_s)
//   logits = W_h h_s + b_h
//
// v, q arrive as the [3, B, d] buffers coattn_forward writes.  No add, concat or tanh pass exists: a layer is
// ONE launch of the MFMA GEMM of gemm.hip whose `inner` dimension sums the products of the pieces,
//   [q_l + v_l | h] W^T = q_l W[:, :d]^T + v_l W[:, :d]^T + h W[:, d:]^T          (pointer tables, 3 pieces)
// with bias and tanh in its epilogue.  Backward: per layer one pass dz = dh (1 - h^2) fused with the bias
// gradient (column sums over the B rows), the weight gradients dW = dz^T [x | h] by the same `inner` trick
// (dz^T q_l + dz^T v_l), and dx / dh = dz W.  The gradient with respect to q_l and v_l is the same tensor
// (they enter through their sum) and is written once, [3, B, d].
// Cross entropy: one workgroup per row (log-sum-exp, loss, and d logits = (softmax - onehot) / B in one pass),
// mean over the batch like nn.CrossEntropyLoss().  Deterministic: no atomics.
#include "common.h"

namespace {

// dz = dh * (1 - h^2), db[j] (+)= sum_r dz[r][j].  h == nullptr: dz = dh (nothing written), only the column sums.
__global__ __launch_bounds__(256) void mlp_dz_colsum_kernel(const float* dh, const float* __restrict__ h,
                                                            float* dz, float* __restrict__ db, int R,
                                                            int n, int accumulate) {
  __shared__ float part[4][64];
  const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  float s = 0.f;
  if (c < n) {
    for (int r = rg; r < R; r += 4) {
      float x = dh[(long)r * n + c];
      if (h) {
        const float o = h[(long)r * n + c];
        x *= 1.f - o * o;
        dz[(long)r * n + c] = x;
      }
      s += x;
    }
  }
  part[rg][cl] = s;
  __syncthreads();
  if (rg == 0 && c < n && db) {
    const float t = (part[0][cl] + part[1][cl]) + (part[2][cl] + part[3][cl]);
    db[c] = accumulate ? db[c] + t : t;
  }
}

__device__ __forceinline__ float block_max(float v, float* sh) {
  v = wave_max(v);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  const float r = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
  __syncthreads();
  return r;
}
__device__ __forceinline__ float block_sum(float v, float* sh) {
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  const float r = (sh[0] + sh[1]) + (sh[2] + sh[3]);
  __syncthreads();
  return r;
}

// row_loss[i] = (logsumexp(z_i) - z_i[label_i]) * inv_b ; dlogits = (softmax(z_i) - onehot(label_i)) * inv_b.
// A label outside [0, K) makes the row's loss NaN (nn.CrossEntropyLoss raises there).
__global__ __launch_bounds__(256) void ce_rows_kernel(const float* __restrict__ logits, const long long* __restrict__ labels,
                                                      float* __restrict__ row_loss, float* __restrict__ dlogits, int K,
                                                      float inv_b) {
  __shared__ float sh[4];
  const int i = blockIdx.x;
  const float* z = logits + (long)i * K;
  float m = -INFINITY;
  for (int k = threadIdx.x; k < K; k += 256) m = fmaxf(m, z[k]);
  m = block_max(m, sh);
  float s = 0.f;
  for (int k = threadIdx.x; k < K; k += 256) s += expf(z[k] - m);
  s = block_sum(s, sh);
  const long long lab = labels[i];
  const bool ok = lab >= 0 && lab < K;
  if (threadIdx.x == 0) row_loss[i] = ok ? (logf(s) + m - z[lab]) * inv_b : NAN;
  if (dlogits) {
    const float inv = inv_b / s;
    for (int k = threadIdx.x; k < K; k += 256)
      dlogits[(long)i * K + k] = expf(z[k] - m) * inv - ((ok && k == lab) ? inv_b : 0.f);
  }
}

// out[i] = act( sum_s part[s][i] + bias[i % n] )   (fixed summation order; bias may be NULL)
__global__ __launch_bounds__(256) void mlp_reduce_kernel(const float* __restrict__ part, int nparts, long mn, int n,
                                                         const float* __restrict__ bias, int act,
                                                         float* __restrict__ out) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= mn) return;
  float v = 0.f;
  for (int s = 0; s < nparts; ++s) v += part[(long)s * mn + i];
  if (bias) v += bias[i % n];
  if (act == 1) v = tanhf(v);
  out[i] = v;
}

inline size_t al64(size_t n) { return (n + 63) & ~(size_t)63; }

// The products of the head have only B (= 160) rows: a 128-row tile grid would be 8 .. 16 workgroups walking
// the whole contraction.  They are split over k instead (64 .. 256 workgroups), partials summed in a fixed order
// by mlp_reduce_kernel together with bias and tanh.
constexpr int kMaxSplit = 16;
int split_len(int K) {
  int ks = 64;
  if ((K + ks - 1) / ks > kMaxSplit) ks = ((K + kMaxSplit - 1) / kMaxSplit + 31) / 32 * 32;
  return ks;
}
size_t part_floats(int B, int d, int mlp, int K) {
  size_t n = (size_t)(d > mlp ? d : mlp);
  if ((size_t)K > n) n = (size_t)K;
  return al64((size_t)kMaxSplit * B * n);
}
int skinny_gemm(coattn_gemm_desc g, float* part, const float* bias, int act, float* out, hipStream_t s) {
  const int ks = split_len(g.K);
  const int S = (g.K + ks - 1) / ks;
  if (S == 1) {
    g.C = out; g.bias_n = bias; g.act = act;
    return launch_gemm_f32(g, s);
  }
  const long mn = (long)g.M * g.N;
  g.ksplit = ks; g.batch = S;
  g.C = part; g.c_sz = mn; g.c_sm = g.N; g.c_sn = 1;
  g.bias_n = nullptr; g.act = 0;
  CA_TRY(launch_gemm_f32(g, s));
  hipLaunchKernelGGL(mlp_reduce_kernel, dim3((unsigned)((mn + 255) / 256)), dim3(256), 0, s, part, S, mn, g.N, bias,
                     act, out);
  CA_CHECK_LAUNCH("mlp_reduce");
  return 0;
}

struct MlpSaved { size_t hw, hp, hs, total; };
MlpSaved plan_saved(int B, int d, int mlp) {
  MlpSaved p;
  size_t o = 0;
  p.hw = o; o += al64((size_t)B * d);
  p.hp = o; o += al64((size_t)B * d);
  p.hs = o; o += al64((size_t)B * mlp);
  p.total = o;
  return p;
}
struct MlpBwd { size_t dhs, dhp, dhw, part, total; };
MlpBwd plan_bwd(int B, int d, int mlp, int K) {
  MlpBwd p;
  size_t o = 0;
  p.dhs = o; o += al64((size_t)B * mlp);
  p.dhp = o; o += al64((size_t)B * d);
  p.dhw = o; o += al64((size_t)B * d);
  p.part = o; o += part_floats(B, d, mlp, K);
  p.total = o;
  return p;
}

int check_mlp_shape(int B, int d, int mlp, int K, int dtype) {
  CA_CHECK_ARG(dtype == COATTN_F32, "unsupported dtype %d (only COATTN_F32)", dtype);
  CA_CHECK_ARG(B > 0 && B <= (1 << 20), "bad batch size B=%d", B);
  CA_CHECK_ARG(d > 0 && d <= 8192 && mlp > 0 && mlp <= 16384 && K > 0 && K <= (1 << 20), "bad d=%d / mlp=%d / K=%d", d, mlp, K);
  return 0;
}

// out[B, n_out] = act( sum_pieces X_i[B, d] . W[:, koff_i : koff_i + d]^T + bias ),  W row stride = w_ld
int layer_forward(const float* const* X, const int* koff, int npieces, const float* W, int w_ld, const float* bias,
                  float* out, float* part, int B, int d, int n_out, int act, hipStream_t s) {
  coattn_gemm_desc g = {};
  for (int i = 0; i < npieces; ++i) {
    g.a_ptrs[i] = X[i];
    g.b_ptrs[i] = W + koff[i];
  }
  g.ptr_by_inner = 1;
  g.inner = npieces;
  g.M = B; g.N = n_out; g.K = d; g.batch = 1;
  g.a_sm = d; g.a_sk = 1;
  g.b_sk = 1; g.b_sn = w_ld;
  g.c_sm = n_out; g.c_sn = 1;
  return skinny_gemm(g, part, bias, act, out, s);
}

// dW[:, col0 : col0 + d] (+)= sum_pieces dz^T X_i      dz [B, n_out], X_i [B, d], dW row stride = w_ld
int layer_dweight(const float* dz, const float* const* X, int npieces, float* dW, int w_ld, int col0, int B, int d,
                  int n_out, int accumulate, hipStream_t s) {
  coattn_gemm_desc g = {};
  for (int i = 0; i < npieces; ++i) {
    g.a_ptrs[i] = dz;
    g.b_ptrs[i] = X[i];
  }
  g.ptr_by_inner = 1;
  g.inner = npieces;
  g.C = dW + col0;
  g.M = n_out; g.N = d; g.K = B; g.batch = 1;
  g.a_sm = 1; g.a_sk = n_out;
  g.b_sk = d; g.b_sn = 1;
  g.c_sm = w_ld; g.c_sn = 1;
  if (accumulate) {
    g.Cin = dW + col0; g.cin_sm = w_ld; g.cin_sn = 1; g.beta = 1.f;
  }
  return launch_gemm_f32(g, s);
}

// dx[B, n_in] = dz[B, n_out] . W[:, col0 : col0 + n_in]
int layer_dinput(const float* dz, const float* W, int w_ld, int col0, float* dx, float* part, int B, int n_out,
                 int n_in, hipStream_t s) {
  coattn_gemm_desc g = {};
  g.A = dz; g.B = W + col0;
  g.M = B; g.N = n_in; g.K = n_out; g.batch = 1;
  g.a_sm = n_out; g.a_sk = 1;
  g.b_sk = w_ld; g.b_sn = 1;
  g.c_sm = n_in; g.c_sn = 1;
  return skinny_gemm(g, part, nullptr, 0, dx, s);
}

int launch_dz_colsum(const float* dh, const float* h, float* dz, float* db, int R, int n, int accumulate,
                     hipStream_t s) {
  hipLaunchKernelGGL(mlp_dz_colsum_kernel, dim3((n + 63) / 64), dim3(256), 0, s, dh, h, dz, db, R, n, accumulate);
  CA_CHECK_LAUNCH("mlp_dz_colsum");
  return 0;
}

}  // namespace

extern "C" int coattn_mlp_workspace_bytes(int B, int d, int mlp, int K, int dtype, size_t* saved, size_t* ws_fwd,
                                          size_t* ws_bwd) {
  CA_TRY(check_mlp_shape(B, d, mlp, K, dtype));
  if (saved) *saved = plan_saved(B, d, mlp).total * sizeof(float);
  // forward scratch: split-k partials, then (inference, saved == NULL) the hidden states
  if (ws_fwd) *ws_fwd = (part_floats(B, d, mlp, K) + plan_saved(B, d, mlp).total) * sizeof(float);
  if (ws_bwd) *ws_bwd = plan_bwd(B, d, mlp, K).total * sizeof(float);
  return 0;
}

extern "C" int coattn_mlp_forward(const void* v, const void* q, const coattn_mlp_params* p, void* logits, void* saved,
                                  void* ws, int B, int d, int mlp, int K, int dtype, int flags, void* stream) {
  (void)flags;
  CA_TRY(check_mlp_shape(B, d, mlp, K, dtype));
  CA_CHECK_ARG(v && q && p && logits && ws, "mlp_forward: null argument");
  CA_CHECK_ARG(p->W_w && p->b_w && p->W_p && p->b_p && p->W_s && p->b_s && p->W_h && p->b_h,
               "mlp_forward: null parameter pointer");
  hipStream_t s = (hipStream_t)stream;
  const MlpSaved sp = plan_saved(B, d, mlp);
  float* part = (float*)ws;
  float* st = saved ? (float*)saved : (float*)ws + part_floats(B, d, mlp, K);
  float* hw = st + sp.hw;
  float* hp = st + sp.hp;
  float* hs = st + sp.hs;
  const float* V = (const float*)v;
  const float* Q = (const float*)q;
  const size_t Bd = (size_t)B * d;
  const int k0[3] = {0, 0, d};
  {
    const float* X[2] = {Q, V};                                        // word level
    CA_TRY(layer_forward(X, k0, 2, (const float*)p->W_w, d, (const float*)p->b_w, hw, part, B, d, d, 1, s));
  }
  {
    const float* X[3] = {Q + Bd, V + Bd, hw};                          // phrase level
    CA_TRY(layer_forward(X, k0, 3, (const float*)p->W_p, 2 * d, (const float*)p->b_p, hp, part, B, d, d, 1, s));
  }
  {
    const float* X[3] = {Q + 2 * Bd, V + 2 * Bd, hp};                  // sentence level
    CA_TRY(layer_forward(X, k0, 3, (const float*)p->W_s, 2 * d, (const float*)p->b_s, hs, part, B, d, mlp, 1, s));
  }
  {
    coattn_gemm_desc g = {};
    g.A = hs; g.B = p->W_h;
    g.M = B; g.N = K; g.K = mlp; g.batch = 1;
    g.a_sm = mlp; g.a_sk = 1;
    g.b_sk = 1; g.b_sn = mlp;
    g.c_sm = K; g.c_sn = 1;
    CA_TRY(skinny_gemm(g, part, (const float*)p->b_h, 0, (float*)logits, s));
  }
  return 0;
}

extern "C" int coattn_mlp_backward(const void* v, const void* q, const coattn_mlp_params* p, const void* saved,
                                   const void* g_logits, void* g_vq, const coattn_mlp_param_grads* pg, int accumulate,
                                   void* ws, int B, int d, int mlp, int K, int dtype, int flags, void* stream) {
  (void)flags;
  CA_TRY(check_mlp_shape(B, d, mlp, K, dtype));
  CA_CHECK_ARG(v && q && p && saved && g_logits && pg && ws, "mlp_backward: null argument");   // g_vq may be NULL
  CA_CHECK_ARG(p->W_w && p->W_p && p->W_s && p->W_h, "mlp_backward: null parameter pointer");
  CA_CHECK_ARG(pg->dW_w && pg->db_w && pg->dW_p && pg->db_p && pg->dW_s && pg->db_s && pg->dW_h && pg->db_h,
               "mlp_backward: null parameter-gradient pointer");
  hipStream_t s = (hipStream_t)stream;
  const MlpSaved sp = plan_saved(B, d, mlp);
  const MlpBwd bp = plan_bwd(B, d, mlp, K);
  const float* st = (const float*)saved;
  const float* hw = st + sp.hw;
  const float* hp = st + sp.hp;
  const float* hs = st + sp.hs;
  float* w = (float*)ws;
  float* dhs = w + bp.dhs;
  float* dhp = w + bp.dhp;
  float* dhw = w + bp.dhw;
  float* part = w + bp.part;
  const float* V = (const float*)v;
  const float* Q = (const float*)q;
  const float* G = (const float*)g_logits;
  float* gx = (float*)g_vq;
  const size_t Bd = (size_t)B * d;
  // logits = W_h h_s + b_h
  {
    const float* X[1] = {hs};
    CA_TRY(layer_dweight(G, X, 1, (float*)pg->dW_h, mlp, 0, B, mlp, K, accumulate, s));
    CA_TRY(launch_dz_colsum(G, nullptr, nullptr, (float*)pg->db_h, B, K, accumulate, s));
    CA_TRY(layer_dinput(G, (const float*)p->W_h, mlp, 0, dhs, part, B, K, mlp, s));
  }
  // sentence level: h_s = tanh(W_s [x_s | h_p] + b_s)
  {
    CA_TRY(launch_dz_colsum(dhs, hs, dhs, (float*)pg->db_s, B, mlp, accumulate, s));
    const float* X[2] = {Q + 2 * Bd, V + 2 * Bd};
    CA_TRY(layer_dweight(dhs, X, 2, (float*)pg->dW_s, 2 * d, 0, B, d, mlp, accumulate, s));
    const float* H[1] = {hp};
    CA_TRY(layer_dweight(dhs, H, 1, (float*)pg->dW_s, 2 * d, d, B, d, mlp, accumulate, s));
    if (gx) CA_TRY(layer_dinput(dhs, (const float*)p->W_s, 2 * d, 0, gx + 2 * Bd, part, B, mlp, d, s));
    CA_TRY(layer_dinput(dhs, (const float*)p->W_s, 2 * d, d, dhp, part, B, mlp, d, s));
  }
  // phrase level: h_p = tanh(W_p [x_p | h_w] + b_p)
  {
    CA_TRY(launch_dz_colsum(dhp, hp, dhp, (float*)pg->db_p, B, d, accumulate, s));
    const float* X[2] = {Q + Bd, V + Bd};
    CA_TRY(layer_dweight(dhp, X, 2, (float*)pg->dW_p, 2 * d, 0, B, d, d, accumulate, s));
    const float* H[1] = {hw};
    CA_TRY(layer_dweight(dhp, H, 1, (float*)pg->dW_p, 2 * d, d, B, d, d, accumulate, s));
    if (gx) CA_TRY(layer_dinput(dhp, (const float*)p->W_p, 2 * d, 0, gx + Bd, part, B, d, d, s));
    CA_TRY(layer_dinput(dhp, (const float*)p->W_p, 2 * d, d, dhw, part, B, d, d, s));
  }
  // word level: h_w = tanh(W_w x_w + b_w)
  {
    CA_TRY(launch_dz_colsum(dhw, hw, dhw, (float*)pg->db_w, B, d, accumulate, s));
    const float* X[2] = {Q, V};
    CA_TRY(layer_dweight(dhw, X, 2, (float*)pg->dW_w, d, 0, B, d, d, accumulate, s));
    if (gx) CA_TRY(layer_dinput(dhw, (const float*)p->W_w, d, 0, gx, part, B, d, d, s));
  }
  return 0;
}

extern "C" int coattn_ce_workspace_bytes(int B, int K, int dtype, size_t* ws) {
  CA_CHECK_ARG(dtype == COATTN_F32, "unsupported dtype %d (only COATTN_F32)", dtype);
  CA_CHECK_ARG(B > 0 && K > 0, "bad B=%d / K=%d", B, K);
  if (ws) *ws = al64((size_t)B) * sizeof(float);
  return 0;
}

extern "C" int coattn_ce_forward(const void* logits, const void* labels, void* loss, void* dlogits, void* ws, int B,
                                 int K, int dtype, void* stream) {
  CA_CHECK_ARG(dtype == COATTN_F32, "unsupported dtype %d (only COATTN_F32)", dtype);
  CA_CHECK_ARG(B > 0 && B <= (1 << 24) && K > 0, "bad B=%d / K=%d", B, K);
  CA_CHECK_ARG(logits && labels && loss && ws, "ce_forward: null argument");                   // dlogits may be NULL
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(ce_rows_kernel, dim3(B), dim3(256), 0, s, (const float*)logits, (const long long*)labels,
                     (float*)ws, (float*)dlogits, K, 1.0f / (float)B);
  CA_CHECK_LAUNCH("ce_rows");
  return launch_sum_all((const float*)ws, (float*)loss, B, 0, s);
}
