// PhraseConvPool (reference model.py:301-334) on gfx950: the three n-gram Conv1d + tanh + the
// reference's max over 3 CONSECUTIVE channels of the concatenated [uni|bi|tri] vector, forward and
// backward, as ONE dense contraction per direction -- on the hand-scheduled bf16-MFMA GEMMs (gemm_w.hip with k bands,
// gemm_tn.hip with a tile mask; fp32-accurate 3-way split) when the channel count is a multiple of 128, else on the
// general GEMM of gemm.hip:
//
//   Xcat[bt] = [x[t-1] | x[t] | x[t+1]]                 (zero rows outside 0 <= t < T)      [B*T, 3E]
//   Wcat[c]  = taps of output channel c of cat(uni, bi, tri) laid out against Xcat          [3E, 3E]
//              uni : (0, W1, 0)   bi : (W2[..,0], W2[..,1], 0)   tri : (W3[..,0], W3[..,1], W3[..,2])
//   Z        = Xcat . Wcat^T + bcat                                                          [B*T, 3E]
//   out[bt,e]= tanh(max_j Z[bt, 3e + j])     (tanh is monotonic: = max_j tanh(Z), model.py:324-332)
//
// backward:  dZ[bt,3e+j] = [j == argmax] g[bt,e] (1 - out^2);  dWcat = dZ^T Xcat;  db = colsum dZ;
//            dXcat = dZ Wcat;  dx[t] = dXcat[t+1, 0:E] + dXcat[t, E:2E] + dXcat[t-1, 2E:3E].
// Padding follows the reference's ConstantPad1d: bigram (1,0), trigram (1,1) (model.py:313-321).
#include "common.h"
#include "fused.h"
#include "gemm_w_body.h"
#include <stdlib.h>

namespace {

// Xcat[bt][j*E + i] = x[b][t + j - 1][i] (0 outside the sequence); one float4 per thread
__global__ __launch_bounds__(256) void phrase_im2col_kernel(const float* __restrict__ X, float* __restrict__ Xcat,
                                                            int T, int E, long n4) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n4) return;
  const int e4 = 3 * E / 4;
  const long bt = idx / e4;
  const int c = (int)(idx % e4) * 4;
  const int j = c / E, i = c % E;
  const int t = (int)(bt % T) + j - 1;
  f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
  if (t >= 0 && t < T) v = *reinterpret_cast<const f32x4*>(X + (bt + j - 1) * E + i);
  *reinterpret_cast<f32x4*>(Xcat + bt * 3 * E + c) = v;
}
__global__ __launch_bounds__(256) void phrase_im2col_scalar_kernel(const float* __restrict__ X, float* __restrict__ Xcat,
                                                                   int T, int E, long n) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n) return;
  const long bt = idx / (3 * E);
  const int c = (int)(idx % (3 * E));
  const int j = c / E, i = c % E;
  const int t = (int)(bt % T) + j - 1;
  Xcat[idx] = (t >= 0 && t < T) ? X[(bt + j - 1) * E + i] : 0.f;
}

// Wcat[c][j*E + i] from the Conv1d weights [E_out][E_in][k]; bcat = [b1 | b2 | b3]
__global__ __launch_bounds__(256) void phrase_pack_w_kernel(const float* __restrict__ W1, const float* __restrict__ W2,
                                                            const float* __restrict__ W3, const float* __restrict__ b1,
                                                            const float* __restrict__ b2, const float* __restrict__ b3,
                                                            float* __restrict__ Wcat, float* __restrict__ bcat, int E) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const long n = 9L * E * E;
  if (idx < 3 * E) bcat[idx] = idx < E ? b1[idx] : (idx < 2 * E ? b2[idx - E] : b3[idx - 2 * E]);
  if (idx >= n) return;
  const int c = (int)(idx / (3 * E)), col = (int)(idx % (3 * E));
  const int j = col / E, i = col % E, g = c / E, co = c % E;
  float v = 0.f;
  if (g == 0) { if (j == 1) v = W1[(long)co * E + i]; }
  else if (g == 1) { if (j < 2) v = W2[((long)co * E + i) * 2 + j]; }
  else v = W3[((long)co * E + i) * 3 + j];
  Wcat[idx] = v;
}

// out[bt][e] = tanh(max_j Z[bt][3e+j]); idx[bt][e] = argmax (first of equals, like MaxPool2d)
__global__ __launch_bounds__(256) void phrase_pool_kernel(const float* __restrict__ Z, float* __restrict__ out,
                                                          unsigned char* __restrict__ amax, long n) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n) return;
  const float a = Z[3 * idx], b = Z[3 * idx + 1], c = Z[3 * idx + 2];
  int k = 0;
  float m = a;
  if (b > m) { m = b; k = 1; }
  if (c > m) { m = c; k = 2; }
  out[idx] = tanhf(m);
  if (amax) amax[idx] = (unsigned char)k;
}

// dZ[bt][3e+j] = (j == amax) ? g (1 - out^2) : 0
__global__ __launch_bounds__(256) void phrase_dz_kernel(const float* __restrict__ g, const float* __restrict__ out,
                                                        const unsigned char* __restrict__ amax, float* __restrict__ dZ,
                                                        long n) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n) return;
  const float o = out[idx];
  const float v = g[idx] * (1.f - o * o);
  const int k = amax[idx];
  dZ[3 * idx] = k == 0 ? v : 0.f;
  dZ[3 * idx + 1] = k == 1 ? v : 0.f;
  dZ[3 * idx + 2] = k == 2 ? v : 0.f;
}

// dx[b][t][i] = dXcat[t+1][i] + dXcat[t][E+i] + dXcat[t-1][2E+i]   (terms outside the sequence dropped)
__global__ __launch_bounds__(256) void phrase_col2im_kernel(const float* __restrict__ dXcat, float* __restrict__ dX,
                                                            int T, int E, long n) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n) return;
  const long bt = idx / E;
  const int i = (int)(idx % E), t = (int)(bt % T);
  float v = dXcat[bt * 3 * E + E + i];
  if (t + 1 < T) v += dXcat[(bt + 1) * 3 * E + i];
  if (t > 0) v += dXcat[(bt - 1) * 3 * E + 2 * E + i];
  dX[idx] = v;
}

// conv weight gradients from the split-K partials of dWcat: dW[co][i][j] (+)= sum_s part[s][c][j*E+i]
__global__ __launch_bounds__(256) void phrase_unpack_dw_kernel(const float* __restrict__ part, int nparts,
                                                               float* __restrict__ dW1, float* __restrict__ dW2,
                                                               float* __restrict__ dW3, int E, int accumulate) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;       // over the 6 E^2 live taps: [uni E^2 | bi 2E^2 | tri 3E^2]
  const long e2 = (long)E * E;
  if (idx >= 6 * e2) return;
  int g, k;
  long r;
  if (idx < e2) { g = 0; k = 1; r = idx; }
  else if (idx < 3 * e2) { g = 1; k = 2; r = idx - e2; }
  else { g = 2; k = 3; r = idx - 3 * e2; }
  const int j = (int)(r % k);
  const long ci = r / k;                                         // co * E + i
  const int co = (int)(ci / E), i = (int)(ci % E);
  const int jj = g == 0 ? 1 : j;                                 // tap position inside Xcat
  const long src = ((long)(g * E + co)) * 3 * E + (long)jj * E + i;
  float v = 0.f;
  for (int s = 0; s < nparts; ++s) v += part[(long)s * 9 * e2 + src];
  float* dst = g == 0 ? dW1 + r : (g == 1 ? dW2 + r : dW3 + r);
  *dst = accumulate ? *dst + v : v;
}

// db[c] (+)= sum_s part[s][c]  split over the three bias vectors
__global__ __launch_bounds__(256) void phrase_unpack_db_kernel(const float* __restrict__ part, int nparts,
                                                               float* __restrict__ db1, float* __restrict__ db2,
                                                               float* __restrict__ db3, int E, int accumulate) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= 3 * E) return;
  float v = 0.f;
  for (int s = 0; s < nparts; ++s) v += part[(long)s * 3 * E + c];
  float* dst = c < E ? db1 + c : (c < 2 * E ? db2 + c - E : db3 + c - 2 * E);
  *dst = accumulate ? *dst + v : v;
}

// Wcat straight into the MFMA-fragment image gemm_w reads (gemm_w.hip: wsplit_kernel's layout), without materialising
// it: one wave per (32-column tile, 16-k step) chunk of Bw(k, n) = Wcat[n][k] (trans = 0: Z = Xcat Wcat^T) or
// Wcat[k][n] (trans = 1: dXcat = dZ Wcat).  Chunks inside an absent tap block are skipped (never read: k bands).
__global__ __launch_bounds__(256) void phrase_wsplit_kernel(const float* __restrict__ W1, const float* __restrict__ W2,
                                                            const float* __restrict__ W3, const float* __restrict__ b1,
                                                            const float* __restrict__ b2, const float* __restrict__ b3,
                                                            char* __restrict__ img, float* __restrict__ bcat, int E,
                                                            int trans, int pieces, float* __restrict__ status) {
  const int gid = blockIdx.x * 256 + threadIdx.x;
  if (status && gid == 0) { status[0] = 0.f; status[1] = pieces == 16 ? 1.f : 0.f; }   // header of the status words (fused.h kStatusHdr)
  if (bcat && gid < 3 * E) bcat[gid] = gid < E ? b1[gid] : (gid < 2 * E ? b2[gid - E] : b3[gid - 2 * E]);
  const int lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
  const int ks16 = 3 * E / 16, chunk = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (chunk >= ks16 * (3 * E / 32)) return;
  const int nt = chunk / ks16, ks = chunk % ks16;
  const int n = 32 * nt + li, k0 = 16 * ks + 8 * lh;
  // (c, col) = (output channel of cat(uni, bi, tri), column of Xcat); a chunk lies inside one E x E block (E % 32 == 0)
  const int c = trans ? k0 : n, col0 = trans ? n : k0;
  const int g = c / E, jb = col0 / E;
  if ((g == 0 && jb != 1) || (g == 1 && jb == 2)) {
    if (status && pieces == 16 && lane == 0) status[kStatusHdr + chunk] = 0.f;
    return;
  }
  f32x8 v;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int cc = (trans ? k0 + e : n) % E, i = (trans ? n : k0 + e) % E;
    v[e] = g == 0 ? W1[(long)cc * E + i] : (g == 1 ? W2[((long)cc * E + i) * 2 + jb] : W3[((long)cc * E + i) * 3 + jb]);
  }
  if (pieces == 16) {                                // two FP16 pieces of kF16WScale * W (fused.h), as gemm_w.hip's wsplit_kernel
    f16_saturating_conversions();
    u32x4 h, m;
    float amax = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      unsigned hh, mm;
      const float w0 = v[2 * e] * kF16WScale, w1 = v[2 * e + 1] * kF16WScale;
      amax = fmaxf(amax, fmaxf(fabsf(w0), fabsf(w1)));
      split_pair_h(w0, w1, hh, mm);
      h[e] = hh; m[e] = mm;
    }
    if (status) {                                    // range report (coattn_phrase_status): one word per wave
      amax = wave_max(amax);
      if (lane == 0) status[kStatusHdr + chunk] = amax;
    }
    char* out = img + (size_t)chunk * gw::kChunkBytes + lane * 16;
    *reinterpret_cast<u32x4*>(out) = h;
    *reinterpret_cast<u32x4*>(out + gw::kFragBytes) = m;
    return;
  }
  bf16x8 pz[3];
  split3(v, pz);
  if (pieces == 1) {                                 // hi piece only, 1 KB chunks: the image gemm_bf.hip reads
    *reinterpret_cast<bf16x8*>(img + (size_t)chunk * gw::kFragBytes + lane * 16) = pz[0];
    return;
  }
  char* out = img + (size_t)chunk * gw::kChunkBytes + lane * 16;
#pragma unroll
  for (int q = 0; q < 3; ++q) *reinterpret_cast<bf16x8*>(out + q * gw::kFragBytes) = pz[q];
}

inline size_t al256(size_t n) { return (n + 255) & ~(size_t)255; }

struct PhrasePlan {
  size_t xcat, z, wcat, bcat, wimg, part, bpart, total_fwd, total_bwd;
  int ksplit, nsplit, rows_per_chunk, nchunks;
};

PhrasePlan plan_phrase(int B, int T, int E) {
  PhrasePlan p = {};
  const size_t bt = (size_t)B * T, e3 = 3 * (size_t)E;
  size_t o = 0;
  p.xcat = o; o += al256(bt * e3 * 4);
  p.z = o; o += al256(bt * e3 * 4);
  p.wcat = o; o += al256(e3 * e3 * 4);
  p.bcat = o; o += al256(e3 * 4);
  p.wimg = o; o += al256(wsplit_bytes(3 * E, 3 * E));       // Wcat split into MFMA-fragment order (gemm_w.hip)
  p.total_fwd = o;
  // split-K of dWcat = dZ^T Xcat over the B*T rows: ~8 slices, each a multiple of 32 rows
  const long K = (long)bt;
  long ks = ((K + 7) / 8 + 31) / 32 * 32;
  if (ks < 32) ks = 32;
  p.ksplit = (int)ks;
  p.nsplit = (int)((K + ks - 1) / ks);
  p.part = o; o += al256((size_t)p.nsplit * e3 * e3 * 4);
  p.rows_per_chunk = 32;
  p.nchunks = (int)((bt + 31) / 32);
  p.bpart = o; o += al256((size_t)p.nchunks * e3 * 4);
  p.total_bwd = o;
  return p;
}

// COATTN_GEMM_W=0 (developer switch): the general GEMM of gemm.hip instead of the hand-scheduled kernels
bool hand_gemms() {
  static const int on = dev_env_int("COATTN_GEMM_W", 1);
  return on != 0;
}

int check_phrase(const void* X, const coattn_phrase_params* p, int B, int T, int E, int dtype) {
  CA_CHECK_ARG(dtype == COATTN_F32, "phrase: only COATTN_F32 is implemented");
  CA_CHECK_ARG(B >= 1 && T >= 1 && E >= 1, "phrase: B, T, E must be >= 1 (got %d, %d, %d)", B, T, E);
  CA_CHECK_ARG((long)B * T * 3 * E < 2147483647L, "phrase: B*T*3E exceeds 32-bit element offsets");
  CA_CHECK_ARG(X && p && p->W1 && p->b1 && p->W2 && p->b2 && p->W3 && p->b3, "phrase: NULL input or parameter pointer");
  return 0;
}

// wimg_trans < 0: Wcat and bcat as fp32 arrays (general GEMM); 0 / 1: the weight image of gemm_w for the forward /
// the dXcat product instead (+ bcat)
int build_operands(const float* X, const coattn_phrase_params* p, char* ws, const PhrasePlan& pl, int B, int T, int E,
                   hipStream_t s, int wimg_trans = -1, int pieces = 3, float* status = nullptr) {
  const long bt = (long)B * T;
  float* Xcat = reinterpret_cast<float*>(ws + pl.xcat);
  if ((E & 3) == 0 && ((((uintptr_t)X) | ((uintptr_t)Xcat)) & 15) == 0) {
    const long n4 = bt * 3 * E / 4;
    hipLaunchKernelGGL(phrase_im2col_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, X, Xcat, T, E, n4);
  } else {
    const long n = bt * 3 * E;
    hipLaunchKernelGGL(phrase_im2col_scalar_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, X, Xcat, T, E, n);
  }
  CA_CHECK_LAUNCH("phrase_im2col");
  if (wimg_trans >= 0) {
    const int chunks = (3 * E / 16) * (3 * E / 32);
    hipLaunchKernelGGL(phrase_wsplit_kernel, dim3((unsigned)((chunks + 3) / 4)), dim3(256), 0, s, (const float*)p->W1,
                       (const float*)p->W2, (const float*)p->W3, (const float*)p->b1, (const float*)p->b2,
                       (const float*)p->b3, ws + pl.wimg, reinterpret_cast<float*>(ws + pl.bcat), E, wimg_trans, pieces, status);
    CA_CHECK_LAUNCH("phrase_wsplit");
    return 0;
  }
  if (status && hipMemsetAsync(status, 0, 2 * sizeof(float), s) != hipSuccess) {   // (no weight-split launch: header = "no FP16 pieces")
    coattn_set_error("phrase: hipMemsetAsync failed");
    return -3;
  }
  const long nw = 9L * E * E;
  hipLaunchKernelGGL(phrase_pack_w_kernel, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, s,
                     (const float*)p->W1, (const float*)p->W2, (const float*)p->W3, (const float*)p->b1,
                     (const float*)p->b2, (const float*)p->b3, reinterpret_cast<float*>(ws + pl.wcat),
                     reinterpret_cast<float*>(ws + pl.bcat), E);
  CA_CHECK_LAUNCH("phrase_pack_w");
  return 0;
}

}  // namespace

extern "C" int coattn_phrase_workspace_bytes(int B, int T, int E, int dtype, size_t* saved, size_t* ws_fwd,
                                             size_t* ws_bwd) {
  CA_CHECK_ARG(dtype == COATTN_F32, "phrase: only COATTN_F32 is implemented");
  CA_CHECK_ARG(B >= 1 && T >= 1 && E >= 1, "phrase: B, T, E must be >= 1 (got %d, %d, %d)", B, T, E);
  const PhrasePlan pl = plan_phrase(B, T, E);
  // argmax index per output element, 1 byte each; then the status words of coattn_phrase_status
  if (saved) *saved = al256((size_t)B * T * E) + al256(status_floats(3 * E, 3 * E, 1) * sizeof(float));
  if (ws_fwd) *ws_fwd = pl.total_fwd;
  if (ws_bwd) *ws_bwd = pl.total_bwd;
  return 0;
}

extern "C" int coattn_phrase_forward(const void* X, const coattn_phrase_params* p, void* out, void* saved, void* ws,
                                     int B, int T, int E, int dtype, int flags, void* stream) {
  CA_TRY(check_phrase(X, p, B, T, E, dtype));
  CA_CHECK_ARG(out && ws, "phrase_forward: NULL output or workspace");
  hipStream_t s = (hipStream_t)stream;
  const bool bf16 = (flags & COATTN_FLAG_BF16_PROJ) != 0;
  auto gemm = [&](const coattn_gemm_desc& g) { return bf16 ? launch_gemm_bf16in(g, s) : launch_gemm_f32(g, s); };
  const PhrasePlan pl = plan_phrase(B, T, E);
  char* w = static_cast<char*>(ws);
  WGemm wg = {};
  wg.A = reinterpret_cast<const float*>(w + pl.xcat); wg.a_sm = 3 * E; wg.Wf = w + pl.wimg;
  wg.C = reinterpret_cast<float*>(w + pl.z); wg.c_sm = 3 * E; wg.bias_n = reinterpret_cast<const float*>(w + pl.bcat);
  wg.M = B * T; wg.N = 3 * E; wg.K = 3 * E; wg.batch = 1;
  wg.bf16 = bf16 ? 1 : 0;                                // reduced precision: one MFMA per product (gemm_bf.hip / gemm_w.hip)
  // fp32 mode, COATTN_FLAG_FAST16: the forward product Z = Xcat Wcat^T on two FP16 pieces per operand, as the co-attention's
  // projections (fused.h; its operands are word features and conv weights; range report: coattn_phrase_status, which needs
  // `saved`) -- default: the exact bf16 split
  static const int f16_env = dev_env_int("COATTN_FWD_F16", 1), split_env = dev_env_int("COATTN_SPLIT", 2);
  const bool fast = (flags & COATTN_FLAG_FAST16) && !(flags & COATTN_FLAG_EXACT3) && split_env != 3;
  float* status = saved ? reinterpret_cast<float*>(static_cast<char*>(saved) + al256((size_t)B * T * E)) : nullptr;
  if (!bf16 && fast && f16_env && status) { wg.np = 2; wg.f16 = 1; wg.status = status; }
  if (E % 128 == 0) {                                    // (k bands first: the kernel and its weight-image format depend on them)
    wg.kband_n = E;
    wg.kband_lo[0] = E; wg.kband_hi[0] = 2 * E;
    wg.kband_lo[1] = 0; wg.kband_hi[1] = 2 * E;
    wg.kband_lo[2] = 0; wg.kband_hi[2] = 3 * E;
  }
  const bool hand = hand_gemms() && E % 128 == 0 && gemm_w_supported(wg);
  if (!hand) { wg.np = 0; wg.f16 = 0; wg.status = nullptr; }   // general GEMM: exact
  CA_TRY(build_operands((const float*)X, p, w, pl, B, T, E, s, hand ? 0 : -1, wimg_pieces(wg), status));
  coattn_gemm_desc g = {};
  g.A = w + pl.xcat; g.B = w + pl.wcat; g.C = w + pl.z; g.bias_n = w + pl.bcat;
  g.M = B * T; g.N = 3 * E; g.K = 3 * E; g.batch = 1; g.inner = 1;
  g.a_sm = 3 * E; g.a_sk = 1; g.b_sk = 1; g.b_sn = 3 * E; g.c_sm = 3 * E; g.c_sn = 1;
  if (E % 128 == 0) {       // skip the zero tap blocks of Wcat: uni uses x[t], bi x[t-1..t], tri x[t-1..t+1]
    g.kband_n = E;
    g.kband_lo[0] = E; g.kband_hi[0] = 2 * E;
    g.kband_lo[1] = 0; g.kband_hi[1] = 2 * E;
    g.kband_lo[2] = 0; g.kband_hi[2] = 3 * E;
  }
  // fp32, 128-aligned channels: the pre-split-weight kernel (Wcat split once into MFMA-fragment order; its zero tap
  // blocks are never read)
  if (hand) {
    CA_TRY(launch_gemm_wx(&wg, 1, s));
  } else {
    CA_TRY(gemm(g));
  }
  const long n = (long)B * T * E;
  hipLaunchKernelGGL(phrase_pool_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s,
                     reinterpret_cast<const float*>(w + pl.z), (float*)out, (unsigned char*)saved, n);
  CA_CHECK_LAUNCH("phrase_pool");
  return 0;
}

extern "C" int coattn_phrase_status_accumulate(const void* saved, int B, int T, int E, void* acc, void* stream) {
  CA_CHECK_ARG(saved != nullptr && B >= 1 && T >= 1 && E >= 1, "phrase_status_accumulate: bad argument");
  const float* status = reinterpret_cast<const float*>(static_cast<const char*>(saved) + al256((size_t)B * T * E));
  return fold_status_words(status, (int)status_floats(3 * E, 3 * E, 1), (float*)acc, (hipStream_t)stream, "coattn_phrase_status_accumulate");
}

extern "C" int coattn_phrase_status(const void* saved, int B, int T, int E, void* stream, float* amax) {
  CA_CHECK_ARG(saved != nullptr && B >= 1 && T >= 1 && E >= 1, "phrase_status: bad argument");
  const float* status = reinterpret_cast<const float*>(static_cast<const char*>(saved) + al256((size_t)B * T * E));
  return read_status_words(status, (int)status_floats(3 * E, 3 * E, 1), (hipStream_t)stream, amax, "coattn_phrase_status");
}

extern "C" int coattn_phrase_backward(const void* X, const coattn_phrase_params* p, const void* out, const void* saved,
                                      const void* g_out, void* dX, const coattn_phrase_param_grads* pg, int accumulate,
                                      void* ws, int B, int T, int E, int dtype, int flags, void* stream) {
  CA_TRY(check_phrase(X, p, B, T, E, dtype));
  CA_CHECK_ARG(out && saved && g_out && pg && ws, "phrase_backward: NULL pointer");
  const bool bf16 = (flags & COATTN_FLAG_BF16_PROJ) != 0;
  auto gemm = [&](const coattn_gemm_desc& g) {
    return bf16 ? launch_gemm_bf16in(g, (hipStream_t)stream) : launch_gemm_f32(g, (hipStream_t)stream);
  };
  CA_CHECK_ARG(pg->dW1 && pg->db1 && pg->dW2 && pg->db2 && pg->dW3 && pg->db3, "phrase_backward: NULL gradient pointer");
  hipStream_t s = (hipStream_t)stream;
  const PhrasePlan pl = plan_phrase(B, T, E);
  char* w = static_cast<char*>(ws);
  const long bt = (long)B * T, n = bt * E;
  float* dZ = reinterpret_cast<float*>(w + pl.z);
  WGemm wdx = {};                                        // dXcat = dZ Wcat (as the forward: Wcat split the other way round)
  wdx.A = dZ; wdx.a_sm = 3 * E; wdx.Wf = w + pl.wimg; wdx.C = reinterpret_cast<float*>(w + pl.xcat); wdx.c_sm = 3 * E;
  wdx.M = (int)bt; wdx.N = 3 * E; wdx.K = 3 * E; wdx.batch = 1;
  wdx.bf16 = bf16 ? 1 : 0;
  // fp32 mode, COATTN_FLAG_FAST16: the two gradient products (dXcat = dZ Wcat, dWcat = dZ^T Xcat) on two bf16 pieces per
  // operand, as the co-attention's backward (fused.h "Widths": gradients keep bf16's range) -- default: three
  static const int split_env = dev_env_int("COATTN_SPLIT", 2);
  const int np_b = (!bf16 && (flags & COATTN_FLAG_FAST16) && !(flags & COATTN_FLAG_EXACT3) && split_env != 3) ? 2 : 3;
  wdx.np = np_b;
  if (E % 128 == 0) {                                    // tap block j of dXcat receives only the n-grams that have that tap
    wdx.kband_n = E;
    wdx.kband_lo[0] = E; wdx.kband_hi[0] = 3 * E;        // x[t-1]: bi, tri
    wdx.kband_lo[1] = 0; wdx.kband_hi[1] = 3 * E;        // x[t]  : all
    wdx.kband_lo[2] = 2 * E; wdx.kband_hi[2] = 3 * E;    // x[t+1]: tri
  }
  const bool hand_dx = dX && hand_gemms() && E % 128 == 0 && gemm_w_supported(wdx);
  CA_TRY(build_operands((const float*)X, p, w, pl, B, T, E, s, hand_dx ? 1 : -1, wimg_pieces(wdx)));
  hipLaunchKernelGGL(phrase_dz_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const float*)g_out,
                     (const float*)out, (const unsigned char*)saved, dZ, n);
  CA_CHECK_LAUNCH("phrase_dz");
  // bias gradients: column sums of dZ
  int nchunks = 0;
  CA_TRY(launch_colsum_partial(nullptr, dZ, reinterpret_cast<float*>(w + pl.bpart), (int)bt, 3 * E, pl.rows_per_chunk,
                               &nchunks, s));
  // dWcat partials = dZ^T Xcat over split row ranges
  // one launch per n-gram: only the tap blocks that exist (uni: x[t]; bi: x[t-1..t]; tri: all three)
  // fp32, 128-aligned channels: ONE split-K launch of the hand-scheduled A^T B kernel over the [3E x 3E] result with
  // the three absent tap blocks masked out (gemm_tn.hip); parts [S][3E][3E] as the unpack kernel reads them
  int nparts = pl.nsplit;
  TnGemm tn = {};
  tn.A = dZ; tn.a_ld = 3 * E; tn.B = reinterpret_cast<const float*>(w + pl.xcat); tn.b_ld = 3 * E;
  tn.C = reinterpret_cast<float*>(w + pl.part); tn.M = 3 * E; tn.N = 3 * E; tn.K = (int)bt; tn.levels = 1;
  tn.mask_blk = E / 128;                             // rows: n-gram (uni, bi, tri); columns: tap x[t-1], x[t], x[t+1]
  tn.tile_mask = (1u << 1) | (3u << 3) | (7u << 6);
  tn.bf16 = bf16 ? 1 : 0;
  tn.np = np_b;
  const bool tn_ok = hand_gemms() && E % 128 == 0 && gemm_tn_supported(tn);
  if (tn_ok && gemm_bf_tn_supported(tn)) {
    // reduced-precision mode at wide shapes: the single-product 256 x 256 kernel of gemm_bf.hip (same mask, same part layout)
    const int live = 6 * (E / 256) * (E / 256);
    int want = (bf_tn_rounds() * 256 + live - 1) / live, spp;
    if (want > pl.nsplit) want = pl.nsplit;
    const int parts = gemm_bf_tn_plan(tn, want, &spp);
    hipLaunchKernelGGL(phrase_unpack_db_kernel, dim3((unsigned)((3 * E + 255) / 256)), dim3(256), 0, s,
                       reinterpret_cast<const float*>(w + pl.bpart), nchunks, (float*)pg->db1, (float*)pg->db2,
                       (float*)pg->db3, E, accumulate);
    CA_CHECK_LAUNCH("phrase_unpack_db");
    CA_TRY(launch_gemm_bf_tn(&tn, &spp, &parts, 1, s));
    nparts = parts;
  } else if (tn_ok) {
    const int live = 6 * (E / 128) * (E / 128);      // tiles that exist
    int S = (512 + live - 1) / live;
    if (S > pl.nsplit) S = pl.nsplit;
    int ks = (int)((bt + S - 1) / S);
    ks = (ks + 15) / 16 * 16;
    S = (int)((bt + ks - 1) / ks);
    // (the bias-gradient partials are summed by a few extra workgroups of the same launch)
    TnReduce red = {};
    float* db[3] = {(float*)pg->db1, (float*)pg->db2, (float*)pg->db3};
    for (int i = 0; i < 3; ++i) { red.src[i] = reinterpret_cast<const float*>(w + pl.bpart) + (long)i * E; red.dst[i] = db[i]; }
    red.njobs = 3; red.nparts = nchunks; red.n = E; red.ld = 3L * E; red.accumulate = accumulate;
    CA_TRY(launch_gemm_tn(&tn, &ks, &S, 1, s, &red));
    nparts = S;
  } else {
    hipLaunchKernelGGL(phrase_unpack_db_kernel, dim3((unsigned)((3 * E + 255) / 256)), dim3(256), 0, s,
                       reinterpret_cast<const float*>(w + pl.bpart), nchunks, (float*)pg->db1, (float*)pg->db2,
                       (float*)pg->db3, E, accumulate);
    CA_CHECK_LAUNCH("phrase_unpack_db");
  }
  for (int gr = 0; gr < 3 && !tn_ok; ++gr) {
    const int lo = gr == 0 ? E : 0, hi = gr == 0 ? 2 * E : (gr == 1 ? 2 * E : 3 * E);
    coattn_gemm_desc g = {};
    g.A = dZ + (long)gr * E;
    g.B = reinterpret_cast<const float*>(w + pl.xcat) + lo;
    g.C = reinterpret_cast<float*>(w + pl.part) + (long)gr * E * 3 * E + lo;
    g.M = E; g.N = hi - lo; g.K = (int)bt; g.batch = pl.nsplit; g.inner = 1; g.ksplit = pl.ksplit;
    g.a_sm = 1; g.a_sk = 3 * E; g.b_sk = 3 * E; g.b_sn = 1; g.c_sm = 3 * E; g.c_sn = 1; g.c_sz = 9L * E * E;
    CA_TRY(gemm(g));
  }
  const long nt = 6L * E * E;
  hipLaunchKernelGGL(phrase_unpack_dw_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, s,
                     reinterpret_cast<const float*>(w + pl.part), nparts, (float*)pg->dW1, (float*)pg->dW2,
                     (float*)pg->dW3, E, accumulate);
  CA_CHECK_LAUNCH("phrase_unpack_dw");
  if (dX) {
    // dXcat = dZ Wcat into the Xcat buffer (its last reader, the weight-gradient GEMM, is queued before)
    coattn_gemm_desc h = {};
    h.A = dZ; h.B = w + pl.wcat; h.C = w + pl.xcat;
    h.M = (int)bt; h.N = 3 * E; h.K = 3 * E; h.batch = 1; h.inner = 1;
    h.a_sm = 3 * E; h.a_sk = 1; h.b_sk = 3 * E; h.b_sn = 1; h.c_sm = 3 * E; h.c_sn = 1;
    if (E % 128 == 0) {     // tap block j of dXcat receives only the n-grams that have that tap
      h.kband_n = E;
      h.kband_lo[0] = E; h.kband_hi[0] = 3 * E;           // x[t-1]: bi, tri
      h.kband_lo[1] = 0; h.kband_hi[1] = 3 * E;           // x[t]  : all
      h.kband_lo[2] = 2 * E; h.kband_hi[2] = 3 * E;       // x[t+1]: tri
    }
    if (hand_dx) {
      CA_TRY(launch_gemm_wx(&wdx, 1, s));
    } else {
      CA_TRY(gemm(h));
    }
    hipLaunchKernelGGL(phrase_col2im_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s,
                       reinterpret_cast<const float*>(w + pl.xcat), (float*)dX, T, E, n);
    CA_CHECK_LAUNCH("phrase_col2im");
  }
  return 0;
}
