// C-ABI of the co-attention path (include/coattn.h): argument checking, workspace planning
// and kernel orchestration.  No torch types, no allocation, no synchronisation: every call
// only enqueues kernels on the caller's stream.
#include "common.h"
#include "fused.h"

#include <string.h>

// ---------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

void coattn_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int coattn_version(void) { return 601; }   // 0.6.1: coattn_status_accumulate / coattn_phrase_status_accumulate (sticky range report in a caller-owned accumulator); 0.6.0: flags = 0 is the exact mode, COATTN_FLAG_FAST16 the tolerance mode, coattn_status / coattn_phrase_status; 0.5.2: forward-side contractions on two FP16 pieces (COATTN_FLAG_F16PAIR); 0.5.1: coattn_features_native; 0.5.0: widths of the fp32 mode (COATTN_FLAG_EXACT3 / _SPLIT2), coattn_profile_*

// ---------------------------------------------------------------------------------------
// per-kernel timing (bench.py's backward roofline legs): HIP events recorded between the launches of the calls made
// on this thread between coattn_profile_begin and coattn_profile_end
// ---------------------------------------------------------------------------------------
namespace {
constexpr int kProfMax = 48;
struct Prof {
  bool on = false;
  int n = 0, created = 0;
  hipEvent_t ev[kProfMax + 1];
  const char* name[kProfMax];
};
thread_local Prof g_prof;
}  // namespace

void prof_mark(hipStream_t s, const char* name) {
  Prof& p = g_prof;
  if (!p.on || p.n >= kProfMax) return;
  if (hipEventRecord(p.ev[p.n + 1], s) != hipSuccess) { p.on = false; return; }
  p.name[p.n++] = name;
}

extern "C" int coattn_profile_begin(void* stream) {
  Prof& p = g_prof;
  while (p.created <= kProfMax) {
    if (hipEventCreate(&p.ev[p.created]) != hipSuccess) { coattn_set_error("profile: hipEventCreate failed"); return -3; }
    ++p.created;
  }
  p.n = 0;
  if (hipEventRecord(p.ev[0], (hipStream_t)stream) != hipSuccess) { coattn_set_error("profile: hipEventRecord failed"); return -3; }
  p.on = true;
  return 0;
}

extern "C" int coattn_profile_end(float* us, char* names, int names_bytes, int max_marks) {
  Prof& p = g_prof;
  CA_CHECK_ARG(p.on || p.n > 0, "profile: no coattn_profile_begin on this thread (or an event could not be recorded)");
  p.on = false;
  CA_CHECK_ARG(us && max_marks >= p.n, "profile: room for %d marks needed", p.n);
  if (p.n == 0) return 0;
  if (hipEventSynchronize(p.ev[p.n]) != hipSuccess) { coattn_set_error("profile: hipEventSynchronize failed"); return -3; }
  int o = 0;
  if (names && names_bytes > 0) names[0] = 0;
  for (int i = 0; i < p.n; ++i) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, p.ev[i], p.ev[i + 1]) != hipSuccess) { coattn_set_error("profile: hipEventElapsedTime failed"); return -3; }
    us[i] = ms * 1000.f;
    if (names && o < names_bytes - 1) o += snprintf(names + o, (size_t)(names_bytes - o), "%s%s", i ? "\n" : "", p.name[i]);
  }
  const int n = p.n;
  p.n = 0;
  return n;
}
extern "C" const char* coattn_last_error(void) { return g_err; }

// ---------------------------------------------------------------------------------------
// buffer plans (in floats, every region aligned to 64 floats = 256 B)
// ---------------------------------------------------------------------------------------
static inline size_t al64(size_t n) { return (n + 63) & ~(size_t)63; }

typedef SavedOff SavedPlan;
static SavedPlan plan_saved(int B, int N, int T, int d, int L) { return saved_off(B, N, T, d, L); }

static const int kMaxSplits = 32;

struct BwdPlan {
  size_t Hv, dPv, dZq, dPq, dC, dav, dsv, daq, dsq, part, total;
};
static BwdPlan plan_bwd(int B, int N, int T, int d, int L) {
  BwdPlan p;
  size_t o = 0;
  p.Hv = o;  o += al64((size_t)B * N * d);
  p.dPv = o; o += al64((size_t)B * N * d);
  p.dZq = o; o += al64((size_t)B * T * d);
  p.dPq = o; o += al64((size_t)L * B * T * d);
  p.dC = o;  o += al64((size_t)L * B * T * N);
  p.dav = o; o += al64((size_t)L * B * N);
  p.dsv = o; o += al64((size_t)L * B * N);
  p.daq = o; o += al64((size_t)L * B * T);
  p.dsq = o; o += al64((size_t)L * B * T);
  size_t part = (size_t)kMaxSplits * d * d;
  size_t cs = (size_t)260 * d;
  p.part = o; o += al64(part > cs ? part : cs);
  p.total = o;
  return p;
}

static size_t fwd_ws_floats(int B, int N, int T, int d, int L) {
  return plan_saved(B, N, T, d, L).total + al64((size_t)B * N * d);
}
static size_t bwd_ws_floats(int B, int N, int T, int d, int L) {
  size_t bw = plan_bwd(B, N, T, d, L).total;
  if (fused_supported(B, N, T, d, L)) {
    const size_t fb = fused_bwd_ws_floats(B, N, T, d, L);
    if (fb > bw) bw = fb;
  }
  return bw;
}

static int check_shape(int B, int N, int T, int d, int L, int dtype) {
  CA_CHECK_ARG(dtype == COATTN_F32, "unsupported dtype %d (only COATTN_F32)", dtype);
  CA_CHECK_ARG(B > 0 && B <= 65535, "bad batch size B=%d", B);
  CA_CHECK_ARG(N > 0 && N <= 4096 && T > 0 && T <= 4096, "bad N=%d / T=%d", N, T);
  CA_CHECK_ARG(d > 0 && d <= 8192, "bad hidden size d=%d", d);
  CA_CHECK_ARG(L > 0 && L <= 8, "bad number of levels L=%d", L);
  return 0;
}

extern "C" int coattn_fused_supported(int B, int N, int T, int d, int L, int dtype) {
  if (dtype != COATTN_F32) return 0;
  return fused_supported(B, N, T, d, L);
}

extern "C" int coattn_workspace_bytes(int B, int N, int T, int d, int L, int dtype, int flags, size_t* saved,
                                      size_t* ws_fwd, size_t* ws_bwd) {
  (void)flags;
  CA_TRY(check_shape(B, N, T, d, L, dtype));
  const SavedPlan sp = plan_saved(B, N, T, d, L);
  if (saved) *saved = sp.total * sizeof(float);
  // the forward workspace ends with room for two pre-split weight images (gemm_w.hip)
  if (ws_fwd) *ws_fwd = fwd_ws_floats(B, N, T, d, L) * sizeof(float) + 2 * wsplit_bytes(d, d);
  if (ws_bwd) *ws_bwd = bwd_ws_floats(B, N, T, d, L) * sizeof(float);
  return 0;
}

extern "C" int coattn_gemm_f32(const coattn_gemm_desc* g, void* stream) {
  CA_CHECK_ARG(g != nullptr, "gemm: null descriptor");
  return launch_gemm_f32(*g, (hipStream_t)stream);
}

extern "C" size_t coattn_linear_workspace_bytes(int N, int K) { return (N > 0 && K > 0) ? wsplit_bytes(N, K) : 0; }

extern "C" int coattn_linear_forward(const void* x, int64_t ld_x, const void* W, const void* bias, void* y, void* wimg,
                                     int M, int N, int K, float out_scale, int flags, void* stream) {
  CA_CHECK_ARG(x && W && y && wimg && M > 0 && N > 0 && K > 0 && ld_x >= K && ld_x < (1L << 24), "linear: bad argument");
  WGemm g = {};
  g.A = (const float*)x; g.a_sm = (int)ld_x; g.Wf = wimg; g.C = (float*)y; g.c_sm = N; g.bias_n = (const float*)bias;
  g.out_scale = out_scale; g.M = M; g.N = N; g.K = K; g.batch = 1;
  g.bf16 = (flags & COATTN_FLAG_BF16_PROJ) ? 1 : 0;
  g.np = (flags & COATTN_FLAG_SPLIT2) ? 2 : 3;
  if ((flags & COATTN_FLAG_F16PAIR) && !g.bf16) { g.np = 2; g.f16 = 1; }
  g.a_bf16 = (flags & COATTN_FLAG_BF16_IN) ? 1 : 0;
  CA_CHECK_ARG(!g.a_bf16 || g.bf16, "linear: COATTN_FLAG_BF16_IN needs COATTN_FLAG_BF16_PROJ");
  CA_CHECK_ARG(g.a_bf16 ? gemm_bf_supported(g) : gemm_w_supported(g), "linear: shape M=%d N=%d K=%d ld=%ld not supported (see coattn.h)", M, N, K, (long)ld_x);
  if (!(flags & 1)) {
    const WSplit job{(const float*)W, wimg, N, K, 0, K, wimg_pieces(g), nullptr};
    CA_TRY(launch_wsplit(&job, 1, (hipStream_t)stream));
  }
  return launch_gemm_wx(&g, 1, (hipStream_t)stream);
}

extern "C" size_t coattn_linear_wgrad_workspace_bytes(int n_out, int n_in) {
  return (n_out > 0 && n_in > 0) ? (size_t)32 * n_out * n_in * sizeof(float) : 0;
}

extern "C" int coattn_linear_weight_grad(const void* dy, int64_t ld_dy, const void* x, int64_t ld_x, void* dW, void* ws,
                                         int M, int n_out, int n_in, int accumulate, void* stream) {
  CA_CHECK_ARG(dy && x && dW && ws && M > 0 && n_out > 0 && n_in > 0 && ld_dy >= n_out && ld_x >= n_in &&
               ld_dy < (1L << 24) && ld_x < (1L << 24), "linear weight grad: bad argument");
  TnGemm g = {};
  g.A = (const float*)dy; g.a_ld = (int)ld_dy; g.B = (const float*)x; g.b_ld = (int)ld_x; g.C = (float*)ws;
  g.M = n_out; g.N = n_in; g.K = M; g.levels = 1;
  g.bf16 = (accumulate & COATTN_FLAG_BF16_PROJ) ? 1 : 0;
  g.np = (accumulate & COATTN_FLAG_SPLIT2) ? 2 : 3;
  g.a_bf16 = (accumulate & COATTN_FLAG_BF16_IN) ? 1 : 0;
  accumulate &= 1;
  CA_CHECK_ARG(!g.a_bf16 || g.bf16, "linear weight grad: COATTN_FLAG_BF16_IN needs COATTN_FLAG_BF16_PROJ");
  CA_CHECK_ARG(g.a_bf16 ? gemm_bf_tn_supported(g) : gemm_tn_supported(g), "linear weight grad: shape M=%d n_out=%d n_in=%d not supported (see coattn.h)", M, n_out, n_in);
  if (gemm_bf_tn_supported(g)) {                     // reduced-precision mode, wide shape: the single-product kernel (gemm_bf.hip)
    const int ntiles = (n_out / 256) * (n_in / 256);
    int want = (bf_tn_rounds() * 256 + ntiles - 1) / ntiles, spp, parts;
    want = want > 32 ? 32 : want;
    parts = gemm_bf_tn_plan(g, want, &spp);
    CA_TRY(launch_gemm_bf_tn(&g, &spp, &parts, 1, (hipStream_t)stream));
    return launch_reduce_partials((const float*)ws, (float*)dW, parts, (int64_t)n_out * n_in, accumulate, (hipStream_t)stream);
  }
  int ks, S;
  const int parts = gemm_tn_plan(g, 32, &ks, &S);
  CA_TRY(launch_gemm_tn(&g, &ks, &S, 1, (hipStream_t)stream));
  return launch_reduce_partials((const float*)ws, (float*)dW, parts, (int64_t)n_out * n_in, accumulate, (hipStream_t)stream);
}

extern "C" int coattn_gemm_bf16(const coattn_gemm_desc* g, void* stream) {
  CA_CHECK_ARG(g != nullptr, "gemm: null descriptor");
  return launch_gemm_bf16in(*g, (hipStream_t)stream);
}

// Range report of the tolerance mode (include/coattn.h): header + per-chunk weight maxima from the status words.
int read_status_words(const float* status, int n_words, hipStream_t s, float* amax, const char* what) {
  if (hipStreamSynchronize(s) != hipSuccess) { coattn_set_error("%s: hipStreamSynchronize failed", what); return -3; }
  float hdr[2] = {0.f, 0.f};
  if (hipMemcpy(hdr, status, sizeof(hdr), hipMemcpyDeviceToHost) != hipSuccess) { coattn_set_error("%s: reading the status words failed", what); return -3; }
  if (amax) amax[0] = amax[1] = 0.f;
  if (hdr[1] != 1.f) return 0;                       // the call converted nothing to FP16 pieces
  float wmax = 0.f;
  {
    const int n = n_words - kStatusHdr;
    float* w = (float*)malloc((size_t)n * sizeof(float));
    CA_CHECK_ARG(w != nullptr, "%s: out of host memory", what);
    const hipError_t e = hipMemcpy(w, status + kStatusHdr, (size_t)n * sizeof(float), hipMemcpyDeviceToHost);
    for (int i = 0; e == hipSuccess && i < n; ++i) wmax = (w[i] > wmax || w[i] != w[i]) ? w[i] : wmax;
    free(w);
    if (e != hipSuccess) { coattn_set_error("%s: reading the status words failed", what); return -3; }
  }
  if (amax) { amax[0] = hdr[0]; amax[1] = wmax; }
  const bool act = !(hdr[0] <= kF16Exact), wgt = !(wmax <= kF16Exact);
  if (act || wgt) {
    coattn_set_error("%s: FP16-piece range exceeded in the last forward (COATTN_FLAG_FAST16): %s%s%s -- pieces were clamped; "
                     "re-run with flags = 0 (exact)", what,
                     act ? "an activation (feature or stored projection) of magnitude > 65504" : "", act && wgt ? " and " : "",
                     wgt ? "a projection weight of magnitude > 255.87" : "");
    if (amax) { /* magnitudes are in amax */ }
    return -4;
  }
  return 0;
}

// Folds a call's status words into a caller-owned accumulator (two floats, device): acc[0] = max over calls of the largest
// out-of-range activation, acc[1] = max over calls of the largest |256 W| -- bit-pattern maxima of non-negative values, so a
// NaN wins over everything.  One 256-thread workgroup, asynchronous.
__global__ __launch_bounds__(256) void status_fold_kernel(const float* __restrict__ status, int n_words, float* __restrict__ acc) {
  if (status[1] != 1.f) return;                       // the call converted nothing to FP16 pieces
  unsigned m = 0;
  for (int i = kStatusHdr + (int)threadIdx.x; i < n_words; i += 256) {
    const unsigned b = __builtin_bit_cast(unsigned, status[i]) & 0x7fffffffu;
    m = b > m ? b : m;
  }
  for (int o = 32; o > 0; o >>= 1) { const unsigned x = (unsigned)__shfl_xor((int)m, o, 64); m = x > m ? x : m; }
  if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<unsigned*>(acc) + 1, m);
  if (threadIdx.x == 0) atomicMax(reinterpret_cast<unsigned*>(acc), __builtin_bit_cast(unsigned, status[0]) & 0x7fffffffu);
}

int fold_status_words(const float* status, int n_words, float* acc, hipStream_t s, const char* what) {
  CA_CHECK_ARG(acc != nullptr, "%s: null accumulator", what);
  hipLaunchKernelGGL(status_fold_kernel, dim3(1), dim3(256), 0, s, status, n_words, acc);
  if (hipGetLastError() != hipSuccess) { coattn_set_error("%s: launch failed", what); return -3; }
  return 0;
}

extern "C" int coattn_status_accumulate(const void* saved, int B, int N, int T, int d, int L, int dtype, void* acc, void* stream) {
  CA_TRY(check_shape(B, N, T, d, L, dtype));
  CA_CHECK_ARG(saved != nullptr, "status_accumulate: null `saved`");
  const SavedPlan sp = plan_saved(B, N, T, d, L);
  return fold_status_words((const float*)saved + sp.status, (int)status_floats(d, d, 2), (float*)acc, (hipStream_t)stream,
                           "coattn_status_accumulate");
}

extern "C" int coattn_status(const void* saved, int B, int N, int T, int d, int L, int dtype, void* stream, float* amax) {
  CA_TRY(check_shape(B, N, T, d, L, dtype));
  CA_CHECK_ARG(saved != nullptr, "status: null `saved`");
  const SavedPlan sp = plan_saved(B, N, T, d, L);
  return read_status_words((const float*)saved + sp.status, (int)status_floats(d, d, 2), (hipStream_t)stream, amax, "coattn_status");
}

// ---------------------------------------------------------------------------------------
// general-shape implementation: MFMA GEMM composition
// ---------------------------------------------------------------------------------------
// COATTN_BF_TN_ROUNDS (developer switch): rounds of workgroups the split-K parts of gemm_bf.hip's weight-gradient kernel fill
int bf_tn_rounds() {
  static const int r = dev_env_int("COATTN_BF_TN_ROUNDS", 1);   // (measured: 1 round 101 us, 2: 128, 3: 152 -- the partial results' round trip)
  return r < 1 ? 1 : r;
}

namespace {

struct Ctx {
  int B, N, T, d, L;
  hipStream_t s;
  VLayout vl;                 // element strides of x_img[B,N,d]
  bool bf16_proj = false;     // COATTN_FLAG_BF16_PROJ: the projections and their gradients on the bf16 MFMA
  int np_pq = 3;              // width of the P_q projection in the fp32 mode (3 | 2)
  bool f16_proj = false;      // fp32 mode: both projections on two FP16 pieces (fused.h kF16WScale)
  float pscale = 1.f;         // factor on P_v, P_q as stored (fused path: kPScale, fused.h)
};

int launch_proj(const Ctx& c, const coattn_gemm_desc& g) {
  return c.bf16_proj ? launch_gemm_bf16in(g, c.s) : launch_gemm_f32(g, c.s);
}

// P_v = V W_v^T + b_v   (model.py:380/384, evaluated once per sample)
int proj_v(const Ctx& c, const float* V, const float* Wv, const float* bv, float* Pv) {
  coattn_gemm_desc g = {};
  g.A = V; g.B = Wv; g.C = Pv; g.bias_n = bv; g.out_scale = c.pscale;
  g.M = c.B * c.N; g.N = c.d; g.K = c.d; g.batch = 1;
  g.a_sm = c.vl.sN; g.a_sk = c.vl.sD;            // row m = (b, n): split rows unless the samples abut
  if (c.vl.sB != (int64_t)c.N * c.vl.sN) { g.a_mdiv = c.N; g.a_sdiv = c.vl.sB; }
  g.b_sk = 1; g.b_sn = c.d;
  g.c_sm = c.d; g.c_sn = 1;
  return launch_proj(c, g);
}
// P_q = Q W_q^T + b_q   (model.py:381/383)
int proj_q(const Ctx& c, const float* Q, const float* Wq, const float* bq, float* Pq) {
  coattn_gemm_desc g = {};
  g.A = Q; g.B = Wq; g.C = Pq; g.bias_n = bq;
  g.M = c.B * c.T; g.N = c.d; g.K = c.d; g.batch = 1;
  g.a_sm = c.d; g.a_sk = 1;
  g.b_sk = 1; g.b_sn = c.d;
  g.c_sm = c.d; g.c_sn = 1;
  return launch_proj(c, g);
}
// C = tanh(Q V)   (model.py:377)
int affinity(const Ctx& c, const float* Q, const float* V, float* C) {
  coattn_gemm_desc g = {};
  g.A = Q; g.a_sz = (int64_t)c.T * c.d; g.a_sm = c.d; g.a_sk = 1;
  g.B = V; g.b_sz = c.vl.sB; g.b_sk = c.vl.sD; g.b_sn = c.vl.sN;
  g.C = C; g.c_sz = (int64_t)c.T * c.N; g.c_sm = c.N; g.c_sn = 1;
  g.M = c.T; g.N = c.N; g.K = c.d; g.batch = c.B; g.act = 1;
  return launch_gemm_f32(g, c.s);
}
// out = act(X + C^T Y): X,out [B,N,d], Y [B,T,d]   (H_v, model.py:380-381; dP_v in backward)
int ct_times(const Ctx& c, const float* C, const float* Y, const float* X, float* out, int act) {
  coattn_gemm_desc g = {};
  g.A = C; g.a_sz = (int64_t)c.T * c.N; g.a_sm = 1; g.a_sk = c.N;
  g.B = Y; g.b_sz = (int64_t)c.T * c.d; g.b_sk = c.d; g.b_sn = 1;
  g.Cin = X; g.cin_sz = (int64_t)c.N * c.d; g.cin_sm = c.d; g.cin_sn = 1; g.beta = 1.f;
  g.C = out; g.c_sz = (int64_t)c.N * c.d; g.c_sm = c.d; g.c_sn = 1;
  g.M = c.N; g.N = c.d; g.K = c.T; g.batch = c.B; g.act = act;
  return launch_gemm_f32(g, c.s);
}
// out = act(X + C Y): X,out [B,T,d], Y [B,N,d]   (H_q, model.py:383-384; dP_q in backward)
int c_times(const Ctx& c, const float* C, const float* Y, const float* X, float* out, int act) {
  coattn_gemm_desc g = {};
  g.A = C; g.a_sz = (int64_t)c.T * c.N; g.a_sm = c.N; g.a_sk = 1;
  g.B = Y; g.b_sz = (int64_t)c.N * c.d; g.b_sk = c.d; g.b_sn = 1;
  g.Cin = X; g.cin_sz = (int64_t)c.T * c.d; g.cin_sm = c.d; g.cin_sn = 1; g.beta = 1.f;
  g.C = out; g.c_sz = (int64_t)c.T * c.d; g.c_sm = c.d; g.c_sn = 1;
  g.M = c.T; g.N = c.d; g.K = c.N; g.batch = c.B; g.act = act;
  return launch_gemm_f32(g, c.s);
}

// Width of the fp32 mode's contractions (fused.h, include/coattn.h "Widths of the fp32 mode").  flags = 0: every
// contraction on the exact three-piece split.  COATTN_FLAG_FAST16: the forward-side contractions on two FP16 pieces, the
// backward's on two bf16 pieces.  Developer switches (builds with -DCOATTN_DEV_SWITCHES only): COATTN_SPLIT=3 forces the exact
// split, COATTN_FWD_F16=0 / COATTN_FWD_F16_KERNEL=0 / COATTN_SPLIT_FWD=3 / COATTN_SPLIT_PQ=3 the bf16 widths of round 4's A/B runs.
static int env_int(const char* name, int dflt) { return dev_env_int(name, dflt); }
static bool fast16(int flags) {
  static const int split = env_int("COATTN_SPLIT", 2);
  return (flags & COATTN_FLAG_FAST16) && !(flags & COATTN_FLAG_EXACT3) && split != 3;
}
static int np_bwd(int flags) { return fast16(flags) ? 2 : 3; }
static int np_projq(int flags) {                    // P_q = Q W_q^T: its error reaches H_v summed over T <= 28 tokens only
  static const int pq = env_int("COATTN_SPLIT_PQ", 2);
  return fast16(flags) ? (pq == 2 ? 2 : 3) : 3;
}
// The forward's projections P_v, P_q on two FP16 pieces (fused.h; tests/test_split_emulation.py: less error than the exact
// split of P_v next to two bf16 pieces of P_q, at half the MFMAs of the former).
static bool f16_fwd(int flags) {
  static const int on = env_int("COATTN_FWD_F16", 1);
  return fast16(flags) && on != 0;
}
static int np_fwd(int flags, bool f16) {            // 4: both phases of the forward kernel on two FP16 pieces (coattn_fwd32.hip)
  static const int fwd = env_int("COATTN_SPLIT_FWD", 2), hk = env_int("COATTN_FWD_F16_KERNEL", 1);
  if (!fast16(flags)) return 3;
  if (f16) return (hk != 0 && fwd == 2) ? 4 : (fwd == 2 ? 2 : 3);
  return f16_fwd(flags) ? 3 : (fwd == 2 ? 2 : 3);   // FP16 pieces wanted but not available for this shape: exact
}

// COATTN_GEMM_W=0 (developer switch): the projections through gemm.hip instead of the pre-split-weight kernel
static bool gemm_w_enabled() {
  static const int on = dev_env_int("COATTN_GEMM_W", 1);
  return on != 0;
}

// The two projection jobs of a forward call on the pre-split-weight kernel (gemm_w.hip): P_v from the image features in
// either layout, P_q of all levels from the pointer table.  Returns through v_w / q_w which of them that kernel takes.
void projection_jobs(const Ctx& c, const float* V, const float* const* Q, const coattn_params* p, float* sv, char* wimg,
                     WGemm& wv, WGemm& wq, bool& v_w, bool& q_w, bool want_f16) {
  const SavedPlan sp = plan_saved(c.B, c.N, c.T, c.d, c.L);
  const size_t BTd = (size_t)c.B * c.T * c.d;
  wv = WGemm{}; wq = WGemm{};
  wv.A = V; wv.a_sm = (int)c.vl.sN; wv.Wf = wimg; wv.C = sv + sp.Pv; wv.c_sm = c.d;
  wv.bias_n = p ? (const float*)p->b_v : nullptr; wv.out_scale = c.pscale; wv.M = c.B * c.N; wv.N = c.d; wv.K = c.d; wv.batch = 1;
  for (int l = 0; l < c.L; ++l) wq.a_ptrs[l] = Q[l];
  wq.a_sm = c.d; wq.Wf = wimg + wsplit_bytes(c.d, c.d); wq.C = sv + sp.Pq; wq.c_sz = (long)BTd; wq.c_sm = c.d;
  wq.bias_n = p ? (const float*)p->b_q : nullptr; wq.out_scale = c.pscale; wq.M = c.B * c.T; wq.N = c.d; wq.K = c.d; wq.batch = c.L;
  wv.bf16 = wq.bf16 = c.bf16_proj ? 1 : 0;          // reduced precision: the same kernels, hi pieces only, one MFMA per product
  wv.np = 3; wq.np = c.np_pq;
  if (want_f16 && !c.bf16_proj) { wv.np = wq.np = 2; wv.f16 = wq.f16 = 1; }   // two FP16 pieces (any M runs on gemm_w then)
  const bool w_ok = gemm_w_enabled();
  v_w = false;
  if (w_ok && c.vl.sD == 1 && c.vl.sB == (long)c.N * c.vl.sN && c.vl.sN < (1L << 24)) {
    v_w = gemm_w_supported(wv) != 0;                 // location-major rows, samples abutting
  } else if (w_ok && c.vl.sN == 1 && c.vl.sD < (1L << 24)) {
    wv.a_sm = 0; wv.a_sk = (int)c.vl.sD; wv.a_mdiv = c.N; wv.a_sdiv = c.vl.sB;   // channel-major, read in place
    v_w = gemm_w_supported(wv) != 0;
  }
  q_w = w_ok && gemm_w_supported(wq);
}
// Tolerance mode: FP16 pieces are used only when BOTH projections run on the pre-split-weight kernel -- it is that launch
// which range-checks the image and question features and the stored projections for the fused kernel behind it
// (coattn_status); any other shape computes exactly.
bool f16_path(const Ctx& c, const float* V, const float* const* Q, int flags, int fused) {
  if (!fused || c.bf16_proj || !f16_fwd(flags)) return false;
  WGemm wv, wq;
  bool v_w, q_w;
  projection_jobs(c, V, Q, nullptr, nullptr, nullptr, wv, wq, v_w, q_w, true);
  return v_w && q_w;
}

// Does the forward write the bitmap of the live question rows (and run P_q over them alone)?  A pure function of the call's
// shapes, pointers and mode: the backward asks the same question about the same call (rowbits_in_saved).
static bool rowbits_predicate(const Ctx& c, const WGemm& wq, const coattn_params* p, const float* sv, bool f16) {
  static const int rows_env = dev_env_int("COATTN_SKIP_ZERO_ROWS", 1);   // developer switch
  const SavedPlan sp = plan_saved(c.B, c.N, c.T, c.d, c.L);
  return rows_env && !f16 && !c.bf16_proj && gemm_wx_kernel(wq) == 0 && wq.a_sk == 0 && p->b_q &&
         (wq.M + 31) / 32 <= kRowBitsMaxWords && c.d % 256 == 0 && c.d <= 1024 &&
         ((((uintptr_t)p->b_q) | ((uintptr_t)(sv + sp.Pq))) & 15) == 0;      // (16-byte accesses of the flag job)
}
bool rowbits_in_saved(const Ctx& c, const float* V, const float* const* Q, const coattn_params* p, const float* sv, int flags, int fused) {
  WGemm wv, wq;
  bool v_w, q_w;
  Ctx cc = c;
  cc.f16_proj = f16_path(c, V, Q, flags, fused);
  const bool f16 = cc.f16_proj && !cc.bf16_proj;
  projection_jobs(cc, V, Q, p, const_cast<float*>(sv), nullptr, wv, wq, v_w, q_w, f16);
  return q_w && rowbits_predicate(cc, wq, p, sv, f16);
}

// wimg: room for two pre-split weight images (wsplit_bytes(d, d) each) at the end of the forward workspace
// keep_wqT: also split W_q the other way round into the saved state (sp.wqT) for the backward's dQ projection
int general_projections(const Ctx& c, const float* V, const float* const* Q, const coattn_params* p, float* sv, char* wimg,
                        bool keep_wqT) {
  const SavedPlan sp = plan_saved(c.B, c.N, c.T, c.d, c.L);
  const size_t BTd = (size_t)c.B * c.T * c.d;
  // fp32 projections of row-major activations: the weight is split once, the GEMM reads it as MFMA fragments
  WGemm wv, wq;
  bool v_w, q_w;
  const bool f16 = c.f16_proj && !c.bf16_proj;        // (forward_impl: only when both projections run on gemm_w)
  projection_jobs(c, V, Q, p, sv, wimg, wv, wq, v_w, q_w, f16);
  float* status = sv + sp.status;
  if (f16) wv.status = wq.status = status;            // both on two FP16 pieces, range-checked
  RowFlagJob rj = {};
  bool rows_in_gemm = false;
  if (v_w || q_w) {
    WSplit jobs[3];
    int nj = 0;
    // (the image of W_q^T is read by the backward's dQ projection: a GEMM of P_q's shape with row-major A in the same
    //  precision mode, so it runs on the same kernel as P_q and wants the same image format)
    const int chunks = ((c.d + 31) / 32) * ((c.d + 15) / 16);
    if (v_w) jobs[nj++] = WSplit{(const float*)p->W_v, const_cast<void*>(wv.Wf), c.d, c.d, 0, c.d, wimg_pieces(wv), status + kStatusHdr};
    if (q_w) jobs[nj++] = WSplit{(const float*)p->W_q, const_cast<void*>(wq.Wf), c.d, c.d, 0, c.d, wimg_pieces(wq), status + kStatusHdr + chunks};
    WGemm wqb = wq;                                   // (the backward's operands are gradients: bf16 pieces, fused.h)
    wqb.f16 = 0;
    if (q_w && keep_wqT) jobs[nj++] = WSplit{(const float*)p->W_q, sv + sp.wqT, c.d, c.d, 1, c.d, wimg_pieces(wqb), nullptr};
    // Question rows of exact zeros -- the pad tokens (model.py:263, :292-296) -- project to the bias alone: the launch's extra
    // workgroups flag the rows of Q_l that hold anything, write (0 + b_q) * scale into the others' rows of P_q, and the exact
    // four-wave GEMM runs over the flagged rows only (same values bit for bit; 44 % fewer rows on BASELINE's synthetic
    // questions, lengths U{3..26} of 26).  Row-major A on gemm_w_kernel only: the other kernels compute every row.
    const bool skip_rows = q_w && rowbits_predicate(c, wq, p, sv, f16);
    if (skip_rows) {
      unsigned* bits = reinterpret_cast<unsigned*>(sv + sp.rowbits);        // (kept in `saved`: the backward's dW_q reads it)
      for (int l = 0; l < c.L; ++l) rj.a_ptrs[l] = Q[l];
      rj.a_sm = c.d; rj.C = sv + sp.Pq; rj.c_sz = (long)BTd; rj.c_sm = c.d;
      rj.bias_n = (const float*)p->b_q; rj.out_scale = c.pscale; rj.M = c.B * c.T; rj.N = c.d; rj.K = c.d; rj.batch = c.L;
      rj.rowbits = bits;
      wq.rowbits = bits;
      // COATTN_FLAGS_IN_GEMM=1 (developer switch, DEV builds): the flag workgroups at the head of the PROJECTION launch instead,
      // its P_q tiles waiting on a counter the weight-split launch zeroes.  Measured and not adopted: the weight-split launch
      // gets its 6.6 us back, the projection launch loses 12.5 at N = 49 (392 flag workgroups hold the slots the first tiles
      // want) and 5 at N = 196 (LAB_NOTES A6.3)
      static const int in_gemm_env = dev_env_int("COATTN_FLAGS_IN_GEMM", 0);
      if (in_gemm_env) rj.rowcnt = reinterpret_cast<unsigned*>(sv + sp.rowcnt);
    }
    rows_in_gemm = skip_rows && rj.rowcnt != nullptr;
    CA_TRY(launch_wsplit(jobs, nj, c.s, status, f16 ? 1 : 0, (skip_rows && !rows_in_gemm) ? &rj : nullptr,
                         rows_in_gemm ? rj.rowcnt : nullptr));   // (also writes the header of the call's status words)
    prof_mark(c.s, "wsplit");
  } else {                                            // no weight-split launch on this path: header = "no FP16 pieces"
    if (hipMemsetAsync(status, 0, 2 * sizeof(float), c.s) != hipSuccess) { coattn_set_error("forward: hipMemsetAsync failed"); return -3; }
  }
  if (v_w && q_w) {                                   // both projections in one launch
    const WGemm both[2] = {wv, wq};
    CA_TRY(launch_gemm_wx(both, 2, c.s, rows_in_gemm ? &rj : nullptr));
    prof_mark(c.s, "projections");
    return 0;
  }
  if (v_w) CA_TRY(launch_gemm_wx(&wv, 1, c.s));
  else CA_TRY(proj_v(c, V, (const float*)p->W_v, (const float*)p->b_v, sv + sp.Pv));
  if (q_w) return launch_gemm_wx(&wq, 1, c.s, rows_in_gemm ? &rj : nullptr);
  // P_q of all levels in one launch: batch z = level, A from the pointer table
  coattn_gemm_desc g = {};
  for (int l = 0; l < c.L; ++l) g.a_ptrs[l] = Q[l];
  g.B = p->W_q; g.C = sv + sp.Pq; g.c_sz = (int64_t)BTd; g.bias_n = p->b_q; g.out_scale = c.pscale;
  g.M = c.B * c.T; g.N = c.d; g.K = c.d; g.batch = c.L;
  g.a_sm = c.d; g.a_sk = 1;
  g.b_sk = 1; g.b_sn = c.d;
  g.c_sm = c.d; g.c_sn = 1;
  return launch_proj(c, g);
}

// everything after the projections: affinity, H_v / H_q, scores, softmax, attended reductions
int general_attention(const Ctx& c, const float* V, const float* const* Q, const coattn_params* p, float* v_out,
                      float* q_out, float* sv, float* Hv) {
  const SavedPlan sp = plan_saved(c.B, c.N, c.T, c.d, c.L);
  float* Pv = sv + sp.Pv;
  const size_t BTd = (size_t)c.B * c.T * c.d, BTN = (size_t)c.B * c.T * c.N;
  for (int l = 0; l < c.L; ++l) {
    float* Pq = sv + sp.Pq + l * BTd;
    float* C = sv + sp.C + l * BTN;
    float* Hq = sv + sp.Hq + l * BTd;
    float* av = sv + sp.av + (size_t)l * c.B * c.N;
    float* aq = sv + sp.aq + (size_t)l * c.B * c.T;
    CA_TRY(affinity(c, Q[l], V, C));
    CA_TRY(ct_times(c, C, Pq, Pv, Hv, 1));
    CA_TRY(c_times(c, C, Pv, Pq, Hq, 1));
    CA_TRY(launch_score_softmax(Hv, (const float*)p->w_v, (const float*)p->c_v, av, c.B, c.N, c.d, c.s));
    CA_TRY(launch_score_softmax(Hq, (const float*)p->w_q, (const float*)p->c_q, aq, c.B, c.T, c.d, c.s));
    // v = sum_n a_v[n] V[:,n]   (model.py:391);  q = sum_t a_q[t] Q[t,:]   (model.py:392)
    CA_TRY(launch_gemv(V, av, v_out + (size_t)l * c.B * c.d, c.B, c.d, c.N, c.vl.sB, c.vl.sD, c.vl.sN, c.N, c.d, c.s));
    CA_TRY(launch_gemv(Q[l], aq, q_out + (size_t)l * c.B * c.d, c.B, c.d, c.T, (int64_t)c.T * c.d, 1, c.d, c.T, c.d, c.s));
  }
  return 0;
}

int backward_general(const Ctx& c, const float* V, const float* const* Q, const coattn_params* p, const float* sv,
                     const float* gv, const float* gq, float* dV, const VLayout& dvl, float* const* dQ,
                     const coattn_param_grads* pg, int accumulate, float* ws) {
  const SavedPlan sp = plan_saved(c.B, c.N, c.T, c.d, c.L);
  const BwdPlan bp = plan_bwd(c.B, c.N, c.T, c.d, c.L);
  const int B = c.B, N = c.N, T = c.T, d = c.d, L = c.L;
  const size_t BTd = (size_t)B * T * d, BTN = (size_t)B * T * N, BNd = (size_t)B * N * d, Bd = (size_t)B * d;
  const float* Pv = sv + sp.Pv;
  const float* wv = (const float*)p->w_v;
  const float* wq = (const float*)p->w_q;
  float* Hv = ws + bp.Hv;
  float* dPv = ws + bp.dPv;
  float* dZq = ws + bp.dZq;
  float* dC = ws + bp.dC;
  float* dav = ws + bp.dav;
  float* dsv = ws + bp.dsv;
  float* daq = ws + bp.daq;
  float* dsq = ws + bp.dsq;
  float* part = ws + bp.part;
  int nch = 0;
  for (int l = 0; l < L; ++l) {
    const float* Pq = sv + sp.Pq + l * BTd;
    const float* C = sv + sp.C + l * BTN;
    const float* Hq = sv + sp.Hq + l * BTd;
    const float* av = sv + sp.av + (size_t)l * B * N;
    const float* aq = sv + sp.aq + (size_t)l * B * T;
    float* dPq = ws + bp.dPq + l * BTd;
    const int acc_l = (accumulate || l > 0) ? 1 : 0;
    // recompute H_v = tanh(P_v + C^T P_q)
    CA_TRY(ct_times(c, C, Pq, Pv, Hv, 1));
    // softmax backward of a_v, a_q
    CA_TRY(launch_gemv(V, gv + l * Bd, dav, B, N, d, c.vl.sB, c.vl.sN, c.vl.sD, d, N, c.s));
    CA_TRY(launch_softmax_bwd(av, dav, dsv, B, N, c.s));
    CA_TRY(launch_gemv(Q[l], gq + l * Bd, daq, B, T, d, (int64_t)T * d, d, 1, d, T, c.s));
    CA_TRY(launch_softmax_bwd(aq, daq, dsq, B, T, c.s));
    // dw_v += ds_v^T H_v ; dc_v += sum ds_v ; same for q
    const int rpc_v = (B * N + 255) / 256 > 32 ? (B * N + 255) / 256 : 32;
    CA_TRY(launch_colsum_partial(dsv, Hv, part, B * N, d, rpc_v, &nch, c.s));
    CA_TRY(launch_reduce_partials(part, (float*)pg->dw_v, nch, d, acc_l, c.s));
    CA_TRY(launch_sum_all(dsv, (float*)pg->dc_v, (int64_t)B * N, acc_l, c.s));
    const int rpc_q = (B * T + 255) / 256 > 32 ? (B * T + 255) / 256 : 32;
    CA_TRY(launch_colsum_partial(dsq, Hq, part, B * T, d, rpc_q, &nch, c.s));
    CA_TRY(launch_reduce_partials(part, (float*)pg->dw_q, nch, d, acc_l, c.s));
    CA_TRY(launch_sum_all(dsq, (float*)pg->dc_q, (int64_t)B * T, acc_l, c.s));
    // dZ_v (in place over H_v), dZ_q
    CA_TRY(launch_dz(dsv, wv, Hv, Hv, (int64_t)B * N, d, c.s));
    CA_TRY(launch_dz(dsq, wq, Hq, dZq, (int64_t)B * T, d, c.s));
    // dP_q = dZ_q + C dZ_v
    CA_TRY(c_times(c, C, Hv, dZq, dPq, 0));
    // dC = P_q dZ_v^T + dZ_q P_v^T ; dA = dC (1 - C^2)
    {
      coattn_gemm_desc g = {};
      g.A = Pq; g.a_sz = (int64_t)T * d; g.a_sm = d; g.a_sk = 1;
      g.B = Hv; g.b_sz = (int64_t)N * d; g.b_sk = 1; g.b_sn = d;
      g.C = dC; g.c_sz = (int64_t)T * N; g.c_sm = N; g.c_sn = 1;
      g.M = T; g.N = N; g.K = d; g.batch = B;
      CA_TRY(launch_gemm_f32(g, c.s));
      g.A = dZq; g.B = Pv; g.Cin = dC; g.cin_sz = g.c_sz; g.cin_sm = N; g.cin_sn = 1; g.beta = 1.f;
      CA_TRY(launch_gemm_f32(g, c.s));
    }
    CA_TRY(launch_dtanh(dC, C, dC, (int64_t)BTN, c.s));
    // dP_v(level) = dZ_v + C^T dZ_q   (in place), accumulate over levels
    CA_TRY(ct_times(c, C, dZq, Hv, Hv, 0));
    CA_TRY(launch_add_inplace(dPv, Hv, (int64_t)BNd, l > 0 ? 1 : 0, c.s));
    // dQ_l = a_q (x) gq + dA V^T   (+ dP_q W_q below)
    CA_TRY(launch_rank1(aq, gq + l * Bd, dQ[l], B, T, d, (int64_t)T * d, d, 1, 0, c.s));
    {
      coattn_gemm_desc g = {};
      g.A = dC; g.a_sz = (int64_t)T * N; g.a_sm = N; g.a_sk = 1;
      g.B = V; g.b_sz = c.vl.sB; g.b_sk = c.vl.sN; g.b_sn = c.vl.sD;
      g.Cin = dQ[l]; g.cin_sz = (int64_t)T * d; g.cin_sm = d; g.cin_sn = 1; g.beta = 1.f;
      g.C = dQ[l]; g.c_sz = (int64_t)T * d; g.c_sm = d; g.c_sn = 1;
      g.M = T; g.N = d; g.K = N; g.batch = B;
      CA_TRY(launch_gemm_f32(g, c.s));
    }
    // dV (+)= a_v (x) gv + Q^T dA
    if (dV) CA_TRY(launch_rank1(av, gv + l * Bd, dV, B, N, d, dvl.sB, dvl.sN, dvl.sD, l > 0 ? 1 : 0, c.s));
    if (dV) {
      coattn_gemm_desc g = {};
      g.A = Q[l]; g.a_sz = (int64_t)T * d; g.a_sm = 1; g.a_sk = d;
      g.B = dC; g.b_sz = (int64_t)T * N; g.b_sk = N; g.b_sn = 1;
      g.Cin = dV; g.cin_sz = dvl.sB; g.cin_sm = dvl.sD; g.cin_sn = dvl.sN; g.beta = 1.f;
      g.C = dV; g.c_sz = dvl.sB; g.c_sm = dvl.sD; g.c_sn = dvl.sN;
      g.M = d; g.N = N; g.K = T; g.batch = B;
      CA_TRY(launch_gemm_f32(g, c.s));
    }
  }
  // projections backward
  for (int l = 0; l < L; ++l) {
    const float* dPq = ws + bp.dPq + l * BTd;
    coattn_gemm_desc g = {};
    g.A = dPq; g.a_sm = d; g.a_sk = 1;
    g.B = p->W_q; g.b_sk = d; g.b_sn = 1;
    g.Cin = dQ[l]; g.cin_sm = d; g.cin_sn = 1; g.beta = 1.f;
    g.C = dQ[l]; g.c_sm = d; g.c_sn = 1;
    g.M = B * T; g.N = d; g.K = d; g.batch = 1;
    CA_TRY(c.bf16_proj ? launch_gemm_bf16in(g, c.s) : launch_gemm_f32(g, c.s));
  }
  if (dV) {
    // dV[b][k][n] += sum_j W_v[j][k] dP_v[b][n][j]
    coattn_gemm_desc g = {};
    g.A = p->W_v; g.a_sm = 1; g.a_sk = d; g.a_sz = 0;
    g.B = dPv; g.b_sz = (int64_t)N * d; g.b_sk = 1; g.b_sn = d;
    g.Cin = dV; g.cin_sz = dvl.sB; g.cin_sm = dvl.sD; g.cin_sn = dvl.sN; g.beta = 1.f;
    g.C = dV; g.c_sz = dvl.sB; g.c_sm = dvl.sD; g.c_sn = dvl.sN;
    g.M = d; g.N = N; g.K = d; g.batch = B;
    CA_TRY(c.bf16_proj ? launch_gemm_bf16in(g, c.s) : launch_gemm_f32(g, c.s));
  }
  {
    // dW_v[j][k] = sum_b sum_n dP_v[b][n][j] V[b][k][n]  -> split over sample groups
    const int G = (B + kMaxSplits - 1) / kMaxSplits;
    const int S = (B + G - 1) / G;
    coattn_gemm_desc g = {};
    g.A = dPv; g.a_sm = 1; g.a_sk = d; g.a_si = (int64_t)N * d; g.a_sz = (int64_t)G * N * d;
    g.B = V; g.b_sk = c.vl.sN; g.b_sn = c.vl.sD; g.b_si = c.vl.sB; g.b_sz = (int64_t)G * c.vl.sB;
    g.C = part; g.c_sz = (int64_t)d * d; g.c_sm = d; g.c_sn = 1;
    g.M = d; g.N = d; g.K = N; g.batch = S; g.inner = G; g.inner_total = B;
    CA_TRY(c.bf16_proj ? launch_gemm_bf16in(g, c.s) : launch_gemm_f32(g, c.s));
    CA_TRY(launch_reduce_partials(part, (float*)pg->dW_v, S, (int64_t)d * d, accumulate, c.s));
    const int rpc = (B * N + 255) / 256 > 32 ? (B * N + 255) / 256 : 32;
    CA_TRY(launch_colsum_partial(nullptr, dPv, part, B * N, d, rpc, &nch, c.s));
    CA_TRY(launch_reduce_partials(part, (float*)pg->db_v, nch, d, accumulate, c.s));
  }
  for (int l = 0; l < L; ++l) {
    // dW_q[j][k] += sum_m dP_q[m][j] Q_l[m][k],  m over (b,t): split-K
    const float* dPq = ws + bp.dPq + l * BTd;
    const int K = B * T;
    int ks = (K + kMaxSplits - 1) / kMaxSplits;
    ks = (ks + 15) / 16 * 16;
    const int S = (K + ks - 1) / ks;
    coattn_gemm_desc g = {};
    g.A = dPq; g.a_sm = 1; g.a_sk = d;
    g.B = Q[l]; g.b_sk = d; g.b_sn = 1;
    g.C = part; g.c_sz = (int64_t)d * d; g.c_sm = d; g.c_sn = 1;
    g.M = d; g.N = d; g.K = K; g.batch = S; g.ksplit = ks;
    CA_TRY(c.bf16_proj ? launch_gemm_bf16in(g, c.s) : launch_gemm_f32(g, c.s));
    CA_TRY(launch_reduce_partials(part, (float*)pg->dW_q, S, (int64_t)d * d, (accumulate || l > 0) ? 1 : 0, c.s));
  }
  {
    const int R = L * B * T;
    const int rpc = (R + 255) / 256 > 32 ? (R + 255) / 256 : 32;
    CA_TRY(launch_colsum_partial(nullptr, ws + bp.dPq, part, R, d, rpc, &nch, c.s));
    CA_TRY(launch_reduce_partials(part, (float*)pg->db_q, nch, d, accumulate, c.s));
  }
  return 0;
}

int pick_impl(int flags, int B, int N, int T, int d, int L, const VLayout& vl, int* fused) {
  const int sel = flags & 3;
  const int ok = fused_supported(B, N, T, d, L) && fused_layout_ok(vl, N, d);
  if (sel == COATTN_IMPL_FUSED) {
    CA_CHECK_ARG(ok, "fused kernels do not support B=%d N=%d T=%d d=%d L=%d with V strides (%ld, %ld, %ld)", B, N, T, d, L,
                 vl.sB, vl.sN, vl.sD);
    *fused = 1;
  } else if (sel == COATTN_IMPL_GENERAL) {
    *fused = 0;
  } else {
    *fused = ok;
  }
  return 0;
}

}  // namespace

static int check_vlayout(const VLayout& v, int B, int N, int d, const char* what) {
  (void)B;
  CA_CHECK_ARG(v.sN > 0 && v.sD > 0 && v.sB > 0, "%s: strides must be positive (sB=%ld sN=%ld sD=%ld)", what, v.sB, v.sN, v.sD);
  // the extent of one sample must not reach into the next one
  CA_CHECK_ARG((long)(N - 1) * v.sN + (long)(d - 1) * v.sD < v.sB, "%s: sample stride %ld is smaller than a sample's extent", what, v.sB);
  return 0;
}

static int forward_impl(const void* V, const VLayout& vl, const void* const* Q, const coattn_params* p, void* v_out,
                        void* q_out, void* saved, void* ws, int B, int N, int T, int d, int L, int dtype, int flags,
                        void* stream, bool do_proj, bool do_attn) {
  CA_TRY(check_shape(B, N, T, d, L, dtype));
  CA_CHECK_ARG(V && Q && p && v_out && q_out && ws, "forward: null argument");
  CA_TRY(check_vlayout(vl, B, N, d, "forward: V"));
  for (int l = 0; l < L; ++l) CA_CHECK_ARG(Q[l] != nullptr, "forward: Q[%d] is null", l);
  CA_CHECK_ARG(p->W_v && p->b_v && p->W_q && p->b_q && p->w_v && p->c_v && p->w_q && p->c_q,
               "forward: null parameter pointer");
  int fused = 0;
  CA_TRY(pick_impl(flags, B, N, T, d, L, vl, &fused));
  const SavedPlan sp = plan_saved(B, N, T, d, L);
  float* sv = saved ? (float*)saved : (float*)ws;      // inference: state lives in the workspace
  float* tail = (float*)ws + sp.total;
  Ctx c{B, N, T, d, L, (hipStream_t)stream, vl};
  c.bf16_proj = (flags & COATTN_FLAG_BF16_PROJ) != 0;
  c.f16_proj = f16_path(c, (const float*)V, (const float* const*)Q, flags, fused);
  // (the general-shape path stays exact throughout; a fused shape without the FP16 path too -- unless the developer switch
  //  COATTN_FWD_F16=0 asks for round 4's bf16 widths)
  c.np_pq = (fused && fast16(flags) && !f16_fwd(flags)) ? np_projq(flags) : 3;
  c.pscale = fused ? kPScale : 1.f;
  if (do_proj)
    CA_TRY(general_projections(c, (const float*)V, (const float* const*)Q, p, sv,
                               (char*)ws + fwd_ws_floats(B, N, T, d, L) * sizeof(float), saved != nullptr));
  if (!do_attn) return 0;
  if (fused)
    return fused_attention_forward(B, N, T, d, L, (const float*)V, vl, (const float* const*)Q, p, (float*)v_out,
                                   (float*)q_out, sv, tail, c.s, c.bf16_proj ? 1 : 0, np_fwd(flags, c.f16_proj));
  return general_attention(c, (const float*)V, (const float* const*)Q, p, (float*)v_out, (float*)q_out, sv, tail);
}

extern "C" int coattn_forward(const void* V, int64_t v_sB, int64_t v_sN, int64_t v_sD, const void* const* Q,
                              const coattn_params* p, void* v_out, void* q_out, void* saved, void* ws, int B, int N,
                              int T, int d, int L, int dtype, int flags, void* stream) {
  return forward_impl(V, VLayout{(long)v_sB, (long)v_sN, (long)v_sD}, Q, p, v_out, q_out, saved, ws, B, N, T, d, L,
                      dtype, flags, stream, true, true);
}

extern "C" int coattn_attention_forward(const void* V, int64_t v_sB, int64_t v_sN, int64_t v_sD, const void* const* Q,
                                        const coattn_params* p, void* v_out, void* q_out, void* saved, void* ws,
                                        int B, int N, int T, int d, int L, int dtype, int flags, void* stream) {
  CA_CHECK_ARG(saved != nullptr, "attention_forward: needs the saved buffer of a previous coattn_forward");
  return forward_impl(V, VLayout{(long)v_sB, (long)v_sN, (long)v_sD}, Q, p, v_out, q_out, saved, ws, B, N, T, d, L,
                      dtype, flags, stream, false, true);
}

extern "C" int coattn_backward(const void* V, int64_t v_sB, int64_t v_sN, int64_t v_sD, const void* const* Q,
                               const coattn_params* p, const void* saved, const void* gv, const void* gq, void* dV,
                               int64_t dv_sB, int64_t dv_sN, int64_t dv_sD, void* const* dQ,
                               const coattn_param_grads* pg, int accumulate, void* ws, int B, int N, int T, int d,
                               int L, int dtype, int flags, void* stream) {
  CA_TRY(check_shape(B, N, T, d, L, dtype));
  CA_CHECK_ARG(V && Q && p && saved && gv && gq && dQ && pg && ws, "backward: null argument");  // dV may be NULL
  const VLayout vl{(long)v_sB, (long)v_sN, (long)v_sD};
  VLayout dvl{(long)dv_sB, (long)dv_sN, (long)dv_sD};
  CA_TRY(check_vlayout(vl, B, N, d, "backward: V"));
  if (dV) CA_TRY(check_vlayout(dvl, B, N, d, "backward: dV"));
  else dvl = vl;
  for (int l = 0; l < L; ++l) CA_CHECK_ARG(Q[l] && dQ[l], "backward: Q[%d]/dQ[%d] is null", l, l);
  CA_CHECK_ARG(pg->dW_v && pg->db_v && pg->dW_q && pg->db_q && pg->dw_v && pg->dc_v && pg->dw_q && pg->dc_q,
               "backward: null parameter-gradient pointer");
  int fused = 0;
  CA_TRY(pick_impl(flags, B, N, T, d, L, vl, &fused));
  Ctx c{B, N, T, d, L, (hipStream_t)stream, vl};
  c.bf16_proj = (flags & COATTN_FLAG_BF16_PROJ) != 0;
  // (`saved` of the fused forward holds P_v, P_q scaled by kPScale: only the fused backward may read it)
  CA_CHECK_ARG(!fused || fused_backward_supported(B, N, T, d, L), "backward: the fused forward's saved state has no fused backward for this shape");
  if (fused)
    return fused_backward(B, N, T, d, L, (const float*)V, vl, (const float* const*)Q, p, (const float*)saved,
                          (const float*)gv, (const float*)gq, (float*)dV, dvl, (float* const*)dQ, pg, accumulate,
                          (float*)ws, c.s, c.bf16_proj ? 1 : 0,
                          gemm_w_enabled() ? 1 : 0, np_bwd(flags),
                          rowbits_in_saved(c, (const float*)V, (const float* const*)Q, p, (const float*)saved, flags, fused) ? 1 : 0);
  return backward_general(c, (const float*)V, (const float* const*)Q, p, (const float*)saved, (const float*)gv,
                          (const float*)gq, (float*)dV, dvl, (float* const*)dQ, pg, accumulate, (float*)ws);
}
