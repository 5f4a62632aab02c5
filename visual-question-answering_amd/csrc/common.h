// Shared device/host helpers for the gfx950 co-attention kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <mutex>

#include "../../include/coattn.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- error plumbing (thread-local message, negative return codes) -----------------------
void coattn_set_error(const char* fmt, ...);

#define CA_CHECK_ARG(cond, ...)                 \
  do {                                          \
    if (!(cond)) {                              \
      coattn_set_error(__VA_ARGS__);            \
      return -1;                                \
    }                                           \
  } while (0)

#define CA_CHECK_LAUNCH(what)                                                        \
  do {                                                                               \
    hipError_t e_ = hipGetLastError();                                               \
    if (e_ != hipSuccess) {                                                          \
      coattn_set_error("%s: launch failed: %s", what, hipGetErrorString(e_));        \
      return -3;                                                                     \
    }                                                                                \
  } while (0)

#define CA_TRY(expr)            \
  do {                          \
    int rc_ = (expr);           \
    if (rc_ != 0) return rc_;   \
  } while (0)

// One-time, per-device setup (hipFuncSetAttribute is a per-device property): thread-safe, keyed by the calling
// thread's current device.  run() returns 0, or -3 with the error message set.
struct DeviceOnce {
  static constexpr int kMaxDev = 64;
  std::once_flag flag[kMaxDev];
  hipError_t err[kMaxDev];
  template <typename F>
  int run(F f, const char* what) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e == hipSuccess) {
      if (dev < 0 || dev >= kMaxDev) e = f();            // beyond the table: set it every time
      else {
        std::call_once(flag[dev], [&] { err[dev] = f(); });
        e = err[dev];
      }
    }
    if (e != hipSuccess) {
      coattn_set_error("%s: per-device setup failed: %s", what, hipGetErrorString(e));
      return -3;
    }
    return 0;
  }
};

// ---- developer switches --------------------------------------------------------------------------------------------
// Environment variables that select kernels / widths for A/B timing (tools/ab_*.sh) exist only in builds made with
// -DCOATTN_DEV_SWITCHES; the shipped library ignores the environment (numerics must not depend on it).
#include <stdlib.h>
static inline int dev_env_int(const char* name, int dflt) {
#ifdef COATTN_DEV_SWITCHES
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
#else
  (void)name;
  return dflt;
#endif
}

// ---- per-kernel timing marks (coattn_profile_begin / _end, api.hip) ----------------------------------------------
// Between coattn_profile_begin and coattn_profile_end on the calling thread, prof_mark(s, name) records a HIP event on `s`:
// the time since the previous mark is attributed to `name` (the launches issued in between).  Off (one thread-local
// load, no HIP call) otherwise -- in particular under graph capture.
void prof_mark(hipStream_t s, const char* name);

// ---- wave64 reductions ------------------------------------------------------------------
// All-reduce butterflies without the LDS crossbar (__shfl_xor compiles to ds_bpermute_b32: a ~100-cycle round trip and an
// lgkmcnt wait per step): halves and 16-lane rows meet through gfx950's lane-swap instructions (after the swap the two
// registers hold [lo, lo] and [hi, hi]), the rest through DPP operands of the add itself -- row_ror:8 (= xor 8), row_half_mirror
// (lane i <-> i ^ 7), quad_perm xor 2, xor 1: masks 8, 7, 2, 1 span the row, so every lane ends with the whole sum, in one fixed order.
// (the swaps are inline asm: with the same value as both operands hipcc 7.2 adds the first result to itself)
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
#define CA_WAVE_REDUCE(OP)                                                                    \
  float a = v, b = v;                                                                         \
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));                 \
  v = OP(a, b);                                                                               \
  a = v; b = v;                                                                               \
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));                 \
  v = OP(a, b);                                                                               \
  v = OP(v, dpp_f32<0x128>(v));   /* row_ror:8 */                                             \
  v = OP(v, dpp_f32<0x141>(v));   /* row_half_mirror */                                       \
  v = OP(v, dpp_f32<0x4E>(v));    /* quad_perm [2,3,0,1] */                                   \
  v = OP(v, dpp_f32<0xB1>(v));    /* quad_perm [1,0,3,2] */                                   \
  return v;
__device__ __forceinline__ float ca_addf(float x, float y) { return x + y; }
// v(lane) + v(lane ^ 32): the two 32-lane halves of the wave, in every lane
__device__ __forceinline__ float half_sum(float v) {
  float a = v, b = v;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return a + b;
}
__device__ __forceinline__ float wave_sum(float v) { CA_WAVE_REDUCE(ca_addf) }
__device__ __forceinline__ float wave_max(float v) { CA_WAVE_REDUCE(fmaxf) }
#undef CA_WAVE_REDUCE

// ---- small reductions shared by reduce_jobs_kernel (small_kernels.hip) and gemm_tn_kernel (extra workgroups) ----
// up to 4 independent reductions of [nparts][n] partial buffers (row `by` = job, 64 columns per block `bx`)
// plus up to 2 whole-array sums (rows by >= njobs, block bx = 0 only): out = sum of sum_x[0 .. sum_n)
struct ReduceJobs {
  const float* src[4];
  float* dst[4];
  int njobs;
  const float* sum_x[2];
  float* sum_out[2];
  long sum_n;
  long ld;                                           // row stride of the partial buffers (0: n)
};
// (`active`: workgroups of more than 256 threads -- gemm_tn_wide_kernel -- pass threadIdx.x < 256: the other threads only
//  keep the barriers company)
__device__ __forceinline__ void reduce_jobs_block(const ReduceJobs& jobs, int nparts, long n, int accumulate, int bx,
                                                  int by, float (*red)[64], bool active = true) {
  // the four waves take interleaved parts and are combined in a fixed order
  if (by >= jobs.njobs) {                            // whole-array sum (fixed order: as sum_all_kernel)
    if (bx) return;
    const float* x = jobs.sum_x[by - jobs.njobs];
    float* out = jobs.sum_out[by - jobs.njobs];
    float acc = 0.f;
    if (active)
      for (long i = threadIdx.x; i < jobs.sum_n; i += 256) acc += x[i];
    acc = wave_sum(acc);
    if (active && (threadIdx.x & 63) == 0) red[0][threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
      const float t = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
      out[0] = accumulate ? out[0] + t : t;
    }
    return;
  }
  const float* part = jobs.src[by];
  float* out = jobs.dst[by];
  const int col = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const long j = (long)bx * 64 + col, ld = jobs.ld ? jobs.ld : n;
  float acc = 0.f;
  if (active && j < n) {
    int c = grp;
    for (; c + 28 < nparts; c += 32) {               // 8 loads in flight per thread
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = part[(long)(c + 4 * u) * ld + j];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; c < nparts; c += 4) acc += part[(long)c * ld + j];
  }
  if (active) red[grp][col] = acc;
  __syncthreads();
  if (grp == 0 && j < n) {
    const float t = (red[0][col] + red[1][col]) + (red[2][col] + red[3][col]);
    out[j] = accumulate ? out[j] + t : t;
  }
}


// tanh with ~2e-7 absolute error: 1 v_exp + 1 v_rcp.  (1-e)/(1+e), e = exp(-2|x|) in (0,1].
// (__fdividef lowers to the full IEEE division sequence on gfx950; v_rcp_f32 is 1 ulp.)
// 5 VALU ops: tanh(x) = 1 - 2 / (1 + exp(2x)); saturates cleanly (exp -> inf gives 1, -> 0 gives -1).
__device__ __forceinline__ float tanh_fast(float x) {
  const float e = __builtin_amdgcn_exp2f(2.8853900817779268f * x);            // exp(2x)
  return fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + e), 1.0f);
}

// 1 / (1 + exp(2x)): tanh(x) = 1 - 2 sig2(x)
__device__ __forceinline__ float sig2_fast(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.8853900817779268f * x));
}

// the same for an argument that already carries the factor 2 log2(e) (fused.h: kPScale)
__device__ __forceinline__ float sig2_scaled(float xs) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(xs)); }
__device__ __forceinline__ float tanh_scaled(float xs) { return fmaf(-2.0f, sig2_scaled(xs), 1.0f); }

// sum over the 16 lanes of a DPP row (lanes sharing lane>>4); result valid in every lane
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // row_mirror
  return v;
}

// Transposing sum of 16 registers over a DPP row: lane j of every 16-lane row ends with sum_i x[j][lane i].
// A butterfly whose halving steps also halve the register set (45 VALU operations; 16 separate row16_sum chains are 64
// plus the selects): steps A / B pair registers (g, g + 8) / (g, g + 4) and exchange across lane distance 8 / 4 -- the
// bank mask of a DPP move picks which half of the lanes takes the partner's value --, steps C / D finish inside a quad.
template <int CTRL, int BANK>
__device__ __forceinline__ float dpp_mov(float old, float src) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src),
                                                               CTRL, 0xF, BANK, false));
}
// lanes 0-7 of a row: lo[i] + lo[i + 8]; lanes 8-15: hi[i] + hi[i - 8]      (row_ror:8)
__device__ __forceinline__ float bfly_a(float lo, float hi) { return dpp_mov<0x128, 0x3>(hi, lo) + dpp_mov<0x128, 0xC>(lo, hi); }
// lanes with bit 2 clear: lo[i] + lo[i + 4] (row_shl:4); set: hi[i] + hi[i - 4] (row_shr:4)
__device__ __forceinline__ float bfly_b(float lo, float hi) { return dpp_mov<0x104, 0x5>(hi, lo) + dpp_mov<0x114, 0xA>(lo, hi); }
// lanes with bit 1 clear: lo[i] + lo[i ^ 2]; set: hi[i] + hi[i ^ 2]         (quad_perm [2,3,0,1])
__device__ __forceinline__ float bfly_c(float lo, float hi, int lane) {
  const float a = lo + dpp_mov<0x4E, 0xF>(0.f, lo), b = hi + dpp_mov<0x4E, 0xF>(0.f, hi);
  return (lane & 2) ? b : a;
}
// lanes with bit 0 clear: lo[i] + lo[i ^ 1]; set: hi[i] + hi[i ^ 1]         (quad_perm [1,0,3,2])
__device__ __forceinline__ float bfly_d(float lo, float hi, int lane) {
  const float a = lo + dpp_mov<0xB1, 0xF>(0.f, lo), b = hi + dpp_mov<0xB1, 0xF>(0.f, hi);
  return (lane & 1) ? b : a;
}
__device__ __forceinline__ float row16_sum16(const float (&x)[16], int lane) {
  float y[8], z[4], w[2];
#pragma unroll
  for (int g = 0; g < 8; ++g) y[g] = bfly_a(x[g], x[g + 8]);
#pragma unroll
  for (int g = 0; g < 4; ++g) z[g] = bfly_b(y[g], y[g + 4]);
#pragma unroll
  for (int g = 0; g < 2; ++g) w[g] = bfly_c(z[g], z[g + 2], lane);
  return bfly_d(w[0], w[1], lane);
}

// ---- generic GEMM (gemm.hip) ------------------------------------------------------------
int launch_gemm_f32(const coattn_gemm_desc& g, hipStream_t s);
// same contract, operands rounded to bf16 for v_mfma_f32_32x32x16_bf16 (falls back to fp32 when unsupported)
int launch_gemm_bf16in(const coattn_gemm_desc& g, hipStream_t s);

// ---- small general-shape kernels (small_kernels.hip) ------------------------------------
// y[z][i] = sum_k X[z*x_sz + i*x_si + k*x_sk] * u[z*u_sz + k]      (i < I, k < K)
int launch_gemv(const float* X, const float* u, float* y, int Z, int I, int K,
                int64_t x_sz, int64_t x_si, int64_t x_sk, int64_t u_sz, int64_t y_sz, hipStream_t s);
// a[z][r] = softmax_r( H[z][r][:] . w + c[0] ),  H rows contiguous (length d)
int launch_score_softmax(const float* H, const float* w, const float* c, float* a,
                         int Z, int R, int d, hipStream_t s);
// ds[z][r] = a[z][r] * (da[z][r] - sum_r a*da)
int launch_softmax_bwd(const float* a, const float* da, float* ds, int Z, int R, hipStream_t s);
// part[chunk][j] = sum_{r in chunk} s[r] * X[r][j]   (s may be NULL -> 1); X: [R][d] contiguous
// returns number of chunks through *nchunks; part must hold nchunks*d floats
int launch_colsum_partial(const float* s, const float* X, float* part, int R, int d, int rows_per_chunk,
                          int* nchunks, hipStream_t s_);
// out[j] (+)= sum_c part[c][j]   (n = elements per partial)
int launch_reduce_partials(const float* part, float* out, int nparts, int64_t n, int accumulate, hipStream_t s);
int launch_reduce_partials2(const float* part0, float* out0, int nparts0, const float* part1, float* out1, int nparts1,
                            int64_t n, int accumulate, hipStream_t s);
// up to 4 reductions dst[i][j] (+)= sum_c src[i][c][j] in one launch
int launch_reduce_jobs(const float* const* src, float* const* dst, int njobs, int nparts, int64_t n, int accumulate,
                       hipStream_t s, const float* const* sum_x = nullptr, float* const* sum_out = nullptr,
                       int64_t sum_n = 0);
// total (+)= sum of x[0..n)   (single block)
int launch_sum_all(const float* x, float* out, int64_t n, int accumulate, hipStream_t s);
int launch_sum_all2(const float* x0, float* out0, const float* x1, float* out1, int64_t n, int accumulate, hipStream_t s);
// H[z][r][j] = ds[z][r] * w[j] * (1 - H^2)      (in place or out of place)
int launch_dz(const float* ds, const float* w, const float* H, float* out, int64_t rows, int d, hipStream_t s);
// dA = dC * (1 - C^2)
int launch_dtanh(const float* dC, const float* C, float* out, int64_t n, hipStream_t s);
// y (+)= x
int launch_add_inplace(float* y, const float* x, int64_t n, int accumulate, hipStream_t s);
// y += x1 + x2 in one pass
int launch_add3_inplace(float* y, const float* x1, const float* x2, int64_t n, hipStream_t s);
// out[z][i][j] (+)= a[z][i] * g[z][j] with strides (o_sz, o_si, o_sj)
int launch_rank1(const float* a, const float* g, float* out, int Z, int I, int J,
                 int64_t o_sz, int64_t o_si, int64_t o_sj, int accumulate, hipStream_t s);

// ---- fused kernels (coattn_fused.hip) ---------------------------------------------------
int fused_supported(int B, int N, int T, int d, int L);
