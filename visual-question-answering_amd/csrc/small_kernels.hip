// General-shape (any N, T, d) kernels around the MFMA GEMM: attention scores + row softmax
// (model.py:387-388), attended-feature reductions (model.py:391-392) and the element-wise
// pieces of the hand-derived backward.  All HBM-bound; lanes run along
// the contiguous axis, reductions are wave64 shuffles, results are deterministic (no atomics).
#include "common.h"

namespace {

// ---- batched GEMV ------------------------------------------------------------------------
// path A (x_sk == 1): one wave per output i, lanes over k, shuffle reduce.
__global__ __launch_bounds__(256) void gemv_kcontig(const float* X, const float* u, float* y, int I, int K,
                                                    long x_sz, long x_si, long u_sz, long y_sz) {
  const int z = blockIdx.y, lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= I) return;
  const float* xr = X + z * x_sz + i * x_si;
  const float* uz = u + z * u_sz;
  float acc = 0.f;
  for (int k = lane; k < K; k += 64) acc = fmaf(xr[k], uz[k], acc);
  acc = wave_sum(acc);
  if (lane == 0) y[z * y_sz + i] = acc;
}
// path B (x_si == 1): one thread per output i, loop over k (coalesced across threads).
__global__ __launch_bounds__(256) void gemv_icontig(const float* X, const float* u, float* y, int I, int K,
                                                    long x_sz, long x_sk, long u_sz, long y_sz) {
  const int z = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= I) return;
  const float* xz = X + z * x_sz + i;
  const float* uz = u + z * u_sz;
  float acc = 0.f;
  for (int k = 0; k < K; ++k) acc = fmaf(xz[(long)k * x_sk], uz[k], acc);
  y[z * y_sz + i] = acc;
}
// fallback: arbitrary strides
__global__ __launch_bounds__(256) void gemv_generic(const float* X, const float* u, float* y, int I, int K,
                                                    long x_sz, long x_si, long x_sk, long u_sz, long y_sz) {
  const int z = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= I) return;
  float acc = 0.f;
  for (int k = 0; k < K; ++k) acc = fmaf(X[z * x_sz + i * x_si + k * x_sk], u[z * u_sz + k], acc);
  y[z * y_sz + i] = acc;
}

// ---- scores + softmax over the R rows of one batch item -----------------------------------
// one workgroup per z; wave w handles rows w, w+4, ...; scores kept in LDS (R <= 4096).
__global__ __launch_bounds__(256) void score_softmax_kernel(const float* H, const float* w, const float* c,
                                                            float* a, int R, int d) {
  extern __shared__ __attribute__((aligned(16))) float sc[];
  __shared__ float red[8];
  const int z = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* Hz = H + (long)z * R * d;
  const float c0 = c[0];
  for (int r = wave; r < R; r += 4) {
    const float* hr = Hz + (long)r * d;
    float acc = 0.f;
    for (int k = lane; k < d; k += 64) acc = fmaf(hr[k], w[k], acc);
    acc = wave_sum(acc);
    if (lane == 0) sc[r] = acc + c0;
  }
  __syncthreads();
  float m = -INFINITY;
  for (int r = threadIdx.x; r < R; r += 256) m = fmaxf(m, sc[r]);
  m = wave_max(m);
  if (lane == 0) red[wave] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float s = 0.f;
  for (int r = threadIdx.x; r < R; r += 256) {
    const float e = expf(sc[r] - m);
    sc[r] = e;
    s += e;
  }
  s = wave_sum(s);
  if (lane == 0) red[4 + wave] = s;
  __syncthreads();
  s = (red[4] + red[5]) + (red[6] + red[7]);
  const float inv = 1.0f / s;
  for (int r = threadIdx.x; r < R; r += 256) a[(long)z * R + r] = sc[r] * inv;
}

// ds = a * (da - <a,da>), one wave per z
__global__ __launch_bounds__(64) void softmax_bwd_kernel(const float* a, const float* da, float* ds, int R) {
  const int z = blockIdx.x, lane = threadIdx.x;
  const float* az = a + (long)z * R;
  const float* dz = da + (long)z * R;
  float dot = 0.f;
  for (int r = lane; r < R; r += 64) dot = fmaf(az[r], dz[r], dot);
  dot = wave_sum(dot);
  for (int r = lane; r < R; r += 64) ds[(long)z * R + r] = az[r] * (dz[r] - dot);
}

// part[chunk][j] = sum_{r in chunk} s[r] * X[r][j]
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* s, const float* X, float* part,
                                                             int R, int d, int rpc) {
  const int chunk = blockIdx.y;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= d) return;
  const int r0 = chunk * rpc, r1 = min(R, r0 + rpc);
  float acc = 0.f;
  for (int r = r0; r < r1; ++r) {
    const float sv = s ? s[r] : 1.0f;
    acc = fmaf(sv, X[(long)r * d + j], acc);
  }
  part[(long)chunk * d + j] = acc;
}

__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* part, float* out, int nparts, long n,
                                                              int accumulate) {
  const long j = (long)blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  float acc = 0.f;
  int c = 0;
  for (; c + 8 <= nparts; c += 8) {                  // 8 independent loads in flight, fixed add order
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = part[(long)(c + u) * n + j];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u];
  }
  for (; c < nparts; ++c) acc += part[(long)c * n + j];
  out[j] = accumulate ? out[j] + acc : acc;
}

// out[j] (+)= sum_c part[c][j], four float columns per thread (n % 4 == 0, 16-byte aligned)
// (blockIdx.y = 1: the second job, same n)
__global__ __launch_bounds__(256) void reduce_partials4_kernel(const float* part, float* out, int nparts, long n,
                                                               int accumulate, const float* part1, float* out1,
                                                               int nparts1) {
  if (blockIdx.y) { part = part1; out = out1; nparts = nparts1; }
  const long j = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (j >= n) return;
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
  int c = 0;
  for (; c + 8 <= nparts; c += 8) {
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(part + (long)(c + u) * n + j);
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u];
  }
  for (; c < nparts; ++c) acc += *reinterpret_cast<const f32x4*>(part + (long)c * n + j);
  f32x4* o = reinterpret_cast<f32x4*>(out + j);
  *o = accumulate ? *o + acc : acc;
}

__global__ __launch_bounds__(256) void reduce_jobs_kernel(const ReduceJobs jobs, int nparts, long n, int accumulate) {
  __shared__ float red[4][64];
  reduce_jobs_block(jobs, nparts, n, accumulate, (int)blockIdx.x, (int)blockIdx.y, red);
}

__global__ __launch_bounds__(256) void add4_inplace_kernel(float* y, const float* x, long n4, int accumulate) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n4) return;
  f32x4* yp = reinterpret_cast<f32x4*>(y) + idx;
  const f32x4 xv = reinterpret_cast<const f32x4*>(x)[idx];
  *yp = accumulate ? *yp + xv : xv;
}

// y += x1 + x2 (fixed order (y + x1) + x2), 16 bytes per thread
__global__ __launch_bounds__(256) void add3_inplace_kernel(float* y, const float* x1, const float* x2, long n4) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n4) return;
  f32x4* yp = reinterpret_cast<f32x4*>(y) + idx;
  *yp = (*yp + reinterpret_cast<const f32x4*>(x1)[idx]) + reinterpret_cast<const f32x4*>(x2)[idx];
}

// one workgroup per job (blockIdx.x): out_j = sum of x_j[0 .. n)   (fixed order)
__global__ __launch_bounds__(256) void sum_all_kernel(const float* x0, float* out0, const float* x1, float* out1, long n,
                                                      int accumulate) {
  const float* x = blockIdx.x ? x1 : x0;
  float* out = blockIdx.x ? out1 : out0;
  __shared__ float red[4];
  float acc = 0.f;
  for (long i = threadIdx.x; i < n; i += 256) acc += x[i];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float t = (red[0] + red[1]) + (red[2] + red[3]);
    out[0] = accumulate ? out[0] + t : t;
  }
}

__global__ __launch_bounds__(256) void dz_kernel(const float* ds, const float* w, const float* H, float* out,
                                                 long rows, int d) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= rows * d) return;
  const long r = idx / d;
  const int j = (int)(idx - r * d);
  const float h = H[idx];
  out[idx] = ds[r] * w[j] * (1.0f - h * h);
}

__global__ __launch_bounds__(256) void dtanh_kernel(const float* dC, const float* C, float* out, long n) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n) return;
  const float c = C[idx];
  out[idx] = dC[idx] * (1.0f - c * c);
}

__global__ __launch_bounds__(256) void add_inplace_kernel(float* y, const float* x, long n, int accumulate) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n) return;
  y[idx] = accumulate ? y[idx] + x[idx] : x[idx];
}

// out[z][i][j] (+)= a[z][i] * g[z][j]; threads run along whichever of i/j has stride 1
__global__ __launch_bounds__(256) void rank1_kernel(const float* a, const float* g, float* out, int I, int J,
                                                    long o_sz, long o_si, long o_sj, int accumulate, int ifast) {
  const int z = blockIdx.y;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)I * J) return;
  int i, j;
  if (ifast) { i = (int)(idx % I); j = (int)(idx / I); } else { j = (int)(idx % J); i = (int)(idx / J); }
  const float v = a[(long)z * I + i] * g[(long)z * J + j];
  float* o = out + z * o_sz + i * o_si + j * o_sj;
  *o = accumulate ? *o + v : v;
}

}  // namespace

int launch_gemv(const float* X, const float* u, float* y, int Z, int I, int K, int64_t x_sz, int64_t x_si,
                int64_t x_sk, int64_t u_sz, int64_t y_sz, hipStream_t s) {
  CA_CHECK_ARG(Z > 0 && I > 0 && K > 0 && Z <= 65535, "gemv: bad shape");
  if (x_sk == 1) {
    hipLaunchKernelGGL(gemv_kcontig, dim3((I + 3) / 4, Z), dim3(256), 0, s, X, u, y, I, K, (long)x_sz, (long)x_si,
                       (long)u_sz, (long)y_sz);
  } else if (x_si == 1) {
    hipLaunchKernelGGL(gemv_icontig, dim3((I + 255) / 256, Z), dim3(256), 0, s, X, u, y, I, K, (long)x_sz,
                       (long)x_sk, (long)u_sz, (long)y_sz);
  } else {
    hipLaunchKernelGGL(gemv_generic, dim3((I + 255) / 256, Z), dim3(256), 0, s, X, u, y, I, K, (long)x_sz,
                       (long)x_si, (long)x_sk, (long)u_sz, (long)y_sz);
  }
  CA_CHECK_LAUNCH("gemv");
  return 0;
}

int launch_score_softmax(const float* H, const float* w, const float* c, float* a, int Z, int R, int d,
                         hipStream_t s) {
  CA_CHECK_ARG(Z > 0 && R > 0 && R <= 4096 && d > 0, "score_softmax: bad shape Z=%d R=%d d=%d", Z, R, d);
  hipLaunchKernelGGL(score_softmax_kernel, dim3(Z), dim3(256), (size_t)R * sizeof(float), s, H, w, c, a, R, d);
  CA_CHECK_LAUNCH("score_softmax");
  return 0;
}

int launch_softmax_bwd(const float* a, const float* da, float* ds, int Z, int R, hipStream_t s) {
  hipLaunchKernelGGL(softmax_bwd_kernel, dim3(Z), dim3(64), 0, s, a, da, ds, R);
  CA_CHECK_LAUNCH("softmax_bwd");
  return 0;
}

int launch_colsum_partial(const float* sv, const float* X, float* part, int R, int d, int rpc, int* nchunks,
                          hipStream_t s) {
  const int nc = (R + rpc - 1) / rpc;
  CA_CHECK_ARG(nc <= 65535, "colsum: too many chunks");
  *nchunks = nc;
  hipLaunchKernelGGL(colsum_partial_kernel, dim3((d + 255) / 256, nc), dim3(256), 0, s, sv, X, part, R, d, rpc);
  CA_CHECK_LAUNCH("colsum_partial");
  return 0;
}

// sum_x / sum_out (may be NULL): two whole-array sums of sum_n floats each, done by the same launch
int launch_reduce_jobs(const float* const* src, float* const* dst, int njobs, int nparts, int64_t n, int accumulate,
                       hipStream_t s, const float* const* sum_x, float* const* sum_out, int64_t sum_n) {
  CA_CHECK_ARG(njobs >= 1 && njobs <= 4, "reduce_jobs: 1..4 jobs");
  ReduceJobs jobs = {};
  for (int i = 0; i < 4; ++i) { jobs.src[i] = i < njobs ? src[i] : nullptr; jobs.dst[i] = i < njobs ? dst[i] : nullptr; }
  jobs.njobs = njobs;
  const int nsum = sum_x ? 2 : 0;
  for (int i = 0; i < nsum; ++i) { jobs.sum_x[i] = sum_x[i]; jobs.sum_out[i] = sum_out[i]; }
  jobs.sum_n = (long)sum_n;
  hipLaunchKernelGGL(reduce_jobs_kernel, dim3((unsigned)((n + 63) / 64), njobs + nsum), dim3(256), 0, s, jobs, nparts,
                     (long)n, accumulate);
  CA_CHECK_LAUNCH("reduce_jobs");
  return 0;
}

int launch_reduce_partials(const float* part, float* out, int nparts, int64_t n, int accumulate, hipStream_t s) {
  if ((n & 3) == 0 && ((((uintptr_t)part) | ((uintptr_t)out)) & 15) == 0) {
    hipLaunchKernelGGL(reduce_partials4_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, s, part, out,
                       nparts, (long)n, accumulate, (const float*)nullptr, (float*)nullptr, 0);
    CA_CHECK_LAUNCH("reduce_partials4");
    return 0;
  }
  hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, part, out, nparts,
                     (long)n, accumulate);
  CA_CHECK_LAUNCH("reduce_partials");
  return 0;
}

// two reductions of equal width n in one launch (n % 4 == 0, 16-byte aligned buffers)
int launch_reduce_partials2(const float* part0, float* out0, int nparts0, const float* part1, float* out1, int nparts1,
                            int64_t n, int accumulate, hipStream_t s) {
  const bool al = (n & 3) == 0 && ((((uintptr_t)part0) | ((uintptr_t)out0) | ((uintptr_t)part1) | ((uintptr_t)out1)) & 15) == 0;
  if (!al) {
    CA_TRY(launch_reduce_partials(part0, out0, nparts0, n, accumulate, s));
    return launch_reduce_partials(part1, out1, nparts1, n, accumulate, s);
  }
  hipLaunchKernelGGL(reduce_partials4_kernel, dim3((unsigned)((n / 4 + 255) / 256), 2), dim3(256), 0, s, part0, out0,
                     nparts0, (long)n, accumulate, part1, out1, nparts1);
  CA_CHECK_LAUNCH("reduce_partials4");
  return 0;
}

int launch_sum_all(const float* x, float* out, int64_t n, int accumulate, hipStream_t s) {
  hipLaunchKernelGGL(sum_all_kernel, dim3(1), dim3(256), 0, s, x, out, (const float*)nullptr, (float*)nullptr, (long)n,
                     accumulate);
  CA_CHECK_LAUNCH("sum_all");
  return 0;
}

// two sums of the same length in one launch
int launch_sum_all2(const float* x0, float* out0, const float* x1, float* out1, int64_t n, int accumulate, hipStream_t s) {
  hipLaunchKernelGGL(sum_all_kernel, dim3(2), dim3(256), 0, s, x0, out0, x1, out1, (long)n, accumulate);
  CA_CHECK_LAUNCH("sum_all2");
  return 0;
}

int launch_dz(const float* ds, const float* w, const float* H, float* out, int64_t rows, int d, hipStream_t s) {
  const long n = rows * d;
  hipLaunchKernelGGL(dz_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, ds, w, H, out, (long)rows, d);
  CA_CHECK_LAUNCH("dz");
  return 0;
}

int launch_dtanh(const float* dC, const float* C, float* out, int64_t n, hipStream_t s) {
  hipLaunchKernelGGL(dtanh_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dC, C, out, (long)n);
  CA_CHECK_LAUNCH("dtanh");
  return 0;
}

int launch_add_inplace(float* y, const float* x, int64_t n, int accumulate, hipStream_t s) {
  if ((n & 3) == 0 && ((((uintptr_t)y) | ((uintptr_t)x)) & 15) == 0) {
    hipLaunchKernelGGL(add4_inplace_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, s, y, x, (long)(n / 4),
                       accumulate);
    CA_CHECK_LAUNCH("add4_inplace");
    return 0;
  }
  hipLaunchKernelGGL(add_inplace_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, y, x, (long)n,
                     accumulate);
  CA_CHECK_LAUNCH("add_inplace");
  return 0;
}

int launch_add3_inplace(float* y, const float* x1, const float* x2, int64_t n, hipStream_t s) {
  if ((n & 3) == 0 && ((((uintptr_t)y) | ((uintptr_t)x1) | ((uintptr_t)x2)) & 15) == 0) {
    hipLaunchKernelGGL(add3_inplace_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, s, y, x1, x2, (long)(n / 4));
    CA_CHECK_LAUNCH("add3_inplace");
    return 0;
  }
  CA_TRY(launch_add_inplace(y, x1, n, 1, s));
  return launch_add_inplace(y, x2, n, 1, s);
}

int launch_rank1(const float* a, const float* g, float* out, int Z, int I, int J, int64_t o_sz, int64_t o_si,
                 int64_t o_sj, int accumulate, hipStream_t s) {
  const long n = (long)I * J;
  hipLaunchKernelGGL(rank1_kernel, dim3((unsigned)((n + 255) / 256), Z), dim3(256), 0, s, a, g, out, I, J,
                     (long)o_sz, (long)o_si, (long)o_sj, accumulate, o_si == 1 ? 1 : 0);
  CA_CHECK_LAUNCH("rank1");
  return 0;
}

// ---- image features -> the layout the kernels run on ------------------------------------------------------------------
// out[b][n][c] (fp32, contiguous [B, N, d]) = x[b sB + n sN + c sD]  (fp32 or bf16, any strides).  The case it exists
// for: the permuted view of an NCHW encoder output (model.py:215-217: strides (d N, 1, N)) whose rows are not 16-byte
// multiples (N = 49: the 7 x 7 grids of BASELINE configs 2 - 5 at 224 x 224) or which an autocast encoder left in bf16
// (config 4, main.py:73, :185) -- one pass, read along n, written along c, instead of the up-cast and the strided copy
// the host framework would make of it (two passes, the second at a quarter of the memory rate).
namespace {
template <typename TIn>
__device__ __forceinline__ float feat_load(const TIn* p);
template <>
__device__ __forceinline__ float feat_load<float>(const float* p) { return *p; }
template <>
__device__ __forceinline__ float feat_load<unsigned short>(const unsigned short* p) {
  return __builtin_bit_cast(float, (unsigned)(*p) << 16);
}

// TR: the input is contiguous along n (sN == 1): 64 (c) x 64 (n) tiles through LDS.  Else: lanes along c, no LDS.
// grid (ceil(N / 64), ceil(d / 64), B), 256 threads.
template <typename TIn, bool TR>
__global__ __launch_bounds__(256) void features_native_kernel(const TIn* __restrict__ x, long sB, long sN, long sD,
                                                              float* __restrict__ out, int N, int d) {
  __shared__ float tile[64][65];
  const int n0 = blockIdx.x * 64, c0 = blockIdx.y * 64, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const TIn* xb = x + (long)blockIdx.z * sB;
  float* ob = out + (long)blockIdx.z * N * d;
  if (TR) {
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {               // all the requests first
      const int c = c0 + w + 4 * i, n = n0 + lane;
      v[i] = (c < d && n < N) ? feat_load<TIn>(xb + (long)n * sN + (long)c * sD) : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) tile[w + 4 * i][lane] = v[i];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int n = n0 + w + 4 * i, c = c0 + lane;
      if (n < N && c < d) ob[(long)n * d + c] = tile[lane][w + 4 * i];
    }
  } else {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int n = n0 + w + 4 * i, c = c0 + lane;
      if (n < N && c < d) ob[(long)n * d + c] = feat_load<TIn>(xb + (long)n * sN + (long)c * sD);
    }
  }
}
}  // namespace

extern "C" int coattn_features_native(const void* x, int x_dtype, int64_t sB, int64_t sN, int64_t sD, void* out, int B,
                                      int N, int d, void* stream) {
  CA_CHECK_ARG(x_dtype == COATTN_F32 || x_dtype == COATTN_BF16, "features_native: dtype %d (COATTN_F32 or COATTN_BF16)", x_dtype);
  CA_CHECK_ARG(x && out && B > 0 && N > 0 && d > 0 && B <= 65535, "features_native: bad argument (B=%d N=%d d=%d)", B, N, d);
  CA_CHECK_ARG(sB >= 0 && sN >= 0 && sD >= 0, "features_native: negative strides");
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((unsigned)((N + 63) / 64), (unsigned)((d + 63) / 64), (unsigned)B);
  const bool tr = sN == 1 && sD != 1;
  float* o = (float*)out;
  if (x_dtype == COATTN_F32) {
    if (tr) hipLaunchKernelGGL((features_native_kernel<float, true>), grid, dim3(256), 0, s, (const float*)x, (long)sB, (long)sN, (long)sD, o, N, d);
    else hipLaunchKernelGGL((features_native_kernel<float, false>), grid, dim3(256), 0, s, (const float*)x, (long)sB, (long)sN, (long)sD, o, N, d);
  } else {
    if (tr) hipLaunchKernelGGL((features_native_kernel<unsigned short, true>), grid, dim3(256), 0, s, (const unsigned short*)x, (long)sB, (long)sN, (long)sD, o, N, d);
    else hipLaunchKernelGGL((features_native_kernel<unsigned short, false>), grid, dim3(256), 0, s, (const unsigned short*)x, (long)sB, (long)sN, (long)sD, o, N, d);
  }
  CA_CHECK_LAUNCH("features_native");
  return 0;
}
