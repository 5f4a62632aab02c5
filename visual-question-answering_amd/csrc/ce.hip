// Cross entropy of the train step on gfx950 (reference main.py:94 / :214: nn.CrossEntropyLoss() on the logits of
// MLPClassifier, mean over the batch) together with its gradient: one workgroup per row -- log-sum-exp, loss and
// d logits = (softmax - onehot) / B in one pass -- and a fixed-order sum of the row losses by the workgroup that finishes
// last (an integer ticket; no float atomics: deterministic).  (SURVEY.md section 8f-1.  The MLPClassifier itself stays on the stock PyTorch-ROCm modules: a
// composition of this library's GEMM for its four B-row products was built and measured in round 1 -- 30 launches,
// 0.53 ms against ~0.2 ms for the stock head -- and was removed again in round 2; see DESIGN.md.)
#include "common.h"

namespace {

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
// A label outside [0, K) makes the row's loss NaN and raises the status word (nn.CrossEntropyLoss raises there:
// coattn_ce_status reports it at the caller's next synchronisation point).
// The mean rides in the same launch: the workgroup that finishes LAST (a ticket from status[1], which it leaves at 0 for the
// next call) adds the B row losses in sum_all_kernel's fixed order -- whichever workgroup that is, the same bits.  status[0]
// and status[1] are 0 when the launch starts (the caller's memset, or the answer head's last layer: head.hip).
__global__ __launch_bounds__(256) void ce_rows_kernel(const float* __restrict__ logits, const long long* __restrict__ labels,
                                                      float* row_loss, float* __restrict__ dlogits, int K,
                                                      float inv_b, int* status, int ldd, float* __restrict__ loss, int B) {
  __shared__ float sh[4];
  __shared__ unsigned ticket;
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
  if (threadIdx.x == 0) {                             // (before the gradient row is stored: the fence has nothing of it to wait for)
    __hip_atomic_store(&row_loss[i], ok ? (logf(s) + m - z[lab]) * inv_b : NAN, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!ok) status[0] = i + 1;                       // (any offending row: the writers race benignly)
    __threadfence();                                  // the row loss is visible device-wide before the ticket is taken
    ticket = atomicAdd(reinterpret_cast<unsigned*>(status) + 1, 1u);
  }
  if (dlogits) {
    const float inv = inv_b / s;
    for (int k = threadIdx.x; k < ldd; k += 256)     // rows ldd >= K floats apart, the padding zeroed
      dlogits[(long)i * ldd + k] = k < K ? expf(z[k] - m) * inv - ((ok && k == lab) ? inv_b : 0.f) : 0.f;
  }
  __syncthreads();
  if (ticket != (unsigned)(B - 1)) return;
  __threadfence();
  float acc = 0.f;                                    // (sum_all_kernel's order: strided per thread, wave sums, (0 + 1) + (2 + 3))
  for (int r = threadIdx.x; r < B; r += 256) acc += __hip_atomic_load(&row_loss[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    loss[0] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
    reinterpret_cast<unsigned*>(status)[1] = 0u;
  }
}

inline size_t al64(size_t n) { return (n + 63) & ~(size_t)63; }

}  // namespace

// rows + mean for the answer head (head.hip): row_loss [B] scratch, dlogits [B][ldd] (ldd = 0: K) or NULL.
// zeroed: status[0] and status[1] have been cleared by an earlier launch on this stream (head.hip's logits layer)
int launch_ce_rows(const float* logits, const void* labels, float* row_loss, float* dlogits, float* loss, int B, int K,
                   int* status, hipStream_t s, int ldd, bool zeroed) {
  if (!zeroed && hipMemsetAsync(status, 0, 16, s) != hipSuccess) {       // (a memset node under graph capture)
    coattn_set_error("ce: clearing the status word failed");
    return -3;
  }
  hipLaunchKernelGGL(ce_rows_kernel, dim3(B), dim3(256), 0, s, logits, (const long long*)labels, row_loss, dlogits, K,
                     1.0f / (float)B, status, ldd > 0 ? ldd : K, loss, B);
  CA_CHECK_LAUNCH("ce_rows");
  return 0;
}

extern "C" int coattn_ce_workspace_bytes(int B, int K, int dtype, size_t* ws) {
  CA_CHECK_ARG(dtype == COATTN_F32, "unsupported dtype %d (only COATTN_F32)", dtype);
  CA_CHECK_ARG(B > 0 && K > 0, "bad B=%d / K=%d", B, K);
  if (ws) *ws = (al64((size_t)B) + 64) * sizeof(float);        // row losses + the status word (its own 256 bytes)
  return 0;
}

// Synchronises `stream` and reports whether the coattn_ce_forward / coattn_head_forward call that last used this
// workspace / saved buffer met a label outside [0, K): -2 (message: the offending row) or 0.
static int status_check(const int* status_dev, void* stream, const char* what) {
  int host = 0;
  if (hipMemcpyAsync(&host, status_dev, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess ||
      hipStreamSynchronize((hipStream_t)stream) != hipSuccess) {
    coattn_set_error("%s: reading the status word failed", what);
    return -3;
  }
  if (host != 0) {
    coattn_set_error("%s: label of row %d is outside [0, K) (nn.CrossEntropyLoss: 'Target out of bounds')", what, host - 1);
    return -2;
  }
  return 0;
}

extern "C" int coattn_ce_status(const void* ws, int B, void* stream) {
  CA_CHECK_ARG(ws && B > 0, "ce_status: bad argument");
  return status_check(reinterpret_cast<const int*>((const float*)ws + al64((size_t)B)), stream, "cross entropy");
}
int head_status_check(const int* status_dev, void* stream) { return status_check(status_dev, stream, "answer head"); }

extern "C" int coattn_ce_forward(const void* logits, const void* labels, void* loss, void* dlogits, void* ws, int B,
                                 int K, int dtype, void* stream) {
  CA_CHECK_ARG(dtype == COATTN_F32, "unsupported dtype %d (only COATTN_F32)", dtype);
  CA_CHECK_ARG(B > 0 && B <= (1 << 24) && K > 0, "bad B=%d / K=%d", B, K);
  CA_CHECK_ARG(logits && labels && loss && ws, "ce_forward: null argument");                   // dlogits may be NULL
  hipStream_t s = (hipStream_t)stream;
  return launch_ce_rows((const float*)logits, labels, (float*)ws, (float*)dlogits, (float*)loss, B, K,
                        reinterpret_cast<int*>((float*)ws + al64((size_t)B)), s, 0, false);
}
