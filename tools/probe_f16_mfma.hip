// Developer probe: does v_mfma_f32_32x32x16_f16 keep subnormal fp16 inputs on gfx950?  And what do the f32 -> f16
// conversions do with values below fp16's normal range?  hipcc --offload-arch=gfx950 -o probe_f16_mfma probe_f16_mfma.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(float a_val, float b_val, float* out, float* cv) {
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)0.f; b[i] = (_Float16)0.f; }
  const int lane = threadIdx.x;
  // A[m][k]: lane (m = lane & 31, k half = lane >> 5); put a_val at k = 0 of row m; B[k][n]: b_val at k = 0, column n
  if (lane < 32) { a[0] = (_Float16)a_val; b[0] = (_Float16)b_val; }
  f32x16 c;
  for (int i = 0; i < 16; ++i) c[i] = 0.f;
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  if (lane == 0) { out[0] = c[0]; cv[0] = (float)(_Float16)a_val; }
}
int main() {
  float *o, *cv;
  hipMalloc(&o, 4); hipMalloc(&cv, 4);
  const float tests[][2] = {{1.0f, 1.0f}, {ldexpf(1.f, -14), 1024.f}, {ldexpf(1.f, -16), 1024.f}, {ldexpf(1.f, -20), 1024.f},
                            {ldexpf(1.5f, -22), 1024.f}, {ldexpf(1.f, -24), 1024.f}, {1024.f, ldexpf(1.f, -20)}};
  for (auto& t : tests) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, t[0], t[1], o, cv);
    float h, c;
    hipMemcpy(&h, o, 4, hipMemcpyDeviceToHost); hipMemcpy(&c, cv, 4, hipMemcpyDeviceToHost);
    printf("a=%g b=%g: mfma %g (exact %g), cvt(a)=%g\n", t[0], t[1], h, t[0] * t[1], c);
  }
  return 0;
}
