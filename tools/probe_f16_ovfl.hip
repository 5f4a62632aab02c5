// Developer probe: MODE.FP16_OVFL (bit 23) and v_cvt_pk_f16_f32 on gfx950: saturate to +-65504 instead of inf?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 hv2 __attribute__((ext_vector_type(2)));
__global__ void k(float x, int ovfl, float* out) {
  if (ovfl) __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1);
  const hv2 h = __builtin_convertvector((f32x2{x, -x}), hv2);
  const float r = x - (float)h[0];
  const hv2 l = __builtin_convertvector((f32x2{r, r}), hv2);
  out[0] = (float)h[0]; out[1] = (float)h[1]; out[2] = r; out[3] = (float)l[0];
}
int main() {
  float* o; hipMalloc(&o, 16);
  for (int ov = 0; ov < 2; ++ov)
    for (float x : {1000.f, 65504.f, 70000.f, 100000.f, 131000.f, 200000.f, 1e30f}) {
      hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, x, ov, o);
      float h[4]; hipMemcpy(h, o, 16, hipMemcpyDeviceToHost);
      printf("ovfl=%d x=%g: hi=%g (-x: %g) resid=%g lo=%g  hi+lo=%g\n", ov, x, h[0], h[1], h[2], h[3], h[0] + h[3]);
    }
  return 0;
}
