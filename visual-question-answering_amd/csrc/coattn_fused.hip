// Fused co-attention kernels -- placeholder until the fused path lands.
#include "fused.h"

int fused_supported(int, int, int, int, int) { return 0; }
int fused_attention_forward(int, int, int, int, int, const float*, const float* const*, const coattn_params*, float*,
                            float*, float*, float*, hipStream_t) {
  coattn_set_error("fused forward not built");
  return -2;
}
int fused_backward(int, int, int, int, int, const float*, const float* const*, const coattn_params*, const float*,
                   const float*, const float*, float*, float* const*, const coattn_param_grads*, int, float*,
                   hipStream_t) {
  coattn_set_error("fused backward not built");
  return -2;
}
