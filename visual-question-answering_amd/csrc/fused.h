// Fused co-attention kernels (coattn_fused.hip): entry points used by api.hip.
#pragma once
#include "common.h"

int fused_supported(int B, int N, int T, int d, int L);
// everything after the projections (P_v, P_q already in `saved`)
int fused_attention_forward(int B, int N, int T, int d, int L, const float* V, const float* const* Q,
                            const coattn_params* p, float* v_out, float* q_out, float* saved, float* ws,
                            hipStream_t s);
int fused_backward_supported(int B, int N, int T, int d, int L);
int fused_backward(int B, int N, int T, int d, int L, const float* V, const float* const* Q, const coattn_params* p,
                   const float* saved, const float* gv, const float* gq, float* dV, float* const* dQ,
                   const coattn_param_grads* pg, int accumulate, float* ws, hipStream_t s);
