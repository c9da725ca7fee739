// Shared helpers for the gfx950 kernels of libaot_hip.so.
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/aot_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

#define AOT_LAUNCH_CHECK()                      \
  do {                                          \
    hipError_t e__ = hipGetLastError();         \
    if (e__ != hipSuccess) return (int)e__;     \
    return AOT_OK;                              \
  } while (0)

// row of the 32x32 MFMA C/D fragment held in accumulator register r by a lane of half `hi`
// (cdna_hip_programming.md section 3: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5))
__device__ __forceinline__ int mfma32_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

__device__ __forceinline__ float apply_act(float v, int act) {
  if (act == AOT_ACT_RELU) return fmaxf(v, 0.f);
  if (act == AOT_ACT_RELU6) return fminf(fmaxf(v, 0.f), 6.f);
  if (act == AOT_ACT_GELU) return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));   // exact-erf GELU (nn.GELU)
  if (act == AOT_ACT_SILU) return v * (1.f / (1.f + expf(-v)));   // x * sigmoid(x), attention.py:585-586
  return v;
}

// One wave-uniform branch on the activation around a whole store loop instead of apply_act's if-chain per element:
//   with_act(p.act, [&](auto ACT) { ... apply_act(v, decltype(ACT)::value) ... });
template <class F>
__device__ __forceinline__ void with_act(int act, F&& f) {
  switch (act) {
    case AOT_ACT_RELU: f(std::integral_constant<int, AOT_ACT_RELU>{}); break;
    case AOT_ACT_RELU6: f(std::integral_constant<int, AOT_ACT_RELU6>{}); break;
    case AOT_ACT_GELU: f(std::integral_constant<int, AOT_ACT_GELU>{}); break;
    case AOT_ACT_SILU: f(std::integral_constant<int, AOT_ACT_SILU>{}); break;
    default: f(std::integral_constant<int, AOT_ACT_NONE>{}); break;
  }
}
// (the register-staged and the wave-independent GEMM kernels of gemm_conv.hip end their tiles through with_act() too)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
