// Shared helpers for the gfx950 kernels of libaot_hip.so.
#pragma once
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

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
