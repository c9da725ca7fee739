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

// gfx950: a packed (VOP3P) instruction whose destination pair is also a source with the halves CROSSED by op_sel -- e.g.
//   v_pk_add_f32 v[0:1], v[18:19], v[0:1] op_sel:[0,1] op_sel_hi:[1,0]      (lo = a.lo + b.hi, hi = a.hi + b.lo, b == dst)
// -- is not safe when other waves share the SIMD: lanes 48-63 of the LOW result can be formed from the instruction's own new high
// half.  hipcc's SLP vectoriser emits exactly that for pairs of `f0 * a0 + f1 * a1` sums (the (O, m, l) merge of the flash kernels
// was hit: wrong O for 16 of 32 queries in some instruction orders, round 3; root-caused in round 4 by an ISA-level bisect,
// profiles/r04_hazard.txt).  scalar_fp32() pins a value in a register of its own: the vectoriser cannot pair what flows through
// it.  tests/test_host.py::test_no_inplace_crossed_packed_ops audits the ISA of every kernel of the library for the pattern
// (tools/dev/isa_pk_inplace_audit.py).
__device__ __forceinline__ float scalar_fp32(float x) {
  asm volatile("" : "+v"(x));
  return x;
}
// ((f0 a0 + f1 a1) + f2 a2) + f3 a3 in that order, in scalar fp32 (the merge of four partial accumulators)
__device__ __forceinline__ float merge4_scalar(const float (&f)[4], float a0, float a1, float a2, float a3) {
  const float p0 = scalar_fp32(f[0] * a0), p1 = scalar_fp32(f[1] * a1), p2 = scalar_fp32(f[2] * a2), p3 = scalar_fp32(f[3] * a3);
  return scalar_fp32(scalar_fp32(p0 + p1) + p2) + p3;
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
