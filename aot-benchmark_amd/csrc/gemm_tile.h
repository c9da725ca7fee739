// Device helpers shared by the tile GEMM kernels of gemm_lds.hip (fp32 matrix cores) and gemm_x6.hip (bf16x6 family): the LDS image
// of an fp32 operand tile, the counted-wait immediate, the work item, LDS-DMA pieces and the inline-asm buffer accesses of the tile ends.
#pragma once
#include "conv_params.h"
#include <cstdlib>
#include <type_traits>

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

namespace {

constexpr int BK = 32;
constexpr int GROUP_STRIDE = 8 * 128 + 16;   // bytes: 8 rows x 32 floats + one 16-byte pad

__device__ __forceinline__ int chunk_off(int row, int c) {
  const int g = row >> 3, r = row & 7;
  return g * GROUP_STRIDE + r * 128 + ((c ^ r) << 4);
}

// s_waitcnt simm16 on gfx9: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[5:4] << 14 (expcnt: no wait)
constexpr int waitcnt_imm(int vm, int lgkm) { return (vm & 15) | (7 << 4) | ((lgkm & 15) << 8) | ((vm >> 4) << 14); }

struct Item {
  int bm, bn, kt0;   // tile coordinates and first k-step of the slice
};

// one LDS-DMA piece: 64 lanes x 16 bytes, lane address = descriptor base + voff (+ the wave-uniform soff); an offset beyond the
// descriptor's range (0x80000000) delivers zeros.  (A __device__ function of its own: the target builtin inside a generic
// lambda silently keeps hipcc's HOST pass from emitting the kernel's launch stub.)
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned char* dst, int voff, int soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)dst, 16, voff, soff, 0, 0);
}
// raw buffer descriptor (stride 0) as four scalars, for the inline-asm buffer loads / stores of the epilogue
__device__ __forceinline__ i32x4 raw_desc(const void* base, long bytes) {
  const unsigned long long a = (unsigned long long)(uintptr_t)base;
  i32x4 d;
  d[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
  d[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32) & 0xffff);
  d[2] = __builtin_amdgcn_readfirstlane(base ? (int)(bytes < 0x7fffffffL ? bytes : 0x7fffffffL) : 0);
  d[3] = 0x00020000;
  return d;
}
// asynchronous (the compiler does not see them as memory operations: the callers place the vmcnt waits).  The s_nop covers
// the "VALU writes an SGPR -> vector-memory instruction reads it" hazard (5 wait states), which hipcc cannot insert for an
// instruction hidden in inline asm: under register pressure it restores the descriptor from spill lanes with v_readlane
// right in front of the asm (seen in the KxK kernel: the load then went out with a stale base address).
__device__ __forceinline__ float buf_load(const i32x4& desc, int voff) {
  float v;
  asm volatile("s_nop 4\n\tbuffer_load_dword %0, %1, %2, 0 offen" : "=v"(v) : "v"(voff), "s"(desc));
  return v;
}
__device__ __forceinline__ void buf_store(const i32x4& desc, int voff, float v) {
  asm volatile("s_nop 4\n\tbuffer_store_dword %0, %1, %2, 0 offen" : : "v"(v), "v"(voff), "s"(desc) : "memory");
}
// the same with a wave-uniform byte offset in an SGPR next to the lane's offset (address = base + soff + voff; an out-of-range
// voff still masks the access whatever soff is)
__device__ __forceinline__ float buf_load_s(const i32x4& desc, int voff, int soff) {
  float v;
  asm volatile("s_nop 4\n\tbuffer_load_dword %0, %1, %2, %3 offen" : "=v"(v) : "v"(voff), "s"(desc), "s"(soff));
  return v;
}
__device__ __forceinline__ void buf_store_s(const i32x4& desc, int voff, int soff, float v) {
  asm volatile("s_nop 4\n\tbuffer_store_dword %0, %1, %2, %3 offen" : : "v"(v), "v"(voff), "s"(desc), "s"(soff) : "memory");
}
// bits [31:16] of v as one 16-bit store (a truncated-bf16 plane element)
__device__ __forceinline__ void buf_store_hi16(const i32x4& desc, int voff, int soff, float v) {
  asm volatile("s_nop 4\n\tbuffer_store_short_d16_hi %0, %1, %2, %3 offen" : : "v"(v), "v"(voff), "s"(desc), "s"(soff) : "memory");
}
}  // namespace

// splitk_reduce_kernel (gemm_lds.hip): sums the [ksplit][M][Cout] slabs in slice order and applies bias / residual / activation
void launch_splitk_reduce(const ConvParams& p, int ksplit, const float* scratch, hipStream_t s);
// ... the same launch writing LayerNorm(result) * gamma + beta as a second output (splitk_reduce_kernel<true>; Cout == 256)
void launch_splitk_reduce_ln(const ConvParams& p, int ksplit, const float* scratch, const float* gamma, const float* beta, float* ln_out,
                             int ld_ln, float eps, hipStream_t s);
