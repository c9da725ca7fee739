// Does a write to the A / B registers of an ISSUED bf16 MFMA, placed right behind it in program order, change its result?
// Each wave runs chains of 8 dependent v_mfma_f32_32x32x16_bf16 whose A operand is then overwritten (a) by a VALU v_mov, (b) by
// a global load that returns other data, with nothing in between; the accumulated result is compared with the value the
// original operands give.  Run at 1 and 4 waves per SIMD (the matrix pipe shared between waves).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ void __launch_bounds__(256) probe(const u32x4* __restrict__ mem, float* __restrict__ out, int iters) {
  const u32x4 one = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};      // bf16 1.0 x 8
  u32x4 a = one;
  const bf16x8 b = __builtin_bit_cast(bf16x8, one);
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const u32x4* src = mem + (blockIdx.x * 256 + threadIdx.x) % 4096;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
      if (MODE == 1) {          // VALU overwrite of A (bf16 2.0), then restore for the next MFMA after some distance
        asm volatile("v_mov_b32 %0, 0x40004000\n v_mov_b32 %1, 0x40004000\n v_mov_b32 %2, 0x40004000\n v_mov_b32 %3, 0x40004000"
                     : "=v"(a[0]), "=v"(a[1]), "=v"(a[2]), "=v"(a[3]));
        asm volatile("s_nop 7\n v_mov_b32 %0, 0x3f803f80\n v_mov_b32 %1, 0x3f803f80\n v_mov_b32 %2, 0x3f803f80\n v_mov_b32 %3, 0x3f803f80\n s_nop 7"
                     : "=v"(a[0]), "=v"(a[1]), "=v"(a[2]), "=v"(a[3]));
      }
      if (MODE == 2) {          // a load (memory holds bf16 2.0) lands in A; wait for it, then restore
        asm volatile("global_load_dwordx4 %0, %1, off\n s_waitcnt vmcnt(0)" : "=v"(a) : "v"(src) : "memory");
        asm volatile("v_mov_b32 %0, 0x3f803f80\n v_mov_b32 %1, 0x3f803f80\n v_mov_b32 %2, 0x3f803f80\n v_mov_b32 %3, 0x3f803f80\n s_nop 7"
                     : "=v"(a[0]), "=v"(a[1]), "=v"(a[2]), "=v"(a[3]));
      }
    }
  }
  float s = 0.f;
  asm volatile("s_nop 7\n s_nop 7\n s_nop 7" ::: "memory");
  for (int r = 0; r < 16; ++r) s += acc[r];
  out[blockIdx.x * 256 + threadIdx.x] = s;          // expected 16 regs x (iters * 8 MFMAs x 16 products of 1 x 1)
}
template <int MODE> void run(const u32x4* mem, float* d, float* h, int occ, int iters, const char* what) {
  const int blocks = 256 * occ, n = blocks * 256;
  hipLaunchKernelGGL((probe<MODE>), dim3(blocks), dim3(256), 0, 0, mem, d, iters);
  hipMemcpy(h, d, n * 4, hipMemcpyDeviceToHost);
  const float want = 16.f * iters * 8 * 16;
  long bad = 0; float worst = 0;
  for (int i = 0; i < n; ++i) if (h[i] != want) { ++bad; if (fabsf(h[i] - want) > worst) worst = fabsf(h[i] - want); }
  printf("%d waves/SIMD  %-44s wrong lanes %ld of %d (largest deviation %.0f of %.0f)\n", occ, what, bad, n, worst, want);
}
int main() {
  u32x4* mem; float *d, *h = (float*)malloc(1024 * 256 * 4);
  hipMalloc(&mem, 4096 * 16); hipMalloc(&d, 1024 * 256 * 4);
  unsigned* hm = (unsigned*)malloc(4096 * 16);
  for (int i = 0; i < 4096 * 4; ++i) hm[i] = 0x40004000u;
  hipMemcpy(mem, hm, 4096 * 16, hipMemcpyHostToDevice);
  for (int occ : {1, 4}) {
    run<0>(mem, d, h, occ, 200, "chain only");
    run<1>(mem, d, h, occ, 200, "VALU overwrite of A right behind each MFMA");
    run<2>(mem, d, h, occ, 200, "global load into A right behind each MFMA");
  }
  return 0;
}
