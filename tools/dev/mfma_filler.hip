// What does an instruction cost beside fp32 MFMAs on gfx950?  (MI355X: v_mfma_f32_32x32x2_f32 runs at the fp32 VECTOR rate.)
// Each wave issues chains of v_mfma_f32_32x32x2_f32; after every MFMA it issues N fillers of one kind.  Reported: ns per MFMA per
// SIMD (64 cycles = 26.7 ns at 2.4 GHz) and the extra cycles per filler.
//   finding (profiles/r02_mfma_filler.txt): VALU work does NOT hide under an fp32 MFMA -- every v_fma_f32 beside the chain adds
//   ~5 cycles, whether the neighbours are dependent MFMAs or not, at 1, 2 or 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
enum { K_NONE, K_FMA, K_EXP, K_PKFMA, K_MAX, K_SALU, K_MUL, K_PKMUL, K_DSREAD };
template <int KIND, int N>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
  __shared__ float lds[1024];
  lds[threadIdx.x] = threadIdx.x;
  __syncthreads();
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float a = threadIdx.x * 1e-3f, b = 1.0001f, x[8] = {1.f, 2.f, 3.f, 4.f, 5.f, 6.f, 7.f, 8.f};
  f32x2 px[4] = {{1.f, 2.f}, {3.f, 4.f}, {5.f, 6.f}, {7.f, 8.f}}, pb = {1.0001f, 0.999f};
  int sacc = iters;
  unsigned ladr = threadIdx.x * 4;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
#pragma unroll
      for (int f = 0; f < N; ++f) {
        if (KIND == K_FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[f]) : "v"(b), "v"(a));
        if (KIND == K_EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(x[f]));
        if (KIND == K_PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(px[f & 3]) : "v"(pb));
        if (KIND == K_MAX) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x[f]) : "v"(a));
        if (KIND == K_SALU) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sacc));
        if (KIND == K_MUL) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[f]) : "v"(b));
        if (KIND == K_PKMUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(px[f & 3]) : "v"(pb));
        if (KIND == K_DSREAD) asm volatile("ds_read_b32 %0, %1" : "=v"(x[f]) : "v"(ladr));
      }
      if (KIND == K_DSREAD) asm volatile("s_waitcnt lgkmcnt(0)");
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = sacc;
  for (int f = 0; f < 8; ++f) s += x[f];
  for (int f = 0; f < 4; ++f) s += px[f][0] + px[f][1];
  for (int r = 0; r < 16; ++r) s += acc[r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
static double base_ns[5];
template <int KIND, int N> void run(float* d, int occ, int iters, const char* what) {
  const int blocks = 256 * occ;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<KIND, N>), dim3(blocks), dim3(256), 0, 0, d, 10); hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<KIND, N>), dim3(blocks), dim3(256), 0, 0, d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double ns = ms * 1e6 / ((double)occ * iters * 16);
  if (KIND == K_NONE) base_ns[occ] = ns;
  printf("%d waves/SIMD  %-28s %6.1f ns per MFMA", occ, what, ns);
  if (KIND != K_NONE) printf("   = +%.1f cycles per filler (at the clock of the bare chain)", (ns - base_ns[occ]) / base_ns[occ] * 64 / N);
  printf("\n");
}
int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 3000;
  float* d; hipMalloc(&d, 1024 * 256 * 4);
  for (int occ = 1; occ <= 4; occ *= 4) {
    run<K_NONE, 0>(d, occ, iters, "bare MFMA chain");
    run<K_FMA, 5>(d, occ, iters, "5 x v_fma_f32");
    run<K_MUL, 5>(d, occ, iters, "5 x v_mul_f32");
    run<K_MAX, 5>(d, occ, iters, "5 x v_max_f32");
    run<K_PKFMA, 4>(d, occ, iters, "4 x v_pk_fma_f32 (8 fma)");
    run<K_PKMUL, 4>(d, occ, iters, "4 x v_pk_mul_f32 (8 mul)");
    run<K_EXP, 2>(d, occ, iters, "2 x v_exp_f32");
    run<K_SALU, 8>(d, occ, iters, "8 x s_add_u32");
    run<K_DSREAD, 2>(d, occ, iters, "2 x ds_read_b32 + wait");
  }
  return 0;
}
