// PROTOTYPE (next-round work item 1 of DESIGN.md section 8; not part of libaot_hip.so).
// fp32 GEMM  C[M][N] = relu(A[M][K] . B[K][N] + bias)  on v_mfma_f32_32x32x2_f32 with
//   * LDS-direct tile loads (global_load_lds_dwordx4, gfx950): no VGPR staging, no ds_write transposes;
//   * both operands K-CONTIGUOUS in LDS (A rows as they are; the weight is pre-packed transposed, Bt[N][K]) with an XOR
//     swizzle of the 16-byte chunks of a row, so one ds_read_b128 feeds four MFMA steps and is bank-conflict free;
//   * a 3-stage ring: the loads of k-step t+2 are issued before the MFMAs of step t, one barrier per step.
// The contraction index inside a 32-wide k-step is enumerated as k = 4*(2j + half) + i (j, i = 0..3; half = lane >> 5)
// for BOTH operands, which is all the MFMA needs.
// Measured on MI355X (end of round 1; shipped kernels in brackets): l3.c3 256>1024 15.5 us [18.8], l1.c1 256>64 15.8 [19.5],
// l2.c1 512>128 17.1 [19.0], l2.c3 128>512 18.8 [18.5], l1.c3 64>256 20.3 [20.7], dec ad4 256>128 29.4 [27.5], lstt 256>256 10.2
// [9.0], l3.c1 1024>256 28.6 [18.2: the shipped kernel splits K]; all results within 6e-5 of an fp64-accumulated reference.
// I.e. 10-20 % on the short-K shapes, but the ~55 TF ceiling stays: with K = 256 a tile is 8 k-steps (3.6 us of MFMA work)
// behind a cold first fetch and an epilogue -- the next step is a PERSISTENT grid that issues the next tile's first stages
// during the current tile's last k-steps and epilogue, plus split-K for the 424-tile shapes.
// Build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 scratch/proto/gemm_ldsdirect.hip -o scratch/proto/gemm_ldsdirect
// Run on the GPU box:  scratch/proto/gemm_ldsdirect      (self-checks against a naive kernel, prints TFLOP/s per shape)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int BM = 64, BN = 64, BK = 32, NSTAGE = 3;
constexpr int GROUP_BYTES = 8 * 128;                 // 8 rows x 32 floats: what one wave instruction moves
constexpr int GROUP_STRIDE = GROUP_BYTES + 16;       // consecutive 8-row groups are skewed by one 16-byte chunk (bank phase)
constexpr int OP_BYTES = 8 * GROUP_STRIDE;
constexpr int STAGE_BYTES = 2 * OP_BYTES;

// byte offset of 16-byte chunk c (0..7) of tile row m (0..63) inside an operand buffer
__device__ __forceinline__ int chunk_off(int m, int c) {
  const int g = m >> 3, r = m & 7;
  return g * GROUP_STRIDE + r * 128 + ((c ^ r) << 4);
}

__global__ void __launch_bounds__(256) gemm_ldsdirect_kernel(const float* __restrict__ A, const float* __restrict__ Bt,
                                                            const float* __restrict__ bias, float* __restrict__ C, int M,
                                                            int N, int K, int lda, int ldbt, int ldc, int relu) {
  // The operand reads are inline-asm ds_read_b128: hipcc's wait-count pass treats a C++ LDS load after LDS-direct loads as
  // possibly aliasing ALL of them and inserts s_waitcnt vmcnt(0) (seen in the ISA), which serialises the ring; the asm reads
  // are ordered by the explicit vmcnt wait + barrier below instead.
  __shared__ __attribute__((aligned(16))) unsigned char lds[NSTAGE * STAGE_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int nbn = N / BN;
  const int bm = blockIdx.x / nbn, bn = blockIdx.x - bm * nbn;
  const int m0 = bm * BM, n0 = bn * BN;
  const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;

  // loader: this wave moves row groups 2*wave and 2*wave+1 of both operands; lane -> (row r = lane>>3, slot p = lane&7),
  // and fetches the chunk that belongs in that slot after the swizzle: c = p ^ r
  const int lr = lane >> 3, lp = lane & 7;
  const float* a_src[2];
  const float* b_src[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int g = 2 * wave + i;
    const int am = min(m0 + 8 * g + lr, M - 1);
    a_src[i] = A + (long)am * lda + ((lp ^ lr) << 2);
    b_src[i] = Bt + (long)(n0 + 8 * g + lr) * ldbt + ((lp ^ lr) << 2);
  }
  auto issue = [&](int kt, int stage) {
    unsigned char* sa = lds + stage * STAGE_BYTES;
    unsigned char* sb = sa + OP_BYTES;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int g = 2 * wave + i;
      const int goff = g * GROUP_STRIDE;
      __builtin_amdgcn_global_load_lds((gptr_t)(a_src[i] + kt * BK), (lptr_t)(sa + goff), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(b_src[i] + kt * BK), (lptr_t)(sb + goff), 16, 0, 0);
    }
  };

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  const int nk = K / BK;
  const unsigned lds_base = (unsigned)(size_t)(lptr_t)lds;
  unsigned aoff[4], boff[4];      // byte offsets of this lane's four chunks (c = 2j + half) inside a stage
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    aoff[j] = lds_base + chunk_off(wm + l31, 2 * j + half);
    boff[j] = lds_base + OP_BYTES + chunk_off(wn + l31, 2 * j + half);
  }
  issue(0, 0);
  if (nk > 1) issue(1, 1);
  for (int kt = 0; kt < nk; ++kt) {
    // this wave's loads of step kt have landed when at most the 4 of step kt+1 are still in flight
    if (kt + 1 < nk) __builtin_amdgcn_s_waitcnt(0xF74); else __builtin_amdgcn_s_waitcnt(0xF70);
    __builtin_amdgcn_s_barrier();      // everyone's step-kt data visible; ring slot (kt+2)%3 (= step kt-1) is free
    if (kt + 2 < nk) issue(kt + 2, (kt + 2) % NSTAGE);
    const unsigned so = (unsigned)((kt % NSTAGE) * STAGE_BYTES);
    float4 a4[4], b4[4];
    asm volatile(
        "ds_read_b128 %0, %8\n\tds_read_b128 %4, %12\n\t"
        "ds_read_b128 %1, %9\n\tds_read_b128 %5, %13\n\t"
        "ds_read_b128 %2, %10\n\tds_read_b128 %6, %14\n\t"
        "ds_read_b128 %3, %11\n\tds_read_b128 %7, %15\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(a4[0]), "=&v"(a4[1]), "=&v"(a4[2]), "=&v"(a4[3]), "=&v"(b4[0]), "=&v"(b4[1]), "=&v"(b4[2]), "=&v"(b4[3])
        : "v"(aoff[0] + so), "v"(aoff[1] + so), "v"(aoff[2] + so), "v"(aoff[3] + so), "v"(boff[0] + so), "v"(boff[1] + so),
          "v"(boff[2] + so), "v"(boff[3] + so));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[j].x, b4[j].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[j].y, b4[j].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[j].z, b4[j].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[j].w, b4[j].w, acc, 0, 0, 0);
    }
  }
  // epilogue: C/D layout col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * half
  const int n = n0 + wn + l31;
  const float bv = bias ? bias[n] : 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = m0 + wm + (r & 3) + 8 * (r >> 2) + 4 * half;
    if (m < M) {
      float v = acc[r] + bv;
      if (relu) v = fmaxf(v, 0.f);
      C[(long)m * ldc + n] = v;
    }
  }
}

__global__ void naive_kernel(const float* A, const float* Bt, const float* bias, float* C, int M, int N, int K, int relu) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
  if (n >= N) return;
  double s = 0.0;
  for (int k = 0; k < K; ++k) s += (double)A[(long)m * K + k] * Bt[(long)n * K + k];
  float v = (float)s + bias[n];
  if (relu) v = fmaxf(v, 0.f);
  C[(long)m * N + n] = v;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

int main() {
  struct Shape { const char* name; int M, K, N; };
  const Shape shapes[] = {{"l3.c1 1024>256", 1674, 1024, 256}, {"l3.c3 256>1024", 1674, 256, 1024}, {"lstt 256>256", 1674, 256, 256},
                          {"l2.c1 512>128", 6527, 512, 128},  {"l2.c3 128>512", 6527, 128, 512},   {"l1.c1 256>64", 25773, 256, 64},
                          {"l1.c3 64>256", 25773, 64, 256},   {"dec ad4 256>128", 25773, 256, 128}, {"ragged", 333, 96, 128}};
  for (const Shape& s : shapes) {
    const int M = s.M, K = s.K, N = s.N;
    std::vector<float> hA((size_t)M * K), hBt((size_t)N * K), hb(N);
    unsigned seed = 12345u + M + K;
    auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return ((int)(seed >> 9) - (1 << 22)) * (1.f / (1 << 22)); };
    for (auto& v : hA) v = rnd();
    for (auto& v : hBt) v = rnd();
    for (auto& v : hb) v = rnd();
    float *dA, *dBt, *db, *dC, *dR;
    CK(hipMalloc(&dA, hA.size() * 4)); CK(hipMalloc(&dBt, hBt.size() * 4)); CK(hipMalloc(&db, N * 4));
    CK(hipMalloc(&dC, (size_t)M * N * 4)); CK(hipMalloc(&dR, (size_t)M * N * 4));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dBt, hBt.data(), hBt.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), N * 4, hipMemcpyHostToDevice));
    const int grid = ((M + BM - 1) / BM) * (N / BN);
    hipLaunchKernelGGL(naive_kernel, dim3((N + 63) / 64, M), dim3(64), 0, 0, dA, dBt, db, dR, M, N, K, 1);
    hipLaunchKernelGGL(gemm_ldsdirect_kernel, dim3(grid), dim3(256), 0, 0, dA, dBt, db, dC, M, N, K, K, K, N, 1);
    CK(hipDeviceSynchronize());
    std::vector<float> hC((size_t)M * N), hR((size_t)M * N);
    CK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hR.data(), dR, hR.size() * 4, hipMemcpyDeviceToHost));
    double err = 0;
    for (size_t i = 0; i < hC.size(); ++i) err = fmax(err, fabs((double)hC[i] - hR[i]));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 30;
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i)
      hipLaunchKernelGGL(gemm_ldsdirect_kernel, dim3(grid), dim3(256), 0, 0, dA, dBt, db, dC, M, N, K, K, K, N, 1);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps, gf = 2.0 * M * K * N / 1e9;
    printf("%-18s M=%6d K=%5d N=%5d  %7.1f us  %6.1f TFLOP/s   max|err| %.2e %s\n", s.name, M, K, N, us, gf * 1e3 / us, err,
           err < 1e-3 ? "ok" : "MISMATCH");
    hipFree(dA); hipFree(dBt); hipFree(db); hipFree(dC); hipFree(dR);
  }
  return 0;
}
