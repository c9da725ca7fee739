// PROTOTYPE 4 (next-round work item 1 of DESIGN.md section 8; not part of libaot_hip.so).
// gemm_tile128.hip with a 128x64x32 workgroup tile: four waves of 64x32 (two MFMA blocks), 21 FLOP per operand byte, a
// 3-stage ring of 25 KB so that TWO workgroups share a CU (two waves per SIMD), six LDS-direct loads per wave and k-step.  A wave's 16 operand ds_read_b128 per step are issued in two halves (lgkmcnt is a 4-bit counter):
//   iteration s:  wait(vmcnt) ; barrier ; issue loads of step s+P ; read A(s+1) ; wait lgkmcnt(8) [all of step s is in
//                 registers] ; MFMAs 0..31 of step s ; read B(s+1) ; MFMAs 32..63 of step s ; epilogue if last k-step
// Measured on MI355X (end of round 1): correct on every shape; 16384x1024x1024 93.6 TF; 25440x128x512 and 25440x512x128 (Swin
// MLP) 49 us = 68 TF each; 25773x256x128 27.8 us (61 TF); 25773x256x64 16.7 us (shipped 19.5); 1674x256x1024 17.0 us (shipped
// 18.8); 6527x512x128 27.2 us with 102 tiles (needs split-K).  Two workgroups per CU recover what 128x128 loses to its single
// wave per SIMD, but ~93 TF is again the steady-state ceiling: next try 8-wave 256x128 tiles and a deeper ring.
// Build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 scratch/proto/gemm_tile128x64.hip -o scratch/proto/gemm_tile128x64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

#ifndef PF
#define PF 2                 // k-steps fetched ahead
#endif
#ifndef BLOCKS_PER_CU
#define BLOCKS_PER_CU 2
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int BM = 128, BN = 64, BK = 32, NSTAGE = PF + 1;
constexpr int GROUP_STRIDE = 8 * 128 + 16;
constexpr int OP_BYTES = 16 * GROUP_STRIDE;          // A: 128 rows = 16 groups of 8
constexpr int OPB_BYTES = 8 * GROUP_STRIDE;          // B: 64 rows
constexpr int STAGE_BYTES = OP_BYTES + OPB_BYTES;
static_assert(PF == 2, "prefetch distance (LDS: 3 stages of 25 KB, two workgroups per CU)");
static_assert(NSTAGE * STAGE_BYTES + 4096 <= 160 * 1024 / BLOCKS_PER_CU, "LDS budget");

__device__ __forceinline__ int chunk_off(int m, int c) {
  const int g = m >> 3, r = m & 7;
  return g * GROUP_STRIDE + r * 128 + ((c ^ r) << 4);
}

// wait until at most 6*n LDS-direct loads of this wave are still in flight (6 per k-step)
__device__ __forceinline__ void wait_steps_in_flight(int n) {
  switch (n) {
    case 0: __builtin_amdgcn_s_waitcnt(0xF70); break;
    case 1: __builtin_amdgcn_s_waitcnt(0xF76); break;            // 6
    default: __builtin_amdgcn_s_waitcnt(0xF7C); break;           // 12
  }
}

__global__ void __launch_bounds__(256) gemm_tile128x64_kernel(const float* __restrict__ A, const float* __restrict__ Bt,
                                                             const float* __restrict__ bias, float* __restrict__ C, int M,
                                                             int N, int K, int lda, int ldbt, int ldc, int relu) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[NSTAGE * STAGE_BYTES];
  __shared__ float sbias[1024];     // the whole bias vector (N <= 1024 in this prototype), read with asm ds_read in the epilogue:
                                    // a global load there would make the compiler drain the prefetch queue (vmcnt(0))
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int nbn = N / BN, nbm = (M + BM - 1) / BM, ntile = nbm * nbn;
  const int nk = K / BK;
  const int first = blockIdx.x, stride = gridDim.x;
  const int mine = first < ntile ? (ntile - first + stride - 1) / stride : 0;
  const int total = mine * nk;
  for (int i = tid; i < N; i += 256) sbias[i] = bias ? bias[i] : 0.f;
  __syncthreads();
  if (total == 0) return;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 32;
  const int lr = lane >> 3, lp = lane & 7;
  const int cofs = (lp ^ lr) << 2;

  // ---- issue side: walks (tile, k-step) independently of the compute side ----
  int is_tile = first, is_kt = 0, is_slot = 0, issued = 0;
  auto issue_next = [&]() {
    const int bm = is_tile / nbn, bn = is_tile - bm * nbn;
    unsigned char* sa = lds + is_slot * STAGE_BYTES;
    unsigned char* sb = sa + OP_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int g = 4 * wave + i;
      const int am = min(bm * BM + 8 * g + lr, M - 1);
      const float* ap = A + (long)am * lda + is_kt * BK + cofs;
      __builtin_amdgcn_global_load_lds((gptr_t)ap, (lptr_t)(sa + g * GROUP_STRIDE), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int g = 2 * wave + i;
      const float* bp = Bt + (long)(bn * BN + 8 * g + lr) * ldbt + is_kt * BK + cofs;
      __builtin_amdgcn_global_load_lds((gptr_t)bp, (lptr_t)(sb + g * GROUP_STRIDE), 16, 0, 0);
    }
    ++issued;
    if (++is_kt == nk) { is_kt = 0; is_tile += stride; }
    if (++is_slot == NSTAGE) is_slot = 0;
  };

  const unsigned lds_base = (unsigned)(size_t)(lptr_t)lds;
  unsigned aoff[2][4], boff[4];     // A: [32-row block of the wave tile][chunk j]; B: one 32-column block
#pragma unroll
  for (int j = 0; j < 4; ++j) {
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) aoff[h2][j] = lds_base + chunk_off(wm + 32 * h2 + l31, 2 * j + half);
    boff[j] = lds_base + OP_BYTES + chunk_off(wn + l31, 2 * j + half);
  }
  float4 ra[2][2][4], rb[2][4];     // [register set]([32-row block])[chunk j]
#define LDS_FETCH8(DST, OFF, SLOT)                                                                                   \
  {                                                                                                                  \
    const unsigned so_ = (unsigned)((SLOT) * STAGE_BYTES);                                                           \
    asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %9\n\tds_read_b128 %2, %10\n\tds_read_b128 %3, %11\n\t"     \
                 "ds_read_b128 %4, %12\n\tds_read_b128 %5, %13\n\tds_read_b128 %6, %14\n\tds_read_b128 %7, %15"         \
                 : "=&v"(DST[0][0]), "=&v"(DST[0][1]), "=&v"(DST[0][2]), "=&v"(DST[0][3]), "=&v"(DST[1][0]),         \
                   "=&v"(DST[1][1]), "=&v"(DST[1][2]), "=&v"(DST[1][3])                                              \
                 : "v"(OFF[0][0] + so_), "v"(OFF[0][1] + so_), "v"(OFF[0][2] + so_), "v"(OFF[0][3] + so_),           \
                   "v"(OFF[1][0] + so_), "v"(OFF[1][1] + so_), "v"(OFF[1][2] + so_), "v"(OFF[1][3] + so_));          \
  }
#define LDS_FETCH4(DST, OFF, SLOT)                                                                                   \
  {                                                                                                                  \
    const unsigned so_ = (unsigned)((SLOT) * STAGE_BYTES);                                                           \
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %7"            \
                 : "=&v"(DST[0]), "=&v"(DST[1]), "=&v"(DST[2]), "=&v"(DST[3])                                        \
                 : "v"(OFF[0] + so_), "v"(OFF[1] + so_), "v"(OFF[2] + so_), "v"(OFF[3] + so_));                      \
  }

  f32x16 acc[2];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[x][r] = 0.f;
  const unsigned sbias_base = (unsigned)(size_t)(lptr_t)sbias;

#pragma unroll 1
  for (int i = 0; i < PF && issued < total; ++i) issue_next();
  wait_steps_in_flight(min(PF - 1, total - 1));
  __builtin_amdgcn_s_barrier();
  LDS_FETCH8(ra[0], aoff, 0)
  LDS_FETCH4(rb[0], boff, 0)

  int c_tile = first, c_kt = 0, rd_slot = 1;      // rd_slot: ring slot of step s+1
#define MFMA_HALF(U, J0)                                                                                       \
  _Pragma("unroll") for (int j = J0; j < J0 + 2; ++j)                                                          \
  _Pragma("unroll") for (int x = 0; x < 2; ++x) {                                                              \
    const float4 a4 = ra[U][x][j], b4 = rb[U][j];                                                              \
    acc[x] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, acc[x], 0, 0, 0);                                \
    acc[x] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, acc[x], 0, 0, 0);                                \
    acc[x] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, acc[x], 0, 0, 0);                                \
    acc[x] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, acc[x], 0, 0, 0);                                \
  }
#pragma unroll 1
  for (int s = 0; s < total; s += 2) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {                 // unrolled by two so the register-set index is static
      const int ss = s + u;
      if (ss < total) {
        const bool more = ss + 1 < total;
        if (more) {
          wait_steps_in_flight(min(PF - 2, total - 2 - ss));   // step ss+1 has landed
          __builtin_amdgcn_s_barrier();                        // ... for every wave; slot of step ss-1 is free
          if (issued < total) issue_next();                    // step ss+P
          if (u == 0) LDS_FETCH8(ra[1], aoff, rd_slot) else LDS_FETCH8(ra[0], aoff, rd_slot)
          asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");   // everything of step ss is in registers
        } else {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        MFMA_HALF(u, 0)
        if (more) {
          if (u == 0) LDS_FETCH4(rb[1], boff, rd_slot) else LDS_FETCH4(rb[0], boff, rd_slot)
          if (++rd_slot == NSTAGE) rd_slot = 0;
        }
        MFMA_HALF(u, 2)
        if (++c_kt == nk) {       // tile finished: bias + relu + store
          const int bm = c_tile / nbn, bn = c_tile - bm * nbn;
          const int n = bn * BN + wn + l31;
          float bv;
          asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(bv) : "v"(sbias_base + (unsigned)n * 4u));
#pragma unroll
          for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int m = bm * BM + wm + 32 * x + (r & 3) + 8 * (r >> 2) + 4 * half;
              if (m < M) {
                float v = acc[x][r] + bv;
                if (relu) v = fmaxf(v, 0.f);
                C[(long)m * ldc + n] = v;
              }
              acc[x][r] = 0.f;
            }
          c_kt = 0;
          c_tile += stride;
        }
      }
    }
  }
#undef MFMA_HALF
#undef LDS_FETCH8
#undef LDS_FETCH4
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
  const Shape shapes[] = {{"l3.c3 256>1024", 1674, 256, 1024}, {"l2.c1 512>128", 6527, 512, 128}, {"l2.c3 128>512", 6527, 128, 512},
                          {"l1.c1 256>64", 25773, 256, 64},   {"l1.c3 64>256", 25773, 64, 256},   {"dec ad4 256>128", 25773, 256, 128},
                          {"swin mlp 128>512", 25440, 128, 512}, {"swin mlp 512>128", 25440, 512, 128}, {"ragged", 333, 96, 128},
                          {"big 16384x1024x1024", 16384, 1024, 1024}};
  printf("PF=%d BLOCKS_PER_CU=%d LDS=%d KB\n", PF, BLOCKS_PER_CU, NSTAGE * STAGE_BYTES / 1024);
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
    CK(hipMemset(dC, 0xFF, (size_t)M * N * 4));
    const int ntile = ((M + BM - 1) / BM) * (N / BN);
    const int grid = ntile < 256 * BLOCKS_PER_CU ? ntile : 256 * BLOCKS_PER_CU;
    hipLaunchKernelGGL(naive_kernel, dim3((N + 63) / 64, M), dim3(64), 0, 0, dA, dBt, db, dR, M, N, K, 1);
    hipLaunchKernelGGL(gemm_tile128x64_kernel, dim3(grid), dim3(256), 0, 0, dA, dBt, db, dC, M, N, K, K, K, N, 1);
    CK(hipDeviceSynchronize());
    std::vector<float> hC((size_t)M * N), hR((size_t)M * N);
    CK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hR.data(), dR, hR.size() * 4, hipMemcpyDeviceToHost));
    double err = 0;
    for (size_t i = 0; i < hC.size(); ++i) { double d = fabs((double)hC[i] - hR[i]); if (!(d <= err)) err = d; }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 30;
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i)
      hipLaunchKernelGGL(gemm_tile128x64_kernel, dim3(grid), dim3(256), 0, 0, dA, dBt, db, dC, M, N, K, K, K, N, 1);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps, gf = 2.0 * M * K * N / 1e9;
    printf("%-18s M=%6d K=%5d N=%5d grid=%4d  %7.1f us  %6.1f TFLOP/s   max|err| %.2e %s\n", s.name, M, K, N, grid, us,
           gf * 1e3 / us, err, err < 1e-3 ? "ok" : "MISMATCH");
    hipFree(dA); hipFree(dBt); hipFree(db); hipFree(dC); hipFree(dR);
  }
  return 0;
}
