// Does gfx950 interlock a packed-fp32 (VOP3P) read of a register that a transcendental op (v_exp_f32) is still producing?
// (VERDICT r3 next #1a.  The failing instruction orders of the bf16x6 attention kernel -- tools/dev/x6_hazard.hip, variants
// "pad P->B" / "pad rescale": wrong O for the queries j >= 16 of a tile = lanes 16-31 and 48-63 -- have ONE thing in common that the
// passing orders lack: `alpha = v_exp_f32(m - mnew)` is followed within a few issue slots by `v_pk_mul_f32 o, o, alpha`, right
// behind a burst of 16-17 other v_exp_f32.  LLVM's gfx940 table asks for one wait state between a trans op and a VALU use.)
// Every sequence is one inline-asm block on fixed registers; the spacing is exactly the s_nop written here:
//     K x v_exp_f32 (independent: the backlog)  ;  v_exp_f32 v66, v64  ;  s_nop (W wait states)  ;  consumer of v66
// consumers: PK = v_pk_mul_f32 v[68:69], v[68:69], v[66:67] op_sel_hi:[1,0];  MUL = v_mul_f32 v68, v68, v66;
//            FMA = v_fma_f32 v68, v68, v66, 0;  MFMAB = v_mfma_f32_32x32x16_bf16 with v66 inside SrcB
// v66 holds a sentinel before; a lane that reads it too early multiplies by the sentinel instead of 2^x.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/dev/trans_hazard_probe.hip -o tools/dev/trans_hazard_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
#define CLOB "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", \
             "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", \
             "v100", "v101", "v102", "v103"
#define SETTLE "s_nop 15\ns_nop 15\n"
#define GAP ".if %[w] > 0\ns_nop %[k]\n.endif\n"

template <int W> struct Gap { static constexpr int K = W > 0 ? W - 1 : 0; };

// MODE 0 PK, 1 MUL, 2 FMA, 3 MFMA SrcB
template <int MODE, int K, int W>
__global__ void __launch_bounds__(1024) probe(float* out, int iters) {
  const int lane = threadIdx.x & 63;
  const float x = -(float)(lane % 7), xb = -(float)(lane % 5);
  float bad = 0.f;
  const int wave = threadIdx.x >> 6;
  for (int i = 0; i < wave; ++i) __builtin_amdgcn_s_sleep(3);
  for (int it = 0; it < iters; ++it) {
    float r0, r1;
    if constexpr (MODE < 3) {
      asm volatile("v_mov_b32 v64, %[x]\nv_mov_b32 v70, %[xb]\nv_mov_b32 v66, 0x42f60000\nv_mov_b32 v67, 0x42f60000\n"
                   "v_mov_b32 v68, 1.0\nv_mov_b32 v69, 1.0\n" SETTLE
                   ".rept %[kk]\nv_exp_f32 v72, v70\n.endr\n"
                   "v_exp_f32 v66, v64\n" GAP
                   ".if %[mode] == 0\nv_pk_mul_f32 v[68:69], v[68:69], v[66:67] op_sel_hi:[1,0]\n.endif\n"
                   ".if %[mode] == 1\nv_mul_f32 v68, v68, v66\nv_mul_f32 v69, v69, v66\n.endif\n"
                   ".if %[mode] == 2\nv_fma_f32 v68, v68, v66, 0\nv_fma_f32 v69, v69, v66, 0\n.endif\n" SETTLE
                   "v_mov_b32 %0, v68\nv_mov_b32 %1, v69\n"
                   : "=v"(r0), "=v"(r1) : [x] "v"(x), [xb] "v"(xb), [w] "n"(W), [k] "n"(Gap<W>::K), [kk] "n"(K), [mode] "n"(MODE) : CLOB);
      const float want = __builtin_amdgcn_exp2f(x);
      if (r0 != want || (MODE != 1 && MODE != 2 ? r1 != want : r1 != want)) bad += 1.f;
    } else {
      // A = 1.0 (bf16 pairs), B = v[84:87] with v84 = the trans result reinterpreted: use x such that 2^x has zero low 16 bits -> the
      // dword is [bf16 0 | bf16 2^x]: contributes 2^x once per lane-half... keep it simple: all four B dwords from v66
      asm volatile("v_mov_b32 v64, %[x]\nv_mov_b32 v70, %[xb]\nv_mov_b32 v66, 0x42f60000\n"
                   "v_mov_b32 v80, 0x3f803f80\nv_mov_b32 v81, 0x3f803f80\nv_mov_b32 v82, 0x3f803f80\nv_mov_b32 v83, 0x3f803f80\n"
                   "v_mov_b32 v85, 0\nv_mov_b32 v86, 0\nv_mov_b32 v87, 0\n"
                   ".irp r,88,89,90,91,92,93,94,95,96,97,98,99,100,101,102,103\nv_mov_b32 v\\r, 0\n.endr\n" SETTLE
                   ".rept %[kk]\nv_exp_f32 v72, v70\n.endr\n"
                   "v_exp_f32 v66, v64\n" GAP
                   "v_mov_b32 v84, v66\n"          // (one VALU hop, as the P split has between exp and MFMA)
                   "s_nop 1\n"
                   "v_mfma_f32_32x32x16_bf16 v[88:103], v[80:83], v[84:87], v[88:103]\n" SETTLE SETTLE
                   "v_mov_b32 %0, v88\nv_mov_b32 %1, v103\n"
                   : "=v"(r0), "=v"(r1) : [x] "v"(x), [xb] "v"(xb), [w] "n"(W), [k] "n"(Gap<W>::K), [kk] "n"(K) : CLOB);
      // column j = lane & 31 of B: k = 0 (+8 for the upper half) holds bf16(0) in the low half, 2^x in the high half of v84 of lanes
      // j and j + 32: every output of column j = 2^x(lane j) + 2^x(lane j + 32)
      const float want = __builtin_amdgcn_exp2f(-(float)((lane & 31) % 7)) + __builtin_amdgcn_exp2f(-(float)(((lane & 31) + 32) % 7));
      if (r0 != want || r1 != want) bad += 1.f;
    }
  }
  out[(long)blockIdx.x * blockDim.x + threadIdx.x] = bad;
}

template <int MODE, int K, int W>
static void run(float* dev, std::vector<float>& host, const char* what) {
  const int iters = 500;
  printf("%s backlog %2d, %d wait states:", what, K, W);
  for (int wps : {1, 2, 4}) {
    const int threads = 256 * wps, blocks = 256;
    hipLaunchKernelGGL((probe<MODE, K, W>), dim3(blocks), dim3(threads), 0, nullptr, dev, iters);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(host.data(), dev, (size_t)blocks * threads * 4, hipMemcpyDeviceToHost));
    long q[4] = {0, 0, 0, 0};
    for (long t = 0; t < (long)blocks * threads; ++t) q[(t & 63) >> 4] += (long)host[t];
    printf("   %d/SIMD: bad [lanes 0-15: %ld  16-31: %ld  32-47: %ld  48-63: %ld]", wps, q[0], q[1], q[2], q[3]);
  }
  printf("\n");
}

int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  float* dev;
  CK(hipMalloc(&dev, (size_t)256 * 1024 * 4));
  std::vector<float> host((size_t)256 * 1024);
#define ROW(MODE, NAME, K) run<MODE, K, 0>(dev, host, NAME); run<MODE, K, 1>(dev, host, NAME); run<MODE, K, 2>(dev, host, NAME); \
  run<MODE, K, 3>(dev, host, NAME); run<MODE, K, 4>(dev, host, NAME); run<MODE, K, 6>(dev, host, NAME); run<MODE, K, 8>(dev, host, NAME); \
  run<MODE, K, 12>(dev, host, NAME); run<MODE, K, 16>(dev, host, NAME);
  ROW(0, "v_exp -> v_pk_mul_f32 ", 0) ROW(0, "v_exp -> v_pk_mul_f32 ", 4) ROW(0, "v_exp -> v_pk_mul_f32 ", 16)
  ROW(1, "v_exp -> v_mul_f32    ", 0) ROW(1, "v_exp -> v_mul_f32    ", 16)
  ROW(2, "v_exp -> v_fma_f32    ", 0) ROW(2, "v_exp -> v_fma_f32    ", 16)
  ROW(3, "v_exp -> v_mov -> MFMA", 0) ROW(3, "v_exp -> v_mov -> MFMA", 16)
  return 0;
}
