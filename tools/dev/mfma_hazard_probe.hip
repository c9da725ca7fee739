// How many wait states does gfx950 need between v_mfma_f32_32x32x16_bf16 and a vector instruction that touches its registers --
// alone on a SIMD and with 2 / 4 waves per SIMD competing for the matrix pipe?  (VERDICT r3 next #1a: the bf16x6 attention
// kernel's order dependence.)  Every sequence is ONE inline-asm block on fixed registers, so the compiler neither reorders it nor
// adds wait states of its own; the spacing is exactly the `s_nop` written here.
//   RAW  : chain of CH MFMAs into v[64:79]; s_nop; VALU read of v79 and v64.        expected CH * 16 (A = B = 1.0)
//   WAR  : chain of CH MFMAs reading B = v[84:87]; s_nop; VALU overwrites B with 0; expected CH * 16
//   VRAW : VALU writes B (0 -> 1.0); s_nop; one MFMA.                               expected 16
//   CRAW : VALU writes the accumulator (v[64:79] = 3.0); s_nop; one MFMA with SrcC = it.   expected 19
// LLVM's table (GCNHazardRecognizer, gfx950 column, 8-pass XDL op): XDL write -> VALU read/write 12; the other three: 0.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/dev/mfma_hazard_probe.hip -o tools/dev/mfma_hazard_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

#define CLOB "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", \
             "v82", "v83", "v84", "v85", "v86", "v87"
#define ZERO_ACC "v_mov_b32 v64, 0\nv_mov_b32 v65, 0\nv_mov_b32 v66, 0\nv_mov_b32 v67, 0\nv_mov_b32 v68, 0\nv_mov_b32 v69, 0\nv_mov_b32 v70, 0\nv_mov_b32 v71, 0\n" \
                 "v_mov_b32 v72, 0\nv_mov_b32 v73, 0\nv_mov_b32 v74, 0\nv_mov_b32 v75, 0\nv_mov_b32 v76, 0\nv_mov_b32 v77, 0\nv_mov_b32 v78, 0\nv_mov_b32 v79, 0\n"
#define SET_A "v_mov_b32 v80, 0x3f803f80\nv_mov_b32 v81, 0x3f803f80\nv_mov_b32 v82, 0x3f803f80\nv_mov_b32 v83, 0x3f803f80\n"
#define SET_B "v_mov_b32 v84, 0x3f803f80\nv_mov_b32 v85, 0x3f803f80\nv_mov_b32 v86, 0x3f803f80\nv_mov_b32 v87, 0x3f803f80\n"
#define CLR_B "v_mov_b32 v84, 0\nv_mov_b32 v85, 0\nv_mov_b32 v86, 0\nv_mov_b32 v87, 0\n"
#define MFMA "v_mfma_f32_32x32x16_bf16 v[64:79], v[80:83], v[84:87], v[64:79]\n"
#define SETTLE "s_nop 15\ns_nop 15\n"

// W = wait states between the pair (0 = back to back; otherwise s_nop W-1)
template <int W>
struct Gap {
  static constexpr int K = W > 0 ? W - 1 : 0;
};
#define GAP_STR(W) ".if %[w] > 0\ns_nop %[k]\n.endif\n"

template <int MODE, int W, int CH>
__global__ void __launch_bounds__(1024) probe(float* out, int iters) {
  float s0 = 0.f, s1 = 0.f;
  const int wave = threadIdx.x >> 6;
  for (int i = 0; i < wave; ++i) __builtin_amdgcn_s_sleep(3);
  for (int it = 0; it < iters; ++it) {
    float r0, r1;
    if constexpr (MODE == 0) {          // RAW: XDL write -> VALU read
      asm volatile(ZERO_ACC SET_A SET_B SETTLE
                   ".rept %[ch]\n" MFMA ".endr\n" GAP_STR(W)
                   "v_mov_b32 %0, v79\nv_mov_b32 %1, v64\n" SETTLE
                   : "=v"(r0), "=v"(r1) : [w] "n"(W), [k] "n"(Gap<W>::K), [ch] "n"(CH) : CLOB);
    } else if constexpr (MODE == 1) {   // WAR: XDL SrcB read -> VALU write
      asm volatile(ZERO_ACC SET_A SET_B SETTLE
                   ".rept %[ch]\n" MFMA ".endr\n" GAP_STR(W) CLR_B SETTLE SETTLE
                   "v_mov_b32 %0, v79\nv_mov_b32 %1, v64\n"
                   : "=v"(r0), "=v"(r1) : [w] "n"(W), [k] "n"(Gap<W>::K), [ch] "n"(CH) : CLOB);
    } else if constexpr (MODE == 2) {   // VALU write -> XDL SrcB read
      asm volatile(ZERO_ACC SET_A CLR_B SETTLE
                   SET_B GAP_STR(W) MFMA SETTLE SETTLE
                   "v_mov_b32 %0, v79\nv_mov_b32 %1, v64\n"
                   : "=v"(r0), "=v"(r1) : [w] "n"(W), [k] "n"(Gap<W>::K), [ch] "n"(CH) : CLOB);
    } else {                            // VALU write -> XDL SrcC read
      asm volatile(ZERO_ACC SET_A SET_B SETTLE
                   "v_mov_b32 v64, 0x40400000\nv_mov_b32 v79, 0x40400000\n" GAP_STR(W) MFMA SETTLE SETTLE
                   "v_mov_b32 %0, v79\nv_mov_b32 %1, v64\n"
                   : "=v"(r0), "=v"(r1) : [w] "n"(W), [k] "n"(Gap<W>::K), [ch] "n"(CH) : CLOB);
    }
    s0 += r0;
    s1 += r1;
  }
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  out[2 * t] = s0;
  out[2 * t + 1] = s1;
}

template <int MODE, int W, int CH>
static void run(float* dev, std::vector<float>& host, const char* what, float expect) {
  const int iters = 2000;
  printf("%s W=%2d chain %2d :", what, W, CH);
  for (int wps : {1, 2, 4}) {          // waves per SIMD (blocks of 256 * wps threads, one block per CU)
    const int threads = 256 * wps, blocks = 256;
    hipLaunchKernelGGL((probe<MODE, W, CH>), dim3(blocks), dim3(threads), 0, nullptr, dev, iters);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(host.data(), dev, (size_t)blocks * threads * 2 * 4, hipMemcpyDeviceToHost));
    long bad = 0;
    int q[4] = {0, 0, 0, 0};
    for (long t = 0; t < (long)blocks * threads; ++t)
      if (host[2 * t] != expect * iters || host[2 * t + 1] != expect * iters) {
        ++bad;
        q[(t & 63) >> 4]++;
      }
    printf("   %d/SIMD: %7ld bad lanes [q0 %d q1 %d q2 %d q3 %d]", wps, bad, q[0], q[1], q[2], q[3]);
  }
  printf("\n");
}

int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  float* dev;
  CK(hipMalloc(&dev, (size_t)256 * 1024 * 2 * 4));
  std::vector<float> host((size_t)256 * 1024 * 2);
#define RAW(W) run<0, W, 1>(dev, host, "RAW  XDL write -> VALU read ", 16.f); run<0, W, 12>(dev, host, "RAW  XDL write -> VALU read ", 192.f);
  RAW(0) RAW(4) RAW(8) RAW(9) RAW(10) RAW(11) RAW(12) RAW(13) RAW(14) RAW(15) RAW(16)
#define WAR(W) run<1, W, 1>(dev, host, "WAR  XDL SrcB -> VALU write ", 16.f); run<1, W, 12>(dev, host, "WAR  XDL SrcB -> VALU write ", 192.f);
  WAR(0) WAR(1) WAR(2) WAR(3) WAR(4) WAR(6) WAR(8)
#define VRAW(W) run<2, W, 1>(dev, host, "VRAW VALU write -> XDL SrcB ", 16.f);
  VRAW(0) VRAW(1) VRAW(2) VRAW(3) VRAW(4)
#define CRAW(W) run<3, W, 1>(dev, host, "CRAW VALU write -> XDL SrcC ", 19.f);
  CRAW(0) CRAW(1) CRAW(2) CRAW(3) CRAW(4)
  return 0;
}
